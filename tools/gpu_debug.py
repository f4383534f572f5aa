#!/usr/bin/env python3
"""First-contact / debugging script for the GPU box: runs the HIP path stage by stage on the 2-segment
stage input and prints its deviation from the oracle at every stage boundary.

    gpurun -- 'python tools/gpu_debug.py > gpurun_out/debug.log 2>&1'
"""
import importlib
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import css_oracle as O  # noqa: E402

P = importlib.import_module("notsofar1_challenge_amd")
W = importlib.import_module("notsofar1_challenge_amd.weights")
SYN = importlib.import_module("notsofar1_challenge_amd.synth")
L = importlib.import_module("notsofar1_challenge_amd._lib")
CSS = importlib.import_module("notsofar1_challenge_amd.css")
SEP = importlib.import_module("notsofar1_challenge_amd.separator")


def rel_rms(a, b):
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / (np.sqrt(np.mean(np.abs(b) ** 2)) + 1e-300))


def step(name):
    print(f"\n=== {name}", flush=True)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    g = np.load(os.path.join(ROOT, "tests/golden/stage_mc.npz"))
    cal = np.load(os.path.join(ROOT, "tests/golden/calib_mc.npz"))
    desc = W.ModelDesc.mc_v1()
    st = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
    if secs > 0:
        mix = SYN.synth_meeting(60.0, 7, seed=1)[:, :int(secs * 16000)]
    else:
        mix = SYN.synth_meeting(float(g["mix_seconds"]), 7, seed=int(g["mix_seed"]))[:, :int(g["n_samples"])]
    print("mix", mix.shape, "devices", L.load().css_device_count())
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
    ocfg = O.OracleCssCfg(activity_th=0.3)
    params = O.ConformerParams(st)
    sep = SEP.HipSeparator(st, None, device=0)
    h = sep.handle
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7)
    T, F, S = 186, 257, 3

    step("fused run")
    t0 = time.time()
    wav = h.run(mix[0], run_cfg)
    print("first run %.3f s" % (time.time() - t0), h.timings())
    t0 = time.time()
    wav = h.run(mix[0], run_cfg)
    print("second run %.3f s" % (time.time() - t0), h.timings())
    plan = h.get_plan()
    nseg = plan.num_segments
    print("plan", plan.stft_frames, plan.mix_frames, nseg, plan.n_out, plan.last_valid)

    step("STFT")
    X = h.read(L.BUF_X)  # [C, 2F, T_ld]
    Xc = (X[:, :F] + 1j * X[:, F:])[:, :, :plan.stft_frames]  # [C, F, T]
    Xo = O.stft(mix[0])  # [F, T, C]
    print("stft relrms", rel_rms(np.moveaxis(Xc, 0, 2), Xo), "Im(DC) exactly 0:", bool((X[:, F, :plan.stft_frames] == 0).all()),
          "Im(Nyq) exactly 0:", bool((X[:, 2 * F - 1, :plan.stft_frames] == 0).all()))

    step("masks (HIP end-to-end vs oracle end-to-end)")
    M = h.read(L.BUF_MASKS).reshape(S + 1, F, nseg, T)
    oplan = O.make_plan(mix.shape[1], 16000, ocfg)
    assert oplan.num_segments == nseg
    hip_masks = []
    for i in range(nseg):
        s0, e0, tv = oplan.seg_range(i)
        seg = np.zeros((F, T, 7), np.complex64)
        seg[:, :tv] = Xo[:, s0:e0]
        if i < 3:
            taps = {}
            feat = O.features(seg)
            om = O.conformer_forward(params, feat, taps=taps)
            d = np.abs(M[:, :, i, :] - om)
            flips = int((np.argmax(M[:, :, i, :], 0) != np.argmax(om, 0)).sum())
            print(f"seg {i}: masks maxabs {d.max():.3e} mean {d.mean():.3e} wta flips {flips}")
        hip_masks.append((np.ascontiguousarray(np.moveaxis(M[:S, :, i, :], 0, 2)),
                          np.ascontiguousarray(np.moveaxis(M[S:, :, i, :], 0, 2))))

    step("features / hidden of the last batch (segment 0 rows)")
    try:
        feat_h = h.read(L.BUF_FEATURES)  # [tokens, Kp]
        seg = np.zeros((F, T, 7), np.complex64)
        s0, e0, tv = oplan.seg_range(0)
        seg[:, :tv] = Xo[:, s0:e0]
        fo = O.features(seg).T  # [T, 1799]
        fo = (fo + params("input_bias").reshape(-1)) * params("input_scale").reshape(-1)
        if nseg <= 64:
            d = np.abs(feat_h[:T, :1799] - fo)
            print("features maxabs %.3e p99 %.3e, pad cols zero: %s, n(|d|>1): %d" % (
                d.max(), np.percentile(d, 99), bool((feat_h[:, 1799:] == 0).all()), int((d > 1).sum())))
            bad = np.argwhere(d > 1)
            print("first big diffs (t, col):", bad[:8].tolist())
            taps = {}
            O.conformer_forward(params, O.features(seg), taps=taps)
            hid = h.read(L.BUF_HIDDEN)
            print("hidden (after last block) maxabs %.3e" % np.abs(hid[:T] - taps["block17"]).max())
    except Exception:
        traceback.print_exc()

    step("downstream of the masks: oracle driven by HIP masks (float64 MVDR)")
    try:
        ow, oside = O.separate_and_stitch(mix, params, 16000, ocfg, separate_fn=lambda i, seg: hip_masks[i],
                                          mvdr_cplx=np.complex128)
        for k in range(S):
            print(f"stream {k}: wav relrms vs oracle(HIP masks, c128) {rel_rms(wav[k], ow[k]):.3e}")
        perms = h.read(L.BUF_PERMS)
        print("perms equal:", [tuple(p) for p in perms] == [tuple(p) for p in oside["perms"]])
        costs = h.read(L.BUF_PIT_COST)
        print("pit cost maxabs diff:", max(np.abs(costs[b].reshape(S, S) - oside["pit_costs"][b]).max() for b in range(nseg - 1)))
        mst = h.read(L.BUF_MASK_ST)
        print("mask_stitched maxabs", np.abs(np.transpose(mst, (1, 2, 0)) - oside["mask_stitched"][0]).max())
        act = h.read(L.BUF_ACTIVITY)
        print("activity maxabs", np.abs(act.T - oside["activity"]).max())
        print("act_b equal", np.array_equal(h.read(L.BUF_ACT_B).astype(bool).T, oside["activity_b"]),
              "act_final equal", np.array_equal(h.read(L.BUF_ACT_FINAL).astype(bool).T, oside["activity_final"][0]))
        ow64, _ = O.separate_and_stitch(mix, params, 16000, ocfg, separate_fn=lambda i, seg: hip_masks[i])
        for k in range(S):
            print(f"stream {k}: wav relrms vs oracle(HIP masks, c64) {rel_rms(wav[k], ow64[k]):.3e}")
    except Exception:
        traceback.print_exc()

    step("full oracle end to end")
    try:
        ow, oside = O.separate_and_stitch(mix, params, 16000, ocfg)
        for k in range(S):
            print(f"stream {k}: wav relrms vs oracle {rel_rms(wav[k], ow[k]):.3e}  (len {len(wav[k])} vs {len(ow[k])})")
    except Exception:
        traceback.print_exc()

    step("separator protocol: stft / separate / istft")
    try:
        import torch
        xs = sep.stft(torch.from_numpy(mix))
        print("protocol stft", tuple(xs.shape), rel_rms(xs.numpy()[0], Xo))
        segt = xs[:, :, :T]
        ms = sep.separate(segt)
        print("protocol separate", tuple(ms["spk_masks"].shape),
              np.abs(ms["spk_masks"].numpy()[0] - hip_masks[0][0]).max())
        rs = np.random.RandomState(0)
        y = (rs.randn(2, F, 50) + 1j * rs.randn(2, F, 50)).astype(np.complex64)
        wi = sep.istft(torch.from_numpy(y)).numpy()
        print("protocol istft relrms", rel_rms(wi, O.istft(y)))
    except Exception:
        traceback.print_exc()


if __name__ == "__main__":
    main()
