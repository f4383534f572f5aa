#!/usr/bin/env python3
"""Round-6 fixture for BASELINE.json configs[3], produced by running the REAL reference in the build container (rules as in
gen_golden.py: the reference is imported in place, nothing of it is copied; only seeds and output tensors are written).

    python tests/golden/gen_golden_r6b.py            (about ten minutes and 20 GB on 8 cores)
    python tests/golden/gen_golden_r6b.py cut        (only the branch-cut distances, merged into the existing file)
    python tests/golden/gen_golden_r6b.py self 4     (the reference once more at 4 torch threads, against the 8-thread run in the file)
    python tests/golden/gen_golden_r6b.py self 8 nomkldnn   (... at 8 threads with torch.backends.mkldnn off: another evaluation order of its STFT)

  e2e1800_r6.npz   the 1800 s, 7-channel meeting of configs[3] (synth_meeting(1800, 7, seed=1), the weights and head biases of
                   configs[1], no re-calibration -- SURVEY.md 8(d)) through the reference's separate_and_stitch: 1209 segments,
                   T_long 112 499.  Until round 6 this configuration was held to the oracle only.  Kept: every permutation,
                   both activity maps (bits), the stitched masks every 32nd bin / 16th frame, every segment's four masks on a
                   9 x 8 grid (which segments agree at the mask level), per segment the number of winner-take-all decisions
                   with a top-2 margin below 2e-5 / 1e-5 (the decisions a rounding-level difference can flip) and the
                   three waveforms every 128th sample and their RMS; and per segment how close its IPD features come to the
                   atan2 branch cut (oracle features, as gen_golden_r5.py screens with: DESIGN.md hazard 7 -- two correct float32
                   evaluations of an angle at +-pi land 2 pi apart, and the reference's masks of such a segment are its own).
                   `self N`: the reference AGAINST ITSELF at another thread count -- per segment the largest difference of its
                   own grid masks, per hop block the distance of its own waveforms, whole-meeting relative RMS: what "agrees with
                   the reference" can mean on this meeting.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up sys.path / stubs, imports the reference as G.RC)

import torch  # noqa: E402

RC, W, SYN, O = G.RC, G.W, G.SYN, G.O


def _cut_one(args):
    x, i = args
    f = O.features(x)[257:].reshape(6, 257, -1)[:, 1:256]
    return i, float(np.abs(np.abs(f) - np.pi).min())


def cut_distances(mix, workers=8):
    """per segment: min over channel pairs, bins 1..255 and frames of | |IPD| - pi | (oracle features of the zero-padded segment)"""
    import multiprocessing as mp
    X = O.stft(mix[0])                     # [F, T_long, C]
    TL = X.shape[1]
    nseg = int(np.ceil((TL - 93) / 93))
    segs = []
    for i in range(nseg):
        seg = np.zeros((257, 186, X.shape[2]), np.complex64)
        t = min(186, TL - 93 * i)
        seg[:, :t] = X[:, 93 * i:93 * i + t]
        segs.append((seg, i))
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_cut_one, segs, chunksize=8)
    out = np.zeros(nseg)
    for i, d in res:
        out[i] = d
    return out


class LightTap(G.Tap):
    """Tap without the beamformer's operands (3.2 GB of copies at 1209 segments): masks and permutations only, and the masks
    reduced at once to what the fixture keeps."""

    def __enter__(self):
        tap = self
        tap.grid, tap.low2, tap.low1, tap.minmargin = [], [], [], []

        def sep(stft):
            out = tap._orig_sep(stft)
            m = np.concatenate([out["spk_masks"][0].detach().cpu().numpy(), out["noise_masks"][0].detach().cpu().numpy()], -1)   # [F, T, 4]
            tap.grid.append(m[::32, ::24].copy())
            srt = np.sort(m, axis=-1)
            margin = srt[..., -1] - srt[..., -2]
            tap.low2.append(int((margin < 2e-5).sum()))
            tap.low1.append(int((margin < 1e-5).sum()))
            tap.minmargin.append(float(margin.min()))
            return out

        def pit(self_, preds, targets):
            loss, perms = tap._orig_pit(self_, preds, targets)
            tap.pit.append((float(loss[0]), tuple(int(x) for x in perms[0])))
            return loss, perms

        self.model.separate = sep
        RC.PitWrapper.forward = pit
        return self


def main():
    torch.manual_seed(0)
    path_npz = os.path.join(HERE, "e2e1800_r6.npz")
    if len(sys.argv) > 1 and sys.argv[1] == "cut":
        e = dict(np.load(path_npz))
        e["cut_distance_per_segment"] = cut_distances(SYN.synth_meeting(1800.0, 7, seed=1))
        np.savez_compressed(path_npz, **e)
        print("cut distances:", np.sort(e["cut_distance_per_segment"])[:12], flush=True)
        return
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(HERE, "calib_mc.npz"))
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"])
    model = G.build_reference_model(desc, st)
    cfg = RC.CssCfg(show_progressbar=False, activity_th=0.3)
    t0 = time.time()
    mix = SYN.synth_meeting(1800.0, 7, seed=1)
    print(f"meeting generated in {time.time() - t0:.0f} s: {mix.shape}", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "self":
        nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4
        torch.set_num_threads(nt)
        if len(sys.argv) > 3 and sys.argv[3] == "nomkldnn":     # the reference's conv1d / Linear on ATen's own kernels instead of oneDNN's
            torch.backends.mkldnn.enabled = False
            nt = f"{nt}_nomkldnn"
        e = dict(np.load(path_npz))
        with LightTap(model) as tap:
            t0 = time.time()
            wavs, side = RC.separate_and_stitch(mix, model, 16000, torch.device("cpu"), cfg)
            dt = time.time() - t0
        w = np.stack(wavs)[:, ::128]
        grid = np.stack(tap.grid).astype(np.float32)
        e[f"self_t{nt}_grid_max_abs"] = np.abs(grid - e["masks_grid"]).reshape(len(grid), -1).max(axis=1)
        nb = w.shape[1] // 186
        d = (w[:, :nb * 186].astype(np.float64) - e["wav_dec128"][:, :nb * 186]).reshape(3, nb, 186)
        e[f"self_t{nt}_hop_block_rel"] = (np.sqrt((d ** 2).mean(axis=2)) / e["wav_rms"][:, None]).astype(np.float32)
        e[f"self_t{nt}_wav_rel_rms"] = np.array([G.rel_rms(w[k], e["wav_dec128"][k]) for k in range(3)])
        e[f"self_t{nt}_perms_differ"] = int((np.array([p for _, p in tap.pit], np.int32) != e["pit_perm"]).any(axis=1).sum())
        e[f"self_t{nt}_activity_bits_differ"] = int((np.packbits(side["activity_b"].numpy()) != e["activity_b"]).sum())
        np.savez_compressed(path_npz, **e)
        rep = {"threads": nt, "reference_wall_s": dt, "segments_whose_own_grid_masks_differ_beyond_5e-6": int((e[f"self_t{nt}_grid_max_abs"] > 5e-6).sum()),
               "grid_masks_max_abs": float(e[f"self_t{nt}_grid_max_abs"].max()), "wav_rel_rms": [float(x) for x in e[f"self_t{nt}_wav_rel_rms"]],
               "hop_blocks_within_1e-4": [int((e[f"self_t{nt}_hop_block_rel"][k] < 1e-4).sum()) for k in range(3)], "hop_blocks": int(nb)}
        path = os.path.join(HERE, "golden_report_r6.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old[f"e2e1800_r6_self_t{nt}"] = rep
        json.dump(old, open(path, "w"), indent=1)
        print(rep, flush=True)
        return
    cut = cut_distances(mix)
    with LightTap(model) as tap:
        t0 = time.time()
        wavs, side = RC.separate_and_stitch(mix, model, 16000, torch.device("cpu"), cfg)
        dt = time.time() - t0
    print(f"reference: {dt:.0f} s for 1800 s of audio = {1800.0 / dt:.2f} x real time, {len(tap.grid)} segments", flush=True)
    w = np.stack(wavs)
    e = {"mix_seed": 1, "mix_seconds": 1800.0, "num_segments": len(tap.grid),
         "pit_perm": np.array([p for _, p in tap.pit], np.int32),
         "pit_loss": np.array([l for l, _ in tap.pit], np.float64),
         "activity_final": np.packbits(side["activity_final"].numpy()[0]),
         "activity_b": np.packbits(side["activity_b"].numpy()),
         "activity_shape": np.array(side["activity_b"].shape),
         "mask_stitched": side["mask_stitched"].numpy()[0, ::32, ::16].copy(),
         "masks_grid": np.stack(tap.grid).astype(np.float32),              # [1209, 9, 8, 4]
         "wta_margin_below_2e-5_per_segment": np.array(tap.low2, np.int32),
         "wta_margin_below_1e-5_per_segment": np.array(tap.low1, np.int32),
         "wta_margin_min_per_segment": np.array(tap.minmargin, np.float64),
         "cut_distance_per_segment": cut,
         "wav_dec128": w[:, ::128].copy(),
         "wav_rms": np.array([np.sqrt(np.mean(x.astype(np.float64) ** 2)) for x in w]),
         "wav_len": w.shape[1]}
    np.savez_compressed(os.path.join(HERE, "e2e1800_r6.npz"), **e)
    rep = {"reference_wall_s": dt, "x_real_time": 1800.0 / dt, "threads": torch.get_num_threads(), "segments": len(tap.grid),
           "perms_non_identity": int((e["pit_perm"] != np.arange(3)).any(axis=1).sum()),
           "activity_final_open_fraction": float(side["activity_final"].numpy().mean()),
           "wta_margin_below_2e-5": int(sum(tap.low2)), "wta_margin_below_1e-5": int(sum(tap.low1)),
           "segments_with_a_margin_below_1e-5": int((np.array(tap.low1) > 0).sum()),
           "wav_rms": [float(x) for x in e["wav_rms"]], "bytes": os.path.getsize(os.path.join(HERE, "e2e1800_r6.npz"))}
    path = os.path.join(HERE, "golden_report_r6.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old["e2e1800_r6"] = rep
    json.dump(old, open(path, "w"), indent=1)
    print(rep, flush=True)


if __name__ == "__main__":
    main()
