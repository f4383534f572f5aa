#!/usr/bin/env python3
"""Round-4 golden fixtures: the REAL reference in the regime a trained separator works in (see gen_golden.py for the
rules: the reference is imported in place, nothing of it is copied; only seeds and output tensors are written).

    python tests/golden/gen_golden_r4.py       # writes realistic_r4.npz, session_triple_r4.json (+ _pcm16.npz), golden_report_r4.json

  realistic_r4.npz   css/css.py:110 separate_and_stitch driven by a separator-protocol object (css.py:131,199) whose
                     ``separate()`` returns IDEAL RATIO MASKS of a 60 s turn-taking conversation of speech-like talkers
                     (tests/irm_separator.py, synth.synth_conversation): sharp, sparse masks saturated to exactly 0 / 1,
                     talkers silent through whole segments, exact ties between winning masks, speaker order shuffled per
                     segment, the activity gate at the shipped threshold 0.3 toggling by itself.  Multi-channel (MVDR) and
                     single-channel (channel 0, mask multiplication).  Three reference runs of the MC case:
                       c64    the reference as it is (8 threads; a 1-thread run is compared in the report);
                       c128   the reference with ONE change: make_mvdr is handed ``mix_stft.astype(complex128)`` -- the same
                              code (mvdr_util.py:5-80) evaluated in double precision;
                     and per (segment, raw stream) the distance between the two: where the reference's complex64 solve does
                     not reproduce its own complex128 evaluation, no implementation can be held to it.
  session_triple_r4  css_inference (css.py:51-107) on a 19.4 s FULL-SCALE session of the well-conditioned config-2 meeting
                     (PCM16 files in, PCM16 files out through the reference's own load_audio / write_wav), replacing the
                     quiet 5 s clip of round 2 whose comparison needed a 2e-2 bar.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as G  # noqa: E402  (sets up sys.path / stubs, imports the reference as G.RC)
import gen_golden_r2 as G2  # noqa: E402
import irm_separator as IRM  # noqa: E402

import torch  # noqa: E402

RC, W, SYN, O = G.RC, G.W, G.SYN, G.O
CONV_SECONDS, CONV_SEED = 60.0, 11
DEC = 16            # waveforms are stored as every 16th sample
# a (segment, stream) whose complex64 and complex128 evaluations agree to this is "reproduced by the reference itself".  The
# distances are bimodal on this meeting: 100 of the 120 lie in [7e-8, 3.5e-4], the other 20 in [2.3e-2, 3.8] -- a talker
# who is silent (or nearly) in the segment: target covariance ~ 1e-10 x the mixture's, the complex64 solve returns noise
REPRO_TAU = 1e-3


def run_protocol(model, masks, mix, cfg, c128=False, threads=None):
    """the reference's driver with the ideal-mask separator; ``c128``: make_mvdr sees a complex128 mixture"""
    if threads:
        torch.set_num_threads(threads)
    sep = IRM.IdealMaskSeparator(masks, model.stft, model.istft)
    orig = RC.make_mvdr
    if c128:
        def mv(spk, noise, mix_wav=None, mix_stft=None, return_stft=False):
            return orig(spk, noise, mix_wav=mix_wav, mix_stft=mix_stft.astype(np.complex128), return_stft=return_stft)
        RC.make_mvdr = mv
    try:
        with G.Tap(sep) as tap:
            wavs, side = RC.separate_and_stitch(mix, sep, 16000, torch.device("cpu"), cfg)
    finally:
        RC.make_mvdr = orig
    return wavs, side, tap


def decisions(prefix, side, tap):
    return {
        f"{prefix}_pit_perm": np.array([p for _, p in tap.pit], np.int32).reshape(-1, 3),
        f"{prefix}_activity_b": np.packbits(side["activity_b"].numpy()),
        f"{prefix}_activity_final": np.packbits(side["activity_final"].numpy()[0]),
        f"{prefix}_activity_shape": np.array(side["activity_b"].shape),
    }


def main():
    torch.manual_seed(0)
    nthr = torch.get_num_threads()
    report = {"torch": torch.__version__, "numpy": np.__version__, "threads": nthr}
    # the transforms of the reference's own wrapper (a one-block model: its estimator is never called here)
    desc1 = W.ModelDesc(num_blocks=1)
    shell = G.build_reference_model(desc1, W.portable_state_dict(desc1, 0))
    desc1s = W.ModelDesc(num_mics=1, in_features=257, num_blocks=1)
    shell_sc = G.build_reference_model(desc1s, W.portable_state_dict(desc1s, 0))

    # ------------------------------------------------------------------ realistic masks, 60 s
    mix, images = SYN.synth_conversation(CONV_SECONDS, 7, seed=CONV_SEED, return_sources=True)
    masks = IRM.IdealMasks(images)
    cfg = RC.CssCfg(show_progressbar=False, activity_th=0.3)
    w64, s64, t64 = run_protocol(shell, masks, mix, cfg, threads=nthr)
    w64_1, s64_1, t64_1 = run_protocol(shell, masks, mix, cfg, threads=1)
    w128, s128, t128 = run_protocol(shell, masks, mix, cfg, c128=True, threads=nthr)
    nseg = len(t64.masks)
    out = {"mix_seconds": CONV_SECONDS, "mix_seed": CONV_SEED, "num_segments": nseg, "dec": DEC,
           "masks_sha256": masks.sha256(nseg), "repro_tau": REPRO_TAU}
    stats = masks.statistics(nseg)
    report["mask_statistics"] = stats
    out.update(decisions("mc", s64, t64))
    assert [p for _, p in t64.pit] == [p for _, p in t128.pit] == [p for _, p in t64_1.pit]
    assert bool((s64["activity_final"] == s128["activity_final"]).all()) and bool((s64["activity_b"] == s128["activity_b"]).all())
    # the reference against itself
    d = np.array([[G.rel_rms(a[k], b[k]) for k in range(3)] for a, b in zip(t64.mvdr_out, t128.mvdr_out)])   # [nseg, raw stream]
    out["mc_c64_vs_c128_per_segment"] = d
    report["ref_8thr_vs_1thr_relrms"] = [G.rel_rms(w64[k], w64_1[k]) for k in range(3)]
    report["ref_c64_vs_c128_relrms_whole_streams"] = [G.rel_rms(w64[k], w128[k]) for k in range(3)]
    report["segment_streams_reproduced"] = int((d <= REPRO_TAU).sum())
    report["segment_streams_total"] = int(d.size)
    out["mc_wav_c64"] = np.stack(w64)[:, ::DEC]
    out["mc_wav_c128"] = np.stack(w128)[:, ::DEC]
    out["mc_wav_len"] = len(w64[0])
    out["mc_mask_stitched"] = s64["mask_stitched"].numpy()[0, ::16, ::8]
    af = s64["activity_final"].numpy()[0]
    report["gate_open_fraction"] = af.mean(0).tolist()
    report["gate_toggles"] = [int(np.abs(np.diff(af[:, k].astype(int))).sum()) for k in range(3)]
    perms = [p for _, p in t64.pit]
    report["distinct_stitching_permutations"] = len(set(perms))
    act = s64["mask_stitched"].numpy().mean(axis=1)[0]
    report["activity_min_dist_to_th"] = float(np.min(np.abs(act - np.float32(cfg.activity_th))))
    # the oracle on the same masks: decisions, and both evaluations
    sep_fn = lambda i, seg: masks.segment(i)
    ow, oside = O.separate_and_stitch(mix, None, 16000, O.OracleCssCfg(activity_th=0.3), separate_fn=sep_fn, mvdr_cplx=np.complex128)
    report["oracle_perms_equal"] = [tuple(p) for p in oside["perms"][1:]] == perms
    report["oracle_activity_equal"] = bool(np.array_equal(oside["activity_final"], s64["activity_final"].numpy())
                                           and np.array_equal(oside["activity_b"], s64["activity_b"].numpy()))
    report["oracle_f64_vs_ref_c128_relrms"] = [G.rel_rms(ow[k], w128[k]) for k in range(3)]
    # exact stitching-cost ties: boundaries where scipy's rule, not the optimum, decides
    ties = 0
    for c, lp in zip(oside["pit_costs"], oside["perms"][:-1]):
        import itertools
        m = c[list(lp)]
        tot = sorted(sum(m[a, p[a]] for a in range(3)) for p in itertools.permutations(range(3)))
        ties += int(tot[1] == tot[0])
    report["boundaries_with_exactly_tied_optimum"] = ties

    # ------------------------------------------------------------------ same masks, single channel (mask multiplication)
    mix1 = np.ascontiguousarray(mix[:, :, :1])
    ws, ss, ts = run_protocol(shell_sc, masks, mix1, cfg, threads=nthr)
    out.update(decisions("sc", ss, ts))
    out["sc_wav"] = np.stack(ws)[:, ::DEC]
    out["sc_wav_len"] = len(ws[0])
    ows, osides = O.separate_and_stitch(mix1, None, 16000, O.OracleCssCfg(activity_th=0.3), separate_fn=sep_fn)
    report["sc_oracle_vs_ref_relrms"] = [G.rel_rms(ows[k], ws[k]) for k in range(3)]
    report["sc_oracle_perms_equal"] = [tuple(p) for p in osides["perms"][1:]] == [p for _, p in ts.pit]
    np.savez_compressed(os.path.join(HERE, "realistic_r4.npz"), **out)

    # ------------------------------------------------------------------ css_inference triple, 20 s, full scale
    import pandas as pd
    WIO = __import__("importlib").import_module("notsofar1_challenge_amd.wavio")
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(HERE, "calib_mc.npz"))
    st = W.apply_golden_recipe(W.portable_state_dict(desc, int(cal["weight_seed"])), head_bias=cal["head_bias"])
    model = G.build_reference_model(desc, st)
    mix60 = SYN.synth_meeting(60.0, 7, seed=1)
    written = {}
    import soundfile as SF

    def sf_read(path, dtype="float32"):
        pcm, sr = WIO.read_wav_pcm16(path)
        return pcm.astype(np.float32) / np.float32(32768.0), sr

    def sf_write(path, samps, sr):
        written[str(path)] = (np.asarray(samps).copy(), sr)
        WIO.write_pcm16_samples(path, np.clip(np.rint(np.asarray(samps, np.float64) * 32767.0), -32768, 32767).astype(np.int16), sr)

    SF.read, SF.write = sf_read, sf_write
    with tempfile.TemporaryDirectory() as td:
        sess_dir = os.path.join(td, "in")
        os.makedirs(sess_dir)
        # 1 209 frames = 12 segments of which the last is FULL (186 valid frames): a ragged last segment is ill-conditioned in
        # the reference itself (its complex64 solve is O(1) from a float64 one there, DESIGN.md hazard 8; with 20 s + 123
        # samples stream 0 of the oracle is 51 PCM16 steps off over exactly that segment and within 1 step everywhere else)
        off, n = 0, 1208 * 256 + 512 + 100
        clip = mix60[0, off:off + n]
        gain = float(0.9 / np.abs(clip).max())            # full scale: the loudest sample at 0.9
        pcm16 = np.clip(np.rint(clip * np.float32(gain) * 32768.0), -32768, 32767).astype(np.int16)
        names = []
        for c in range(7):
            p = os.path.join(sess_dir, f"ch{c}.wav")
            WIO.write_pcm16_samples(p, pcm16[:, c], 16000)
            names.append(p)
        session = pd.Series({"session_id": "MTG_SYNTH_mc_1", "is_mc": True, "wav_file_names": names, "device_name": "synth"})
        RC.load_css_model = lambda model_dir: (model, None)   # no checkpoint / OmegaConf here: the model is in memory
        out_dir = os.path.join(td, "out")
        base = dict(show_progressbar=False, activity_th=0.3)
        res = RC.css_inference(out_dir, "unused_models_dir", session, RC.CssCfg(**base, device="cpu"), fetch_from_cache=False)
        rel = lambda p: os.path.relpath(str(p), out_dir)
        files = sorted(rel(os.path.join(dp, f)) for dp, _, fs in os.walk(out_dir) for f in fs)
        to16 = lambda v: np.clip(np.rint(np.asarray(v, np.float64) * 32767.0), -32768, 32767).astype(np.int16)
        triple = {
            "input": {"session_id": session.session_id, "is_mc": True, "n_samples": n, "mix_seed": 1, "mix_offset": off,
                      "pcm16_gain": gain, "n_files": 7},
            "output_columns": sorted(res.index.tolist()),
            "sep_wav_file_names": [rel(p) for p in res["sep_wav_file_names"]],
            "files": files,
            "pcm16_sha256": {rel(k): G2.sha(to16(v[0])) for k, v in written.items()},
            "lengths": {rel(k): int(len(v[0])) for k, v in written.items()},
        }
        # the PCM16 samples of the separated streams in full (3 x 320 000 int16, compressed)
        np.savez_compressed(os.path.join(HERE, "session_triple_r4_pcm16.npz"),
                            **{rel(k).replace("/", "__"): to16(v[0]) for k, v in written.items() if "sep_stream" in k})
        # cache rule (css.py:79-82): a second call returns the sorted glob of sep*.wav; pass-through (css.py:73-75)
        res2 = RC.css_inference(out_dir, "unused_models_dir", session, RC.CssCfg(**base, device="cpu"), fetch_from_cache=True)
        triple["cached_sep_wav_file_names"] = [rel(p) for p in res2["sep_wav_file_names"]]
        res3 = RC.css_inference(out_dir, "unused", session, RC.CssCfg(**base, pass_through_ch0=True), fetch_from_cache=False)
        triple["pass_through"] = [os.path.basename(p) for p in res3["sep_wav_file_names"]]
        # what a second evaluation of the same path does to the files: the oracle (float64 MVDR) on the same PCM16 session
        mixf = (pcm16.astype(np.float32) / np.float32(32768.0))[None]
        ow, _ = O.separate_and_stitch(mixf, O.ConformerParams(st), 16000, O.OracleCssCfg(activity_th=0.3), mvdr_cplx=np.complex128)
        lsb = {}
        for k in range(3):
            ref16 = to16(written[os.path.join(out_dir, "css_inference", session.session_id, f"sep_stream{k}.wav")][0]).astype(np.int64)
            x = ow[k].astype(np.float32)
            mine = to16(x * np.float32(0.99) / (np.abs(x).max() + np.float32(1e-7))).astype(np.int64)
            dlt = np.abs(mine - ref16)
            lsb[f"stream{k}"] = {"equal": float((dlt == 0).mean()), "within_1_lsb": float((dlt <= 1).mean()), "max": int(dlt.max())}
        report["triple_oracle_vs_reference_pcm16"] = lsb
    with open(os.path.join(HERE, "session_triple_r4.json"), "w") as f:
        json.dump(triple, f, indent=1)
    with open(os.path.join(HERE, "golden_report_r4.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
