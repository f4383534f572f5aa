#!/usr/bin/env python3
"""Round-6 fixture: how far the REFERENCE is from ITSELF on the screened configs[1] meeting (e2e60_r5.npz, seed 22), and the
list of its winner-take-all decisions that rounding can flip.  Rules as in gen_golden.py (the reference is imported in place,
only seeds and output tensors are written).

    python tests/golden/gen_golden_r6.py

  e2e60_r6_self.npz   the reference's separate_and_stitch (css/css.py:110-338) on the SAME input and weights with torch at
                      8 / 4 / 2 / 1 threads (ATen's GEMM blocking and reduction order depend on the thread count; numpy's
                      OpenBLAS einsum / solve follow the process default).  Per thread count against the 8-thread run (the
                      one e2e60_r5.npz holds): free-running waveform rel-RMS per stream (all samples and the ::64 decimation
                      the fixture uses), mask max-abs difference, the winner sets (mvdr_util.py:50-55) that differ and where.
                      `near_*`: every (segment, bin, frame) whose top-2 mask margin in the 8-thread run is below 2e-5 -- the
                      decisions a rounding-level difference can flip --, the four float32 masks there in each run, and the
                      SAME network evaluated in float64 (oracle with float64 parameters on the oracle's features) there.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import gen_golden_r5 as G5  # noqa: E402

import torch  # noqa: E402

RC, W, SYN, O = G.RC, G.W, G.SYN, G.O
THREADS = (8, 4, 2, 1)


def masks_of(tap):
    return np.stack([np.concatenate([t["spk_masks"][0], t["noise_masks"][0]], -1) for t in tap.masks])   # [40, F, T, 4]


def main():
    torch.manual_seed(0)
    g5 = np.load(os.path.join(HERE, "e2e60_r5.npz"))
    seed = int(g5["mix_seed"])
    desc, st = G5.mc_model()
    model = G.build_reference_model(desc, st)
    cfg = RC.CssCfg(show_progressbar=False, activity_th=0.3)
    mix = G5.meeting(seed)
    runs = {}
    for thr in THREADS:
        w, side, tap, dt = G.run_reference(model, mix, cfg, threads=thr)
        runs[thr] = {"wav": np.stack(w), "masks": masks_of(tap), "perm": np.array([p for _, p in tap.pit], np.int32),
                     "act_f": side["activity_final"].numpy()[0].copy(), "act_b": side["activity_b"].numpy().copy(), "dt": dt}
        print(f"threads {thr}: {dt:.1f} s", flush=True)
    base = runs[THREADS[0]]
    same_as_r5 = bool(np.array_equal(base["wav"][:, ::64], g5["wav_dec64"]))
    print("8-thread run reproduces e2e60_r5.npz's waveforms bit for bit:", same_as_r5, flush=True)
    m0 = base["masks"]
    srt = np.sort(m0, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    near = np.argwhere(margin < 2e-5).astype(np.int32)                     # [n, 3] = (segment, bin, frame)
    e = {"mix_seed": seed, "mix_samples": int(g5["mix_samples"]), "threads": np.array(THREADS, np.int32),
         "base_run_equals_e2e60_r5": same_as_r5, "near_points": near,
         "near_margin": margin[tuple(near.T)].astype(np.float32)}
    report = {"seed": seed, "base_run_equals_e2e60_r5": same_as_r5, "near_points": int(len(near)), "runs": {}}
    win0 = m0 == m0.max(axis=-1, keepdims=True)
    for thr in THREADS:
        r = runs[thr]
        e[f"near_masks_t{thr}"] = r["masks"][tuple(near.T)].astype(np.float32)   # [n, 4]
        if thr == THREADS[0]:
            continue
        win = r["masks"] == r["masks"].max(axis=-1, keepdims=True)
        differ = np.argwhere(np.any(win != win0, axis=-1)).astype(np.int32)
        rr = [G.rel_rms(r["wav"][k], base["wav"][k]) for k in range(3)]
        rr64 = [G.rel_rms(r["wav"][k, ::64], base["wav"][k, ::64]) for k in range(3)]
        e[f"wta_differ_t{thr}"] = differ
        e[f"wav_rel_rms_t{thr}"] = np.array(rr)
        e[f"wav_rel_rms_dec64_t{thr}"] = np.array(rr64)
        e[f"masks_max_abs_t{thr}"] = float(np.abs(r["masks"] - m0).max())
        report["runs"][str(thr)] = {
            "wall_s": r["dt"], "waveform_rel_rms_vs_8_threads": rr, "waveform_rel_rms_dec64_vs_8_threads": rr64,
            "masks_max_abs_vs_8_threads": e[f"masks_max_abs_t{thr}"], "winner_sets_that_differ": int(len(differ)),
            "winner_sets_that_differ_at": differ.tolist(),
            "perms_equal": bool(np.array_equal(r["perm"], base["perm"])),
            "activity_equal": bool(np.array_equal(r["act_f"], base["act_f"]) and np.array_equal(r["act_b"], base["act_b"]))}
        print(thr, report["runs"][str(thr)], flush=True)
    # the same network in float64 at the near points (segment by segment, oracle features of the oracle's STFT)
    X = O.stft(mix[0])
    p64 = O.ConformerParams(st, dtype=np.float64)
    f64 = np.zeros((len(near), 4), np.float64)
    for i in sorted(set(int(s) for s in near[:, 0])):
        seg = np.zeros((X.shape[0], 186, X.shape[2]), X.dtype)          # the last segment is zero-padded (css.py:185-190)
        part = X[:, i * 93:i * 93 + 186]
        seg[:, :part.shape[1]] = part
        m64 = np.moveaxis(O.conformer_forward(p64, O.features(seg).astype(np.float64)), 0, 2)   # [F, T, 4]
        sel = near[:, 0] == i
        f64[sel] = m64[near[sel, 1], near[sel, 2]]
        print("float64 segment", i, int(sel.sum()), "points", flush=True)
    e["near_masks_f64"] = f64
    # where the reference's own (8-thread) decision is not float64's
    ref_arg = np.argmax(e["near_masks_t8"], axis=-1)
    f64_arg = np.argmax(f64, axis=-1)
    report["near_points_where_the_reference_differs_from_float64"] = int((ref_arg != f64_arg).sum())
    report["worst_self_distance_free_running"] = float(max(max(v["waveform_rel_rms_dec64_vs_8_threads"]) for v in report["runs"].values()))
    np.savez_compressed(os.path.join(HERE, "e2e60_r6_self.npz"), **e)
    with open(os.path.join(HERE, "golden_report_r6.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1)[:3000])


if __name__ == "__main__":
    main()
