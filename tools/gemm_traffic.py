#!/usr/bin/env python3
"""Only the headline's estimator launches, for the counter passes (tools/profile_r05.sh): a handle in the given arithmetic
mode runs the headline's shared batch of queued 60 s sessions under the per-launch profile (one lane; exact float32: six
sessions = M 44 640 rows per Linear-layer launch, split-f16: three = 22 320 -- bench.py sessions_in_a_batch) a few times.  Every dispatch of the mode's GEMM kernel in such a run is one of the 110 launches
`roofline.achieved` averages over, so counter sums / launches are per-launch figures of exactly those launches.
    python tools/gemm_traffic.py exact_f32|split_f16 [passes]          (prints the algorithmic bytes per launch as JSON)"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pkg(n):
    return importlib.import_module("notsofar1_challenge_amd." + n)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "exact_f32"
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    mix = SYN.synth_meeting(60.0, 7, seed=1)
    plan = L.plan(desc, run_cfg, mix.shape[1])
    sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=256)
    ns = int(sys.argv[3]) if len(sys.argv) > 3 else (6 if mode == "exact_f32" else 3)
    h = sep.handle
    h.set_linear_mode(mode)
    pcm = L.pinned_copy(np.ascontiguousarray(mix[0]))
    outs = [L.pinned_empty((3, int(plan.n_out)), np.float32) for _ in range(2)]
    h.set_profile(True)
    for _ in range(passes):
        for k in range(ns):
            h.run_enqueue(pcm, run_cfg, outs[k % 2])
        h.wait()
    t = h.timings()
    h.set_profile(False)
    sep.close()
    # algorithmic bytes of the 110 launches (float32 operands: A + W + C, + the residual where there is one)
    M, D, FF, Kp, F4 = ns * 40 * 186, 512, 1024, 1824, 257 * 4
    f = 4.0
    per = {"embed": M * Kp * f + D * Kp * f + M * D * f,
           "ffn_up": M * D * f + FF * D * f + M * FF * f, "ffn_down": M * FF * f + D * FF * f + 2 * M * D * f,
           "qkv": M * D * f + 3 * D * D * f + M * 3 * D * f, "attn_out": M * D * f + D * D * f + 2 * M * D * f,
           "head": F4 * D * f + M * D * f + F4 * M * f}
    total = per["embed"] + 18 * (2 * per["ffn_up"] + 2 * per["ffn_down"] + per["qkv"] + per["attn_out"]) + per["head"]
    print(json.dumps({"mode": mode, "rows_per_launch": M, "launches_per_batch": 110, "gemm_ms_per_batch": t["gemm_ms"],
                      "gemm_launches_profiled": t["gemm_launches"], "algorithmic_bytes_per_launch": total / 110.0}))


if __name__ == "__main__":
    main()
