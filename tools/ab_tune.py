#!/usr/bin/env python3
"""A/B of css_set_tuning(NAME, v) inside ONE process, passes interleaved (box-to-box and run-to-run spread is 3 %,
more than the effect): GEMM time per pass from the library's own HIP-event profile (one lane) and device-resident pass
time on the production schedule (three lanes).   python tools/ab_tune.py NAME|env:VAR v0,v1,... [seconds] [rounds]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)  # noqa: E731


def main():
    import torch
    name = sys.argv[1]
    modes = [int(v) for v in sys.argv[2].split(",")]
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
    run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
    mix = SYN.synth_meeting(seconds, 7, seed=1)
    n = mix.shape[1]
    plan = L.plan(desc, run_cfg, n)
    sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128)
    h = sep.handle
    pcm = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda()
    wav = torch.empty((3, int(plan.n_out)), dtype=torch.float32, device="cuda")
    go = lambda: h.run_device(pcm.data_ptr(), n, 7, run_cfg, wav.data_ptr(), int(plan.n_out))  # noqa: E731
    for _ in range(3):
        go()
    gemm = {m: [] for m in modes}
    dev = {m: [] for m in modes}
    ref = None
    for r in range(rounds):
        for m in modes:
            if name.startswith("env:"):      # a launcher-level switch read from the environment at every launch
                os.environ[name[4:]] = str(m)
            else:
                h.set_tuning(name, m)
            h.set_profile(True)
            go(); go()
            t = h.timings()
            h.set_profile(False)
            gemm[m].append(t["gemm_ms"])
            go()
            h.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                go()
            h.sync(); torch.cuda.synchronize()
            dev[m].append(1e3 * (time.perf_counter() - t0) / 10)
            out = wav.cpu().numpy()
            if ref is None:
                ref = out
            assert np.array_equal(out, ref), m
    print(f"{seconds:g} s meeting, {rounds} interleaved rounds; results bit-identical in every mode")
    for m in modes:
        print(f"{name}={m}: GEMM ms per pass (one lane, HIP events) median {np.median(gemm[m]):.3f} min {min(gemm[m]):.3f} | "
              f"device-resident pass ms median {np.median(dev[m]):.3f} min {min(dev[m]):.3f}")
    sep.close()


if __name__ == "__main__":
    main()
