#!/usr/bin/env python3
"""Device-resident pass time against the meeting length (segments) for 1 and 3 lanes: is a pass bound by the latency of its
kernel chain (time ~ constant) or by throughput (time ~ segments)?   python tools/size_probe.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(240.0, 7, seed=1)
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=256); h = sep.handle
pd = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda()
print("| segments | rows | 1 lane ms | 3 lanes ms | 1 lane us/segment | 3 lanes us/segment | masknet ms (1 lane) |\n|---|---|---|---|---|---|---|")
for nseg in (2, 4, 7, 13, 20, 27, 40, 80, 160):
    n = ((nseg - 1) * 93 + 186 - 1) * 256 + 512
    plan = L.plan(desc, run_cfg, n)
    assert plan.num_segments == nseg, (plan.num_segments, nseg)
    wd = torch.empty((3, int(plan.n_out)), device="cuda")
    res = []
    for lanes in (1, 3):
        h.set_lanes(lanes)
        fn = lambda: h.run_device(pd.data_ptr(), n, 7, run_cfg, wd.data_ptr(), int(plan.n_out))
        for _ in range(3): fn()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize(); best = min(best, 1e3 * (time.perf_counter() - t0) / 10)
        res.append(best)
        if lanes == 1: mk = h.timings()["masknet"]
    print(f"| {nseg} | {nseg * 186} | {res[0]:.3f} | {res[1]:.3f} | {1e3 * res[0] / nseg:.1f} | {1e3 * res[1] / nseg:.1f} | {mk:.3f} |", flush=True)
sep.close()
