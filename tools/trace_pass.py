#!/usr/bin/env python3
"""A few css_run passes (page-locked host buffers) for rocprofv3 --kernel-trace --memory-copy-trace; tools/timeline.py
reads the CSVs.    python tools/trace_pass.py [seconds] [passes] [device|host] [lanes]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "host"
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(seconds, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128); h = sep.handle
plan = L.plan(desc, run_cfg, n)
if len(sys.argv) > 4: h.set_lanes(int(sys.argv[4]))
pcm = L.pinned_copy(np.ascontiguousarray(mix[0])); out = L.pinned_empty((3, int(plan.n_out)), np.float32)
if mode == "device":
    pd = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda(); wd = torch.empty((3, int(plan.n_out)), device="cuda")
    torch.cuda.synchronize()
for _ in range(passes):
    if mode == "device": h.run_device(pd.data_ptr(), n, 7, run_cfg, wd.data_ptr(), int(plan.n_out))
    else: h.run(pcm, run_cfg, out=out)
print("ms of the last pass:", h.timings()["total"])
sep.close()
