#!/usr/bin/env python3
"""Probe: overhead of the staged / sharded driver (parallel.py) over the fused css_run_device on ONE GPU with
world = 1 (no collectives: what is measured is stage-call granularity, host round trips and synchronisations)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L, PAR = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib"), pkg("parallel")
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(60.0, 7, seed=1); n = mix.shape[1]
cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
run_cfg = CSS.make_run_cfg(cfg, 16000, 7, desc.frame_len, desc.frame_hop)
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(np.ascontiguousarray(mix[0])).to(dev)
plan = L.plan(desc, run_cfg, n)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128)
h = sep.handle
out = torch.empty((3, plan.n_out), dtype=torch.float32, device=dev)
be = PAR.HipShardBackend(h, dev, dev)
def fused(): h.run_device(pcm.data_ptr(), n, 7, run_cfg, out.data_ptr(), plan.n_out)
def staged():
    be.begin(pcm, n, 7, run_cfg)
    return PAR.sharded_separate_and_stitch(be, 3, run_cfg.c.segment_frames, run_cfg.c.hop_frames, desc.frame_hop, 0, 1, None)
for f in (fused, staged):
    for _ in range(3): f()
    torch.cuda.synchronize(); h.sync()
    t0 = time.perf_counter()
    for _ in range(20): r = f()
    h.sync(); torch.cuda.synchronize()
    print(f"{f.__name__:7s}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per 60 s meeting")
