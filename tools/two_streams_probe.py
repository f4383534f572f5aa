#!/usr/bin/env python3
"""Probe: do two independent CSS sessions on two HIP streams of one GPU overlap (fill each other's launch
prologue / epilogue bubbles)?  Prints single-session and two-concurrent-session throughput.  GPU box only."""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(seconds, 7, seed=1)
n = mix.shape[1]
cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)
run_cfg = CSS.make_run_cfg(cfg, 16000, 7, desc.frame_len, desc.frame_hop)
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(np.ascontiguousarray(mix[0])).to(dev)
plan = L.plan(desc, run_cfg, n)
seps = [SEP.HipSeparator(state, None, device=0, max_batch_segments=128) for _ in range(2)]
outs = [torch.empty((3, plan.n_out), dtype=torch.float32, device=dev) for _ in range(2)]
torch.cuda.synchronize()
def work(i, steps):
    h = seps[i].handle
    for _ in range(steps):
        h.run_device(pcm.data_ptr(), n, 7, run_cfg, outs[i].data_ptr(), plan.n_out)
for i in range(2): work(i, 3)
K = 20
t0 = time.perf_counter(); work(0, K); t1 = time.perf_counter() - t0
print(f"one session : {1e3 * t1 / K:.3f} ms per meeting, {seconds * K / t1:.0f} x real time")
th = [threading.Thread(target=work, args=(i, K)) for i in range(2)]
t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t2 = time.perf_counter() - t0
print(f"two sessions: {1e3 * t2 / (2 * K):.3f} ms per meeting, {seconds * 2 * K / t2:.0f} x real time  (x{(2 * K / t2) / (K / t1):.2f})")
