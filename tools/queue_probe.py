#!/usr/bin/env python3
"""A queue of sessions: K meetings host -> host, synchronous css_run against css_run_enqueue ... css_wait on one handle
(consecutive passes overlap) and on two handles used in turn (they only contend).   python tools/queue_probe.py [seconds]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(seconds, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
seps = [SEP.HipSeparator(state, None, device=0, max_batch_segments=128) for _ in range(2)]
hs = [s.handle for s in seps]
if os.environ.get('LANES'):
    for h in hs: h.set_lanes(int(os.environ['LANES']))
plan = L.plan(desc, run_cfg, n)
pcm = L.pinned_copy(np.ascontiguousarray(mix[0]))
outs = [L.pinned_empty((3, int(plan.n_out)), np.float32) for _ in range(2)]
pd = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda(); wd = torch.empty((3, int(plan.n_out)), device="cuda")
K = 20 if seconds <= 120 else 4
def sync_run():
    for k in range(K): hs[0].run(pcm, run_cfg, out=outs[0])
def dev_run():
    for k in range(K): hs[0].run_device(pd.data_ptr(), n, 7, run_cfg, wd.data_ptr(), int(plan.n_out))
def queued(two=False):
    for k in range(K):
        h = hs[k % 2] if two else hs[0]
        h.run_enqueue(pcm, run_cfg, outs[k % 2])
    for h in hs: h.wait()
ref = hs[0].run(pcm, run_cfg).copy()
if len(sys.argv) > 2 and sys.argv[2] == "trace":
    queued(); queued(); torch.cuda.synchronize()
    for s in seps: s.close()
    sys.exit(0)
def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, 1e3 * (time.perf_counter() - t0) / K)
    return best
print(f"{seconds:g} s meeting, {K} meetings back to back, ms per meeting:")
print(f"  css_run, synchronous, one handle        : {timed(sync_run):.3f}")
print(f"  css_run_device (resident), one handle   : {timed(dev_run):.3f}")
print(f"  css_run_enqueue, one handle             : {timed(lambda: queued(False)):.3f}")
print(f"  css_run_enqueue, two handles in turn    : {timed(lambda: queued(True)):.3f}")
queued(False)
print("  results equal the synchronous pass:", all(np.array_equal(o[:, :plan.n_out], ref) for o in outs))
for s in seps: s.close()
