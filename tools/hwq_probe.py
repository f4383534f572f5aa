#!/usr/bin/env python3
"""How many hardware queues the HIP runtime gives the handle's streams (GPU_MAX_HW_QUEUES, read by the runtime when it
starts) against the lane count, and two meetings in flight on two handles: the 60 s meeting, host -> host from
page-locked buffers and device-resident.   GPU_MAX_HW_QUEUES=8 python tools/hwq_probe.py"""
import importlib, os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(60.0, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
seps = [SEP.HipSeparator(state, None, device=0, max_batch_segments=128) for _ in range(2)]
hs = [s.handle for s in seps]
plan = L.plan(desc, run_cfg, n)
pcm = [L.pinned_copy(np.ascontiguousarray(mix[0])) for _ in hs]; out = [L.pinned_empty((3, int(plan.n_out)), np.float32) for _ in hs]
pd = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda(); wd = [torch.empty((3, int(plan.n_out)), device="cuda") for _ in hs]
def timed(fn, steps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
def both(kind):
    def one(i):
        if kind == "host": hs[i].run(pcm[i], run_cfg, out=out[i])
        else: hs[i].run_device(pd.data_ptr(), n, 7, run_cfg, wd[i].data_ptr(), int(plan.n_out))
    ts = [threading.Thread(target=one, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(unset)')}")
print("| lanes | host->host ms | device-resident ms (stage sequence) | device-resident ms (unit pipeline) | two handles in flight, per meeting: host->host | device-resident |")
print("|---|---|---|---|---|---|")
for lanes in (1, 2, 3, 4, 6):
    for h in hs: h.set_lanes(lanes)
    a = min(timed(lambda: hs[0].run(pcm[0], run_cfg, out=out[0])) for _ in range(2))
    hs[0].set_tuning("pipeline_device", 0)
    b = min(timed(lambda: hs[0].run_device(pd.data_ptr(), n, 7, run_cfg, wd[0].data_ptr(), int(plan.n_out))) for _ in range(2))
    hs[0].set_tuning("pipeline_device", 1)
    c = min(timed(lambda: hs[0].run_device(pd.data_ptr(), n, 7, run_cfg, wd[0].data_ptr(), int(plan.n_out))) for _ in range(2))
    hs[0].set_tuning("pipeline_device", 0)
    d = min(timed(lambda: both("host"), 10) for _ in range(2)) / 2
    e = min(timed(lambda: both("dev"), 10) for _ in range(2)) / 2
    print(f"| {lanes} | {a:.3f} | {b:.3f} | {c:.3f} | {d:.3f} | {e:.3f} |", flush=True)
for s in seps: s.close()
