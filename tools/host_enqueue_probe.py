#!/usr/bin/env python3
"""How long the HOST needs to enqueue one pass (css_run_enqueue returns when everything is queued) against how long
the device needs to run it: is the device ever waiting for launches?   python tools/host_enqueue_probe.py [seconds]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(seconds, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128); h = sep.handle
plan = L.plan(desc, run_cfg, n)
pcm = L.pinned_copy(np.ascontiguousarray(mix[0]))
outs = [L.pinned_empty((3, int(plan.n_out)), np.float32) for _ in range(3)]
for lanes in (1, 2, 3, 4):
    h.set_lanes(lanes)
    for _ in range(3):
        h.run(pcm, run_cfg, out=outs[0])
    enq, tot = [], []
    for _ in range(10):
        h.sync()
        t0 = time.perf_counter()
        h.run_enqueue(pcm, run_cfg, outs[0])
        t1 = time.perf_counter()
        h.wait()
        t2 = time.perf_counter()
        enq.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
    # a queue of 12: the host may run ahead
    h.sync(); t0 = time.perf_counter()
    for i in range(12):
        h.run_enqueue(pcm, run_cfg, outs[i % 3])
    t1 = time.perf_counter(); h.wait(); t2 = time.perf_counter()
    print(f"lanes={lanes}: one pass: host enqueue {np.median(enq):.3f} ms, until done {np.median(tot):.3f} ms | queue of 12: "
          f"host {1e3 * (t1 - t0) / 12:.3f} ms per pass, done {1e3 * (t2 - t0) / 12:.3f} ms per pass")
sep.close()
