#!/usr/bin/env python3
"""A/B of the css_run* schedule options on ONE box (boxes of the pool differ by +-5 %): every variant in turn, several
rounds, the 60 s meeting host -> host from page-locked buffers.   python tools/ab_tuning.py [seconds] [rounds]"""
import importlib, itertools, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(seconds, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128); h = sep.handle
plan = L.plan(desc, run_cfg, n)
pcm = L.pinned_copy(np.ascontiguousarray(mix[0])); out = L.pinned_empty((3, int(plan.n_out)), np.float32)
pd = torch.from_numpy(np.ascontiguousarray(mix[0])).cuda(); wd = torch.empty((3, int(plan.n_out)), device="cuda")
def timed(fn, steps=20 if seconds <= 120 else 4):
    for _ in range(3): fn()
    h.sync(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    h.sync(); return 1e3 * (time.perf_counter() - t0) / steps
variants = [dict(lanes=l, tail_pieces=p, out_mapped=m, tail_per_unit=u, mvdr_on_lanes=v)
            for l, p, m, u, v in [(3, 1, 1, 0, 1), (3, 1, 1, 0, 0), (3, 1, 0, 0, 1), (3, 1, 0, 0, 0), (3, 2, 0, 0, 0), (3, 1, 1, 1, 1), (2, 1, 1, 0, 1), (4, 1, 1, 0, 1)]]
res = {i: [] for i in range(len(variants))}; dev = {i: [] for i in range(len(variants))}
for r in range(rounds):
    for i, v in enumerate(variants):
        h.set_lanes(v["lanes"])
        for k in ("tail_pieces", "out_mapped", "tail_per_unit", "mvdr_on_lanes"): h.set_tuning(k, v[k])
        res[i].append(timed(lambda: h.run(pcm, run_cfg, out=out)))
        dev[i].append(timed(lambda: h.run_device(pd.data_ptr(), n, 7, run_cfg, wd.data_ptr(), int(plan.n_out))))
print(f"| variant | host -> host ms (rounds) | device-resident ms | ratio |\n|---|---|---|---|")
for i, v in enumerate(variants):
    a, b = np.array(res[i]), np.array(dev[i])
    print(f"| {v} | {a.min():.3f} ({', '.join(f'{x:.3f}' for x in a)}) | {b.min():.3f} | {a.min() / b.min():.3f} |")
sep.close()
