#!/usr/bin/env python3
"""Probe: overhead of the staged / sharded driver (parallel.py) over the fused css_run_device on ONE GPU with
world = 1 (no collectives: what is measured is stage-call granularity and whatever the driver adds on the host), and
the device-side cost of the three exchanges' pack / unpack steps for an 8-rank plan (stacked pieces instead of
collectives).   python tools/shard_overhead_probe.py [seconds]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L, PAR = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib"), pkg("parallel")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(seconds, 7, seed=1); n = mix.shape[1]
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
    return PAR.sharded_separate_and_stitch(be, 3, run_cfg.c.segment_frames, run_cfg.c.hop_frames, desc.frame_hop, 0, 1, None, out=out)
steps = 20 if seconds <= 120 else 5
print(f"## {seconds:g} s meeting, {plan.num_segments} segments, world = 1 on one MI355X\n")
res = {}
for f in (fused, staged, fused, staged):
    for _ in range(2): f()
    torch.cuda.synchronize(); h.sync()
    t0 = time.perf_counter()
    for _ in range(steps): f()
    h.sync(); torch.cuda.synchronize()
    res.setdefault(f.__name__, []).append(1e3 * (time.perf_counter() - t0) / steps)
for k, v in res.items():
    print(f"- {k:7s}: {min(v):.3f} ms per meeting  (runs: {', '.join(f'{x:.3f}' for x in v)})")
print(f"- staged / fused = {min(res['staged']) / min(res['fused']):.3f}  (the staged driver is parallel.py's phases at world = 1; the fused pass is css_run_device, which takes the same plain "
      f"stage sequence for resident samples)")
# ---- the exchanges' own device work for an 8-rank plan: pack, unpack (index_select / slice adds) -- no collective
world = 8
be.begin(pcm, n, 7, run_cfg)
ss = PAR.ShardedSession(be, 3, run_cfg.c.segment_frames, run_cfg.c.hop_frames, desc.frame_hop, 3, world)
with be.on_stream():
    c = ss.segments_and_costs(); costs = torch.stack([c] * world)
    a = ss.masks_and_activity(costs); acts = torch.stack([a] * world)
    s = ss.gate_and_istft(acts); shards = torch.stack([s] * world)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    idx_c = ss._gather_index("idx_costs", [(p.b_lo, p.b_hi) for p in ss.plans], ss.max_b + 1, ss.nseg - 1)
    idx_a = ss._gather_index("idx_act", [(p.t_lo, p.t_hi) for p in ss.plans], ss.max_t, ss.TL)
    ev[0].record()
    for _ in range(10): be.costs_view()[:ss.nseg - 1].copy_(costs.reshape(-1, 9).index_select(0, idx_c))
    ev[1].record()
    for _ in range(10): be.act_view().copy_(acts.permute(1, 0, 2).reshape(3, -1).index_select(1, idx_a))
    ev[2].record()
    for _ in range(10): ss.join_shards(shards, out)
    ev[3].record()
    torch.cuda.synchronize()
print(f"\n## device work of the exchanges' unpack steps, 8-rank plan (us per step; the collectives themselves need 8 GPUs)\n")
print(f"- PIT costs  [{world} x {ss.max_b + 1} x 9] f64 -> costs buffer: {ev[0].elapsed_time(ev[1]) * 100:.1f} us")
print(f"- activity   [{world} x 3 x {ss.max_t}] u8 -> activity bits:   {ev[1].elapsed_time(ev[2]) * 100:.1f} us")
print(f"- waveforms  [{world} x 3 x {ss.max_len}] f32 -> [3 x {ss.n_out}]:  {ev[2].elapsed_time(ev[3]) * 100:.1f} us")
del out, c, a, s, costs, acts, shards
be.close()
sep.close()
