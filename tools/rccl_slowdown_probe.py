#!/usr/bin/env python3
"""Probe: why a process that has initialised RCCL runs the SAME single-GPU workload ~12 % slower
(profiles/r03_rccl_world1.json: 167.5 vs 149.7 ms per 30-min meeting) and why the first upload piece of a sharded step took
14.9 ms.  One process, one GPU, the fused host -> host pass of the 1800 s meeting timed
  (a) before torch.distributed exists,  (b) after init_process_group("nccl", world 1) + one all-reduce (communicator built),
  (c) after destroy_process_group,      and with the suspects toggled: a torch-owned stream for the handle, a second handle,
  the pieces of the upload timed by events.          python tools/rccl_slowdown_probe.py [seconds]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, SYN, CSS, SEP, L, PAR = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("_lib"), pkg("parallel")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1800.0
desc = W.ModelDesc.mc_v1()
cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
mix = SYN.synth_meeting(seconds, 7, seed=1); n = mix.shape[1]
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7, desc.frame_len, desc.frame_hop)
plan = L.plan(desc, run_cfg, n)
dev = torch.device("cuda", 0)
pcm = L.pinned_copy(np.ascontiguousarray(mix[0]))
out = L.pinned_empty((3, int(plan.n_out)), np.float32)
res = {"seconds": seconds, "segments": int(plan.num_segments)}


def timed(h, steps=4, warmup=2):
    for _ in range(warmup):
        h.run(pcm, run_cfg, out=out)
    h.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.run(pcm, run_cfg, out=out)
    h.sync(); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    t = h.timings()
    return {"ms": round(ms, 2), "stage_ms": {k: round(float(t[k]), 2) for k in ("upload", "masknet", "stitch", "istft", "download", "total", "host_enqueue")}}


def first_piece(h, ts=None):
    """device time of css_begin_range's first piece (32 segments' samples) alone, and with the rest following on the copy stream"""
    be = PAR.HipShardBackend(h, dev, dev, torch_stream=ts) if ts is not None else PAR.HipShardBackend(h, dev, dev)
    me = PAR.make_shard_plan(int(plan.num_segments), int(plan.mix_frames), int(plan.stft_frames), 186, 93, 256, 0, 1)
    groups, cuts = PAR.upload_schedule(me, 186, 93, 512, n)
    stream = ts if ts is not None else torch.cuda.ExternalStream(h.stream_ptr(), device=dev)
    outp = {}
    for name, c in (("first_piece_only", cuts[:1] + []), ("with_the_rest_behind_it", cuts)):
        vals = []
        for _ in range(3):
            h.sync(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            if name == "first_piece_only":
                h.begin_range(pcm, n, 7, run_cfg, 0, int(cuts[0]))
            else:
                be.begin(pcm, n, 7, run_cfg, sample_range=(0, n), slice_only=False, cuts=cuts)
            b.record(stream)
            h.sync(); torch.cuda.synchronize()
            vals.append(round(a.elapsed_time(b), 3))
        outp[name] = vals
    outp["first_piece_MB"] = round(cuts[0] * 7 * 4 / 1e6, 1)
    be.close()
    return outp


sep = SEP.HipSeparator(state, None, device=0, max_batch_segments=128)
h = sep.handle
res["a_before_distributed"] = timed(h)
res["a_first_piece"] = first_piece(h)
ts = torch.cuda.Stream(device=dev)
sep2 = SEP.HipSeparator(state, None, device=0, max_batch_segments=128, stream=int(ts.cuda_stream))
res["a_handle_on_a_torch_stream"] = timed(sep2.handle)
res["a_first_piece_torch_stream"] = first_piece(sep2.handle, ts)

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
res["b0_after_init_before_any_collective"] = timed(h)
x = torch.ones(4, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
res["b_after_init_and_a_collective"] = timed(h)
res["b_first_piece"] = first_piece(h)
res["b_handle_on_a_torch_stream"] = timed(sep2.handle)
sep3 = SEP.HipSeparator(state, None, device=0, max_batch_segments=128)
res["b_new_handle_created_after_init"] = timed(sep3.handle)
env = {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "NCCL_", "RCCL_", "GPU_", "AMD_", "ROC"))}
res["env"] = env
dist.destroy_process_group()
res["c_after_destroy"] = timed(h)
print(json.dumps(res, indent=1))
for s_ in (sep, sep2, sep3):
    s_.close()
