#!/usr/bin/env python3
"""One meeting over the GPUs of a node (as a runnable script):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/sharded_meeting.py [seconds]

Every rank uploads only its own slice of the recording and ends with the finished samples of its own range; the ranges
concatenated are the single-GPU result, bit for bit (checked here against a fused run on rank 0's GPU when CHECK=1)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
parallel, css, _lib, sepmod, weights, synth = (pkg(n) for n in ("parallel", "css", "_lib", "separator", "weights", "synth"))

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
desc = weights.ModelDesc.mc_v1()                                     # a real deployment: separator.load_css_model(dir)
state = weights.apply_golden_recipe(weights.portable_state_dict(desc, 0))
pcm_host = np.ascontiguousarray(synth.synth_meeting(seconds, 7, seed=1)[0])    # [n, 7] float32, identical on every rank
n = pcm_host.shape[0]
cfg = css.CssCfg(activity_th=0.3, show_progressbar=False)

ts = torch.cuda.Stream(device=dev)                                   # torch owns the stream the handle works on
sep = sepmod.HipSeparator(state, None, device=dev.index, max_batch_segments=160, stream=int(ts.cuda_stream))
be = parallel.HipShardBackend(sep.handle, dev, torch_stream=ts)
run_cfg = css.make_run_cfg(cfg, 16000, 7)
plan = _lib.plan(desc, run_cfg, n)
me = parallel.make_shard_plan(int(plan.num_segments), int(plan.mix_frames), int(plan.stft_frames), 186, 93, 256, rank, world)
lo, hi = me.pcm_range(512, n)                                        # the samples this rank needs
piece = _lib.pinned_copy(np.ascontiguousarray(pcm_host[lo:hi]))      # page-locked
groups, cuts = parallel.upload_schedule(me, 186, 93, 512, n)         # upload in pieces, under the stages
be.begin(piece, n, 7, run_cfg, sample_range=(lo, hi), slice_only=True, cuts=cuts)
own, (o_lo, o_hi) = parallel.sharded_separate_and_stitch(be, 3, 186, 93, 256, rank, world, dist, gather="range",
                                                          segment_groups=groups)
with be.on_stream():
    mine = own.cpu().numpy()                                         # [3, o_hi - o_lo]: this rank's samples of the result
print(f"[rank {rank}] segments {me.seg_lo}..{me.seg_hi}, samples [{o_lo}, {o_hi}) of {int(plan.n_out)}, rms {float(np.sqrt((mine ** 2).mean())):.4f}")
if os.environ.get("CHECK") == "1":
    ref = sep.handle.run(pcm_host, run_cfg)
    print(f"[rank {rank}] own range == fused single-GPU pass: {bool(np.array_equal(ref[:, o_lo:o_hi], mine))}")
del own
be.close(); sep.close(); dist.destroy_process_group()
