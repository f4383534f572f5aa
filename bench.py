#!/usr/bin/env python3
"""CSS real-time-factor benchmark (BASELINE.json metric) on 1..8 MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the CSS hot path (css/css.py::separate_and_stitch equivalent: STFT ->
Conformer mask estimator -> WTA/SCM/MVDR -> PIT stitch -> activity gate -> iSTFT) over one synthetic
7-channel 16 kHz meeting whose PCM is already resident in HBM.  N = 1: the 60 s meeting of
BASELINE.json configs[1].  N > 1: ONE meeting of N x 60 s, sharded by sliding-window segment across the
ranks with the RCCL all-gather stitch of notsofar1-challenge_amd/parallel.py (weak scaling: 40 segments
per GPU).  `value` = audio seconds separated per wall second over the whole job.

Weights: v1.0-MC architecture (D=512, H=8, 18 blocks, 1799 inputs), seeded portable random init with the
conditioning recipe of the golden tests (no pretrained checkpoint exists offline) -- arithmetic and
memory traffic are weight-independent.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MATRIX_TFLOPS = 2500.0  # same guide: dense f16 / bf16 MFMA peak (v_mfma_f32_32x32x16_f16)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(mix, state, seconds, cfg_kwargs, threads=16):
    """The oracle (numpy restatement of the reference's path, kind = "port") timed on the host cores on a
    bounded slice of the same workload.  This is the ONLY place bench.py touches oracle/.

    BLAS threads are limited to `threads`: on the 256-thread host of the GPU box OpenBLAS' default (64
    threads) runs this workload 3.5x SLOWER than 8-16 threads (measured: 1.5x vs 5.5x real time), and the
    reference's own measurement (5.3x on 8 cores) is an 8-thread figure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import css_oracle as O
    n = int(seconds * 16000)
    sample = np.ascontiguousarray(mix[:, :n])
    params = O.ConformerParams(state)
    threads = min(threads, os.cpu_count() or threads)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:  # pragma: no cover
        import contextlib
        limiter, threads = contextlib.nullcontext(), os.cpu_count()
    with limiter:
        t0 = time.time()
        O.separate_and_stitch(sample, params, 16000, O.OracleCssCfg(**cfg_kwargs))
        dt = time.time() - t0
    return {"value": round(seconds / dt, 3), "unit": "x real-time (audio s / wall s)", "cores": threads,
            "kind": "port", "sample": f"first {seconds:g} s of the same 7-ch meeting, numpy/OpenBLAS float32 oracle on "
                                      f"{threads} threads, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=60.0, help="meeting seconds per GPU")
    ap.add_argument("--max-batch", type=int, default=128, help="segments per batched mask-estimator pass")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=60.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run with N processes")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    # functional-test knobs (never used by the driver): CSS_BENCH_BACKEND=gloo exchanges through host memory,
    # CSS_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 (a 1-GPU box can then exercise the N > 1 code path)
    backend = os.environ.get("CSS_BENCH_BACKEND", "nccl")
    if os.environ.get("CSS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    W = importlib.import_module("notsofar1_challenge_amd.weights")
    SYN = importlib.import_module("notsofar1_challenge_amd.synth")
    CSS = importlib.import_module("notsofar1_challenge_amd.css")
    SEP = importlib.import_module("notsofar1_challenge_amd.separator")
    PAR = importlib.import_module("notsofar1_challenge_amd.parallel")

    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
    total_seconds = args.seconds * world
    t0 = time.time()
    mix = SYN.synth_meeting(total_seconds, 7, seed=1)  # [1, n, 7]; identical on every rank
    n = mix.shape[1]
    log(f"[rank {rank}] synthetic meeting {total_seconds:g} s generated in {time.time() - t0:.1f} s")

    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)  # configs/inference/inference_v1.yaml
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7, desc.frame_len, desc.frame_hop)
    sep = SEP.HipSeparator(state, None, device=local_rank, max_batch_segments=args.max_batch)
    h = sep.handle
    pcm_dev = torch.from_numpy(np.ascontiguousarray(mix[0])).to(dev)  # resident in HBM before timing
    S = desc.num_spks
    L = importlib.import_module("notsofar1_challenge_amd._lib")
    plan = L.plan(desc, run_cfg, n)
    wav_dev = torch.empty((S, plan.n_out), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    be = PAR.HipShardBackend(h, dev, comm_dev)

    def step():
        if world == 1:
            h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, wav_dev.data_ptr(), plan.n_out)
            return wav_dev
        be.begin(pcm_dev, n, 7, run_cfg)
        return PAR.sharded_separate_and_stitch(be, S, run_cfg.c.segment_frames, run_cfg.c.hop_frames,
                                               desc.frame_hop, rank, world, dist)

    def barrier():
        if world > 1:
            dist.barrier()
        h.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()
    if os.environ.get("CSS_BENCH_CHECK") == "1":   # functional test: the sharded result equals the fused single-GPU run
        ref = torch.empty((S, plan.n_out), dtype=torch.float32, device=dev)
        h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, ref.data_ptr(), plan.n_out)
        same = bool(torch.equal(ref.cpu(), out.cpu()))
        log(f"[rank {rank}] sharded == fused single-GPU result: {same}")
        assert same

    result = {
        "metric": "CSS real-time-factor (sep. audio sec/wall sec) on 7-ch 16 kHz",
        "value": round(total_seconds * args.steps / elapsed, 2),
        "unit": "x real-time (audio s / wall s)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 (Linear layers: f32 products as 3 f16 MFMAs on split operands, f32 accumulate; MVDR f64)"
                  if h.linear_mode() == "split_f16" else "f32 (MVDR covariance/solve f64)"), "data": "synthetic",
        "config": {"workload": f"synthetic 7-ch 16 kHz {total_seconds:g} s meeting "
                               f"({plan.num_segments} segments of 3 s / 1.5 s hop), Conformer-CSS v1.0-MC "
                               f"(18 blocks, D=512) + MVDR, PCM resident in HBM",
                   "seconds_per_gpu": args.seconds, "segments": int(plan.num_segments),
                   "sharding": "single GPU" if world == 1 else f"{world} ranks x segment ranges, RCCL all-gather stitch"},
    }

    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel (the fp32 MFMA GEMM): live HIP-event timing of every launch
        h.set_profile(True)
        h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, wav_dev.data_ptr(), plan.n_out)
        h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, wav_dev.data_ptr(), plan.n_out)
        t = h.timings()
        h.set_profile(False)
        # `achieved` counts ALGORITHMIC flops (2*M*N*K of the float32 products the network defines).
        achieved = t["gemm_flops"] / (t["gemm_ms"] * 1e-3) / 1e12 if t["gemm_ms"] > 0 else 0.0
        if h.linear_mode() == "split_f16":
            # every product is three f16 MFMAs (hi*hi + hi*lo + lo*hi, f32 accumulate): the ceiling for
            # float32-grade products on the f16 pipes is the dense f16 peak / 3, and achieved / that ceiling
            # equals executed-MFMA-flops / dense f16 peak (the matrix-core utilisation).
            peak = PEAK_F16_MATRIX_TFLOPS / 3.0
            kernel = "css::gemm_split_wd_kernel (3 x v_mfma_f32_32x32x16_f16 per product, 64x128x32 tiles, weights direct)"
            extra = {"mfma_executed_tflops": round(3 * achieved, 2), "mfma_dense_peak_tflops": PEAK_F16_MATRIX_TFLOPS,
                     "vs_f32_matrix_peak": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 3)}
        else:
            peak = PEAK_FP32_MATRIX_TFLOPS
            kernel = "css::gemm_kernel (v_mfma_f32_32x32x2_f32, 128x128x32 tiles)"
            extra = {}
        # HBM bytes per launch of that kernel: PMC counters cannot be read inside this process, so the figure comes
        # from the committed PMC passes of the same command (profiles/README.md), when present
        traffic = None
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tj = os.path.join(ROOT, "profiles", f"{tag}_gemm_traffic.json")
            if os.path.exists(tj):
                with open(tj) as f:
                    traffic = round(json.load(f)["traffic_bytes_per_launch"])
                extra["traffic_source"] = f"profiles/{tag}_gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
                break
        result["roofline"] = {
            "bound": "mfma", "kernel": kernel,
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "launches_per_step": int(t["gemm_launches"]),
            "avg_launch_us": round(1e3 * t["gemm_ms"] / max(t["gemm_launches"], 1), 2),
            "flops_per_step": t["gemm_flops"], **extra,
        }
        h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, wav_dev.data_ptr(), plan.n_out)
        result["stage_ms"] = {k: round(v, 3) for k, v in h.timings().items()
                              if k in ("upload", "stft", "masknet", "mvdr", "stitch", "istft", "download", "total")}
        # the same pass through the host-buffer entry point (css_run: PCIe upload of the PCM, download of the
        # waveforms) -- reported beside `value`, never as `value`
        h.run(mix[0], run_cfg)
        t0 = time.perf_counter()
        for _ in range(5):
            h.run(mix[0], run_cfg)
        host_ms = 1e3 * (time.perf_counter() - t0) / 5
        result["host_buffers"] = {"ms_per_step": round(host_ms, 3), "value": round(total_seconds / (host_ms * 1e-3), 2),
                                  "note": "css_run from/to pageable host memory (PCIe-inclusive)"}
        # ... and between the wav edges (css_run_pcm16: int16 planes up, peak-normalised PCM16 down, converted on the GPU)
        planes = [np.ascontiguousarray(np.clip(np.rint(mix[0, :, c] * 0.05 * 32768.0), -32768, 32767).astype(np.int16)) for c in range(7)]
        h.run_pcm16(planes, run_cfg)
        t0 = time.perf_counter()
        for _ in range(5):
            h.run_pcm16(planes, run_cfg)
        p16_ms = 1e3 * (time.perf_counter() - t0) / 5
        result["pcm16_edges"] = {"ms_per_step": round(p16_ms, 3), "value": round(total_seconds / (p16_ms * 1e-3), 2),
                                 "note": "css_run_pcm16: 7 int16 planes from host memory -> 3 PCM16 streams in host memory"}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(mix, state, min(args.cpu_baseline_seconds, total_seconds),
                                                  {"activity_th": 0.3})
    if rank == 0:
        print(json.dumps(result), flush=True)
    sep.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
