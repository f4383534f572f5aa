#!/usr/bin/env python3
"""CSS real-time-factor benchmark (BASELINE.json metric) on 1..8 MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the CSS hot path (css/css.py::separate_and_stitch equivalent: STFT -> Conformer mask
estimator -> WTA/SCM/MVDR -> PIT stitch -> activity gate -> iSTFT) over one synthetic 7-channel 16 kHz meeting,
timed from the PCM in (page-locked) HOST memory to the separated waveforms in host memory: both PCIe legs are
inside the timed region (SURVEY.md 8(d) "Metric").  `value` = audio seconds separated per wall second.  At N = 1 the K
steps are K sessions in a queue (css_run_enqueue ... css_wait: a session's PCIe legs run under its neighbours' kernels);
the same K sessions as K synchronous css_run calls are the `synchronous_call` key of the line.

  N = 1   BASELINE.json configs[1]: the 60 s meeting (40 segments).  The same line also carries the 30-min meeting of
          configs[3] on this one GPU (`meeting_1800s`), the device-resident figure, the exact-float32 arithmetic mode,
          the roofline of the dominant kernel, the memory-bound kernels' GB/s, and the CPU baseline.
  N > 1   BASELINE.json configs[3]: ONE fixed 1800 s meeting (1209 segments), strong-scaled: sharded by sliding-window
          segment across the N ranks with the three small RCCL all-gathers of notsofar1-challenge_amd/parallel.py (PIT
          costs, activity bits, one 256-sample seam block per stream).  Every rank uploads only the samples of its own
          segments, piece by piece under the stages, and finishes and downloads only its own range of the result.
          Rank 0 also times the same meeting alone on its GPU after the timed region (`single_gpu_same_workload`), so
          that every line compares like with like.

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
PEAK_HBM_GBS = 8000.0            # same guide: HBM3E spec (6.3 TB/s is what a float4 copy reaches)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_REAL_STDOUT = None


def keep_stdout_for_the_record():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL 2.26 prints a version banner through C
    stdio, flushed at exit -- i.e. AFTER the record): from here on file descriptor 1 is stderr, and emit_record() writes
    the line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_record(obj):
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, line)


def pkg(name):
    return importlib.import_module("notsofar1_challenge_amd." + name)


def gpu_clocks():
    """sclk / mclk / power of GPU 0 as rocm-smi reports them (None when the tool is missing): the boxes of the pool differ"""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "power" in kl or "fclk" in kl:
                keep[k] = v
        return keep
    except Exception:  # pragma: no cover
        return None


def sclk_mhz():
    """the shader clock rocm-smi reports right now (MHz; None when unavailable) -- printed next to every roofline fraction:
    the matrix peaks assume 2.4 GHz, the sustained clock under the MFMA kernels is lower"""
    import re
    c = gpu_clocks() or {}
    for k, v in c.items():
        if "sclk" in k.lower():
            m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
            if m:
                return int(m.group(1))
    return None


def gemm_roofline(t, mode, ks_ref=None):
    # `achieved` counts ALGORITHMIC flops (2*M*N*K of the float32 products the network defines).
    achieved = t["gemm_flops"] / (t["gemm_ms"] * 1e-3) / 1e12 if t["gemm_ms"] > 0 else 0.0
    if mode == "split_f16":
        # every product is three f16 MFMAs (hi*hi + hi*lo + lo*hi, f32 accumulate): the ceiling for float32-grade
        # products on the f16 pipes is the dense f16 peak / 3, and achieved / that ceiling equals
        # executed-MFMA-flops / dense f16 peak (the matrix-core utilisation).
        peak = PEAK_F16_MATRIX_TFLOPS / 3.0
        kernel = "css::gemm_split_wd_kernel (3 x v_mfma_f32_32x32x16_f16 per product, weights direct)"
        extra = {"mfma_executed_tflops": round(3 * achieved, 2), "mfma_dense_peak_tflops": PEAK_F16_MATRIX_TFLOPS,
                 "vs_f32_matrix_peak": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 3)}
    else:
        peak = PEAK_FP32_MATRIX_TFLOPS
        kernel = "css::gemm_f32_kernel (v_mfma_f32_32x32x2_f32; persistent, four 4-wave blocks per CU, tiles of 32..128 x 128 x 16)"
        extra = {}
    out = {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": round(peak, 1),
           "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
           "launches_per_step": int(t["gemm_launches"]),
           "avg_launch_us": round(1e3 * t["gemm_ms"] / max(t["gemm_launches"], 1), 2),
           "flops_per_step": t["gemm_flops"], **extra}
    # `achieved` divides by the HIP-event brackets as they are: a bracket also holds the launch gap and the two event
    # records (an EMPTY bracket behind a 4-byte fill measures `empty_event_bracket_us` on this box).  The kernel trace
    # of the same command (profiles/r03_kernel_stats.md) gives the kernels' own durations: ~3 us less per launch.
    if ks_ref and "event_pair_overhead" in ks_ref:
        out["empty_event_bracket_us"] = round(1e3 * ks_ref["event_pair_overhead"][0] / max(ks_ref["event_pair_overhead"][1], 1), 2)
    return out


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


def hbm_kernel_bytes(plan, desc, T, hop, n, own_planes=False):
    """ALGORITHMIC bytes per pass of the memory-bound kernel families: SURVEY.md 8(d)'s compulsory traffic -- what the
    reference's own stages hand to each other (PCM, spectra, features, masks, covariances, beamformer weights, separated
    spectra, waveforms), each read and written once.  own_planes=True: the bytes THIS implementation moves where it
    differs (round 4: the analysis transform also writes one phase plane per microphone, and a segment's features read the
    two planes of microphone 0 plus the C phase planes instead of all 2 C planes) -- listed beside, never as the roofline's
    numerator."""
    C, F, S = desc.num_mics, desc.num_bins, desc.num_spks
    nseg, TL = int(plan.num_segments), int(plan.mix_frames)
    Kp = (desc.in_features + 31) // 32 * 32
    KIp = (2 * F + 31) // 32 * 32
    planes_seg = C * 2 * F * T * 4
    masks_seg = (S + 1) * F * T * 4
    if own_planes:
        return {"stft": n * C * 4 + C * 3 * F * TL * 4, "features": nseg * ((2 + C) * F * T * 4 + T * Kp * 4)}
    return {
        "deinterleave": 2 * n * C * 4,
        "stft": n * C * 4 + C * 2 * F * TL * 4,
        "features": nseg * (planes_seg + T * Kp * 4),
        "scm": nseg * (planes_seg + masks_seg + (S + 1) * F * 49 * 8),
        "mvdr_solve": nseg * ((S + 1) * F * 49 * 8 + S * F * C * 16),
        "beamform": nseg * (planes_seg + S * F * C * 16 + S * F * T * 4 + S * F * T * 8),
        "pit": max(nseg - 1, 0) * 2 * S * F * (T - hop) * 4,
        "ola_masks": TL * (2 * S * F * 4 + S * F * 4),
        "ola_stft": TL * (2 * S * F * 8 + S * KIp * 4),
        "istft_gemm": S * TL * KIp * 4 + S * TL * desc.frame_len * 4,
        "wave_ola": S * TL * desc.frame_len * 4 + S * int(plan.n_out) * 4,
    }


def spawn_ranks(n):
    """`python bench.py --gpus N` started plainly: start N ranks of this same script, one per GPU, with the environment
    torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT -- what the
    reference's own bootstrap reads, utils/torch_utils.py:102-113).  Rank 0 inherits stdout (the ONE JSON line), every
    rank inherits stderr.  Returns the exit code: the first failing rank's, after the others have been stopped."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's intra-node transport needs it here
        env.setdefault("OMP_NUM_THREADS", str(max((os.cpu_count() or 8) // n, 1)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    log(f"[launcher] rank {procs.index(p)} exited with code {code}: stopping the other ranks")
                    for q in live:
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def rank_census(torch, dist, backend, rank, world, local_rank, comm_dev, dry=False):
    """Evidence that the collective library really joined `world` ranks on distinct devices: every rank's identity is
    all-gathered over the process group itself, and a tensor collective runs on the communication device."""
    ident = {"rank": rank, "local_rank": local_rank, "pid": os.getpid()}
    if not dry:
        pr = torch.cuda.get_device_properties(local_rank)
        ident.update({"device": pr.name, "uuid": str(getattr(pr, "uuid", "")),
                      "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                                 getattr(pr, "pci_device_id", 0))})
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    mine = torch.full((1,), float(rank), device=comm_dev)
    if backend == "nccl":
        seen = torch.empty((world,), device=comm_dev)
        dist.all_gather_into_tensor(seen, mine)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        seen = torch.cat(parts)
    assert seen.cpu().tolist() == [float(r) for r in range(world)], seen
    ev = {"backend": backend, "library": "RCCL (torch.distributed backend 'nccl' on ROCm)" if backend == "nccl" else backend,
          "world": world, "ranks_seen_by_all_gather": [int(x) for x in seen.cpu().tolist()], "ranks": idents}
    if backend == "nccl":
        try:
            ev["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # pragma: no cover
            ev["rccl_version"] = f"unavailable ({e})"
    if not dry:
        ids = sorted({i["uuid"] or i["pci"] for i in idents})
        ev["distinct_devices"] = len(ids)
        ev["device_ids"] = ids
    return ev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=60.0, help="N = 1: length of the headline meeting (configs[1])")
    ap.add_argument("--long-seconds", type=float, default=1800.0, help="the strong-scaling meeting (configs[3])")
    ap.add_argument("--max-batch", type=int, default=256,
                    help="segments per batched mask-estimator pass (the split-f16 mode keeps its batches to 128 segments of 3 s "
                         "by itself: css_set_tuning split_batch_rows)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=60.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long", action="store_true", help="N = 1: skip the 30-min meeting")
    ap.add_argument("--lanes", type=int, default=0, help="kernel chains per mask-estimator batch (0 = the library's default)")
    ap.add_argument("--queue-group", type=int, default=0,
                    help="queued sessions merged into one mask-estimator batch (css_set_queue_group; 0 = the library's default, 1 = none)")
    ap.add_argument("--min-seconds", type=float, default=5.0,
                    help="N = 1: the timed regions (exactly K steps each) are repeated until this much has been timed in total")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=VALUE",
                    help="css_set_tuning options of the handle (A/B runs), e.g. --tune tail_pieces=2")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (how the driver starts a bench): this process becomes the launcher of N ranks
        raise SystemExit(spawn_ranks(args.gpus))
    keep_stdout_for_the_record()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # functional-test knobs (never used by the driver): CSS_BENCH_BACKEND=gloo exchanges through host memory,
    # CSS_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 (RCCL refuses two ranks on one device, gloo does not: a 1-GPU box
    # can then exercise the N > 1 code path); CSS_BENCH_CHECK=1 compares the sharded result with the fused run;
    # CSS_BENCH_DRY=1 stops after the rendezvous and the rank / device census (no GPU needed: tests/test_bench_spawn.py)
    # CSS_BENCH_FORCE_SHARDED=1 takes the N > 1 path with whatever world there is -- with one rank it is the RCCL smoke a
    # 1-GPU box can run: communicator creation on the device, every collective of the path at world 1 (profiles/)
    backend = os.environ.get("CSS_BENCH_BACKEND", "nccl")
    dry = os.environ.get("CSS_BENCH_DRY") == "1"
    sharded = world > 1 or os.environ.get("CSS_BENCH_FORCE_SHARDED") == "1"
    if sharded and world == 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            free_port = sk.getsockname()[1]
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(free_port)), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if dry:
        backend = "gloo"
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    if os.environ.get("CSS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(local_rank)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    evidence = None
    if sharded:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        evidence = rank_census(torch, dist, backend, rank, world, local_rank, comm_dev, dry)
    if dry:
        if rank == 0:
            emit_record({"dry_run": True, "n_gpus": world, "collective": evidence})
        if sharded:
            dist.barrier()
            dist.destroy_process_group()
        return

    W, SYN, CSS, SEP, PAR, L = pkg("weights"), pkg("synth"), pkg("css"), pkg("separator"), pkg("parallel"), pkg("_lib")
    desc = W.ModelDesc.mc_v1()
    cal = np.load(os.path.join(ROOT, "tests", "golden", "calib_mc.npz"))
    state = W.apply_golden_recipe(W.portable_state_dict(desc, 0), head_bias=cal["head_bias"])
    cfg = CSS.CssCfg(activity_th=0.3, show_progressbar=False)  # configs/inference/inference_v1.yaml
    run_cfg = CSS.make_run_cfg(cfg, 16000, 7, desc.frame_len, desc.frame_hop)
    T, hop = int(run_cfg.c.segment_frames), int(run_cfg.c.hop_frames)
    S = desc.num_spks

    def meeting(seconds):
        t0 = time.time()
        m = SYN.synth_meeting(seconds, 7, seed=1)  # [1, n, 7]; identical on every rank
        log(f"[rank {rank}] synthetic meeting {seconds:g} s generated in {time.time() - t0:.1f} s")
        return m

    def fused_host_to_host(h, pcm_pinned, out_pinned, steps, warmup):
        """css_run: page-locked host PCM -> page-locked host waveforms (both PCIe legs timed); ms per step"""
        for _ in range(warmup):
            h.run(pcm_pinned, run_cfg, out=out_pinned)
        h.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            h.run(pcm_pinned, run_cfg, out=out_pinned)
        h.sync()
        return 1e3 * (time.perf_counter() - t0) / steps

    dtype_of = {"split_f16": "split-f16x3 (f32 values carried as two f16 numbers: 22-bit operands, three v_mfma_f32_32x32x16_f16 per product, f32 "
                             "accumulate -- operands NARROWER than the reference's f32; MVDR covariance / solve f64)",
                "exact_f32": "f32 (float32 operands on v_mfma_f32_32x32x2_f32 in every Linear layer, attention product and transform: the "
                             "reference's own operand precision; MVDR covariance / solve f64)"}
    result = {"metric": "CSS real-time-factor (sep. audio sec/wall sec) on 7-ch 16 kHz", "unit": "x real-time (audio s / wall s)",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
              "data": "synthetic"}

    # ================================================================================================ N > 1
    if sharded:
        seconds = args.long_seconds
        mix = meeting(seconds)
        n = mix.shape[1]
        plan = L.plan(desc, run_cfg, n)
        nseg, n_out = int(plan.num_segments), int(plan.n_out)
        per_rank = -(-nseg // world) + 1
        # the handle works on a stream torch owns: collectives, packing copies and the kernels are ordered on it
        ts = torch.cuda.Stream(device=dev)
        sep = SEP.HipSeparator(state, None, device=local_rank, max_batch_segments=max(args.max_batch, per_rank),
                               stream=int(ts.cuda_stream))
        h = sep.handle
        be = PAR.HipShardBackend(h, dev, comm_dev, torch_stream=ts)
        me = PAR.make_shard_plan(nseg, int(plan.mix_frames), int(plan.stft_frames), T, hop, desc.frame_hop, rank, world)
        s_lo, s_hi = me.pcm_range(desc.frame_len, n)
        # this rank's samples and its range of the result, in page-locked host memory
        pcm_slice = L.pinned_copy(np.ascontiguousarray(mix[0, s_lo:s_hi]))
        o_lo = me.sample_lo
        o_hi = (n_out if me.t_hi == int(plan.mix_frames) else me.t_hi * desc.frame_hop) if me.num_frames else o_lo
        out_host = torch.empty((S, max(o_hi - o_lo, 1)), dtype=torch.float32, pin_memory=True)
        full_dev = torch.empty((S, n_out), dtype=torch.float32, device=dev)       # gather="all": every rank's HBM
        full_host = torch.empty((S, n_out), dtype=torch.float32, pin_memory=True) if rank == 0 else None

        # the rank's samples cross PCIe in growing pieces; all but the first hide under the stages of the pieces before
        groups, cuts = PAR.upload_schedule(me, T, hop, desc.frame_len, n)

        # Three ways to end a step (all start from this rank's page-locked PCM slice):
        #   "all"        north_star's "RCCL all-gather to stitch the separated streams": the ranks' waveform shards are
        #                all-gathered (12 B per sample) and joined, every GPU then holds the complete separated streams in
        #                HBM (what css.py:110 returns), and each rank writes its own range of them to host memory -- the
        #                ranks' page-locked buffers together are the meeting's result.  THIS IS `value`.
        #   "range"      only one 256-sample seam block per stream crosses between neighbours; each rank finishes and
        #                downloads its own range (the result in host memory is the same, no GPU holds all of it).
        #   "all_rank0"  as "all", but ONE process (rank 0) downloads the complete waveforms (346 MB over one PCIe link).
        def step(mode, trace=None):
            mark = trace or (lambda label: None)
            mark("start")
            be.begin(pcm_slice, n, 7, run_cfg, sample_range=(s_lo, s_hi), slice_only=True, cuts=cuts)
            mark("upload_first_piece")
            if mode == "range":
                own, rng = PAR.sharded_separate_and_stitch(be, S, T, hop, desc.frame_hop, rank, world, dist, gather="range",
                                                           segment_groups=groups, trace=trace, check_range=False)
                assert rng == (o_lo, o_hi), (rng, o_lo, o_hi)
            else:
                full = PAR.sharded_separate_and_stitch(be, S, T, hop, desc.frame_hop, rank, world, dist, gather="all",
                                                       out=full_dev, segment_groups=groups, trace=trace, check_range=False)
                own = full[:, o_lo:o_hi]
            with be.on_stream():
                if mode == "all_rank0":
                    if rank == 0:
                        full_host.copy_(full, non_blocking=True)
                elif o_hi > o_lo:
                    out_host[:, :o_hi - o_lo].copy_(own, non_blocking=True)
            mark("download")

        def barrier():
            h.sync()
            torch.cuda.synchronize()
            dist.barrier()

        def timed_steps(mode, steps, warmup):
            """W warm-up steps, then exactly K steps between two barriers (+ device synchronisation); max over ranks.
            The host enqueues a step much faster than the device runs it; it stays at most two steps ahead (an unbounded
            lead only makes the runtime grow its command and signal pools inside the timed region)."""
            done = []

            def paced():
                if len(done) >= 2:
                    done[-2].synchronize()
                step(mode)
                ev = torch.cuda.Event()
                ev.record(ts)
                done.append(ev)

            for _ in range(warmup):
                paced()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                paced()
            barrier()
            el = time.perf_counter() - t0
            tmax = torch.tensor([el], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            h.check_range()   # (the staged path has no automatic exact-float32 repeat: fail loudly instead)
            return float(tmax.item())

        def rec_n(el, steps, note):
            return {"ms_per_step": round(1e3 * el / steps, 3), "value": round(seconds * steps / el, 2), "note": note}

        # the headline is the library's default arithmetic = the reference's own (float32 operands), as at N = 1; the opt-in
        # split-f16 mode is timed first and reported beside it (`split_f16`, `value_split_f16`)
        h.set_linear_mode("split_f16")
        step("all"); barrier()   # initialisation, not a step: device buffers, communicator channels, index maps
        el_split = timed_steps("all", args.steps, args.warmup)
        assert torch.isfinite(out_host).all()
        h.sync(); barrier()
        h.set_linear_mode("exact_f32")
        step("all"); barrier()
        el_all = timed_steps("all", args.steps, args.warmup)
        assert torch.isfinite(out_host).all()
        el_range = timed_steps("range", args.steps, max(args.warmup, 1))
        assert torch.isfinite(out_host).all()
        el_r0 = timed_steps("all_rank0", max(args.steps // 2, 3), 1)

        # ---- one instrumented step per mode: device-side time between the phases (events on the handle's stream)
        def phases(mode):
            marks = []

            def trace(label):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(ts)
                marks.append((label, ev))

            barrier()
            step(mode, trace)
            barrier()
            names = [m[0] for m in marks[1:]]
            ms = torch.tensor([a[1].elapsed_time(b[1]) for a, b in zip(marks, marks[1:])], dtype=torch.float64, device=comm_dev)
            allms = [torch.empty_like(ms) for _ in range(world)]
            dist.all_gather(allms, ms)
            allms = torch.stack(allms).cpu().numpy()
            return {"phases": names, "max_over_ranks_ms": [round(float(x), 3) for x in allms.max(axis=0)],
                    "rank0_ms": [round(float(x), 3) for x in allms[0]],
                    "sum_of_max_ms": round(float(allms.max(axis=0).sum()), 3)}

        phase_ms = {"all": phases("all"), "range": phases("range")}

        # ---- roofline of the dominant kernel on THIS workload: one more step with the library's per-launch HIP-event
        # brackets on (every rank takes the step -- it holds collectives --, rank 0 reports its own shard's launches;
        # one lane: the brackets need one ordered stream, as at N = 1)
        barrier()
        h.set_profile(True)
        step("all")
        barrier()
        t_prof, ks_prof = h.timings(), h.kernel_stats()
        h.set_profile(False)
        roof = gemm_roofline(t_prof, h.linear_mode(), ks_prof)
        roof["measured_on"] = f"rank 0's shard ({me.seg_hi - me.seg_lo} segments) during one sharded step"

        # the sharded result equals the fused single-GPU run, bit for bit -- checked by EVERY N > 1 run after its timed regions
        # (CSS_BENCH_CHECK=0 skips it): the first real multi-GPU run is also a correctness run
        if os.environ.get("CSS_BENCH_CHECK", "1") != "0":
            ref = h.run(np.ascontiguousarray(mix[0]), run_cfg)
            step("range"); barrier()
            same_own = bool(np.array_equal(ref[:, o_lo:o_hi], out_host[:, :o_hi - o_lo].numpy()))
            step("all"); barrier()
            same_own_all = bool(np.array_equal(ref[:, o_lo:o_hi], out_host[:, :o_hi - o_lo].numpy()))
            same_all = bool(np.array_equal(ref, full_dev.cpu().numpy()))
            same = same_own and same_own_all and same_all
            log(f"[rank {rank}] sharded == fused single-GPU result, bit for bit: own range [{o_lo}, {o_hi}) via seams "
                f"{same_own}, via the all-gather {same_own_all}, the all-gathered whole {same_all}: {same}")
            assert same
            ok = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            result["parity_checked"] = bool(ok.item() == 1.0)
            result["parity"] = {"sharded_equals_fused_single_gpu_bit_for_bit_on_every_rank": result["parity_checked"],
                                "compared": "each rank's own range via the seam exchange, via the waveform all-gather, and the all-gathered whole"}
        ms_all = 1e3 * el_all / args.steps
        result.update({
            "value": round(seconds * args.steps / el_all, 2), "ms_per_step": round(ms_all, 3),
            "scaling": "strong", "dtype": dtype_of[h.linear_mode()],
            "config": {"workload": f"synthetic 7-ch 16 kHz {seconds:g} s meeting ({nseg} segments of 3 s / 1.5 s hop), "
                                   f"Conformer-CSS v1.0-MC (18 blocks, D=512) + MVDR, host PCM -> host waveforms",
                       "segments": nseg, "segments_per_rank": me.seg_hi - me.seg_lo,
                       "sharding": f"{world} ranks x segment ranges (one halo segment per seam), each rank uploads its own "
                                   f"samples; all-gathers over {'RCCL' if backend == 'nccl' else backend}: PIT costs (72 B / "
                                   f"boundary), activity bits (3 B / frame), then the separated waveform shards (12 B / sample: "
                                   f"every GPU ends with the complete streams in HBM); each rank writes its own range of the "
                                   f"result to page-locked host memory",
                       "gather": "all"},
            "value_is": "gather_all (waveform shards all-gathered over the process group and joined on every GPU; the ranks' "
                        "host buffers together hold the result)",
            "gather_range": rec_n(el_range, args.steps, "one 256-sample seam block per stream crosses between neighbours instead "
                                                        "of the waveform all-gather; same result in host memory"),
            "gather_all_rank0_download": rec_n(el_r0, max(args.steps // 2, 3), "as `value`, but rank 0 alone downloads the complete "
                                                                               "waveforms (S x n_out x 4 B over one PCIe link)"),
            "phase_ms": phase_ms,
            # what the step time SHOULD be from its parts -- the slowest rank's segments + the three exchanges + its download
            # (device time between events on the handle's stream, one instrumented step): a `value` well below this
            # projection is host-side or launch-side overhead, one at it is the sum of the phases
            "projected_from_phase_ms": {
                "ms_per_step": phase_ms["all"]["sum_of_max_ms"],
                "value": round(seconds / (phase_ms["all"]["sum_of_max_ms"] * 1e-3), 2),
                "measured_over_projected": round(ms_all / max(phase_ms["all"]["sum_of_max_ms"], 1e-9), 3),
                "largest_phase": max(zip(phase_ms["all"]["max_over_ranks_ms"], phase_ms["all"]["phases"]))[1],
                "note": "sum over the phases of the maximum over ranks (upload of the first piece, segments, cost exchange, scan + "
                        "mask overlap-add, activity exchange, gate + synthesis, waveform exchange, download)"},
            "collective": evidence,
            "roofline": roof,
            "clocks": gpu_clocks(),
            "value_is_arithmetic": "CSS_LINEAR_EXACT_F32 (float32 operands): the library's default and the reference's operand precision; the opt-in split-f16 mode: `split_f16`",
            "split_f16": {**rec_n(el_split, args.steps, "the same gather_all step in the opt-in split-f16 mode (22-bit operands)"),
                          "dtype": dtype_of["split_f16"]},
            "value_split_f16": round(seconds * args.steps / el_split, 2), "dtype_split_f16": dtype_of["split_f16"],
        })
        if rank == 0:
            # the same meeting alone on this rank's GPU, host to host (what N = 1 would print for this workload)
            # (its own handle with the N = 1 line's batch size: the sharded handle holds a rank's whole shard in one batch)
            pcm_all = L.pinned_copy(np.ascontiguousarray(mix[0]))
            out_all = L.pinned_empty((S, n_out), np.float32)
            sep1 = SEP.HipSeparator(state, None, device=local_rank, max_batch_segments=args.max_batch)
            try:
                ms1 = fused_host_to_host(sep1.handle, pcm_all, out_all, 3, 1)     # like with like: the headline's arithmetic
                sep1.handle.set_linear_mode("split_f16")
                ms1_split = fused_host_to_host(sep1.handle, pcm_all, out_all, 3, 1)
            finally:
                sep1.close()
            result["single_gpu_same_workload"] = {"ms_per_step": round(ms1, 3), "value": round(seconds / (ms1 * 1e-3), 2),
                                                  "speedup": round(ms1 / ms_all, 3),
                                                  "speedup_gather_range": round(ms1 / (1e3 * el_range / args.steps), 3)}
            result["speedup_vs_1gpu_same_workload"] = round(ms1 / ms_all, 3)
            result["split_f16"]["single_gpu_same_workload"] = {"ms_per_step": round(ms1_split, 3), "value": round(seconds / (ms1_split * 1e-3), 2),
                                                                "speedup": round(ms1_split / (1e3 * el_split / args.steps), 3)}
        dist.barrier()
        if rank == 0:
            emit_record(result)
        be.close()
        sep.close()
        dist.destroy_process_group()
        return

    # ================================================================================================ N = 1
    seconds = args.seconds
    mix = meeting(seconds)
    n = mix.shape[1]
    plan = L.plan(desc, run_cfg, n)
    sep = SEP.HipSeparator(state, None, device=local_rank, max_batch_segments=args.max_batch)
    h = sep.handle
    if args.lanes:
        h.set_lanes(args.lanes)
    if args.queue_group:
        h.set_queue_group(args.queue_group)
    result["clocks_at_start"] = gpu_clocks()
    for kv in args.tune:
        name, _, val = kv.partition("=")
        h.set_tuning(name, int(val))
        result.setdefault("tuning", {})[name] = int(val)
    pcm_pin = L.pinned_copy(np.ascontiguousarray(mix[0]))
    out_pin = L.pinned_empty((S, int(plan.n_out)), np.float32)
    out_pin2 = L.pinned_empty((S, int(plan.n_out)), np.float32)
    outs = (out_pin, out_pin2)
    pcm_dev = torch.from_numpy(np.ascontiguousarray(mix[0])).to(dev)
    wav_dev = torch.empty((S, plan.n_out), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    group_limit = args.queue_group if args.queue_group else 8
    split_rows = dict(kv.partition("=")[::2] for kv in args.tune).get("split_batch_rows")
    split_rows = int(split_rows) if split_rows is not None else 24576

    def sessions_in_a_batch(mode):
        """sessions whose segments share one estimator batch in this arithmetic mode (api.hip batch_cap)"""
        cap = args.max_batch
        if mode == "split_f16" and split_rows > 0:
            cap = min(cap, max(1, split_rows // T))
        return max(1, min(group_limit, cap // int(plan.num_segments), args.steps))

    def queued(k_steps):
        for k in range(k_steps):
            h.run_enqueue(pcm_pin, run_cfg, outs[k % 2])
        h.wait()

    def timed_region():
        """exactly K steps between two synchronisations"""
        h.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        queued(args.steps)
        h.sync(); torch.cuda.synchronize()
        return time.perf_counter() - t0

    def timed(fn, steps=args.steps, warmup=2):
        for _ in range(warmup):
            fn()
        h.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        h.sync(); torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    def rec(ms, note):
        return {"ms_per_step": round(ms, 3), "value": round(seconds / (ms * 1e-3), 2), "note": note}

    def profiled_pass():
        """one session alone under the per-launch profile (device-resident, one lane)"""
        h.set_profile(True)
        h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, wav_dev.data_ptr(), plan.n_out)
        h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, wav_dev.data_ptr(), plan.n_out)
        t, ks = h.timings(), h.kernel_stats()
        h.set_profile(False)
        return t, ks

    def profiled_queue(k_sessions):
        """the headline's own schedule under the per-launch profile: k queued sessions = one estimator batch"""
        h.wait()
        h.set_profile(True)
        for _ in range(2):
            for k in range(k_sessions):
                h.run_enqueue(pcm_pin, run_cfg, outs[k % 2])
            h.wait()
        t, ks = h.timings(), h.kernel_stats()
        h.set_profile(False)
        return t, ks

    def attach_traffic(roof, mode):
        # HBM bytes per launch of that kernel: PMC counters cannot be read inside this process, so the figure comes from
        # the committed PMC passes of the same command (profiles/README.md), when present
        suffix = "" if mode == "split_f16" else "_exact_f32"
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tj = os.path.join(ROOT, "profiles", f"{tag}_gemm_traffic{suffix}.json")
            if not os.path.exists(tj):
                continue
            with open(tj) as f:
                tr = json.load(f)
            key = f"M{roof.get('rows_per_launch', int(plan.num_segments) * T)}"
            pick = tr.get(key, tr)
            if "traffic_bytes_per_launch" in pick:
                roof["traffic"] = round(pick["traffic_bytes_per_launch"])
                if "algorithmic_bytes_per_launch" in pick:
                    roof["algorithmic_bytes_per_launch"] = round(pick["algorithmic_bytes_per_launch"])
                    roof["traffic_over_algorithmic"] = round(pick["traffic_bytes_per_launch"] / pick["algorithmic_bytes_per_launch"], 3)
                roof["traffic_source"] = (f"profiles/{tag}_gemm_traffic{suffix}.json [{key if key in tr else 'launch mix of that round'}] (rocprofv3 "
                                          f"--pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, bytes per launch)")
            break
        return roof

    def headline(mode, min_seconds):
        """The metric in one arithmetic mode: K queued sessions, each host -> host (css_run_enqueue ... css_wait), exactly K
        steps per timed region, regions repeated until `min_seconds` have been timed, the MEDIAN region is the value; the
        same sessions as synchronous calls and device-resident; the dominant kernel's roofline on the headline's own
        schedule (the shared estimator batch) and on one session alone."""
        h.set_linear_mode(mode)
        sessions_per_batch = sessions_in_a_batch(mode)
        h.run(pcm_pin, run_cfg, out=out_pin)   # initialisation, not a step: the handle sizes its device buffers on first use
        queued(args.warmup)
        runs = [timed_region()]
        while sum(runs) < min_seconds and len(runs) < 200:
            runs.append(timed_region())
        elapsed = float(np.median(runs))
        assert np.isfinite(out_pin).all() and (args.steps < 2 or np.array_equal(out_pin, out_pin2))
        queued_out = (out_pin.copy(), out_pin2.copy())   # what the TIMED queue left in the caller's buffers (checked below)
        out = {"value": round(seconds * args.steps / elapsed, 2), "ms_per_step": round(1e3 * elapsed / args.steps, 3),
               "dtype": dtype_of[mode],
               "runs_ms": {"per_step_ms_of_each_timed_region": [round(1e3 * r / args.steps, 3) for r in runs], "regions": len(runs),
                           "steps_per_region": args.steps, "value_is": "median region", "timed_seconds_in_total": round(sum(runs), 3),
                           "min": round(1e3 * min(runs) / args.steps, 3), "max": round(1e3 * max(runs) / args.steps, 3)}}
        # the same K sessions as K synchronous calls (css_run returns when the waveforms are in host memory)
        ms_sync = timed(lambda: h.run(pcm_pin, run_cfg, out=out_pin), warmup=args.warmup)
        out["synchronous_call"] = rec(ms_sync, "css_run: one session per call, host -> host, the call's own latency (nothing to overlap with)")
        # the timed schedule's outputs against the same session's synchronous css_run (out_pin holds one now): bit for bit
        out["queue_equals_css_run"] = bool(np.array_equal(queued_out[0], out_pin) and (args.steps < 2 or np.array_equal(queued_out[1], out_pin)))
        assert out["queue_equals_css_run"], "the queued sessions of the timed region differ from css_run on the same session"
        del queued_out
        stage = h.timings()
        out["stage_ms"] = {k: round(v, 3) for k, v in stage.items()
                           if k in ("upload", "stft", "masknet", "mvdr", "stitch", "istft", "download", "total", "host_enqueue", "host_total")}
        ms_dev = timed(lambda: h.run_device(pcm_dev.data_ptr(), n, 7, run_cfg, wav_dev.data_ptr(), plan.n_out))
        out["device_resident"] = rec(ms_dev, "css_run_device: PCM and waveforms in HBM (no PCIe leg)")
        out["host_vs_device_resident"] = round(out["ms_per_step"] / ms_dev, 4)
        out["synchronous_vs_device_resident"] = round(ms_sync / ms_dev, 4)
        t1, ks1 = profiled_pass()
        roof_single = gemm_roofline(t1, mode, ks1)
        if sessions_per_batch > 1:
            tq, ksq = profiled_queue(sessions_per_batch)
            roof = gemm_roofline(tq, mode, ksq)
            roof["sessions_per_launch"] = sessions_per_batch
            roof["rows_per_launch"] = sessions_per_batch * int(plan.num_segments) * T
            out["roofline_single_session"] = roof_single
            out["kernel_family_ms_per_session_in_a_shared_batch"] = {k: round(v[0] / sessions_per_batch, 4) for k, v in ksq.items()
                                                                     if k != "event_pair_overhead"}
            out["_ks_queue"] = {k: (v[0] / sessions_per_batch, v[1]) for k, v in ksq.items()}
        else:
            roof = roof_single
        roof["sclk_mhz_after_the_profiled_passes"] = sclk_mhz()
        out["roofline"] = attach_traffic(roof, mode)
        out["kernel_family_ms"] = {k: round(v[0], 4) for k, v in ks1.items() if k != "event_pair_overhead"}
        out["sessions_per_estimator_batch"] = sessions_per_batch
        out["_ks_single"] = ks1
        return out

    # ---- the headline: the reference's own arithmetic (float32 operands on the float32 matrix instruction) ...
    head = headline("exact_f32", args.min_seconds)
    ks_single = head.pop("_ks_single")
    ks_queue = head.pop("_ks_queue", None)
    sessions_per_batch = head["sessions_per_estimator_batch"]
    result.update(head)
    result.update({
        "scaling": "strong",
        "config": {"workload": f"synthetic 7-ch 16 kHz {seconds:g} s meeting ({plan.num_segments} segments of 3 s / 1.5 s "
                               f"hop), Conformer-CSS v1.0-MC (18 blocks, D=512) + MVDR; a queue of sessions, each host PCM -> "
                               f"host waveforms (css_run_enqueue / css_wait, page-locked buffers, every session's two PCIe "
                               f"legs inside the timed region, overlapped with its neighbours' kernels)",
                   "segments": int(plan.num_segments), "sharding": "single GPU", "arithmetic": "CSS_LINEAR_EXACT_F32",
                   "sessions_per_estimator_batch": sessions_per_batch,
                   "estimator_batching": f"queued sessions share mask-estimator batches: {sessions_per_batch} sessions = "
                                         f"{sessions_per_batch * int(plan.num_segments)} segments = M {sessions_per_batch * int(plan.num_segments) * T} rows per "
                                         f"Linear-layer launch (max_batch_segments {args.max_batch}); everything else per session; "
                                         f"each session's result is bit for bit its css_run result"},
        "value_is": "CSS_LINEAR_EXACT_F32 (float32 operands, v_mfma_f32_32x32x2_f32): the library's DEFAULT mode and the reference's "
                    "operand precision.  The opt-in split-f16 mode (22-bit operands, faster): `split_f16` / `value_split_f16` below",
    })
    # ---- parity of the schedule that was just timed: (1) its outputs equal css_run's bit for bit (asserted
    # inside headline()); (2) the SAME handle, mode and queue shape on the screened configs[1] meeting whose every frame the
    # reference fixture covers (tests/golden/e2e60_r5.npz: the reference's own run, css/css.py:110-338): decisions exact,
    # waveforms free-running against what the reference's own rounding-level flips cost it (e2e60_r6_self.npz: the reference at
    # 8 / 4 / 2 / 1 threads is 1.0e-4 from itself through ONE flipped winner-take-all decision; tests/test_hip_headline.py holds
    # the same schedule to the same fixture with the flips located)
    def parity_of_the_timed_schedule():
        gdir = os.path.join(ROOT, "tests", "golden")
        g, g6 = np.load(os.path.join(gdir, "e2e60_r5.npz")), np.load(os.path.join(gdir, "e2e60_r6_self.npz"))
        n22 = int(g["mix_samples"])
        mix22 = SYN.synth_meeting(n22 / 16000.0, 7, seed=int(g["mix_seed"]))[:, :n22]
        pcm22 = L.pinned_copy(np.ascontiguousarray(mix22[0]))
        plan22 = L.plan(desc, run_cfg, n22)
        bufs = [L.pinned_empty((S, int(plan22.n_out)), np.float32) for _ in range(2)]
        ref22 = h.run(pcm22, run_cfg).copy()
        for b in bufs:
            b[:] = np.nan
        for k in range(2 * sessions_per_batch):
            h.run_enqueue(pcm22, run_cfg, bufs[k % 2])
        h.wait()
        same = bool(np.array_equal(bufs[0], ref22) and np.array_equal(bufs[1], ref22))
        bits = lambda packed, shape: np.unpackbits(packed)[:int(np.prod(shape))].reshape(tuple(shape)).astype(bool)
        shape = tuple(g["activity_shape"])
        perms_ok = [tuple(p) for p in h.read(L.BUF_PERMS)[1:]] == [tuple(p) for p in g["pit_perm"]]
        act_ok = bool(np.array_equal(h.read(L.BUF_ACT_B).astype(bool).T, bits(g["activity_b"], shape)) and
                      np.array_equal(h.read(L.BUF_ACT_FINAL).astype(bool).T, bits(g["activity_final"], shape)))
        ms = float(np.abs(h.read(L.BUF_MASK_ST).transpose(1, 2, 0)[::32, ::16] - g["mask_stitched"]).max())

        def rr(a, b):
            a, b = a.astype(np.float64), b.astype(np.float64)
            return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
        free = [rr(bufs[0][k, ::64], g["wav_dec64"][k]) for k in range(S)]
        thr = [int(t) for t in g6["threads"][1:]]
        self_dist = np.max(np.stack([g6[f"wav_rel_rms_dec64_t{t}"] for t in thr]), axis=0)
        bar = [1e-4 + 2 * float(self_dist[k]) for k in range(S)]   # at most two noise-level flips (tests/test_hip_headline.py locates them)
        ok = same and perms_ok and act_ok and ms < 5e-6 and all(free[k] < bar[k] for k in range(S))
        return ok, {"timed_queue_equals_css_run_bit_for_bit": bool(result.get("queue_equals_css_run")),
                    "fixture": "tests/golden/e2e60_r5.npz (the reference's own run of the screened 61 s configs[1] meeting, seed "
                               f"{int(g['mix_seed'])}; 40 segments, every frame comparable)",
                    "schedule": f"{2 * sessions_per_batch} sessions queued on the timed handle, {sessions_per_batch} per estimator batch, "
                                f"max_batch_segments {args.max_batch}, arithmetic {h.linear_mode()}",
                    "queue_equals_css_run_bit_for_bit": same, "permutations_equal_the_reference": bool(perms_ok),
                    "activity_maps_equal_the_reference": act_ok, "stitched_masks_max_abs_vs_reference": ms,
                    "waveform_rel_rms_free_running_vs_reference_whole_meeting": free, "bar": bar,
                    "reference_free_running_self_distance_between_thread_counts": [float(x) for x in self_dist],
                    "on_the_reference_decisions": "<= 1e-4 on every frame: tests/test_hip_headline.py (same handle configuration), "
                                                  "tests/test_hip_golden_r5.py"}

    h.set_linear_mode("exact_f32")
    result["parity_checked"], result["parity"] = parity_of_the_timed_schedule()

    # ---- the memory-bound kernel families of the profiled single-session pass: algorithmic bytes / live HIP-event time
    def hbm_table(alg, ks, own=None):
        rows = []
        for k in alg:
            if k not in ks or ks[k][0] <= 0:
                continue
            row = {"kernel": k, "bytes": int(alg[k]), "us": round(1e3 * ks[k][0], 2), "launches": ks[k][1],
                   "GBps": round(alg[k] / (ks[k][0] * 1e-3) / 1e9, 1),
                   "frac": round(alg[k] / (ks[k][0] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            if own and k in own:
                row["bytes_this_implementation_moves"] = int(own[k])
            rows.append(row)
        return rows

    result["roofline_hbm"] = hbm_table(hbm_kernel_bytes(plan, desc, T, hop, n), ks_single, hbm_kernel_bytes(plan, desc, T, hop, n, True))
    if ks_queue:   # the same families inside the headline's queue: a session's bytes over its share of the group's launches
        result["roofline_hbm_in_a_shared_batch"] = hbm_table(hbm_kernel_bytes(plan, desc, T, hop, n), ks_queue)
    result["roofline_hbm_bytes_are"] = ("SURVEY.md 8(d)'s compulsory bytes: operands the reference's own stages exchange (PCM, spectra, masks, "
                                        "features, covariances, weights, separated spectra, waveforms); planes the implementation adds for "
                                        "itself (the analysis transform's phase planes) are NOT counted")
    pageable = np.ascontiguousarray(mix[0])
    result["pageable_host"] = rec(timed(lambda: h.run(pageable, run_cfg), steps=5, warmup=1),
                                  "css_run from/to ordinary (pageable) host memory: the driver's staged copies")
    planes = [np.ascontiguousarray(np.clip(np.rint(mix[0, :, c] * 0.05 * 32768.0), -32768, 32767).astype(np.int16)) for c in range(7)]
    result["pcm16_edges"] = rec(timed(lambda: h.run_pcm16(planes, run_cfg), steps=5, warmup=1),
                                "css_run_pcm16: 7 int16 planes in host memory -> 3 peak-normalised PCM16 streams in host memory")

    # ---- files -> files: the product's own session loop (pipeline.css_sessions = the CSS leg of inference_pipeline/inference.py:
    # 59-63, css/css.py:51-107) on N sessions of 7 mono PCM16 wav files each -> input_mixture.wav + sep_stream{0,1,2}.wav per
    # session, I/O INSIDE the timed region: wav decode and file writes on worker threads, the sessions through the same queue the
    # headline times (css_run_enqueue_pcm16 / css_wait: both wav edges on the device).  The model stays resident (the reference
    # reloads its checkpoint per session, css.py:85; not imitated, not timed).
    def sessions_from_files(n_sessions, repeat=1):
        import shutil
        import tempfile
        import pandas as pd
        PIPE, WIO = pkg("pipeline"), pkg("wavio")
        tmp = tempfile.mkdtemp(prefix="css_bench_sessions_")
        planes_r = planes if repeat == 1 else [np.ascontiguousarray(np.tile(p, repeat)) for p in planes]   # (a longer session: the same minute again)
        seconds_r = seconds * repeat
        try:
            rows = []
            for i in range(n_sessions):
                names = []
                for c in range(7):
                    p = os.path.join(tmp, f"in_{i:03d}_ch{c}.wav")
                    WIO.write_pcm16_samples(p, planes_r[c], 16000)
                    names.append(p)
                rows.append({"wav_file_names": names, "session_id": f"bench_{i:03d}", "is_mc": True})
            df = pd.DataFrame(rows)
            direct, _ = h.run_pcm16(planes_r, run_cfg)
            times, loop_stats = [], []
            for rep in range(3):
                h.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = {}
                got = PIPE.css_sessions(os.path.join(tmp, f"out{rep}"), "unused: the model is resident", df, cfg, separators={True: sep}, stats=st)
                times.append(time.perf_counter() - t0)
                loop_stats.append({k: round(v, 4) if isinstance(v, float) else v for k, v in st.items()})
            same = True
            for _, row in got.iterrows():
                for i, f in enumerate(row.sep_wav_file_names):
                    y, sr = WIO.read_wav_pcm16(f)
                    same = same and sr == 16000 and bool(np.array_equal(y, direct[i]))
            dt = min(times[1:])
            return {"value": round(n_sessions * seconds_r / dt, 2), "ms_per_session": round(1e3 * dt / n_sessions, 3), "sessions": n_sessions,
                    "seconds_per_session": seconds_r,
                    "runs_s": [round(t, 4) for t in times], "value_is": "best of the two runs after the first (which also sizes the page-locked pools)",
                    "files_equal_css_run_pcm16_bit_for_bit": same, "where_the_wall_time_went": loop_stats,
                    "vs_queue_value": round((n_sessions * seconds_r / dt) / result["value"], 4),
                    "note": f"pipeline.css_sessions: {n_sessions} sessions x (7 mono PCM16 wav files of {seconds_r:g} s in -> input_mixture.wav + 3 "
                            "sep_stream wav files out), wav decode / file writes on 4 worker threads, the sessions queued with css_run_enqueue_pcm16 "
                            "(a rolling window: css_wait_sessions for the oldest 12 of up to 24 in flight), resident model; file system = the "
                            "box's temporary directory"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    h.set_linear_mode("exact_f32")
    result["sessions_from_files"] = sessions_from_files(48)
    # ... and at the session length of the reference's own data (dev-set-1: ~6 min per session): a session fills an estimator batch
    # by itself (240 segments), seven 11.5 MB files in and four out per session
    result["sessions_from_files_6min"] = sessions_from_files(12, repeat=6)

    # ---- ... and the opt-in, faster mode: same workload, same timing rules, its own roofline
    fast = headline("split_f16", min(args.min_seconds, 3.0))
    fast.pop("_ks_single"); fast.pop("_ks_queue", None)
    result["split_f16"] = fast
    result["value_split_f16"] = fast["value"]
    result["dtype_split_f16"] = fast["dtype"]
    result["roofline_split_f16"] = fast["roofline"]
    # (the keys earlier rounds' records used for the float32 figure)
    result["value_exact_f32"] = result["value"]
    result["roofline_exact_f32"] = result["roofline"]

    # ---- BASELINE.json configs[3] on this one GPU: the fixed 30-min meeting every N > 1 line runs
    if not args.no_long:
        del pcm_dev, wav_dev
        long_mix = meeting(args.long_seconds)
        n_long = long_mix.shape[1]
        plan_long = L.plan(desc, run_cfg, n_long)
        pcm_long = L.pinned_copy(np.ascontiguousarray(long_mix[0]))
        # the same kernel families in the THROUGHPUT regime (1209 segments, 128 per estimator batch): GEMM roofline and
        # the memory-bound kernels' GB/s, live HIP events on a device-resident pass (north_star: "rocprof HBM GB/s on
        # STFT / covariance"; the 60 s figures above are launch-latency-bound at 40 segments)
        pcm_dev = torch.from_numpy(np.ascontiguousarray(long_mix[0])).to(dev)
        wav_dev = torch.empty((S, int(plan_long.n_out)), dtype=torch.float32, device=dev)
        del long_mix
        out_long = L.pinned_empty((S, int(plan_long.n_out)), np.float32)
        out_long2 = L.pinned_empty((S, int(plan_long.n_out)), np.float32)
        for mode in ("exact_f32", "split_f16"):
            h.set_linear_mode(mode)
            h.set_profile(True)
            for _ in range(2):
                h.run_device(pcm_dev.data_ptr(), n_long, 7, run_cfg, wav_dev.data_ptr(), int(plan_long.n_out))
            t_l, ks_l = h.timings(), h.kernel_stats()
            h.set_profile(False)
            ms_long = fused_host_to_host(h, pcm_long, out_long, 3, 1)
            assert np.isfinite(out_long[:, ::4096]).all()
            m1800 = {**rec(ms_long, "the strong-scaling workload of the N > 1 lines on ONE GPU, host -> host, one synchronous css_run per meeting"),
                     "value": round(args.long_seconds / (ms_long * 1e-3), 2), "dtype": dtype_of[mode],
                     "segments": int(plan_long.num_segments), "seconds": args.long_seconds,
                     "roofline": gemm_roofline(t_l, mode, ks_l)}
            h.run_enqueue(pcm_long, run_cfg, out_long2); h.wait()
            h.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(4):
                h.run_enqueue(pcm_long, run_cfg, (out_long, out_long2)[k % 2])
            h.wait(); torch.cuda.synchronize()
            ms_q = 1e3 * (time.perf_counter() - t0) / 4
            m1800["queued"] = {"ms_per_step": round(ms_q, 3), "value": round(args.long_seconds / (ms_q * 1e-3), 2),
                               "note": "four such meetings queued (css_run_enqueue / css_wait)"}
            if mode == "exact_f32":
                result["meeting_1800s"] = m1800
                result["roofline_1800s"] = m1800["roofline"]
                result["roofline_hbm_1800s"] = hbm_table(hbm_kernel_bytes(plan_long, desc, T, hop, n_long), ks_l,
                                                         hbm_kernel_bytes(plan_long, desc, T, hop, n_long, True))
            else:
                result["split_f16"]["meeting_1800s"] = m1800
        del pcm_dev, wav_dev, pcm_long, out_long, out_long2

    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(mix, state, min(args.cpu_baseline_seconds, seconds), {"activity_th": 0.3})
    result["clocks_at_end"] = gpu_clocks()
    emit_record(result)
    sep.close()


if __name__ == "__main__":
    main()
