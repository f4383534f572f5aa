#!/bin/bash
# Collects everything profiles/ holds for one round (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r04
# -> gpurun_out/<tag>_* ; copy the summaries into profiles/ afterwards.
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out
mkdir -p $o
# the -m gpu suite first (writes gpurun_out/parity_coverage.json: how much of each comparison was covered)
rm -f $o/parity_coverage.json $o/split_vs_f64.md
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -6 > $o/${tag}_pytest_gpu.txt
[ -f $o/parity_coverage.json ] && cp $o/parity_coverage.json $o/${tag}_parity_coverage.json
[ -f $o/split_vs_f64.md ] && cp $o/split_vs_f64.md $o/${tag}_split_vs_f64.md
python bench.py --steps 20 --warmup 3 > $o/${tag}_bench.json 2> $o/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -o p -- python bench.py --steps 12 --warmup 3 --min-seconds 1 --no-cpu-baseline --no-long > $o/${tag}_bench_under_rocprof.json 2>/dev/null
python tools/summarize_prof.py $o/${tag}_prof/p_kernel_trace.csv > $o/${tag}_kernel_stats.md
cp $o/${tag}_prof/p_kernel_stats.csv $o/${tag}_rocprofv3_kernel_stats.csv
# PMC: separate passes, --kernel-trace only (FETCH_SIZE and WRITE_SIZE cannot share a pass)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $o/${tag}_pmc1 -o p -- python bench.py --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-long > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/${tag}_pmc2 -o p -- python bench.py --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-long > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/${tag}_pmc3 -o p -- python bench.py --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-long > /dev/null 2>&1
CSS_TRAFFIC_JSON=$o/${tag}_gemm_traffic.json python tools/summarize_pmc.py $o/${tag}_pmc1 $o/${tag}_pmc2 $o/${tag}_pmc3 > $o/${tag}_pmc.md
python tools/parity_margins.py > $o/${tag}_parity_margins.txt 2>/dev/null
python tools/shard_overhead_probe.py 60 > $o/${tag}_shard_overhead.md 2>/dev/null
python tools/shard_overhead_probe.py 1800 >> $o/${tag}_shard_overhead.md 2>/dev/null
# RCCL at world 1: communicator, census and every collective of the sharded path on this one GPU
CSS_BENCH_FORCE_SHARDED=1 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $o/${tag}_rccl_world1.json 2> $o/${tag}_rccl_world1.log
# the N > 1 code path as separate processes on this one GPU (RCCL refuses two ranks per device: gloo carries the pieces)
for w in 2 8; do
  # (started PLAINLY, as the driver starts benches: bench.py spawns its own ranks)
  CSS_BENCH_ONE_DEVICE=1 CSS_BENCH_BACKEND=gloo CSS_BENCH_CHECK=1 timeout 1200 python bench.py --gpus $w --steps 2 --warmup 1 > $o/${tag}_multiprocess_w$w.log 2>&1
done
rm -rf $o/${tag}_prof $o/${tag}_pmc1/*/*.db $o/${tag}_pmc2/*/*.db $o/${tag}_pmc3/*/*.db 2>/dev/null
cat $o/${tag}_pytest_gpu.txt; head -c 1500 $o/${tag}_bench.json; echo; head -30 $o/${tag}_pmc.md; tail -12 $o/${tag}_pmc.md; cat $o/${tag}_shard_overhead.md; tail -3 $o/${tag}_multiprocess_w2.log; tail -3 $o/${tag}_multiprocess_w8.log; head -c 600 $o/${tag}_rccl_world1.json
