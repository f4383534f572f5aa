#!/bin/bash
# Round profiles (run on the GPU box through gpurun):  bash tools/profile_round.sh <tag e.g. r06> [bench|pmc|small|rest|all]  ->  gpurun_out/<tag>_* ;
# copy the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r06}
part=${2:-all}
if [ "$part" = all ] || [ "$part" = bench ]; then
python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python bench.py --steps 12 --warmup 3 --min-seconds 1 --no-cpu-baseline --no-long > gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
python tools/summarize_prof.py gpurun_out/${tag}_prof/p_kernel_trace.csv > gpurun_out/${tag}_kernel_stats.md
cp gpurun_out/${tag}_prof/p_kernel_stats.csv gpurun_out/${tag}_rocprofv3_kernel_stats.csv
fi
if [ "$part" = all ] || [ "$part" = pmc ]; then
# counters of the headline's own GEMM launches, both arithmetic modes; separate passes, --kernel-trace only
for mode in exact_f32 split_f16; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/${tag}_gpmc_${mode}_$i -o p -- python tools/gemm_traffic.py $mode 3 > gpurun_out/${tag}_gpmc_${mode}_$i.json 2>/dev/null
  done
done
python tools/summarize_gemm_pmc.py gpurun_out ${tag} > gpurun_out/${tag}_gemm_pmc.md
python tools/summarize_estimator_pmc.py gpurun_out ${tag} > gpurun_out/${tag}_estimator_pmc.md
fi
if [ "$part" = all ] || [ "$part" = small ]; then
# the memory-bound kernels either side of the estimator on the 30-MIN meeting (north_star: "rocprof HBM GB/s on STFT / covariance"):
# three separate --pmc passes over two device-resident passes of the 1800 s meeting (FETCH_SIZE and WRITE_SIZE cannot share a pass)
o=gpurun_out/${tag}_pmc1800
rm -rf $o; mkdir -p $o
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $o/a -o p -- python tools/trace_pass.py 1800 2 device 1 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/f -o p -- python tools/trace_pass.py 1800 2 device 1 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/w -o p -- python tools/trace_pass.py 1800 2 device 1 > /dev/null 2>&1
python tools/summarize_pmc_small.py $o > gpurun_out/${tag}_pmc_small_kernels.md
rm -rf $o
fi
if [ "$part" = all ] || [ "$part" = rest ]; then
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -6 > gpurun_out/${tag}_pytest_gpu.txt
[ -f gpurun_out/parity_coverage.json ] && cp gpurun_out/parity_coverage.json gpurun_out/${tag}_parity_coverage.json
[ -f gpurun_out/split_vs_f64.md ] && cp gpurun_out/split_vs_f64.md gpurun_out/${tag}_split_vs_f64.md
python tools/parity_margins.py > gpurun_out/${tag}_parity_margins.txt 2>/dev/null
CSS_BENCH_FORCE_SHARDED=1 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/${tag}_rccl_world1.json 2> gpurun_out/${tag}_rccl_world1.log
for w in 2 8; do
  CSS_BENCH_ONE_DEVICE=1 CSS_BENCH_BACKEND=gloo CSS_BENCH_CHECK=1 timeout 1200 python bench.py --gpus $w --steps 2 --warmup 1 > gpurun_out/${tag}_multiprocess_w$w.log 2>&1
done
fi
# (the raw traces are scratch: only the summaries travel back)
find gpurun_out -maxdepth 1 -type d -name "${tag}_prof" -exec rm -r {} +
find gpurun_out -name '*.db' -path "*${tag}_gpmc_*" -delete
ls -la gpurun_out | grep ${tag} | head -40
