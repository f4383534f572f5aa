#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun):  bash tools/profile_r05.sh [bench|pmc|rest|all]  ->  gpurun_out/r05_* ;
# copy the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
part=${1:-all}
if [ "$part" = all ] || [ "$part" = bench ]; then
python bench.py --steps 20 --warmup 3 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_prof -o p -- python bench.py --steps 12 --warmup 3 --min-seconds 1 --no-cpu-baseline --no-long > gpurun_out/r05_bench_under_rocprof.json 2>/dev/null
python tools/summarize_prof.py gpurun_out/r05_prof/p_kernel_trace.csv > gpurun_out/r05_kernel_stats.md
cp gpurun_out/r05_prof/p_kernel_stats.csv gpurun_out/r05_rocprofv3_kernel_stats.csv
fi
if [ "$part" = all ] || [ "$part" = pmc ]; then
# counters of the headline's own GEMM launches (M = 22 320), both arithmetic modes; separate passes, --kernel-trace only
for mode in exact_f32 split_f16; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/r05_gpmc_${mode}_$i -o p -- python tools/gemm_traffic.py $mode 3 > gpurun_out/r05_gpmc_${mode}_$i.json 2>/dev/null
  done
done
python tools/summarize_gemm_pmc.py gpurun_out r05 > gpurun_out/r05_gemm_pmc.md
fi
if [ "$part" = all ] || [ "$part" = rest ]; then
timeout 1500 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -6 > gpurun_out/r05_pytest_gpu.txt
[ -f gpurun_out/parity_coverage.json ] && cp gpurun_out/parity_coverage.json gpurun_out/r05_parity_coverage.json
[ -f gpurun_out/split_vs_f64.md ] && cp gpurun_out/split_vs_f64.md gpurun_out/r05_split_vs_f64.md
python tools/parity_margins.py > gpurun_out/r05_parity_margins.txt 2>/dev/null
CSS_BENCH_FORCE_SHARDED=1 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r05_rccl_world1.json 2> gpurun_out/r05_rccl_world1.log
for w in 2 8; do
  CSS_BENCH_ONE_DEVICE=1 CSS_BENCH_BACKEND=gloo CSS_BENCH_CHECK=1 timeout 1200 python bench.py --gpus $w --steps 2 --warmup 1 > gpurun_out/r05_multiprocess_w$w.log 2>&1
done
fi
# (the raw traces are scratch: only the summaries travel back)
find gpurun_out -maxdepth 1 -type d -name 'r05_prof' -exec rm -r {} +
find gpurun_out -name '*.db' -path '*r05_gpmc_*' -delete
ls -la gpurun_out | grep r05 | head -40
