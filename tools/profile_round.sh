#!/bin/bash
# Collects everything profiles/ holds for one round (run on the GPU box through gpurun):
#   bash tools/profile_round.sh r01
# -> gpurun_out/<tag>_* ; copy the summaries into profiles/ afterwards.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
python tools/summarize_prof.py gpurun_out/${tag}_prof/p_kernel_trace.csv > gpurun_out/${tag}_kernel_stats.md
cp gpurun_out/${tag}_prof/p_kernel_stats.csv gpurun_out/${tag}_rocprofv3_kernel_stats.csv
# PMC: separate passes, --kernel-trace only (FETCH_SIZE and WRITE_SIZE cannot share a pass)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/${tag}_pmc1 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc2 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc3 -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
CSS_TRAFFIC_JSON=gpurun_out/${tag}_gemm_traffic.json python tools/summarize_pmc.py gpurun_out/${tag}_pmc1 gpurun_out/${tag}_pmc2 gpurun_out/${tag}_pmc3 > gpurun_out/${tag}_pmc.md
python tools/parity_margins.py > gpurun_out/${tag}_parity_margins.txt 2>/dev/null
head -c 1500 gpurun_out/${tag}_bench.json; echo; head -30 gpurun_out/${tag}_pmc.md
