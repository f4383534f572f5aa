#!/bin/bash
# round-3 check on the 1-GPU box: new tests, the N = 1 bench line, and the N > 1 path as real processes on ONE device
# (gloo carries the exchanges: RCCL refuses two ranks on one GPU).  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_schedules.py tests/test_hip_gemm.py -x -q 2>&1 | tail -15 ) > gpurun_out/r3_tests.txt
( timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err ) ; echo "n1 rc=$?" >> gpurun_out/r3_tests.txt
( CSS_BENCH_ONE_DEVICE=1 CSS_BENCH_BACKEND=gloo CSS_BENCH_CHECK=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r3_bench_w2.json 2> gpurun_out/r3_bench_w2.err ) ; echo "w2 rc=$?" >> gpurun_out/r3_tests.txt
( CSS_BENCH_ONE_DEVICE=1 CSS_BENCH_BACKEND=gloo CSS_BENCH_CHECK=1 timeout 1200 python bench.py --gpus 8 --steps 2 --warmup 1 > gpurun_out/r3_bench_w8.json 2> gpurun_out/r3_bench_w8.err ) ; echo "w8 rc=$?" >> gpurun_out/r3_tests.txt
cat gpurun_out/r3_tests.txt
tail -5 gpurun_out/r3_bench_w8.err
