#!/bin/bash
# round-4 check A on the 1-GPU box: the whole -m gpu suite (new: realistic-mask fixtures, 1-LSB session triple, left-out
# frames of configs[1] vs the oracle, 8 s segments vs the oracle, scipy's assignment rule on the device) and a bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_coverage.json
( timeout 1500 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_hip_long.py 2>&1 | tail -40 ) > gpurun_out/r4a_tests.txt
( timeout 900 python -m pytest tests/test_hip_long.py -m gpu -q --timeout 400 2>&1 | tail -15 ) >> gpurun_out/r4a_tests.txt
( timeout 600 python bench.py --steps 20 --warmup 5 --no-long > gpurun_out/r4a_bench_n1.json 2> gpurun_out/r4a_bench_n1.err ) ; echo "n1 rc=$?" >> gpurun_out/r4a_tests.txt
cat gpurun_out/r4a_tests.txt
head -c 600 gpurun_out/r4a_bench_n1.json
