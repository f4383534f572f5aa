#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest "tests/test_hip_parity.py::test_short_and_odd_segment_lengths_vs_oracle" -m gpu -q --timeout 400 2>&1 | tail -5 ) > gpurun_out/r4d_tests.txt
( timeout 900 python tools/rccl_slowdown_probe.py 1800 > gpurun_out/r4d_rccl_probe.json 2> gpurun_out/r4d_rccl_probe.err ); echo "probe rc=$?" >> gpurun_out/r4d_tests.txt
cat gpurun_out/r4d_tests.txt; cat gpurun_out/r4d_rccl_probe.json; tail -5 gpurun_out/r4d_rccl_probe.err
