#!/bin/bash
# round-4 check B: the tests that failed in check A (report writer, 499-frame clips), the grouped queue, A/B of the group limit
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_coverage.json
( timeout 900 python -m pytest tests/test_hip_realistic.py tests/test_hip_schedules.py "tests/test_hip_parity.py::test_short_and_odd_segment_lengths_vs_oracle" tests/test_hip_parity.py::test_eight_second_segments_dense_hop_vs_oracle tests/test_hip_session.py -m gpu -q --timeout 300 2>&1 | tail -60 ) > gpurun_out/r4b_tests.txt
for g in 1 3 8; do
  ( timeout 300 python bench.py --steps 21 --warmup 6 --no-long --no-cpu-baseline --queue-group $g > gpurun_out/r4b_bench_g$g.json 2> gpurun_out/r4b_bench_g$g.err ) ; echo "group $g rc=$?" >> gpurun_out/r4b_tests.txt
done
cat gpurun_out/r4b_tests.txt
for g in 1 3 8; do python - <<PY
import json
d=json.loads(open("gpurun_out/r4b_bench_g$g.json").read().strip().splitlines()[-1])
print("group $g", d["value"], d["ms_per_step"], d.get("roofline"), {k: d.get(k) for k in ("device_resident","synchronous_call")})
PY
done
