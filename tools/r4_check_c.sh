#!/bin/bash
# round-4 check C: tests touched since check B, the precision table, and the schedule A/B of a shared estimator batch
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_coverage.json
( timeout 1200 python -m pytest tests/test_hip_realistic.py tests/test_hip_schedules.py tests/test_hip_precision.py "tests/test_hip_parity.py::test_short_and_odd_segment_lengths_vs_oracle" tests/test_hip_long.py tests/test_hip_protocol.py tests/test_hip_golden_r2.py -m gpu -q --timeout 400 2>&1 | tail -40 ) > gpurun_out/r4c_tests.txt
run() { tag=$1; shift; ( timeout 300 python bench.py --steps 21 --warmup 6 --no-long --no-cpu-baseline --min-seconds 2 "$@" > gpurun_out/r4c_bench_$tag.json 2> gpurun_out/r4c_bench_$tag.err ); echo "$tag rc=$?" >> gpurun_out/r4c_tests.txt; }
run l3b128 ; run l1b128 --lanes 1 ; run l2b128 --lanes 2 ; run l3b256 --max-batch 256 ; run l1b256 --lanes 1 --max-batch 256 ; run l3b128again
cat gpurun_out/r4c_tests.txt
for t in l3b128 l1b128 l2b128 l3b256 l1b256 l3b128again; do python - <<PY
import json
d=json.loads(open("gpurun_out/r4c_bench_$t.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$t", d["value"], d["ms_per_step"], "roof", r["achieved"], r["frac"], r["avg_launch_us"], r.get("rows_per_launch"), d["runs_ms"]["min"], d["runs_ms"]["max"])
PY
done
