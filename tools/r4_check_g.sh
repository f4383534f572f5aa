#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_schedules.py tests/test_hip_realistic.py -m gpu -q --timeout 400 2>&1 | tail -8 ) > gpurun_out/r4g_tests.txt
run() { tag=$1; shift; ( timeout 300 python bench.py --steps 21 --warmup 6 --no-long --no-cpu-baseline --min-seconds 2 "$@" > gpurun_out/r4g_bench_$tag.json 2> gpurun_out/r4g_bench_$tag.err ); echo "$tag rc=$?" >> gpurun_out/r4g_tests.txt; }
run default
run lanes3 --tune group_lanes=3
run xfmain --tune group_transform_on_main=1
run mvdrlanes --tune group_mvdr_on_lanes=1
run both --tune group_transform_on_main=1 --tune group_mvdr_on_lanes=1
run first --tune group_transform_on_main=1 --tune group_mvdr_on_lanes=1 --tune group_lanes=3
run lanes1 --tune group_lanes=1
run default2
cat gpurun_out/r4g_tests.txt
for t in default lanes3 xfmain mvdrlanes both first lanes1 default2; do python - <<PY
import json
d=json.loads(open("gpurun_out/r4g_bench_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], d["runs_ms"]["min"], d["runs_ms"]["max"], "dev", d["device_resident"]["ms_per_step"])
PY
done
