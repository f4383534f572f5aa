#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_schedules.py tests/test_hip_realistic.py tests/test_hip_lanes.py -m gpu -q --timeout 400 2>&1 | tail -8 ) > gpurun_out/r4f_tests.txt
run() { tag=$1; shift; ( timeout 300 python bench.py --steps 21 --warmup 6 --no-long --no-cpu-baseline --min-seconds 2 "$@" > gpurun_out/r4f_bench_$tag.json 2> gpurun_out/r4f_bench_$tag.err ); echo "$tag rc=$?" >> gpurun_out/r4f_tests.txt; }
run g8 ; run g1 --queue-group 1 ; run g8again
cat gpurun_out/r4f_tests.txt
for t in g8 g1 g8again; do python - <<PY
import json
d=json.loads(open("gpurun_out/r4f_bench_$t.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$t", d["value"], d["ms_per_step"], "roof", r["achieved"], r["frac"], r["avg_launch_us"], d["runs_ms"]["min"], d["runs_ms"]["max"], "sync", d["synchronous_call"]["ms_per_step"], "dev", d["device_resident"]["ms_per_step"])
print("   hbm", [(x["kernel"], x["us"], x["frac"]) for x in d["roofline_hbm"]])
PY
done
