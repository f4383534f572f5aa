#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_lanes.py tests/test_hip_linear_modes.py tests/test_hip_schedules.py -m gpu -q --timeout 600 2>&1 | tail -6 ) > gpurun_out/r4l.txt
( timeout 500 python bench.py --steps 20 --warmup 3 --no-long --no-cpu-baseline > gpurun_out/r4l_bench.json 2> gpurun_out/r4l_bench.err ); echo "bench rc=$?" >> gpurun_out/r4l.txt
cat gpurun_out/r4l.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r4l_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], "dev", d["device_resident"]["ms_per_step"], "sync", d["synchronous_call"]["ms_per_step"])
print("fam", d["kernel_family_ms"]); print("famg", d["kernel_family_ms_per_session_in_a_shared_batch"])
PY
