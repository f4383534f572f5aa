#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r4h.txt
( timeout 300 python examples/session_queue.py 2>&1 | tail -6 ) >> gpurun_out/r4h.txt
( timeout 300 python examples/sharded_meeting.py 2>&1 | tail -4 ) >> gpurun_out/r4h.txt
( timeout 600 python -m pytest tests/test_hip_schedules.py tests/test_hip_gemm.py -m gpu -q --timeout 300 2>&1 | tail -4 ) >> gpurun_out/r4h.txt
( timeout 400 python bench.py --steps 20 --warmup 3 --no-long > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err ); echo "bench rc=$?" >> gpurun_out/r4h.txt
cat gpurun_out/r4h.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r4h_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("achieved","frac","traffic","algorithmic_bytes_per_launch","traffic_over_algorithmic","traffic_source","rows_per_launch")})
PY
