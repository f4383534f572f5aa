#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_coverage.json
( timeout 1500 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -12 ) > gpurun_out/r4i.txt
( timeout 300 python examples/session_queue.py 2>&1 | tail -2 ) >> gpurun_out/r4i.txt
( timeout 500 python bench.py --steps 20 --warmup 3 > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err ); echo "bench rc=$?" >> gpurun_out/r4i.txt
cat gpurun_out/r4i.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r4i_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], "dev", d["device_resident"]["ms_per_step"], "sync", d["synchronous_call"]["ms_per_step"], "1800", d["meeting_1800s"]["ms_per_step"])
print("hbm", [(x["kernel"], x["us"], x["frac"]) for x in d["roofline_hbm"]])
print("hbm1800", [(x["kernel"], x["us"], x["frac"]) for x in d["roofline_hbm_1800s"]])
PY
