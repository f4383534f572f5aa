#!/bin/bash
# A/B: sessions per shared estimator batch (max_batch_segments 128 = 3 x 40 segments, 160 = 4, 200 = 5), interleaved on one box
mkdir -p gpurun_out
out=gpurun_out/r4q.txt; : > $out
for i in 1 2 3; do
for mb in 128 160 200; do
  timeout 300 python bench.py --steps 20 --warmup 3 --min-seconds 2 --no-long --no-cpu-baseline --max-batch $mb > gpurun_out/r4q_$mb.json 2> gpurun_out/r4q_$mb.err
  python - <<PY >> $out
import json
try:
    d=json.loads(open("gpurun_out/r4q_$mb.json").read().strip().splitlines()[-1])
    print("max_batch $mb run $i", d["value"], d["ms_per_step"], d["config"].get("sessions_per_estimator_batch"), d["roofline"]["frac"])
except Exception as e: print("$mb failed", e)
PY
done; done
cat $out
