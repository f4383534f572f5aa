#!/bin/bash
# probe: tail / copy streams at the lowest stream priority
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4p.txt; : > $out
bench() {  # tag, env value
  CSS_EXPERIMENT_LOW_PRIORITY=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-long --no-cpu-baseline > gpurun_out/r4p_$1.json 2> gpurun_out/r4p_$1.err
  python - <<PY >> $out
import json
try:
    d=json.loads(open("gpurun_out/r4p_$1.json").read().strip().splitlines()[-1])
    r=d["runs_ms"]["per_step_ms_of_each_timed_region"]
    print("$1", d["value"], d["ms_per_step"], "regions", min(r), max(r), len(r), "sync", d["synchronous_call"]["ms_per_step"])
except Exception as e: print("$1 failed", e)
PY
}
for i in 1 2 3; do
bench base_$i 0
bench tail_$i 1
bench both_$i 3
done
bench lanes_hi 4
bench all 7
cat $out
