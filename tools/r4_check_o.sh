#!/bin/bash
# lanes of a shared batch, with the output leg as DMA (default since check n)
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4o.txt; : > $out
bench() {  # tag, extra args
  timeout 300 python bench.py --steps 20 --warmup 3 --no-long --no-cpu-baseline $2 > gpurun_out/r4o_$1.json 2> gpurun_out/r4o_$1.err
  python - <<PY >> $out
import json
try:
    d=json.loads(open("gpurun_out/r4o_$1.json").read().strip().splitlines()[-1])
    r=d["runs_ms"]["per_step_ms_of_each_timed_region"]
    print("$1", d["value"], d["ms_per_step"], "regions", min(r), max(r), len(r))
except Exception as e: print("$1 failed", e)
PY
}
for i in 1 2 3; do
bench l2_$i ""
bench l3_$i "--tune group_lanes=3"
bench l2map_$i "--tune group_out_dma=0"
done
bench l2_g4 "--queue-group 4 --max-batch 160"
bench l3_g4 "--queue-group 4 --max-batch 160 --tune group_lanes=3"
cat $out
