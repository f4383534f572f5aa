#!/bin/bash
# A/B: the queue's output leg written over PCIe by the overlap-add kernel (default) or copied out by DMA
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r4n.txt; : > $out
bench() {  # tag, extra args
  timeout 300 python bench.py --steps 20 --warmup 3 --no-long --no-cpu-baseline $2 > gpurun_out/r4n_$1.json 2> gpurun_out/r4n_$1.err
  python - <<PY >> $out
import json
try:
    d=json.loads(open("gpurun_out/r4n_$1.json").read().strip().splitlines()[-1])
    g=d["kernel_family_ms_per_session_in_a_shared_batch"]
    print("$1", d["value"], d["ms_per_step"], "regions", min(d["runs_ms"]["per_step_ms_of_each_timed_region"]), max(d["runs_ms"]["per_step_ms_of_each_timed_region"]), "wave_ola", g["wave_ola"])
except Exception as e: print("$1 failed", e)
PY
}
bench map_1 ""
bench dma_1 "--tune group_out_dma=1"
bench map_2 ""
bench dma_2 "--tune group_out_dma=1"
bench map_3 ""
bench dma_3 "--tune group_out_dma=1"
bench dma_l3 "--tune group_out_dma=1 --tune group_lanes=3"
bench dma_l1 "--tune group_out_dma=1 --tune group_lanes=1"
( timeout 600 python -m pytest tests/test_hip_schedules.py -m gpu -q -x --timeout 600 2>&1 | tail -3 ) >> $out
cat $out
