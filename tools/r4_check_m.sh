#!/bin/bash
# A/B of the conv module's packed-FMA phase 2 (libraries prebuilt under tools/bin) + the queue's output leg as DMA
mkdir -p gpurun_out
export TMPDIR=/tmp
L=notsofar1-challenge_amd/libcss_mi355.so
out=gpurun_out/r4m.txt; : > $out
for v in hv2 hv1; do
  cp tools/bin/libcss_$v.so $L
  ( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_lanes.py tests/test_hip_schedules.py -m gpu -q -x --timeout 600 2>&1 | tail -4 ) >> $out
done
bench() {  # name, extra args
  cp tools/bin/libcss_$1.so $L
  timeout 300 python bench.py --steps 20 --warmup 3 --no-long --no-cpu-baseline $2 > gpurun_out/r4m_$1$3.json 2> gpurun_out/r4m_$1$3.err
  python - <<PY >> $out
import json
try:
    d=json.loads(open("gpurun_out/r4m_$1$3.json").read().strip().splitlines()[-1])
    f=d["kernel_family_ms"]; g=d["kernel_family_ms_per_session_in_a_shared_batch"]
    print("$1$3", d["value"], d["ms_per_step"], "dev", d["device_resident"]["ms_per_step"], "sync", d["synchronous_call"]["ms_per_step"], "conv", f["conv_module"], g["conv_module"], "wave_ola", g["wave_ola"])
except Exception as e: print("$1$3 failed", e)
PY
}
bench base "" _1
bench hv2 "" _1
bench hv1 "" _1
bench base "" _2
bench hv2 "" _2
bench hv1 "" _2
bench hv2 "--tune group_out_dma=1" _dma1
bench hv2 "" _3
bench hv2 "--tune group_out_dma=1" _dma2
cp tools/bin/libcss_hv2.so $L
cat $out
