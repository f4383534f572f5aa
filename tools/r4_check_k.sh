#!/bin/bash
# hunt for an intermittent crash seen once in check J: the same four files, several times, full output kept on failure
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4k_*.log
for it in 1 2 3 4; do
  timeout 600 python -X faulthandler -m pytest tests/test_hip_parity.py tests/test_feature_options.py tests/test_hip_golden_r2.py tests/test_hip_lanes.py -m gpu -v --timeout 300 -p no:cacheprovider > gpurun_out/r4k_$it.log 2>&1
  rc=$?
  echo "iteration $it rc=$rc $(tail -1 gpurun_out/r4k_$it.log | cut -c1-120)"
  if [ $rc -eq 0 ]; then rm gpurun_out/r4k_$it.log; else dmesg 2>/dev/null | tail -5; fi
done
