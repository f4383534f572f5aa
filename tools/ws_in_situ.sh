#!/bin/bash
# A/B of the specialised-wave GEMM inside the mask estimator (one lane), per layer shape, from two kernel traces
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 2; do
  rm -rf gpurun_out/ws_tr$v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ws_tr$v -o p -- python bench.py --lanes 1 --steps 3 --warmup 1 --no-long --no-cpu-baseline --tune gemm_ws=$v > /dev/null 2>&1
  echo "== gemm_ws=$v"
  python tools/gemm_in_situ.py $(find gpurun_out/ws_tr$v -name p_kernel_trace.csv | head -1)
  rm -rf gpurun_out/ws_tr$v
done
