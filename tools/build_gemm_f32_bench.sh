#!/bin/bash
# tools/build_gemm_f32_bench.sh <name> [extra -D flags ...]  ->  tools/bin/<name> (tools/gemm_f32_bench.hip + the GEMM units), quiet unless it fails
n=$1; shift
C=/root/repo/notsofar1-challenge_amd/csrc
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DF32_TOOLS=1 "$@" /root/repo/tools/gemm_f32_bench.hip $C/gemm.hip $C/gemm_f32.hip $C/gemm_split.hip $C/gemm_split_wd.hip -I$C -o /root/repo/tools/bin/$n > /tmp/build_$n.log 2>&1 || { echo "BUILD FAILED $n"; grep -m3 -A5 error /tmp/build_$n.log; }
