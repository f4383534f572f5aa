#!/bin/bash
# tools/build_attn_bench.sh <name> [extra -D flags ...]  ->  tools/bin/<name> (tools/attn_bench.hip + encoder.hip), quiet unless it fails
n=$1; shift
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value "$@" /root/repo/tools/attn_bench.hip /root/repo/notsofar1-challenge_amd/csrc/encoder.hip -I/root/repo/notsofar1-challenge_amd/csrc -o /root/repo/tools/bin/$n > /tmp/build_$n.log 2>&1 || { echo "BUILD FAILED $n"; grep -m3 -A4 error /tmp/build_$n.log; }
