#!/bin/bash
# A/B of two builds on ONE box (boxes of the pool differ by +-5 %): bench.py with the current library and with
# notsofar1-challenge_amd/libcss_base.so (a build of the commit to compare against, made by hand), interleaved.
#   bash tools/ab_bench.sh [rounds]
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
P=notsofar1-challenge_amd
[ -f $P/libcss_base.so ] || { echo "no $P/libcss_base.so"; exit 1; }
cp $P/libcss_mi355.so /tmp/libcss_new.so
one() { python bench.py --steps 30 --warmup 5 --no-long --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_family_ms']
print('$1', 'host->host', d['ms_per_step'], 'device-resident', d['device_resident']['ms_per_step'], 'one lane: gemm', k['linear_gemm'], 'attention', k['attention'], 'conv', k['conv_module'], 'stft', k['stft'], 'scm', k['scm'], 'features', k['features'])"; }
for i in $(seq 1 ${1:-3}); do
  cp /tmp/libcss_new.so $P/libcss_mi355.so; one new
  cp $P/libcss_base.so $P/libcss_mi355.so; one base
done
cp /tmp/libcss_new.so $P/libcss_mi355.so
