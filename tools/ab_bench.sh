#!/bin/bash
# A/B of two builds on ONE box (boxes of the pool differ by +-5 %): bench.py with the current library and with
# notsofar1-challenge_amd/libcss_base.so (a build of the commit to compare against, made by hand), interleaved.
#   bash tools/ab_bench.sh [rounds]
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
P=notsofar1-challenge_amd
[ -f $P/libcss_base.so ] || { echo "no $P/libcss_base.so"; exit 1; }
cp $P/libcss_mi355.so /tmp/libcss_new.so
one() { python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stage_ms']['masknet'])"; }
for i in $(seq 1 ${1:-3}); do
  cp /tmp/libcss_new.so $P/libcss_mi355.so; one new
  cp $P/libcss_base.so $P/libcss_mi355.so; one base
done
cp /tmp/libcss_new.so $P/libcss_mi355.so
