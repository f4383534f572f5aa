#!/bin/bash
# Per-kernel time of two builds on one box: rocprofv3 --kernel-trace --stats of bench.py with the current library and with
# notsofar1-challenge_amd/libcss_base.so (see tools/ab_bench.sh).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=notsofar1-challenge_amd
cp $P/libcss_mi355.so /tmp/libcss_new.so
for v in new base; do
  [ $v = base ] && cp $P/libcss_base.so $P/libcss_mi355.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$v -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "== $v"
  python - "$v" <<'PY'
import pandas as pd, glob, sys
f = glob.glob("gpurun_out/abp_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
c = pd.read_csv(f[0]).head(9)
for _, r in c.iterrows():
    print("%-60s calls %6d  total %9.2f ms  avg %7.2f us" % (r["Name"][:60], r["Calls"], r["TotalDurationNs"] / 1e6, r["AverageNs"] / 1e3))
PY
done
cp /tmp/libcss_new.so $P/libcss_mi355.so
