#!/usr/bin/env python3
"""How much of the LAST pass in a rocprofv3 --kernel-trace directory ran concurrently: wall time, time with 0 / 1 / 2 /
... kernels in flight, sum of kernel durations per family, and the idle gaps.   python tools/concurrency.py <dir>"""
import glob, os, sys
import numpy as np
import pandas as pd
d = sys.argv[1]
kt = pd.read_csv(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0])
kt["name"] = kt["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("css::", "").str.replace(r"<.*", "", regex=True)
kt = kt.sort_values("Start_Timestamp")
starts = kt[kt["name"].str.contains("pcm_peak|deinterleave")]["Start_Timestamp"].values
t0 = ([starts[0]] + [b for a, b in zip(starts, starts[1:]) if b - a > 2_000_000])[-1] - 1000
k = kt[kt["Start_Timestamp"] >= t0]
s, e = k["Start_Timestamp"].values, k["End_Timestamp"].values
wall = (e.max() - s.min()) / 1e3
ev = sorted([(t, 1) for t in s] + [(t, -1) for t in e])
hist, cur, last = {}, 0, ev[0][0]
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += dlt
print(f"last pass: {len(k)} kernels on {k['Queue_Id'].nunique()} queues, wall {wall:.1f} us, sum of durations {(e - s).sum() / 1e3:.1f} us")
for c in sorted(hist): print(f"  {c} kernels in flight: {hist[c] / 1e3:8.1f} us ({100 * hist[c] / 1e3 / wall:4.1f} %)")
g = k.assign(dur=(e - s) / 1e3).groupby("name")["dur"].agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
print(g.head(14).to_string(float_format=lambda x: f"{x:.1f}"))
