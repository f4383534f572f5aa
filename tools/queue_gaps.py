#!/usr/bin/env python3
"""Per hardware queue of the LAST pass in a rocprofv3 --kernel-trace directory: span, sum of kernel durations, and the
idle time between consecutive kernels of the queue (start[i + 1] - end[i]).   python tools/queue_gaps.py <dir>"""
import glob, os, sys
import numpy as np
import pandas as pd
kt = pd.read_csv(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0])
kt["name"] = kt["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("css::", "").str.replace("void ", "").str.slice(0, 34)
starts = kt[kt["name"].str.contains("deinterleave")]["Start_Timestamp"].sort_values().values
t0 = ([starts[0]] + [b for a, b in zip(starts, starts[1:]) if b - a > 2_000_000])[-1] - 20_000
k = kt[kt["Start_Timestamp"] >= t0].sort_values("Start_Timestamp")
print(f"last pass: {len(k)} kernels, span {(k['End_Timestamp'].max() - k['Start_Timestamp'].min()) / 1e3:.1f} us")
for q, g in k.groupby("Queue_Id"):
    g = g.sort_values("Start_Timestamp")
    s, e = g["Start_Timestamp"].values, g["End_Timestamp"].values
    gaps = (s[1:] - e[:-1]) / 1e3
    dur = (e - s) / 1e3
    if len(g) < 20:
        continue
    print(f"queue {q}: {len(g)} kernels, span {(e[-1] - s[0]) / 1e3:8.1f} us, sum of durations {dur.sum():8.1f}, gaps: sum {gaps.sum():7.1f} "
          f"median {np.median(gaps):5.2f} p90 {np.percentile(gaps, 90):5.2f} max {gaps.max():6.1f}; negative (overlap) {int((gaps < 0).sum())}")
    after = pd.DataFrame({"prev": g["name"].values[:-1], "gap": gaps}).groupby("prev")["gap"].agg(["count", "median", "sum"])
    print(after.sort_values("sum", ascending=False).head(8).to_string())
