#!/usr/bin/env python3
"""Steady state of the session queue in a rocprofv3 --kernel-trace directory: over a window in the middle of the run, per
hardware queue the busy time (sum of kernel durations), the idle time between consecutive kernels and which kernel the
gaps follow; overall the time with at least one kernel running and the mean number running.
    python tools/steady_state_gaps.py <dir> [window_ms]
CAUTION: under rocprofv3 the host needs 6.2 ms to enqueue a session the device finishes in 5.0 ms (r04_bench_under_rocprof.json:
host_enqueue), so the gaps this prints after each block's closing LayerNorm (~ 100 us) and the 12 % of the window with no kernel
running are the PROFILER's host overhead, not the product's: untraced, the host enqueues a session in 1.8 ms."""
import glob, os, sys
import numpy as np
import pandas as pd
kt = pd.read_csv(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 30e6
kt["name"] = kt["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("css::", "").str.replace("void ", "").str.slice(0, 30)
# the queue phase of bench.py: the densest stretch of gemm launches; take the window ending 20 % before the last kernel
g = kt[kt["name"].str.contains("gemm_split_wd")]
t_end = g["End_Timestamp"].quantile(0.6)
t0, t1 = t_end - win, t_end
k = kt[(kt["Start_Timestamp"] >= t0) & (kt["End_Timestamp"] <= t1)].sort_values("Start_Timestamp")
print(f"window {win / 1e6:.0f} ms: {len(k)} kernels, sum of durations {((k['End_Timestamp'] - k['Start_Timestamp']).sum()) / 1e6:.2f} ms")
ev = np.concatenate([np.stack([k["Start_Timestamp"].values, np.ones(len(k))], 1), np.stack([k["End_Timestamp"].values, -np.ones(len(k))], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
run = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0], append=ev[-1, 0])
busy = dt[run > 0].sum(); print(f"at least one kernel running {busy / win * 100:.1f} % of the window; mean kernels running {(run * dt).sum() / win:.2f}")
for n in (1, 2, 3):
    print(f"  exactly {n} running: {dt[run == n].sum() / win * 100:.1f} %", end="")
print(f"  none: {dt[run <= 0].sum() / win * 100:.1f} %")
for q, gq in k.groupby("Queue_Id"):
    s, e = gq["Start_Timestamp"].values, gq["End_Timestamp"].values
    if len(gq) < 50: continue
    gaps = (s[1:] - e[:-1]) / 1e3
    print(f"queue {q}: {len(gq)} kernels, busy {(e - s).sum() / win * 100:.1f} %, gaps: sum {gaps[gaps > 0].sum() / 1e3:.2f} ms, median {np.median(gaps):.2f} us, p90 {np.percentile(gaps, 90):.2f}")
    after = pd.DataFrame({"prev": gq["name"].values[:-1], "gap": gaps}).groupby("prev")["gap"].agg(["count", "median", "sum"])
    print(after.sort_values("sum", ascending=False).head(6).to_string())
