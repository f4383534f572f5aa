#!/usr/bin/env python3
"""Timeline of the LAST css_run pass in a rocprofv3 --kernel-trace --memory-copy-trace output directory: when each
PCIe piece moved, when each stream's first / last kernel ran.   python tools/timeline.py <dir> [prefix]"""
import glob, os, sys
import pandas as pd
d = sys.argv[1]
kt = pd.read_csv(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0])
mc = pd.read_csv(glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)[0])
kt["name"] = kt["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("css::", "").str.slice(0, 40)
# a pass starts with its first host-to-device copy; take the last pass (H2D copies closer than 2 ms belong together)
h2d = mc[mc["Direction"].str.contains("HOST_TO_DEVICE")]["Start_Timestamp"].sort_values().values if len(mc) else []
if len(h2d):
    groups = [h2d[0]] + [b for a, b in zip(h2d, h2d[1:]) if b - a > 2_000_000]
    t0 = groups[-1] - 20_000
else:
    starts = kt[kt["name"].str.contains("deinterleave")]["Start_Timestamp"].values
    t0 = ([starts[0]] + [b for a, b in zip(starts, starts[1:]) if b - a > 2_000_000])[-1] - 20_000
k = kt[kt["Start_Timestamp"] >= t0].copy(); m = mc[mc["Start_Timestamp"] >= t0].copy()
base = min(k["Start_Timestamp"].min(), m["Start_Timestamp"].min() if len(m) else 1 << 62)
rel = lambda x: (x - base) / 1e3
print(f"# last pass: {len(k)} kernels, {len(m)} copies; times in us from the first event\n\n## copies")
for _, r in m.sort_values("Start_Timestamp").iterrows():
    dur = (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3
    b = r.get("Bytes", r.get("Size", 0))
    print(f"{r['Direction'] if 'Direction' in r else r.get('Name', '')}  {rel(r['Start_Timestamp']):9.1f} -> {rel(r['End_Timestamp']):9.1f}  ({dur:7.1f} us, {b / 1e6:7.2f} MB, {b / max(dur, 1e-3) / 1e3:6.1f} GB/s)")
print("\n## kernels per queue")
for q, g in k.groupby("Queue_Id"):
    g = g.sort_values("Start_Timestamp")
    busy = ((g["End_Timestamp"] - g["Start_Timestamp"]).sum()) / 1e3
    print(f"queue {q}: {len(g)} kernels, first {g.iloc[0]['name']} at {rel(g.iloc[0]['Start_Timestamp']):.1f}, last {g.iloc[-1]['name']} ends {rel(g.iloc[-1]['End_Timestamp']):.1f}, sum of durations {busy:.1f} us")
    head = g.head(5); tail = g.tail(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    for _, r in pd.concat([head, tail]).iterrows():
        print(f"    {rel(r['Start_Timestamp']):9.1f} -> {rel(r['End_Timestamp']):9.1f}  {r['name']}")
