#!/usr/bin/env python3
"""Per-kernel PMC summary from three separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and
WRITE_SIZE cannot share a pass; FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950 -- both the raw
and the doubled figure are listed, WRITE_SIZE is uncalibrated).
    python tools/summarize_pmc.py gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 > profiles/r01_pmc.md"""
import sys

import pandas as pd


def load(d):
    c = pd.read_csv(f"{d}/p_counter_collection.csv")
    c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
    c["k"] = c["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 44) + \
        " g=" + (c["Grid_Size"] // c["Workgroup_Size"]).astype(str)
    return c.pivot_table(index=["Dispatch_Id", "k", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()


a, f, w = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
ga = a.groupby("k").agg(n=("dur", "size"), dur_us=("dur", lambda x: x.mean() / 1e3), mfma=("SQ_VALU_MFMA_BUSY_CYCLES", "mean"),
                        gui=("GRBM_GUI_ACTIVE", "mean"), wave=("SQ_WAVE_CYCLES", "mean"), wait=("SQ_WAIT_ANY", "mean"))
gf = f.groupby("k").agg(fetch_kb=("FETCH_SIZE", "mean"), dur_f=("dur", "mean"))
gw = w.groupby("k").agg(write_kb=("WRITE_SIZE", "mean"))
m = ga.join(gf).join(gw)
m["clk_GHz"] = m["gui"] / 8 / (m["dur_us"] * 1e3)          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
m["mfma_util"] = m["mfma"] / (m["gui"] / 8 * 1024)           # 256 CUs x 4 SIMDs
m["wait_frac"] = m["wait"] / m["wave"]
m["fetch_MB"] = m["fetch_kb"] / 1024
m["write_MB"] = m["write_kb"] / 1024
m["hbm_GBps_raw"] = (m["fetch_kb"] + m["write_kb"]) * 1024 / (m["dur_us"] * 1e-6) / 1e9
m["hbm_GBps_fetch_x2"] = (2 * m["fetch_kb"] + m["write_kb"]) * 1024 / (m["dur_us"] * 1e-6) / 1e9
m = m.sort_values("dur_us", ascending=False)
print("# PMC summary per kernel (mean per launch; `python bench.py --steps 3 --warmup 1`, three separate --pmc passes)\n")
print("| kernel (g = workgroups) | launches | avg us | clock GHz | MFMA busy | wave cycles parked (SQ_WAIT_ANY) | FETCH_SIZE MB | WRITE_SIZE MB | HBM GB/s (raw) | HBM GB/s (fetch x2) |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, r in m.iterrows():
    if r["n"] < 3:
        continue
    print(f"| `{k}` | {int(r['n'])} | {r['dur_us']:.1f} | {r['clk_GHz']:.2f} | {100 * r['mfma_util']:.1f}% | {100 * r['wait_frac']:.0f}% | "
          f"{r['fetch_MB']:.1f} | {r['write_MB']:.1f} | {r['hbm_GBps_raw']:.0f} | {r['hbm_GBps_fetch_x2']:.0f} |")

# HBM traffic per launch of the dominant kernel for bench.py's roofline.traffic (FETCH_SIZE doubled as the guide
# prescribes for gfx950's wide reads, + WRITE_SIZE), launch-weighted over the Linear-layer GEMM instantiations
import json, os
# (the full-batch launches of bench.py's single-lane profile pass, which is what roofline.achieved is measured on:
#  117 row tiles x N / 128 column tiles = 468 / 936 / 1404 workgroups; the three-lane launches have a third of that)
g = m[m.index.str.contains("gemm_split_wd_kernel", regex=False) & (m.index.str.extract(r"g=(\d+)", expand=False).astype(float) >= 400)]
if len(g):
    wgt = g["n"]
    out = {"kernel": "css::gemm_split_wd_kernel", "launches": int(wgt.sum()),
           "fetch_bytes_raw": float((g["fetch_kb"] * 1024 * wgt).sum() / wgt.sum()),
           "write_bytes": float((g["write_kb"] * 1024 * wgt).sum() / wgt.sum()),
           "traffic_bytes_per_launch": float(((2 * g["fetch_kb"] + g["write_kb"]) * 1024 * wgt).sum() / wgt.sum()),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x 2 per MI355X_MICROARCH.md"}
    dst = os.environ.get("CSS_TRAFFIC_JSON")
    if dst:
        with open(dst, "w") as f:
            json.dump(out, f, indent=1)
