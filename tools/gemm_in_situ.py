#!/usr/bin/env python3
"""Per-shape launch times of the Linear-layer GEMMs INSIDE the mask estimator, from a rocprofv3 kernel trace of a
single-lane run (the launches of a Conformer block come in a fixed order: ffn-up, ffn-down, qkv, attn-out, ffn-up,
ffn-down; the first GEMM of a pass is the embed layer, the last the mask head).
    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python bench.py --lanes 1 --steps 3 --warmup 1 --no-long --no-cpu-baseline
    python tools/gemm_in_situ.py DIR/**/p_kernel_trace.csv"""
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp")
df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
g = df[df["Kernel_Name"].str.contains("gemm_split_wd_kernel|gemm_split_ws_kernel|gemm_split_dma_kernel")].copy()
g["blocks"] = g["Grid_Size_X"] // g["Workgroup_Size_X"]
g["kern"] = g["Kernel_Name"].str.extract(r"(gemm_split_\w+?_kernel)")[0]
# single-lane launches of the 40-segment meeting: M = 7440 -> 468 / 936 / 1404 blocks (64-row tiles) or 236 / 472 / 708 (128 x 128)
names = {468: "N=512", 236: "N=512", 936: "N=1024", 472: "N=1024", 1404: "N=1536", 708: "N=1536"}
g = g[g["blocks"].isin(names)]
order = ["ffn-up", "ffn-down", "qkv", "attn-out", "ffn-up2", "ffn-down2"]
rows, i, expect_embed = [], 0, True
for _, r in g.iterrows():
    n = names[int(r["blocks"])]
    if expect_embed and n == "N=512":        # the embed layer opens a pass (then 18 blocks x 6 launches)
        rows.append(("embed", r["kern"], r["dur_us"]))
        expect_embed, i = False, 0
        continue
    rows.append((order[i % 6], r["kern"], r["dur_us"]))
    i += 1
    if i == 18 * 6:
        expect_embed = True
out = pd.DataFrame(rows, columns=["layer", "kernel", "us"])
print(out.groupby(["layer", "kernel"])["us"].agg(["count", "mean", "min", "median"]).round(2).to_string())
print("sum of means x launches per pass:", round(sum(out.groupby("layer")["us"].mean()[k] * (1 if k == "embed" else 18) for k in out["layer"].unique()), 1), "us")
