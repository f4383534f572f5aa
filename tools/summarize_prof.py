#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV into the per-kernel table committed under profiles/.
    python tools/summarize_prof.py gpurun_out/prof/r01_kernel_trace.csv > profiles/r01_kernel_stats.md"""
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1])
df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
df["Kernel_Name"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.slice(0, 60)
g = df.groupby("Kernel_Name")["dur_us"].agg(["count", "sum", "mean", "min", "max"]).sort_values("sum", ascending=False)
tot = g["sum"].sum()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print(f"# rocprofv3 --kernel-trace summary ({sys.argv[1]}; {steps} bench steps incl. warm-up/profile passes)\n")
print("| kernel | launches | total ms | avg us | min us | max us | share |")
print("|---|---:|---:|---:|---:|---:|---:|")
for name, r in g.iterrows():
    print(f"| `{name}` | {int(r['count'])} | {r['sum'] / 1e3:.2f} | {r['mean']:.2f} | {r['min']:.2f} | {r['max']:.2f} | {100 * r['sum'] / tot:.1f}% |")
gm = df[df["Kernel_Name"].str.contains("gemm_kernel|gemm_split|gemm_split_ws")].copy()
if len(gm):
    gm["blocks"] = gm["Grid_Size_X"] // gm["Workgroup_Size_X"]
    print("\n## GEMM kernels (css::gemm_kernel, gemm_split_kernel, gemm_split_wd_kernel) by launch shape\n")
    print("| workgroups | threads/wg | launches | avg us |")
    print("|---:|---:|---:|---:|")
    for (kn, b, w), r in gm.groupby(["Kernel_Name", "blocks", "Workgroup_Size_X"])["dur_us"].agg(["count", "mean"]).iterrows():
        print(f"| {kn[:34]} {b} | {w} | {int(r['count'])} | {r['mean']:.2f} |")
    print(f"\nGEMM launches: {len(gm)}, average duration {gm['dur_us'].mean():.2f} us")
