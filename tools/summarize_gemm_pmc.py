#!/usr/bin/env python3
"""Counter passes of tools/profile_r05.sh -> a table per arithmetic mode (mean per launch of the mode's Linear-layer kernel
in the headline's shared batch: M = 44 640 rows in the exact float32 mode, 22 320 in the split-f16 mode) + <tag>_gemm_traffic[_exact_f32].json for bench.py's roofline.traffic (FETCH_SIZE doubled
as MI355X_MICROARCH.md prescribes for gfx950's wide reads, + WRITE_SIZE; KiB units).
    python tools/summarize_gemm_pmc.py gpurun_out r05 > gpurun_out/r05_gemm_pmc.md"""
import glob
import json
import os
import sys

import pandas as pd

o, tag = sys.argv[1], sys.argv[2]
KERNEL = {"exact_f32": "gemm_f32_kernel", "split_f16": "gemm_split_wd_kernel"}
print("# Counters of the Linear-layer launches of the headline's shared batch (queued 60 s sessions: six = M 44 640 rows per launch in the exact float32 mode, three = 22 320 in the split-f16 mode)\n")
print("`rocprofv3 --kernel-trace --pmc <set> -- python tools/gemm_traffic.py <mode> 3`, one counter set per run; mean per launch of the")
print("mode's kernel (exact_f32: css::gemm_f32_kernel incl. the mask head; split_f16: css::gemm_split_wd_kernel incl. the transposed head).\n")
for mode, kern in KERNEL.items():
    vals, dur, n = {}, 0.0, 0
    for d in sorted(glob.glob(f"{o}/{tag}_gpmc_{mode}_*")):
        if not os.path.isdir(d):
            continue
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        c = pd.read_csv(f[0])
        c = c[c["Kernel_Name"].str.contains(kern)].copy()
        if not len(c):
            continue
        c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
        t = c.pivot_table(index=["Dispatch_Id", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
        for col in t.columns:
            if col not in ("Dispatch_Id", "dur"):
                vals[col] = float(t[col].mean())
        dur, n = float(t["dur"].mean()) / 1e3, len(t)
    meta = {}
    for j in sorted(glob.glob(f"{o}/{tag}_gpmc_{mode}_*.json")):
        try:
            meta = json.loads(open(j).read().strip().splitlines()[-1])
        except Exception:
            pass
    print(f"## {mode}: `css::{kern}` ({n} launches in the last pass, {dur:.1f} us mean under the counters)\n")
    print("| counter | mean per launch |")
    print("|---|---:|")
    for k in sorted(vals):
        print(f"| {k} | {vals[k]:.4g} |")
    g = vals.get("GRBM_GUI_ACTIVE")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in vals and dur:
        clk = g / 8 / (dur * 1e3)
        print(f"\nclock {clk:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration); MFMA pipes busy {100 * vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / 8 * 1024):.1f} % "
              f"(SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs))")
    if "SQ_WAVE_CYCLES" in vals:
        w = vals["SQ_WAVE_CYCLES"]
        print(f"wave cycles: {100 * vals.get('SQ_WAIT_INST_ANY', 0) / w:.0f} % waiting to issue (SQ_WAIT_INST_ANY), {100 * vals.get('SQ_WAIT_ANY', 0) / w:.0f} % parked at "
              f"s_waitcnt / barriers (SQ_WAIT_ANY), {100 * vals.get('SQ_ACTIVE_INST_ANY', 0) / w:.0f} % issuing (SQ_ACTIVE_INST_ANY)")
    if "SQ_INSTS_MFMA" in vals:
        m = vals["SQ_INSTS_MFMA"]
        print(f"per MFMA: {vals.get('SQ_INSTS_VMEM_RD', 0) / m:.3f} vector-memory reads, {vals.get('SQ_INSTS_LDS', 0) / m:.3f} LDS instructions, "
              f"{vals.get('SQ_INSTS_VALU', 0) / m - 1:.3f} other VALU; SQ_INST_CYCLES_VMEM per vector-memory read = "
              f"{vals.get('SQ_INST_CYCLES_VMEM', 0) / max(vals.get('SQ_INSTS_VMEM_RD', 1), 1):.1f}")
    if "TCC_HIT_sum" in vals:
        print(f"L2: {100 * vals['TCC_HIT_sum'] / max(vals['TCC_HIT_sum'] + vals.get('TCC_MISS_sum', 0), 1):.1f} % hits (TCC_HIT_sum / (HIT + MISS), all eight XCDs)")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        traffic = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        alg = meta.get("algorithmic_bytes_per_launch")
        print(f"HBM traffic per launch: FETCH_SIZE x 2 + WRITE_SIZE = {traffic / 1e6:.1f} MB" + (f" against {alg / 1e6:.1f} MB algorithmic = {traffic / alg:.2f} x" if alg else ""))
        out = {f"M{meta.get('rows_per_launch', 22320)}": {"traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"],
                          "launches": n, "mean_us_under_counters": dur}}
        suffix = "" if mode == "split_f16" else "_exact_f32"
        with open(f"{o}/{tag}_gemm_traffic{suffix}.json", "w") as fjs:
            json.dump(out, fjs, indent=1)
    print()
