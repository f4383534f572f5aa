#!/usr/bin/env python3
"""The counter passes of tools/profile_round.sh (exact_f32, the headline's shared batch: six queued 60 s sessions, 240 segments,
M = 44 640 token rows) hold every kernel of those passes, not only the GEMM: this prints the mask estimator's OTHER kernels --
attention, LayerNorm, the conv module, the feature kernel -- mean per launch (tools only).

    python tools/summarize_estimator_pmc.py gpurun_out r06 > gpurun_out/r06_estimator_pmc.md

HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (KiB units; gfx950's wide reads counted as /opt/skills/guides/MI355X_MICROARCH.md prescribes)."""
import glob
import sys

import pandas as pd

o, tag = sys.argv[1], sys.argv[2]
ROWS, D = 44640, 512
# algorithmic bytes per launch (what the reference's stage exchanges): rows x width x 4 B in + out
KERNELS = [
    ("relpos_attn_kernel<6, false, false", "relative-position attention, float32 (240 segments x 8 heads x 6 query tiles = 11 520 waves of 608 MFMAs)",
     ROWS * 3 * D * 4 + ROWS * D * 4, 11520 * 608 * 4096.0),
    ("layernorm_kernel<2, 0>", "LayerNorm (one input, one output)", 2 * ROWS * D * 4, 0.0),
    ("layernorm2_kernel<2>", "LayerNorm of a block's end + the next module's (one input, two outputs)", 3 * ROWS * D * 4, 0.0),
    ("conv_module_kernel", "the conv module in one kernel (LayerNorm, GLU, 33-tap depthwise conv, BatchNorm, ReLU, residual, next LayerNorm)", 3 * ROWS * D * 4, 0.0),
    ("features_kernel", "features of one session's 40 segments (per session, not per batch)", None, 0.0),
]
print("# Counters of the mask estimator's kernels OTHER than the GEMM in the headline's shared batch (exact float32 mode; six queued 60 s sessions = 240 segments = 44 640 token rows per launch)\n")
print("From the same `rocprofv3 --kernel-trace --pmc <set> -- python tools/gemm_traffic.py exact_f32 3` passes as `" + tag + "_gemm_pmc.md` (one counter set per run); mean per launch.\n")
passes = []
for d in sorted(glob.glob(f"{o}/{tag}_gpmc_exact_f32_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if f:
        passes.append(pd.read_csv(f[0]))
for pat, what, alg, flops in KERNELS:
    vals, dur, n = {}, 0.0, 0
    for c in passes:
        c = c[c["Kernel_Name"].str.contains(pat, regex=False)].copy()
        if not len(c):
            continue
        c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
        t = c.pivot_table(index=["Dispatch_Id", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
        for col in t.columns:
            if col not in ("Dispatch_Id", "dur"):
                vals[col] = float(t[col].mean())
        dur, n = float(t["dur"].mean()) / 1e3, len(t)
    if not n:
        continue
    print(f"## `{pat}`: {what}\n")
    print(f"{n} launches in the last pass, {dur:.1f} us mean under the counters")
    g = vals.get("GRBM_GUI_ACTIVE")
    if g and dur and flops:   # (a 30 us kernel's GRBM_GUI_ACTIVE also counts its neighbours on the other streams: only the long MFMA kernel's is a clock)
        clk = g / 8 / (dur * 1e3)
        line = f"clock {clk:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)"
        if vals.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            line += f"; MFMA pipes busy {100 * vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / 8 * 1024):.1f} % of the kernel's cycles"
        if flops:
            line += f"; {flops / (dur * 1e-6) / 1e12:.1f} TFLOP/s = {flops / (dur * 1e-6) / 157.3e12:.3f} of 157.3 (= {flops / (dur * 1e-6) / (157.3e12 * clk / 2.4):.3f} of the peak at THIS clock)"
        print(line)
    if "SQ_WAVE_CYCLES" in vals:
        w = vals["SQ_WAVE_CYCLES"]
        print(f"wave cycles: {100 * vals.get('SQ_WAIT_INST_ANY', 0) / w:.0f} % waiting to issue, {100 * vals.get('SQ_WAIT_ANY', 0) / w:.0f} % parked at s_waitcnt / barriers, "
              f"{100 * vals.get('SQ_ACTIVE_INST_ANY', 0) / w:.0f} % issuing")
    if vals.get("SQ_INSTS_MFMA"):
        m = vals["SQ_INSTS_MFMA"]
        print(f"per MFMA: {vals.get('SQ_INSTS_VMEM_RD', 0) / m:.3f} vector-memory reads, {vals.get('SQ_INSTS_LDS', 0) / m:.3f} LDS instructions, {vals.get('SQ_INSTS_VALU', 0) / m - 1:.3f} other VALU")
    elif "SQ_INSTS_VALU" in vals:
        print(f"instructions per launch: {vals['SQ_INSTS_VALU']:.3g} VALU, {vals.get('SQ_INSTS_LDS', 0):.3g} LDS, {vals.get('SQ_INSTS_VMEM_RD', 0):.3g} vector-memory reads")
    if "TCC_HIT_sum" in vals:
        print(f"L2: {100 * vals['TCC_HIT_sum'] / max(vals['TCC_HIT_sum'] + vals.get('TCC_MISS_sum', 0), 1):.1f} % hits")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        traffic = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        line = f"HBM traffic per launch: {traffic / 1e6:.1f} MB = {traffic / (dur * 1e-6) / 1e12:.2f} TB/s"
        if alg:
            line += f"; algorithmic {alg / 1e6:.1f} MB ({traffic / alg:.2f} x) = {alg / (dur * 1e-6) / 1e12:.2f} TB/s = {alg / (dur * 1e-6) / 8e12:.2f} of 8 TB/s"
        print(line)
    print()
