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

# HBM traffic per launch of the dominant kernel for bench.py's roofline.traffic (FETCH_SIZE doubled as the guide prescribes
# for gfx950's wide reads, + WRITE_SIZE), PER LAUNCH SHAPE, beside the algorithmic bytes of that shape (VERDICT r3 #7/#10:
# `traffic` and `achieved` must describe the same launches).  A launch of the weights-direct kernel has
# ceil(M / 64) x (N / 128) workgroups; the rows M a bench run can produce are known (a 60 s meeting has 40 segments of 186
# frames: 2 480 rows per lane of three, 7 440 alone, 11 160 per lane of a two-lane batch shared by three queued sessions,
# 22 320 for that batch on one lane -- the per-launch profile), so (M, N) follows from the workgroup count; the N = 512
# class mixes K = 512 (attention output, x18), 1024 (feed-forward down, x36) and 1824 (embedding, x1) per pass.
import json, os, re


def algorithmic_bytes(M, N):
    f = 4.0
    if N == 1024:   # feed-forward up: A [M, 512], W [1024, 512], C [M, 1024] (split rows: same bytes)
        return M * 512 * f + N * 512 * f + M * N * f
    if N == 1536:   # QKV
        return M * 512 * f + N * 512 * f + M * N * f
    # N = 512: attention output (+ residual), feed-forward down (+ residual), embedding
    wo = M * 512 * f + N * 512 * f + 2 * M * N * f
    dn = M * 1024 * f + N * 1024 * f + 2 * M * N * f
    em = M * 1824 * f + N * 1824 * f + M * N * f
    return (18 * wo + 36 * dn + em) / 55.0


rows_known = {2480: "one lane of three, 60 s meeting alone", 7440: "60 s meeting alone, one lane (per-launch profile of a single session)",
              11160: "one lane of two, three queued sessions sharing the batch (the headline's launches)",
              22320: "three queued sessions sharing the batch, one lane (the headline's per-launch profile)"}
g = m[m.index.str.contains("gemm_split_wd_kernel", regex=False)].copy()
shapes = []
for k, r in g.iterrows():
    wg = int(re.search(r"g=(\d+)", k).group(1))
    hit = [(M, N) for M in rows_known for N in (512, 1024, 1536) if -(-M // 64) * (N // 128) == wg]
    if len(hit) != 1:
        continue   # (ambiguous or another workload's shape)
    M, N = hit[0]
    traffic = (2 * r["fetch_kb"] + r["write_kb"]) * 1024
    alg = algorithmic_bytes(M, N)
    shapes.append({"rows_M": M, "cols_N": N, "workgroups": wg, "launches": int(r["n"]), "avg_us": round(float(r["dur_us"]), 2),
                   "fetch_bytes_raw": float(r["fetch_kb"] * 1024), "write_bytes": float(r["write_kb"] * 1024),
                   "traffic_bytes": float(traffic), "algorithmic_bytes": float(alg), "traffic_over_algorithmic": round(float(traffic / alg), 3),
                   "what": rows_known[M]})
if shapes:
    print("\n## Linear-layer GEMM: HBM traffic per launch shape against the shape's algorithmic bytes\n")
    print("| rows M | cols N | workgroups | launches | avg us | traffic MB (FETCH x2 + WRITE) | algorithmic MB | ratio |")
    print("|---:|---:|---:|---:|---:|---:|---:|---:|")
    for q in sorted(shapes, key=lambda q: (q["rows_M"], q["cols_N"])):
        print(f"| {q['rows_M']} | {q['cols_N']} | {q['workgroups']} | {q['launches']} | {q['avg_us']} | {q['traffic_bytes'] / 1e6:.1f} | "
              f"{q['algorithmic_bytes'] / 1e6:.1f} | {q['traffic_over_algorithmic']} |")
    # per-launch average over one estimator pass (55 / 36 / 18 launches of the three classes), for the profile's launches
    out = {"kernel": "css::gemm_split_wd_kernel", "shapes": shapes,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x 2 per MI355X_MICROARCH.md"}
    for M in (22320, 7440):
        cls = {q["cols_N"]: q for q in shapes if q["rows_M"] == M}
        if len(cls) == 3:
            wgt = {512: 55, 1024: 36, 1536: 18}
            out[f"M{M}"] = {"traffic_bytes_per_launch": sum(cls[N]["traffic_bytes"] * wgt[N] for N in wgt) / 109.0,
                           "algorithmic_bytes_per_launch": sum(cls[N]["algorithmic_bytes"] * wgt[N] for N in wgt) / 109.0}
    best = out.get("M22320") or out.get("M7440")
    if best:
        out["traffic_bytes_per_launch"] = best["traffic_bytes_per_launch"]
        out["algorithmic_bytes_per_launch"] = best["algorithmic_bytes_per_launch"]
        out["rows_of_those_launches"] = 22320 if "M22320" in out else 7440
    dst = os.environ.get("CSS_TRAFFIC_JSON")
    if dst:
        with open(dst, "w") as f:
            json.dump(out, f, indent=1)
