#!/usr/bin/env python3
"""The memory-bound kernels either side of the mask estimator on the 30-min meeting: per kernel (mean per launch over two device-resident
passes) the duration, the HBM bytes the counters saw -- FETCH_SIZE (doubled: gfx950's wide reads, MI355X_MICROARCH.md) + WRITE_SIZE, KiB
units, separate passes --, the algorithmic bytes of SURVEY.md 8(d) (bench.py hbm_kernel_bytes), and where the wave cycles went.
    python tools/summarize_pmc_small.py gpurun_out/r06_pmc1800 > gpurun_out/r06_pmc_small_kernels.md"""
import glob
import importlib
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
o = sys.argv[1]
NAMES = {"deinterleave_kernel": "deinterleave", "stft_fft_kernel": "stft", "features_kernel": "features", "scm_kernel": "scm", "mvdr_solve": "mvdr_solve",
         "beamform_kernel": "beamform", "ola_masks_kernel": "ola_masks", "ola_stft_kernel": "ola_stft", "wave_ola_kernel": "wave_ola", "pit_cost_kernel": "pit"}


def load(d):
    f = glob.glob(f"{o}/{d}/**/p_counter_collection.csv", recursive=True)[0]
    c = pd.read_csv(f)
    c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
    c["k"] = c["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void css::", "").str.replace("css::", "").str.replace(r"<.*", "", regex=True)
    c = c[c["k"].isin(NAMES)]
    return c.pivot_table(index=["Dispatch_Id", "k", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()


a, f, w = load("a"), load("f"), load("w")
# algorithmic bytes of the same meeting
pkg = lambda n: importlib.import_module("notsofar1_challenge_amd." + n)
W, CSS, L = pkg("weights"), pkg("css"), pkg("_lib")
bench = importlib.import_module("bench")
desc = W.ModelDesc.mc_v1()
run_cfg = CSS.make_run_cfg(CSS.CssCfg(activity_th=0.3, show_progressbar=False), 16000, 7)
n = 1800 * 16000
plan = L.plan(desc, run_cfg, n)
alg = bench.hbm_kernel_bytes(plan, desc, 186, 93, n)
print("# HBM counters of the memory-bound kernels on the 30-min meeting (round 6; `bash tools/profile_round.sh r06 small`)\n")
print("Two device-resident passes of the 1 800 s / 1 209-segment meeting (`tools/trace_pass.py 1800 2 device 1`, exact float32 mode, one lane) under")
print("`rocprofv3 --kernel-trace --pmc <set>`, three separate runs (SQ set | FETCH_SIZE | WRITE_SIZE).  Per kernel FAMILY and PASS (a family's launches")
print("of one pass summed; the estimator batches split features / covariance / beamformer into ten launches each): time, HBM bytes the")
print("counters saw (FETCH_SIZE x 2 + WRITE_SIZE, KiB units), SURVEY 8(d)'s algorithmic bytes (bench.py `hbm_kernel_bytes`), their ratio, the")
print("rates against 8 TB/s, and the share of wave cycles parked at `s_waitcnt` / barriers (SQ_WAIT_ANY) or issuing (SQ_ACTIVE_INST_ANY).\n")
print("| kernel family | launches / pass | us / pass | FETCH x2 MB | WRITE MB | counted MB | algorithmic MB | counted / algorithmic | algorithmic TB/s (frac of 8) | counted TB/s | parked | issuing | VALU busy |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, fam in NAMES.items():
    aa, ff, ww = a[a["k"] == k], f[f["k"] == k], w[w["k"] == k]
    if not len(aa) or not len(ff) or not len(ww):
        continue
    passes = 2.0
    us = aa["dur"].sum() / 1e3 / passes
    fetch = 2 * ff["FETCH_SIZE"].sum() * 1024 / passes
    write = ww["WRITE_SIZE"].sum() * 1024 / passes
    wc = aa["SQ_WAVE_CYCLES"].sum()
    al = alg.get(fam)
    row = [fam, f"{len(aa) / passes:.0f}", f"{us:.0f}", f"{fetch / 1e6:.0f}", f"{write / 1e6:.0f}", f"{(fetch + write) / 1e6:.0f}"]
    if al:
        row += [f"{al / 1e6:.0f}", f"{(fetch + write) / al:.2f}", f"{al / us / 1e6:.2f} ({al / us / 1e6 / 8:.2f})"]
    else:
        row += ["", "", ""]
    row += [f"{(fetch + write) / us / 1e6:.2f}", f"{100 * aa['SQ_WAIT_ANY'].sum() / wc:.0f}%", f"{100 * aa['SQ_ACTIVE_INST_ANY'].sum() / wc:.0f}%",
            f"{100 * aa['SQ_ACTIVE_INST_VALU'].sum() / wc:.0f}%"]
    print("| " + " | ".join(row) + " |")
