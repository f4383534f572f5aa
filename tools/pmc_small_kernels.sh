#!/bin/bash
# Where the waves of the non-GEMM kernels spend their cycles: two PMC passes (8 SQ counters each) over a single-lane
# device-resident run, summarised per kernel.   bash tools/pmc_small_kernels.sh   (GPU box, through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc_small
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/a -o p -- python tools/trace_pass.py 60 3 device 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $o/b -o p -- python tools/trace_pass.py 60 3 device 1 > /dev/null 2>&1
python - <<PY
import glob, pandas as pd
def load(d):
    f = glob.glob(f"$o/{d}/**/p_counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f)
    df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void css::", "").str.replace("css::", "").str.slice(0, 42)
    return df.pivot_table(index=["k", "Dispatch_Id"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index().groupby("k").mean(numeric_only=True)
a, b = load("a"), load("b")
t = a.join(b, lsuffix="_a", rsuffix="_b")
keep = [k for k in t.index if any(s in k for s in ("features", "scm_kernel", "layernorm", "conv_module", "relpos", "beamform", "mvdr_solve", "ola_", "stft_fft", "pit_cost", "wave_ola", "deinterleave"))]
rows = []
for k in keep:
    r = t.loc[k]
    wc = r["SQ_WAVE_CYCLES"]
    rows.append((k, int(r["SQ_WAVES"]), round(wc / r["SQ_WAVES"]), f"{100*r['SQ_ACTIVE_INST_ANY']/wc:.0f}%", f"{100*r['SQ_WAIT_INST_ANY']/wc:.0f}%", f"{100*r['SQ_WAIT_ANY']/wc:.0f}%",
                 f"{100*r['SQ_ACTIVE_INST_VALU']/wc:.0f}%", f"{100*r['SQ_ACTIVE_INST_LDS']/wc:.0f}%", f"{100*r['SQ_ACTIVE_INST_VMEM']/wc:.0f}%",
                 round(r["SQ_INSTS_VALU"] / r["SQ_WAVES"]), round(r["SQ_INSTS_LDS"] / r["SQ_WAVES"]), round((r["SQ_INSTS_VMEM_RD"] + r["SQ_INSTS_VMEM_WR"]) / r["SQ_WAVES"]), round(r["SQ_INSTS_SALU"] / r["SQ_WAVES"]),
                 round(r["SQ_LDS_BANK_CONFLICT"] / r["SQ_WAVES"])))
print("| kernel | waves | quad-cycles per wave | issuing | issue-stalled | parked | VALU busy | LDS busy | VMEM busy | VALU instr / wave | LDS instr | VMEM instr | SALU instr | LDS conflict cycles / wave |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for r in rows: print("| " + " | ".join(str(x) for x in r) + " |")
PY
rm -rf $o
