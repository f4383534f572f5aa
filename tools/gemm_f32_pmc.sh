#!/bin/bash
# SQ counters of the exact float32 GEMM (gemm_f32.hip) on the census cases of tools/gemm_f32_bench.hip (run on the GPU box):
#   bash tools/gemm_f32_pmc.sh [binary]  ->  gpurun_out/f32_pmc*/ + a per-kernel table on stdout
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BIN=${1:-tools/bin/gemm_f32_bench}
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/f32_pmc1 -o p -- $BIN > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/f32_pmc2 -o p -- $BIN > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/f32_pmc3 -o p -- $BIN > /dev/null 2>&1
python - <<'PY'
import pandas as pd, glob
pd.set_option("display.width", 300); pd.set_option("display.max_columns", 40)
for d in ("gpurun_out/f32_pmc1", "gpurun_out/f32_pmc2", "gpurun_out/f32_pmc3"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file in", d); continue
    c = pd.read_csv(f[0])
    c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
    c["k"] = c["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 30)
    t = c.pivot_table(index=["Dispatch_Id", "k", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    t["bucket"] = (t["dur"] / 20000).round() * 20   # us buckets separate the launch shapes
    g = t[t.k.str.contains("gemm")].groupby(["k", "bucket"]).mean(numeric_only=True).drop(columns=["Dispatch_Id"])
    print(g.round(0).to_string())
PY
