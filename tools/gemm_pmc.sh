#!/bin/bash
# SQ counters of the GEMM variants on one long-K shape (run on the GPU box through gpurun):
#   bash tools/gemm_pmc.sh  ->  gpurun_out/gemm_pmc/*.csv + a per-kernel table on stdout
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip notsofar1-challenge_amd/csrc/gemm.hip notsofar1-challenge_amd/csrc/gemm_split.hip notsofar1-challenge_amd/csrc/gemm_split_wd.hip -Inotsofar1-challenge_amd/csrc -o /tmp/gemm_bench 2>/dev/null
export GEMM_BENCH_ONLY="K=4096"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/gemm_pmc -o p -- /tmp/gemm_bench > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/gemm_pmc2 -o p -- /tmp/gemm_bench > /dev/null 2>&1
python - <<'PY'
import pandas as pd, glob
for d in ("gpurun_out/gemm_pmc", "gpurun_out/gemm_pmc2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file in", d); continue
    c = pd.read_csv(f[0])
    c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
    c["k"] = c["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 48)
    t = c.pivot_table(index=["Dispatch_Id", "k", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    g = t.groupby("k").mean(numeric_only=True).drop(columns=["Dispatch_Id"])
    pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
    print(g.round(0).to_string())
PY
