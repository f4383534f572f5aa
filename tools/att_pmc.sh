#!/bin/bash
# SQ counters of the attention kernel inside the real pipeline (GPU box, through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/att_pmc -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/att_pmc2 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import pandas as pd, glob
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
for d in ("gpurun_out/att_pmc", "gpurun_out/att_pmc2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    c = pd.read_csv(f[0])
    c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
    c["k"] = c["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 40)
    t = c.pivot_table(index=["Dispatch_Id", "k", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    g = t.groupby("k").mean(numeric_only=True).drop(columns=["Dispatch_Id"])
    g = g[g.index.str.contains("attn|gemm_split_wd|layernorm_kernel<2, 0>|dwconv|scm")]
    print((g / 1e3).round(0).to_string())
PY
