#!/bin/bash
# Counters of the float32 attention kernel alone (tools/bin/aw_1 = tools/attn_bench.hip; row-major and fragment operands), on the GPU box.
# Separate --pmc passes, each under its own timeout (TA_* counters hang rocprofv3 on this pool and are not asked for).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=${1:-120}
B=${2:-tools/bin/aw_1}
run() { d=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$d -o p -- $B $N > /dev/null 2>&1; }
run attf1 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run attf2 GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
run attf3 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
run attf4 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
python - <<'PY'
import pandas as pd, glob
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
for d in ("attf1", "attf2", "attf3", "attf4"):
    f = glob.glob("gpurun_out/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f: print("no counters in", d); continue
    c = pd.read_csv(f[0])
    c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
    c["k"] = c["Kernel_Name"].str.replace("void css::", "").str.slice(0, 60)
    t = c.pivot_table(index=["Dispatch_Id", "k", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    g = t.groupby("k").mean(numeric_only=True).drop(columns=["Dispatch_Id"])
    print(g.round(0).to_string())
PY
