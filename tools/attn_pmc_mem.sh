#!/bin/bash
# (each pass under its own timeout; passes with TA_* counters -- TA_TA_BUSY, TA_ADDR_STALLED_BY_TC, TA_BUSY_avr -- did not
# return on this pool and were dropped)
# Memory-pipe counters of the attention kernel alone (tools/attn_bench.hip, 40 segments), GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/attn_bench.hip notsofar1-challenge_amd/csrc/encoder.hip -Inotsofar1-challenge_amd/csrc -o /tmp/ab 2>/dev/null
N=${1:-40}
timeout 90 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/attm2 -o p -- /tmp/ab $N > /dev/null 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/attm3 -o p -- /tmp/ab $N > /dev/null 2>&1
python - <<'PY'
import pandas as pd, glob
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
for d in ("gpurun_out/attm2", "gpurun_out/attm3"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print("no counters in", d); continue
    c = pd.read_csv(f[0])
    c["dur"] = c["End_Timestamp"] - c["Start_Timestamp"]
    c["k"] = c["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 44)
    t = c.pivot_table(index=["Dispatch_Id", "k", "dur"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    g = t.groupby("k").mean(numeric_only=True).drop(columns=["Dispatch_Id"])
    print(g.round(0).to_string())
PY
