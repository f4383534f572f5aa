#!/bin/bash
# round-4 check E: grouped passes with the transforms on the copy stream and everything behind the mask head on the tail
# stream; streams dealt onto hardware queues by measurement
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_coverage.json
( timeout 1500 python -m pytest tests -m gpu -q --timeout 400 2>&1 | tail -25 ) > gpurun_out/r4e_tests.txt
run() { tag=$1; shift; ( timeout 300 python bench.py --steps 21 --warmup 6 --no-long --no-cpu-baseline --min-seconds 2 "$@" > gpurun_out/r4e_bench_$tag.json 2> gpurun_out/r4e_bench_$tag.err ); echo "$tag rc=$?" >> gpurun_out/r4e_tests.txt; }
run g8 ; run g1 --queue-group 1 ; run g8l2 --lanes 2 ; run g8again
( timeout 900 python tools/rccl_slowdown_probe.py 1800 > gpurun_out/r4e_rccl_probe.json 2> gpurun_out/r4e_rccl_probe.err ); echo "probe rc=$?" >> gpurun_out/r4e_tests.txt
cat gpurun_out/r4e_tests.txt
for t in g8 g1 g8l2 g8again; do python - <<PY
import json
d=json.loads(open("gpurun_out/r4e_bench_$t.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$t", d["value"], d["ms_per_step"], "roof", r["achieved"], r["frac"], r["avg_launch_us"], d["runs_ms"]["min"], d["runs_ms"]["max"], "sync", d["synchronous_call"]["ms_per_step"], "dev", d["device_resident"]["ms_per_step"])
PY
done
python - <<PY
import json
s=open("gpurun_out/r4e_rccl_probe.json").read()
d,_=json.JSONDecoder().raw_decode(s)
for k,v in d.items(): print(k, json.dumps(v))
PY
