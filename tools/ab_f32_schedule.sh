#!/bin/bash
# A/B of the schedule knobs (run on the GPU box) in both arithmetic modes: lanes of one session's pass (css_run), of a shared batch, sessions per batch
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --no-cpu-baseline --no-long --min-seconds 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['split_f16']; print('%-28s f32: queue %.3f ms, css_run %.3f ms, device-resident %.3f ms   split: queue %.3f, css_run %.3f, device-resident %.3f' % ('$*', d['ms_per_step'], d['synchronous_call']['ms_per_step'], d['device_resident']['ms_per_step'], s['ms_per_step'], s['synchronous_call']['ms_per_step'], s['device_resident']['ms_per_step']))"; }
for rep in 1 2; do
run
run --lanes 1
run --lanes 2
run --lanes 4
done
