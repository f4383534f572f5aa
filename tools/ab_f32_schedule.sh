#!/bin/bash
# A/B of the schedule knobs (run on the GPU box) in both arithmetic modes: lanes of a shared batch, rows per split-mode batch
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --no-cpu-baseline --no-long --min-seconds 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['split_f16']; print('%-44s f32 %.3f ms (%d / batch)   split %.3f ms (%d / batch)' % ('$*', d['ms_per_step'], d['sessions_per_estimator_batch'], s['ms_per_step'], s['sessions_per_estimator_batch']))"; }
for rep in 1 2; do
run
run --tune group_lanes=3
run --tune split_batch_rows=30000
run --tune split_batch_rows=16000
run --tune split_batch_rows=30000 --tune group_lanes=3
done
