#!/bin/bash
# A/B of the queue's schedule knobs (run on the GPU box): lanes of a shared batch, sessions per batch, in both arithmetic modes
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --no-cpu-baseline --no-long --min-seconds 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s f32 %.3f ms (%.0f x, frac %.3f, %d sessions / batch)   split %.3f ms (%d)' % ('$*', d['ms_per_step'], d['value'], d['roofline']['frac'], d['sessions_per_estimator_batch'], d['split_f16']['ms_per_step'], d['split_f16']['sessions_per_estimator_batch']))"; }
for rep in 1 2; do
run
run --max-batch 128
run --tune split_batch_rows=0
run --queue-group 3
done
