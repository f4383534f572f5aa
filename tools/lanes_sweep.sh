for l in 1 2 3 4; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-long --lanes $l 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes=$l value', d['value'], 'ms', d['ms_per_step'], 'dev', d['device_resident']['ms_per_step'], 'sync', d['synchronous_call']['ms_per_step'])"; done
