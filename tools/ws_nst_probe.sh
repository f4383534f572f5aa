for v in 0 1 0 1 0 1; do
  CSS_WD_GLOBAL=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-long 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GLOBAL=$v', 'value', d['value'], 'dev', d['device_resident']['ms_per_step'], 'gemm_ms', d['kernel_family_ms']['linear_gemm'], 'frac', d['roofline']['frac'])"
done
