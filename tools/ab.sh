#!/bin/bash
# A/B of kernel tuning knobs on one workload: prints value / spmv us / gs sweep us per configuration
WL=${1:-C3}
for C in 64 128; do for R in 0 1; do
  SMG_SELL_C=$C SMG_REGION_ORDER=$R python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C=$C region=$R  vcyc/s %.1f  ms/step %.4f  spmv %.2f us (%.0f GB/s, %.1f%%)  gs_sweep %.2f us  pad %.3f' % (d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['achieved'], 100*d['roofline']['frac'], d['roofline_gs_sweep']['us_per_sweep'], d['roofline']['sell_padding']))"
done; done
