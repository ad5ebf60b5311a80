#!/bin/bash
# quick perf line: tools/quick.sh [workload] [steps] [smoother]
python bench.py --workload ${1:-C3} --steps ${2:-800} --warmup 100 --no-cpu --smoother ${3:-hybrid_chebyshev} --no-c5 --no-c4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('vcyc/s %.1f  ms/step %.4f  spmv %.2f us (%.1f%%)  gs_sweep %.2f us  colors %s' % (d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], 100*d['roofline']['frac'], d['roofline_gs_sweep']['us_per_sweep'], d['config']['colors']))
for k, v in (d.get('smoothers') or {}).items(): print('  %-13s %-7s cycles_to_tol %3d  ms/step %.4f  time_to_tol %.3f ms  solve_wall %.3f ms' % (k, v['smoother'], v['cycles_to_tol'], v['ms_per_step'], v['time_to_tol_ms'], v['solve_wall_ms']))"
