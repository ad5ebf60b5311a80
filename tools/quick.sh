#!/bin/bash
# quick perf line: tools/quick.sh [workload] [steps] [smoother]      (reads the FULL record, bench_extra.json: stdout's last line is the compact one)
X=$(mktemp -d)
SMG_BENCH_EXTRA_DIR=$X python bench.py --workload ${1:-C3} --steps ${2:-800} --warmup 100 --no-cpu --smoother ${3:-gs} --no-c5 --no-c4 --no-c3k3 --no-c3k64 --no-reprecompute --no-c3dec --no-c1 --no-block3 --no-multi-mesh >/dev/null 2>&1
python -c "
import json,sys
d=json.load(open('$X/bench_extra.json'))
print('vcyc/s %.1f  ms/step %.4f  spmv %.2f us (%.1f%%)  gs_sweep %.2f us  colors %s  live %.1f MB' % (d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], 100*d['roofline']['frac'], d['roofline_gs_sweep']['us_per_sweep'], d['config']['colors'], d['device_bytes']['libsmg_live'] / 1e6))
for k, v in (d.get('smoothers') or {}).items(): print('  %-13s %-7s cycles_to_tol %3d  ms/step %.4f  time_to_tol %.3f ms  solve_wall %.3f ms' % (k, v['smoother'], v['cycles_to_tol'], v['ms_per_step'], v['time_to_tol_ms'], v['solve_wall_ms']))"
rm -rf $X
