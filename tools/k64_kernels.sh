#!/bin/bash
# per-kernel durations of C3 V-cycles with k right-hand-side columns (rocprofv3 kernel trace): usage tools/k64_kernels.sh [k]
K=${1:-64}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/k64_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python tools/prof_kernels.py --workload C3 --k $K --reps 3 --cycles 10 --smoother gs > $OUT/run.log 2>&1
python tools/rocpd_stats.py $OUT/t_results.db $OUT/k${K}_kernel_stats.csv | head -30
python tools/rocpd_timeline.py $OUT/t_results.db > $OUT/k${K}_timeline.txt; head -70 $OUT/k${K}_timeline.txt
rm -f $OUT/t_results.db
