#!/bin/bash
# PMC passes (counters only, one counter set per run) + kernel trace for the wide (k >= 8) kernels: C3 with k columns.
#   tools/pmc_wide.sh [tag] [k ...]     outputs gpurun_out/pmc_wide/<tag>_pmc_summary_C3_k<k>.json, <tag>_c3_k<k>_kernel_stats.csv
TAG=${1:-r04}; shift
KS=${@:-"64 8"}
OUT=gpurun_out/pmc_wide
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in $KS; do
  D=$OUT/k$k; mkdir -p $D
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr " " "_")
    rocprofv3 --pmc $c --kernel-trace -d $D/pmc_$n -o p -- python tools/prof_kernels.py --workload C3 --k $k --reps 3 --cycles 4 --smoother gs > $D/pmc_$n.log 2>&1
  done
  python tools/pmc_summary.py $D > $OUT/${TAG}_pmc_summary_C3_k$k.json
  rocprofv3 --kernel-trace --stats -d $D/t -o t -- python tools/prof_kernels.py --workload C3 --k $k --reps 3 --cycles 10 --smoother gs > $D/t.log 2>&1
  python tools/rocpd_stats.py $D/t/t_results.db $OUT/${TAG}_c3_k${k}_kernel_stats.csv > /dev/null
  python tools/rocpd_timeline.py $D/t/t_results.db > $OUT/${TAG}_c3_k${k}_timeline.txt
  rm -rf $D
done
ls -la $OUT
