#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun).  Outputs under gpurun_out/prof_round/.
#   tools/profile_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/prof_round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. kernel trace + stats of the very command the driver runs
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
python tools/rocpd_stats.py $OUT/trace/bench_results.db $OUT/${TAG}_bench_kernel_stats.csv > /dev/null
python tools/rocpd_timeline.py $OUT/trace/bench_results.db > $OUT/${TAG}_vcycle_timeline.txt
# 2. PMC passes (counters only, each in its own run), fine-level kernels isolated by grid size
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr " " "_")
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$n -o p -- python tools/prof_kernels.py --reps 20 --cycles 3 > $OUT/pmc_$n.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/${TAG}_pmc_summary.json
rm -rf $OUT/trace $OUT/pmc_*/  # keep the summaries only (the dbs are large)
ls -la $OUT
