#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun).  Outputs under gpurun_out/prof_round/.
#   tools/profile_round.sh [tag]
TAG=${1:-r06}
OUT=gpurun_out/prof_round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. kernel trace + stats of the very command the driver runs (C3 main loop = the reference's Gauss-Seidel cycle, smoother comparison, C5, C3 x 64 columns,
#    C4 legs; the block leg's scalar comparison is left out: host precompute that adds nothing to the kernel table; the multi-mesh leg too:
#    its 8 host threads launching at once crash rocprofv3's tool thread)
SMG_BENCH_EXTRA_DIR=$OUT rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu --no-block3-scalar --no-multi-mesh > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/trace.err
mv $OUT/bench_extra.json $OUT/${TAG}_bench_extra_under_rocprof.json 2>/dev/null
python tools/rocpd_stats.py $OUT/trace/bench_results.db $OUT/${TAG}_bench_kernel_stats.csv > /dev/null
# 2. one outer iteration of the timed configuration (GS above 300 k rows, Chebyshev-Jacobi below), kernel by kernel -- and of the
#    reference's Gauss-Seidel everywhere; and of the wide (k = 64) kernels of C4
rocprofv3 --kernel-trace -d $OUT/tl -o t -- python tools/prof_kernels.py --reps 5 --cycles 12 --smoother hybrid_chebyshev > $OUT/tl.log 2>&1
python tools/rocpd_timeline.py $OUT/tl/t_results.db > $OUT/${TAG}_vcycle_timeline_hybrid_chebyshev.txt
rocprofv3 --kernel-trace -d $OUT/tlg -o t -- python tools/prof_kernels.py --reps 5 --cycles 12 --smoother gs > $OUT/tlg.log 2>&1
python tools/rocpd_timeline.py $OUT/tlg/t_results.db > $OUT/${TAG}_vcycle_timeline_gs.txt
rocprofv3 --kernel-trace --stats -d $OUT/c4 -o t -- python tools/prof_kernels.py --workload C4k64 --reps 5 --cycles 30 --smoother hybrid_chebyshev > $OUT/c4.log 2>&1
python tools/rocpd_stats.py $OUT/c4/t_results.db $OUT/${TAG}_c4_k64_kernel_stats.csv > /dev/null
python tools/rocpd_timeline.py $OUT/c4/t_results.db > $OUT/${TAG}_c4_k64_timeline.txt
# 2b. the reference's own kind of hierarchy (mg_precompute) on the C3 mesh and on ogre.obj: kernel table + one outer iteration kernel by kernel
rocprofv3 --kernel-trace --stats -d $OUT/dec -o t -- python tools/prof_kernels.py --workload C3dec --reps 5 --cycles 12 --smoother gs > $OUT/dec.log 2>&1
python tools/rocpd_stats.py $OUT/dec/t_results.db $OUT/${TAG}_c3dec_kernel_stats.csv > /dev/null
python tools/rocpd_timeline.py $OUT/dec/t_results.db > $OUT/${TAG}_c3dec_timeline_gs.txt
rocprofv3 --kernel-trace -d $OUT/ogre -o t -- python tools/prof_kernels.py --workload ogre --reps 5 --cycles 12 --smoother gs > $OUT/ogre.log 2>&1
python tools/rocpd_timeline.py $OUT/ogre/t_results.db > $OUT/${TAG}_ogre_timeline_gs.txt
# 3. PMC passes (counters only, each in its own run), fine-level kernels isolated by grid size: C3 (in cache) and C5 (beyond it)
for wl in C3 C5 C3dec; do
  mkdir -p $OUT/$wl
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr " " "_")
    rocprofv3 --pmc $c --kernel-trace -d $OUT/$wl/pmc_$n -o p -- python tools/prof_kernels.py --workload $wl --reps 20 --cycles 3 > $OUT/$wl/pmc_$n.log 2>&1
  done
  python tools/pmc_summary.py $OUT/$wl > $OUT/${TAG}_pmc_summary_$wl.json
done
rocprofv3 --kernel-trace --stats -d $OUT/c5t -o t -- python tools/prof_kernels.py --workload C5 --reps 50 --cycles 5 > $OUT/c5t.log 2>&1
python tools/rocpd_stats.py $OUT/c5t/t_results.db $OUT/${TAG}_c5_kernel_stats.csv > /dev/null
# 3b. the wide (k >= 8) kernels a column-sharded job runs on every GPU: C3 with 64 and with 8 columns (8 = the 8-way shard of 64), counters + kernel table
tools/pmc_wide.sh $TAG 64 8 > $OUT/pmc_wide.log 2>&1
cp gpurun_out/pmc_wide/${TAG}_* $OUT/ 2>/dev/null
# 3c. the block (3-DOF) kernels: k_bsr3<...> on the C3 x 3 system (tools/prof_block3.py; rocprofv3 --pmc sometimes dies at start-up on this pool: up to three tries)
mkdir -p $OUT/B3
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr " " "_")
  for try in 1 2 3; do
    rm -rf $OUT/B3/pmc_$n
    rocprofv3 --pmc $c --kernel-trace -d $OUT/B3/pmc_$n -o p -- python tools/prof_block3.py C3 > $OUT/B3/pmc_$n.log 2>&1 && break
  done
done
python tools/pmc_summary.py $OUT/B3 > $OUT/${TAG}_pmc_summary_block3.json
python tools/make_traffic.py $OUT/${TAG}_pmc_summary_C3.json $OUT/${TAG}_pmc_summary_C5.json $TAG > $OUT/traffic.json
# 4. where the cycle's time goes, level by level (graph replay, hipEvents): the reference's Gauss-Seidel cycle with and without the one-launch
#    relax() of the small levels, and the Chebyshev hybrid
{ echo "== Gauss-Seidel everywhere (reference cycle)"; python tools/level_times.py 2>/dev/null; echo "== the same with SMG_TILED=0 (one launch per colour on every level)"; SMG_TILED=0 python tools/level_times.py 2>/dev/null;
  echo "== hybrid Chebyshev (GS above 300 k rows)"; SMG_TOOL_SMOOTHER=hybrid_chebyshev:300000 python tools/level_times.py 2>/dev/null; } > $OUT/${TAG}_level_times.txt
{ echo "== decimated hierarchies (mg_precompute): level table, per-level times, outer iteration"; python tools/dec_probe.py ogre C3pdec C3dec 2>/dev/null;
  echo "== the same with one launch per colour (SMG_WGS=0: round 4's path)"; SMG_WGS=0 python tools/dec_probe.py ogre C3dec 2>/dev/null | grep -v "^level [0-9] rows";
  echo "== relax(2) per level, graph-replayed"; python tools/wgs_probe.py C3dec 2 2>/dev/null | grep level; python tools/wgs_probe.py ogre 2 2>/dev/null | grep level;
  echo "== ogre.obj by column count"; python tools/c4_time.py 2>/dev/null | grep "k = "; } > $OUT/${TAG}_decimated.txt
python tools/mem_probe.py C3 > $OUT/${TAG}_device_bytes.txt 2>/dev/null
python tools/block3_time.py C3 > $OUT/${TAG}_block3_c3.txt 2>/dev/null
python tools/multi_mesh.py > $OUT/${TAG}_multi_mesh.txt 2>/dev/null
python tools/reprecompute_time.py > $OUT/${TAG}_reprecompute.txt 2>/dev/null
# 5. the coarse solver chosen by cost: dense inverse against the Schur-complement solver -- re-precompute, V-cycle and coarse solve per column count,
#    the flow step, the block system; the kernels of three re-precomputes and of the coarse solves under rocprofv3
bash tools/schur_legs.sh > /dev/null 2>&1
rm -rf $OUT/trace $OUT/tl $OUT/tlg $OUT/c4 $OUT/c5t $OUT/C3 $OUT/C5 $OUT/C3dec $OUT/B3 $OUT/sch $OUT/dec $OUT/ogre  # keep the summaries only (the dbs are large)
ls -la $OUT
