TAG=r04; OUT=gpurun_out/prof_round; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ SMG_DEBUG_SCHUR=1 python tools/schur_time.py C3 2>&1 | grep -v amdgpu.ids; echo; echo "== mean-curvature-flow step (tools/mcf_step_time.py)"; python tools/mcf_step_time.py 2>/dev/null | tail -4;
  echo; echo "== block (3-DOF) benchmark system (tools/block_reprecompute.py)"; (cd tools && SMG_DEBUG_SCHUR=1 python block_reprecompute.py 2>&1 | grep -v amdgpu.ids | tail -4);
  echo; echo "== large coarsest levels (tools/coarse_big.py)"; python tools/coarse_big.py 2>&1 | grep -v "amdgpu.ids\|smg schur" | tail -6; } > $OUT/${TAG}_schur.txt
rocprofv3 --kernel-trace --stats -d $OUT/sch -o t -- python tools/schur_prof.py > $OUT/sch.log 2>&1
python tools/rocpd_stats.py $OUT/sch/t_results.db $OUT/${TAG}_schur_kernel_stats.csv > /dev/null
rm -rf $OUT/sch
cat $OUT/${TAG}_schur.txt
