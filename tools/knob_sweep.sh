#!/bin/bash
# Sweep of the A/B knobs around their defaults on one box ("the defaults are what was measured best"): us per outer iteration's V-cycle from level 0
# (tools/level_times.py, graph-replayed).  usage: tools/knob_sweep.sh > gpurun_out/knob_sweep.txt
run() { # workload, then VAR=value ...
  wl=$1; shift
  t=$(env "$@" timeout 300 python tools/level_times.py $wl 2>/dev/null | grep "^level 0" | sed -E 's/.*cycle from here +([0-9.]+) us.*/\1/')
  printf "%-8s %-40s %s us\n" "$wl" "$*" "$t"
}
for rep in 1 2; do run C3 SMG_NONE=0; done
for v in 128 256 384 512; do run C3 SMG_TILED_ROWS=$v; done
for v in 256 512; do run C3 SMG_TILED_NT=$v; done
for v in 0; do run C3 SMG_TILED=$v; done
for v in 2 8; do run C3 SMG_GS_WPB=$v; done
for v in 0 16 64 128; do run C3 SMG_ONE_XCD_MAX=$v; done
for v in 0; do run C3 SMG_FUSE_HEAD=$v; run C3 SMG_FUSE_FIRST=$v; run C3 SMG_SYM_COARSE=$v; run C3 SMG_REGION_ORDER=$v; run C3 SMG_TRANSFER_REGION_ORDER=$v; run C3 SMG_SELL_STRIDE=$v; done
for v in 1 2 8; do run C3 SMG_GRAPH_ITERS=$v; done
for v in 1 4; do run C3 SMG_COARSE_CPW=$v; done
for rep in 1 2; do run ogre SMG_NONE=0; done
for v in 128 256 384; do run ogre SMG_TILED_ROWS=$v; done
for v in 0 16 64; do run ogre SMG_ONE_XCD_MAX=$v; done
for v in 0; do run ogre SMG_DEEP=$v; run ogre SMG_WGS_PIECES=$v; run ogre SMG_FUSE_FIRST=$v; done
for v in 9 17 25; do run ogre SMG_DEEP_MIN_W=$v; done
for v in 32 48; do run ogre SMG_WGS_ROWS=$v; done
for rep in 1 2; do run C3pdec SMG_NONE=0; done
for v in 0; do run C3pdec SMG_DEEP=$v; run C3pdec SMG_WGS_PIECES=$v; done
for v in 32 48; do run C3pdec SMG_WGS_ROWS=$v; done
for v in 9 17 25; do run C3pdec SMG_DEEP_MIN_W=$v; done
for v in 33 129; do run C3pdec SMG_LONG_ROW_MIN=$v; done
