#!/bin/bash
# A/B of an environment knob inside one GPU run, order alternated (the first run of a pair is ~1.5 % slower):
#   tools/ab.sh VAR a b [workload] [rounds]
export SMG_EXPECT_GPU=1
for r in $(seq 1 ${5:-2}); do
if [ $((r % 2)) = 1 ]; then order="$2 $3"; else order="$3 $2"; fi
for v in $order; do
echo "--- $1=$v (round $r)"
env $1=$v timeout 300 bash tools/quick.sh ${4:-C3} 900
done
done
for v in $2 $3; do
echo "--- $1=$v"
env $1=$v timeout 300 python tools/level_times.py ${4:-C3} 2>&1 | tail -5
done
