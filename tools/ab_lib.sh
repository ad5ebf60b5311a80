#!/bin/bash
# A/B of two builds of libsmg.so in one run, alternating order: tools/ab_lib.sh path/to/other.so [workload] [rounds]
export SMG_EXPECT_GPU=1
ALT=$(readlink -f $1)
for r in $(seq 1 ${3:-2}); do
if [ $((r % 2)) = 1 ]; then
echo "--- this build (round $r)";  timeout 300 bash tools/quick.sh ${2:-C3} 900
echo "--- other build (round $r)"; SMG_LIB=$ALT timeout 300 bash tools/quick.sh ${2:-C3} 900
else
echo "--- other build (round $r)"; SMG_LIB=$ALT timeout 300 bash tools/quick.sh ${2:-C3} 900
echo "--- this build (round $r)";  timeout 300 bash tools/quick.sh ${2:-C3} 900
fi
done
echo "--- other build"; SMG_LIB=$ALT timeout 300 python tools/level_times.py ${2:-C3} 2>&1 | tail -5
echo "--- this build"; timeout 300 python tools/level_times.py ${2:-C3} 2>&1 | tail -5
