#!/bin/bash
# round 3, GPU call 28/29: soak until a bounded wait times out; the library dumps what every chain role / wide workgroup held
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_28
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do
env DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_28/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_28/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_28/soak_$i.log; then grep "hiop_amd" gpurun_out/r03_28/soak_$i.log | cut -c1-300 | head -30; grep "failed after" gpurun_out/r03_28/soak_$i.log; break; fi
done
