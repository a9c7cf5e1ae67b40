#!/bin/bash
# round 3, GPU call 36: soak until a time-out, with the dump of what every waiting task misses
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_36
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_36/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_36/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_36/soak_$i.log; then grep "hiop_amd\|failed after" gpurun_out/r03_36/soak_$i.log | grep -v "chain role [0-9]\|holds subst" | cut -c1-220 | head -120; break; fi
done
