#!/bin/bash
# round 3, GPU call 29: the same soak with the library of the last commit (before this session's changes to the wide kernel): is the rare
# lost-task time-out older than today?
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_29
cp gpurun_exp_libhiopamd.so hiop_amd/lib/libhiopamd.so
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do
env DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_29/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_29/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_29/soak_$i.log; then grep "hiop_amd\|failed after" gpurun_out/r03_29/soak_$i.log | cut -c1-300 | head; break; fi
done
