#!/bin/bash
# round 3, GPU call 26: repeat the retirement setting, keep the diagnostics of a bounded-wait time-out if one happens
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_26
for i in 1 2 3 4 5 6 7 8 9 10; do
env HIOPAMD_DF_RETIRE=16 DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py > gpurun_out/r03_26/run_$i.log 2>&1
echo "run $i: $(grep -c 'timed out' gpurun_out/r03_26/run_$i.log) $(tail -1 gpurun_out/r03_26/run_$i.log | cut -c1-100)"
grep "timed out" -A2 gpurun_out/r03_26/run_$i.log | cut -c1-400
done
