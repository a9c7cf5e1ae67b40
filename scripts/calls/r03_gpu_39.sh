#!/bin/bash
# round 3, GPU call 39: waiters poll with a read-modify-write (+0) instead of a load: timing, long soak
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_39
for r in 1 2; do env DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1; done
for i in $(seq 1 40); do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_39/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_39/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_39/soak_$i.log; then grep "dataflow LDL\|failed after\|super-panel [0-9]*: update" gpurun_out/r03_39/soak_$i.log | cut -c1-260 | head -12; break; fi
done
