#!/bin/bash
# round 3, GPU call 46: soak until a time-out; the dump now shows what every workgroup held AT the time-out (snapshot by the waiter)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_46
for i in $(seq 1 34); do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_46/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_46/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_46/soak_$i.log; then grep "hiop_amd\|failed after" gpurun_out/r03_46/soak_$i.log | grep -v "chain role [0-9]\|writers of" | cut -c1-240 | head -70; break; fi
done
