#!/bin/bash
# round 3, GPU call 37: polling back-off in the wide kernel — timing, then the soak
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_37
for r in 1 2; do env DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1; done
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27 28 29 30; do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_37/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_37/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_37/soak_$i.log; then grep "hiop_amd\|failed after" gpurun_out/r03_37/soak_$i.log | grep -v "chain role [0-9]\|holds subst" | cut -c1-220 | head -40; break; fi
done
