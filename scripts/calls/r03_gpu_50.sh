#!/bin/bash
# round 3, GPU call 50: does a time-out coincide with an EVICTION of the process's queues by the kernel driver (KFD evicted_ms)?
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_50
ls /sys/class/kfd/kfd/proc/*/ 2>&1 | head -12
for i in $(seq 1 30); do
env DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_50/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(grep 'kfd evicted' gpurun_out/r03_50/soak_$i.log | tr '\n' ' ' | cut -c1-260)"
if grep -q "timed out" gpurun_out/r03_50/soak_$i.log; then grep "bounded wait\|failed after\|kfd" gpurun_out/r03_50/soak_$i.log | cut -c1-220 | head; break; fi
done
