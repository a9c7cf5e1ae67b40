#!/bin/bash
# Next round, first GPU call: the frozen workgroups of the dataflow LDL^T.
#  (a) soak with the debug dump until a bounded wait expires: does any workgroup publish its task on another CU than it took it on
#      (= it was context-saved and restored)?  what do the driver's eviction counters say at that moment?
#  (b) the soak with ONE wide workgroup per CU to 1e5 factorisations (round 3: none in 41 600).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_first
for i in $(seq 1 30); do
  env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r04_first/soak_$i.log 2>&1; rc=$?
  echo "soak $i exit $rc: $(tail -1 gpurun_out/r04_first/soak_$i.log | cut -c1-160)"
  if grep -q "timed out" gpurun_out/r04_first/soak_$i.log; then
    grep "bounded wait\|another CU\|in the tile loop at stage\|wide kernel:\|kfd evicted" gpurun_out/r04_first/soak_$i.log | cut -c1-220 | head -40
    break
  fi
done
for i in $(seq 1 62); do
  env HIOPAMD_DF_WGS=240 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r04_first/one_per_cu_$i.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "one workgroup per CU: soak $i exit $rc"; grep "bounded wait\|failed after" gpurun_out/r04_first/one_per_cu_$i.log | head -3; break; fi
done
echo "one workgroup per CU: $i soaks of 1600 factorisations"
