#!/bin/bash
# round 3, GPU call 27: soak test of the dataflow factorisation with workgroup retirement (and without)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_27
for r in 16 99 16 16; do
env HIOPAMD_DF_RETIRE=$r timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_27/soak_$r.log 2>&1; echo "RETIRE=$r exit $?: $(tail -1 gpurun_out/r03_27/soak_$r.log | cut -c1-200)"
grep -i "timed out" gpurun_out/r03_27/soak_$r.log | cut -c1-500
done
for n in 8191 4096 2049 1536 777; do
env DF_N=$n DF_REPS=200 DF_OBJECTS=2 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_27/soak_n$n.log 2>&1; echo "N=$n exit $?: $(tail -1 gpurun_out/r03_27/soak_n$n.log | cut -c1-200)"
grep -i "timed out" gpurun_out/r03_27/soak_n$n.log | cut -c1-500
done
