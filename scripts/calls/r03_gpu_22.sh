#!/bin/bash
# round 3, GPU call 22: TIMING EXPERIMENT ONLY — operand loads of the update tile as plain (L2-cacheable) loads instead of sc1
# (a library built with -DHIOPAMD_DF_OPAUX=0; its factors may be wrong: stale L2 lines).  What would L2 reuse of the row panels be worth?
set -u
export TMPDIR=/tmp
echo "=== production library"
DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
cp hiop_amd/lib/libhiopamd.so /tmp/prod.so
cp gpurun_exp_libhiopamd.so hiop_amd/lib/libhiopamd.so
for cfg in "HIOPAMD_DF_ONE=0" "HIOPAMD_DF_ONE=1" "HIOPAMD_DF_K512=0"; do
echo "=== plain operand loads $cfg"
env $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
env DF_TIMELINE=1 DF_MODES=5 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | grep "wide kernel phases"
cp /tmp/prod.so hiop_amd/lib/libhiopamd.so
