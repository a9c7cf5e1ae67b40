#!/bin/bash
# round 3, GPU call 23: phase accounting of the fused update tasks with ONE workgroup per CU (240) against two (480)
set -u
export TMPDIR=/tmp
for w in 480 240 360 120; do
echo "=== HIOPAMD_DF_WGS=$w"
env HIOPAMD_DF_WGS=$w DF_TIMELINE=1 DF_MODES=5,1 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | grep "wide kernel phases\|matrixChanged" | cut -c1-330
done
