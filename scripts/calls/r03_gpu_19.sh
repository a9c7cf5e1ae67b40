#!/bin/bash
# round 3, GPU call 19: phase accounting of the wide kernel restricted to super-panels of the update-bound phase
set -u
mkdir -p gpurun_out/r03_19
export TMPDIR=/tmp
DF_MODES=5,7,11,15,6 timeout 600 python scripts/df_stamps.py > gpurun_out/r03_19/stamps.log 2>&1; echo "exit $?"
grep -v "^  [ 0-9][0-9] |" gpurun_out/r03_19/stamps.log | cut -c1-400 | tail -60
