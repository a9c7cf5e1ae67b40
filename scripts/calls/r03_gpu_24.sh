#!/bin/bash
# round 3, GPU call 24: time lines with 480 / 240 workgroups of the wide kernel — which half (update-bound / chain-bound) changes
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_24
for w in 480 240; do
env HIOPAMD_DF_WGS=$w DF_TIMELINE=1 DF_MODES=1 timeout -s KILL 180 python scripts/df_stamps.py > gpurun_out/r03_24/timeline_$w.txt 2>&1
done
paste <(grep "^  [ 0-9][0-9] |" gpurun_out/r03_24/timeline_480.txt | cut -c1-28) <(grep "^  [ 0-9][0-9] |" gpurun_out/r03_24/timeline_240.txt | cut -c6-28)
grep "spine\|inside F" gpurun_out/r03_24/timeline_480.txt gpurun_out/r03_24/timeline_240.txt | cut -c1-250
