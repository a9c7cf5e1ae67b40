#!/bin/bash
# round 3, GPU call 31: version words / substitution counters one cache line each — does the rare lost flag update go away?
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_31
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -q -x -k "dataflow or ragged or full_size" > gpurun_out/r03_31/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03_31/pytest.log
env DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24; do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > gpurun_out/r03_31/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 gpurun_out/r03_31/soak_$i.log | cut -c1-160)"
if grep -q "timed out" gpurun_out/r03_31/soak_$i.log; then grep "hiop_amd\|failed after" gpurun_out/r03_31/soak_$i.log | grep -v "chain role [0-9]" | cut -c1-250 | head -12; break; fi
done
