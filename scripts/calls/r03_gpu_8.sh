#!/bin/bash
# timelines of the dataflow factorisation with and without the fused K = 512 tasks
set -u
mkdir -p gpurun_out/r03_8
export TMPDIR=/tmp
O=gpurun_out/r03_8
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -q -x -k "dataflow_factorisation_equals_stepwise" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for k in 1073741824 0; do
  HIOPAMD_DF_K512=$k DF_TIMELINE=1 DF_MODES=1 timeout -s KILL 180 python scripts/df_stamps.py > $O/timeline_$k.txt 2>&1
  tail -60 $O/timeline_$k.txt
done
