#!/bin/bash
# Round 4, GPU call 6: what bounds the eight-wave tile loop?  Timing experiments of the profiling instantiation (HIOPAMD_DF_EXP: 1 no
# operand loads, 2 no stage barrier, 4 no LDS operand reads; the factor is garbage, the phase sums are what is read) and the
# shader clock inside the loop; the hand-out check after the fix of the publication race; recovery / dump tests.
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_6
mkdir -p $O
t() { name=$1; shift; env "$@" DF_TIMELINE=${TL:-0} timeout -s KILL 150 python scripts/df_stamps.py > $O/$name.log 2>&1; echo "$name: exit $? | $(grep -h 'matrixChanged' $O/$name.log | tr '\n' ' ')"; }
for e in 0 1 2 4 3 7; do TL=1 t exp$e HIOPAMD_DF_PIPE=0 HIOPAMD_DF_EXP=$e; grep -h "wide kernel phases\|shader clock" $O/exp$e.log | cut -c1-330; done
TL=1 t legacy240_stamps HIOPAMD_DF_FORM=4
grep -h "wide kernel phases" $O/legacy240_stamps.log | cut -c1-330
t w8_pipe3 HIOPAMD_DF_PIPE=3
t w8_pipe0 HIOPAMD_DF_PIPE=0
env HIOPAMD_DF_CHECK=1 DF_REPS=150 DF_OBJECTS=2 DF_VERIFY=1 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_check.log 2>&1; echo "soak(check) exit $?: $(tail -1 $O/soak_check.log | cut -c1-200)"
timeout -s KILL 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_zz_gpu_dataflow_debug_dump.py tests/test_gpu_dense_sparse.py -x -q > $O/pytest.log 2>&1; echo "pytest exit $?: $(tail -3 $O/pytest.log | tr '\n' ' ')"
