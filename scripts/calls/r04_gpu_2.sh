#!/bin/bash
# Round 4, GPU call 2: eight-wave wide kernel with two operand stages in flight, 32-column substitution tasks on eight waves, the
# selection-ahead state machine; the retry copy of the solver object (recovery tests).
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_2
mkdir -p $O
t() { name=$1; shift; env "$@" DF_TIMELINE=${TL:-0} timeout -s KILL 150 python scripts/df_stamps.py > $O/$name.log 2>&1; echo "$name: exit $? | $(grep -h 'matrixChanged\|3 solves' $O/$name.log | tr '\n' ' ')"; }
t legacy480 HIOPAMD_DF_FORM=4 HIOPAMD_DF_WGS=480
t w8_pipe0 HIOPAMD_DF_PIPE=0
t w8_pipe1 HIOPAMD_DF_PIPE=1
t w8_pipe3 HIOPAMD_DF_PIPE=3
t w8_pipe3_all HIOPAMD_DF_PIPE=3 HIOPAMD_DF_PIPEJ=31
t w8_pipe3_j24 HIOPAMD_DF_PIPE=3 HIOPAMD_DF_PIPEJ=24
t w8_pipe3_k24 HIOPAMD_DF_PIPE=3 HIOPAMD_DF_K512=24 HIOPAMD_DF_PIPEJ=24
t w8_pipe3_k32 HIOPAMD_DF_PIPE=3 HIOPAMD_DF_K512=32 HIOPAMD_DF_PIPEJ=31
TL=1 t w8_pipe3_stamps HIOPAMD_DF_PIPE=3
TL=1 t w8_pipe0_stamps HIOPAMD_DF_PIPE=0
grep -h "wide kernel phases\|spine steps\|inside F\|spine wait" $O/w8_pipe3_stamps.log $O/w8_pipe0_stamps.log | cut -c1-400
timeout -s KILL 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_zz_gpu_dataflow_debug_dump.py -x -q > $O/pytest_ldlt.log 2>&1; echo "pytest ldlt exit $?: $(tail -3 $O/pytest_ldlt.log | tr '\n' ' ')"
env HIOPAMD_DF_CHECK=1 DF_REPS=100 DF_OBJECTS=2 DF_VERIFY=1 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_check.log 2>&1; echo "soak(check) exit $?: $(tail -1 $O/soak_check.log | cut -c1-200)"
env DF_REPS=1500 DF_OBJECTS=2 DF_VERIFY=1 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_verify.log 2>&1; echo "soak(verify) exit $?: $(tail -1 $O/soak_verify.log | cut -c1-200)"
for n in 4096 6144; do DF_N=$n t w8_n$n HIOPAMD_DF_PIPE=3; done
