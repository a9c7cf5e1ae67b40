#!/bin/bash
# Round 4, call 17: the four-wave wide kernel with the task selected ahead and its selection state in LDS (HIOPAMD_DF_FORM=5) against
# the shipped form (4) and round 3's two-per-CU shape, inside one box; per-task counters (HIOPAMD_DF_CHECK) and a verified soak
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_17
mkdir -p $O
t() { name=$1; shift; env "$@" DF_TIMELINE=${TL:-0} timeout -s KILL 150 python scripts/df_stamps.py > $O/$name.log 2>&1; echo "$name: exit $? | $(grep -h 'matrixChanged' $O/$name.log | tr '\n' ' ')"; }
t form4 HIOPAMD_DF_FORM=4
t form5 HIOPAMD_DF_FORM=5
t form5_pipe0 HIOPAMD_DF_FORM=5 HIOPAMD_DF_PIPE=0
t form5_all HIOPAMD_DF_FORM=5 HIOPAMD_DF_PIPEJ=31
t form5_lead12 HIOPAMD_DF_FORM=5 HIOPAMD_DF_SELLEAD=12
t form5_lead6 HIOPAMD_DF_FORM=5 HIOPAMD_DF_SELLEAD=6
t form4_480 HIOPAMD_DF_FORM=4 HIOPAMD_DF_WGS=480
t form4_again HIOPAMD_DF_FORM=4
t form5_again HIOPAMD_DF_FORM=5
TL=1 t form5_stamps HIOPAMD_DF_FORM=5
TL=1 t form4_stamps HIOPAMD_DF_FORM=4
grep -h "wide kernel phases" $O/form5_stamps.log $O/form4_stamps.log | cut -c1-330
env HIOPAMD_DF_FORM=5 timeout -s KILL 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest(form 5) exit $?: $(tail -2 $O/pytest.log | tr '\n' ' ')"
env HIOPAMD_DF_FORM=5 HIOPAMD_DF_CHECK=1 DF_REPS=150 DF_OBJECTS=2 DF_VERIFY=1 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_check.log 2>&1; echo "soak(check, form 5) exit $?: $(tail -1 $O/soak_check.log | cut -c1-200)"
env HIOPAMD_DF_FORM=5 DF_REPS=1500 DF_OBJECTS=2 DF_VERIFY=1 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_verify.log 2>&1; echo "soak(verify, form 5) exit $?: $(tail -1 $O/soak_verify.log | cut -c1-200)"
