#!/bin/bash
# Round 4, GPU call 8: the shipped form (four waves, one workgroup per CU, selection ahead) against the kernel of rounds 2-3 with one /
# two workgroups per CU and the eight-wave form; three reserved CUs per XCD for the chain kernel (24 CUs for its 16 workgroups);
# fused pre / post solve kernels of the MDS KKT; LDL^T tests; a verified soak.
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_8
mkdir -p $O
t() { name=$1; shift; env "$@" DF_TIMELINE=${TL:-0} timeout -s KILL 150 python scripts/df_stamps.py > $O/$name.log 2>&1; echo "$name: exit $? | $(grep -h 'matrixChanged' $O/$name.log | tr '\n' ' ')"; }
t form0_pipe1 HIOPAMD_DF_PIPE=1
t form0_pipe0 HIOPAMD_DF_PIPE=0
t form0_pipe1_sd3 HIOPAMD_DF_PIPE=1 HIOPAMD_SD_CUS=3
t form4_240 HIOPAMD_DF_FORM=4
t form4_480 HIOPAMD_DF_FORM=4 HIOPAMD_DF_WGS=480
t form8 HIOPAMD_DF_FORM=8 HIOPAMD_DF_PIPE=3
t form0_pipe1_lead12 HIOPAMD_DF_PIPE=1 HIOPAMD_DF_SELLEAD=12
t form0_pipe1_all HIOPAMD_DF_PIPE=1 HIOPAMD_DF_PIPEJ=31
TL=1 t form0_stamps HIOPAMD_DF_PIPE=1
grep -h "wide kernel phases\|shader clock" $O/form0_stamps.log | cut -c1-330
timeout -s KILL 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_zz_gpu_dataflow_debug_dump.py tests/test_gpu_vector.py tests/test_gpu_dense_sparse.py -x -q > $O/pytest.log 2>&1; echo "pytest exit $?: $(tail -3 $O/pytest.log | tr '\n' ' ')"
env HIOPAMD_DF_CHECK=1 DF_REPS=150 DF_OBJECTS=2 DF_VERIFY=1 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_check.log 2>&1; echo "soak(check) exit $?: $(tail -1 $O/soak_check.log | cut -c1-200)"
env DF_REPS=1500 DF_OBJECTS=2 DF_VERIFY=1 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_verify.log 2>&1; echo "soak(verify) exit $?: $(tail -1 $O/soak_verify.log | cut -c1-200)"
for f in 1 0; do env HIOPAMD_MDS_FUSED=$f timeout -s KILL 300 python bench.py --steps 20 --warmup 5 > $O/bench_fused$f.json 2> $O/bench_fused$f.err; python - <<PY
import json
d=json.loads(open('$O/bench_fused$f.json').read().strip().splitlines()[-1])
s=d['kkt_spans']
print('fused=$f', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms | rhs manip', round(s['kkt.tmSolveRhsManip']['ms_per_step'],3), '| solves', round(s['linsolv.tmTriuSolves']['ms_per_step'],3), '| fact', round(s['linsolv.tmFactTime']['ms_per_step'],3), '| frac', round(d['roofline']['frac'],3))
PY
done
