#!/bin/bash
# Round 6, call 4: (a) the new GPU tests of the round (known answers 97-137, Gram closed forms, one-workgroup pivoted panel kernel);
# (b) A/B of the wide kernel compiled for ONE wave per SIMD (no scratch spills, AGPRs as spill space, a CU cannot hold two) vs the shipped form
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_04
mkdir -p $O
timeout 900 python -m pytest tests/test_reference_known_answers.py tests/test_gram_closed_forms.py tests/test_gpu_ldlt_bk.py -m gpu -q -p no:cacheprovider > $O/pytest_new.log 2>&1
echo "pytest exit: $?"; tail -15 $O/pytest_new.log
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
STEPS=20 bash scripts/ab_bench.sh base occ1 2>&1 | tee $O/ab_occ1.txt
cp build_variants/occ1.so hiop_amd/lib/libhiopamd.so
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py -m gpu -q -x -p no:cacheprovider > $O/pytest_occ1.log 2>&1
echo "pytest(occ1) exit: $?"; tail -5 $O/pytest_occ1.log
python scripts/factor_time.py 8192 2>&1 | tail -1 | tee -a $O/ab_occ1.txt
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
python scripts/factor_time.py 8192 2>&1 | tail -1 | tee -a $O/ab_occ1.txt
