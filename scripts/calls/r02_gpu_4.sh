#!/bin/bash
# tile form 2 validation: LDL^T parity tests first (hard limits), the A/B timing of the two tile forms, then the full pass
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== ldlt parity (form 2 default) ==="
timeout -s KILL 300 python -m pytest tests/test_gpu_ldlt_kkt.py -x -q 2>&1 | tail -4
echo "=== A/B ==="
for form in 1 2; do
  HIOPAMD_DF_TILE=$form timeout -s KILL 120 python -u scripts/df_stamps.py > gpurun_out/stamps_form$form.txt 2>&1
  echo "form $form: $(grep matrixChanged gpurun_out/stamps_form$form.txt)"
done
DO_PROF=${DO_PROF:-1} bash scripts/calls/r02_gpu_full.sh
