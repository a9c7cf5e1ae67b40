#!/bin/bash
# Round 6, call 12: the 147 known answers on HIP, then this round's PMC passes (one-dispatch factorisation; dense kernels)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_12
timeout 600 python -m pytest tests/test_reference_known_answers.py -m gpu -q -p no:cacheprovider > gpurun_out/r06_12/pytest_known.log 2>&1
echo "pytest exit: $?"; tail -3 gpurun_out/r06_12/pytest_known.log
bash scripts/calls/r06_pmc.sh 2>&1 | tail -60
