#!/bin/bash
# round 3, GPU call 42: C interface tests (fixed variable in the dense interface, gradient-based scaling in both)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_42
timeout 1200 python -m pytest tests/test_c_interface.py -m gpu -q > gpurun_out/r03_42/pytest.log 2>&1; echo "pytest exit $?"; tail -60 gpurun_out/r03_42/pytest.log | cut -c1-220
