#!/bin/bash
# round 3, GPU call 18/19: SparseEx2 selfcheck through the device (dense XDYcYd class with inertia correction; sparse condensed class with
# the direct inner solver) + the sparse KKT tests
set -u
mkdir -p gpurun_out/r03_18
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ipm_device.py tests/test_gpu_kkt_sparse.py -m gpu -q -k "sparse" > gpurun_out/r03_18/pytest.log 2>&1; echo "pytest exit $?"; tail -40 gpurun_out/r03_18/pytest.log
