#!/bin/bash
# round 3, GPU call 44: slab operations (adjust_bounds) and the C interface again
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_44
timeout 1200 python -m pytest tests/test_gpu_ipm_slab.py tests/test_c_interface.py tests/test_gpu_ipm_device.py -m gpu -q > gpurun_out/r03_44/pytest.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/r03_44/pytest.log | cut -c1-220
