#!/bin/bash
# Round 6, call 14: the factorisation's solve-only epilogue on the side stream (overlapping the read-back, the host's wake-up and the next
# solve's preparation kernels): tests of everything that factors and solves, then A/B of the bench line against the previous library
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_14
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
timeout 1200 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_ldlt_exact_closed_form.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_kkt_xycyd.py tests/test_gpu_sparse_ldl.py tests/test_gpu_lowrank.py tests/test_c_interface.py tests/test_gpu_ipm_device.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed" $O/pytest.log | tail -2
STEPS=20 bash scripts/ab_bench.sh base2 epi epi_async 2>&1 | tee $O/ab_epilogue_side_stream.txt
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
