#!/bin/bash
# Round 6, call 23: odd orders factored AND solved at the padded order: LDL^T / KKT / C-interface / poison tests, timing
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_23
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_ldlt_exact_closed_form.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_ldlt_bk.py tests/test_c_interface.py tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py tests/test_gpu_lowrank.py tests/test_gpu_example_mds.py tests/test_gpu_ipm_device.py tests/test_gpu_poisoned_allocations.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest.txt
for N in 8191 4097 2047 1025 12289 8193; do
  for pad in 1 0; do
    echo "== N $N pad $pad" | tee -a $O/factor_time.txt
    HIOPAMD_LDLT_PAD=$pad timeout 300 python scripts/factor_time.py $N 2>&1 | tail -2 | tee -a $O/factor_time.txt
  done
done
exit 0
