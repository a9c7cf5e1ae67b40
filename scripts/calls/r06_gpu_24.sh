#!/bin/bash
# Round 6, call 24: padded orders chosen by the cost model (HIOPAMD_LDLT_PAD=2, default) against parity only (=1): timing over many orders,
# then the LDL^T / KKT / C-interface / poison tests
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_24
mkdir -p $O
for N in 1025 1300 1537 2000 2049 2500 3000 4000 4097 5000 6000 7000 8000 8193 8500 9000 10000 12289; do
  for pad in 1 2; do
    echo "== N $N pad $pad" | tee -a $O/factor_time.txt
    HIOPAMD_LDLT_PAD=$pad timeout 300 python scripts/factor_time.py $N 2>&1 | tail -2 | tee -a $O/factor_time.txt
  done
done
timeout 1800 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_ldlt_exact_closed_form.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_ldlt_bk.py tests/test_c_interface.py tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py tests/test_gpu_lowrank.py tests/test_gpu_example_mds.py tests/test_gpu_ipm_device.py tests/test_gpu_poisoned_allocations.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest.txt
exit 0
