#!/bin/bash
# Round 6, call 20: odd orders factored as the even order n + 1 (16-byte tile form) — LDL^T tests, then timing with / without
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_20
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_ldlt_exact_closed_form.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_ldlt_bk.py tests/test_c_interface.py tests/test_gpu_poisoned_allocations.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest.txt
for N in 8191 8192 4097 4096 2047 2049 1025 12289; do
  for pad in 1 0; do
    echo "== N $N pad $pad" | tee -a $O/factor_time.txt
    HIOPAMD_LDLT_PAD=$pad timeout 300 python scripts/factor_time.py $N 2>&1 | tail -2 | tee -a $O/factor_time.txt
  done
done
exit 0
