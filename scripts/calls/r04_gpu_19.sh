#!/bin/bash
# Round 4, call 19: y = Jd x of the sparse condensed KKT through the CSR kernels (a thread per short row) instead of a wave per row:
# parity of the sparse paths, then the bench entry
set -u
O=gpurun_out/r04_19; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_kkt_sparse.py tests/test_gpu_csr_condensed.py tests/test_gpu_ipm_device.py tests/test_gpu_kkt_xycyd.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert|^E " $O/pytest.log | head -20
timeout -s KILL 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/sparse.log
import argparse, bench
from hiop_amd.runtime import Context
ctx = Context(0)
a = argparse.Namespace(steps=20, warmup=3, solves=3)
for _ in range(2):
    d = bench.sparse_condensed_bench(ctx, a)
    print("sparse condensed n=1e6: %.3f ms per KKT iteration (%s)" % (d["ms_per_step"], d["inner_solver"]), flush=True)
PY
