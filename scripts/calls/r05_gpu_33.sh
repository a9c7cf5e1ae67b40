#!/bin/bash
# Round 5, call 33: sparse LDL^T solve sweeps with the front's L panel staged in LDS: parity (sparse LDL^T + condensed sparse KKT) and the two sparse bench entries
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_33
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py -x -q 2>&1 | grep "passed\|failed\|error" | tail -2 | tee $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense --ns 2000 --nd 256 --neq 253 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('sparse_condensed_n1e6','sparse_condensed_banded_n1e6'):
    print(k, '%.3f ms/step' % d[k]['ms_per_step'])" | tee $O/bench.txt
