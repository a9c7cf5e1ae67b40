#!/bin/bash
# round 3, GPU call 15: sparse condensed KKT after the long-row SpMV; kernel trace of its bench entry
set -u
mkdir -p gpurun_out/r03_15
export TMPDIR=/tmp
O=gpurun_out/r03_15
timeout 900 python -m pytest tests/test_gpu_kkt_sparse.py tests/test_gpu_csr_condensed.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --ns 2000 --nd 256 --neq 253 > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err; echo "bench exit $?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_15/bench.json"))
print(json.dumps({k: v for k, v in d.get("sparse_condensed_n1e6", {}).items() if k not in ("note", "workload")}, indent=1))
PY
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
find $O/prof -type f ! -name "*stats*.csv" -delete
