#!/bin/bash
# Round 6, call 26: leaf width 32: sparse tests (normal + poison), bench entries
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_26
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
HIOPAMD_BUILD_VARIANT=poison timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_26/bench.json"))
print("headline %.2f it/s | dense_sharded %.3f | dense_n1e6 %.3f | sparse %.3f / banded %.3f ms" % (d["value"], d["dense_sharded"]["ms_per_step"], d["dense_n1e6_m100"]["ms_per_step"], d["sparse_condensed_n1e6"]["ms_per_step"], d["sparse_condensed_banded_n1e6"]["ms_per_step"]))
print(json.dumps(d["sparse_condensed_banded_n1e6"])[:600])
PY
exit 0
