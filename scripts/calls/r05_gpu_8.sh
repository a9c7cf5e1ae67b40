#!/bin/bash
# round 5, call 8: deferred look at the solves' error word (no synchronisation in front of the factorisation)
# multi-workgroup inertia kernel, bench.py calling the C ABI with resolved addresses: tests, time line, bench
set -u
mkdir -p gpurun_out/r05_8
export TMPDIR=/tmp
echo "=== pytest ==="
timeout 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_vector.py tests/test_gpu_full_size.py tests/test_gpu_kkt_sparse.py tests/test_gpu_example_mds.py tests/test_zz_gpu_dataflow_debug_dump.py -q -x > gpurun_out/r05_8/pytest.log 2>&1; echo "pytest exit: $?"
tail -6 gpurun_out/r05_8/pytest.log
echo "=== bench ==="
timeout 600 python bench.py --steps 20 --warmup 5 --no-dense --no-cpu-baseline > gpurun_out/r05_8/bench.json 2> gpurun_out/r05_8/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_8/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["kkt_spans"].items():
    if isinstance(v,dict): print(" ",k,round(v["ms_per_step"],4))
PY
tail -3 gpurun_out/r05_8/bench.err
bash scripts/gpu_trace_factor.sh > gpurun_out/r05_8/trace.txt 2>&1
cp gpurun_out/timeline.txt gpurun_out/r05_8/timeline.txt
cat gpurun_out/r05_8/timeline.txt
