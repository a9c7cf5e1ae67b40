#!/bin/bash
# round 5, call 1: the clean-up commits on hardware (full GPU suite), hipStreamWaitValue32 probe, time line of the factorisation, bench
set -u
mkdir -p gpurun_out/r05_1
export TMPDIR=/tmp
echo "=== stream wait value probe ==="
true
true
echo "=== pytest -m gpu ==="
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r05_1/pytest_gpu.log 2>&1; echo "pytest exit: $?"
tail -15 gpurun_out/r05_1/pytest_gpu.log
echo "=== stamps ==="
timeout 200 python scripts/df_stamps.py > gpurun_out/r05_1/stamps.log 2>&1; echo "stamps exit: $?"
grep -v "^  [ 0-9][0-9] |" gpurun_out/r05_1/stamps.log | tail -20
echo "=== bench ==="
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_1/bench.json 2> gpurun_out/r05_1/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_1/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["kkt_spans"].items():
    if isinstance(v,dict): print(" ",k,round(v["ms_per_step"],4))
for k in ("dense_sharded","dense_n1e6_m100","sparse_condensed_n1e6","ipm_end_to_end_N8192"):
    if k in d: print(k, d[k].get("ms_per_step"), d[k].get("value"))
PY
tail -3 gpurun_out/r05_1/bench.err
