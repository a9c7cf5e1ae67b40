#!/bin/bash
# round 3, GPU call 12: whole GPU suite after the LDL^T schedule / GEMV / Gram / C-interface changes + bench
set -u
mkdir -p gpurun_out/r03_12
export TMPDIR=/tmp
O=gpurun_out/r03_12
timeout 1700 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -14 $O/pytest.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_12/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic/alg", d["roofline"].get("traffic_over_algorithmic"))
for k,v in d["kkt_spans"].items():
    if isinstance(v,dict): print(" ", k, round(v["ms_per_step"],4))
for key in ("dense_sharded","dense_n1e6_m100"):
    e=d[key]; print(key, round(e["ms_per_step"],3), [ (r["kernel"][:12], round(r["avg_launch_ms"],3), round(r["frac"],3)) for r in e["roofline"]])
print("cpu", d.get("cpu_baseline"))
PY
tail -3 $O/bench.err
