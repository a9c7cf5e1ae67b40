#!/bin/bash
# round 3, GPU call 2: the whole GPU suite (new: sparse assembly, sparse condensed KKT, full-size properties, 2 ranks at k = 200,
# robustness changes of the linear solver), factor / solve timing after the faster W512 products, bench.
set -u
mkdir -p gpurun_out/r03_2
export TMPDIR=/tmp
O=gpurun_out/r03_2
echo "=== pytest -m gpu (all) ==="
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -30 $O/pytest.log
echo "=== factor + 3 solves ==="
DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -2
HIOPAMD_SOLVE_LEAD=3 DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -2 | head -1
echo "=== bench ==="
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_2/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic/alg", d["roofline"].get("traffic_over_algorithmic"))
for k,v in d["kkt_spans"].items():
    if isinstance(v,dict): print(" ", k, round(v["ms_per_step"],4))
for key in ("dense_sharded","dense_n1e6_m100"):
    e=d[key]; print(key, round(e["ms_per_step"],3), [ (r["kernel"][:12], round(r["avg_launch_ms"],3), round(r["frac"],3)) for r in e["roofline"]])
print("cpu", d.get("cpu_baseline"))
PY
tail -3 $O/bench.err
