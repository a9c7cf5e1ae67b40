#!/bin/bash
# Round 6, call 28: the default bench run with the new `mds_other_orders` entry (wall time of the whole command, the JSON line)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_28
mkdir -p $O
s=$(date +%s)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit: $?"
e=$(date +%s); echo "bench wall time: $((e - s)) s" | tee $O/wall.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_28/bench.json"))
print("headline %.2f it/s %.3f ms | frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print(json.dumps(d.get("mds_other_orders")))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
PY
exit 0
