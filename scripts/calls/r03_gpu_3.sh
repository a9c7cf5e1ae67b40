#!/bin/bash
# round 3, GPU call 3: whole GPU suite (new: reference-trajectory IPM runs on the device for MDS and the dense quasi-Newton
# examples, randomized / dual-first regularisation), short bench.
set -u
mkdir -p gpurun_out/r03_3
export TMPDIR=/tmp
O=gpurun_out/r03_3
echo "=== pytest -m gpu (all) ==="
timeout 1700 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -40 $O/pytest.log
echo "=== bench ==="
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_3/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
tail -3 $O/bench.err
