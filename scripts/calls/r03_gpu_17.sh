#!/bin/bash
# round 3, GPU call 17: bench with every entry (incl. sparse condensed and the end-to-end IPM run through the C interface)
set -u
mkdir -p gpurun_out/r03_17
export TMPDIR=/tmp
O=gpurun_out/r03_17
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_17/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("ipm_end_to_end_N8192"), indent=1)[:1500])
print({k: v for k, v in d.get("sparse_condensed_n1e6", {}).items() if k in ("value", "ms_per_step", "pcg_iterations_per_solve", "error")})
PY
tail -3 $O/bench.err
