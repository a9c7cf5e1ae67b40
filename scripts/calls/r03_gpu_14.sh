#!/bin/bash
# round 3, GPU call 14: dense C interface (quasi-Newton), MDS C interface again, bench with the sparse-condensed entry
set -u
mkdir -p gpurun_out/r03_14
export TMPDIR=/tmp
O=gpurun_out/r03_14
timeout 900 python -m pytest tests/test_c_interface.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -12 $O/pytest.log
gcc -std=c11 -O1 -Iinclude tests/c/dense_c_interface.c -o /tmp/dense_c -Lhiop_amd/lib -lhiopamd -lm -Wl,-rpath,$PWD/hiop_amd/lib
( time timeout 300 /tmp/dense_c 500 ) 2>&1 | tail -12
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_14/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("sparse_condensed_n1e6"), indent=1)[:900])
PY
tail -3 $O/bench.err
