#!/bin/bash
# round 5, call 2: the 16 x 16 sub-block factor with four pivots per pass (factor16r): LDL^T tests, time line, short bench
set -u
mkdir -p gpurun_out/r05_2
export TMPDIR=/tmp
echo "=== pytest ldlt ==="
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_ldlt_bk.py tests/test_gpu_full_size.py tests/test_gpu_kkt_xycyd.py -q -x > gpurun_out/r05_2/pytest.log 2>&1; echo "pytest exit: $?"
tail -8 gpurun_out/r05_2/pytest.log
echo "=== stamps ==="
timeout 200 python scripts/df_stamps.py > gpurun_out/r05_2/stamps.log 2>&1; echo "stamps exit: $?"
grep -v "^  [ 0-9][0-9] |" gpurun_out/r05_2/stamps.log | tail -16
grep "^  [ 0-9][0-9] |" gpurun_out/r05_2/stamps.log | awk 'NR%4==1' | head -12
echo "=== bench ==="
timeout 600 python bench.py --steps 20 --warmup 5 --no-dense --no-cpu-baseline > gpurun_out/r05_2/bench.json 2> gpurun_out/r05_2/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_2/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["kkt_spans"].items():
    if isinstance(v,dict): print(" ",k,round(v["ms_per_step"],4))
PY
tail -3 gpurun_out/r05_2/bench.err
