#!/bin/bash
# round 5, call 4: dataflow solve with 512 x 32 pieces (two workgroups per CU) against 512 x 64, A/B in one box
set -u
mkdir -p gpurun_out/r05_4
export TMPDIR=/tmp
echo "=== pytest ldlt ==="
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_full_size.py -q -x > gpurun_out/r05_4/pytest.log 2>&1; echo "pytest exit: $?"
tail -4 gpurun_out/r05_4/pytest.log
for v in 32 64 32 64; do
  if [ $v = 64 ]; then export HIOPAMD_DEV_SOLVE_CW64=1; else unset HIOPAMD_DEV_SOLVE_CW64; fi
  echo "--- CW=$v"
  DF_TIMELINE=0 timeout 200 python scripts/df_stamps.py 2>&1 | grep "solves best\|matrixChanged best"
done
unset HIOPAMD_DEV_SOLVE_CW64
echo "=== bench (CW=32) ==="
timeout 600 python bench.py --steps 20 --warmup 5 --no-dense --no-cpu-baseline > gpurun_out/r05_4/bench.json 2> gpurun_out/r05_4/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_4/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["kkt_spans"].items():
    if isinstance(v,dict): print(" ",k,round(v["ms_per_step"],4))
PY
