#!/bin/bash
# round 2: full GPU pass — parity suite, smoke, bench, rocprofv3 kernel trace of the bench.  Hard per-command limits.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "=== pytest -m gpu ==="
timeout -s KILL ${PYTEST_TIMEOUT:-600} python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?"; tail -6 gpurun_out/pytest_gpu.log
echo "=== smoke ==="
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "=== bench ==="
timeout -s KILL 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("value %.2f it/s  %.3f ms/step  roofline %.1f TF (%.3f)  fact %.3f ms  dense_sharded %.2f ms  dense_c2 %.2f ms  cpu %.3f it/s" % (
    d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["kkt_spans"]["linsolv.tmFactTime"]["ms_per_step"],
    d["dense_sharded"]["ms_per_step"], d["dense_n1e6_m100"]["ms_per_step"], d["cpu_baseline"]["value"]))
PY
if [ "${DO_PROF:-1}" = "1" ]; then
  echo "=== rocprofv3 kernel-trace ==="
  (cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "rocprof exit: $?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -16
fi
