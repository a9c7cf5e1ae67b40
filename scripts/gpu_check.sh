#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, rocprofv3 kernel trace of the bench.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
echo "=== pytest -m gpu ===" 
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
echo "=== smoke ==="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" | tee -a gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
echo "=== bench ==="
timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ "${DO_PROF:-1}" = "1" ]; then
  echo "=== rocprofv3 kernel-trace ==="
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-dense > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err); echo "rocprof exit: $?"
  find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
fi
