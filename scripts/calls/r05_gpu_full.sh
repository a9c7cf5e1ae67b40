#!/bin/bash
# Round 5: the full pass — parity suite, smoke, bench (JSON line kept), rocprofv3 kernel trace + stats of the same bench command,
# PMC passes (LDL^T one-dispatch form, dense kernels), pivoted-mode timing.  What is to be judged goes to gpurun_out/r05_full (-> profiles/).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_full
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != 1 ]; then
echo "=== pytest -m gpu ==="
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?"; grep -h "passed\|failed\|error" $O/pytest_gpu.log | tail -3
echo "=== smoke ==="
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
echo "=== bench ==="
timeout -s KILL 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_full/bench.json").read().strip().splitlines()[-1])
s = d["kkt_spans"]
print("value %.2f it/s  %.3f ms/step  roofline %.1f TF (%.3f)  wide kernel %.3f ms | fact %.3f  solves %.3f  assembly %.3f  rhs %.3f | dense_sharded %.2f ms  dense_c2 %.2f ms  sparse %.2f ms  banded %.2f ms  cpu %.3f it/s" % (
    d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], s["linsolv.tmFactTime"]["ms_per_step"],
    s["linsolv.tmTriuSolves"]["ms_per_step"], s["kkt.tmUpdateLinsys"]["ms_per_step"], s["kkt.tmSolveRhsManip"]["ms_per_step"],
    d["dense_sharded"]["ms_per_step"], d["dense_n1e6_m100"]["ms_per_step"], d["sparse_condensed_n1e6"]["ms_per_step"],
    d["sparse_condensed_banded_n1e6"]["ms_per_step"], d["cpu_baseline"]["value"]))
e = d.get("ipm_end_to_end_N8192") or {}
print("ipm_end_to_end:", {k: e.get(k) for k in ("value", "iterations", "solve_seconds")})
PY
echo "=== rocprofv3 kernel-trace of the same command ==="
(cd /tmp && timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2> $R/$O/prof.err); echo "rocprof exit: $?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && cut -c1-150 "$f" | head -8
rm -rf $O/prof
if [ "${SKIP_PMC:-0}" != 1 ]; then
bash scripts/calls/r05_pmc.sh 2>&1 | tail -60
echo "=== pivoted factorisation: timing ==="
timeout -s KILL 300 python scripts/bk_time.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee $O/bk_time.log
fi
