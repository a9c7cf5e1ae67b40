#!/bin/bash
# Round 6, final validation of a binary: library hash, the whole GPU suite (incl. the poison-build runs) $PASSES times in fresh processes,
# smoke, the fresh-process soak (>= 1000 solves), bench + kernel trace.   PASSES (default 3), RUNS (soak runs per case, default 130)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_final
mkdir -p $O
sha256sum hiop_amd/lib/libhiopamd.so hiop_amd/lib_poison/libhiopamd.so | tee $O/libraries.txt
rocm-smi --showproductname 2>/dev/null | head -8 > $O/device.txt
: > $O/full_pytest_gpu.txt
for p in $(seq 1 ${PASSES:-3}); do
  echo "=== whole GPU suite, pass $p (pytest -m gpu -x -q, fresh process) ===" | tee -a $O/full_pytest_gpu.txt
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 | tee -a $O/full_pytest_gpu.txt
done
echo "=== smoke ===" | tee -a $O/full_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|error" | tee -a $O/full_pytest_gpu.txt
echo "=== cold-start soak ==="
scripts/cold_start_soak.sh ${RUNS:-130} $O/cold_start_soak.txt "host:400:100 device:400:100 host:401:100 host:40:12 device:40:12 host:1000:1044 device:1000:1045"
scripts/cold_start_soak.sh 40 $O/cold_start_soak.txt "host:4092:4096 device:4092:4096"
HIOPAMD_BUILD_VARIANT=poison scripts/cold_start_soak.sh 20 $O/cold_start_soak_poison.txt "host:400:100 device:400:100 host:401:100 host:1000:1044 host:4092:4096"
echo "=== bench ==="
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_final/bench.json"))
print("headline %.2f it/s %.3f ms | frac %.3f | fact %.3f ms | dense_sharded %.3f ms | dense_n1e6 %.3f ms | sparse %.3f / banded %.3f ms" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kkt_spans"]["linsolv.tmFactTime"]["ms_per_step"], d["dense_sharded"]["ms_per_step"],
    d["dense_n1e6_m100"]["ms_per_step"], d["sparse_condensed_n1e6"]["ms_per_step"], d["sparse_condensed_banded_n1e6"]["ms_per_step"]))
PY
echo "=== rocprofv3 kernel trace of the bench ==="
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err); echo "rocprof exit: $?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
exit 0
