#!/bin/bash
# Round 6, call 18: sparse LDL^T — the register-resident factor kernel on the levels with many pivots only, readlane broadcasts in the sweeps
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_18
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_sparse.txt
HIOPAMD_BUILD_VARIANT=poison timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/pytest_sparse.txt
for regs in 0 1 0 1; do
  echo "== regs $regs" | tee -a $O/sparse_time.txt
  for pat in "1000000 5" "1000000 7" "1000000 3" "200000 20" "500000 10"; do
    HIOPAMD_SL_REGS=$regs timeout 300 python scripts/sparse_ldl_time.py $pat 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  done
done
for regs in 0 1; do
  (cd /tmp && HIOPAMD_SL_REGS=$regs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$regs -o t -- python $GRAFT_REPO_ROOT/scripts/sparse_ldl_time.py 1000000 7 > /dev/null 2>&1)
  f=$(find /tmp/prof_$regs -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $O/trace_bw7_regs$regs.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%9.1f  %8.1f us  grid %8s wg %5s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), r["Kernel_Name"][:64]))
PY
done
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_18/bench.json"))
print("headline %.2f it/s | sparse %.3f / banded %.3f ms" % (d["value"], d["sparse_condensed_n1e6"]["ms_per_step"], d["sparse_condensed_banded_n1e6"]["ms_per_step"]))
PY
exit 0
