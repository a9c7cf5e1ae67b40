#!/bin/bash
# Round 6, call 16: sparse LDL^T groups with a workgroup-scope hand-over; where a leaf front's time goes (experiment switches); kernel trace
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_sparse.txt
for cfg in "1 0 0" "1 1 0" "6 1 0" "6 0 0" "4 1 0" "1 1 1" "1 1 2" "1 1 4" "1 1 8" "1 1 15" "6 1 15"; do
  set -- $cfg
  echo "== depth $1 regs $2 exp $3" | tee -a $O/sparse_time.txt
  HIOPAMD_SL_GROUP_DEPTH=$1 HIOPAMD_SL_REGS=$2 HIOPAMD_SL_EXP=$3 timeout 300 python scripts/sparse_ldl_time.py 1000000 5 2>&1 | tail -1 | tee -a $O/sparse_time.txt
done
for cfg in "1 0" "1 1" "6 1"; do
  set -- $cfg
  (cd /tmp && HIOPAMD_SL_GROUP_DEPTH=$1 HIOPAMD_SL_REGS=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -o t -- python $GRAFT_REPO_ROOT/scripts/sparse_ldl_time.py 1000000 5 > /dev/null 2>&1)
  f=$(find /tmp/prof_$1_$2 -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $O/trace_depth$1_regs$2.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation + solve sequence: print the last 120 kernels' names and durations
for r in rows[-120:]:
    print("%8.1f us  grid %8s wg %5s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), r["Kernel_Name"][:70]))
PY
done
exit 0
