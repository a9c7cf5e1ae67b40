#!/bin/bash
# Round 6, call 17: sparse LDL^T sweeps with every independent load requested up front + readlane broadcasts; fused narrow levels on / off
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_17
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_sparse.txt
HIOPAMD_SL_GROUP_DEPTH=1 timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $O/pytest_sparse.txt
for cfg in "1 0 2048" "1 1 2048" "4 0 2048" "4 0 512" "4 0 8192" "3 0 2048" "4 1 2048" "1 0 2048" "4 0 2048"; do
  set -- $cfg
  echo "== depth $1 regs $2 fusemax $3" | tee -a $O/sparse_time.txt
  HIOPAMD_SL_GROUP_DEPTH=$1 HIOPAMD_SL_REGS=$2 HIOPAMD_SL_FUSE_MAX=$3 timeout 300 python scripts/sparse_ldl_time.py 1000000 5 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  HIOPAMD_SL_GROUP_DEPTH=$1 HIOPAMD_SL_REGS=$2 HIOPAMD_SL_FUSE_MAX=$3 timeout 300 python scripts/sparse_ldl_time.py 200000 20 2>&1 | tail -1 | tee -a $O/sparse_time.txt
done
for cfg in "1 0" "4 0" "4 1"; do
  set -- $cfg
  (cd /tmp && HIOPAMD_SL_GROUP_DEPTH=$1 HIOPAMD_SL_REGS=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -o t -- python $GRAFT_REPO_ROOT/scripts/sparse_ldl_time.py 1000000 5 > /dev/null 2>&1)
  f=$(find /tmp/prof_$1_$2 -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $O/trace_depth$1_regs$2.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation + its three solves
last = max(i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%9.1f  %8.1f us  grid %8s wg %5s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), r["Kernel_Name"][:64]))
PY
done
exit 0
