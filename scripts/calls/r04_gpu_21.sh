#!/bin/bash
# Round 4, call 21: kernel statistics of the end-to-end IPM solve (hiop_mds_solve_problem on MdsEx1, N = 8192, device callbacks)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_21; mkdir -p $O
cat > /tmp/e2e.py <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import bench
d = bench.ipm_end_to_end_bench()
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["device"].items()}, flush=True)
PY
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o e2e -- python /tmp/e2e.py 2>&1 | grep "objective")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/e2e_kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04_21/e2e_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print("total kernel time %.1f ms" % tot)
for r in rows[:32]:
    print("%-100s %5s calls %8.1f us avg %7.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
