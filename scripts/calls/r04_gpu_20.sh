#!/bin/bash
# Round 4, call 20: kernel statistics of the sparse condensed KKT iteration (n = 1e6) alone
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_20; mkdir -p $O
cat > /tmp/sp.py <<'PY'
import argparse, sys
sys.path.insert(0, "/root/repo")
import bench
from hiop_amd.runtime import Context
ctx = Context(0)
a = argparse.Namespace(steps=20, warmup=3, solves=3)
d = bench.sparse_condensed_bench(ctx, a)
print("sparse condensed n=1e6: %.3f ms per KKT iteration (%s)" % (d["ms_per_step"], d["inner_solver"]), flush=True)
PY
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o sp -- python /tmp/sp.py 2>&1 | grep "sparse condensed")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/sparse_kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04_20/sparse_kernel_stats.csv")))
for r in rows[:22]:
    print("%-90s %5s calls %8.1f us avg %7.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
