#!/bin/bash
# round 5, call 7: HIP API + kernel trace of a few steps: where the host stands while the device waits in front of the chain kernel
set -u
mkdir -p gpurun_out/r05_7
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/proft; mkdir -p $R/gpurun_out/proft
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace -d $R/gpurun_out/proft -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense > /dev/null 2> $R/gpurun_out/proft/err.txt)
DB=$(find $R/gpurun_out/proft -name "*.db" | head -1)
python - "$DB" > gpurun_out/r05_7/api_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
print("tables:", [t for t in tabs if 'rocpd' not in t][:40])
k = db.execute("select start, end, name from kernels order by start").fetchall()
marks=[i for i,r in enumerate(k) if "ldlt_chain_kernel" in r[2]]
c = k[marks[-1]]
t0 = c[0]
print("chain kernel start (ref 0), dur us", (c[1]-c[0])/1e3)
lo, hi = t0 - 400000, t0 + 80000
for r in k:
    if lo <= r[0] <= hi: print("K %9.1f %9.1f %s" % ((r[0]-t0)/1e3, (r[1]-t0)/1e3, r[2].split('(')[0][-45:]))
try:
    cols=[r[1] for r in db.execute("pragma table_info(regions)").fetchall()]
    print("regions cols", cols)
    rows = db.execute("select start, end, name from regions where start between ? and ? order by start", (lo, hi)).fetchall()
    for r in rows: print("A %9.1f %9.1f %s" % ((r[0]-t0)/1e3, (r[1]-t0)/1e3, r[2]))
except Exception as e:
    print("regions query failed", e)
PY
head -150 gpurun_out/r05_7/api_timeline.txt
rm -f "$DB"
