#!/bin/bash
# rocprofv3 kernel trace of a short MDS bench run; prints per-kernel duration summaries of the factorisation chain.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/proft; mkdir -p $R/gpurun_out/proft
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/proft -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense > /dev/null 2> $R/gpurun_out/proft/err.txt)
DB=$(find $R/gpurun_out/proft -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, (end-start)/1000.0 d from kernels").fetchall()
import collections
agg = collections.defaultdict(list)
for n, d in rows:
    agg[n.split('(')[0][:60]].append(d)
tot = sum(sum(v) for v in agg.values())
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    v2 = sorted(v)
    print(f"{n:60s} n={len(v):5d} sum={sum(v)/1e3:8.3f} ms  med={v2[len(v2)//2]:8.1f} us  max={v2[-1]:8.1f}")
PY
python $R/scripts/rocpd_timeline.py "$DB" $R/gpurun_out/timeline.txt 2 || true
rm -f "$DB"
