#!/bin/bash
# kernel trace of the bench, csv summaries only (the box's rocprofv3 writes a rocpd database unless told otherwise)
set -u
mkdir -p gpurun_out/r03_4
export TMPDIR=/tmp
O=gpurun_out/r03_4
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err; echo "prof exit $?"
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f"
find $O/prof -type f ! -name "*stats*.csv" -delete
