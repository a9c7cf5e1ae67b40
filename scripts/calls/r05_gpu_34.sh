#!/bin/bash
# Round 5, call 34: rocprofv3 kernel statistics of the pivoted factorisation + solves (scripts/bk_time.py 8192)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_34
mkdir -p $O
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bk -- python $R/scripts/bk_time.py 8192 > $R/$O/bk_under_rocprof.txt 2> $R/$O/prof.err); echo "rocprof exit: $?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bk_kernel_stats.csv && cut -c1-160 "$f" | head -14
rm -rf $O/prof
grep -v amdgpu $O/bk_under_rocprof.txt | tail -2
