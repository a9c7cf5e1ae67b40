#!/bin/bash
# Round 4, call 12: kernel time line of the dense low-rank step (rocprofv3 --kernel-trace of scripts/dense_step_time.py) -> where the
# step's wall time goes that is not kernel time
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_12; mkdir -p $O
STEPS=15 timeout -s KILL 300 python scripts/dense_step_time.py 2>&1 | grep -v amdgpu.ids | tee $O/dense.log
(cd /tmp && STEPS=6 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o dense -- python $R/scripts/dense_step_time.py > $R/$O/dense_prof.log 2>&1); echo "rocprof exit $?"
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $O/dense_kernel_trace.csv; rm -rf $O/prof; ls -la $O
