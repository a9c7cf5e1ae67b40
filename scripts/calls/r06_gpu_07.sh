#!/bin/bash
# Round 6, call 7: kernel trace of the dense low-rank step with the fused solveCompressed (both bench shapes)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_07
mkdir -p $O
cd /tmp
for cfg in "1000000 100" "1250000 200"; do
  set -- $cfg
  DENSE_N=$1 DENSE_K=$2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_k$2 -o dense -- python $GRAFT_REPO_ROOT/scripts/dense_step_phases.py > $O/phases_k$2.txt 2>&1
  f=$(find $O/prof_k$2 -name "*kernel_trace.csv" | head -1)
  python $GRAFT_REPO_ROOT/scripts/step_timeline.py $f secant_jac 30 2 > $O/timeline_k$2.txt 2>&1
  tail -9 $O/phases_k$2.txt
  rm -rf $O/prof_k$2
done
