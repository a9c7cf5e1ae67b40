#!/bin/bash
# Round 5: soak (chain kernel with the four-pivots-per-pass 16 x 16 factor, one preparation launch) of the dataflow LDL^T in its SHIPPED configuration (four-wave wide kernel, ONE workgroup per CU of the wide stream,
# retry copy on): 10 x 10 000 factorisations at N = 8192 on fresh solver objects; every 2 500th factor is used for a solve and the
# residual checked.  A bounded wait that expires is absorbed by the retry copy and counted (hiopamd_linsolver_timeouts): the run
# stops at the first one.  The summary goes to profiles/r05_soak.txt.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_soak
mkdir -p $O
total=0
for i in $(seq 1 ${SOAK_CHUNKS:-10}); do
  env DF_RETRY_COPY=1 DF_REPS=2500 DF_OBJECTS=4 timeout -s KILL 600 python scripts/df_repeat.py > $O/chunk_$i.log 2>&1; rc=$?
  tail -1 $O/chunk_$i.log | cut -c1-160
  if [ $rc -ne 0 ]; then echo "chunk $i: exit $rc"; grep -h "time-outs absorbed\|bounded wait\|failed" $O/chunk_$i.log | head -5; break; fi
  total=$((total + 10000))
done
echo "soak: $total factorisations of order 8192 without a time-out" | tee $O/summary.txt
