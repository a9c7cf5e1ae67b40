#!/bin/bash
# Round 6: soak of the dataflow LDL^T in its shipped configuration (retry copy on; a bounded wait that expires is counted and stops the run)
# on the round's final library: 5 x 10 000 factorisations at N = 8192, then the shapes of the round-5 gate failure's neighbourhood that the
# earlier soaks never ran — ODD order on the dataflow path (8191: 8-byte tile form), ragged orders 2047 and 4097 — 10 000 each, every factor of
# the ragged orders used for a solve with the residual checked (DF_VERIFY=1: the inverted ragged last block is what round 5 got wrong).
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_soak
mkdir -p $O
sha256sum hiop_amd/lib/libhiopamd.so | tee $O/summary.txt
total=0
for i in $(seq 1 ${SOAK_CHUNKS:-5}); do
  env DF_RETRY_COPY=1 DF_REPS=2500 DF_OBJECTS=4 timeout -s KILL 600 python scripts/df_repeat.py > $O/chunk_$i.log 2>&1; rc=$?
  tail -1 $O/chunk_$i.log | cut -c1-160 | tee -a $O/summary.txt
  if [ $rc -ne 0 ]; then echo "chunk $i: exit $rc" | tee -a $O/summary.txt; grep -h "time-outs absorbed\|bounded wait\|failed" $O/chunk_$i.log | head -5 | tee -a $O/summary.txt; break; fi
  total=$((total + 10000))
done
echo "soak: $total factorisations of order 8192 without a time-out" | tee -a $O/summary.txt
for n in 8191 2047 4097; do
  v=1; [ $n = 8191 ] && v=0
  env DF_N=$n DF_VERIFY=$v DF_RETRY_COPY=1 DF_REPS=2500 DF_OBJECTS=4 timeout -s KILL 900 python scripts/df_repeat.py > $O/n$n.log 2>&1; rc=$?
  echo "N = $n (DF_VERIFY=$v): exit $rc: $(tail -1 $O/n$n.log | cut -c1-160)" | tee -a $O/summary.txt
done
