#!/bin/bash
# Round 6, second soak: the padded orders (round 6: an order the fast forms do not take is factored and solved at a larger one) on the round's
# final library, shipped configuration (retry copy on; an expired wait is counted and stops the run), EVERY factor used for a solve with the
# residual checked: 10 000 factorisations each of N = 8191 (-> 8192), 8003 (-> 8192), 4097 (-> 4608), 2047 (-> 2048), 6503 (-> 6656), then
# 20 000 of the unpadded 8192 without the solves.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_soak2
mkdir -p $O
sha256sum hiop_amd/lib/libhiopamd.so | tee $O/summary.txt
for n in 8191 8003 4097 2047 6503; do
  env DF_N=$n DF_VERIFY=1 DF_RETRY_COPY=1 DF_REPS=2500 DF_OBJECTS=4 timeout -s KILL 1200 python scripts/df_repeat.py > $O/n$n.log 2>&1; rc=$?
  echo "N = $n (DF_VERIFY=1): exit $rc: $(tail -1 $O/n$n.log | cut -c1-200)" | tee -a $O/summary.txt
done
for i in 1 2; do
  env DF_RETRY_COPY=1 DF_REPS=2500 DF_OBJECTS=4 timeout -s KILL 600 python scripts/df_repeat.py > $O/chunk_$i.log 2>&1; rc=$?
  echo "N = 8192: exit $rc: $(tail -1 $O/chunk_$i.log | cut -c1-200)" | tee -a $O/summary.txt
done
exit 0
