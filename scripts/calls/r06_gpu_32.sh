#!/bin/bash
# Round 6, call 32: one more pass of the whole GPU suite (with the tests added after the final validation; the library is unchanged),
# then the headline step at further orders
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_32
mkdir -p $O
sha256sum hiop_amd/lib/libhiopamd.so | tee $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/log.txt 2>&1; grep -E "passed|failed" $O/log.txt | tail -2 | tee -a $O/summary.txt
timeout 600 python scripts/other_orders.py 4097,4093 5000,4200 2500,2497 6000,6286 3000,2117 7000,6000 2>&1 | grep "^N " | tee $O/other_orders.txt
for pad in 0; do
  HIOPAMD_LDLT_PAD=$pad timeout 600 python scripts/other_orders.py 2500,2497 6000,6286 3000,2117 7000,6000 2>&1 | grep "^N " | sed "s/^/PAD=$pad /" | tee -a $O/other_orders.txt
done
exit 0
