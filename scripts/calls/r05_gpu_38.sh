#!/bin/bash
# Round 5, call 38: the host waits for the factorisation's info words (an event) instead of for the whole stream, deferred reductions are
# flushed on an event behind the last of them: whole GPU suite on the new form, then A/B of the bench line
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_38
mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?"; grep -h "passed\|failed\|error" $O/pytest_gpu.log | tail -3
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
STEPS=20 bash scripts/ab_bench.sh base evsync 2>&1 | tee $O/ab.txt
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
