#!/bin/bash
# round 3, GPU call 54: sanity of the last instrumentation change (forced time-out with the dump; hand-out check; one full-size parity test)
set -u
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_ldlt_timeout_recovery.py -m gpu -q 2>&1 | tail -3
HIOPAMD_DF_TIMEOUT_MS=0.001 HIOPAMD_DF_DEBUG=1 DF_N=4096 DF_REPS=1 DF_OBJECTS=1 timeout 60 python scripts/df_repeat.py 2>&1 | grep "another CU\|never started" | head -4
env DF_TIMELINE=0 timeout -s KILL 60 python scripts/df_stamps.py 2>&1 | tail -1
