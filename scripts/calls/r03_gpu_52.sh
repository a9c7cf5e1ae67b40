#!/bin/bash
# round 3, GPU call 52: the hand-out check as a test
set -u
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ldlt_timeout_recovery.py -m gpu -q 2>&1 | tail -5
