#!/bin/bash
# round 3, GPU call 34: phases of the dense low-rank step
set -u
export TMPDIR=/tmp
timeout 300 python scripts/dense_step_phases.py 2>&1 | tail -10
DENSE_N=1250000 DENSE_K=200 timeout 300 python scripts/dense_step_phases.py 2>&1 | tail -10
