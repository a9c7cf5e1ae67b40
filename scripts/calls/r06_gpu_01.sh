#!/bin/bash
# Round 6, call 1: reproduce the round-5 gate failure.  (a) fresh-process soak of the C interface on the shipped library; (b) the whole
# GPU suite and a short soak on the POISON build (every device allocation of the library filled with NaN / -1).
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_01
mkdir -p $O
echo "=== first process on this box: the failing test's program ==="
scripts/cold_start_soak.sh 1 $O/first.txt host:400:100
echo "=== soak, shipped library ==="
scripts/cold_start_soak.sh ${RUNS:-150} $O/soak_shipped.txt
echo "=== poison build: whole GPU suite ==="
HIOPAMD_BUILD_VARIANT=poison timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_zz_gpu_dataflow_debug_dump.py > $O/pytest_poison.log 2>&1
echo "pytest(poison) exit: $?"; tail -40 $O/pytest_poison.log
echo "=== poison build: soak ==="
HIOPAMD_BUILD_VARIANT=poison scripts/cold_start_soak.sh 20 $O/soak_poison.txt
