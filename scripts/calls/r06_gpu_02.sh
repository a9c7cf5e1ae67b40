#!/bin/bash
# Round 6, call 2: after the fix of the block inversion's padded reads — the whole GPU suite on the POISON build (no -x: every test
# that depends on memory nobody wrote shows), the fresh-process soak on it and on the shipped build.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_02
mkdir -p $O
echo "=== poison build: whole GPU suite (all failures) ==="
HIOPAMD_BUILD_VARIANT=poison timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_zz_gpu_dataflow_debug_dump.py -p no:cacheprovider > $O/pytest_poison.log 2>&1
echo "pytest(poison) exit: $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_poison.log | head -80
echo "=== poison build: soak ==="
HIOPAMD_BUILD_VARIANT=poison scripts/cold_start_soak.sh 20 $O/soak_poison.txt
echo "=== shipped build: soak ==="
scripts/cold_start_soak.sh 20 $O/soak_shipped.txt
