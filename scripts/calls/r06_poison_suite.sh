#!/bin/bash
# Round 6: the WHOLE GPU suite (every test but the gate's own poison wrapper) on the poison build of the round's final sources
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_poison_suite
mkdir -p $O
sha256sum hiop_amd/lib_poison/libhiopamd.so | tee $O/summary.txt
HIOPAMD_BUILD_VARIANT=poison timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_poisoned_allocations.py > $O/log.txt 2>&1; grep -E "passed|failed|error" $O/log.txt | tail -3 | tee -a $O/summary.txt
exit 0
