#!/bin/bash
# full pass + PMC passes of the bench (dominant kernel = ldlt_wide_kernel)
set -u
bash scripts/calls/r02_gpu_full.sh
echo "=== PMC ==="
bash scripts/gpu_pmc.sh 2>&1 | tail -30
