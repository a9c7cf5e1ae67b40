#!/bin/bash
# round 3, GPU call 35: probe — contended read-modify-writes on flag words under the dataflow kernels' access pattern
set -u
export TMPDIR=/tmp
timeout 200 scripts/probes/atomic_soak_probe 3000 64 4 2>&1 | tail -12
timeout 200 scripts/probes/atomic_soak_probe 1000 256 1 2>&1 | tail -12
