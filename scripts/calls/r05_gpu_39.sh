#!/bin/bash
# Round 5, call 39: after the event-wait change: 20 000 back-to-back factorisations (the next one is now enqueued while the previous one's
# epilogue kernels may still run) and the bench line of the final library
set -u
export TMPDIR=/tmp
SOAK_CHUNKS=2 bash scripts/calls/r05_soak.sh
bash scripts/calls/r05_gpu_36.sh
