#!/bin/bash
# Round 6, call 5: cold-start soak of the fixed library: >= 1000 fresh-process solves over odd / even / ragged orders, host and device
# callbacks, plus the same on the poison build
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_05
mkdir -p $O
sha256sum hiop_amd/lib/libhiopamd.so | tee $O/library.txt
scripts/cold_start_soak.sh ${RUNS:-160} $O/cold_start_soak.txt "host:400:100 device:400:100 host:401:100 host:40:12 device:40:12 host:1000:1044 device:1000:1045 host:4092:4096"
HIOPAMD_BUILD_VARIANT=poison scripts/cold_start_soak.sh 30 $O/cold_start_soak_poison.txt "host:400:100 device:400:100 host:401:100 host:40:12 host:1000:1044 device:1000:1045 host:4092:4096"
