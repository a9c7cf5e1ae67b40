#!/bin/bash
# Round 4: soak with the debug dump until a bounded wait expires (eight-wave form): who holds what at that moment?
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_soak_debug
mkdir -p $O
for i in $(seq 1 ${SOAK_CHUNKS:-12}); do
  env HIOPAMD_DF_DEBUG=1 ${SOAK_ENV:-} DF_REPS=625 DF_OBJECTS=4 timeout -s KILL 600 python scripts/df_repeat.py > $O/chunk_$i.log 2>&1; rc=$?
  echo "chunk $i exit $rc: $(tail -1 $O/chunk_$i.log | cut -c1-140)"
  if grep -q "timed out" $O/chunk_$i.log; then
    grep -v "kfd evicted" $O/chunk_$i.log | grep "hiop_amd" | cut -c1-260 | head -120
    break
  fi
done
