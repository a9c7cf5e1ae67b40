#!/bin/bash
# Round 5, call 33b: the two sparse bench entries after the LDS staging of the sparse solve sweeps
set -u
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ns 2000 --nd 256 --neq 253 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('sparse_condensed_n1e6','sparse_condensed_banded_n1e6'):
    print(k, '%.3f ms/step' % d[k]['ms_per_step'])" | tee gpurun_out/r05_33/bench.txt
