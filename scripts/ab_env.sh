#!/bin/bash
# A/B timing of environment switches of ONE library on ONE box: each argument is an env assignment ("X=1" or "X=1 Y=2")
for round in 1 2; do
  for v in "$@"; do
    r=$(env $v timeout 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-dense 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f it/s %.3f ms  upd %.3f ms %.1f TF  bwd.err %.2e' % (d['value'], d['ms_per_step'], r['update_ms_per_step'], r['achieved'], d['check']['kkt_backward_error']))")
    echo "$v: $r"
  done
done
