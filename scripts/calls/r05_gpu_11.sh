#!/bin/bash
# round 5, call 11: is the wide kernel dealt one workgroup per CU?  A/B: grid = 240 against grid = 480 with the second workgroup of every CU leaving at once
for round in 1 2 3; do
  for v in base wide2x; do
    if [ $v = wide2x ]; then export HIOPAMD_DEV_WIDE2X=1; else unset HIOPAMD_DEV_WIDE2X; fi
    r=$(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dense 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s %.3f ms fact %.3f wide %.3f' % (d['value'], d['ms_per_step'], d['kkt_spans']['linsolv.tmFactTime']['ms_per_step'], d['roofline']['avg_launch_ms']))")
    echo "$v: $r"
  done
done
