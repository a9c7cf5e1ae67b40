#!/bin/bash
# A/B timing of library variants on ONE box: build_variants/<name>.so are swapped in turn, two rounds each.
for round in 1 2; do
  for v in "$@"; do
    cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
    r=$(timeout 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-dense 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "$v: $r"
  done
done
