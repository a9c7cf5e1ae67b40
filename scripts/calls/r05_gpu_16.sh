#!/bin/bash
# round 5, call 16: host waits on the critical path poll hipStreamQuery instead of sleeping in hipStreamSynchronize: A/B (sync = the library before)
for round in 1 2; do
  for v in sync spin; do
    cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
    r=$(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s %.3f ms fact %.3f | dense %.3f %.3f | sparse %.3f %.3f' % (d['value'], d['ms_per_step'], d['kkt_spans']['linsolv.tmFactTime']['ms_per_step'], d['dense_sharded']['ms_per_step'], d['dense_n1e6_m100']['ms_per_step'], d['sparse_condensed_n1e6']['ms_per_step'], d['sparse_condensed_banded_n1e6']['ms_per_step']))")
    echo "$v: $r"
  done
done
cp build_variants/spin.so hiop_amd/lib/libhiopamd.so
