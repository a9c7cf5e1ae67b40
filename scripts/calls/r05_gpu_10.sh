for round in 1 2; do
  for v in sync defer2 defer2s; do
    if [ $v = defer2s ]; then export HIOPAMD_DEV_SYNC_BEFORE_CHAIN=1; cp build_variants/defer2.so hiop_amd/lib/libhiopamd.so; else unset HIOPAMD_DEV_SYNC_BEFORE_CHAIN; cp build_variants/$v.so hiop_amd/lib/libhiopamd.so; fi
    r=$(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dense 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s %.3f ms fact %.3f' % (d['value'], d['ms_per_step'], d['kkt_spans']['linsolv.tmFactTime']['ms_per_step']))")
    echo "$v: $r"
  done
done
