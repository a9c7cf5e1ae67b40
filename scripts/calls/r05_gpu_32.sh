#!/bin/bash
# Round 5, call 32: y = J x with the XCD-aware block map (x fetched once per chunk group instead of once per row tile): parity + A/B of the dense entries
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_32
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dense or matrix or lowrank or known_answers" 2>&1 | tail -3 | tee $O/pytest.log
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in base gemvmap base gemvmap; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ns 2000 --nd 256 --neq 253 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['dense_sharded']; b=d['dense_n1e6_m100']
print('$v: sharded %.3f ms  c2 %.3f ms | y=Jx %.3f ms (%.0f GB/s)  x=JTy %.3f ms | c2 y=Jx %.3f ms' % (a['ms_per_step'], b['ms_per_step'], a['roofline'][1]['avg_launch_ms'], a['roofline'][1]['achieved'], a['roofline'][2]['avg_launch_ms'], b['roofline'][1]['avg_launch_ms']))" | tee -a $O/ab.txt
done
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
