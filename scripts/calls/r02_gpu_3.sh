#!/bin/bash
# round 2, GPU pass 3: dataflow LDL^T — parity, then timing (dataflow on / off).  Hard per-command limits.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== df_debug 8192 ==="
timeout -s KILL 60 python -u scripts/df_debug.py 8192 2>&1 | tail -4
echo "=== pytest dataflow ==="
timeout -s KILL 200 python -m pytest tests/test_gpu_ldlt_kkt.py -x -q > gpurun_out/pytest_3.log 2>&1
echo "pytest exit: $?"; tail -8 gpurun_out/pytest_3.log
for v in 1 0; do
  echo "=== bench HIOPAMD_DF=$v ==="
  HIOPAMD_DF=$v timeout -s KILL 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dense 2> gpurun_out/bench_df$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s %.3f ms; upd %.3f ms/step %.1f TF; fact %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['update_ms_per_step'], d['roofline']['achieved'], d['kkt_spans']['linsolv.tmFactTime']['ms_per_step']))"
  tail -3 gpurun_out/bench_df$v.err
done
