#!/bin/bash
# round 2, GPU pass 2: double-buffered 128x128 update kernel (parity + A/B timing), 2-rank diagnostics
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest ldlt + two-rank ==="
timeout 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_two_rank.py -q > gpurun_out/pytest_2.log 2>&1
echo "pytest exit: $?"; tail -12 gpurun_out/pytest_2.log
for v in 0 1; do
  echo "=== update kernel timing, HIOPAMD_UPD_OLD=$v ==="
  HIOPAMD_UPD_OLD=$v timeout 300 python scripts/upd_time.py 2>&1 | tail -1
  HIOPAMD_UPD_OLD=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dense 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s %.3f ms; upd %.3f ms/step %.1f TF' % (d['value'], d['ms_per_step'], d['roofline']['update_ms_per_step'], d['roofline']['achieved']))"
done
cat gpurun_out/two_rank_debug_4001.json | head -60
