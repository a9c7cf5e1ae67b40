#!/bin/bash
# round 5, call 19: wide kernel launched with two workgroups per CU, the second of every CU leaving at once (exactly one working workgroup per CU):
# LDL^T tests, bench x3
set -u
mkdir -p gpurun_out/r05_19
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_gpu_full_size.py tests/test_zz_gpu_dataflow_debug_dump.py tests/test_gpu_kkt_xycyd.py -q -x > gpurun_out/r05_19/pytest.log 2>&1; echo "pytest exit: $?"
tail -4 gpurun_out/r05_19/pytest.log
for r in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dense 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s %.3f ms fact %.3f wide %.3f frac %.4f' % (d['value'], d['ms_per_step'], d['kkt_spans']['linsolv.tmFactTime']['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
