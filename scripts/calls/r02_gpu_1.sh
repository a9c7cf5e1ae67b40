#!/bin/bash
# round 2, GPU pass 1: new multi-rank test + touched paths, bench with spans/dense rooflines, vendor DGEMM yardstick
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "=== pytest (two-rank, lowrank, xycyd, slab, abi) ==="
timeout 900 python -m pytest tests/test_gpu_two_rank.py tests/test_gpu_lowrank.py tests/test_gpu_kkt_xycyd.py tests/test_gpu_ipm_slab.py tests/test_gpu_ldlt_kkt.py -x -q > gpurun_out/pytest_1.log 2>&1
echo "pytest exit: $?"; tail -15 gpurun_out/pytest_1.log
echo "=== bench --gpus 2 on a 1-GPU box must fail loudly ==="
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_g2.out 2>&1; echo "bench --gpus 2 exit: $?"; tail -2 gpurun_out/bench_g2.out
echo "=== bench ==="
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
echo "=== vendor DGEMM yardstick ==="
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dgemm_prof -o dgemm -- python $R/scripts/probes/dgemm_yardstick.py $R/gpurun_out/dgemm_yardstick.json > $R/gpurun_out/dgemm.log 2>&1); echo "dgemm exit: $?"
cat gpurun_out/dgemm.log | tail -8
f=$(find gpurun_out/dgemm_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -8
