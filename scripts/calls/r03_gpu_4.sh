#!/bin/bash
# round 3, GPU call 4: the tests that failed in call 3 + kernel trace of the bench (dense step breakdown).
set -u
mkdir -p gpurun_out/r03_4
export TMPDIR=/tmp
O=gpurun_out/r03_4
echo "=== pytest (subset) ==="
timeout 900 python -m pytest tests/test_gpu_kkt_xycyd.py tests/test_gpu_krylov.py tests/test_gpu_ipm_device.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest.log
echo "=== rocprofv3 kernel trace of bench ==="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err; echo "prof exit $?"
cd $GRAFT_REPO_ROOT
ls $O/prof | head; find $O/prof -name "*kernel_stats*" | head -2
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
# keep only the small csv summaries (the merge back is limited to 64 MiB)
find $O/prof -type f ! -name "*stats*.csv" -delete
