#!/bin/bash
# Round 4, call 14: dense low-rank step after software-pipelined loads in x = J^T y and the secant pass, the tile-driven Gram fold and
# the in-place shift + append of the secant memory: parity tests of the touched paths, step time, kernel time line
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_14; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_full_size.py tests/test_gpu_two_rank.py tests/test_gpu_dense_sparse.py tests/test_gpu_ipm_device.py tests/test_c_interface.py tests/test_reference_known_answers.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -20
for i in 1 2; do STEPS=15 timeout -s KILL 300 python scripts/dense_step_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dense.log; done
(cd /tmp && STEPS=6 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o dense -- python $R/scripts/dense_step_time.py > $R/$O/dense_prof.log 2>&1); echo "rocprof exit $?"
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $O/dense_kernel_trace.csv; rm -rf $O/prof
python scripts/dense_trace_summary.py $O/dense_kernel_trace.csv > $O/dense_trace_summary.txt 2>&1; rm -f $O/dense_kernel_trace.csv; head -30 $O/dense_trace_summary.txt
