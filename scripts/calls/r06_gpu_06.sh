#!/bin/bash
# Round 6, call 6: the fused solveCompressed of the low-rank KKT (8 launches per solve): parity tests of everything that goes through it,
# then the dense step timings (before: 2.50 ms at n = 1e6, m = 100; 5.57 ms at n_local = 1.25e6, m = 200 — call 3 of this round)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_06
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_kkt_xycyd.py tests/test_gpu_two_rank.py tests/test_gpu_full_size.py tests/test_gpu_dense_sparse.py tests/test_c_interface.py tests/test_gpu_ipm_slab.py tests/test_reference_known_answers.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest exit: $?"; tail -15 $O/pytest.log
python scripts/dense_step_time.py 2>&1 | tee $O/dense_step_time.txt
python scripts/dense_step_phases.py 2>&1 | tee $O/dense_step_phases.txt
# the sparse LDL^T with the row-by-row backward sweep: its tests, the per-level timings, the bench entries
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py tests/test_gpu_kkt_sparse.py tests/test_gpu_csr_condensed.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sparse.log 2>&1
echo "pytest(sparse) exit: $?"; tail -4 $O/pytest_sparse.log
python scripts/sparse_ldl_levels.py 2>&1 | tee $O/sparse_ldl_levels.txt
