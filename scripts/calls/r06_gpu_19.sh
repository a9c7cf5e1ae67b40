#!/bin/bash
# Round 6, call 19: the register-resident factor kernel — column broadcast through LDS (0), half LDS / half readlane (1), readlane only (2)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_19
mkdir -p $O
cp hiop_amd/lib/libhiopamd.so /tmp/shipped.so
for v in split0 split1 split2 split0 split1 split2; do
  cp build_variants/$v.so hiop_amd/lib/libhiopamd.so
  echo "== $v" | tee -a $O/sparse_time.txt
  timeout 300 python scripts/sparse_ldl_time.py 1000000 5 2>&1 | tail -1 | tee -a $O/sparse_time.txt
  timeout 300 python scripts/sparse_ldl_time.py 1000000 3 2>&1 | tail -1 | tee -a $O/sparse_time.txt
done
cp build_variants/split1.so hiop_amd/lib/libhiopamd.so
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_sparse_split1.txt
cp build_variants/split2.so hiop_amd/lib/libhiopamd.so
timeout 900 python -m pytest tests/test_gpu_sparse_ldl.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_sparse_split2.txt
cp /tmp/shipped.so hiop_amd/lib/libhiopamd.so
exit 0
