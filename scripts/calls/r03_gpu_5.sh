#!/bin/bash
# round 3, GPU call 5: the C interface (host and device callbacks) + regularisation tests
set -u
mkdir -p gpurun_out/r03_5
export TMPDIR=/tmp
O=gpurun_out/r03_5
timeout 900 python -m pytest tests/test_c_interface.py tests/test_gpu_kkt_xycyd.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -30 $O/pytest.log
gcc -std=c11 -O1 -Iinclude tests/c/mds_c_interface.c -o /tmp/mds_c -Lhiop_amd/lib -lhiopamd -lm -Wl,-rpath,$PWD/hiop_amd/lib
( time timeout 300 /tmp/mds_c host ) 2>&1 | tail -30
( time timeout 300 /tmp/mds_c device ) 2>&1 | tail -8
