#!/bin/bash
# round 3, GPU call 16: the C interface at the headline size (stock MdsEx1 with N = nd + ns + 3 = 8192), device callbacks
set -u
export TMPDIR=/tmp
gcc -std=c11 -O1 -Iinclude tests/c/mds_c_interface.c -o /tmp/mds_c -Lhiop_amd/lib -lhiopamd -lm -Wl,-rpath,$PWD/hiop_amd/lib
( time timeout 600 /tmp/mds_c device 4092 4097 ) 2>&1 | tail -45
( time timeout 600 /tmp/mds_c host 4092 4097 ) 2>&1 | tail -6
