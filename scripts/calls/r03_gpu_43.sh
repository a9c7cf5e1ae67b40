#!/bin/bash
# round 3, GPU call 43: the badly scaled dense problem through the C interface, with the iteration table
set -u
export TMPDIR=/tmp
gcc -O2 -I include tests/c/dense_c_interface.c -o /tmp/dense_c -L hiop_amd/lib -lhiopamd -Wl,-rpath,$PWD/hiop_amd/lib -lm 2>&1 | tail -3
DENSE_KOBJ=1e4 DENSE_KROW=1e3 timeout 300 /tmp/dense_c 500 2>&1 | tail -45
