#!/bin/bash
# round 3, GPU call 55: recovery test after the wrapper's retry switch
timeout 20 python -m pytest tests/test_gpu_ldlt_timeout_recovery.py -m gpu -q -k timeout_recovery 2>&1 | tail -3
