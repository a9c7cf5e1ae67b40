#!/bin/bash
# Round 6, call 31: KKT object at a padded order with mode switches between assembly and factorisation (normal + poison build)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_31
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -x -q -p no:cacheprovider -k "mode_switches or assemble_factor_solve or vector_regularisation" > $O/log.txt 2>&1; grep -E "passed|failed|Error|assert" $O/log.txt | tail -8 | tee $O/pytest.txt
HIOPAMD_BUILD_VARIANT=poison timeout 600 python -m pytest tests/test_gpu_ldlt_kkt.py -m gpu -x -q -p no:cacheprovider -k "mode_switches" > $O/logp.txt 2>&1; grep -E "passed|failed|Error|assert" $O/logp.txt | tail -8 | tee -a $O/pytest.txt
exit 0
