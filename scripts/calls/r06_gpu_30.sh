#!/bin/bash
# Round 6, call 30: the MDS KKT object assembles straight into the solver's padded copy: KKT / IPM / C-interface / poison tests, bench other orders
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_30
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_ldlt_timeout_recovery.py tests/test_c_interface.py tests/test_gpu_example_mds.py tests/test_gpu_ipm_device.py tests/test_gpu_kkt_xycyd.py tests/test_gpu_poisoned_allocations.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -5 | tee $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-dense > $O/b0.json 2> $O/b.err
timeout 600 python bench.py --no-cpu-baseline > $O/b.json 2>> $O/b.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_30/b.json"))
print("headline %.2f it/s %.3f ms | frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for o in d["mds_other_orders"]["orders"]:
    print("N %d: %.2f it/s %.3f ms" % (o["N"], o["value"], o["ms_per_step"]))
PY
exit 0
