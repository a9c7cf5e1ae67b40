#!/bin/bash
# round 3, GPU call 6: Gram strip kernel with the swizzled plane LDS layout — parity, timing, LDS bank-conflict counters.
set -u
mkdir -p gpurun_out/r03_6
export TMPDIR=/tmp
O=gpurun_out/r03_6
timeout 900 python -m pytest tests/test_gpu_dense_sparse.py tests/test_gpu_lowrank.py tests/test_gpu_full_size.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --ns 2000 --nd 256 --neq 253 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_6/bench.json"))
for key in ("dense_sharded","dense_n1e6_m100"):
    e=d[key]; print(key, round(e["ms_per_step"],3), [ (r["kernel"][:12], round(r["avg_launch_ms"],3), round(r["frac"],3)) for r in e["roofline"]])
PY
R=$GRAFT_REPO_ROOT
(cd /tmp && HIOPAMD_DF=0 timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $R/$O/lds -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --ns 2000 --nd 256 --neq 253 > /dev/null 2> $R/$O/lds.err); echo "lds exit $?"
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/r03_6/lds/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(fs[0])):
    k = row.get("Kernel_Name", "?")
    if "gram_strip2" in k: agg[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items(): print(k, {c: f"{x:.3e}" for c, x in v.items()})
PY
find $O/lds -type f -delete
