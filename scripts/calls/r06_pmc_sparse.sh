#!/bin/bash
# Round 6: PMC passes over the sparse LDL^T (banded n = 1e6): HBM bytes of the level kernels (FETCH_SIZE, WRITE_SIZE: separate passes,
# --kernel-trace only) and the LDS / VALU share of the register-resident factor kernel -> gpurun_out/r06_pmc_sparse/summary.json
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06_pmc_sparse
mkdir -p $O
pass() {
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/$name -o pmc --output-format csv -- python $R/scripts/sparse_ldl_time.py 1000000 5 > $R/$O/$name.out 2> $R/$O/$name.err); echo "$name exit $?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
python - <<'PY'
import csv, glob, json, collections
out = {}
for d in ("fetch", "write", "lds", "valu"):
    fs = glob.glob(f"gpurun_out/r06_pmc_sparse/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        out[d] = "no counter file"; continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "sl_" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    out[d] = {k: dict(launches=len(cnt[k]), **{c: v for c, v in acc[k].items()}) for k in acc}
json.dump(out, open("gpurun_out/r06_pmc_sparse/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf $O/fetch $O/write $O/lds $O/valu
exit 0
