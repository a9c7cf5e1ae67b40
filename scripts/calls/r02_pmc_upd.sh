#!/bin/bash
# PMC passes over the trailing-update kernel only (scripts/upd_time.py), kernel-trace only (pool rule)
set -u
mkdir -p gpurun_out/pmcu
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT

run() {
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmcu/$name -o pmc --output-format csv -- env PYTHONPATH=$R python $R/scripts/upd_time.py > $R/gpurun_out/pmcu/$name.out 2> $R/gpurun_out/pmcu/$name.err); echo "$name exit $?"
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS
run p2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p3 SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
python3 - <<'PY'
import csv, glob, collections
for d in ["p1", "p2", "p3"]:
    fs = glob.glob(f"gpurun_out/pmcu/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name", "?").split("(")[0][:50]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row.get("Dispatch_Id"))
    for k in agg:
        if "update_db" in k or "update_kernel_t" in k:
            print(d, k, len(n[k]), {c: f"{v:.4g}" for c, v in agg[k].items()})
PY
