#!/bin/bash
# PMC passes over the bench (separate rocprofv3 runs, kernel-trace only, as the GPU pool requires).
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, counters...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc/$name -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense > $R/gpurun_out/pmc/$name.json 2> $R/gpurun_out/pmc/$name.err); echo "$name exit $?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64
find gpurun_out/pmc -name "*.csv" | head -20
python3 - <<'PY'
import csv, glob, collections
for d in ["fetch","write","mfma","wait"]:
    fs = glob.glob(f"gpurun_out/pmc/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name","?")[:50]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:8]:
        print(d, k, dict(agg[k]))
PY
