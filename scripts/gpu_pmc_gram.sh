#!/bin/bash
# TLB / latency counters of the Gram kernel (separate --pmc passes, kernel-trace only)
set -u
mkdir -p gpurun_out/pmcg
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmcg/$name -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmcg/$name.err); echo "$name exit $?"
}
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
python3 - <<'PY'
import csv, glob, collections
for d in ["tlb", "lat"]:
    fs = glob.glob(f"gpurun_out/pmcg/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name", "?")[:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k].add(row.get("Dispatch_Id"))
    for k in agg:
        if "gram_partial" in k or "ldlt_update_kernel" in k or "gemv_t" in k:
            print(d, k, len(cnt[k]), {c: f"{v:.3e}" for c, v in agg[k].items()})
PY
