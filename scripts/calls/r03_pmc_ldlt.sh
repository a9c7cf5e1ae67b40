#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA — separate rocprofv3 runs, --kernel-trace only) on the ONE-dispatch form of the dataflow LDL^T
# (HIOPAMD_DF_ONE=1: chain roles + wide workgroups in one kernel, the only form counter collection can observe); summary ->
# profiles/r03_pmc/summary.json (bench.py reads roofline.traffic from it)
set -u
mkdir -p gpurun_out/r03_pmc2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03_pmc2
echo "=== PMC passes on the one-dispatch factorisation ==="
export HIOPAMD_DF_ONE=1
run() {  # name, counters...
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/pmc_$name -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense > $R/$O/pmc_$name.json 2> $R/$O/pmc_$name.err); echo "$name exit $?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d in ["fetch", "write", "mfma"]:
    fs = glob.glob(f"gpurun_out/r03_pmc2/pmc_{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name", "?")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
    for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:5]:
        print(d, k[:70], len(disp[k]), dict(agg[k]))
    for k in agg:
        if "ldlt_df_one_kernel" in k:
            e = out.setdefault("ldlt_df_one_kernel", {})
            e["dispatches_" + d] = len(disp[k])
            for c, v in agg[k].items():
                e[c] = v
e = out.get("ldlt_df_one_kernel")
if e and "FETCH_SIZE" in e and "WRITE_SIZE" in e:
    e["hbm_bytes_per_launch"] = 2.0 * e["FETCH_SIZE"] * 1024.0 / e["dispatches_fetch"] + e["WRITE_SIZE"] * 1024.0 / e["dispatches_write"]
    e["note"] = "FETCH_SIZE[KB]*1024*2 (gfx950 correction) + WRITE_SIZE[KB]*1024, per launch; separate --pmc passes, kernel-trace only; chain + wide roles of the dataflow LDL^T as ONE dispatch (HIOPAMD_DF_ONE=1)"
json.dump(out, open("gpurun_out/r03_pmc2/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

find $O -type f ! -name "pmc_summary.json" ! -name "*.err" -delete
