#!/bin/bash
# Round 6 PMC passes (separate rocprofv3 runs per counter group, --kernel-trace only):
#  1. the ONE-dispatch form of the dataflow LDL^T (HIOPAMD_DF_ONE=1: chain roles + wide workgroups in one kernel — counter collection
#     serialises dispatches, so the shipped pair of concurrently running kernels cannot be observed: the wide kernel would wait for a
#     chain kernel that is not started before it ends) -> gpurun_out/r06_pmc/summary.json
#  2. the three kernels of the dense low-rank step at bench.py's two shapes (scripts/pmc_dense_kernels.py) -> gpurun_out/r06_pmc_dense/summary.json
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r06_pmc
mkdir -p $O gpurun_out/r06_pmc_dense
if [ "${1:-all}" != dense_only ]; then
echo "=== PMC passes on the one-dispatch factorisation ==="
run() {  # name, counters...
  name=$1; shift
  (cd /tmp && HIOPAMD_DF_ONE=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/pmc_$name -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense > $R/$O/pmc_$name.json 2> $R/$O/pmc_$name.err); echo "$name exit $?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d in ["fetch", "write", "mfma"]:
    fs = glob.glob(f"gpurun_out/r06_pmc/pmc_{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name", "?")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
    for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:4]:
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
    e["note"] = ("FETCH_SIZE[KB]*1024*2 (gfx950 correction) + WRITE_SIZE[KB]*1024, per launch; separate --pmc passes, kernel-trace only; chain + wide "
                 "roles of the dataflow LDL^T as ONE dispatch (HIOPAMD_DF_ONE=1); round 6 library (round 5 kernels; padding of the compact diagonal blocks defined, inversion masked)")
json.dump(out, open("gpurun_out/r06_pmc/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -type f ! -name "summary.json" ! -name "*.err" -delete
find $O -type d -empty -delete
fi

echo "=== PMC passes on the dense low-rank kernels ==="
O=gpurun_out/r06_pmc_dense
rund() {  # shape-name k n l pass-name counters...
  sh=$1; k=$2; n=$3; l=$4; name=$5; shift 5
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/${sh}_$name -o pmc --output-format csv -- python $R/scripts/pmc_dense_kernels.py $k $n $l 4 > $R/$O/${sh}_$name.out 2> $R/$O/${sh}_$name.err); echo "$sh $name exit $?"
}
for shape in "k200_n1250000 200 1250000 6" "k100_n1000000 100 1000000 6"; do
  rund $shape fetch FETCH_SIZE
  rund $shape write WRITE_SIZE
  rund $shape mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64
done
python3 - <<'PY'
import csv, glob, collections, json
KEYS = {"gram_strip2_kernel": "gram_weighted_stacked", "gemv_n_stage1": "mat_times_vec", "gemv_t_kernel": "mat_trans_times_vec"}
summary = {}
for sh in ["k200_n1250000", "k100_n1000000"]:
    S = summary.setdefault(sh, {})
    for d in ["fetch", "write", "mfma"]:
        fs = glob.glob(f"gpurun_out/r06_pmc_dense/{sh}_{d}/**/*counter_collection.csv", recursive=True)
        if not fs:
            print(sh, d, "no csv"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for row in csv.DictReader(open(fs[0])):
            k = row.get("Kernel_Name", "?")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
        for k in agg:
            print(sh, d, k[:60], len(disp[k]), {c: "%.4g" % v for c, v in agg[k].items()})
            for pat, short in KEYS.items():
                if pat in k:
                    u = S.setdefault(short, {"kernel": k[:80]})
                    u["dispatches_" + d] = len(disp[k])
                    for c, v in agg[k].items():
                        u[c] = v
    for short, u in S.items():
        if "FETCH_SIZE" in u and "WRITE_SIZE" in u:
            u["hbm_bytes_per_launch"] = 2.0 * u["FETCH_SIZE"] * 1024.0 / u["dispatches_fetch"] + u["WRITE_SIZE"] * 1024.0 / u["dispatches_write"]
summary["note"] = ("FETCH_SIZE[KB]*1024*2 (gfx950 correction) + WRITE_SIZE[KB]*1024 per launch of the kernel at ONE shape (scripts/pmc_dense_kernels.py: "
                   "k x n Jacobian, l = 6 secant pairs, 4 launches each); separate --pmc passes, kernel-trace only")
json.dump(summary, open("gpurun_out/r06_pmc_dense/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
find $O -type f ! -name "summary.json" ! -name "*.err" -delete
find $O -type d -empty -delete
