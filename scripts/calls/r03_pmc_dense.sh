#!/bin/bash
# PMC passes over the DENSE low-rank step (Gram strip kernel, the two GEMVs, the one-pass secant update): separate rocprofv3 runs,
# --kernel-trace only.  The MDS part of the bench is shrunk (N = 512, stepwise LDL^T) so that the passes spend their time on the dense
# kernels; the dense shapes are the bench's (k = 200, n_local = 1.25e6 and k = 100, n = 1e6).
export HIOPAMD_DF=0
set -u
mkdir -p gpurun_out/pmcd
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, counters...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmcd/$name -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --ns 2000 --nd 256 --neq 253 > $R/gpurun_out/pmcd/$name.json 2> $R/gpurun_out/pmcd/$name.err); echo "$name exit $?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
python3 - <<'PY'
import csv, glob, collections, json
KEYS = {"gram_strip2_kernel<8, 13, 256>": "gram_strip2_k200", "gram_strip2_kernel<4, 7, 128>": "gram_strip2_k100", "gemv_n_stage1": "gemv_n_stage1",
        "gemv_t_kernel": "gemv_t", "secant_jac_kernel": "secant_jac"}
summary = collections.defaultdict(dict)
for d in ["fetch", "write", "mfma", "lds"]:
    fs = glob.glob(f"gpurun_out/pmcd/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    big = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name", "?")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
    names = sorted({c for k in agg for c in agg[k]})
    with open(f"gpurun_out/pmcd/{d}_by_kernel.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Kernel_Name", "Dispatches"] + names)
        for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
            w.writerow([k, len(disp[k])] + [agg[k].get(c, 0.0) for c in names])
    for k in agg:
        for pat, short in KEYS.items():
            if pat in k:
                summary[short]["dispatches_" + d] = len(disp[k])
                for c in names:
                    summary[short][c] = agg[k].get(c, 0.0)
for short, u in summary.items():
    if "FETCH_SIZE" in u and "WRITE_SIZE" in u:
        u["hbm_bytes_per_launch_avg"] = 2.0 * u["FETCH_SIZE"] * 1024.0 / u["dispatches_fetch"] + u["WRITE_SIZE"] * 1024.0 / u["dispatches_write"]
summary["note"] = ("FETCH_SIZE[KB]*1024*2 (gfx950 correction) + WRITE_SIZE[KB]*1024, averaged over the launches of the kernel in the run (gemv_*: launches of "
                   "different shapes, k x n and l x n, are averaged together); separate --pmc passes, kernel-trace only")
json.dump(summary, open("gpurun_out/pmcd/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
find gpurun_out/pmcd -type f ! -name "*by_kernel.csv" ! -name "summary.json" ! -name "*.err" -delete
