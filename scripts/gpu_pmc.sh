#!/bin/bash
# PMC passes over the bench (separate rocprofv3 runs, kernel-trace only, as the GPU pool requires).
# Counter collection serialises the dispatches of a process, and the dataflow LDL^T is a PAIR of persistent kernels that
# must run concurrently (each waits for the other's flags): under --pmc the pair cannot make progress.  The passes are
# therefore taken with HIOPAMD_DF=0 — the stepwise kernels, same tile algorithm launched once per super-panel step — and
# labelled so; bench.py reports `traffic: null` for the dataflow kernel.
export HIOPAMD_DF=0
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
# per-kernel aggregation of each pass + the per-launch HBM traffic of the dominant kernel
# (MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports by 2x on gfx950)
import csv, glob, collections, json
summary = {}
for d in ["fetch", "write", "mfma", "wait"]:
    fs = glob.glob(f"gpurun_out/pmc/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row.get("Kernel_Name", "?")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
    names = sorted({c for k in agg for c in agg[k]})
    with open(f"gpurun_out/pmc/{d}_by_kernel.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Kernel_Name", "Dispatches"] + names)
        for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
            w.writerow([k, len(disp[k])] + [agg[k].get(c, 0.0) for c in names])
    # the trailing update of the STEPWISE path = the instantiation of the tile kernel with the largest counter totals
    cand = [k for k in agg if "ldlt_update_kernel" in k]
    if cand:
        k = max(cand, key=lambda k: sum(agg[k].values()))
        summary.setdefault("ldlt_update_kernel_stepwise", {})["dispatches_" + d] = len(disp[k])
        summary["ldlt_update_kernel_stepwise"]["kernel_" + d] = k.split("(")[0]
        for c in names:
            summary["ldlt_update_kernel_stepwise"][c] = agg[k].get(c, 0.0)
    for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:6]:
        print(d, k[:60], len(disp[k]), dict(agg[k]))
u = summary.get("ldlt_update_kernel_stepwise")
if u and "FETCH_SIZE" in u and "WRITE_SIZE" in u:
    n = u["dispatches_fetch"]
    u["hbm_bytes_per_launch"] = (2.0 * u["FETCH_SIZE"] * 1024.0 / n) + (u["WRITE_SIZE"] * 1024.0 / u["dispatches_write"])
    u["note"] = "FETCH_SIZE[KB]*1024*2 (gfx950 correction) + WRITE_SIZE[KB]*1024, per launch; separate --pmc passes, kernel-trace only"
json.dump(summary, open("gpurun_out/pmc/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
