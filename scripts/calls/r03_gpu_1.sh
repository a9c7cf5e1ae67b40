#!/bin/bash
# round 3, GPU call 1: parity of the changed kernels (512-block solve, fused assembly, vector deltas, strip Gram v2, fused secant),
# A/B timings (solve block size / lead, Gram strip form), the one-dispatch dataflow factorisation and its PMC passes.
set -u
mkdir -p gpurun_out/r03_1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r03_1
echo "=== pytest (changed areas) ==="
timeout 900 python -m pytest tests/test_gpu_ldlt_kkt.py tests/test_gpu_dense_sparse.py tests/test_gpu_lowrank.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest.log
echo "=== factor + 3 solves: default (B=512) ==="
DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -2
for lead in 1 3 4; do echo "--- lead $lead"; HIOPAMD_SOLVE_LEAD=$lead DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -2 | head -1; done
echo "--- B=256"; HIOPAMD_SOLVE_B=256 DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -2
echo "=== one-dispatch dataflow factorisation (HIOPAMD_DF_ONE=1) ==="
HIOPAMD_DF_ONE=1 DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -2
echo "=== bench (default) ==="
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_1/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["kkt_spans"].items():
    if isinstance(v,dict): print(" ", k, round(v["ms_per_step"],4))
for key in ("dense_sharded","dense_n1e6_m100"):
    e=d[key]; print(key, round(e["ms_per_step"],3), [ (r["kernel"][:12], round(r["avg_launch_ms"],3), round(r["frac"],3)) for r in e["roofline"]])
PY
echo "=== bench with the first strip Gram form ==="
HIOPAMD_GRAM_STRIP=1 timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_strip1.json 2> $O/bench_strip1.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_1/bench_strip1.json"))
for key in ("dense_sharded","dense_n1e6_m100"):
    e=d[key]; print(key, round(e["ms_per_step"],3), [ (r["kernel"][:12], round(r["avg_launch_ms"],3), round(r["frac"],3)) for r in e["roofline"]])
PY
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
    fs = glob.glob(f"gpurun_out/r03_1/pmc_{d}/**/*counter_collection.csv", recursive=True)
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
json.dump(out, open("gpurun_out/r03_1/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
