#!/bin/bash
# Round 5, call 36: the bench line once more on the final library and the final committed PMC summaries (profiles/r05_full_bench.json)
set -u
export TMPDIR=/tmp
O=gpurun_out/r05_36
mkdir -p $O
timeout -s KILL 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit: $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_36/bench.json").read().strip().splitlines()[-1])
s = d["kkt_spans"]
print("value %.2f it/s  %.3f ms/step  frac %.3f  wide %.3f ms | fact %.3f solves %.3f | dense_sharded %.2f  c2 %.2f  sparse %.2f  banded %.2f  cpu %.3f" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], s["linsolv.tmFactTime"]["ms_per_step"], s["linsolv.tmTriuSolves"]["ms_per_step"],
    d["dense_sharded"]["ms_per_step"], d["dense_n1e6_m100"]["ms_per_step"], d["sparse_condensed_n1e6"]["ms_per_step"], d["sparse_condensed_banded_n1e6"]["ms_per_step"], d["cpu_baseline"]["value"]))
for r in d["dense_sharded"]["roofline"]:
    print(r["kernel"][:28], "frac %.3f traffic/alg %.3f" % (r["frac"], r["traffic"] / r["algorithmic_bytes_per_launch"]))
print("e2e", d["ipm_end_to_end_N8192"]["device"]["kkt_iterations_per_s"])
PY
