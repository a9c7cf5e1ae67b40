#!/bin/bash
# round 3, GPU call 40: small sweep of the pairing / retirement boundary with the final polling set-up, then the full pass
set -u
mkdir -p gpurun_out/r03_40
export TMPDIR=/tmp
O=gpurun_out/r03_40
for cfg in "HIOPAMD_DF_K512=16 HIOPAMD_DF_RETIRE=16" "HIOPAMD_DF_K512=20 HIOPAMD_DF_RETIRE=20" "HIOPAMD_DF_K512=12 HIOPAMD_DF_RETIRE=12" "HIOPAMD_DF_K512=16 HIOPAMD_DF_RETIRE=12" "HIOPAMD_DF_K512=16 HIOPAMD_DF_RETIRE=16"; do
  echo "=== $cfg"; env $cfg DF_TIMELINE=0 timeout -s KILL 180 python scripts/df_stamps.py 2>&1 | tail -1
done
( time timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 ) 2>&1 | grep real; grep -h "passed\|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_40/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "fact ms", d["kkt_spans"]["linsolv.tmFactTime"]["ms_per_step"])
for k in ("dense_sharded", "dense_n1e6_m100", "sparse_condensed_n1e6"):
    print(k, d[k].get("value"), d[k].get("ms_per_step"))
print(d.get("ipm_end_to_end_N8192", {}).get("device"))
PY
tail -2 $O/bench.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "rocprof exit $?"; cd $GRAFT_REPO_ROOT
head -4 $O/prof/bench_kernel_stats.csv | cut -c1-150
