#!/bin/bash
# round 3, GPU call 51: final pass — GPU suite, smoke, bench, kernel trace of the bench, then a soak with the pinned result buffer
set -u
mkdir -p gpurun_out/r03_51
export TMPDIR=/tmp
O=gpurun_out/r03_51
( time timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1 ) 2>&1 | grep real; grep -h "passed\|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_51/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "fact ms", d["kkt_spans"]["linsolv.tmFactTime"]["ms_per_step"])
for k in ("dense_sharded", "dense_n1e6_m100", "sparse_condensed_n1e6"):
    print(k, d[k].get("value"), d[k].get("ms_per_step"))
print(d.get("ipm_end_to_end_N8192", {}).get("device"))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "rocprof exit $?"; cd $GRAFT_REPO_ROOT
for i in $(seq 1 14); do
env HIOPAMD_DF_DEBUG=1 DF_REPS=400 DF_OBJECTS=4 timeout -s KILL 300 python scripts/df_repeat.py > $O/soak_$i.log 2>&1; rc=$?
echo "soak $i exit $rc: $(tail -1 $O/soak_$i.log | cut -c1-140)"
if grep -q "timed out" $O/soak_$i.log; then grep "bounded wait\|failed after\|in the tile loop at stage" $O/soak_$i.log | cut -c1-200 | head -12; break; fi
done
