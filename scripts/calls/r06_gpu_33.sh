#!/bin/bash
# Round 6, call 33: the driver's commands once more on the final tree: default bench, the 2-rank control flow rehearsed on one GPU, smoke
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_33
mkdir -p $O
s=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $? in $(( $(date +%s) - s )) s" | tee $O/summary.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print('headline %.2f it/s | frac %.3f | other orders %s | cpu %.2f it/s (%s cores)' % (d['value'], d['roofline']['frac'], [(o['N'], round(o['value'],1)) for o in d['mds_other_orders']['orders']], d['cpu_baseline']['value'], d['cpu_baseline']['cores']))" | tee -a $O/summary.txt
s=$(date +%s); HIOPAMD_BENCH_FAKE_MULTI=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; echo "2-rank rehearsal exit $? in $(( $(date +%s) - s )) s" | tee -a $O/summary.txt
python -c "
import json; d=json.load(open('$O/bench2.json')); print('n_gpus', d['n_gpus'], 'value %.2f' % d['value'], 'dense_sharded' in d)" | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee -a $O/summary.txt
exit 0
