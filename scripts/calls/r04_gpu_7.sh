#!/bin/bash
# Round 4, GPU call 7: staggered LDS-DMA issue (waves 4-7 two k-steps behind waves 0-3) against issuing together (HIOPAMD_DF_EXP=8),
# the whole GPU suite (exact transposed / symmetric SpMV, batched reductions, retry copy), bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_7
mkdir -p $O
t() { name=$1; shift; env "$@" DF_TIMELINE=${TL:-0} timeout -s KILL 150 python scripts/df_stamps.py > $O/$name.log 2>&1; echo "$name: exit $? | $(grep -h 'matrixChanged' $O/$name.log | tr '\n' ' ')"; }
t w8_pipe3 HIOPAMD_DF_PIPE=3
t w8_pipe0 HIOPAMD_DF_PIPE=0
t legacy240 HIOPAMD_DF_FORM=4
t legacy480 HIOPAMD_DF_FORM=4 HIOPAMD_DF_WGS=480
for e in 0 8; do TL=1 t exp$e HIOPAMD_DF_PIPE=3 HIOPAMD_DF_EXP=$e; grep -h "wide kernel phases\|shader clock" $O/exp$e.log | cut -c1-330; done
timeout -s KILL 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?: $(tail -3 $O/pytest.log | tr '\n' ' ')"
timeout -s KILL 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_7/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('kkt_spans'))
PY
