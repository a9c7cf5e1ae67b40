#!/bin/bash
# Round 6, call 27: the headline step (bench.py) at orders that are NOT the bench's 8192: no padding / parity only / the cost model's choice
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_27
mkdir -p $O
for dims in "4000 4000" "4096 4092" "4097 4093" "3500 3000" "5000 4200"; do
  set -- $dims
  for pad in 0 1 2; do
    HIOPAMD_LDLT_PAD=$pad timeout 600 python bench.py --nd $1 --neq $2 --no-cpu-baseline --no-dense > $O/b.json 2> $O/b.err
    python - "$1" "$2" "$pad" <<'PY' | tee -a $O/steps.txt
import json, sys
d = json.load(open("gpurun_out/r06_27/b.json"))
sp = d["kkt_spans"]
print("nd %s neq %s pad %s: N = %s | %.2f it/s %.3f ms | fact %.3f ms | solves %.3f ms" % (sys.argv[1], sys.argv[2], sys.argv[3], d["config"].get("N", "?"), d["value"], d["ms_per_step"],
      sp["linsolv.tmFactTime"]["ms_per_step"], sp["linsolv.tmTriuSolves"]["ms_per_step"]))
PY
  done
done
exit 0
