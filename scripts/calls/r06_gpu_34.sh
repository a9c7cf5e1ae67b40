#!/bin/bash
# Round 6, call 34: the default bench once more (bench.py now reads the sparse PMC summary into the banded entry's roofline object)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_34
mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python -c "
import json; d=json.load(open('$O/bench.json')); print('headline %.2f it/s frac %.3f' % (d['value'], d['roofline']['frac'])); print(json.dumps(d['sparse_condensed_banded_n1e6']['roofline'])[:900])"
exit 0
