#!/bin/bash
# Round 6, last call: the whole suite on the final poison build, then the padded-order soak, on the round's final library
set -u
bash scripts/calls/r06_poison_suite.sh
bash scripts/calls/r06_soak2.sh
exit 0
