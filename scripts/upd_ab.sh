#!/bin/bash
for v in "$@"; do
  r=$(env HIOPAMD_UPD4=$v timeout 200 python scripts/upd_time.py 2>&1 | tail -1)
  echo "UPD4=$v: $r"
done
