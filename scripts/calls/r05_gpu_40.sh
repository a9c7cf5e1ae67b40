#!/bin/bash
# Round 5, call 40: per-level kernel durations of the sparse LDL^T (banded n = 1e6): factor, forward, backward sweeps
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_40
mkdir -p $O
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -o sl -- python $R/scripts/sparse_ldl_levels.py > $R/$O/run.txt 2> $R/$O/prof.err); echo "rocprof exit: $?"
cat $O/run.txt | grep -v amdgpu
python3 - <<'PY'
import csv, glob
fs = glob.glob("gpurun_out/r05_40/prof/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows if "sl_" in r["Kernel_Name"]]
half = len(sel) // 2
sel = sel[half:]          # second round
out = open("gpurun_out/r05_40/levels.txt", "w")
for kind in ("sl_factor_level", "sl_fwd_level", "sl_bwd_level"):
    ks = [s for s in sel if kind in s[0]]
    line = "%s: %d launches, total %.1f us | per level (workgroups: us): %s" % (kind, len(ks), sum(k[1] for k in ks), "  ".join("%d: %.1f" % (k[3] // max(k[2], 1), k[1]) for k in ks))
    print(line); out.write(line + "\n")
PY
rm -rf $O/prof
