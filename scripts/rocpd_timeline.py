#!/usr/bin/env python3
"""Dump the kernel timeline (start/end relative to the first selected kernel, queue, name) of one factorisation
from a rocprofv3 rocpd sqlite trace: rows between the N-th and (N+1)-th `ldlt_inertia_kernel`."""
import sqlite3
import sys


def main(db_path, out_txt, which=3):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else None
    sel = "start, end, name" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
    rows = db.execute(f"select {sel} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "ldlt_inertia_kernel" in r[2]]
    if len(marks) <= which:
        print("not enough factorizations in trace", len(marks))
        return
    lo, hi = marks[which - 1] + 1, marks[which] + 1
    t0 = rows[lo][0]
    with open(out_txt, "w") as f:
        f.write("columns: " + ",".join(cols) + "\n")
        for r in rows[lo:hi]:
            nm = r[2].split("(")[0][-40:]
            f.write(f"{(r[0]-t0)/1e3:10.1f} {(r[1]-t0)/1e3:10.1f} {(r[1]-r[0])/1e3:8.1f} q={r[3]} s={r[4]} {nm}\n")
    print(f"{out_txt}: {hi-lo} kernels, span {(rows[hi-1][1]-t0)/1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
