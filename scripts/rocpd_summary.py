#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into a per-kernel CSV (name, calls, total/avg/min/max)."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPRs", "AccVGPRs", "LDS", "Scratch"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3), r[6], r[7], r[8], r[9]])
    print(f"{out_csv}: {len(rows)} kernels, total {tot/1e6:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
