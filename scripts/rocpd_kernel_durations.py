#!/usr/bin/env python3
"""List the durations (us) of every dispatch of the kernels whose name contains a pattern, with their grid sizes."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
rows = db.execute("select name, (end-start)/1000.0, grid_x, grid_y, workgroup_x from kernels where name like ? order by start",
                  (f"%{pat}%",)).fetchall()
for r in rows:
    print(f"{r[1]:10.1f} us  grid=({r[2]},{r[3]}) wg={r[4]}")
