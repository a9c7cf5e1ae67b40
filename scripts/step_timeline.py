"""Time line of ONE step out of a rocprofv3 kernel trace (CSV, --output-format csv): every kernel between two consecutive launches of a
marker kernel, with its start offset, duration and the idle gap in front of it; then the per-kernel sums of that step.
usage: step_timeline.py <kernel_trace.csv> <marker substring> [which occurrence from the end, default 3] [stride: marker launches per step, default 1]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][::stride]
a, b = idx[-back - 1], idx[-back]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  +%6.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = max(prev_end, e)
    busy += e - s
wall = int(rows[b]["Start_Timestamp"]) - t0
print("step: %d kernels, wall %.3f ms, kernel time %.3f ms, idle %.3f ms" % (len(seg), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"][:80]
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("   %-80s %4d %9.1f us" % (k, v[0], v[1] / 1e3))
