"""Summarise one step of the dense low-rank KKT bench from a rocprofv3 kernel trace (scripts/calls/r04_gpu_12.sh): kernel time by name,
idle time between kernels.  usage: dense_trace_summary.py <kernel_trace.csv> [step index of the first config]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'secant_jac' in n]
starts = idx[0::2]
nper = len(starts) // 2


def analyze(s_i, e_i, label):
    seg = rows[s_i:e_i]
    t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    print(label, 'kernels', len(seg), 'wall %.3f ms busy %.3f ms idle %.3f ms' % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        k = r['Kernel_Name'][:78]; agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print('   %-78s %4d %8.1f us' % (k, v[0], v[1] / 1e3))
    gaps = []
    for a, b in zip(seg[:-1], seg[1:]):
        gaps.append((int(b['Start_Timestamp']) - int(a['End_Timestamp']), a['Kernel_Name'][:50], b['Kernel_Name'][:50]))
    gaps.sort(reverse=True)
    for g in gaps[:8]:
        print('   gap %.1f us after %s before %s' % (g[0] / 1e3, g[1], g[2]))


analyze(starts[nper - 3] - 3, starts[nper - 2] - 3, 'first config, a late step:')
analyze(starts[2 * nper - 3] - 3, starts[2 * nper - 2] - 3, 'second config, a late step:')
