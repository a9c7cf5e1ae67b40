"""Summarise a HIOPAMD_FLOW_TRACE file: the critical path of the dataflow solve, step by step."""
import sys, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
kind, I, J, ch, t0, t1, t2 = a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5] * 0.01, a[:, 6] * 0.01, a[:, 7] * 0.01
nb = J.max() + 1
print(f"tasks {len(a)}  nb {nb}  span {t2.max():.1f} us; task start of last ticket {t0.max():.1f} us")
for k, name in ((1, "fwd"), (3, "bwd")):
    d = kind == k
    off = kind == k - 1
    print(f"-- {name}: per block step: diag start-lag (start before its input was complete, us; negative = started late), "
          f"input->signal of diag, signal(prev diag) -> last offdiag critical signal, ...")
    order = range(nb) if k == 1 else range(nb - 1, -1, -1)
    prev = None
    rows = []
    for b in order:
        m = d & (I == b)
        sig = t2[m].max()
        inp = t1[m].max()
        st = t0[m].max()
        if k == 1:
            c = off & (J == b) & (I == b - 1)
        else:
            c = off & (I == b) & (J == b + 1)
        if c.any():
            rows.append((b, sig - prev if prev is not None else 0.0, t0[c].max() - prev, t1[c].max() - prev, t2[c].max() - prev,
                         inp - prev, sig - prev, st - prev))
        prev = sig
    r = np.array(rows)
    print("  blk  step_us | crit-offdiag: start  input  signal | diag: input signal start   (all relative to previous diag signal)")
    for row in r[:: max(1, len(r) // 16)]:
        print("  %3d  %6.2f | %7.2f %6.2f %6.2f | %6.2f %6.2f %8.2f" % tuple(row))
    print(f"  mean step {r[:,1].mean():.2f} us; crit offdiag input {r[:,3].mean():.2f} signal {r[:,4].mean():.2f}; diag input {r[:,5].mean():.2f} signal {r[:,6].mean():.2f}")
