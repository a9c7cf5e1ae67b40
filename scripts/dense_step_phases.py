"""Where a dense low-rank KKT step goes: wall time (host clock around ctx.sync) of each phase of bench.py's dense step,
and the same phases back to back without the syncs.  DENSE_N (1000000), DENSE_K (100)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hiop_amd.runtime import Context
from hiop_amd.kkt import HessianLowRank, KKTLinSysLowRank
n = int(os.environ.get("DENSE_N", "1000000")); k = int(os.environ.get("DENSE_K", "100")); l = 6
me, mi = k // 2, k - k // 2
ctx = Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
U = lambda *shape, lo=-1.0, hi=1.0: torch.rand(*shape, generator=g, device="cuda", dtype=torch.float64) * (hi - lo) + lo
J = U(me + mi, n); Jc, Jd = J[:me], J[me:]
q = U(n, lo=0.5, hi=3.0); x = U(n)
H = HessianLowRank(ctx, n, me, mi, l_max=l, sigma0=1.0, sigma_update_strategy="sty"); K = KKTLinSysLowRank(ctx, H)
Dx = U(n, lo=0.0, hi=2.0); Dd = U(mi, lo=0.5, hi=2.0)
rx0 = U(n); ryc, ryd = U(me), U(mi); rx = rx0.clone()
dx, dyc, dyd = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros_like(ryc), torch.zeros_like(ryd)
yc, yd = U(me, lo=-0.1, hi=0.1), U(mi, lo=-0.1, hi=0.1)
steps_x = [U(n, lo=-0.05, hi=0.05) for _ in range(4)]
xs = [x.clone(), x.clone()]; gg = torch.empty_like(x)
torch.cuda.synchronize()
acc = {}
def phase(name, fn, timed):
    if timed:
        ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    fn()
    if timed:
        ctx.sync(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
def step(i, timed):
    xn, xo = xs[(i + 1) & 1], xs[i & 1]
    def nlp():
        torch.add(xo, steps_x[i % 4], out=xn); torch.mul(q, xn, out=gg)
    phase("nlp stand-in", nlp, timed)
    phase("H.update", lambda: H.update(xn, gg, Jc, Jd, yc, yd), timed)
    phase("K.update_diag", lambda: K.update_diag(Dx, Dd, Jc, Jd), timed)
    for s in range(3):
        def solve():
            rx.copy_(rx0)
            assert K.solve_compressed(rx, ryc, ryd, dx, dyc, dyd)
        phase("solve %d" % s, solve, timed)
with torch.cuda.stream(ctx.torch_stream):
    for i in range(12): step(i, False)
    ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): step(12 + i, False)
    ctx.sync(); torch.cuda.synchronize(); whole = (time.perf_counter() - t0) / 20
    for i in range(20): step(32 + i, True)
print("n = %d, k = %d: step %.3f ms back to back; phases with a sync around each (ms):" % (n, k, whole * 1e3))
for name, t in acc.items(): print("  %-14s %.3f" % (name, t / 20 * 1e3))
print("  sum            %.3f" % (sum(acc.values()) / 20 * 1e3))
