"""Time the pivoted (Bunch-Kaufman) factorisation and solve at the headline order (scripts/calls/r04_gpu_15.sh)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense

ctx = Context(0)
for n in (int(a) for a in (sys.argv[1:] or ["2048", "8192"])):
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    A = torch.rand(n, n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    A = A + A.T
    ls = LinSolverSymDense(ctx, n)
    ls.set_pivoting(True)
    best = 1e9
    for rep in range(3):
        ls.set_sys_matrix(torch.triu(A))
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        nneg = ls.matrix_changed()
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    b = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
    x = b.clone()
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ls.solve(x); ctx.sync(); ts = time.perf_counter() - t0
    res = float((A @ x - b).abs().max() / (A.abs().max() * x.abs().max() + 1.0))
    # later solves with the same factor: the second captures the sweeps as a HIP graph, the third and following replay it
    tl, xs = [], []
    for rep in range(4):
        y = b.clone()
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ls.solve(y); ctx.sync(); tl.append(time.perf_counter() - t0)
        xs.append(y)
    same = all(bool(torch.equal(x, y)) for y in xs)
    w = torch.linalg.eigvalsh(A) if n <= 4096 else None
    print("n %d: factor %.1f ms, solve %.2f ms (first, eager), later solves %s ms (identical results: %s), negative eigenvalues %d%s, residual %.2e" % (
        n, best * 1e3, ts * 1e3, " ".join("%.2f" % (t * 1e3) for t in tl), same, nneg, "" if w is None else " (eigvalsh: %d)" % int((w < 0).sum()), res), flush=True)
    ls.close()
ctx.close()
