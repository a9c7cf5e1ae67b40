"""Wall time of hiopamd_linsolver_matrix_changed (factorisation incl. the final inertia read-back) and of one solve."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = Context(0)
ls = LinSolverSymDense(ctx, N)
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) - 0.5
M = M + M.T
sgn = torch.ones(N, device="cuda", dtype=torch.float64); sgn[N // 2:] = -1.0
M = torch.triu(M + torch.diag(sgn * (0.6 * N ** 0.5 * 4)))
ts = []
for rep in range(6):
    ls.set_sys_matrix(M); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    nneg = ls.matrix_changed()
    ts.append(time.perf_counter() - t0)
b = torch.rand(N, generator=g, device="cuda", dtype=torch.float64)
x = b.clone(); ls.solve(x); ctx.sync()
Mf = M + torch.triu(M, 1).T
r = (Mf @ x - b).abs().max().item() / b.abs().max().item()
print(f"N={N} factor ms: " + " ".join(f"{t*1e3:.3f}" for t in ts) + f"  nneg={nneg} resid={r:.1e}")
ss = []
for rep in range(6):
    x = b.clone(); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ls.solve(x)
    ctx.sync()
    ss.append((time.perf_counter() - t0) / 3)
print(f"N={N} solve ms (of 3 in a row): " + " ".join(f"{t*1e3:.3f}" for t in ss))
