"""Time hiopamd_linsolver_solve alone and report the residual of the solution (N from argv, default 8192)."""
import sys, time, torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = Context(0)
ls = LinSolverSymDense(ctx, N)
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) - 0.5
M = M + M.T
# quasi-definite: positive block on top, negative below, as a KKT matrix has
sgn = torch.ones(N, device="cuda", dtype=torch.float64); sgn[N // 2:] = -1.0
M = M + torch.diag(sgn * (0.6 * N ** 0.5 * 4))
ls.set_sys_matrix(M)
nneg = ls.matrix_changed()
b = torch.rand(N, generator=g, device="cuda", dtype=torch.float64)
x = b.clone()
ls.solve(x); ctx.sync()
r = (M @ x - b).abs().max().item() / b.abs().max().item()
for _ in range(3): ls.solve(x)
ctx.sync()
t0 = time.perf_counter()
K = 20
for _ in range(K): ls.solve(x)
ctx.sync()
dt = (time.perf_counter() - t0) / K
x2 = b.clone(); ls.solve(x2); ctx.sync()
x3 = b.clone(); ls.solve(x3); ctx.sync()
print(f"N={N} nneg={nneg} solve {dt*1e3:.3f} ms  rel.resid {r:.2e}  reproducible={bool((x2 == x3).all())}")
