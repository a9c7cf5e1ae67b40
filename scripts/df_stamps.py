"""Dataflow LDL^T at N = 8192: wall time of matrixChanged (best of 8, it synchronises to return the inertia) and, with
HIOPAMD_DF_STAMPS=1, the per-super-panel timeline the library prints."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
N = int(os.environ.get("DF_N", "8192"))
ctx = Context(0); ls = LinSolverSymDense(ctx, N)
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) * 1e-3
M = M + M.T + torch.eye(N, device="cuda", dtype=torch.float64) * 10.0
os.environ.pop("HIOPAMD_DF_STAMPS", None)
best = 1e9
for rep in range(10):
    ls.set_sys_matrix(M); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ls.matrix_changed(); dt = time.perf_counter() - t0
    if rep >= 2: best = min(best, dt)
b = torch.rand(N, generator=g, device="cuda", dtype=torch.float64)
bs = 1e9
for rep in range(10):
    xs = [b.clone() for _ in range(3)]; ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for xq in xs: ls.solve(xq)
    ctx.sync(); dt = time.perf_counter() - t0
    if rep >= 2: bs = min(bs, dt)
print("3 solves best of 8: %.3f ms" % (bs * 1e3))
x = b.clone(); ls.solve(x); ctx.sync()
res = float((M @ x - b).abs().max() / b.abs().max())
print("matrixChanged best of 8: %.3f ms  (%.1f TFLOP/s on n^3/3)  residual %.2e" % (best * 1e3, N ** 3 / 3 / best / 1e12, res))
if os.environ.get("DF_TIMELINE", "1") == "1":
    for mode in os.environ.get("DF_MODES", "1").split(","):
        os.environ["HIOPAMD_DF_STAMPS"] = mode
        ls.set_sys_matrix(M); ctx.sync()
        ls.matrix_changed()
