import sys, numpy as np, torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
N = 8192
ctx = Context(0); ls = LinSolverSymDense(ctx, N)
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) * 1e-3
M = M + M.T + torch.eye(N, device="cuda", dtype=torch.float64) * 10.0
for rep in range(3):
    ls.set_sys_matrix(M); ctx.sync()
    if rep < 2:
        import os; os.environ.pop("HIOPAMD_DF_STAMPS", None)
    else:
        os.environ["HIOPAMD_DF_STAMPS"] = "1"
    ls.matrix_changed()
