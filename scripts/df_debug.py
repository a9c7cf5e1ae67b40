"""Dataflow vs stepwise factorisation on the same matrix: where do they differ?  (bring-up aid; run under `timeout`)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
N = int(os.environ.get("DF_N", "1024"))
ctx = Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) * 1e-3
M = M + M.T + torch.eye(N, device="cuda", dtype=torch.float64) * 10.0
M[2 * N // 3:, 2 * N // 3:] -= torch.eye(N - 2 * N // 3, device="cuda", dtype=torch.float64) * 20.0
M = torch.triu(M).contiguous()
out = {}
for df in (0, 1):
    ls = LinSolverSymDense(ctx, N)
    ls.set_dataflow(bool(df))
    ls.set_sys_matrix(M); ctx.sync()
    nneg = ls.matrix_changed()
    out[df] = (nneg, ls.get_sys_matrix().cpu().numpy())
    ls.close()
F0, F1 = out[0][1], out[1][1]
print("N", N, "nneg stepwise/dataflow", out[0][0], out[1][0], "finite", np.isfinite(F1).all())
E = np.abs(np.triu(F1) - np.triu(F0)) / (1e-300 + np.abs(np.triu(F0)).max())
print("max rel diff", E.max())
T = 128
nt = (N + T - 1) // T
for I in range(nt):
    print(" ".join("%8.1e" % E[I * T:(I + 1) * T, J * T:(J + 1) * T].max() if J >= I else "        " for J in range(nt)))
bad = np.argwhere(E > 1e-9)
if bad.size:
    print("first bad entries (row, col, dataflow, stepwise):")
    for r, c in bad[:12]:
        print(r, c, F1[r, c], F0[r, c])
    I, J = bad[0][0] // T, bad[0][1] // T
    sub = E[I * T:(I + 1) * T, J * T:(J + 1) * T] > 1e-9
    print("tile", I, J, "bad rows", np.nonzero(sub.any(axis=1))[0][:40], "bad cols", np.nonzero(sub.any(axis=0))[0][:40])
