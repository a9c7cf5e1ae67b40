"""Dataflow LDL^T smoke/debug: factor one quasi-definite matrix with the dataflow kernels and with the stepwise kernels."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r = np.random.Generator(np.random.PCG64(N))
n1 = 2 * N // 3
G = r.uniform(-1, 1, (N, N)) * (1.0 / np.sqrt(N))
A = G + G.T + np.diag(np.concatenate([np.full(n1, 4.0), np.full(N - n1, -4.0)]))
ctx = Context(0)
ls = LinSolverSymDense(ctx, N)
M = torch.as_tensor(np.triu(A)).cuda()
res = {}
for mode in (False, True, True):
    ls.set_dataflow(mode)
    ls.set_sys_matrix(M)
    ctx.sync()
    t0 = time.perf_counter()
    try:
        nneg = ls.matrix_changed()
    except Exception as e:
        print("mode", mode, "FAILED:", e); sys.exit(3)
    dt = time.perf_counter() - t0
    F = np.triu(ls.get_sys_matrix().cpu().numpy())
    res.setdefault(mode, []).append(F)
    print(f"mode dataflow={mode}: nneg={nneg} ({N - n1} expected) {dt*1e3:.3f} ms, finite={np.isfinite(F).all()}")
d = np.abs(res[True][0] - res[False][0]).max() / np.abs(res[False][0]).max()
print("max rel difference dataflow vs stepwise:", d, " repeat identical:", np.array_equal(res[True][0], res[True][1]))
