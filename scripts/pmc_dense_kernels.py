"""The three kernels that carry the dense low-rank step, alone, at one of bench.py's shapes (k x n Jacobian, l secant pairs): `reps`
launches each of the weighted stacked Gram, y = J x and x = J^T y through the C ABI.  Meant to run under `rocprofv3 --pmc ...` so that
the counters of a kernel belong to ONE shape (inside bench.py the GEMV kernels also run on l x n operands).
    python scripts/pmc_dense_kernels.py 200 1250000 6 [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from hiop_amd.runtime import Context

k, n, l = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = Context(0)
L = ctx._L
g = torch.Generator(device="cuda").manual_seed(1)
J = torch.rand(k * n, dtype=torch.float64, device="cuda", generator=g) - 0.5
S = torch.rand(l * n, dtype=torch.float64, device="cuda", generator=g) - 0.5
Y = torch.rand(l * n, dtype=torch.float64, device="cuda", generator=g) - 0.5
q = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) + 0.5
x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
dx = torch.zeros(n, dtype=torch.float64, device="cuda")
yk = torch.zeros(k, dtype=torch.float64, device="cuda")
kw = k + 2 * l
G = torch.zeros(k * kw, dtype=torch.float64, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
torch.cuda.synchronize()
for _ in range(reps):
    assert L.hiopamd_gram_weighted_stacked(ctx.h, k, n, p(J), n, k, p(J), n, l, p(S), n, l, p(Y), n, p(q), C.c_double(0.0), p(G), kw,
                                           C.c_double(1.0)) == 0
    assert L.hiopamd_mat_times_vec(ctx.h, k, n, p(J), n, C.c_double(0.0), p(yk), C.c_double(1.0), p(x)) == 0
    assert L.hiopamd_mat_trans_times_vec(ctx.h, k, n, p(J), n, C.c_double(0.0), p(dx), C.c_double(1.0), p(yk)) == 0
ctx.sync()
print("ok", k, n, l, reps, float(G[0]), float(yk[0]), float(dx[0]))
