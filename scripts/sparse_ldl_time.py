"""Wall time of hiopamd_sparse_ldl_factorize and _solve on the banded n = 1e6 pattern of the bench entry (and the analysis time)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiop_amd.runtime import Context
from tests.test_sparse_ldl_plan import banded, csr_full

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = Context(0)
L = ctx._L
A = banded(n, bw, seed=3)
rp, ci, vals = csr_full(A)
h = C.c_void_p()
t0 = time.perf_counter()
assert L.hiopamd_sparse_ldl_create(C.byref(h), ctx.h, n, rp.ctypes.data, ci.ctypes.data) == 0
t_sym = time.perf_counter() - t0
i8 = np.zeros(8, dtype=np.int64)
L.hiopamd_sparse_ldl_info(h, i8.ctypes.data)
v = torch.as_tensor(vals).cuda()
x = torch.ones(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
nneg, nzero = C.c_int(0), C.c_int(0)
tf, ts = [], []
for rep in range(8):
    ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    assert L.hiopamd_sparse_ldl_factorize(h, C.c_void_p(v.data_ptr()), C.byref(nneg), C.byref(nzero)) == 0
    ctx.sync(); tf.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(3):
        assert L.hiopamd_sparse_ldl_solve(h, C.c_void_p(x.data_ptr())) == 0
    ctx.sync(); ts.append((time.perf_counter() - t0) / 3)
print("n %d bw %d: supernodes %d levels %d nnzL %d | analysis %.2f s | factor %.3f ms | solve %.3f ms  (best of 8)" % (
    n, bw, i8[0], i8[2], i8[4], t_sym, min(tf) * 1e3, min(ts) * 1e3))
L.hiopamd_sparse_ldl_destroy(h)
ctx.close()
