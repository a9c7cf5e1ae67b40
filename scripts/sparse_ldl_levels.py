"""One factorisation + one solve of the general sparse LDL^T on the banded n = 1e6 pattern of the bench entry: meant to run under
`rocprofv3 --kernel-trace --output-format csv` so that the per-level kernel durations can be read off (scripts/calls/r05_gpu_40.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiop_amd.runtime import Context
from tests.test_sparse_ldl_plan import banded, csr_full

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = Context(0)
L = ctx._L
A = banded(n, 5, seed=3)
rp, ci, vals = csr_full(A)
h = C.c_void_p()
assert L.hiopamd_sparse_ldl_create(C.byref(h), ctx.h, n, rp.ctypes.data, ci.ctypes.data) == 0
i8 = np.zeros(8, dtype=np.int64)
L.hiopamd_sparse_ldl_info(h, i8.ctypes.data)
print("supernodes %d fronts %d levels %d root %d nnzL %d" % tuple(int(v) for v in i8[:5]), flush=True)
v = torch.as_tensor(vals).cuda()
x = torch.ones(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for rep in range(2):   # (the second round is the one to read: first-touch effects are in the first)
    nneg, nzero = C.c_int(0), C.c_int(0)
    assert L.hiopamd_sparse_ldl_factorize(h, C.c_void_p(v.data_ptr()), C.byref(nneg), C.byref(nzero)) == 0
    assert L.hiopamd_sparse_ldl_solve(h, C.c_void_p(x.data_ptr())) == 0
    ctx.sync()
L.hiopamd_sparse_ldl_destroy(h)
ctx.close()
