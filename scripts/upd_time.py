"""Time the LDL^T trailing-update launches only (HIP events inside the library), no correctness check: used with the
HIOPAMD_UPD4 debug variants, whose results may be numerically meaningless."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from hiop_amd.runtime import Context
from hiop_amd.kkt import LinSolverSymDense
from hiop_amd._lib import lib
N = 8192
ctx = Context(0)
ls = LinSolverSymDense(ctx, N)
g = torch.Generator(device="cuda"); g.manual_seed(1)
M = torch.rand(N, N, generator=g, device="cuda", dtype=torch.float64) * 1e-3
M = M + M.T + torch.eye(N, device="cuda", dtype=torch.float64) * 10.0
L = lib()
h = C.c_void_p(ls.h.value)
for rep in range(4):
    ls.set_sys_matrix(M)
    if rep == 1:
        L.hiopamd_linsolver_profile(h, 1)
    ls.matrix_changed()
ums, ufl, ul = C.c_double(0), C.c_double(0), C.c_int64(0)
L.hiopamd_linsolver_profile_read(h, C.byref(ums), C.byref(ufl), C.byref(ul))
print(f"update ms per factorisation {ums.value/3:.3f}  launches {ul.value/3:.0f}  TF {ufl.value/ (ums.value*1e-3)/1e12:.2f}")
