"""Soak of the pivoted panel kernel's device protocol (rendezvous on one XCD, tagged granules, one grid barrier per column): the same
matrix factorised again and again — every factorisation must return status 0 (an expired wait is HIOPAMD_ERR_TIMEOUT) and the SAME
pivots and factor, bit for bit, as the first one (a stale or torn cross-workgroup read would change a decision or a value).
    python scripts/bk_repeat.py n reps [n reps ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiop_amd.runtime import Context
from hiop_amd._lib import lib

ctx = Context(0)
L = lib()
args = [int(a) for a in (sys.argv[1:] or ["2048", "500"])]
for n, reps in zip(args[0::2], args[1::2]):
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    A = torch.rand(n, n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    A = torch.triu(A + A.T).contiguous()
    h = C.c_void_p()
    assert L.hiopamd_ldlt_bk_create(C.byref(h), ctx.h, n) == 0
    M = torch.empty_like(A)
    ref = None
    bad = 0
    t0 = time.perf_counter()
    for rep in range(reps):
        M.copy_(A)
        torch.cuda.synchronize()
        ine, info = (C.c_int * 3)(), C.c_int(0)
        rc = L.hiopamd_ldlt_bk_factor(h, C.c_void_p(M.data_ptr()), n, ine, C.byref(info))
        if rc != 0:
            print("factorisation %d of order %d: status %d" % (rep, n, rc), flush=True)
            bad += 1
            continue
        ipiv, perm, e = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
        assert L.hiopamd_ldlt_bk_pivots(h, ipiv.ctypes.data, perm.ctypes.data, e.ctypes.data) == 0
        if ref is None:
            ref = (ipiv, e, M.clone(), tuple(ine))
        else:
            same = np.array_equal(ipiv, ref[0]) and np.array_equal(e, ref[1]) and bool(torch.equal(M, ref[2])) and tuple(ine) == ref[3]
            if not same:
                print("factorisation %d of order %d differs from the first one" % (rep, n), flush=True)
                bad += 1
    dt = time.perf_counter() - t0
    print("%d pivoted factorisations of order %d: %d failures, every other one bitwise equal to the first (%.1f ms each incl. the copy and the read-back)" % (
        reps, n, bad, 1e3 * dt / reps), flush=True)
    L.hiopamd_ldlt_bk_destroy(h)
    if bad:
        sys.exit(1)
ctx.close()
