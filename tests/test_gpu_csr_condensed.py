"""Sparse condensed KKT matrix M = Jd^T diag(Hd) Jd + H + Dx + delta_wx I in CSR on the device (first
pinnable piece; reference: hiopKKTLinSysCondensedSparse::build_kkt_matrix, src/Optimization/hiopKKTLinSysSparseCondensed.cpp:205-335
and the CSR operations of src/LinAlg/hiopMatrixSparseCSR.hpp:97-290).  Checked against the DENSE product; the six CSR diagonal
kernels against numpy; then hiopPCGSolver (hiopamd_krylov_*) with a Jacobi preconditioner on that matrix through the C-level
operator callbacks.  The reference's sparse Cholesky of M is not part of this library (no oracle for it in the image)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def D(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def sparse_ex1_like(n, m, seed):
    """The shape of the reference's NlpSparseEx1 family: n variables, m inequality rows with 2-4 entries each (banded
    couplings plus a few long-range ones), a Hessian with a full diagonal and a sparse set of upper off-diagonals."""
    r = rng(seed)
    iJ, jJ = [], []
    for i in range(m):
        cols = sorted(set([i % n, (i + 1) % n] + list(r.integers(0, n, r.integers(0, 3)))))
        iJ += [i] * len(cols); jJ += cols
    iJ, jJ = np.array(iJ, np.int32), np.array(jJ, np.int32)
    vJ = r.uniform(0.5, 2.0, iJ.size) * r.choice([-1.0, 1.0], iJ.size)
    iH, jH = list(range(n)), list(range(n))
    for _ in range(n // 2):
        a, b = sorted(r.integers(0, n, 2))
        if a != b:
            iH.append(int(a)); jH.append(int(b))
    key = sorted(set(zip(iH, jH)))
    iH, jH = np.array([k[0] for k in key], np.int32), np.array([k[1] for k in key], np.int32)
    vH = np.where(iH == jH, r.uniform(1.0, 2.0, iH.size), r.uniform(-0.1, 0.1, iH.size))
    return iJ, jJ, vJ, iH, jH, vH


def dense_M(n, m, iJ, jJ, vJ, iH, jH, vH, Hd, Dx, delta):
    J = np.zeros((m, n)); np.add.at(J, (iJ, jJ), vJ)
    H = np.zeros((n, n)); np.add.at(H, (iH, jH), vH)
    H = H + np.triu(H, 1).T
    return J.T @ (Hd[:, None] * J) + H + np.diag(Dx + delta)


@pytest.mark.parametrize("n,m,seed", [(1, 1, 0), (50, 30, 1), (500, 700, 2), (3000, 2000, 3)])
def test_condensed_csr_equals_the_dense_product(ctx, n, m, seed):
    L = ctx._L
    iJ, jJ, vJ, iH, jH, vH = sparse_ex1_like(n, m, seed)
    r = rng(100 + seed)
    Hd, Dx, delta = r.uniform(0.1, 10.0, m), r.uniform(0.0, 5.0, n), 1e-3
    c = C.c_void_p()
    assert L.hiopamd_csr_condensed_create(C.byref(c), ctx.h, n, m, iJ.size, iJ.ctypes.data, jJ.ctypes.data, iH.size,
                                          iH.ctypes.data, jH.ctypes.data) == 0
    L.hiopamd_csr_condensed_nnz.restype = C.c_int64
    nnz = L.hiopamd_csr_condensed_nnz(c)
    rowptr, colidx = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32)
    assert L.hiopamd_csr_condensed_pattern(c, rowptr.ctypes.data, colidx.ctypes.data) == 0
    # CSR is well formed: sorted unique columns per row, the diagonal present
    for i in range(n):
        cols = colidx[rowptr[i]:rowptr[i + 1]]
        assert np.all(np.diff(cols) > 0) and i in cols
    vals_ptr = L.hiopamd_csr_condensed_values
    vals_ptr.restype = C.c_void_p
    for rep in range(2):      # the numeric phase is repeatable on new values (symbolic reuse)
        sc = 1.0 + rep
        keep = [D(vJ * sc), D(vH), D(Hd), D(Dx)]
        torch.cuda.synchronize()
        assert L.hiopamd_csr_condensed_numeric(c, C.c_void_p(keep[0].data_ptr()), C.c_void_p(keep[1].data_ptr()),
                                               C.c_void_p(keep[2].data_ptr()), C.c_void_p(keep[3].data_ptr()), C.c_double(delta)) == 0
        ctx.sync()
        from hiop_amd.krylov import _view
        vals = _view(vals_ptr(c), nnz).cpu().numpy()
        M = np.zeros((n, n))
        for i in range(n):
            M[i, colidx[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]
        want = dense_M(n, m, iJ, jJ, vJ * sc, iH, jH, vH, Hd, Dx, delta)
        np.testing.assert_allclose(M, want, rtol=1e-13, atol=1e-13 * np.abs(want).max())
        np.testing.assert_allclose(M, M.T, rtol=1e-15, atol=1e-15 * np.abs(want).max())   # (a/d)*b vs (b/d)*a: symmetric to rounding
    # ---- CSR kernels on (rowptr, colidx, values)
    L.hiopamd_csr_condensed_rowptr.restype = C.c_void_p
    L.hiopamd_csr_condensed_colidx.restype = C.c_void_p
    rp, ci = C.c_void_p(L.hiopamd_csr_condensed_rowptr(c)), C.c_void_p(L.hiopamd_csr_condensed_colidx(c))
    vp = C.c_void_p(vals_ptr(c))
    # (device temporaries go through ctx.call: it keeps them alive until the next synchronisation)
    x = r.uniform(-1, 1, n)
    y = D(r.uniform(-1, 1, n)); y0 = y.cpu().numpy()
    torch.cuda.synchronize()
    ctx.call("hiopamd_csr_times_vec", n, rp, ci, vp, 0.5, y, -2.0, D(x))
    ctx.sync()
    np.testing.assert_allclose(y.cpu().numpy(), 0.5 * y0 - 2.0 * (want @ x), rtol=1e-12, atol=1e-12 * np.abs(want).max())
    dg = D(np.zeros(n))
    torch.cuda.synchronize()
    ctx.call("hiopamd_csr_extract_diagonal", n, rp, ci, vp, dg)
    ctx.sync()
    np.testing.assert_allclose(dg.cpu().numpy(), np.diag(want), rtol=1e-13)
    sr, scol = r.uniform(0.5, 2.0, n), r.uniform(0.5, 2.0, n)
    srd, scd = D(sr), D(scol)
    torch.cuda.synchronize()
    ctx.call("hiopamd_csr_scale_rows", n, rp, vp, srd)
    ctx.call("hiopamd_csr_scale_cols", nnz, ci, vp, scd)
    ctx.call("hiopamd_csr_set_diagonal", n, rp, ci, vp, 7.0)
    ctx.sync()
    vals = _view(vals_ptr(c), nnz).cpu().numpy()
    M2 = np.zeros((n, n))
    for i in range(n):
        M2[i, colidx[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]
    e = sr[:, None] * want * scol[None, :]
    np.fill_diagonal(e, 7.0)
    np.testing.assert_allclose(M2, e, rtol=1e-13, atol=1e-13 * np.abs(e).max())
    # form_diag: symbolic + numeric
    drp, dci, dv = D(np.zeros(n + 1), torch.int32), D(np.zeros(n), torch.int32), D(np.zeros(n))
    torch.cuda.synchronize()
    ctx.call("hiopamd_csr_form_diag_symbolic", n, drp, dci)
    ctx.call("hiopamd_csr_form_diag_numeric", n, dv, srd)
    ctx.sync()
    assert np.array_equal(drp.cpu().numpy(), np.arange(n + 1)) and np.array_equal(dci.cpu().numpy(), np.arange(n))
    assert np.array_equal(dv.cpu().numpy(), sr)
    L.hiopamd_csr_condensed_destroy(c)


def test_pcg_with_jacobi_preconditioner_on_the_condensed_matrix(ctx):
    """hiopPCGSolver (src/LinAlg/hiopKrylovSolver.cpp:152-373) on M through the C-level callbacks hiopamd_csr_condensed_apply /
    _jacobi: converges to the dense solve; the preconditioner cuts the iteration count on a badly scaled Dx."""
    L = ctx._L
    n, m = 4000, 3000
    iJ, jJ, vJ, iH, jH, vH = sparse_ex1_like(n, m, 9)
    r = rng(10)
    Hd = r.uniform(0.1, 10.0, m)
    Dx = 10.0 ** r.uniform(-2, 4, n)          # barrier diagonals spread over six decades
    c = C.c_void_p()
    assert L.hiopamd_csr_condensed_create(C.byref(c), ctx.h, n, m, iJ.size, iJ.ctypes.data, jJ.ctypes.data, iH.size,
                                          iH.ctypes.data, jH.ctypes.data) == 0
    keep = [D(vJ), D(vH), D(Hd), D(Dx)]
    torch.cuda.synchronize()
    assert L.hiopamd_csr_condensed_numeric(c, *[C.c_void_p(t.data_ptr()) for t in keep], C.c_double(1e-8)) == 0
    want = dense_M(n, m, iJ, jJ, vJ, iH, jH, vH, Hd, Dx, 1e-8)
    b = r.uniform(-1, 1, n)
    xs = np.linalg.solve(want, b)
    from hiop_amd._lib import LINOP_FN
    # the library's own C entry points as operator callbacks (no Python in the Krylov loop)
    apply_fn = LINOP_FN(C.cast(L.hiopamd_csr_condensed_apply, C.c_void_p).value)
    jac_fn = LINOP_FN(C.cast(L.hiopamd_csr_condensed_jacobi, C.c_void_p).value)
    no_fn = C.cast(None, LINOP_FN)
    iters = {}
    for name, prec in (("none", no_fn), ("jacobi", jac_fn)):
        k = C.c_void_p()
        assert L.hiopamd_krylov_create(C.byref(k), ctx.h, 0, C.c_int64(n), apply_fn, c, prec, c, no_fn, None) == 0
        L.hiopamd_krylov_set_tol(k, C.c_double(1e-10))
        L.hiopamd_krylov_set_max_num_iter(k, 20000)
        bd = D(b)
        conv = C.c_int(0)
        torch.cuda.synchronize()
        assert L.hiopamd_krylov_solve(k, C.c_void_p(bd.data_ptr()), C.byref(conv)) == 0
        ctx.sync()
        L.hiopamd_krylov_get_sol_num_iter.restype = C.c_double
        iters[name] = L.hiopamd_krylov_get_sol_num_iter(k)
        assert conv.value == 1, (name, iters)
        x = bd.cpu().numpy()
        assert np.abs(want @ x - b).max() <= 1e-8 * np.abs(b).max()
        np.testing.assert_allclose(x, xs, rtol=1e-6, atol=1e-8 * np.abs(xs).max())
        L.hiopamd_krylov_destroy(k)
    assert iters["jacobi"] < 0.5 * iters["none"]
    L.hiopamd_csr_condensed_destroy(c)
