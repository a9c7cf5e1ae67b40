"""GPU parity of hiopKKTLinSysCondensedSparse (BASELINE configs[4] / SURVEY section 8 row f2): the condensed sparse KKT of the
inequality-only sparse formulation, CSR assembly + PCG(Jacobi) inner solve on the device, against the oracle's restatement
(oracle/kkt_sparse.py: the same condensed matrix solved by LAPACK Cholesky — the published algorithm of the reference's
MA57 / cuSOLVER solver, which is not in the image) and against the UNcondensed XDYcYd system; then behind the full-space
layer (computeDirections, the 12-block operator, compute_directions_w_IR) against oracle/kkt_full.py.
Tolerances: componentwise backward error of the uncondensed system <= 1e-10 (PCG relative tolerance 1e-12 on a system of
condition ~1e4), directions vs the Cholesky path 1e-8 relative."""
import ctypes as C

import numpy as np
import pytest
import torch

from hiop_amd import problems as pr
from oracle import kkt_full as kf
from oracle import kkt_sparse as ks
from tests import kkt_full_cases as cases

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def D(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def pair(ctx, p):
    from hiop_amd.kkt import KKTLinSysSparseCondensed
    ko = ks.KKTLinSysCondensedSparse(p.nx, p.nineq, (p.Jd_i, p.Jd_j), (p.H_i, p.H_j))
    kg = KKTLinSysSparseCondensed(ctx, p.nx, p.nineq, p.Jd_i, p.Jd_j, p.H_i, p.H_j)
    return ko, kg


@pytest.mark.parametrize("n", [3, 10, 500, 20000, 1_000_000])
def test_condensed_sparse_solve_compressed(ctx, n):
    r = rng(n)
    p = pr.sparse_ex2_ineq(n, x=r.uniform(0.5, 2.0, n))
    ko, kg = pair(ctx, p)
    Dx, Dd = r.uniform(0, 3, n), r.uniform(0.1, 5, p.nineq)
    ko.set_values(p.Jd_v, p.H_v, Dx, Dd)
    kg.set_values(D(p.Jd_v), D(p.H_v), D(Dx), D(Dd))
    dvec = (r.uniform(0.5e-4, 1.5e-4, n), r.uniform(0.5e-6, 1.5e-6, p.nineq))
    for deltas in ((0.0, 0.0), (1e-4, 1e-6), dvec):
        ko.build_kkt_matrix(*deltas)
        if np.isscalar(deltas[0]):
            kg.build_kkt_matrix(*deltas)
        else:
            kg.build_kkt_matrix(D(deltas[0]), D(deltas[1]))
        assert kg.factorize() == 0
        rx, rd, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, p.nineq), r.uniform(-1, 1, p.nineq)
        dx, dd, dyd = D(np.zeros(n)), D(np.zeros(p.nineq)), D(np.zeros(p.nineq))
        torch.cuda.synchronize()
        assert kg.solve_compressed(D(rx), D(rd), D(ryd), dx, dd, dyd); ctx.sync()
        flag, iters, rel = kg.last_solve()
        assert flag == 0 and rel <= 1e-12
        assert iters <= 12       # M = diagonal + (dense first row/column) + rank-2 terms: a handful of Krylov directions
        dx, dd, dyd = dx.cpu().numpy(), dd.cpu().numpy(), dyd.cpu().numpy()
        assert max(ks.xdycyd_residual(ko, deltas[0], deltas[1], rx, rd, ryd, dx, dd, dyd)) < 1e-10
        if n <= 2000:            # (the oracle's Cholesky is dense)
            assert ko.factorize() == 0
            ok, dx_o, dd_o, dyd_o = ko.solve_compressed(rx, rd, ryd)
            assert ok
            for a, b in ((dx, dx_o), (dd, dd_o), (dyd, dyd_o)):
                assert np.abs(a - b).max() <= 1e-8 * max(1.0, np.abs(b).max())
    kg.close()


def test_krylov_inner_solver_stays_available(ctx, monkeypatch):
    """HIOPAMD_SPARSE_ARROW=0: the PCG + Jacobi inner solver of round 3 on the same system gives the same direction as the sparse
    direct solver (which takes over by default when the pattern is a bordered diagonal)."""
    n = 20000
    r = rng(5)
    p = pr.sparse_ex2_ineq(n, x=r.uniform(0.5, 2.0, n))
    Dx, Dd = r.uniform(0, 3, n), r.uniform(0.1, 5, p.nineq)
    rx, rd, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, p.nineq), r.uniform(-1, 1, p.nineq)
    sols = []
    for arrow in ("1", "0"):
        monkeypatch.setenv("HIOPAMD_SPARSE_ARROW", arrow)
        ko, kg = pair(ctx, p)
        kg.set_values(D(p.Jd_v), D(p.H_v), D(Dx), D(Dd))
        kg.build_kkt_matrix(1e-4, 1e-6)
        assert kg.factorize() == 0
        dx, dd, dyd = D(np.zeros(n)), D(np.zeros(p.nineq)), D(np.zeros(p.nineq))
        torch.cuda.synchronize()
        assert kg.solve_compressed(D(rx), D(rd), D(ryd), dx, dd, dyd); ctx.sync()
        flag, iters, rel = kg.last_solve()
        assert (iters == 0) == (arrow == "1")
        sols.append(dx.cpu().numpy())
        kg.close()
    assert np.abs(sols[0] - sols[1]).max() <= 1e-9 * max(1.0, np.abs(sols[0]).max())


def _bordered(r, n, border, dens):
    """random symmetric bordered-diagonal matrix as dense array + full CSR pattern (columns sorted)"""
    M = np.diag(r.uniform(1.0, 3.0, n))
    for b in border:
        col = (r.uniform(0, 1, n) < dens) * r.uniform(-0.5, 0.5, n)
        col[b] = 0.0
        M[b, :] += col
        M[:, b] += col
    M = 0.5 * (M + M.T)
    for b in border:
        M[b, b] = r.uniform(1.0, 3.0) + np.abs(M[b]).sum()
    rp, ci, v = [0], [], []
    for i in range(n):
        c = np.nonzero(M[i])[0]
        c = np.union1d(c, [i])
        ci += list(c); v += list(M[i, c]); rp.append(len(ci))
    return M, np.array(rp, np.int32), np.array(ci, np.int32), np.array(v)


@pytest.mark.parametrize("n,border,dens", [(1, [], 0.0), (50, [], 0.0), (400, [7], 1.0), (3000, [0, 1500, 2999], 0.3), (700, list(range(0, 700, 25)), 0.05)])
def test_bordered_diagonal_direct_solver(ctx, n, border, dens):
    """csrc/arrow_ldl.hip against numpy: the border found by the greedy cover, the exact inertia (negative entries in D, an
    indefinite Schur complement, a singular one), solves to rounding; the role of the reference's sparse Cholesky
    (hiopKKTLinSysSparseCondensed.cpp:469-496) for the patterns of its sparse examples."""
    from hiop_amd._lib import lib
    L = lib()
    r = rng(n + len(border))
    M, rp, ci, v = _bordered(r, n, border, dens)
    h = C.c_void_p()
    assert L.hiopamd_arrow_ldl_create(C.byref(h), ctx.h, n, rp.ctypes.data, ci.ctypes.data) == 0
    pb, bl = C.c_int(0), (C.c_int * 32)()
    assert L.hiopamd_arrow_ldl_border(h, C.byref(pb), bl) == 0
    cover = set(bl[:pb.value])
    offd = [(i, int(c)) for i in range(n) for c in ci[rp[i]:rp[i + 1]] if c != i]
    assert all(i in cover or c in cover for i, c in offd) and pb.value <= max(len(border), 1)
    vd = D(v)
    nneg, nzero = C.c_int(0), C.c_int(0)

    def fact(vals):
        torch.cuda.synchronize()
        assert L.hiopamd_arrow_ldl_factorize(h, C.c_void_p(vals.data_ptr()), C.byref(nneg), C.byref(nzero)) == 0
        return nneg.value, nzero.value

    assert fact(vd) == (0, 0)
    for rep in range(2):
        b = r.uniform(-1, 1, n)
        xd = D(b)
        torch.cuda.synchronize()
        assert L.hiopamd_arrow_ldl_solve(h, C.c_void_p(xd.data_ptr())) == 0
        ctx.sync()
        x = xd.cpu().numpy()
        assert np.abs(M @ x - b).max() <= 1e-12 * (np.abs(M).sum(1).max() * np.abs(x).max() + 1.0)
    if n >= 50:
        # inertia: flip diagonal entries of non-border variables, then make the Schur complement indefinite / singular
        nb = [i for i in range(n) if i not in cover]
        M2 = M.copy()
        for i in nb[:3]:
            M2[i, i] = -2.0
        v2 = np.concatenate([M2[i, ci[rp[i]:rp[i + 1]]] for i in range(n)])
        got = fact(D(v2))
        w = np.linalg.eigvalsh(M2)
        assert got == (int((w < 0).sum()), 0)
        if cover:
            b0 = sorted(cover)[0]
            M3 = M.copy(); M3[b0, b0] = -5.0
            v3 = np.concatenate([M3[i, ci[rp[i]:rp[i + 1]]] for i in range(n)])
            got = fact(D(v3))
            assert got == (int((np.linalg.eigvalsh(M3) < 0).sum()), 0)
            # singular: the Schur complement of the first border variable made exactly zero (p = 1 only: S is a scalar)
            if len(cover) == 1:
                d = np.array([M[i, i] for i in range(n)])
                e = M[b0].copy(); e[b0] = 0.0
                M4 = M.copy(); M4[b0, b0] = float((e * e / np.where(np.arange(n) == b0, 1.0, d)).sum())
                v4 = np.concatenate([M4[i, ci[rp[i]:rp[i + 1]]] for i in range(n)])
                g4 = fact(D(v4))
                assert g4[1] >= 1 or abs(np.linalg.eigvalsh(M4)).min() > 1e-13
                assert L.hiopamd_arrow_ldl_solve(h, C.c_void_p(D(r.uniform(-1, 1, n)).data_ptr())) != 0 or g4[1] == 0
    assert L.hiopamd_arrow_ldl_destroy(h) == 0
    # a tridiagonal pattern needs n / 2 border variables: not this solver's
    if n >= 400:
        rp2 = np.array([0] + [min(3 * i + 2, 3 * n - 2) for i in range(n)], np.int32)
        ci2 = np.concatenate([[c for c in (i - 1, i, i + 1) if 0 <= c < n] for i in range(n)]).astype(np.int32)
        rp2 = np.concatenate([[0], np.cumsum([len([c for c in (i - 1, i, i + 1) if 0 <= c < n]) for i in range(n)])]).astype(np.int32)
        h2 = C.c_void_p()
        assert L.hiopamd_arrow_ldl_create(C.byref(h2), ctx.h, n, rp2.ctypes.data, ci2.ctypes.data) == -5 and not h2.value


def test_condensed_sparse_negative_curvature_is_reported(ctx):
    """A non-convex Hessian entry: the diagonal test of factorize() or PCG's curvature test must say "not positive definite"
    (the reference's Cholesky fails, hiopKKTLinSysSparseCondensed.cpp:386-388 returns false) until delta_wx is large enough."""
    n = 400
    p = pr.sparse_ex2_ineq(n)
    ko, kg = pair(ctx, p)
    Hneg = p.H_v.copy(); Hneg[5] = -50.0
    Dx, Dd = np.zeros(n), np.full(p.nineq, 0.5)
    kg.set_values(D(p.Jd_v), D(Hneg), D(Dx), D(Dd))
    kg.build_kkt_matrix(0.0, 0.0)
    nneg = kg.factorize()
    r = rng(1)
    rx, rd, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, p.nineq), r.uniform(-1, 1, p.nineq)
    dx, dd, dyd = D(np.zeros(n)), D(np.zeros(p.nineq)), D(np.zeros(p.nineq))
    torch.cuda.synchronize()
    ok = kg.solve_compressed(D(rx), D(rd), D(ryd), dx, dd, dyd); ctx.sync()
    assert nneg == -1 or not ok
    kg.build_kkt_matrix(60.0, 0.0)
    assert kg.factorize() == 0
    torch.cuda.synchronize()
    assert kg.solve_compressed(D(rx), D(rd), D(ryd), dx, dd, dyd); ctx.sync()
    kg.close()


@pytest.mark.parametrize("n", [12, 300])
def test_full_space_layer_on_the_sparse_condensed_backend(ctx, n):
    """hiopKKTLinSysCompressedSparseXDYcYd semantics through hiopamd_kkt_xycyd_*: update (+ inertia-correction loop),
    computeDirections, the 12-block operator, compute_directions_w_IR — against oracle/kkt_full.py on the sparse provider."""
    from hiop_amd.kkt import KKTLinSysXYcYd
    r = rng(7 + n)
    p = pr.sparse_ex2_ineq(n, x=r.uniform(0.5, 2.0, n))
    ko, kg = pair(ctx, p)
    ixl = (r.uniform(0, 1, n) < 0.7).astype(np.float64); ixu = (r.uniform(0, 1, n) < 0.3).astype(np.float64)
    idl = np.ones(p.nineq); idu = (r.uniform(0, 1, p.nineq) < 0.5).astype(np.float64)
    ko.set_values(p.Jd_v, p.H_v, None, None)
    kg.set_values(D(p.Jd_v), D(p.H_v), None, None)
    fo = kf.KKTLinSysFull(ks.SparseCondensedProvider(ko), ixl, ixu, idl, idu)
    fg = KKTLinSysXYcYd(ctx, kg, D(ixl), D(ixu), D(idl), D(idu))
    it = cases.random_iterate(n, p.nineq, 0, p.nineq, ixl, ixu, idl, idu, seed=3)
    res = cases.random_resid(fo.sizes, ixl, ixu, idl, idu)
    it_g, r_g = fg.pack(it, kf.ITER_PARTS), fg.pack(res, kf.RESID_PARTS)
    fo.perturb.set_mu(1e-2); fg.set_mu(1e-2)
    assert fo.update(it) and fg.update(it_g)
    assert fg.num_refact == fo.num_refact
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert fg.compute_directions(r_g, d_g); ctx.sync()
    ok_o, d_o = fo.compute_directions(res)
    assert ok_o
    got = kf.pack(fg.unpack(d_g, kf.ITER_PARTS), kf.ITER_PARTS)
    want = kf.pack(d_o, kf.ITER_PARTS)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * np.abs(want).max())
    # K d = r on the 12-block operator (hiopKKTLinSys.cpp:1600-1617)
    y = torch.zeros_like(d_g)
    torch.cuda.synchronize()
    fg.times_vec(y, d_g); ctx.sync()
    rr = kf.pack(res, kf.RESID_PARTS)
    assert np.linalg.norm(y.cpu().numpy() - rr) <= 1e-9 * np.linalg.norm(rr)
    ok_g, info_g = fg.compute_directions_w_IR(r_g, d_g); ctx.sync()
    ok_o, d_o, info_o = fo.compute_directions_w_IR(res, mu=1e-2)
    assert ok_g and info_g["converged"] and info_o["converged"]
    got = kf.pack(fg.unpack(d_g, kf.ITER_PARTS), kf.ITER_PARTS)
    want = kf.pack(d_o, kf.ITER_PARTS)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * np.abs(want).max())
    fg.close(); kg.close()
