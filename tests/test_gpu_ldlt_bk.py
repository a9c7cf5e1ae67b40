"""GPU parity of the pivoted (Bunch-Kaufman) factorisation, csrc/ldlt_bk.hip, against the oracle restatement of LAPACK DSYTRF
(oracle/bunch_kaufman.py, itself pinned on scipy's DSYTRF / DSYTRS in tests/test_oracle_bunch_kaufman.py): the same pivots (IPIV),
the same permutation, D and L to rounding, the reference's inertia rule, DSYTRS solutions -- through the C ABI, on matrices that need
2 x 2 pivots, far interchanges, several panels, and on a singular one.  Then the linear-solver object in pivoted mode on a KKT matrix
the no-pivot factorisation cannot be trusted on."""
import ctypes as C

import numpy as np
import pytest
import torch
from scipy.linalg import lapack

from oracle import bunch_kaufman as bk
from tests.test_oracle_bunch_kaufman import make, kkt, rng

pytestmark = pytest.mark.gpu


def D(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float64).cuda()


class DevBK:
    def __init__(self, ctx, n):
        from hiop_amd._lib import lib
        self.L, self.ctx, self.n = lib(), ctx, n
        self.h = C.c_void_p()
        assert self.L.hiopamd_ldlt_bk_create(C.byref(self.h), ctx.h, n) == 0

    def factor(self, A):
        self.M = D(np.triu(A))          # only the upper triangle is populated, like sysMatrix()
        ine, info = (C.c_int * 3)(), C.c_int(0)
        torch.cuda.synchronize()
        assert self.L.hiopamd_ldlt_bk_factor(self.h, C.c_void_p(self.M.data_ptr()), self.n, ine, C.byref(info)) == 0
        n = self.n
        ipiv, perm, e = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
        assert self.L.hiopamd_ldlt_bk_pivots(self.h, ipiv.ctypes.data, perm.ctypes.data, e.ctypes.data) == 0
        F = self.M.cpu().numpy()
        return dict(L=np.tril(F.T, -1) + np.eye(n), d=np.diag(F).copy(), e=e, ipiv=ipiv, perm=perm, inertia=tuple(ine), info=info.value)

    def solve(self, b):
        x = D(b)
        torch.cuda.synchronize()
        assert self.L.hiopamd_ldlt_bk_solve(self.h, C.c_void_p(self.M.data_ptr()), self.n, C.c_void_p(x.data_ptr()), 1) == 0
        self.ctx.sync()
        return x.cpu().numpy()

    def close(self):
        self.L.hiopamd_ldlt_bk_destroy(self.h)


CASES = [("rand", 1), ("rand", 2), ("rand", 7), ("rand", 63), ("rand", 64), ("rand", 65), ("rand", 130), ("rand", 257), ("rand", 700), ("rand", 2500),
         ("kkt", 96), ("kkt", 200), ("kkt", 1100), ("zero_diag", 50), ("zero_diag", 150), ("arrow", 90), ("arrow", 1500), ("graded", 120)]


@pytest.mark.parametrize("kind,n", CASES)
def test_factor_equals_oracle_and_lapack(ctx, kind, n):
    A = make(kind, n)
    fo = bk.factor(A)
    dev = DevBK(ctx, n)
    f = dev.factor(A)
    ldu, ipiv_l, info_l = lapack.dsytrf(A, lower=1)
    assert f["info"] == 0 and info_l == 0
    np.testing.assert_array_equal(f["ipiv"], ipiv_l)          # LAPACK's pivots
    np.testing.assert_array_equal(f["ipiv"], fo.ipiv)
    np.testing.assert_array_equal(f["perm"], fo.perm)
    nrm = np.abs(A).max()
    g = max(1.0, np.abs(fo.L).max())
    np.testing.assert_allclose(f["d"], fo.d, rtol=1e-9, atol=1e-11 * nrm * g * g)
    np.testing.assert_allclose(f["e"], fo.e, rtol=1e-9, atol=1e-11 * nrm * g * g)
    assert np.array_equal(f["e"] != 0.0, fo.e != 0.0)
    np.testing.assert_allclose(f["L"], fo.L, rtol=1e-8, atol=1e-10 * g * g)
    # backward error of the device factor itself
    Dm = np.diag(f["d"])
    for k in np.nonzero(f["e"])[0]:
        Dm[k + 1, k] = Dm[k, k + 1] = f["e"][k]
    R = f["L"] @ Dm @ f["L"].T - A[np.ix_(f["perm"], f["perm"])]
    assert np.abs(R).max() <= 1e-13 * n * nrm * g * g
    assert f["inertia"] == bk.inertia(fo)
    if kind != "graded":
        w = np.linalg.eigvalsh(A)
        assert f["inertia"] == (int((w > 0).sum()), int((w < 0).sum()), 0)
    # DSYTRS
    b = rng(n + 5).uniform(-1, 1, n)
    x = dev.solve(b)
    xl, _ = lapack.dsytrs(ldu, ipiv_l, b, lower=1)
    res = np.abs(A @ x - b).max() / (nrm * np.abs(x).max() + 1.0)
    assert res <= 1e-12 * max(1.0, g)
    np.testing.assert_allclose(x, xl, rtol=0, atol=1e-8 * np.abs(xl).max() * max(1.0, np.linalg.cond(A) * 1e-8))
    dev.close()


@pytest.mark.parametrize("kind,n", [("rand", 700), ("kkt", 1100), ("arrow", 1500)])
def test_one_workgroup_panel_kernel_is_the_same_factorisation(ctx, kind, n):
    """The fallback a solver object switches to after a grid barrier of the multi-workgroup panel kernel expired (advisor, round 5: the
    pivoted mode is the SAFE solver and must not be the path that can hard-fail on a busy device): one workgroup, nobody to wait for.
    Same pivots as LAPACK, same bits as the multi-workgroup form (the decisions and the arithmetic of a row do not depend on who owns it)."""
    A = make(kind, n)
    dev = DevBK(ctx, n)
    f16 = dev.factor(A)
    assert dev.L.hiopamd_ldlt_bk_set_single_workgroup(dev.h, 1) == 0
    f1 = dev.factor(A)
    _, ipiv_l, info_l = lapack.dsytrf(A, lower=1)
    assert f1["info"] == 0 and info_l == 0
    np.testing.assert_array_equal(f1["ipiv"], ipiv_l)
    np.testing.assert_array_equal(f1["perm"], f16["perm"])
    assert np.array_equal(f1["d"], f16["d"]) and np.array_equal(f1["e"], f16["e"]) and np.array_equal(f1["L"], f16["L"])
    assert f1["inertia"] == f16["inertia"]
    b = rng(n + 5).uniform(-1, 1, n)
    x = dev.solve(b)
    assert np.abs(A @ x - b).max() <= 1e-11 * n * np.abs(A).max() * max(1.0, np.abs(x).max())
    dev.close()


@pytest.mark.parametrize("n", [1024, 1300, 2500])
def test_repeated_solves_replay_a_graph_and_survive_a_new_factorisation(ctx, n):
    """from n = 1024 on the two sweeps of a solve are a HIP graph: the first solve with a matrix address runs eagerly, the second captures,
    later ones replay.  All of them must give the bits of the first; a new factorisation in the same storage (other values, other
    pivots, other inverted diagonal blocks) is picked up by the replayed graph; a matrix at ANOTHER address drops the graph."""
    A = make("rand", n)
    dev = DevBK(ctx, n)
    dev.factor(A)
    b = rng(n).uniform(-1, 1, n)
    x0 = dev.solve(b)
    assert np.abs(A @ x0 - b).max() <= 1e-11 * n * np.abs(A).max() * np.abs(x0).max()
    for _ in range(4):
        assert np.array_equal(dev.solve(b), x0)
    # same storage, new values: copy into the tensor the graph's kernels point at
    A2 = make("kkt", n)
    M = dev.M
    M.copy_(D(np.triu(A2)))
    ine, info = (C.c_int * 3)(), C.c_int(0)
    torch.cuda.synchronize()
    assert dev.L.hiopamd_ldlt_bk_factor(dev.h, C.c_void_p(M.data_ptr()), n, ine, C.byref(info)) == 0 and info.value == 0
    x2 = dev.solve(b)
    ldu, ipiv_l, _ = lapack.dsytrf(A2, lower=1)
    xl, _ = lapack.dsytrs(ldu, ipiv_l, b, lower=1)
    assert np.abs(A2 @ x2 - b).max() <= 1e-11 * n * np.abs(A2).max() * np.abs(x2).max()
    np.testing.assert_allclose(x2, xl, rtol=0, atol=1e-8 * np.abs(xl).max() * max(1.0, np.linalg.cond(A2) * 1e-8))
    assert np.array_equal(dev.solve(b), x2)
    # another address
    dev.factor(A)            # (allocates a new tensor)
    assert dev.M.data_ptr() != M.data_ptr()
    for _ in range(3):
        assert np.array_equal(dev.solve(b), x0)
    dev.close()


def test_singular_matrices(ctx):
    A = np.zeros((70, 70))
    A[0, 0] = 1.0
    dev = DevBK(ctx, 70)
    f = dev.factor(A)
    assert f["info"] == 2 and f["info"] == bk.factor(A).info
    B = np.ones((66, 66))                  # rank one: null pivots after the first step (INFO > 0 as well: exact zeros)
    dev2 = DevBK(ctx, 66)
    f2 = dev2.factor(B)
    assert f2["info"] > 0 or f2["inertia"][2] > 0
    dev.close(); dev2.close()


def test_linsolver_in_pivoted_mode(ctx):
    """hiopamd_linsolver_set_pivoting: matrixChanged returns the exact number of negative eigenvalues of a KKT matrix whose (1,1)
    block is indefinite (not quasi-definite: the static-regularisation safe mode and the no-pivot factor have no guarantee there),
    -1 for a singular matrix; solve is accurate to rounding."""
    from hiop_amd.kkt import LinSolverSymDense
    r = rng(77)
    nx, m = 300, 140
    K = kkt(r, nx, m)
    n = nx + m
    ls = LinSolverSymDense(ctx, n)
    ls.set_pivoting(True)
    ls.set_sys_matrix(D(np.triu(K)))
    nneg = ls.matrix_changed()
    w = np.linalg.eigvalsh(K)
    assert nneg == int((w < 0).sum()) and nneg != m          # (more negative eigenvalues than constraints: H is indefinite)
    b = r.uniform(-1, 1, n)
    x = D(b)
    ls.solve(x)
    ctx.sync()
    xs = x.cpu().numpy()
    assert np.abs(K @ xs - b).max() <= 1e-11 * (np.abs(K).max() * np.abs(xs).max() + 1.0)
    # singular: two equal rows / columns
    K2 = K.copy(); K2[5, :] = K2[4, :]; K2[:, 5] = K2[:, 4]; K2[5, 5] = K2[4, 4]
    ls.set_sys_matrix(D(np.triu(K2)))
    assert ls.matrix_changed() == -1
    # back to the no-pivot path: the same object still works
    ls.set_pivoting(False)
    Q = np.diag(r.uniform(1, 2, n)); Q[nx:, nx:] *= -1.0
    ls.set_sys_matrix(D(np.triu(Q)))
    assert ls.matrix_changed() == m
    ls.close()


def test_mds_kkt_safe_mode_two_is_the_pivoted_solver(ctx):
    """hiopamd_kkt_mds_set_safe_mode(k, 2): the condensed MDS system factored with Bunch-Kaufman -- same inertia verdict and the same
    directions as the oracle's LAPACK path (the reference switches solvers, not systems: hiopKKTLinSysMDS.cpp:408-430)."""
    from hiop_amd import problems as pr
    from hiop_amd._lib import lib
    from oracle import hiop_oracle as ho
    from tests.test_gpu_ldlt_kkt import _kkt_pair
    p = pr.mds_ex1_g(600, 130, 257)
    ko, kg, dv = _kkt_pair(ctx, p)
    assert lib().hiopamd_kkt_mds_set_safe_mode(kg.h, 2) == 0
    deltas = (1e-4, 1e-4, 1e-8, 1e-8)
    ko.build_kkt_matrix(*deltas)
    kg.build_kkt_matrix(*deltas)
    assert kg.factorize_with_curv_check() == ko.factorize_with_curv_check() == p.neq + p.nineq
    rx, ryc, ryd = pr.random_rhs(p)
    ok, dx_o, dyc_o, dyd_o = ko.solve_compressed(rx, ryc, ryd)
    dx, dyc, dyd = D(np.zeros_like(rx)), D(np.zeros_like(ryc)), D(np.zeros_like(ryd))
    rxd, rycd, rydd = D(rx), D(ryc), D(ryd)
    torch.cuda.synchronize()
    kg.solve_compressed(rxd, rycd, rydd, dx, dyc, dyd)
    ctx.sync()
    dx, dyc, dyd = dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy()
    res = ho.kkt_mds_full_residual(ko, deltas, rx, ryc, ryd, dx, dyc, dyd)
    assert max(res) < 1e-12, res
    scale = max(np.abs(dx_o).max(), np.abs(dyc_o).max(), np.abs(dyd_o).max())
    assert max(np.abs(dx - dx_o).max(), np.abs(dyc - dyc_o).max(), np.abs(dyd - dyd_o).max()) / scale < 1e-8
    kg.close()
