"""The general sparse LDL^T (csrc/sparse_ldl.hip, SURVEY section 8 row f2: the sparse-Cholesky role of
hiopKKTLinSysSparseCondensed.cpp:469-496) on the device.
  1. the solver on its own: the matrices of tests/test_sparse_ldl_plan.py (where the same plans are replayed in numpy) and large ones —
     banded n = 2e4 ... 1e6, block-arrow with borders of 33 ... 512, random fill n = 2e4, positive definite AND indefinite — checked by
     the residual of M x = b (scipy product on the host), the exact inertia (strictly diagonally dominant matrices: #negative
     eigenvalues = #negative diagonal entries), bitwise reproducibility of two factorisations;
  2. behind hiopamd_kkt_sparse_condensed on a sparse problem whose condensed matrix is BANDED (chain constraints x_i + x_{i+1} ...):
     inner_kind 'sparse_ldl', never PCG; directions against the uncondensed XDYcYd residual and (small n) the oracle's Cholesky path;
     a non-convex Hessian must be reported as "not positive definite" by factorize() alone — no right-hand side involved."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import kkt_sparse as ks
from tests.test_sparse_ldl_plan import CASES, banded, block_arrow, csr_full, quasi_definite, random_fill

pytestmark = pytest.mark.gpu


def D(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


class SparseLdl:
    def __init__(self, ctx, A):
        self.ctx, self.L = ctx, ctx._L
        self.n = A.shape[0]
        self.rp, self.ci, vals = csr_full(A)
        self.vals = D(vals)
        self.h = C.c_void_p()
        self.rc = self.L.hiopamd_sparse_ldl_create(C.byref(self.h), ctx.h, self.n, self.rp.ctypes.data, self.ci.ctypes.data)

    def info(self):
        i8 = np.zeros(8, dtype=np.int64)
        assert self.L.hiopamd_sparse_ldl_info(self.h, i8.ctypes.data) == 0
        return dict(supernodes=int(i8[0]), fronts=int(i8[1]), levels=int(i8[2]), root=int(i8[3]), nnzL=int(i8[4]), reg_levels=int(i8[5]))

    def factorize(self, vals=None):
        nneg, nzero = C.c_int(-7), C.c_int(-7)
        v = self.vals if vals is None else vals
        torch.cuda.synchronize()
        assert self.L.hiopamd_sparse_ldl_factorize(self.h, C.c_void_p(v.data_ptr()), C.byref(nneg), C.byref(nzero)) == 0
        return nneg.value, nzero.value

    def solve(self, b):
        x = D(b)
        torch.cuda.synchronize()
        rc = self.L.hiopamd_sparse_ldl_solve(self.h, C.c_void_p(x.data_ptr()))
        self.ctx.sync()
        return rc, x.cpu().numpy()

    def close(self):
        if self.h:
            self.L.hiopamd_sparse_ldl_destroy(self.h)
            self.h = C.c_void_p()


def check(ctx, A, expect_neg, tol=1e-10, max_root=None):
    S = SparseLdl(ctx, A)
    assert S.rc == 0, S.rc
    inf = S.info()
    if max_root is not None:
        assert inf["root"] <= max_root, inf
    nneg, nzero = S.factorize()
    assert (nneg, nzero) == (expect_neg, 0), (nneg, nzero, expect_neg, inf)
    b = np.random.default_rng(5).uniform(-1, 1, A.shape[0])
    rc, x = S.solve(b)
    assert rc == 0
    res = np.abs(A @ x - b).max() / max(1.0, np.abs(b).max())
    assert res <= tol, (res, inf)
    # a second factorisation and solve: the same bits (gather plans sum in a fixed order, pivots are counted with integer atomics)
    assert S.factorize() == (expect_neg, 0)
    rc, x2 = S.solve(b)
    assert rc == 0 and np.array_equal(x, x2)
    S.close()
    return inf


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_small_matrices_of_the_cpu_replay(ctx, name, make):
    A = make()
    ev = np.linalg.eigvalsh(A.toarray())
    check(ctx, A, int((ev < 0).sum()))


def n_negative_diagonal(A):
    return int((A.diagonal() < 0).sum())


@pytest.mark.parametrize("n,bw", [(20000, 3), (200000, 5), (1000000, 5), (1000000, 1), (50000, 20)])
def test_banded(ctx, n, bw):
    A = banded(n, bw, seed=n % 97)
    inf = check(ctx, A, 0, max_root=128 if bw <= 5 else None)
    assert inf["levels"] <= 4 * int(np.ceil(np.log2(n)))
    Ai = quasi_definite(A, n // 3, seed=3) if n <= 200000 else None     # (lil_matrix edits are slow beyond that)
    if Ai is not None:
        check(ctx, Ai, n_negative_diagonal(Ai))


# (orders of 30 x a power of two: the balanced dissection ends in leaves of ~25-28 columns + 2 bw rows, inside the register kernel's 40 rows)
@pytest.mark.parametrize("n,bw,indef", [(30 * 1024, 3, False), (30 * 4096, 5, False), (30 * 2048, 5, True), (28 * 1024, 2, True)])
def test_register_resident_fronts_against_the_lds_kernel(ctx, n, bw, indef, monkeypatch):
    """Round 6: the levels with small fronts and many pivots (the leaves of a banded pattern) are factored by one wave per front with the
    front in registers (sl_factor_regs_kernel); HIOPAMD_SL_REGS=0 (read when the object is created) keeps the LDS kernel of rounds 4-5
    everywhere.  Same inertia, same solution to rounding, and the register kernel is really the one in use."""
    A = banded(n, bw, seed=11)
    if indef:
        A = quasi_definite(A, n // 4, seed=5)
    b = np.random.default_rng(8).uniform(-1, 1, n)
    out = {}
    for regs in ("1", "0"):
        monkeypatch.setenv("HIOPAMD_SL_REGS", regs)
        S = SparseLdl(ctx, A)
        assert S.rc == 0
        inf = S.info()
        assert (inf["reg_levels"] >= 1) == (regs == "1"), inf
        inertia = S.factorize()
        rc, x = S.solve(b)
        assert rc == 0
        out[regs] = (inertia, x)
        S.close()
    assert out["1"][0] == out["0"][0] == ((n_negative_diagonal(A), 0) if indef else (0, 0))
    x1, x0 = out["1"][1], out["0"][1]
    assert np.abs(A @ x1 - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    assert np.abs(x1 - x0).max() <= 1e-11 * max(1.0, np.abs(x0).max())


@pytest.mark.parametrize("nblocks,bs,border", [(2000, 6, 33), (20000, 5, 128), (60000, 4, 512)])
def test_block_arrow(ctx, nblocks, bs, border):
    A = block_arrow(nblocks, bs, border, seed=border)
    inf = check(ctx, A, 0, tol=1e-9)
    assert inf["root"] <= border + 128, inf          # the border (and nothing much else) is the dense root
    if nblocks <= 20000:
        # indefinite: negate the diagonal blocks of a third of the block rows (every block stays definite: inertia = their sizes)
        A2 = sp.lil_matrix(A)
        flip = np.arange(0, nblocks, 3)
        for q in flip[:400]:
            A2[q * bs:(q + 1) * bs, q * bs:(q + 1) * bs] = -A2[q * bs:(q + 1) * bs, q * bs:(q + 1) * bs]
        A2 = A2.tocsr()
        ev_neg = min(len(flip), 400) * bs
        S = SparseLdl(ctx, A2)
        assert S.rc == 0
        nneg, nzero = S.factorize()
        # Haynsworth: inertia = inertia(blocks) + inertia(Schur complement of the border); the flipped blocks ADD to the border's Schur
        # complement (it stays positive definite), so exactly the flipped blocks' variables are negative
        assert nzero == 0 and nneg == ev_neg
        S.close()


def test_random_fill_goes_to_the_dense_root(ctx):
    n = 20000
    A = random_fill(n, 2.5, seed=1)
    inf = check(ctx, A, 0, tol=1e-9)
    assert inf["root"] > n // 2          # an expander has no small separators: this is the dense LDL^T with a sparse fringe
    Ai = quasi_definite(A, 5000, seed=2)
    check(ctx, Ai, n_negative_diagonal(Ai), tol=1e-9)


def test_zero_pivot_is_reported_and_solve_refuses(ctx):
    """row and column 100 numerically zero (structurally still there): an isolated zero pivot"""
    A = banded(3000, 2, seed=9)
    rp, ci, vals = csr_full(A)
    S = SparseLdl(ctx, A)
    assert S.rc == 0
    v = vals.copy()
    rows = np.repeat(np.arange(3000), np.diff(rp))
    v[(rows == 100) | (ci == 100)] = 0.0
    nneg, nzero = S.factorize(D(v))
    assert nzero >= 1
    rc, _ = S.solve(np.ones(3000))
    assert rc == -5                        # HIOPAMD_ERR_STATE
    assert S.factorize() == (0, 0)         # the object recovers with the next matrix
    S.close()


def test_the_recipe_of_the_sparse_solver_adapter(ctx):
    """adapters/hiopLinSolverSymSparseHipNative.cpp cannot run here (it needs libhiop); this is its sequence of C-ABI calls, restated:
    the reference's symmetric TRIPLET (one triangle, unique entries, row-sorted) -> both triangles as CSR + a gather map CSR position ->
    triplet entry (host, once) -> per factorisation: gather the values on the device, factorise, solve."""
    n = 30000
    A = quasi_definite(banded(n, 4, seed=2), n // 4, seed=5)
    U = sp.triu(A).tocoo()
    order = np.lexsort((U.col, U.row))
    ti, tj, tv = U.row[order].astype(np.int32), U.col[order].astype(np.int32), U.data[order]
    # first_call(): symmetrise, sort by (row, column), refuse duplicates
    r = np.concatenate([ti, tj[ti != tj]])
    c = np.concatenate([tj, ti[ti != tj]])
    t = np.concatenate([np.arange(ti.size), np.arange(ti.size)[ti != tj]])
    o = np.lexsort((c, r))
    r, c, t = r[o], c[o], t[o].astype(np.int32)
    assert not np.any((r[1:] == r[:-1]) & (c[1:] == c[:-1]))
    rp = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    ci = c.astype(np.int32)
    L = ctx._L
    h = C.c_void_p()
    assert L.hiopamd_sparse_ldl_create(C.byref(h), ctx.h, n, rp.ctypes.data, ci.ctypes.data) == 0
    gather = torch.as_tensor(t).cuda()
    csr_vals = torch.zeros(t.size, dtype=torch.float64, device="cuda")
    b = np.random.default_rng(1).uniform(-1, 1, n)
    for scale in (1.0, 2.5):              # two "IPM iterations": same pattern, new values
        tvals = D(tv * scale)
        torch.cuda.synchronize()
        assert L.hiopamd_vec_copy_from_indexes(ctx.h, t.size, C.c_void_p(csr_vals.data_ptr()), C.c_void_p(tvals.data_ptr()),
                                               C.c_void_p(gather.data_ptr())) == 0
        nneg, nzero = C.c_int(-7), C.c_int(-7)
        assert L.hiopamd_sparse_ldl_factorize(h, C.c_void_p(csr_vals.data_ptr()), C.byref(nneg), C.byref(nzero)) == 0
        assert (nneg.value, nzero.value) == (n_negative_diagonal(A), 0)        # matrixChanged() would return nneg
        x = D(b)
        torch.cuda.synchronize()
        assert L.hiopamd_sparse_ldl_solve(h, C.c_void_p(x.data_ptr())) == 0
        ctx.sync()
        assert np.abs(scale * (A @ x.cpu().numpy()) - b).max() <= 1e-10
    L.hiopamd_sparse_ldl_destroy(h)


# ---- behind the condensed sparse KKT -------------------------------------------------------------------------------------------------
def chain_problem(n, seed, couple=2):
    """inequalities d_i = sum_{q < couple} a_iq x_{i+q} (i = 0 .. n - couple), diagonal Hessian: M = H + Dx + Jd^T Dd Jd is banded"""
    r = np.random.default_rng(seed)
    m = n - couple + 1
    Ji = np.repeat(np.arange(m), couple).astype(np.int32)
    Jj = (Ji + np.tile(np.arange(couple), m)).astype(np.int32)
    Jv = r.uniform(0.5, 1.5, Ji.size)
    Hi = np.arange(n, dtype=np.int32)
    Hv = r.uniform(0.5, 2.0, n)
    return dict(n=n, m=m, Ji=Ji, Jj=Jj, Jv=Jv, Hi=Hi, Hj=Hi.copy(), Hv=Hv)


@pytest.mark.parametrize("n,couple", [(6000, 2), (20000, 3), (1000000, 2)])
def test_condensed_sparse_kkt_on_a_banded_pattern_uses_the_sparse_ldl(ctx, n, couple):
    from hiop_amd.kkt import KKTLinSysSparseCondensed
    p = chain_problem(n, seed=n % 89, couple=couple)
    r = np.random.default_rng(4)
    ko = ks.KKTLinSysCondensedSparse(n, p["m"], (p["Ji"], p["Jj"]), (p["Hi"], p["Hj"]))
    kg = KKTLinSysSparseCondensed(ctx, n, p["m"], p["Ji"], p["Jj"], p["Hi"], p["Hj"])
    assert kg.inner_kind() == "sparse_ldl"
    Dx, Dd = r.uniform(0, 3, n), r.uniform(0.1, 5, p["m"])
    ko.set_values(p["Jv"], p["Hv"], Dx, Dd)
    kg.set_values(D(p["Jv"]), D(p["Hv"]), D(Dx), D(Dd))
    for deltas in ((0.0, 0.0), (1e-4, 1e-6)):
        ko.build_kkt_matrix(*deltas)
        kg.build_kkt_matrix(*deltas)
        assert kg.factorize() == 0
        rx, rd, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, p["m"]), r.uniform(-1, 1, p["m"])
        dx, dd, dyd = D(np.zeros(n)), D(np.zeros(p["m"])), D(np.zeros(p["m"]))
        torch.cuda.synchronize()
        assert kg.solve_compressed(D(rx), D(rd), D(ryd), dx, dd, dyd); ctx.sync()
        flag, iters, rel = kg.last_solve()
        assert flag == 0 and iters == 0            # a direct solve: no Krylov iteration
        dx, dd, dyd = dx.cpu().numpy(), dd.cpu().numpy(), dyd.cpu().numpy()
        assert max(ks.xdycyd_residual(ko, deltas[0], deltas[1], rx, rd, ryd, dx, dd, dyd)) < 1e-10
        if n <= 6000:
            assert ko.factorize() == 0
            ok, dx_o, dd_o, dyd_o = ko.solve_compressed(rx, rd, ryd)
            assert ok
            for a, b in ((dx, dx_o), (dd, dd_o), (dyd, dyd_o)):
                assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
    # negative curvature in the Hessian: factorize() itself says "not positive definite" (-1), like a failed Cholesky
    # (hiopKKTLinSysSparseCondensed.cpp:386-388); with delta_wx large enough the verdict turns
    Hneg = p["Hv"].copy()
    Hneg[n // 2] = -50.0
    kg.set_values(D(p["Jv"]), D(Hneg), D(np.zeros(n)), D(Dd))
    kg.build_kkt_matrix(0.0, 0.0)
    assert kg.factorize() == -1
    kg.build_kkt_matrix(100.0, 0.0)
    assert kg.factorize() == 0
    kg.close()
