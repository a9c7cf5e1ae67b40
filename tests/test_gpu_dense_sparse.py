"""GPU parity of the hiopMatrixDense / hiopMatrixSparseTriplet kernels against the numpy oracle.
Mirrors tests/LinAlg/matrixTestsDense.hpp / matrixTestsSparse.hpp of the reference (one test per method),
with the reference's driver sizes (tests/testMatrixDense.cpp:168-170: M=50, K=100, N=500) plus ragged
and tall-skinny shapes of the KKT hot path.  fp64 tolerance: rtol 1e-12 on GEMV/GEMM-like sums."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hiop_oracle as ho

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def D(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def run(ctx, name, *args):
    torch.cuda.synchronize()
    ctx.call(name, *args)
    ctx.sync()


SHAPES = [(1, 1), (4, 1000), (50, 500), (7, 2049), (100, 100003), (129, 257), (300, 300), (1000, 33)]


@pytest.mark.parametrize("m,n", SHAPES)
def test_times_vec_and_trans(ctx, m, n):
    r = rng(m * 1000 + n)
    A = r.uniform(-1, 1, (m, n))
    x, y = r.uniform(-1, 1, n), r.uniform(-1, 1, m)
    Ad = D(A)
    for beta, alpha in ((0.0, 1.0), (1.0, -1.0), (0.5, 2.0)):
        yd = D(y)
        run(ctx, "hiopamd_mat_times_vec", m, n, Ad, n, beta, yd, alpha, D(x))
        e = y.copy(); ho.times_vec(A, beta, e, alpha, x)
        np.testing.assert_allclose(yd.cpu().numpy(), e, rtol=1e-12, atol=1e-12)
        xd = D(x)
        run(ctx, "hiopamd_mat_trans_times_vec", m, n, Ad, n, beta, xd, alpha, D(y))
        e = x.copy(); ho.trans_times_vec(A, beta, e, alpha, y)
        np.testing.assert_allclose(xd.cpu().numpy(), e, rtol=1e-12, atol=1e-12)


def test_times_vec_beta_zero_ignores_nan_in_y(ctx):
    A = rng(1).uniform(-1, 1, (5, 300)); x = rng(2).uniform(-1, 1, 300)
    yd = D(np.full(5, np.nan))
    run(ctx, "hiopamd_mat_times_vec", 5, 300, D(A), 300, 0.0, yd, 1.0, D(x))
    np.testing.assert_allclose(yd.cpu().numpy(), A @ x, rtol=1e-12)


def test_leading_dimension(ctx):
    big = rng(3).uniform(-1, 1, (20, 700))
    A = big[:, 100:600]  # lda = 700, odd offset -> unaligned double2 path
    Ad = D(big)
    x = rng(4).uniform(-1, 1, 500); y = rng(5).uniform(-1, 1, 20)
    torch.cuda.synchronize()
    off = C.c_void_p(Ad.data_ptr() + 100 * 8)
    yd = D(y)
    ctx.call("hiopamd_mat_times_vec", 20, 500, off, 700, 0.0, yd, 1.0, D(x)); ctx.sync()
    np.testing.assert_allclose(yd.cpu().numpy(), A @ x, rtol=1e-12)
    off = C.c_void_p(Ad.data_ptr() + 101 * 8)
    xd = D(x[:499])
    ctx.call("hiopamd_mat_trans_times_vec", 20, 499, off, 700, 0.0, xd, 1.0, D(y)); ctx.sync()
    np.testing.assert_allclose(xd.cpu().numpy(), big[:, 101:600].T @ y, rtol=1e-12)


@pytest.mark.parametrize("m,n,k", [(50, 100, 500), (12, 12, 12), (3, 200, 7), (65, 33, 129), (200, 12, 200), (300, 257, 70), (64, 64, 64),
                                   (1000, 130, 513)])   # outputs of 32 x 32 and more run on fp64 MFMA tiles (ragged tiles, K not a multiple of 16)
def test_small_gemm_family(ctx, m, n, k):
    r = rng(m + n + k)
    A = r.uniform(-1, 1, (m, n)); X = r.uniform(-1, 1, (n, k)); W = r.uniform(-1, 1, (m, k))
    Wd = D(W)
    run(ctx, "hiopamd_mat_times_mat", m, n, k, D(A), n, 0.5, Wd, k, 2.0, D(X), k)
    np.testing.assert_allclose(Wd.cpu().numpy(), 0.5 * W + 2.0 * A @ X, rtol=1e-12, atol=1e-12)
    X2 = r.uniform(-1, 1, (m, k)); W2 = r.uniform(-1, 1, (n, k))
    Wd = D(W2)
    run(ctx, "hiopamd_mat_trans_times_mat", m, n, k, D(A), n, 1.0, Wd, k, -1.0, D(X2), k)
    np.testing.assert_allclose(Wd.cpu().numpy(), W2 - A.T @ X2, rtol=1e-12, atol=1e-12)
    X3 = r.uniform(-1, 1, (k, n)); W3 = r.uniform(-1, 1, (m, k))
    Wd = D(W3)
    run(ctx, "hiopamd_mat_times_mat_trans", m, n, k, D(A), n, 0.0, Wd, k, 1.0, D(X3), n)
    np.testing.assert_allclose(Wd.cpu().numpy(), A @ X3.T, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("ma,mb,n", [(4, 6, 5000), (100, 100, 70001), (130, 12, 9999), (200, 200, 20000), (6, 6, 33)])
def test_gram_weighted_mfma(ctx, ma, mb, n):
    """fp64-MFMA weighted Gram against numpy; asymmetric operands catch a transposed fragment layout."""
    r = rng(ma * 7 + mb)
    A = r.uniform(-1, 1, (ma, n)); B = r.uniform(-1, 1, (mb, n)); d = r.uniform(0.1, 2.0, n)
    W = r.uniform(-1, 1, (ma, mb))
    Wd = D(W)
    run(ctx, "hiopamd_gram_weighted", ma, mb, n, D(A), n, D(B), n, D(d), 0.5, Wd, mb, 2.0, 0)
    e = 0.5 * W + 2.0 * (A * d) @ B.T
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-11, atol=1e-10)
    Wd = D(W)
    run(ctx, "hiopamd_gram_weighted", ma, mb, n, D(A), n, D(B), n, None, 0.0, Wd, mb, 1.0, 0)
    np.testing.assert_allclose(Wd.cpu().numpy(), A @ B.T, rtol=1e-11, atol=1e-10)
    if ma == mb:
        Ad = D(A)
        W0 = r.uniform(-1, 1, (ma, ma))
        Wd = D(W0)
        run(ctx, "hiopamd_gram_weighted", ma, ma, n, Ad, n, Ad, n, D(d), 1.0, Wd, ma, -1.0, 1)
        e = W0.copy(); ho.symm_mat_times_diag_times_mat_trans_local(1.0, e, -1.0, A, d)   # both triangles (:1079)
        got = Wd.cpu().numpy()
        np.testing.assert_allclose(got, e, rtol=1e-11, atol=1e-10)
        np.testing.assert_array_equal(got, got.T)


@pytest.mark.parametrize("k,l,n", [(200, 6, 9000), (130, 3, 4097), (256, 8, 5000), (100, 6, 3000), (64, 6, 4096), (150, 6, 5001),
                                   (176, 4, 8192), (16, 1, 2500), (112, 8, 6000), (40, 6, 2049), (208, 0, 4100)])
def test_gram_stacked_symmetric_block_is_mirrored(ctx, k, l, n):
    """X D [X; S; Y]^T in one pass: the tiles below the diagonal inside the X x X block are not computed but mirrored by the
    fold kernel (k > 128: two tile rows) — every entry, both triangles, against numpy.  The shapes walk through the variants of
    the strip kernel (4 waves x 4 / 6 / 7 / 8 tiles with two workgroups per CU, 8 waves x 8 / 11 / 13 tiles, the first form for
    14-16 tiles per wave, 16-byte and 8-byte staging for even / odd n) and the 128-tile kernel (more than 256 stacked rows)."""
    r = rng(k + l)
    X = r.uniform(-1, 1, (k, n)); S = r.uniform(-1, 1, (l, n)); Y = r.uniform(-1, 1, (l, n)); d = r.uniform(0.1, 2.0, n)
    kw = k + 2 * l
    W0 = r.uniform(-1, 1, (k, kw))
    Wd = D(W0)
    Xd = D(X)
    run(ctx, "hiopamd_gram_weighted_stacked", k, n, Xd, n, k, Xd, n, l, D(S), n, l, D(Y), n, D(d), 0.25, Wd, kw, 1.5)
    e = 0.25 * W0 + 1.5 * (X * d) @ np.vstack([X, S, Y]).T
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-11, atol=1e-10)


@pytest.mark.parametrize("l,n,ld", [(6, 50000, 50000), (1, 777, 777), (8, 4099, 4100), (3, 63, 64), (5, 1, 8)])
def test_gram_lowrank_blocks_one_pass(ctx, l, n, ld):
    """The four l x l blocks of updateInternalBFGSRepresentation (hiopHessianLowRank.cpp:400-460) in one pass, against numpy with the
    weights formed as the reference forms them (DhInv * sigma; (DhInv * sigma - 1) * sigma); symmetric blocks exactly symmetric;
    the columns between n and ld are never read (NaN-filled)."""
    r = rng(l * 1000 + n)
    S = np.full((l, ld), np.nan); Y = np.full((l, ld), np.nan)
    S[:, :n] = r.uniform(-1, 1, (l, n)); Y[:, :n] = r.uniform(-1, 1, (l, n))
    d = r.uniform(0.1, 2.0, n); sigma = 0.73
    G = D(np.full(4 * l * l, np.nan))
    run(ctx, "hiopamd_gram_lowrank_blocks", l, n, D(S), D(Y), ld, D(d), sigma, G)
    got = G.cpu().numpy().reshape(4, l, l)
    s_, y_ = S[:, :n], Y[:, :n]
    e = [(y_ * d) @ y_.T, (s_ * (d * sigma)) @ y_.T, (s_ * ((d * sigma - 1.0) * sigma)) @ s_.T, sigma * (s_ @ s_.T)]
    for q in range(4):
        np.testing.assert_allclose(got[q], e[q], rtol=1e-12, atol=1e-12 * max(1.0, n))
        if q != 1:
            np.testing.assert_array_equal(got[q], got[q].T)
    G2 = D(np.full(4 * l * l, np.nan))          # run to run: bitwise (fixed order of the partial sums)
    run(ctx, "hiopamd_gram_lowrank_blocks", l, n, D(S), D(Y), ld, D(d), sigma, G2)
    np.testing.assert_array_equal(G2.cpu().numpy(), G.cpu().numpy())


def test_assembly_kernels(ctx):
    r = rng(99)
    nW = 301
    W = r.uniform(-1, 1, (nW, nW))
    A = r.uniform(-1, 1, (37, 70))       # m x n, goes transposed into W rows [10,80) cols [100,137)
    Wd = D(W)
    run(ctx, "hiopamd_mat_trans_add_to_sym_upper", 37, 70, D(A), 70, 10, 100, 0.5, Wd, nW)
    e = W.copy(); ho.trans_add_to_sym_upper(A, 10, 100, 0.5, e)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-15, atol=1e-16)
    H = r.uniform(-1, 1, (90, 90))
    Wd = D(W)
    run(ctx, "hiopamd_mat_add_upper_to_sym_upper", 90, D(H), 90, 5, -2.0, Wd, nW)
    e = W.copy(); ho.add_upper_to_sym_upper(H, 5, -2.0, e)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-15, atol=1e-16)
    d = r.uniform(-1, 1, 200)
    Wd = D(W)
    run(ctx, "hiopamd_mat_add_sub_diagonal", Wd, nW, 7, 0.25, D(d), 13, 50)
    e = W.copy(); ho.add_sub_diagonal(e, 7, 0.25, d, 13, 50)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-15, atol=1e-16)
    Wd = D(W)
    run(ctx, "hiopamd_mat_add_sub_diagonal_const", Wd, nW, 3, 20, 1.5)
    e = W.copy(); e[np.arange(3, 23), np.arange(3, 23)] += 1.5
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-15, atol=1e-16)
    Wd = D(W)
    run(ctx, "hiopamd_mat_add_diagonal_vec", nW, Wd, nW, 2.0, D(r.uniform(-1, 1, nW) * 0 + 1))
    e = W.copy(); e[np.arange(nW), np.arange(nW)] += 2.0
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-15, atol=1e-16)
    X = r.uniform(-1, 1, (nW, nW))
    Wd = D(W)
    run(ctx, "hiopamd_mat_add_matrix", nW, nW, Wd, nW, 0.3, D(X), nW)
    np.testing.assert_allclose(Wd.cpu().numpy(), W + 0.3 * X, rtol=4e-16, atol=1e-16)
    Wd = D(W)
    run(ctx, "hiopamd_mat_symmetrize", nW, Wd, nW)
    e = np.triu(W) + np.triu(W, 1).T
    np.testing.assert_array_equal(Wd.cpu().numpy(), e)
    Wd = D(W)
    run(ctx, "hiopamd_mat_set_to_constant", nW, nW, Wd, nW, 0.125)
    assert torch.all(Wd == 0.125).item()


def test_row_ops(ctx):
    r = rng(5)
    A = r.uniform(-1, 1, (12, 1000))
    for shift in (1, -1, 3, -5):
        Ad = D(A)
        run(ctx, "hiopamd_mat_shift_rows", 12, 1000, Ad, 1000, shift)
        e = A.copy(); ho.shift_rows(e, shift)
        np.testing.assert_array_equal(Ad.cpu().numpy(), e)
    src = r.uniform(-1, 1, (5, 1000))
    Ad = D(A)
    run(ctx, "hiopamd_mat_copy_rows_from", 5, 1000, Ad, 1000, 4, D(src), 1000)
    e = A.copy(); e[4:9] = src
    np.testing.assert_array_equal(Ad.cpu().numpy(), e)
    idx = np.array([3, 0, 4], np.int32)
    Ad = D(A)
    run(ctx, "hiopamd_mat_copy_rows_from_idx", 3, 1000, Ad, 1000, D(src), 1000, D(idx, torch.int32))
    e = A.copy(); e[:3] = src[idx]
    np.testing.assert_array_equal(Ad.cpu().numpy(), e)
    torch.cuda.synchronize()
    assert ctx.reduce_double("hiopamd_mat_max_abs", 12, 1000, D(A), 1000) == np.abs(A).max()
    assert ctx.reduce_int("hiopamd_mat_is_finite", 12, 1000, D(A), 1000) == 1
    rm = D(np.zeros(12))
    run(ctx, "hiopamd_mat_row_max_abs", 12, 1000, D(A), 1000, rm)
    np.testing.assert_array_equal(rm.cpu().numpy(), np.abs(A).max(axis=1))
    sc = r.uniform(0.5, 2, 12)
    Ad = D(A)
    run(ctx, "hiopamd_mat_scale_rows", 12, 1000, Ad, 1000, D(sc), 1)
    np.testing.assert_allclose(Ad.cpu().numpy(), A * (1.0 / sc)[:, None], rtol=4e-16, atol=1e-16)


def _rand_sparse(r, m, n, density):
    M = (r.uniform(0, 1, (m, n)) < density) * r.uniform(0.5, 2.0, (m, n))
    i, j = np.nonzero(M)   # row-major order == row-sorted, columns ascending
    return i.astype(np.int32), j.astype(np.int32), M[i, j]


@pytest.mark.parametrize("m,n,dens", [(1, 1, 1.0), (40, 80, 0.1), (300, 1000, 0.01), (5, 3000, 0.5), (64, 64, 0.0)])
def test_sparse_spmv(ctx, m, n, dens):
    r = rng(m + n)
    i, j, v = _rand_sparse(r, m, n, dens)
    x, y = r.uniform(-1, 1, n), r.uniform(-1, 1, m)
    id_, jd, vd = D(i, torch.int32), D(j, torch.int32), D(v)
    for beta, alpha in ((1.0, -1.0), (0.0, 1.0), (0.5, 2.0)):
        yd = D(y)
        run(ctx, "hiopamd_sp_times_vec", m, n, v.size, id_, jd, vd, beta, yd, alpha, D(x))
        e = y.copy(); ho.sp_times_vec(m, i, j, v, beta, e, alpha, x)
        np.testing.assert_allclose(yd.cpu().numpy(), e, rtol=1e-12, atol=1e-13)
        xd = D(x)
        run(ctx, "hiopamd_sp_trans_times_vec", m, n, v.size, id_, jd, vd, beta, xd, alpha, D(y))
        e = x.copy(); ho.sp_trans_times_vec(n, i, j, v, beta, e, alpha, y)
        np.testing.assert_allclose(xd.cpu().numpy(), e, rtol=1e-12, atol=1e-13)


def test_transposed_and_symmetric_products_are_bitwise_reproducible(ctx):
    """transTimesVec (hiopMatrixSparseTriplet.cpp:110) and the symmetric product (:941-958): thousands of entries add to one output
    element.  Rounds 1-3 used fp64 atomics (the sum depended on the order in which the hardware served them); now every contribution is
    accumulated exactly (96-bit fixed point, integer atomics) and rounded once: every run gives the same bits, terms of wildly different
    magnitude and cancelling signs included, and the result is the correctly rounded sum to ~1 ulp."""
    from fractions import Fraction
    r = rng(77)
    m, n = 6000, 37                      # ~160 entries per output column, far more than one wave
    M = (r.uniform(0, 1, (m, n)) < 0.6) * r.uniform(-1, 1, (m, n)) * 10.0 ** r.integers(-12, 12, (m, n))
    i, j = np.nonzero(M)
    i, j, v = i.astype(np.int32), j.astype(np.int32), M[i, j]
    y = r.uniform(-1, 1, m) * 10.0 ** r.integers(-6, 6, m)
    x0 = r.uniform(-1, 1, n)
    id_, jd, vd, yd = D(i, torch.int32), D(j, torch.int32), D(v), D(y)
    outs = []
    for rep in range(12):
        xd = D(x0)
        run(ctx, "hiopamd_sp_trans_times_vec", m, n, v.size, id_, jd, vd, 0.5, xd, -1.5, yd)
        outs.append(xd.cpu().numpy().copy())
    for o in outs[1:]:
        assert np.array_equal(o.view(np.int64), outs[0].view(np.int64))
    # against exact rational arithmetic of the SAME rounded products (alpha * y_i * v rounded as the kernel forms it)
    for c in (0, 5, n - 1):
        k = np.nonzero(j == c)[0]
        prods = (-1.5 * y[i[k]]) * v[k]
        exact = float(sum(Fraction(float(t)) for t in prods))
        got = outs[0][c] - 0.5 * x0[c]
        assert abs(got - exact) <= 4e-16 * max(abs(exact), np.abs(prods).max() * 1e-10) + 1e-300, (c, got, exact)
    # symmetric product: upper-triangle triplets, each off-diagonal entry feeds two outputs
    ns = 900
    S = np.triu((r.uniform(0, 1, (ns, ns)) < 0.3) * r.uniform(-1, 1, (ns, ns)) * 10.0 ** r.integers(-8, 8, (ns, ns)))
    si, sj = np.nonzero(S)
    si, sj, sv = si.astype(np.int32), sj.astype(np.int32), S[si, sj]
    xs, ys = r.uniform(-1, 1, ns), r.uniform(-1, 1, ns)
    sid, sjd, svd, xsd = D(si, torch.int32), D(sj, torch.int32), D(sv), D(xs)
    souts = []
    for rep in range(8):
        yd2 = D(ys)
        run(ctx, "hiopamd_spsym_times_vec", ns, sv.size, sid, sjd, svd, 1.0, yd2, 2.0, xsd)
        souts.append(yd2.cpu().numpy().copy())
    for o in souts[1:]:
        assert np.array_equal(o.view(np.int64), souts[0].view(np.int64))
    full = S + np.triu(S, 1).T
    np.testing.assert_allclose(souts[0], ys + 2.0 * (full @ xs), rtol=1e-9, atol=1e-9 * np.abs(full).max())
    # non-finite contributions do not disappear
    v2 = v.copy(); v2[3] = np.inf
    xd = D(x0)
    run(ctx, "hiopamd_sp_trans_times_vec", m, n, v2.size, id_, jd, D(v2), 0.0, xd, 1.0, yd)
    assert not np.isfinite(xd.cpu().numpy()).all()


@pytest.mark.parametrize("m1,m2,n,dens", [(30, 30, 60, 0.15), (100, 7, 400, 0.05), (3, 3, 5000, 0.6), (50, 50, 50, 0.0)])
def test_sparse_schur_rowbuild(ctx, m1, m2, n, dens):
    """addMDinvMtransToDiagBlockOfSymDeMatUTri / addMDinvNtransToSymDeMatUTri vs the literal row-merge loop."""
    from hiop_amd._lib import lib
    L = lib()
    r = rng(m1 * 31 + m2)
    i1, j1, v1 = _rand_sparse(r, m1, n, dens)
    i2, j2, v2 = _rand_sparse(r, m2, n, dens)
    Dv = r.uniform(0.5, 3.0, n)
    nW = m1 + m2 + 9
    W = r.uniform(-1, 1, (nW, nW))
    # same-matrix diagonal block (upper triangle only)
    plan = C.c_void_p()
    rc = L.hiopamd_sp_plan_create(C.byref(plan), m1, m1, n, v1.size, i1.ctypes.data, j1.ctypes.data, v1.size,
                                  i1.ctypes.data, j1.ctypes.data, 1)
    assert rc == 0
    Wd = D(W)
    v1d, Dd = D(v1), D(Dv)
    run(ctx, "hiopamd_sp_add_MDinvNt", plan, v1d, v1d, Dd, -1.0, Wd, nW, 4, 4)
    e = W.copy(); ho.sp_add_MDinvMtrans_rowmerge(m1, i1, j1, v1, 4, -1.0, Dv, e)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-13, atol=1e-14)
    L.hiopamd_sp_plan_destroy(plan)
    # two-matrix off-diagonal block
    rc = L.hiopamd_sp_plan_create(C.byref(plan), m1, m2, n, v1.size, i1.ctypes.data, j1.ctypes.data, v2.size,
                                  i2.ctypes.data, j2.ctypes.data, 0)
    assert rc == 0
    Wd = D(W)
    run(ctx, "hiopamd_sp_add_MDinvNt", plan, v1d, D(v2), Dd, 0.5, Wd, nW, 2, 5 + m1)
    e = W.copy(); ho.sp_add_MDinvNtrans(m1, m2, n, i1, j1, v1, i2, j2, v2, 2, 5 + m1, 0.5, Dv, e)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-13, atol=1e-14)
    L.hiopamd_sp_plan_destroy(plan)


def test_sparse_plan_rejects_unsorted_rows():
    from hiop_amd._lib import lib
    L = lib()
    i = np.array([1, 0], np.int32); j = np.array([0, 0], np.int32)
    plan = C.c_void_p()
    rc = L.hiopamd_sp_plan_create(C.byref(plan), 2, 2, 1, 2, i.ctypes.data, j.ctypes.data, 2, i.ctypes.data,
                                  j.ctypes.data, 0)
    assert rc == -2


def test_symsparse_diag_ops(ctx):
    r = rng(8)
    n = 500
    i = np.arange(n, dtype=np.int32); j = i.copy()
    # add a few off-diagonal upper entries
    i = np.concatenate([i, np.array([0, 3, 10], np.int32)]); j = np.concatenate([j, np.array([5, 9, 499], np.int32)])
    o = np.lexsort((j, i)); i, j = i[o], j[o]
    v = r.uniform(-1, 1, i.size)
    y = r.uniform(-1, 1, n)
    yd = D(y)
    run(ctx, "hiopamd_spsym_add_diag_to_vec", v.size, D(i, torch.int32), D(j, torch.int32), D(v), 0.5, yd, 0, n, 0, n)
    e = y.copy(); ho.spsym_add_diag_to_vec(i, j, v, 0.5, e, 0)
    np.testing.assert_allclose(yd.cpu().numpy(), e, rtol=1e-15, atol=1e-16)
    W = r.uniform(-1, 1, (n + 3, n + 3))
    Wd = D(W)
    run(ctx, "hiopamd_spsym_add_upper_to_sym_upper", v.size, D(i, torch.int32), D(j, torch.int32), D(v), 2, -1.0, Wd, n + 3)
    e = W.copy(); np.add.at(e, (i + 2, j + 2), -v)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-15, atol=1e-16)


@pytest.mark.parametrize("m,n,dens", [(1, 1, 1.0), (40, 80, 0.1), (300, 1000, 0.01), (64, 64, 0.0), (500, 20, 0.3)])
def test_triplet_surface_outside_the_schur_build(ctx, m, n, dens):
    """transAddToSymDenseMatrixUpperTriangle :255, row_max_abs_value :285, scale_row :303, copy_to(dense) :363,
    checkIndexesAreOrdered :377, is_diagonal :1338, extract_diagonal :1355 of hiopMatrixSparseTriplet.cpp.
    Each output element receives at most one product / is a max: exact up to the fused multiply-add of W += alpha*v."""
    r = rng(7 * m + n)
    i, j, v = _rand_sparse(r, m, n, dens)
    v = v * r.choice([-1.0, 1.0], v.size)
    id_, jd, vd = D(i, torch.int32), D(j, torch.int32), D(v)
    nnz = v.size
    # W (n+m+3)^2, destination block rows [1, 1+n) x cols [n+2, n+2+m): strictly above the diagonal
    nW = n + m + 3
    W = r.uniform(-1, 1, (nW, nW))
    Wd = D(W)
    run(ctx, "hiopamd_sp_trans_add_to_sym_upper", nnz, id_, jd, vd, 1, n + 2, -0.75, Wd, nW)
    e = W.copy(); ho.sp_trans_add_to_sym_upper(i, j, v, 1, n + 2, -0.75, e)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-15, atol=5e-16)   # one rounding (fma) vs two
    # row max
    ret = D(r.uniform(5, 6, m))     # must be overwritten, also for empty rows
    run(ctx, "hiopamd_sp_row_max_abs", m, nnz, id_, vd, ret)
    assert np.array_equal(ret.cpu().numpy(), ho.sp_row_max_abs(m, i, v))
    # scale rows, both directions
    sc = r.uniform(0.5, 2.0, m)
    for inv in (0, 1):
        v2 = D(v)
        run(ctx, "hiopamd_sp_scale_rows", nnz, id_, v2, D(sc), inv)
        e = v.copy(); ho.sp_scale_rows(i, e, sc, inv)
        assert np.array_equal(v2.cpu().numpy(), e)
    # densify
    Md = D(r.uniform(-1, 1, (m, n + 2)))
    run(ctx, "hiopamd_sp_copy_to_dense", m, n, nnz, id_, jd, vd, Md, n + 2)
    assert np.array_equal(Md.cpu().numpy()[:, :n], ho.sp_copy_to_dense(m, n, i, j, v))
    # ordering / diagonal queries
    assert ctx.reduce_int("hiopamd_sp_indexes_ordered", nnz, id_, jd) == 1
    if nnz >= 2:
        assert ctx.reduce_int("hiopamd_sp_indexes_ordered", nnz, D(i[::-1].copy(), torch.int32), D(j[::-1].copy(), torch.int32)) == \
            int(ho.sp_indexes_ordered(i[::-1], j[::-1]))
    assert ctx.reduce_int64("hiopamd_sp_num_offdiag", nnz, id_, jd) == int(np.sum(i != j))
    k = min(m, n)
    dg = D(r.uniform(5, 6, k))
    inside = (i < k) & (j < k)
    run(ctx, "hiopamd_sp_extract_diagonal", k, int(inside.sum()), D(i[inside], torch.int32), D(j[inside], torch.int32),
        D(v[inside]), dg)
    assert np.array_equal(dg.cpu().numpy(), np.diag(ho.sp_copy_to_dense(m, n, i, j, v))[:k])


def test_triplet_times_mat_trans_through_the_plan(ctx):
    """hiopMatrixSparseTriplet::timesMatTrans :144-201 = scale W, then the Schur row-build with D = ones (what the HiOp-side
    adapter does)."""
    r = rng(3)
    m1, m2, n = 37, 11, 300
    i1, j1, v1 = _rand_sparse(r, m1, n, 0.08)
    i2, j2, v2 = _rand_sparse(r, m2, n, 0.2)
    L = ctx._L
    plan = C.c_void_p()
    assert L.hiopamd_sp_plan_create(C.byref(plan), m1, m2, n, v1.size, i1.ctypes.data, j1.ctypes.data, v2.size,
                                    i2.ctypes.data, j2.ctypes.data, 0) == 0
    W = r.uniform(-1, 1, (m1, m2))
    Wd = D(W)
    run(ctx, "hiopamd_vec_scale", m1 * m2, Wd, 0.5)
    run(ctx, "hiopamd_sp_add_MDinvNt", plan, D(v1), D(v2), D(np.ones(n)), -2.0, Wd, m2, 0, 0)
    e = W.copy(); ho.sp_times_mat_trans(m1, m2, n, i1, j1, v1, i2, j2, v2, 0.5, e, -2.0)
    np.testing.assert_allclose(Wd.cpu().numpy(), e, rtol=1e-13, atol=1e-14)
    L.hiopamd_sp_plan_destroy(plan)
