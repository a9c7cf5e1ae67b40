"""hiopMatrixSparseTriplet's assembly surface (copyRowsFrom, copyRowsBlockFrom, copySubmatrixFrom(+Trans), copySubDiagonalFrom,
setSubDiagonalTo, copyDiagMatrixToSubblock(_w_pattern), setSubmatrixToConstantDiag_w_{col,row}pattern, set_Jac_FR, set_Hess_FR).

CPU part: the oracle's restatements reproduce the closed forms the reference's own unit tests assert on the reference's test
matrices (tests/LinAlg/matrixTestsSparse.hpp:298-411, :939-965, :1070-1216, :1256-1405, :1408-1568, :1570-1678 on the pattern of
tests/LinAlg/matrixTestsSparseTriplet.cpp:303-330: M rows, `entries_per_row` entries per row at equally spaced columns, the
last one in the last column).  GPU part (-m gpu): the HIP kernels reproduce the oracle bit for bit on those and on random
row-sorted triplets (index work: exact)."""
import numpy as np
import pytest

from oracle import hiop_oracle as ho


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def ref_pattern(M, N, entries_per_row, val):
    """initializeMatrix of the reference's sparse tests (matrixTestsSparseTriplet.cpp:303-330)."""
    i, j = [], []
    for r in range(M):
        for e in range(entries_per_row - 1):
            i.append(r); j.append(e * (N // entries_per_row))
        i.append(r); j.append(N - 1)
    return np.array(i, np.int32), np.array(j, np.int32), np.full(len(i), float(val))


def dense(T, m, n, upto=None):
    W = np.zeros((m, n))
    i, j, v = T
    k = i.size if upto is None else upto
    np.add.at(W, (i[:k], j[:k]), v[:k])
    return W


def random_sorted(m, n, per_row, seed):
    r = rng(seed)
    i, j = [], []
    for row in range(m):
        cols = sorted(set(r.integers(0, n, r.integers(0, per_row + 1)).tolist()))
        i += [row] * len(cols); j += cols
    return np.array(i, np.int32), np.array(j, np.int32), r.uniform(-1, 1, len(i))


# ------------------------------------------------------------------ reference known answers (oracle)
def test_set_jac_fr_known_answer():
    """matrix_set_Jac_FR (:1570-1678): W = [C -I I 0 0; D 0 0 -I I] with C = 5 x 50 (value 2 on its pattern), D = 10 x 50."""
    M, N, M2 = 5, 50, 10
    C_, D_ = ref_pattern(M, N, 5, 2.0), ref_pattern(M2, N, 5, 2.0)
    nnz = C_[0].size + D_[0].size + 2 * M + 2 * M2
    T = (np.zeros(nnz, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz))
    assert ho.sp_set_jac_fr(T, N, C_, M, D_, M2) == nnz
    W = dense(T, M + M2, N + 2 * (M + M2))
    E = np.zeros_like(W)
    E[:M, :N] = dense(C_, M, N); E[M:, :N] = dense(D_, M2, N)
    E[:M, N:N + M] = -np.eye(M); E[:M, N + M:N + 2 * M] = np.eye(M)
    E[M:, N + 2 * M:N + 2 * M + M2] = -np.eye(M2); E[M:, N + 2 * M + M2:] = np.eye(M2)
    np.testing.assert_array_equal(W, E)
    # sorted by (row, column) like the reference leaves it
    assert np.all(np.diff(T[0].astype(np.int64) * 10**6 + T[1]) > 0)


def test_copy_subdiagonal_and_set_subdiagonal_known_answer():
    """matrix_copy_subdiagonal_from (:298-351) / matrix_set_subdiagonal_to (:353-411): the last entries of the triplet arrays are
    replaced by a diagonal block; an (i, i) that is also in the untouched pattern sums up in the dense copy."""
    M, N = 15, 80
    for use_vec in (True, False):
        T = ref_pattern(M, N, 5, 2.0)
        nnz = T[0].size
        nd = 5 if use_vec else M // 2
        if use_vec:
            ho.sp_copy_sub_diagonal_from(T, M - nd, nd, np.full(nd, 3.0), nnz - nd)
        else:
            ho.sp_set_sub_diagonal_to(T, M - nd, nd, 3.0, nnz - nd)
        W = dense(T, M, N)
        Wkept = dense(T, M, N, upto=nnz - nd)
        for i in range(M):
            for j in range(N):
                ans = 2.0 if Wkept[i, j] != 0 else 0.0
                if i == j and i >= M - nd:
                    ans += 3.0
                assert W[i, j] == ans


def test_copy_submatrix_from_known_answer():
    """matrix_copy_submatrix_from / _trans (:1070-1216): A (5 x 50, value 1 after setToConstant? the test uses two values) copied
    into the tail of a 60 x 60 matrix at (M, 2M): here checked as "the dense image is the shifted (transposed) source"."""
    M, N = 5, 50
    A = ref_pattern(M, N, 5, 3.0)
    n4 = 2 * M + N
    for trans in (False, True):
        B = ref_pattern(n4, n4, 5, 2.0)
        nnz4 = B[0].size
        end = ho.sp_copy_submatrix_from(B, A, M, 2 * M, nnz4 - A[0].size, False, trans)
        assert end == nnz4
        W = dense(B, n4, n4)
        E = dense(B, n4, n4, upto=nnz4 - A[0].size)
        Ad = dense(A, M, N)
        if trans:
            E[M:M + N, 2 * M:2 * M + M] += Ad.T
        else:
            E[M:M + M, 2 * M:2 * M + N] += Ad
        np.testing.assert_array_equal(W, E)


def test_copy_rows_known_answer():
    """matrix_copy_rows_from (:939-965): B's first M rows selected with select = 0..M-1 reproduce A when A = those rows;
    copy_rows_block_from (testMatrixSparse.cpp:179-182): row 0 of A becomes the last row of B."""
    M, N = 5, 50
    A = ref_pattern(M, N, 5, 1.0)
    B = ref_pattern(2 * M, N, 5, 2.0)
    assert ho.sp_copy_rows_from(A, B, np.arange(M)) == A[0].size
    np.testing.assert_array_equal(dense(A, M, N), dense(B, 2 * M, N)[:M])
    A = ref_pattern(M, N, 5, 1.0)
    B2 = ref_pattern(2 * M, N, 5, 2.0)
    ho.sp_copy_rows_block_from(B2, A, 0, 1, M - 1, A[0].size - 5)
    assert np.all(B2[0][A[0].size - 5:A[0].size] == M - 1) and np.all(B2[2][A[0].size - 5:A[0].size] == 1.0)


def test_hess_fr_merges_and_inserts_the_diagonal():
    H = (np.array([0, 0, 1, 2, 2], np.int32), np.array([0, 2, 2, 2, 3], np.int32), np.array([1.0, 2.0, 3.0, 4.0, 5.0]))
    add = np.array([10.0, 20.0, 30.0, 40.0])
    nnz = 4 + 3        # one diagonal per row + the three off-diagonal entries
    T = (np.zeros(nnz, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz))
    assert ho.spsym_set_hess_fr(T, H, 4, add) == nnz
    W = dense(T, 4, 4)
    E = np.diag(add).astype(float); E[0, 0] += 1.0; E[2, 2] += 4.0; E[0, 2] = 2.0; E[1, 2] = 3.0; E[2, 3] = 5.0
    np.testing.assert_array_equal(W, E)
    T0 = (np.zeros(3, np.int32), np.zeros(3, np.int32), np.zeros(3))
    empty = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    assert ho.spsym_set_hess_fr(T0, empty, 0, add[:3]) == 3
    np.testing.assert_array_equal(dense(T0, 3, 3), np.diag(add[:3]))


# ------------------------------------------------------------------ HIP vs oracle
def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _trip_dev(T):
    return [_dev(T[0].copy()), _dev(T[1].copy()), _dev(T[2].copy())]


def _eq(Tg, T):
    for a, b in zip(Tg, T):
        np.testing.assert_array_equal(a.cpu().numpy(), b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hip_assembly_kernels_equal_the_oracle(ctx, seed):
    import ctypes as C
    import torch
    from hiop_amd.runtime import dptr
    L = ctx._L
    r = rng(seed)
    m, n = [(7, 30), (200, 5000), (3000, 70000)][seed]
    S = random_sorted(m, n, 6, 10 + seed)
    nnzS = S[0].size
    Sd = _trip_dev(S)
    big = nnzS + 3 * max(m, n) + 64
    fresh = lambda: (np.full(big, -7, np.int32), np.full(big, -7, np.int32), np.full(big, -7.0))
    P = lambda t: dptr(t, ctx)

    # copySubDiagonalFrom / setSubDiagonalTo / copyDiagMatrixToSubblock
    d = r.uniform(-1, 1, m)
    T = fresh(); Tg = _trip_dev(T)
    ho.sp_copy_sub_diagonal_from(T, 3, m - 3, d, 11, 0.5)
    assert L.hiopamd_sp_copy_sub_diagonal_from(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), 3, m - 3, P(_dev(d)), 11, C.c_double(0.5)) == 0
    ho.sp_set_sub_diagonal_to(T, 1, m - 1, 2.5, 11 + m)
    assert L.hiopamd_sp_set_sub_diagonal_to(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), 1, m - 1, C.c_double(2.5), 11 + m) == 0
    ho.sp_copy_diag_matrix_to_subblock(T, -1.5, 2, 5, 11 + 2 * m, m)
    assert L.hiopamd_sp_copy_diag_matrix_to_subblock(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), C.c_double(-1.5), 2, 5, 11 + 2 * m, m) == 0
    ctx.sync(); _eq(Tg, T)

    # pattern-driven blocks
    ix = (r.uniform(0, 1, n) < 0.4).astype(np.float64)
    dx = r.uniform(-1, 1, n)
    for rowpattern in (0, 1):
        T = fresh(); Tg = _trip_dev(T)
        found = ho.sp_set_submatrix_to_constant_diag_w_pattern(T, 4.0, 3, 2, 5, ix, rowpattern)
        nf = C.c_int(-1)
        assert L.hiopamd_sp_set_submatrix_to_constant_diag_w_pattern(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), C.c_double(4.0), 3, 2, 5, n,
                                                                     P(_dev(ix)), rowpattern, C.byref(nf)) == 0
        ctx.sync(); assert nf.value == found; _eq(Tg, T)
    T = fresh(); Tg = _trip_dev(T)
    found = ho.sp_copy_diag_matrix_to_subblock_w_pattern(T, dx, 1, 4, 9, ix)
    nf = C.c_int(-1)
    assert L.hiopamd_sp_copy_diag_matrix_to_subblock_w_pattern(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), P(_dev(dx)), 1, 4, 9, n, P(_dev(ix)),
                                                               C.byref(nf)) == 0
    ctx.sync(); assert nf.value == found; _eq(Tg, T)

    # copySubmatrixFrom (+Trans), with and without the diagonal
    for trans in (0, 1):
        for off in (0, 1):
            T = fresh(); Tg = _trip_dev(T)
            ho.sp_copy_submatrix_from(T, S, 2, 3, 7, bool(off), bool(trans))
            assert L.hiopamd_sp_copy_submatrix_from(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), nnzS, P(Sd[0]), P(Sd[1]), P(Sd[2]), 2, 3, 7, off,
                                                    trans) == 0
            ctx.sync(); _eq(Tg, T)

    # copyRowsFrom / copyRowsBlockFrom
    rows = np.sort(r.choice(m, size=max(1, m // 3), replace=False)).astype(np.int32)
    T = fresh(); Tg = _trip_dev(T)
    ho.sp_copy_rows_from(T, S, rows)
    assert L.hiopamd_sp_copy_rows_from(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), nnzS, P(Sd[0]), P(Sd[1]), P(Sd[2]), P(_dev(rows)), rows.size) == 0
    ctx.sync(); _eq(Tg, T)
    T = fresh(); Tg = _trip_dev(T)
    ho.sp_copy_rows_block_from(T, S, 2, m - 4, 5, 13)
    assert L.hiopamd_sp_copy_rows_block_from(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), nnzS, P(Sd[0]), P(Sd[1]), P(Sd[2]), 2, m - 4, 5, 13) == 0
    ctx.sync(); _eq(Tg, T)

    # set_Jac_FR: structure, then values only
    m2 = m // 2 + 1
    S2 = random_sorted(m2, n, 4, 20 + seed)
    S2d = _trip_dev(S2)
    nn = nnzS + S2[0].size + 2 * m + 2 * m2
    T = (np.zeros(nn, np.int32), np.zeros(nn, np.int32), np.zeros(nn))
    Tg = _trip_dev(T); Ug = _trip_dev(T)
    assert ho.sp_set_jac_fr(T, n, S, m, S2, m2) == nn
    assert L.hiopamd_sp_set_jac_fr(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), n, m, nnzS, P(Sd[0]), P(Sd[1]), P(Sd[2]), m2, S2[0].size, P(S2d[0]),
                                   P(S2d[1]), P(S2d[2]), P(Ug[0]), P(Ug[1]), P(Ug[2])) == 0
    ctx.sync(); _eq(Tg, T); _eq(Ug, T)
    Vg = _trip_dev((np.zeros(nn, np.int32), np.zeros(nn, np.int32), np.zeros(nn)))
    Wv = torch.zeros(nn, dtype=torch.float64, device="cuda")
    assert L.hiopamd_sp_set_jac_fr(ctx.h, P(Vg[0]), P(Vg[1]), P(Vg[2]), n, m, nnzS, P(Sd[0]), P(Sd[1]), P(Sd[2]), m2, S2[0].size, P(S2d[0]),
                                   P(S2d[1]), P(S2d[2]), None, None, P(Wv)) == 0
    ctx.sync()
    np.testing.assert_array_equal(Vg[2].cpu().numpy(), T[2]); np.testing.assert_array_equal(Wv.cpu().numpy(), T[2])
    assert not Vg[0].any() and not Vg[1].any()          # no index arrays given: indices untouched (:838 `if(iJacS != nullptr ...)`)

    # set_Hess_FR on an upper-triangle Hessian (some rows with, some without a diagonal entry) and on an empty one
    mh = min(m, n)
    Hraw = random_sorted(mh, mh, 5, 30 + seed)
    keep = Hraw[1] >= Hraw[0]
    H = (Hraw[0][keep], Hraw[1][keep], Hraw[2][keep])
    Hd = _trip_dev(H)
    add = r.uniform(0.5, 2.0, mh)
    noff = int(np.sum(H[0] != H[1]))
    nh = mh + noff
    T = (np.zeros(nh, np.int32), np.zeros(nh, np.int32), np.zeros(nh))
    Tg = _trip_dev(T); Ug = _trip_dev(T)
    assert ho.spsym_set_hess_fr(T, H, mh, add) == nh
    assert L.hiopamd_spsym_set_hess_fr(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), mh, H[0].size, P(Hd[0]), P(Hd[1]), P(Hd[2]), mh, P(_dev(add)),
                                       P(Ug[0]), P(Ug[1]), P(Ug[2])) == 0
    ctx.sync(); _eq(Tg, T); _eq(Ug, T)
    T = (np.zeros(mh, np.int32), np.zeros(mh, np.int32), np.zeros(mh))
    Tg = _trip_dev(T); Ug = _trip_dev(T)
    empty = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    ho.spsym_set_hess_fr(T, empty, 0, add)
    assert L.hiopamd_spsym_set_hess_fr(ctx.h, P(Tg[0]), P(Tg[1]), P(Tg[2]), 0, 0, None, None, None, mh, P(_dev(add)), P(Ug[0]), P(Ug[1]),
                                       P(Ug[2])) == 0
    ctx.sync(); _eq(Tg, T)
