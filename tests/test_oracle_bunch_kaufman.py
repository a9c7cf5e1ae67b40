"""The Bunch-Kaufman oracle (oracle/bunch_kaufman.py) against the reference's own solver, LAPACK DSYTRF / DSYTRS through scipy
(hiopLinSolverSymDenseLapack.hpp:75-195): pivots and D identical, P A P^T = L D L^T, solutions, inertia rule."""
import numpy as np
import pytest
from scipy.linalg import lapack

from oracle import bunch_kaufman as bk


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def sym(r, n, scale=1.0):
    a = r.uniform(-1, 1, (n, n)) * scale
    return a + a.T


def kkt(r, nx, m):
    """[[H + Dx, J^T], [J, -Dd]] with an indefinite H: neither quasi-definite nor singular -- the case the no-pivot factor cannot
    be trusted on and the reference answers with its pivoted solver"""
    H = sym(r, nx)
    J = r.uniform(-1, 1, (m, nx))
    K = np.zeros((nx + m, nx + m))
    K[:nx, :nx] = H + np.diag(r.uniform(0, 2, nx))
    K[nx:, :nx] = J
    K[:nx, nx:] = J.T
    K[nx:, nx:] = -np.diag(r.uniform(1e-3, 1, m))
    return K


CASES = [("rand", 1), ("rand", 2), ("rand", 7), ("rand", 63), ("rand", 64), ("rand", 65), ("rand", 130), ("rand", 257),
         ("kkt", 96), ("kkt", 200), ("zero_diag", 50), ("arrow", 90), ("graded", 120)]


def make(kind, n):
    r = rng(n * 31 + len(kind))
    if kind == "rand":
        return sym(r, n)
    if kind == "kkt":
        return kkt(r, n - n // 3, n // 3)
    if kind == "zero_diag":            # every first pivot test fails: 2 x 2 pivots and interchanges everywhere
        a = sym(r, n)
        np.fill_diagonal(a, 0.0)
        return a
    if kind == "arrow":                # a border that forces far interchanges
        a = np.diag(r.uniform(1e-3, 1e-2, n))
        a[-1, :] = a[:, -1] = r.uniform(1, 2, n)
        return a
    if kind == "graded":
        s = np.logspace(0, -8, n)
        return sym(r, n) * np.outer(s, s)
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n", CASES)
@pytest.mark.parametrize("nb", [8, 64])
def test_pivots_and_D_equal_lapack(kind, n, nb):
    A = make(kind, n)
    f = bk.factor(A, nb=nb)
    ldu, ipiv, info = lapack.dsytrf(A, lower=1)
    assert info == 0 and f.info == 0
    np.testing.assert_array_equal(f.ipiv, ipiv)
    nrm = np.abs(A).max()
    np.testing.assert_allclose(f.d, np.diag(ldu), rtol=1e-9, atol=1e-12 * nrm)
    # sub-diagonals of the 2 x 2 blocks
    k = 0
    while k < n:
        if ipiv[k] < 0:
            assert f.e[k] != 0.0
            np.testing.assert_allclose(f.e[k], ldu[k + 1, k], rtol=1e-9, atol=1e-12 * nrm)
            k += 2
        else:
            assert f.e[k] == 0.0
            k += 1
    # P A P^T = L D L^T
    PAP = A[np.ix_(f.perm, f.perm)]
    R = f.L @ f.D() @ f.L.T - PAP
    growth = max(1.0, np.abs(f.L).max()) ** 2
    assert np.abs(R).max() <= 1e-13 * n * nrm * growth
    assert np.array_equal(np.triu(f.L, 1), np.zeros_like(f.L)) and np.all(np.diag(f.L) == 1.0)


@pytest.mark.parametrize("kind,n", CASES)
def test_solve_equals_dsytrs(kind, n):
    A = make(kind, n)
    r = rng(n)
    b = r.uniform(-1, 1, n)
    f = bk.factor(A)
    x = bk.solve(f, b)
    ldu, ipiv, info = lapack.dsytrf(A, lower=1)
    xl, info2 = lapack.dsytrs(ldu, ipiv, b, lower=1)
    assert info == 0 and info2 == 0
    scale = np.abs(xl).max()
    np.testing.assert_allclose(x, xl, rtol=0, atol=1e-9 * scale * max(1.0, np.linalg.cond(A) * 1e-7))
    res = np.abs(A @ x - b).max() / (np.abs(A).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= 1e-12


@pytest.mark.parametrize("kind,n", [c for c in CASES if c[0] != "graded"])
def test_inertia_rule_against_eigenvalues(kind, n):
    A = make(kind, n)
    nneg, f = bk.matrix_changed(A)
    w = np.linalg.eigvalsh(A)
    assert np.abs(w).min() > 1e-10          # (the cases are non-singular)
    assert nneg == int((w < 0).sum())
    pos, neg, null = bk.inertia(f)
    assert (pos, neg, null) == (int((w > 0).sum()), int((w < 0).sum()), 0)


def test_singular_matrix_is_reported():
    A = np.zeros((5, 5))
    A[0, 0] = 1.0
    nneg, f = bk.matrix_changed(A)
    assert nneg == -1 and f.info == 2         # DSYTRF: INFO = 2, the first exactly zero pivot
    ldu, ipiv, info = lapack.dsytrf(A, lower=1)
    assert info == f.info
    B = np.ones((4, 4))                         # rank one: a null pivot after the first step
    nneg, f = bk.matrix_changed(B)
    assert nneg == -1
