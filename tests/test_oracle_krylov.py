"""oracle/krylov.py (hiopPCGSolver / hiopBiCGStabSolver restatements) checked against independent statements of the same
mathematics on the matrix of the reference's own tests/test_pcg.cpp and tests/test_bicgstab.cpp."""
import numpy as np
import pytest

from oracle import krylov as kr


def _setup(n):
    ii, jj, vv, minv = kr.krylov_test_matrix(n)
    A = lambda x: kr.sym_times_vec(n, ii, jj, vv, x)
    M = lambda x: minv * x
    dense = np.zeros((n, n))
    dense[ii, jj] = vv
    dense = dense + np.triu(dense, 1).T
    return A, M, dense


def _textbook_pcg(A, M, b, k):
    """k steps of preconditioned CG from x = 0 (Golub & Van Loan alg. 11.5.1), written without reference to the oracle."""
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z
    for _ in range(k):
        q = A(p); a = rz / (p @ q); x = x + a * p; r = r - a * q
        z = M(r); rz_new = r @ z; p = z + (rz_new / rz) * p; rz = rz_new
    return x, np.linalg.norm(r)


def test_reference_test_matrix_is_the_one_of_test_pcg_cpp():
    ii, jj, vv, minv = kr.krylov_test_matrix(50)
    assert len(vv) == 50 + 49 + 48                                    # nnz = M + M-1 + M-2 (test_pcg.cpp:157)
    assert vv[0] == 5.0 and vv[1] == 2.0 and vv[2] == 1.0 and minv[0] == 0.2
    assert (ii <= jj).all()


@pytest.mark.parametrize("n", [50, 400])
def test_pcg_default_parameters_as_in_the_reference_test(n):
    """tests/test_pcg.cpp: rhs = 1, Jacobi preconditioner, tol 1e-9, maxit 8: does not converge in 8 iterations; the
    minimal-residual iterate comes back, and it is the textbook iterate of that index."""
    A, M, dense = _setup(n)
    b = np.ones(n)
    x, ok, flag, it, ares, rres, xk = kr.pcg(A, M, None, b)
    # the reference reports ii + 1 with ii == maxit when the loop runs out and the last iterate is the best one (:345-348)
    assert not ok and flag == 1 and 1 <= it <= 9
    xt, rt = _textbook_pcg(A, M, b, min(int(it), 8))
    np.testing.assert_allclose(x, xt, rtol=1e-12, atol=1e-14)
    assert ares == pytest.approx(np.linalg.norm(b - dense @ x), rel=1e-10)
    assert rres == pytest.approx(ares / np.sqrt(n), rel=1e-14)


def test_pcg_converges_to_the_direct_solution():
    A, M, dense = _setup(50)
    b = np.ones(50)
    x, ok, flag, it, ares, rres, xk = kr.pcg(A, M, None, b, tol=1e-12, maxit=200)
    assert ok and flag == 0 and it < 60
    np.testing.assert_allclose(x, np.linalg.solve(dense, b), rtol=1e-9)
    assert ares <= 1e-12 * np.sqrt(50) and np.array_equal(xk, x)
    # warm start from the solution: "initial guess is good enough" (:195-201), zero iterations
    x2, ok2, flag2, it2, *_ = kr.pcg(A, M, None, b, tol=1e-10, maxit=5, x0=x)
    assert ok2 and flag2 == 0 and it2 == 0.0


def test_pcg_exit_paths():
    n = 20
    A, M, dense = _setup(n)
    assert kr.pcg(A, M, None, np.zeros(n))[1:4] == (True, 0, 0.0)                     # rhs = 0
    # indefinite operator: p'Ap <= 0 -> flag 4 at the first iteration (:256-259)
    r = kr.pcg(lambda v: -A(v), None, None, np.ones(n))
    assert r[1] is False and r[2] == 4
    # right preconditioner is applied after the left one
    r1 = kr.pcg(A, M, None, np.ones(n), maxit=5)
    r2 = kr.pcg(A, None, M, np.ones(n), maxit=5)
    np.testing.assert_allclose(r1[0], r2[0], rtol=1e-14)


@pytest.mark.parametrize("n", [50, 400])
def test_bicgstab_on_the_reference_test_matrix(n):
    """tests/test_bicgstab.cpp: same matrix, rhs = 1, left Jacobi preconditioner, defaults tol 1e-9, maxit 8."""
    A, M, dense = _setup(n)
    b = np.ones(n)
    x, ok, flag, it, ares, rres = kr.bicgstab(A, M, b, 1e-9, 8)
    assert flag in (0, 1) and (it * 2) == int(it * 2)                                # whole or half iterations
    assert ares == pytest.approx(np.linalg.norm(b - dense @ x), rel=1e-8)
    x, ok, flag, it, ares, rres = kr.bicgstab(A, M, b, 1e-12, 200)
    assert ok and flag == 0
    np.testing.assert_allclose(x, np.linalg.solve(dense, b), rtol=1e-9)
    # MR(ML(v)) composition
    xa = kr.bicgstab(A, M, b, 1e-9, 6)[0]
    xb = kr.bicgstab(A, None, b, 1e-9, 6, MR=M)[0]
    np.testing.assert_allclose(xa, xb, rtol=1e-14)


def hard_system(n=40, cond=1e4, seed=0):
    """an indefinite symmetric matrix with a log-spaced spectrum: BiCGStab reaches ~1e-11 relative residual and then drifts"""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    d = np.logspace(0, np.log10(cond), n) * np.where(np.arange(n) % 3 == 0, -1, 1)
    return (Q * d) @ Q.T, rng.standard_normal(n)


def test_bicgstab_tolerance_too_small_exit_follows_the_reference():
    """hiopKrylovSolver.cpp:561-566 / :639-644: after 100 extra (half) steps that do not reach the tolerance the reference copies xk over b
    and breaks with flag 3; its closing comparison of the minimal-residual iterate (:671-688) then runs against that overwritten vector:
    ||K xmin - xk|| is of the size of ||xk||, not of a residual, so the comparison fails and the LAST iterate is returned with its own
    iteration index.  The restatement does the same by default (ref_exit=True); ref_exit=False compares against the original b and returns
    the minimal-residual iterate.  (On well-conditioned systems the exit is not reachable: the method breaks down, flag 4, or stagnates
    first — e.g. the reference's own test matrix with tol = 1e-17.)"""
    A, M, dense = _setup(50)
    assert kr.bicgstab(A, M, np.ones(50), 1e-17, 400)[2] == 4
    Amat, b = hard_system()
    op = lambda v: Amat @ v
    xr, okr, flagr, itr, aresr, _ = kr.bicgstab(op, None, b, 1e-16, 2000)
    xf, okf, flagf, itf, aresf, _ = kr.bicgstab(op, None, b, 1e-16, 2000, ref_exit=False)
    assert (okr, flagr) == (False, 3) and (okf, flagf) == (False, 3)
    assert itf < itr                                                                    # the minimal-residual iterate is an earlier one
    rr, rf = np.linalg.norm(b - Amat @ xr), np.linalg.norm(b - Amat @ xf)
    assert rf < rr < 1e-9 * np.linalg.norm(b)                                           # both accurate, the reference's choice slightly less
    assert aresr == pytest.approx(rr, rel=1e-6) and aresf == pytest.approx(rf, rel=1e-6)
