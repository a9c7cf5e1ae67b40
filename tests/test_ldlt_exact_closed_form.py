"""A known answer for the dominant operator that needs NO oracle (hiopLinSolverSymDense: no-pivot LDL^T + inertia +
solve, reference semantics hiopLinSolverSymDenseMagma.cpp:324-480, thresholds hiopLinSolverSymDenseLapack.hpp:154-161).

A = L D L^T is BUILT from a unit lower triangular L = I + N, N with entries in {-1, 0, 1} on a few sub-diagonals (odd rows, even columns:
N^2 = 0), and a diagonal D of signed powers of two.  Every entry of A, every Schur complement, every multiplier and every intermediate of a right-looking factorisation — blocked or
not, in any order of the additions — is then a dyadic rational of small magnitude and exactly representable: the factorisation of A must
return L and D EXACTLY (bit for bit), the inertia is the sign count of D, and A x = b for an integer x is solved exactly.  A kernel that
drops or doubles a term, mis-addresses a tile, or reads padding cannot pass; rounding cannot hide or fake anything.

CPU: the oracle's restatement of the no-pivot recurrence (oracle/hiop_oracle.py::ldlt_nopiv) reproduces the closed form — an
independent pin of that function.  GPU: the stepwise kernels (N < 768), the dataflow pair (N >= 768: 16-byte tile form for even N, 8-byte
form for odd N), ragged last blocks, and the one-launch solve with its inverted diagonal blocks, all against the same exact answer."""
import numpy as np
import pytest

from oracle import hiop_oracle as ho


def exact_case(n, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    # L = I + N with N_ij != 0 only for ODD rows i and EVEN columns j (odd offsets): N^2 = 0, so L^-1 = I - N and the inverse of every
    # principal block of L has entries in {-1, 0, 1} as well — the inverted 16 x 16 / 256 x 256 / 512 x 512 blocks the substitution and
    # the one-launch solve multiply with stay exact too (a general integer L would have inverses that grow beyond 2^53)
    L = np.eye(n)
    for off in (1, 3, 17, 65, 257, 511):      # inside a 16-block, across 16-, 64-, 256- and 512-blocks
        if off < n:
            idx = np.arange(off, n)
            idx = idx[idx % 2 == 1]
            L[idx, idx - off] = r.integers(-1, 2, idx.size).astype(np.float64)
    d = r.choice([0.5, 1.0, 2.0, 4.0], n) * r.choice([-1.0, 1.0], n)
    A = (L * d) @ L.T
    # exactness of the construction itself: every entry is a multiple of 1/2 of small magnitude
    assert np.all(A * 2 == np.round(A * 2)) and np.abs(A).max() < 64
    x = r.integers(-3, 4, n).astype(np.float64)
    return L, d, A, x, A @ x


@pytest.mark.parametrize("n", [1, 7, 64, 65, 200, 503])
def test_oracle_recurrence_returns_the_closed_form_exactly(n):
    L, d, A, x, b = exact_case(n, 100 + n)
    U, dd = ho.ldlt_nopiv(np.triu(A))
    assert np.array_equal(dd, d) and np.array_equal(U, L.T)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 55, 64, 300, 503, 504, 767, 768, 1025, 2047, 2048, 4097])
def test_hip_factor_inertia_and_solve_equal_the_closed_form_exactly(ctx, n):
    import torch
    from hiop_amd.kkt import LinSolverSymDense
    L, d, A, x, b = exact_case(n, 100 + n)
    ls = LinSolverSymDense(ctx, n)
    ls.set_sys_matrix(torch.as_tensor(np.triu(A)).cuda())
    nneg = ls.matrix_changed()
    assert nneg == int((d < 0).sum())
    F = ls.get_sys_matrix().cpu().numpy()
    assert np.array_equal(np.diag(F), d)                               # D, bit for bit
    assert np.array_equal(np.triu(F, 1), np.triu(L.T, 1))              # U = L^T strictly above the diagonal, bit for bit
    rhs = torch.as_tensor(b.copy()).cuda()
    ls.solve(rhs)
    ctx.sync()
    assert ls.solve_status()
    assert np.array_equal(rhs.cpu().numpy(), x)                        # the integer solution, exactly
    assert ls.inertia() == (int((d > 0).sum()), int((d < 0).sum()), 0)
    ls.close()
