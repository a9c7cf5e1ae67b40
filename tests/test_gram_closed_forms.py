"""Kernel-level known answers for the weighted Gram family built from the CONSTANT patterns of the
reference's own dense product tests (tests/LinAlg/matrixTestsDense.hpp:308 matrixTransTimesMat, :360 matrixTimesMatTrans: constant
operands, one row / column zeroed, so that every entry of the result is a closed form and a dropped or doubled term shows):

    X = c everywhere except its LAST column (zero), D = d, W = w0:
        W <- beta W + alpha X diag(D) X^T        =  beta w0 + alpha (n - 1) c^2 d                       (a4, both triangles)
        W <- W + S diag(D) X^T  (S = s, full)    =  w0 + (n - 1) s c d                                  (a5)
        W <- beta W + alpha X (sigma I + Dx)^-1 X^T with an empty secant memory
                                                 =  beta w0 + alpha (n - 1) c^2 / (sigma + dx)          (a8)

These are DERIVED closed forms on the reference's test pattern, not values the reference holds (its tests never call the three
methods of hiopHessianLowRank): they pin the oracle's restatement of hiopHessianLowRank.cpp:1079-1160, :495-633 at the kernel level
— independent of any whole-solve objective — and the HIP kernels against the same numbers.  Tolerance: the reference's isEqual
(relative 10 eps) scaled by the length of the sum (the products are exact in binary for the constants used: 1/2, 2, 3, 1/4)."""
import numpy as np
import pytest

from oracle import hiop_oracle as ho

EPS = np.finfo(np.float64).eps
SHAPES = [(3, 7), (10, 100), (16, 4096), (100, 1000), (200, 5000), (33, 100003)]   # (k rows, n columns); odd n: unaligned tails


def pattern(k, n, c):
    X = np.full((k, n), c)
    X[:, -1] = 0.0
    return X


def close(got, want, n):
    return np.all(np.abs(got - want) <= 10 * EPS * n * np.maximum(1.0, np.abs(want)))


@pytest.mark.parametrize("k,n", SHAPES)
def test_oracle_symm_gram_closed_form(k, n):
    c, d, w0, alpha, beta = 2.0, 0.5, 3.0, 0.5, 2.0
    W = np.full((k, k), w0)
    ho.symm_mat_times_diag_times_mat_trans_local(beta, W, alpha, pattern(k, n, c), np.full(n, d))
    assert close(W, beta * w0 + alpha * (n - 1) * c * c * d, n)


@pytest.mark.parametrize("k,n", SHAPES)
def test_oracle_gram_closed_form(k, n):
    c, s, d, w0, l = 2.0, 3.0, 0.25, 0.5, 6
    W = np.full((l, k), w0)
    ho.mat_times_diag_times_mat_trans_local(W, np.full((l, n), s), np.full(n, d), pattern(k, n, c))
    # the reference's method OVERWRITES W (hiopHessianLowRank.cpp:1119-1160: W = S D X^T): w0 must be gone
    assert close(W, (n - 1) * s * c * d, n)


@pytest.mark.parametrize("k,n", SHAPES)
def test_oracle_sym_mat_times_inverse_closed_form(k, n):
    c, sigma, dx, w0, alpha, beta = 2.0, 1.0, 3.0, 0.5, 2.0, 0.5
    H = ho.HessianLowRank(n, l_max=6, sigma0=sigma)
    H.update_log_barrier_diagonal(np.full(n, dx))
    W = np.full((k, k), w0)
    H.sym_mat_times_inverse_times_mat_trans(beta, W, alpha, pattern(k, n, c))
    assert close(W, beta * w0 + alpha * (n - 1) * c * c / (sigma + dx), n)


# ------------------------------------------------------------------ HIP
def _D(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float64).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", SHAPES)
def test_hip_symm_gram_closed_form(ctx, k, n):
    import torch
    c, d, w0, alpha, beta = 2.0, 0.5, 3.0, 0.5, 2.0
    X, W = _D(pattern(k, n, c)), _D(np.full((k, k), w0))
    torch.cuda.synchronize()
    ctx.call("hiopamd_gram_weighted", k, k, n, X, n, X, n, _D(np.full(n, d)), beta, W, k, alpha, 1)
    ctx.sync()
    assert close(W.cpu().numpy(), beta * w0 + alpha * (n - 1) * c * c * d, n)     # BOTH triangles, like the reference's method


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", SHAPES)
def test_hip_gram_closed_form_plain_and_stacked(ctx, k, n):
    import torch
    c, s, d, w0, l = 2.0, 3.0, 0.25, 0.5, 6
    X, S, dv = _D(pattern(k, n, c)), _D(np.full((l, n), s)), _D(np.full(n, d))
    W = _D(np.full((l, k), w0))
    torch.cuda.synchronize()
    ctx.call("hiopamd_gram_weighted", l, k, n, S, n, X, n, dv, 0.0, W, k, 1.0, 0)
    ctx.sync()
    assert close(W.cpu().numpy(), (n - 1) * s * c * d, n)
    # the one-pass form the low-rank KKT uses: X D [X; S; Y]^T with Y = -S
    Ws = _D(np.full((k, k + 2 * l), w0))
    Y = _D(np.full((l, n), -s))
    torch.cuda.synchronize()
    ctx.call("hiopamd_gram_weighted_stacked", k, n, X, n, k, X, n, l, S, n, l, Y, n, dv, 2.0, Ws, k + 2 * l, 0.5)
    ctx.sync()
    got = Ws.cpu().numpy()
    assert close(got[:, :k], 2.0 * w0 + 0.5 * (n - 1) * c * c * d, n)
    assert close(got[:, k:k + l], 2.0 * w0 + 0.5 * (n - 1) * c * s * d, n)
    assert close(got[:, k + l:], 2.0 * w0 - 0.5 * (n - 1) * c * s * d, n)


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", SHAPES)
def test_hip_sym_mat_times_inverse_closed_form(ctx, k, n):
    import torch
    from hiop_amd.kkt import HessianLowRank
    c, sigma, dx, w0, alpha, beta = 2.0, 1.0, 3.0, 0.5, 2.0, 0.5
    me = k // 2
    H = HessianLowRank(ctx, n, me, k - me, l_max=6, sigma0=sigma)
    H.update_log_barrier_diagonal(_D(np.full(n, dx)))
    W = _D(np.full((k, k), w0))
    torch.cuda.synchronize()
    H.sym_mat_times_inverse_times_mat_trans(beta, W, alpha, _D(pattern(k, n, c)))
    ctx.sync()
    assert close(W.cpu().numpy(), beta * w0 + alpha * (n - 1) * c * c / (sigma + dx), n)
    H.close()
