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


# ------------------------------------------------------------------ the secant Hessian with ONE pair: textbook closed forms (a6, a7, a9)
# s = cs 1, y = cy 1, sigma = 1 (strategy sigma0):  B = sigma I - sigma^2 s s^T / (sigma s^T s) + y y^T / (s^T y) = I + (cy / cs - 1) / n 1 1^T,
# so with Dx = dx I:  (B + Dx) 1 = (cy / cs + dx) 1,  (B + Dx)^-1 1 = 1 / (cy / cs + dx) 1,  and for a vector e with sum(e) = 0:
# (B + Dx) e = (1 + dx) e.  These follow from the definition of the BFGS update the reference implements
# (hiopHessianLowRank.cpp:262-475, :495-545 solve, :925-1075 timesVec) and from nothing in this repository.
def _one_pair_inputs(n, cs, cy):
    x0, x1 = np.zeros(n), np.full(n, cs)
    g0, g1 = np.zeros(n), np.full(n, cy)
    Jc = np.zeros((1, n))          # a constant Jacobian: (J_k - J_{k-1})^T lambda = 0, y = g_1 - g_0
    Jd = np.zeros((0, n))
    return (x0, g0), (x1, g1), Jc, Jd, np.zeros(1), np.zeros(0)


@pytest.mark.parametrize("n", [64, 1000, 100003])
def test_oracle_one_pair_secant_hessian_closed_forms(n):
    cs, cy, dx = 0.5, 2.0, 1.0
    (x0, g0), (x1, g1), Jc, Jd, yc, yd = _one_pair_inputs(n, cs, cy)
    H = ho.HessianLowRank(n, l_max=6, sigma0=1.0, sigma_update_strategy="sigma0")
    assert H.update(x0, g0, Jc, Jd, yc, yd) is False and H.update(x1, g1, Jc, Jd, yc, yd) is True
    H.update_log_barrier_diagonal(np.full(n, dx))
    ones = np.ones(n)
    assert close(H.solve(ones), ones / (cy / cs + dx), n)
    y = np.zeros(n)
    H.times_vec(0.0, y, 1.0, ones)
    assert close(y, (cy / cs + dx) * ones, n)
    e = np.zeros(n); e[0], e[-1] = 1.0, -1.0
    y = np.zeros(n)
    H.times_vec(0.0, y, 1.0, e)
    assert close(y, (1.0 + dx) * e, n)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 1000, 100003])
def test_hip_one_pair_secant_hessian_closed_forms(ctx, n):
    import torch
    from hiop_amd.kkt import HessianLowRank
    cs, cy, dx = 0.5, 2.0, 1.0
    (x0, g0), (x1, g1), Jc, Jd, yc, yd = _one_pair_inputs(n, cs, cy)
    H = HessianLowRank(ctx, n, 1, 0, l_max=6, sigma0=1.0, sigma_update_strategy="sigma0")
    torch.cuda.synchronize()
    assert H.update(_D(x0), _D(g0), _D(Jc), _D(Jd), _D(yc), _D(yd)) is False
    assert H.update(_D(x1), _D(g1), _D(Jc), _D(Jd), _D(yc), _D(yd)) is True
    H.update_log_barrier_diagonal(_D(np.full(n, dx)))
    ones = np.ones(n)
    x = _D(np.zeros(n))
    torch.cuda.synchronize()
    H.solve(_D(ones), x); ctx.sync()
    assert close(x.cpu().numpy(), ones / (cy / cs + dx), n)
    y = _D(np.zeros(n))
    H.times_vec(0.0, y, 1.0, _D(ones)); ctx.sync()
    assert close(y.cpu().numpy(), (cy / cs + dx) * ones, n)
    e = np.zeros(n); e[0], e[-1] = 1.0, -1.0
    y = _D(np.zeros(n))
    H.times_vec(0.0, y, 1.0, _D(e)); ctx.sync()
    assert close(y.cpu().numpy(), (1.0 + dx) * e, n)
    H.close()


# ------------------------------------------------------------------ ... and the low-rank KKT on top of it (a10): one equality row J = c 1^T
# With q = cy / cs + dx (above):  N = J (B + Dx)^-1 J^T = c^2 n / q;   solveCompressed(rx = 1, ryc = r):
#   dyc = (J (B + Dx)^-1 rx - ryc) / N = (c n / q - r) q / (c^2 n),   dx = (B + Dx)^-1 (rx - J^T dyc) = (1 - c dyc) / q 1
# (hiopKKTLinSysLowRank::solveCompressed, hiopKKTLinSys.cpp:1110-1187: the block elimination written out for a rank-one J).
def _kkt_closed_form(n, cs, cy, dx, c, r):
    q = cy / cs + dx
    dyc = (c * n / q - r) * q / (c * c * n)
    return q, dyc, (1.0 - c * dyc) / q


@pytest.mark.parametrize("n", [64, 1000, 100003])
def test_oracle_lowrank_kkt_closed_form(n):
    cs, cy, dx, c, r = 0.5, 2.0, 1.0, 0.25, 3.0
    (x0, g0), (x1, g1), Jc, Jd, yc, yd = _one_pair_inputs(n, cs, cy)
    Jc = np.full((1, n), c)
    H = ho.HessianLowRank(n, l_max=6, sigma0=1.0, sigma_update_strategy="sigma0")
    H.update(x0, g0, Jc, Jd, yc, yd); H.update(x1, g1, Jc, Jd, yc, yd)
    K = ho.KKTLinSysLowRank(H, 1, 0)
    K.update(np.full(n, dx), np.zeros(0), Jc, Jd)
    ok, dxs, dyc, dyd = K.solve_compressed(np.ones(n), np.array([r]), np.zeros(0))
    q, dyc_w, dx_w = _kkt_closed_form(n, cs, cy, dx, c, r)
    assert ok and close(K.last_N, np.array([[c * c * n / q]]), n) and close(dyc, dyc_w, n) and close(dxs, np.full(n, dx_w), n)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 1000, 100003])
def test_hip_lowrank_kkt_closed_form(ctx, n):
    import torch
    from hiop_amd.kkt import HessianLowRank, KKTLinSysLowRank
    cs, cy, dx, c, r = 0.5, 2.0, 1.0, 0.25, 3.0
    (x0, g0), (x1, g1), Jc, Jd, yc, yd = _one_pair_inputs(n, cs, cy)
    Jc = np.full((1, n), c)
    H = HessianLowRank(ctx, n, 1, 0, l_max=6, sigma0=1.0, sigma_update_strategy="sigma0")
    Jcd, Jdd = _D(Jc), _D(Jd)
    torch.cuda.synchronize()
    H.update(_D(x0), _D(g0), Jcd, Jdd, _D(yc), _D(yd)); H.update(_D(x1), _D(g1), Jcd, Jdd, _D(yc), _D(yd))
    K = KKTLinSysLowRank(ctx, H)
    K.update_diag(_D(np.full(n, dx)), _D(np.zeros(0)), Jcd, Jdd)
    rx, dxs, dyc, dyd = _D(np.ones(n)), _D(np.zeros(n)), _D(np.zeros(1)), _D(np.zeros(0))
    torch.cuda.synchronize()
    assert K.solve_compressed(rx, _D(np.array([r])), _D(np.zeros(0)), dxs, dyc, dyd); ctx.sync()
    q, dyc_w, dx_w = _kkt_closed_form(n, cs, cy, dx, c, r)
    assert close(K.N().cpu().numpy(), np.array([[c * c * n / q]]), n)
    assert close(dyc.cpu().numpy(), dyc_w, n) and close(dxs.cpu().numpy(), np.full(n, dx_w), n)
    K.close(); H.close()
