"""Full-size parity of the single-GPU configurations the bench reports (BASELINE configs[1] and the per-GPU shard of configs[3]).

The oracle (numpy) covers this path up to n ~ 1e5; at n = 1e6 / 1.25e6 the checks are the size-independent properties of the
domain, computed with plain torch fp64 operations that share nothing with the library:
  * the direction returned by hiopKKTLinSysLowRank::solveCompressed (hiopKKTLinSys.cpp:1110-1190) satisfies the UNCONDENSED
    XYcYd system  [B + Dx, Jc^T, Jd^T; Jc, 0, 0; Jd, 0, -Dd^-1] [dx; dyc; dyd] = [rx; ryc; ryd]  with B the compact
    Byrd-Nocedal-Schnabel matrix rebuilt here from the library's secant pairs (S, Y, sigma);
  * the reduced k x k system matrix the library formed equals J (B + Dx)^-1 J^T + diag(0, Dd^-1) applied through the same
    independent operator (one column probe);
  * solves with the cached N are BIT-IDENTICAL to solves that rebuild it;
  * the secant update's fused Jacobian pass leaves the stored previous Jacobians equal to the current ones and contributes
    nothing when the Jacobian did not change (linear constraints).
Tolerances: componentwise backward error <= 1e-9 (fp64, n = 1e6 sums, cond(N) ~ 1e4); bit identity where stated."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "C2 n=1e6 m=100": (1_000_000, 50, 50),
    "C4 shard n_local=1.25e6 m=200": (1_250_000, 100, 100),
}


def _bns_times(S, Y, sigma, v):
    """B v for the compact representation B = sigma I - [sigma S, Y] [[sigma S^T S, L], [L^T, -D]]^-1 [sigma S^T; Y^T]
    (Byrd, Nocedal, Schnabel 1994, eq. 3.5 — the matrix hiopHessianLowRank represents, hiopHessianLowRank.hpp:70-95);
    S, Y: l x n (rows = pairs, oldest first)."""
    l = S.shape[0]
    if l == 0:
        return sigma * v
    StS = S @ S.T
    SY = S @ Y.T
    Lm = torch.tril(SY, -1)
    Dm = torch.diag(torch.diagonal(SY))
    M = torch.cat([torch.cat([sigma * StS, Lm], 1), torch.cat([Lm.T, -Dm], 1)], 0)
    w = torch.cat([sigma * (S @ v), Y @ v])
    z = torch.linalg.solve(M, w)
    return sigma * v - (sigma * (S.T @ z[:l]) + Y.T @ z[l:])


@pytest.mark.parametrize("case", list(CASES))
def test_dense_lowrank_full_size_properties(ctx, case):
    from hiop_amd.kkt import HessianLowRank, KKTLinSysLowRank
    n, me, mi = CASES[case]
    l = 6
    g = torch.Generator(device="cuda"); g.manual_seed(n % 9973)
    U = lambda *shape, lo=-1.0, hi=1.0: torch.rand(*shape, generator=g, device="cuda", dtype=torch.float64) * (hi - lo) + lo
    J = U(me + mi, n)
    Jc, Jd = J[:me], J[me:]
    q = U(n, lo=0.5, hi=3.0)
    x = U(n)
    H = HessianLowRank(ctx, n, me, mi, l_max=l, sigma0=1.0, sigma_update_strategy="sty")
    K = KKTLinSysLowRank(ctx, H)
    yc, yd = U(me, lo=-0.1, hi=0.1), U(mi, lo=-0.1, hi=0.1)
    torch.cuda.synchronize()
    stored = 0
    for it in range(9):                      # fills the secant memory and shifts it twice
        gx = q * x
        torch.cuda.synchronize()
        stored += int(H.update(x, gx, Jc, Jd, yc, yd)); ctx.sync()
        x = x + U(n, lo=-0.05, hi=0.05)
    assert H.l_curr == l and stored >= l
    Dx, Dd = U(n, lo=0.0, hi=2.0), U(mi, lo=0.5, hi=2.0)
    rx0, ryc, ryd = U(n), U(me), U(mi)
    dx, dyc, dyd = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros_like(ryc), torch.zeros_like(ryd)
    torch.cuda.synchronize()
    K.update_diag(Dx, Dd, Jc, Jd)
    rx = rx0.clone(); torch.cuda.synchronize()
    assert K.solve_compressed(rx, ryc, ryd, dx, dyc, dyd); ctx.sync()

    # ---- (1) backward error of the uncondensed system, everything recomputed with torch
    S, Y, sigma = H.St(), H.Yt(), H.sigma
    Bdx = _bns_times(S, Y, sigma, dx) + Dx * dx
    r1 = Bdx + Jc.T @ dyc + Jd.T @ dyd - rx0
    nrm = lambda v: float(v.abs().max())
    s1 = nrm(Bdx) + nrm(Jc.T @ dyc) + nrm(Jd.T @ dyd) + nrm(rx0)                 # normwise (B is only available as an operator)
    r2 = Jc @ dx - ryc
    s2 = Jc.abs() @ dx.abs() + ryc.abs()                                          # componentwise (Oettli-Prager)
    r3 = Jd @ dx - dyd / Dd - ryd
    s3 = Jd.abs() @ dx.abs() + (dyd / Dd).abs() + ryd.abs()
    be = max(nrm(r1) / s1, float((r2.abs() / s2).max()), float((r3.abs() / s3).max()))
    assert be < 1e-9, be

    # ---- (2) one column of the reduced matrix N = J (B + Dx)^-1 J^T + diag(0, Dd^-1) against an independent operator:
    # w = (B + Dx)^-1 J^T e_c by conjugate gradients on the torch operator (B + Dx is SPD), then J w
    c = me + 3
    e = torch.zeros(me + mi, dtype=torch.float64, device="cuda"); e[c] = 1.0
    b = J.T @ e
    A = lambda v: _bns_times(S, Y, sigma, v) + Dx * v
    w = b / (sigma + Dx); r = b - A(w); pvec = r.clone(); rr = float(r @ r)
    for _ in range(200):
        Ap = A(pvec); alpha = rr / float(pvec @ Ap)
        w = w + alpha * pvec; r = r - alpha * Ap
        rr_new = float(r @ r)
        if rr_new ** 0.5 <= 1e-13 * float(b.norm()):
            break
        pvec = r + (rr_new / rr) * pvec; rr = rr_new
    col = J @ w
    col[c] += 1.0 / float(Dd[c - me])
    Nlib = K.N()
    ref = col
    got = torch.where(torch.arange(me + mi, device="cuda") <= c, Nlib[:, c], Nlib[c, :])   # upper triangle stored
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-9

    # ---- (3) cached N vs rebuilt N: bit identity
    out = []
    for cache in (1, 1, 0):
        ctx._L.hiopamd_kkt_lowrank_set_cache(K.h, cache)
        rx = rx0.clone(); torch.cuda.synchronize()
        a, b_, c_ = torch.zeros_like(dx), torch.zeros_like(dyc), torch.zeros_like(dyd)
        assert K.solve_compressed(rx, ryc, ryd, a, b_, c_); ctx.sync()
        out.append((a.clone(), b_.clone(), c_.clone()))
    for o in out:
        assert torch.equal(o[0], dx) and torch.equal(o[1], dyc) and torch.equal(o[2], dyd)
    ctx._L.hiopamd_kkt_lowrank_set_cache(K.h, 1)

    # ---- (4) the secant update with an unchanged Jacobian: the Jacobian term of y_new vanishes exactly (fused difference
    # pass), so the stored pair is (x_new - x_old, g_new - g_old)
    x2 = x + U(n, lo=-0.05, hi=0.05)
    gx_prev, gx2 = q * x, q * x2
    torch.cuda.synchronize()                               # torch's stream -> the context's stream
    H.update(x, gx_prev, Jc, Jd, yc, yd); ctx.sync()
    assert H.update(x2, gx2, Jc, Jd, yc, yd); ctx.sync()
    S2, Y2 = H.St(), H.Yt()
    assert torch.equal(S2[-1], x2 - x) and torch.equal(Y2[-1], gx2 - gx_prev)
    K.close(); H.close()
