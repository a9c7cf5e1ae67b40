"""GPU parity of the quasi-Newton low-rank path (hiopHessianLowRank + hiopKKTLinSysLowRank) against the
oracle restatement of src/Optimization/hiopHessianLowRank.cpp and hiopKKTLinSys.cpp:1057-1330.

A sequence of secant updates is driven through both implementations with identical inputs (memory growth,
then memory shift), then every operator is compared.  fp64 tolerances: multivectors bit-exact (copies),
sigma 1e-13 rel, operator results 1e-9 relative (they go through 2l x 2l and k x k solves whose condition
numbers are ~1e3-1e6), KKT residual of the full XYcYd system <= 1e-10."""
import numpy as np
import pytest
import torch

from oracle import hiop_oracle as ho

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def D(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float64).cuda()


def drive(ctx, n, me, mi, l_max, nupd, strategy, seed=0):
    from hiop_amd.kkt import HessianLowRank
    r = rng(seed)
    Ho = ho.HessianLowRank(n, l_max=l_max, sigma0=1.0, sigma_update_strategy=strategy)
    Hg = HessianLowRank(ctx, n, me, mi, l_max=l_max, sigma0=1.0, sigma_update_strategy=strategy)
    q = r.uniform(0.5, 3.0, n)       # objective 0.5 sum q_i x_i^2  -> grad = q*x (strictly convex: s^T y > 0)
    Jc0 = r.uniform(-1, 1, (me, n)); Jd0 = r.uniform(-1, 1, (mi, n))
    x = r.uniform(-1, 1, n)
    stored = []
    for it in range(nupd):
        g = q * x
        # mildly nonlinear constraints: Jacobian drifts a little, duals random
        Jc = Jc0 + 0.01 * it * np.sin(np.arange(me * n).reshape(me, n))
        Jd = Jd0 + 0.01 * it * np.cos(np.arange(mi * n).reshape(mi, n))
        yc, yd = r.uniform(-0.1, 0.1, me), r.uniform(-0.1, 0.1, mi)
        so = Ho.update(x, g, Jc, Jd, yc, yd)
        torch.cuda.synchronize()
        sg = Hg.update(D(x), D(g), D(Jc), D(Jd), D(yc), D(yd))
        ctx.sync()
        assert so == sg
        stored.append(so)
        x = x + r.uniform(-0.2, 0.2, n) * (1.0 if it != 3 else 0.0)   # it==3: zero step -> update must be skipped
    return Ho, Hg, (Jc, Jd), r, stored


@pytest.mark.parametrize("n,me,mi,l_max,nupd,strategy", [
    (300, 3, 4, 6, 5, "sigma0"),          # memory still growing
    (1000, 1, 0, 6, 12, "snrm_ynrm"),     # memory full + shifts, Dense Ex1 shape (m=1)
    (5001, 2, 2, 6, 10, "sty"),           # odd n (unaligned double2 path), Dense Ex2 shape (m=4)
    (20000, 60, 40, 4, 7, "sty_inv"),
    (777, 0, 3, 2, 6, "sty_srnm_ynrm"),
    (4000, 2, 3, 12, 16, "sty"),          # secant memory beyond 8 pairs (blocked form of the four l x l Gram blocks), growing then shifting
    (1500, 1, 1, 20, 14, "sigma0"),       # ... still growing at 13 pairs
])
def test_hessian_lowrank_against_oracle(ctx, n, me, mi, l_max, nupd, strategy):
    Ho, Hg, (Jc, Jd), r, stored = drive(ctx, n, me, mi, l_max, nupd, strategy, seed=n)
    assert Hg.l_curr == Ho.St.shape[0] == min(l_max, sum(stored))
    assert stored[4] is False if nupd > 4 else True      # the zero step was rejected
    np.testing.assert_array_equal(Hg.St().cpu().numpy(), Ho.St)
    np.testing.assert_allclose(Hg.Yt().cpu().numpy(), Ho.Yt, rtol=1e-12, atol=1e-13)
    assert Hg.sigma == pytest.approx(Ho.sigma, rel=1e-12)
    Dx = r.uniform(0, 2, n) * (r.uniform(0, 1, n) < 0.5)     # free variables have Dx = 0 (singular leading block of V)
    Ho.update_log_barrier_diagonal(Dx)
    Hg.update_log_barrier_diagonal(D(Dx))
    rhs = r.uniform(-1, 1, n)
    xo = Ho.solve(rhs)
    xg = D(np.zeros(n))
    torch.cuda.synchronize()
    Hg.solve(D(rhs), xg); ctx.sync()
    np.testing.assert_allclose(xg.cpu().numpy(), xo, rtol=1e-9, atol=1e-9 * np.abs(xo).max())
    # B x through the compact form vs the reference's a_k/b_k recursion
    yo = r.uniform(-1, 1, n); yg = D(yo)
    Ho.times_vec(0.5, yo, -2.0, rhs)
    torch.cuda.synchronize()
    Hg.times_vec(0.5, yg, -2.0, D(rhs)); ctx.sync()
    np.testing.assert_allclose(yg.cpu().numpy(), yo, rtol=1e-9, atol=1e-9 * np.abs(yo).max())
    # solve and times_vec are inverses of each other
    back = D(np.zeros(n))
    torch.cuda.synchronize()
    Hg.times_vec(0.0, back, 1.0, xg); ctx.sync()
    np.testing.assert_allclose(back.cpu().numpy(), rhs, rtol=1e-9, atol=1e-9)
    # W = beta*W + alpha*X (B+Dx)^-1 X^T
    k = me + mi
    if k:
        X = np.vstack([Jc, Jd])
        W0 = r.uniform(-1, 1, (k, k)); W0 = W0 + W0.T
        Wo = W0.copy()
        Ho.sym_mat_times_inverse_times_mat_trans(0.5, Wo, 2.0, X)
        Wg = D(W0)
        Hg.sym_mat_times_inverse_times_mat_trans(0.5, Wg, 2.0, D(X))
        np.testing.assert_allclose(Wg.cpu().numpy(), Wo, rtol=1e-9, atol=1e-9 * np.abs(Wo).max())
    Hg.close()


@pytest.mark.parametrize("n,me,mi", [(1000, 1, 0), (5000, 2, 2), (100003, 30, 70), (4000, 90, 110)])   # last: k = 200, 2 x 2 Gram tiles (mirrored tile)
def test_kkt_lowrank_solve_compressed(ctx, n, me, mi):
    from hiop_amd.kkt import KKTLinSysLowRank
    Ho, Hg, (Jc, Jd), r, _ = drive(ctx, n, me, mi, 6, 9, "sigma0", seed=n + 1)
    ixl = (r.uniform(0, 1, n) < 0.6).astype(np.float64); ixu = (r.uniform(0, 1, n) < 0.3).astype(np.float64)
    zl, zu = r.uniform(0.1, 1, n) * ixl, r.uniform(0.1, 1, n) * ixu
    sxl, sxu = r.uniform(0.1, 2, n), r.uniform(0.1, 2, n)
    idl, idu = np.ones(mi), (r.uniform(0, 1, mi) < 0.5).astype(np.float64)
    vl, vu = r.uniform(0.1, 1, mi), r.uniform(0.1, 1, mi) * idu
    sdl, sdu = r.uniform(0.1, 2, mi), r.uniform(0.1, 2, mi)
    Dx = np.zeros(n); ho.axdzpy_w_pattern(Dx, 1.0, zl, sxl, ixl); ho.axdzpy_w_pattern(Dx, 1.0, zu, sxu, ixu)
    Dd = np.zeros(mi); ho.axdzpy_w_pattern(Dd, 1.0, vl, sdl, idl); ho.axdzpy_w_pattern(Dd, 1.0, vu, sdu, idu)
    Ko = ho.KKTLinSysLowRank(Ho, me, mi)
    Ko.update(Dx, Dd, Jc, Jd)
    Kg = KKTLinSysLowRank(ctx, Hg)
    dev = [D(a) for a in (zl, sxl, ixl, zu, sxu, ixu, vl, sdl, idl, vu, sdu, idu, Jc, Jd)]
    torch.cuda.synchronize()
    Kg.update(*dev)
    rx, ryc, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, me), r.uniform(-1, 1, mi)
    ok_o, dx_o, dyc_o, dyd_o = Ko.solve_compressed(rx.copy(), ryc, ryd)
    rxd, dx, dyc, dyd = D(rx), D(np.zeros(n)), D(np.zeros(me)), D(np.zeros(mi))
    torch.cuda.synchronize()
    ok_g = Kg.solve_compressed(rxd, D(ryc), D(ryd), dx, dyc, dyd); ctx.sync()
    assert ok_o and ok_g
    np.testing.assert_allclose(Kg.N().cpu().numpy(), Ko.last_N, rtol=1e-9, atol=1e-9 * np.abs(Ko.last_N).max())
    dx, dyc, dyd = dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy()
    # residual of the full XYcYd system with the explicit dense Hessian (single rank)
    if n <= 5000:
        B = Ho.dense_matrix_local() + np.diag(Dx)
        r1 = B @ dx + Jc.T @ dyc + Jd.T @ dyd - rx
        r2 = Jc @ dx - ryc
        r3 = Jd @ dx - dyd / Dd - ryd
        sc = max(1.0, np.abs(dx).max(), np.abs(dyc).max() if me else 0, np.abs(dyd).max() if mi else 0)
        assert max(np.abs(r1).max(), np.abs(r2).max() if me else 0, np.abs(r3).max() if mi else 0) / sc < 1e-10
    sc = max(np.abs(dx_o).max(), 1e-300)
    assert np.abs(dx - dx_o).max() / sc < 1e-7
    if me:
        assert np.abs(dyc - dyc_o).max() / max(np.abs(dyc_o).max(), 1e-300) < 1e-7
    if mi:
        assert np.abs(dyd - dyd_o).max() / max(np.abs(dyd_o).max(), 1e-300) < 1e-7
    Kg.close(); Hg.close()


def test_kkt_lowrank_solve_compressed_takes_the_refinement_path(ctx):
    """The solve of N dy = rhs is followed by the reference's residual loop (inf-norm 1e-8 ABSOLUTE, up to three refinement steps,
    hiopKKTLinSys.cpp:1192-1330).  The device queues the end of solveCompressed before it knows the first residual and, when that
    residual does not pass, takes back what it applied to rx, refines and repeats the end of the call.  Right-hand sides of 1e9 make
    eps |N| |dy| exceed 1e-8, so the loop runs (and cannot succeed: the three steps are all taken).  The answer must still be the
    oracle's, rx must come out as rx - J^T dy of the FINAL dy, and a second, well-scaled solve on the same object must be unaffected."""
    import ctypes as C
    from hiop_amd.kkt import KKTLinSysLowRank
    n, me, mi = 3000, 20, 30
    Ho, Hg, (Jc, Jd), r, _ = drive(ctx, n, me, mi, 6, 9, "sigma0", seed=77)
    Dx = r.uniform(0.1, 2, n); Dd = r.uniform(0.5, 2, mi)
    Ko = ho.KKTLinSysLowRank(Ho, me, mi)
    Ko.update(Dx, Dd, Jc, Jd)
    Kg = KKTLinSysLowRank(ctx, Hg)
    Kg.update_diag(D(Dx), D(Dd), D(Jc), D(Jd))
    ctx._L.hiopamd_kkt_lowrank_last_residual.restype = C.c_double
    for scale, expect_refine in ((1e9, True), (1.0, False)):
        rx, ryc, ryd = r.uniform(-1, 1, n), scale * r.uniform(-1, 1, me), scale * r.uniform(-1, 1, mi)
        ok_o, dx_o, dyc_o, dyd_o = Ko.solve_compressed(rx.copy(), ryc, ryd)
        rxd, dx, dyc, dyd = D(rx), D(np.zeros(n)), D(np.zeros(me)), D(np.zeros(mi))
        torch.cuda.synchronize()
        assert Kg.solve_compressed(rxd, D(ryc), D(ryd), dx, dyc, dyd); ctx.sync()
        resid = ctx._L.hiopamd_kkt_lowrank_last_residual(Kg.h)
        assert (resid >= 1e-8) == expect_refine, resid
        dxg, dycg, dydg = dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy()
        for a, b in ((dxg, dx_o), (dycg, dyc_o), (dydg, dyd_o)):
            assert np.abs(a - b).max() / max(np.abs(b).max(), 1e-300) < 1e-7
        # rx is left as rx - J^T dy (reference :1178), with the dy that was returned
        want = rx - Jc.T @ dycg - Jd.T @ dydg
        assert np.abs(rxd.cpu().numpy() - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    Kg.close(); Hg.close()


def test_posv_refine(ctx):
    import ctypes as C
    r = rng(5)
    for k in (1, 7, 100, 200):
        G = r.uniform(-1, 1, (k, 3 * k + 5)); Nm = G @ G.T + np.diag(r.uniform(0.1, 10, k) ** 3)   # badly scaled SPD
        b = r.uniform(-1, 1, k)
        bd = D(b); work = torch.zeros(3 * k * k + 8 * k + 8, dtype=torch.float64, device="cuda")
        info, res = C.c_int(-1), C.c_double(-1)
        torch.cuda.synchronize()
        ctx.call("hiopamd_posv_refine", k, D(np.triu(Nm)), k, bd, work, C.byref(info), C.byref(res)); ctx.sync()
        assert info.value == 0 and res.value < 1e-8
        xo, _ = ho.solve_with_refin(Nm, b)
        np.testing.assert_allclose(bd.cpu().numpy(), xo, rtol=1e-8, atol=1e-10)
    # not positive definite -> info != 0 (DPOSVX INFO > 0)
    Nm = np.diag([1.0, -2.0, 3.0])
    info = C.c_int(-1)
    ctx.call("hiopamd_posv_refine", 3, D(Nm), 3, D(np.ones(3)), torch.zeros(200, dtype=torch.float64, device="cuda"),
             C.byref(info), None); ctx.sync()
    assert info.value != 0


def test_rccl_allreduce_hook_single_rank(ctx):
    """RCCL communicator (world_size 1) installed as the context's all-reduce hook: every reduction site of the
    low-rank path goes through ncclAllReduce on the context's stream; results must be unchanged."""
    import os
    import torch.distributed as dist
    from hiop_amd.runtime import Context
    from hiop_amd.kkt import HessianLowRank, KKTLinSysLowRank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1)
    c2 = Context(0)
    try:
        c2.init_rccl_from_torch_distributed()
        n, me, mi = 4000, 3, 2
        r = rng(77)
        Ho = ho.HessianLowRank(n, l_max=6, sigma0=1.0, sigma_update_strategy="sty")
        Hg = HessianLowRank(c2, n, me, mi, l_max=6, sigma0=1.0, sigma_update_strategy="sty")
        q = r.uniform(0.5, 3.0, n); Jc = r.uniform(-1, 1, (me, n)); Jd = r.uniform(-1, 1, (mi, n))
        x = r.uniform(-1, 1, n)
        for it in range(8):
            yc, yd = r.uniform(-0.1, 0.1, me), r.uniform(-0.1, 0.1, mi)
            Ho.update(x, q * x, Jc, Jd, yc, yd)
            torch.cuda.synchronize()
            Hg.update(D(x), D(q * x), D(Jc), D(Jd), D(yc), D(yd)); c2.sync()
            x = x + r.uniform(-0.2, 0.2, n)
        Dx = r.uniform(0, 2, n); Dd = r.uniform(0.5, 2, mi)
        Ko = ho.KKTLinSysLowRank(Ho, me, mi); Ko.update(Dx, Dd, Jc, Jd)
        Kg = KKTLinSysLowRank(c2, Hg)
        torch.cuda.synchronize()
        Kg.update_diag(D(Dx), D(Dd), D(Jc), D(Jd))
        rx, ryc, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, me), r.uniform(-1, 1, mi)
        _, dx_o, dyc_o, dyd_o = Ko.solve_compressed(rx.copy(), ryc, ryd)
        dx, dyc, dyd = D(np.zeros(n)), D(np.zeros(me)), D(np.zeros(mi))
        torch.cuda.synchronize()
        assert Kg.solve_compressed(D(rx), D(ryc), D(ryd), dx, dyc, dyd); c2.sync()
        np.testing.assert_allclose(dx.cpu().numpy(), dx_o, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(dyc.cpu().numpy(), dyc_o, rtol=1e-7, atol=1e-9)
        # the full-space layer on the same (1-rank) partition: compound-vector dots split into their distributed and
        # replicated parts, Jacobian products all-reduced (hiopamd_kkt_xycyd_*, sharded code path)
        from hiop_amd.kkt import KKTLinSysXYcYd
        from oracle import kkt_full as kf
        from tests import kkt_full_cases as cases
        ixl = (r.uniform(0, 1, n) < 0.7).astype(np.float64); ixu = (r.uniform(0, 1, n) < 0.3).astype(np.float64)
        idl = np.ones(mi); idu = np.array([1.0, 0.0])
        fo = kf.KKTLinSysFull(kf.LowRankProvider(Ko, Jc, Jd), ixl, ixu, idl, idu, perturb=kf.PDPerturbationNull())
        fg = KKTLinSysXYcYd(c2, Kg, D(ixl), D(ixu), D(idl), D(idu))
        fg.set_matrices(None, D(Jc), D(Jd))
        it = cases.random_iterate(n, mi, me, mi, ixl, ixu, idl, idu, seed=3)
        res = cases.random_resid(fo.sizes, ixl, ixu, idl, idu)
        it_g, r_g = fg.pack(it, kf.ITER_PARTS), fg.pack(res, kf.RESID_PARTS)
        assert fo.update(it) and fg.update(it_g)
        d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ok_g, info_g = fg.compute_directions_w_IR(r_g, d_g); c2.sync()
        ok_o, d_o, info_o = fo.compute_directions_w_IR(res, mu=1e-8)
        assert ok_g and info_g["converged"] and info_o["converged"]
        got = kf.pack(fg.unpack(d_g, kf.ITER_PARTS), kf.ITER_PARTS)
        want = kf.pack(d_o, kf.ITER_PARTS)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * np.abs(want).max())
        acc, dWd, nrm = fg.test_direction(d_g)
        assert acc == fo.test_direction(d_o) and dWd == pytest.approx(fo.last_dWd, rel=1e-7)
        fg.close(); Kg.close(); Hg.close()
    finally:
        c2.close()
        if own:
            dist.destroy_process_group()
