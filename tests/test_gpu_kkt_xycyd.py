"""GPU parity of the full-space XYcYd layer (hiopamd_kkt_xycyd_*) against oracle/kkt_full.py, the restatement of
hiopKKTLinSys.cpp:218-376,543-690,911-961,1619-1736, hiopKrylovSolver.cpp:390-700, hiopPDPerturbation.cpp and
hiopKKTLinSysDense.hpp:84-212.  Same seeded iterate/residual through both; fp64 tolerances: directions 1e-9
relative to the slab's inf-norm (they pass through a condensed LDL^T solve), operator products 1e-13,
perturbation values and re-factorization counts exact."""
import numpy as np
import pytest
import torch

from oracle import hiop_oracle as ho
from oracle import kkt_full as kf
from oracle import problems
from tests import kkt_full_cases as cases

pytestmark = pytest.mark.gpu


def D(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float64).cuda()


def close(a, b, rtol):
    scale = max(np.abs(b).max(), 1e-300)
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * scale)


def gpu_mds(ctx, p, k_oracle, ixl, ixu, idl, idu):
    from hiop_amd.kkt import KKTLinSysXYcYd, mds_from_problem
    kg, d = mds_from_problem(ctx, p)
    d["Hdd"] = D(k_oracle.Hdd)
    d["Jcd"] = D(k_oracle.Jcd)
    d["Jcs_v"] = D(k_oracle.Jcs_val)
    kg.set_values(d["Jcs_v"], d["Jds_v"], d["Hss_v"], d["Jcd"], d["Jdd"], d["Hdd"], None, None)
    pats = [D(ixl), D(ixu), D(idl), D(idu)]
    full = KKTLinSysXYcYd(ctx, kg, *pats)
    return kg, full, d


def compare_dirs(fg, dg_slab, d_oracle, rtol=1e-9):
    dg = fg.unpack(dg_slab, kf.ITER_PARTS)
    flat_g = kf.pack(dg, kf.ITER_PARTS)
    flat_o = kf.pack(d_oracle, kf.ITER_PARTS)
    close(flat_g, flat_o, rtol)
    return dg


@pytest.mark.parametrize("ns,nd,neq", [(8, 6, None), (40, 33, 17), (300, 70, 129)])
def test_mds_update_directions_operator(ctx, ns, nd, neq):
    p, k, fo, it = cases.mds_case(ns, nd, neq)
    kg, fg, keep = gpu_mds(ctx, p, k, fo.ixl, fo.ixu, fo.idl, fo.idu)
    assert fg.dim == sum(fo.sizes)
    it_g = fg.pack(it, kf.ITER_PARTS)
    assert fo.update(it) and fg.update(it_g)
    assert fg.num_refact == fo.num_refact == 0 and fg.deltas() == fo.perturb.deltas()
    r = cases.random_resid(fo.sizes, fo.ixl, fo.ixu, fo.idl, fo.idu)
    r_g = fg.pack(r, kf.RESID_PARTS)
    ok_o, d_o = fo.compute_directions(r)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert fg.compute_directions(r_g, d_g) and ok_o
    ctx.sync()
    compare_dirs(fg, d_g, d_o)
    # the 12-block operator: y = K d must reproduce the residual on both sides, and agree part by part
    y_g = torch.zeros_like(d_g)
    torch.cuda.synchronize()
    fg.times_vec(y_g, d_g); ctx.sync()
    y_o = kf.pack(fo.times_vec(d_o), kf.RESID_PARTS)
    close(y_g.cpu().numpy(), y_o, 1e-9)
    close(y_g.cpu().numpy(), kf.pack(r, kf.RESID_PARTS), 1e-9)
    # operator alone on an arbitrary vector (no solve in between): tight tolerance
    xr = np.random.Generator(np.random.PCG64(9)).uniform(-1, 1, fg.dim)
    y2 = torch.zeros_like(d_g)
    xr_g = D(xr)
    torch.cuda.synchronize()
    fg.times_vec(y2, xr_g); ctx.sync()
    y2o = fo.times_vec_flat(xr)
    close(y2.cpu().numpy(), y2o, 1e-13)


def test_mds_inertia_correction_matches_reference_sequence(ctx):
    p, k, fo, it = cases.mds_case(16, 9, nonconvex=True)
    kg, fg, keep = gpu_mds(ctx, p, k, fo.ixl, fo.ixu, fo.idl, fo.idu)
    it_g = fg.pack(it, kf.ITER_PARTS)
    for _ in range(3):     # wrong inertia -> x100 growth; then restart from last/3 on the next matrix
        assert fo.update(it) and fg.update(it_g)
        assert fg.num_refact == fo.num_refact
        assert fg.deltas() == fo.perturb.deltas()
    assert fg.deltas()[0] > 0
    # directions with delta_w > 0 in the system and in the operator
    r = cases.random_resid(fo.sizes, fo.ixl, fo.ixu, fo.idl, fo.idu)
    ok_o, d_o = fo.compute_directions(r)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert fg.compute_directions(fg.pack(r, kf.RESID_PARTS), d_g)
    ctx.sync()
    compare_dirs(fg, d_g, d_o)


def test_mds_singular_jacobian_delta_c(ctx):
    p, k, fo, it = cases.mds_case(8, 6)
    cases.zero_equality_row(k, 1)
    kg, fg, keep = gpu_mds(ctx, p, k, fo.ixl, fo.ixu, fo.idl, fo.idu)
    fo.perturb.set_mu(1e-2)
    fg.set_mu(1e-2)
    it_g = fg.pack(it, kf.ITER_PARTS)
    assert fo.update(it) and fg.update(it_g)
    assert fg.num_refact == fo.num_refact == 1
    assert fg.deltas() == fo.perturb.deltas() and fg.deltas()[2] == pytest.approx(1e-8 * 1e-2 ** 0.25)


def test_max_refactorizations_reports_failure(ctx):
    """delta_w_max_bar tiny: the perturbation overflows its cap, the acceptor gives up and update returns false
    on both sides (hiopKKTLinSys.cpp:355-357, hiopPDPerturbation.cpp:349-354)."""
    p, k, fo, it = cases.mds_case(8, 6, nonconvex=True)
    kg, fg, keep = gpu_mds(ctx, p, k, fo.ixl, fo.ixu, fo.idl, fo.idu)
    fo.perturb.delta_w_max_bar = 1e-3
    fo.perturb.delta_c_bar = 1e-8
    fg.set_perturbation_options([1e-20, 1e-3, 1e-4, 1. / 3, 100., 8., 1e-8, 0.25])
    it_g = fg.pack(it, kf.ITER_PARTS)
    ro, rg = fo.update(it), fg.update(it_g)
    assert ro == rg is False
    assert fg.deltas() == fo.perturb.deltas()


@pytest.mark.parametrize("pivoted", [False, True])
@pytest.mark.parametrize("nx,neq,nineq,nonconvex", [(12, 3, 4, False), (70, 9, 30, True), (33, 0, 5, False)])
def test_dense_xycyd_build_and_directions(ctx, nx, neq, nineq, nonconvex, pivoted):
    """pivoted: the back-end's linear solver in Bunch-Kaufman mode (hiopamd_linsolver_set_pivoting on hiopamd_kkt_xycyd_linsolver) -- what
    the reference's dense KKT classes run with compute_mode = cpu (hiopKKTLinSysDense.hpp:100-119: hiopLinSolverSymDenseLapack); the
    inertia-correction sequence of the non-convex case must be the oracle's (which factors with LAPACK DSYTRF) either way."""
    from hiop_amd.kkt import KKTLinSysXYcYd
    from hiop_amd._lib import lib
    (H, Jc, Jd, ixl, ixu, idl, idu), fo, it = cases.dense_case(nx, neq, nineq, seed=nx, nonconvex=nonconvex)
    fg = KKTLinSysXYcYd(ctx, None, D(ixl), D(ixu), D(idl), D(idu), dense_dims=(nx, neq, nineq))
    if pivoted:
        L = lib()
        import ctypes as C
        L.hiopamd_kkt_xycyd_linsolver.restype = C.c_void_p
        assert L.hiopamd_linsolver_set_pivoting(C.c_void_p(L.hiopamd_kkt_xycyd_linsolver(fg.h)), 1) == 0
    fg.set_matrices(D(H), D(Jc), D(Jd))
    it_g = fg.pack(it, kf.ITER_PARTS)
    assert fo.update(it) and fg.update(it_g)
    assert fg.num_refact == fo.num_refact and fg.deltas() == fo.perturb.deltas()
    if nonconvex:
        assert fg.num_refact > 0
    # the assembled matrix before factorisation is compared through a rebuild on the oracle side (upper triangle)
    r = cases.random_resid(fo.sizes, ixl, ixu, idl, idu)
    ok_o, d_o = fo.compute_directions(r)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert fg.compute_directions(fg.pack(r, kf.RESID_PARTS), d_g) and ok_o
    ctx.sync()
    compare_dirs(fg, d_g, d_o)
    ok_g, info_g = fg.compute_directions_w_IR(fg.pack(r, kf.RESID_PARTS), d_g)
    ctx.sync()
    ok_o, d_o2, info_o = fo.compute_directions_w_IR(r, mu=1e-8)
    assert ok_g and info_g["converged"] == info_o["converged"] is True
    assert info_g["flag"] == info_o["flag"] == 0 and info_g["iter"] == info_o["iter"]
    compare_dirs(fg, d_g, d_o2)


def test_ir_with_inexact_preconditioner_iterates(ctx):
    """Factor with H, then apply the operator with H + E (E small, symmetric): the condensed solve is now only an
    approximate inverse and BiCGStab has to iterate.  Same perturbation on both sides."""
    from hiop_amd.kkt import KKTLinSysXYcYd
    nx, neq, nineq = 40, 6, 9
    (H, Jc, Jd, ixl, ixu, idl, idu), fo, it = cases.dense_case(nx, neq, nineq, seed=77)
    fg = KKTLinSysXYcYd(ctx, None, D(ixl), D(ixu), D(idl), D(idu), dense_dims=(nx, neq, nineq))
    Hg = D(H)
    fg.set_matrices(Hg, D(Jc), D(Jd))
    it_g = fg.pack(it, kf.ITER_PARTS)
    assert fo.update(it) and fg.update(it_g)
    rng = np.random.Generator(np.random.PCG64(5))
    E = rng.uniform(-1, 1, (nx, nx)) * 2e-2
    H2 = H + (E + E.T) / 2
    fo.p.H = H2
    H2g = D(H2)
    fg.set_matrices(H2g, fg._keep[1], fg._keep[2])
    fo.perturb.set_mu(1.0)
    fg.set_mu(1.0)
    r = cases.random_resid(fo.sizes, ixl, ixu, idl, idu)
    r_g = fg.pack(r, kf.RESID_PARTS)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ok_g, info_g = fg.compute_directions_w_IR(r_g, d_g, ir_outer_tol_factor=1e-2, ir_outer_tol_min=1e-9, ir_outer_maxit=30)
    ctx.sync()
    ok_o, d_o, info_o = fo.compute_directions_w_IR(r, mu=1.0, ir_outer_tol_factor=1e-2, ir_outer_tol_min=1e-9,
                                                   ir_outer_maxit=30)
    assert info_o["converged"] and info_g["converged"] and info_o["iter"] >= 1.5
    assert abs(info_g["iter"] - info_o["iter"]) <= 0.5
    assert info_g["rel_resid"] <= 1e-9
    compare_dirs(fg, d_g, d_o, rtol=1e-7)
    # true residual of the returned direction w.r.t. the (H + E) operator
    y = torch.zeros_like(d_g)
    torch.cuda.synchronize()
    fg.times_vec(y, d_g); ctx.sync()
    rr = np.linalg.norm(y.cpu().numpy() - kf.pack(r, kf.RESID_PARTS)) / np.linalg.norm(kf.pack(r, kf.RESID_PARTS))
    assert rr <= 1.05e-9
    # maxit too small: reported as not converged, step still accepted (hiopKKTLinSys.cpp:949-953), min-residual iterate
    ok_g, info_g = fg.compute_directions_w_IR(r_g, d_g, 1e-2, 1e-9, 1)
    ctx.sync()
    ok_o, d_o, info_o = fo.compute_directions_w_IR(r, mu=1.0, ir_outer_tol_factor=1e-2, ir_outer_tol_min=1e-9,
                                                   ir_outer_maxit=1)
    assert ok_g and not info_g["converged"] and not info_o["converged"]
    assert info_g["flag"] == info_o["flag"] == 1 and info_g["iter"] == info_o["iter"]
    assert info_g["abs_resid"] == pytest.approx(info_o["abs_resid"], rel=1e-6)
    compare_dirs(fg, d_g, d_o, rtol=1e-7)
    # zero right-hand side -> zero direction
    z = torch.zeros_like(d_g)
    ok_g, info_g = fg.compute_directions_w_IR(z, d_g)
    ctx.sync()
    assert ok_g and info_g["converged"] and not d_g.any()
    # ir_outer_maxit = 0 falls back to computeDirections (:916-919)
    ok_g, info_g = fg.compute_directions_w_IR(r_g, d_g, ir_outer_maxit=0)
    ctx.sync()
    ok_o, d_o = fo.compute_directions(r)
    compare_dirs(fg, d_g, d_o)


def lowrank_pair(ctx, n, me, mi, seed):
    from tests.test_gpu_lowrank import drive
    Ho, Hg, (Jc, Jd), r, stored = drive(ctx, n, me, mi, 6, 9, "sigma0", seed=seed)
    return Ho, Hg, Jc, Jd, r


@pytest.mark.parametrize("n,me,mi", [(400, 3, 4), (6000, 1, 0), (2500, 0, 7)])
def test_lowrank_full_space(ctx, n, me, mi):
    from hiop_amd.kkt import KKTLinSysLowRank, KKTLinSysXYcYd
    Ho, Hg, Jc, Jd, r = lowrank_pair(ctx, n, me, mi, seed=n)
    ixl = (r.uniform(0, 1, n) < 0.7).astype(np.float64)
    ixu = (r.uniform(0, 1, n) < 0.3).astype(np.float64)
    idl = np.ones(mi)
    idu = (r.uniform(0, 1, mi) < 0.5).astype(np.float64)
    Ko = ho.KKTLinSysLowRank(Ho, me, mi)
    fo = kf.KKTLinSysFull(kf.LowRankProvider(Ko, Jc, Jd), ixl, ixu, idl, idu, perturb=kf.PDPerturbationNull())
    Kg = KKTLinSysLowRank(ctx, Hg)
    fg = KKTLinSysXYcYd(ctx, Kg, D(ixl), D(ixu), D(idl), D(idu))
    fg.set_matrices(None, D(Jc), D(Jd))
    it = cases.random_iterate(n, mi, me, mi, ixl, ixu, idl, idu, seed=3)
    # every variable needs a positive barrier diagonal or sigma > 0: fine, B0 = sigma I
    it_g = fg.pack(it, kf.ITER_PARTS)
    assert fo.update(it) and fg.update(it_g)
    res = cases.random_resid(fo.sizes, ixl, ixu, idl, idu)
    r_g = fg.pack(res, kf.RESID_PARTS)
    ok_o, d_o = fo.compute_directions(res)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert fg.compute_directions(r_g, d_g) and ok_o
    ctx.sync()
    compare_dirs(fg, d_g, d_o, rtol=1e-8)
    xr = r.uniform(-1, 1, fg.dim)
    y = torch.zeros_like(d_g)
    xr_g = D(xr)
    torch.cuda.synchronize()
    fg.times_vec(y, xr_g); ctx.sync()
    close(y.cpu().numpy(), fo.times_vec_flat(xr), 1e-11)
    ok_g, info_g = fg.compute_directions_w_IR(r_g, d_g)
    ctx.sync()
    ok_o, d_o2, info_o = fo.compute_directions_w_IR(res, mu=1e-8)
    assert info_g["converged"] and info_o["converged"]
    compare_dirs(fg, d_g, d_o2, rtol=1e-8)


def test_full_size_config_directions_and_ir(ctx):
    """BASELINE config C3 (n_sparse = 1e5, n_dense = 4096, 4093+3 constraints, N = 8192): no oracle solve at this
    size in the test; size-independent property instead — the returned direction satisfies the 12-block system it
    was computed for, to the IR tolerance, and the patterns of the bound parts are respected."""
    p = problems.mds_ex1_g(50000, 4096, 4093)
    ixl, ixu, idl, idu = cases.patterns(p)
    k = cases.oracle_mds(p)     # holds the numpy values only; nothing is solved on the oracle side
    kg, fg, keep = gpu_mds(ctx, p, k, ixl, ixu, idl, idu)
    nx, nd = p.nxs + p.nxd, p.nineq
    it = cases.random_iterate(nx, nd, p.neq, nd, ixl, ixu, idl, idu, seed=1)
    sizes = kf.part_sizes(nx, nd, p.neq, nd)
    r = cases.random_resid(sizes, ixl, ixu, idl, idu, seed=2)
    it_g, r_g = fg.pack(it, kf.ITER_PARTS), fg.pack(r, kf.RESID_PARTS)
    fg.set_mu(1e-3)
    assert fg.update(it_g) and fg.num_refact == 0
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ok, info = fg.compute_directions_w_IR(r_g, d_g)
    ctx.sync()
    assert ok and info["converged"] and info["rel_resid"] <= 1e-5
    y = torch.zeros_like(d_g)
    fg.times_vec(y, d_g); ctx.sync()
    rel = (torch.linalg.norm(y - r_g) / torch.linalg.norm(r_g)).item()
    assert rel <= 1.01e-5
    d = fg.unpack(d_g, kf.ITER_PARTS)
    for s, pat in (("sxl", ixl), ("zl", ixl), ("sxu", ixu), ("zu", ixu), ("sdl", idl), ("vl", idl), ("sdu", idu),
                   ("vu", idu)):
        assert not d[s][pat == 0].any()


@pytest.mark.parametrize("nx,neq,nineq,nonconvex", [(14, 3, 5, False), (60, 8, 21, True), (20, 0, 4, False)])
def test_dense_xdycyd_form(ctx, nx, neq, nineq, nonconvex):
    """hiopKKTLinSysDenseXDYcYd + hiopKKTLinSysCompressedXDYcYd::computeDirections (a12, second form)."""
    from hiop_amd.kkt import KKTLinSysXYcYd
    (H, Jc, Jd, ixl, ixu, idl, idu), fo, it = cases.dense_case(nx, neq, nineq, seed=nx + 1, nonconvex=nonconvex,
                                                               xd_form=True)
    fg = KKTLinSysXYcYd(ctx, None, D(ixl), D(ixu), D(idl), D(idu), dense_dims=(nx, neq, nineq), xd_form=True)
    fg.set_matrices(D(H), D(Jc), D(Jd))
    it_g = fg.pack(it, kf.ITER_PARTS)
    # assembled matrix (upper triangle) before factorisation: build only, compare, then update
    assert fo.update(it) and fg.update(it_g)
    assert fg.num_refact == fo.num_refact and fg.deltas() == fo.perturb.deltas()
    r = cases.random_resid(fo.sizes, ixl, ixu, idl, idu)
    ok_o, d_o = fo.compute_directions(r)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert fg.compute_directions(fg.pack(r, kf.RESID_PARTS), d_g) and ok_o
    ctx.sync()
    compare_dirs(fg, d_g, d_o)
    y = torch.zeros_like(d_g)
    fg.times_vec(y, d_g); ctx.sync()
    close(y.cpu().numpy(), kf.pack(r, kf.RESID_PARTS), 1e-9)


def test_inertia_free_acceptor_test_direction_and_refactorize(ctx):
    """hiopFactAcceptorInertiaFreeDWD + test_direction + factorize_inertia_free, driven the way
    hiopAlgFilterIPMBase::compute_search_direction_inertia_free does (hiopAlgFilterIPM.cpp:3374-3440)."""
    from hiop_amd.kkt import KKTLinSysXYcYd
    nx, neq, nineq = 16, 3, 4
    (H, Jc, Jd, ixl, ixu, idl, idu), fo, it = cases.dense_case(nx, neq, nineq, seed=21, nonconvex=True, inertia_free=True,
                                                               neg_value=-50.0, free_nonconvex=True)
    fg = KKTLinSysXYcYd(ctx, None, D(ixl), D(ixu), D(idl), D(idu), dense_dims=(nx, neq, nineq))
    fg.set_fact_acceptor(True)
    fg.set_matrices(D(H), D(Jc), D(Jd))
    it_g = fg.pack(it, kf.ITER_PARTS)
    assert fo.update(it) and fg.update(it_g)
    assert fg.num_refact == fo.num_refact == 0 and fg.deltas() == fo.perturb.deltas() == (0.0, 0.0, 0.0, 0.0)
    r = cases.negative_curvature_resid(fo.sizes)
    r_g = fg.pack(r, kf.RESID_PARTS)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    rounds = 0
    while True:
        ok_o, d_o = fo.compute_directions(r)
        torch.cuda.synchronize()
        assert fg.compute_directions(r_g, d_g) and ok_o
        ctx.sync()
        compare_dirs(fg, d_g, d_o, rtol=1e-8)
        acc_o = fo.test_direction(d_o)
        acc_g, dWd, nrm = fg.test_direction(d_g)
        assert acc_g == acc_o
        assert dWd == pytest.approx(fo.last_dWd, rel=1e-7, abs=1e-9 * fo.last_xs_nrmsq)
        assert nrm == pytest.approx(fo.last_xs_nrmsq, rel=1e-7)
        if acc_g:
            break
        assert fo.factorize_inertia_free() and fg.factorize_inertia_free()
        assert fg.deltas() == fo.perturb.deltas()
        rounds += 1
        assert rounds <= 10
    assert rounds >= 1 and fg.deltas()[0] > 0


class _FrozenDeltas:
    """The oracle-side stand-in for a *Rand perturbation whose draw was made elsewhere: hands out the given vectors and never
    changes them (any request for a correction means the oracle disagrees with the inertia the device accepted)."""

    def __init__(self, vecs):
        self.vecs = tuple(vecs)
        self.asked = 0

    def set_mu(self, mu):
        pass

    def deltas(self):
        return self.vecs

    def compute_initial_deltas(self):
        return True

    def compute_perturb_wrong_inertia(self):
        self.asked += 1
        return True

    compute_perturb_singularity = compute_perturb_wrong_inertia


@pytest.mark.parametrize("backend", ["mds", "dense_xycyd", "dense_xdycyd"])
@pytest.mark.parametrize("dual_first", [False, True])
def test_randomized_regularisation_vectors(ctx, backend, dual_first):
    """hiopPDPerturbationPrimalFirstRand / DualFirstRand (regularization_method = randomized): on a non-convex system the
    inertia-correction loop ends with VECTOR regularisations, uniform in [0.9, 1.0] x the scalar of the state machine
    (hiopPDPerturbation.hpp:53-54).  Checked: the range and the spread of the drawn vectors, reproducibility from the seed,
    and that build, solve, 12-block operator and test_direction all consume the same vectors — through the oracle's
    restatements evaluated WITH THE DEVICE'S VECTORS (the reference draws from the host's std generator: there is no stream
    to be bit-identical with)."""
    from hiop_amd.kkt import KKTLinSysXYcYd
    if backend == "mds":
        p, k, fo, it = cases.mds_case(16, 9, nonconvex=True)
        kg, fg, keep = gpu_mds(ctx, p, k, fo.ixl, fo.ixu, fo.idl, fo.idu)
        kg2, fg2, keep2 = gpu_mds(ctx, p, k, fo.ixl, fo.ixu, fo.idl, fo.idu)
    else:
        xd = backend == "dense_xdycyd"
        nx, neq, nineq = 60, 8, 21
        (H, Jc, Jd, ixl, ixu, idl, idu), fo, it = cases.dense_case(nx, neq, nineq, seed=nx + int(xd), nonconvex=True, xd_form=xd,
                                                                   neg_value=-50.0, free_nonconvex=True)
        mk = lambda: KKTLinSysXYcYd(ctx, None, D(ixl), D(ixu), D(idl), D(idu), dense_dims=(nx, neq, nineq), xd_form=xd)
        fg, fg2 = mk(), mk()
        mats = (D(H), D(Jc), D(Jd))
        fg.set_matrices(*mats); fg2.set_matrices(*mats)
    fg.set_regularization(dual_first=dual_first, randomized=True, seed=1234)
    fg2.set_regularization(dual_first=dual_first, randomized=True, seed=1234)
    it_g = fg.pack(it, kf.ITER_PARTS)
    ok = fg.update(it_g)
    sc = fg.deltas()
    if dual_first and not ok:
        # dual regularisation alone cannot repair a negative Hessian block: the machine raises delta_c to its cap, then switches
        # to the primal one (hiopPDPerturbation.cpp:529-540); within 10 re-factorisations that may not finish — same on the oracle
        fo.perturb = kf.PDPerturbationDualFirstScalar()
        assert fo.update(it) is False
        return
    assert ok and fg.num_refact > 0
    vecs = [v.cpu().numpy() for v in fg.delta_vectors()]
    lo, hi = 0.9, 1.0
    for v, s in zip(vecs, sc):
        if v.size == 0:
            continue
        assert (v >= lo * s).all() and (v <= hi * s).all()
        if s > 0 and v.size >= 8:
            assert v.std() > 0.01 * s and abs(v.mean() - 0.95 * s) < 0.04 * s
    assert sc[0] > 0 or sc[2] > 0
    # same seed, same sequence of draws -> identical vectors; another seed -> different
    assert fg2.update(it_g) and fg2.deltas() == sc
    for a, b in zip(fg2.delta_vectors(), vecs):
        assert np.array_equal(a.cpu().numpy(), b)
    fg2.set_regularization(dual_first=dual_first, randomized=True, seed=99)
    assert fg2.update(it_g)
    assert any(v.size and not np.array_equal(a.cpu().numpy(), v) for a, v in zip(fg2.delta_vectors(), vecs))
    # the oracle with the device's vectors: its factorisation of that matrix has the inertia the device accepted ...
    fo.perturb = _FrozenDeltas(vecs)
    assert fo.update(it) and fo.perturb.asked == 0
    # ... and directions, 12-block operator and test_direction agree
    r = cases.random_resid(fo.sizes, fo.ixl, fo.ixu, fo.idl, fo.idu)
    r_g = fg.pack(r, kf.RESID_PARTS)
    ok_o, d_o = fo.compute_directions(r)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert fg.compute_directions(r_g, d_g) and ok_o
    ctx.sync()
    compare_dirs(fg, d_g, d_o, rtol=1e-8)
    xr = np.random.Generator(np.random.PCG64(5)).uniform(-1, 1, fg.dim)
    y = torch.zeros_like(d_g)
    xr_g = D(xr)
    torch.cuda.synchronize()
    fg.times_vec(y, xr_g); ctx.sync()
    close(y.cpu().numpy(), fo.times_vec_flat(xr), 1e-13)
    acc_g, dWd, nrm = fg.test_direction(d_g)
    acc_o = fo.test_direction(d_o)
    assert acc_g == acc_o and dWd == pytest.approx(fo.last_dWd, rel=1e-6, abs=1e-9 * fo.last_xs_nrmsq)
    ok_g, info_g = fg.compute_directions_w_IR(r_g, d_g)
    ctx.sync()
    ok_o, d_o2, info_o = fo.compute_directions_w_IR(r, mu=1e-8)
    assert ok_g and info_g["converged"] and info_o["converged"]
    compare_dirs(fg, d_g, d_o2, rtol=1e-8)


@pytest.mark.parametrize("randomized", [False, True])
def test_dual_first_regularisation_on_a_singular_jacobian(ctx, randomized):
    """hiopPDPerturbationDualFirstScalar / DualFirstRand on a rank-deficient equality Jacobian: the first correction is the dual
    one (delta_c = max(1e-20, 1e-8 mu^0.25)), the primal deltas stay zero; scalar mode: deltas and re-factorisation count equal
    to the oracle's machine; randomized: delta_cc / delta_cd vectors in [0.9, 1.0] x delta_c and the oracle agrees on the
    inertia of the matrix built with them."""
    p, k, fo, it = cases.mds_case(8, 6)
    cases.zero_equality_row(k, 1)
    kg, fg, keep = gpu_mds(ctx, p, k, fo.ixl, fo.ixu, fo.idl, fo.idu)
    fo.perturb = kf.PDPerturbationDualFirstScalar()
    fo.perturb.set_mu(1e-2)
    fg.set_regularization(dual_first=True, randomized=randomized, seed=7)
    fg.set_mu(1e-2)
    it_g = fg.pack(it, kf.ITER_PARTS)
    assert fo.update(it) and fg.update(it_g)
    assert fg.num_refact == fo.num_refact == 1
    assert fg.deltas() == fo.perturb.deltas() == (0.0, 0.0, 1e-8 * 1e-2 ** 0.25, 1e-8 * 1e-2 ** 0.25)
    if randomized:
        vecs = [v.cpu().numpy() for v in fg.delta_vectors()]
        dc = fg.deltas()[2]
        assert not vecs[0].any() and not vecs[1].any()
        for v in vecs[2:]:
            assert (v >= 0.9 * dc).all() and (v <= dc).all() and v.std() > 0
        fo.perturb = _FrozenDeltas(vecs)
        assert fo.update(it) and fo.perturb.asked == 0
