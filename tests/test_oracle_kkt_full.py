"""CPU checks pinning oracle/kkt_full.py (the restatement of HiOp's full-space KKT layer) through algebraic
identities of the 12-block system it encodes (hiopKKTLinSys.cpp:1619 block matrix)."""
import numpy as np
import pytest

from oracle import kkt_full as kf
from tests import kkt_full_cases as cases


def _flat_err(full, r, d):
    y = full.times_vec(d)
    e = kf.pack(y, kf.RESID_PARTS) - kf.pack(r, kf.RESID_PARTS)
    return np.linalg.norm(e) / np.linalg.norm(kf.pack(r, kf.RESID_PARTS))


@pytest.mark.parametrize("ns,nd,neq", [(8, 6, None), (12, 5, 7)])
def test_mds_compute_directions_solves_full_system(ns, nd, neq):
    p, k, full, it = cases.mds_case(ns, nd, neq)
    assert full.update(it)
    assert full.num_refact == 0 and full.perturb.deltas() == (0.0, 0.0, 0.0, 0.0)
    r = cases.random_resid(full.sizes, full.ixl, full.ixu, full.idl, full.idu)
    ok, d = full.compute_directions(r)
    assert ok
    assert _flat_err(full, r, d) < 1e-11
    # directions respect the bound patterns (the reference's DEEPCHECKS asserts, hiopKKTLinSys.cpp:296-303)
    for s, pat in (("sxl", full.ixl), ("zl", full.ixl), ("sxu", full.ixu), ("zu", full.ixu), ("sdl", full.idl),
                   ("vl", full.idl), ("sdu", full.idu), ("vu", full.idu)):
        assert np.all(d[s][pat == 0] == 0.0)


def test_dense_compute_directions_and_ir():
    _, full, it = cases.dense_case()
    assert full.update(it)
    r = cases.random_resid(full.sizes, full.ixl, full.ixu, full.idl, full.idu)
    ok, d = full.compute_directions(r)
    assert ok and _flat_err(full, r, d) < 1e-11
    ok, d2, info = full.compute_directions_w_IR(r, mu=1e-3)
    assert ok and info["converged"] and info["flag"] == 0
    assert info["iter"] <= 1.0          # the preconditioner is the exact inverse: converges in the first half step
    assert _flat_err(full, r, d2) <= 1e-5 * 1.0001


def test_inertia_correction_loop_nonconvex():
    p, k, full, it = cases.mds_case(8, 6, nonconvex=True)
    assert full.update(it)
    dwx, dwd, dcc, dcd = full.perturb.deltas()
    # first trial delta_0_bar = 1e-4, then x kappa_w_plus_bar = 100 while no earlier successful delta is known
    # (delta_last == 0), until the inertia is (n, m, 0)                      (hiopPDPerturbation.cpp:335-346)
    assert full.num_refact >= 2 and dwx == dwd and dwx > 0 and dcc == dcd == 0.0
    assert np.isclose(dwx, 1e-4 * 100.0 ** (full.num_refact - 1))
    assert full.p.factorize() == p.neq + p.nineq
    # next factorization starts from the last successful delta / 3  (kappa_w_minus)
    last = dwx
    assert full.update(it)
    assert np.isclose(full.perturb.deltas()[0], last / 3) and full.num_refact == 1


def test_singular_jacobian_gets_delta_c():
    p, k, full, it = cases.mds_case(8, 6)
    # an all-zero equality row -> exactly singular KKT at delta_c = 0 (detected by pivoted and unpivoted LDL^T alike)
    cases.zero_equality_row(k, 1)
    full.perturb.set_mu(1e-2)
    ok = full.update(it)
    assert ok
    dwx, dwd, dcc, dcd = full.perturb.deltas()
    assert dcc == dcd == pytest.approx(1e-8 * (1e-2) ** 0.25)


def test_bicgstab_plain_and_preconditioned():
    rng = np.random.Generator(np.random.PCG64(1))
    n = 40
    A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    b = rng.uniform(-1, 1, n)
    x, conv, flag, it, absr, relr = kf.bicgstab(lambda v: A @ v, None, b, 1e-10, 200)
    assert conv and flag == 0 and np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b) * 1.01
    Ainv = np.linalg.inv(A)
    x, conv, flag, it, absr, relr = kf.bicgstab(lambda v: A @ v, lambda v: Ainv @ v, b, 1e-10, 8)
    assert conv and it == 0.5
    # zero rhs -> zero solution, converged with 0 iterations (hiopKrylovSolver.cpp:405-413)
    x, conv, flag, it, absr, relr = kf.bicgstab(lambda v: A @ v, None, np.zeros(n), 1e-10, 8)
    assert conv and it == 0 and not x.any()
    # max-iter exhaustion returns the minimum-residual iterate and reports failure
    x, conv, flag, it, absr, relr = kf.bicgstab(lambda v: A @ v, None, b, 1e-300, 2)
    assert not conv and np.isclose(np.linalg.norm(A @ x - b), absr)


def test_ir_improves_perturbed_preconditioner():
    """With delta_w > 0 the compressed solve is still an exact inverse of the *perturbed* full system the operator
    applies (times_vec includes the deltas, hiopKKTLinSys.cpp:1672-1690), so IR converges at once."""
    p, k, full, it = cases.mds_case(8, 6, nonconvex=True)
    assert full.update(it)
    r = cases.random_resid(full.sizes, full.ixl, full.ixu, full.idl, full.idu)
    ok, d, info = full.compute_directions_w_IR(r, mu=1e-4)
    assert ok and info["converged"] and _flat_err(full, r, d) <= 1e-6 * 1.0001


def test_xdycyd_form_gives_the_same_directions():
    _, f1, it = cases.dense_case(14, 3, 5, seed=4)
    _, f2, _ = cases.dense_case(14, 3, 5, seed=4, xd_form=True)
    assert f1.update(it) and f2.update(it)
    r = cases.random_resid(f1.sizes, f1.ixl, f1.ixu, f1.idl, f1.idu)
    ok1, d1 = f1.compute_directions(r)
    ok2, d2 = f2.compute_directions(r)
    assert ok1 and ok2
    np.testing.assert_allclose(kf.pack(d2, kf.ITER_PARTS), kf.pack(d1, kf.ITER_PARTS), rtol=1e-9, atol=1e-11)
    assert _flat_err(f2, r, d2) < 1e-11


def test_inertia_free_path_regularises_until_curvature_is_positive():
    """The inertia-free acceptor takes the first factorisation as is; the direction then fails the curvature test and
    factorize_inertia_free adds delta_w until test_direction accepts (hiopAlgFilterIPM.cpp:3374-3440)."""
    _, full, it = cases.dense_case(16, 3, 4, seed=21, nonconvex=True, inertia_free=True,
                                                               neg_value=-50.0, free_nonconvex=True)
    assert full.update(it) and full.num_refact == 0 and full.perturb.deltas()[0] == 0.0
    r = cases.negative_curvature_resid(full.sizes)
    rounds = 0
    while True:
        ok, d = full.compute_directions(r)
        assert ok
        if full.test_direction(d):
            break
        assert full.factorize_inertia_free()
        rounds += 1
        assert rounds <= 10
    assert full.last_dWd >= 1e-11 * full.last_xs_nrmsq
    assert rounds >= 1 and full.perturb.deltas()[0] > 0


def test_ipm_slab_restatement_consistency():
    """oracle/ipm_slab.py: the residual parts feed computeDirections (K dir = resid), the norms are those of the parts,
    a full step keeps the patterns, adjust_small_slacks restores strictly positive slacks."""
    from oracle import ipm_slab as osl
    p, k, full, it = cases.mds_case(12, 5, 7)
    rng = np.random.Generator(np.random.PCG64(2))
    nx = p.nxs + p.nxd
    bounds = (np.where(full.ixl == 1.0, p.xl, -1e20), np.where(full.ixu == 1.0, p.xu, 1e20),
              np.where(full.idl == 1.0, p.dl, -1e20), np.where(full.idu == 1.0, p.du, 1e20), rng.uniform(-1, 1, p.neq))
    c, d, g = rng.uniform(-1, 1, p.neq), rng.uniform(-3, 3, p.nineq), rng.uniform(-1, 1, nx)
    mu = 0.2
    assert full.update(it)
    r, n = osl.residual_update(full, it, c, d, g, bounds, mu, 1e-5)
    assert n["nrmInf_bar_feasib"] == max(np.abs(r["ryc"]).max(), np.abs(r["ryd"]).max())
    assert n["nrmInf_bar_complem"] == max(np.abs(r[q]).max() for q in ("rszl", "rszu", "rsvl", "rsvu"))
    ok, dr = full.compute_directions(r)
    assert ok and _flat_err(full, r, dr) < 1e-11
    ap, ad = osl.fraction_to_the_bdry(full, it, dr, 0.995)
    assert 0 < ap <= 1 and 0 < ad <= 1
    trial = osl.take_step(it, dr, ap, ad)
    osl.determine_slacks(full, trial, bounds)
    trial["x"][np.nonzero(full.ixl == 1.0)[0][0]] = bounds[0][np.nonzero(full.ixl == 1.0)[0][0]]   # on its lower bound
    osl.determine_slacks(full, trial, bounds)
    assert osl.adjust_small_slacks(full, trial, it, bounds, mu) >= 1
    for s, pat in (("sxl", full.ixl), ("sxu", full.ixu), ("sdl", full.idl), ("sdu", full.idu)):
        assert np.all(trial[s][pat == 1.0] > 0) and not trial[s][pat == 0.0].any()
