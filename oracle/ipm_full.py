"""TEST INFRASTRUCTURE ONLY — a compact barrier IPM written ENTIRELY in terms of the full-space layer the product
implements (residual update -> KKT update with inertia correction -> directions with BiCGStab IR -> fraction to the
boundary -> primal/dual step -> safe slacks -> dual safeguard), so that one driver can run on two interchangeable
sets of operations: `OracleOps` (numpy restatements: oracle/kkt_full.py, oracle/ipm_slab.py) and the device ops of the
GPU tests (hiop_amd.kkt.KKTLinSysXYcYd + IpmSlabOps).

It is NOT hiopAlgFilterIPM (no filter line search, no second-order correction, no restoration): monotone barrier
updates and full fraction-to-the-boundary steps, enough for the reference's convex example problems.  Purpose: pin the
composition of the layer end to end against the optimal objective values stored in the reference drivers' -selfcheck
(tests/golden/selfcheck_objectives.json) and let the GPU path be compared iteration by iteration with the CPU path."""
import numpy as np

from . import hiop_oracle as ho
from . import ipm_slab as osl
from . import kkt_full as kf


class OracleOps:
    """The operations of one IPM iteration on numpy dicts; `model` supplies f, grad, c(x), d(x) (constant Jacobians
    and Hessian live in the provider)."""

    def __init__(self, full, bounds, model):
        self.full, self.bounds, self.model = full, bounds, model

    # iterate container helpers
    def copy(self, it):
        return {k: v.copy() for k, v in it.items()}

    def from_host(self, it):
        return self.copy(it)

    def primal(self, it):
        return it["x"].copy()

    def evaluate(self, it):
        return self.model(it["x"])          # f, grad, c, d

    def residual(self, it, ev, mu, kappa_d):
        self.full.it = it
        r, n = osl.residual_update(self.full, it, ev[2], ev[3], ev[1], self.bounds, mu, kappa_d)
        return r, [n[k] for k in osl.NORM_ORDER]

    def kkt_update(self, it, mu):
        self.full.perturb.set_mu(mu)
        self.mu = mu
        return self.full.update(it)

    def directions(self, resid):
        ok, d, info = self.full.compute_directions_w_IR(resid, self.mu)
        return ok, d

    def fraction_to_the_bdry(self, it, d, tau):
        return osl.fraction_to_the_bdry(self.full, it, d, tau)

    def step(self, it, d, ap, ad, mu):
        trial = osl.take_step(it, d, ap, ad)
        osl.determine_slacks(self.full, trial, self.bounds)
        nadj = osl.adjust_small_slacks(self.full, trial, it, self.bounds, mu)
        osl.adjust_duals_plh(self.full, trial, mu, 1e10)
        return trial, nadj

    def n_refactorizations(self):
        return self.full.num_refact


def initial_iterate(full, bounds, x0, d_of_x, mu):
    """x0 projected into its bounds (hiopVector::projectIntoBounds, kappa1 = kappa2 = 1e-2), d = d(x0) projected, slacks from
    the bounds, bound duals mu / slack, yc = yd = 0 — the start of hiopAlgFilterIPM::startingProcedure in spirit."""
    xl, xu, dl, du, _ = bounds
    x = x0.copy()
    ho.project_into_bounds(x, xl, full.ixl, xu, full.ixu, 1e-2, 1e-2)
    d = d_of_x(x).copy()
    ho.project_into_bounds(d, dl, full.idl, du, full.idu, 1e-2, 1e-2)
    p = full.p
    it = {"x": x, "d": d, "yc": np.zeros(p.nyc), "yd": np.zeros(p.nyd)}
    for k in ("sxl", "sxu", "zl", "zu"):
        it[k] = np.zeros(p.nx)
    for k in ("sdl", "sdu", "vl", "vu"):
        it[k] = np.zeros(p.nd)
    osl.determine_slacks(full, it, bounds)
    with np.errstate(divide="ignore", invalid="ignore"):
        it["zl"] = np.where(full.ixl == 0.0, 0.0, mu / it["sxl"])
        it["zu"] = np.where(full.ixu == 0.0, 0.0, mu / it["sxu"])
    osl.determine_duals_bounds_d(full, it, mu)
    return it


def solve(ops, it0, mu0=0.1, tol=1e-8, max_iter=200, kappa_d=1e-5, trace=None, table=None):
    """`table` (a list) receives one record per iteration with the columns of the reference's iteration table
    (hiopAlgFilterIPM.cpp:2783-2812): iter, objective, inf_pr, inf_du, mu, alpha_du, alpha_pr of the step that led here."""
    it = ops.from_host(it0)
    mu = mu0
    nfact = 0
    ap = ad = 0.0
    for k in range(max_iter):
        ev = ops.evaluate(it)
        resid, n = ops.residual(it, ev, mu, kappa_d)
        err0 = max(n[0], n[1], n[2])                       # nrmInf_nlp_{optim, feasib, complem}
        if trace is not None:
            trace.append((ev[0], err0, mu))
        if table is not None:
            table.append(dict(iter=k, objective=float(ev[0]), inf_pr=float(n[1]), inf_du=float(n[0]), mu=float(mu),
                              alpha_du=float(ad), alpha_pr=float(ap)))
        if err0 < tol:
            break
        changed = False
        while max(n[3], n[4], n[5]) < 10 * mu and mu > tol / 10:     # barrier subproblem solved well enough
            mu = max(tol / 10, min(0.2 * mu, mu ** 1.5))
            resid, n = ops.residual(it, ev, mu, kappa_d)
            changed = True
        if not ops.kkt_update(it, mu):
            raise RuntimeError("KKT update failed (inertia correction exhausted)")
        nfact += 1 + ops.n_refactorizations()
        ok, d = ops.directions(resid)
        if not ok:
            raise RuntimeError("direction computation failed")
        tau = max(0.99, 1.0 - mu)
        ap, ad = ops.fraction_to_the_bdry(it, d, tau)
        it, _ = ops.step(it, d, ap, ad, mu)
    ev = ops.evaluate(it)
    return dict(x=ops.primal(it), obj=ev[0], iters=k, n_fact=nfact, err=err0, mu=mu)


def mds_model(p):
    """f, grad, c(x), d(x) of the MdsEx1 family (quadratic objective, linear constraints; hiop_amd/problems.py)."""
    import scipy.sparse as sp
    nxs = p.nxs
    n = p.nxs + p.nxd
    Jcs = sp.csr_matrix((p.Jcs_v, (p.Jcs_i, p.Jcs_j)), shape=(p.neq, nxs))
    Jds = sp.csr_matrix((p.Jds_v, (p.Jds_i, p.Jds_j)), shape=(p.nineq, nxs))
    Hs = np.zeros(nxs)
    ho.spsym_add_diag_to_vec(p.Hss_i, p.Hss_j, p.Hss_v, 1.0, Hs, 0)
    q = getattr(p, "q_lin", None)
    if q is None:
        q = np.zeros(n)
        q[:nxs // 2] = -0.5

    def model(x):
        Hx = np.concatenate([Hs * x[:nxs], p.Hdd @ x[nxs:]])
        return (0.5 * x @ Hx + q @ x, Hx + q, Jcs @ x[:nxs] + p.Jcd @ x[nxs:], Jds @ x[:nxs] + p.Jdd @ x[nxs:])
    return model, q
