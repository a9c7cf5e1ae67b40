"""TEST INFRASTRUCTURE ONLY — restatement of the reference's Newton filter line-search interior-point loop,
hiopAlgFilterIPMNewton::run (src/Optimization/hiopAlgFilterIPM.cpp:2101-2770), on top of the full-space layer
(oracle/kkt_full.py) and the iterate/residual steps (oracle/ipm_slab.py).  Written so that the KKT systems the loop
produces can be compared with the systems the reference itself wrote (`write_kkt yes`): the purpose is to pin the
oracle's KKT rows (SURVEY.md §8 a/f1) on the reference's OWN trajectory, iteration by iteration, not only at x0.

What is restated (reference line per block below): the bounds relaxation of hiopBoundsRelaxer::relax, startingProcedure
with `duals_init zero`, evalNlpAndLogErrors (the sd / sc scaling), checkTermination, update_log_barrier_params, the
filter (hiopFilter) and accept_line_search_conditions, the backtracking loop with compute_safe_slacks, the post
line-search filter augmentation, hiopDualsNewtonLinearUpdate::go.
the second-order correction (apply_second_order_correction).
What is NOT restated (and raises if reached): feasibility restoration, the safe-mode switch of
the linear solver, elastic mode, NLP scaling (the example problems' gradients at x0 are below scaling_max_grad = 100).

The loop is written against an `ops` object so the same driver runs the numpy restatements (FilterOracleOps) and the
device operations of the GPU tests.

Parity pin: tests/test_oracle_reference_trajectory.py — with the MdsEx1 driver's options (NlpMdsEx1Driver.cpp:130-139) the
KKT matrices of iterations 0, 5 and 10 equal the matrices in tests/golden/kkt_linsys_{0,5,10}.iajaaa, the first
right-hand side of each equals the file's, the run takes the reference's 14 factorizations, and the final objective of
MdsEx1(400, 100) equals the driver's stored -selfcheck value to 1e-6 relative."""
import numpy as np

from . import hiop_oracle as ho
from . import ipm_slab as osl

# option defaults, src/Utils/hiopOptions.cpp:560-720
DEFAULTS = dict(mu0=1.0, tolerance=1e-8, kappa_mu=0.2, theta_mu=1.5, kappa_eps=10.0, tau_min=0.99, kappa1=1e-2, kappa2=1e-2,
                smax=100.0, kappa_d=1e-5, eta_phi=1e-8, gamma_theta=1e-5, gamma_phi=1e-8, s_theta=1.1, s_phi=2.3, delta=1.0,
                theta_max_fact=1e4, theta_min_fact=1e-4, dual_tol=1.0, cons_tol=1e-4, comp_tol=1e-4, rel_tolerance=0.0,
                acceptable_tolerance=1e-6, acceptable_iterations=10, max_iter=3000, min_step_size=1e-16, kappa_Sigma=1e10,
                bound_relax_perturb=1e-8, duals_lsq_ini_max=1e3, recalc_lsq_duals_tol=1e-6, max_soc_iter=4,
                kappa_soc=0.99,
                # not a reference option.  False (default): after a step accepted through the second-order correction, filter.add and the
                # LSQ-duals test keep the FIRST trial point's theta: the reference passes theta_trial to apply_second_order_correction BY
                # VALUE (hiopAlgFilterIPM.cpp:2949-2973, call :2561-2570; infeas_nrm_trial :2537 is set in the outer loop only).
                # True: the corrected point's theta (what rounds 3-4 of this restatement used).
                soc_theta_corrected=False)


def relax_bounds(xl, xu, dl, du, rel):
    """hiopBoundsRelaxer::relax (src/Optimization/hiopNlpTransforms.cpp:366-389): every entry, 'infinite' ones included
    (they stay beyond +-1e20)."""
    r = lambda b, s: b + s * rel * np.maximum(np.abs(b), 1.0)
    return r(xl, -1.0), r(xu, 1.0), r(dl, -1.0), r(du, 1.0)


class Filter:                                                       # src/Optimization/hiopFilter.hpp:60-75, .cpp:55-68
    def __init__(self):
        self.entries = []

    def initialize(self, theta_max):
        self.entries = [(theta_max, -1e20)]

    def add(self, theta, phi):
        self.entries.insert(0, (theta, phi))

    def contains(self, theta, phi):
        return any(theta >= t and phi >= p for t, p in self.entries)


class FilterOracleOps:
    """numpy operations of one iteration; `model(x)` -> f, grad, c, d (constant Jacobians / Hessian in the provider)."""

    def __init__(self, full, bounds, model, kappa_d=1e-5, kappa_sigma=1e10):
        self.full, self.bounds, self.model = full, bounds, model
        self.kappa_d, self.kappa_sigma = kappa_d, kappa_sigma
        p = full.p
        self.n_complem = int(full.ixl.sum() + full.ixu.sum() + full.idl.sum() + full.idu.sum())   # hiopNlpFormulation.hpp:244
        self.m = p.nyc + p.nyd

    # ---- startingProcedure (hiopAlgFilterIPM.cpp:290-425), duals_init = zero, no warm start
    def start(self, x0, mu0, kappa1, kappa2):
        full = self.full
        xl, xu, dl, du, _ = self.bounds
        x = x0.copy()
        ho.project_into_bounds(x, xl, full.ixl, xu, full.ixu, kappa1, kappa2)          # :355
        d = self.model(x)[3].copy()                                                    # :362, :374
        ho.project_into_bounds(d, dl, full.idl, du, full.idu, kappa1, kappa2)          # :378
        p = full.p
        it = {"x": x, "d": d, "yc": np.zeros(p.nyc), "yd": np.zeros(p.nyd)}
        for k in ("sxl", "sxu", "zl", "zu"):
            it[k] = np.zeros(p.nx)
        for k in ("sdl", "sdu", "vl", "vu"):
            it[k] = np.zeros(p.nd)
        osl.determine_slacks(full, it, self.bounds)                                    # :380 compute_safe_slacks
        if osl.adjust_small_slacks(full, it, it, self.bounds, mu0) > 0:
            raise NotImplementedError("adjust_bounds at the starting point")
        it["zl"], it["zu"], it["vl"], it["vu"] = full.ixl.copy(), full.ixu.copy(), full.idl.copy(), full.idu.copy()   # :390
        return it

    def copy(self, it):
        return {k: v.copy() for k, v in it.items()}

    def primal(self, it):
        return it["x"].copy()

    def evaluate(self, it):
        return self.model(it["x"])

    def residual(self, it, ev, mu):
        self.full.it = it
        r, n = osl.residual_update(self.full, it, ev[2], ev[3], ev[1], self.bounds, mu, self.kappa_d)
        return r, n

    def dual_norms(self, it):                                                          # hiopIterate.cpp:239-257
        bnd = ho.onenorm(it["zl"]) + ho.onenorm(it["zu"]) + ho.onenorm(it["vl"]) + ho.onenorm(it["vu"])
        return ho.onenorm(it["yc"]) + ho.onenorm(it["yd"]), bnd

    def logbar(self, it, f, mu):                                                       # hiopLogBarProblem.hpp:94-113, :128-129
        v = f - mu * osl.eval_log_barrier(self.full, it)
        if self.kappa_d > 0:
            v += osl.linear_damping_term(self.full, it, mu, self.kappa_d)
        return float(v)

    def grad_phi_dx(self, it, dr, grad_f, mu):                                         # hiopLogBarProblem.hpp:91-117, :149-156
        full = self.full
        gx = grad_f.copy()
        gd = np.zeros_like(it["d"])
        ho.add_log_barrier_grad(gx, -mu, it["sxl"], full.ixl)                          # hiopIterate.cpp:539-550
        ho.add_log_barrier_grad(gx, mu, it["sxu"], full.ixu)
        ho.add_log_barrier_grad(gd, -mu, it["sdl"], full.idl)
        ho.add_log_barrier_grad(gd, mu, it["sdu"], full.idu)
        if self.kappa_d > 0:
            ho.add_linear_damping_term(gx, full.ixl, full.ixu, 1.0, self.kappa_d * mu)  # hiopIterate.cpp:568-588
            ho.add_linear_damping_term(gd, full.idl, full.idu, 1.0, self.kappa_d * mu)
        return float(dr["x"] @ gx + dr["d"] @ gd)

    def kkt_update(self, it, mu):
        self.full.perturb.set_mu(mu)
        self.mu = mu
        return self.full.update(it)

    def directions(self, resid):
        ok, d, info = self.full.compute_directions_w_IR(resid, self.mu)
        return ok, d

    def fraction_to_the_bdry(self, it, d, tau):
        return osl.fraction_to_the_bdry(self.full, it, d, tau)

    def trial_primals(self, it, d, ap, ad, mu):                                        # :2527-2528
        trial = osl.take_step(it, d, ap, ad, primals=True, duals=False)
        osl.determine_slacks(self.full, trial, self.bounds)                            # compute_safe_slacks, hiopIterate.cpp:293-304
        nadj = osl.adjust_small_slacks(self.full, trial, it, self.bounds, mu)
        return trial, nadj

    def theta(self, it, c, d):                                                         # hiopResidual.cpp:101-115
        return ho.onenorm(self.bounds[4] - c) + ho.onenorm(it["d"] - d)

    def duals_update(self, it, trial, d, ap, ad, mu):                                  # hiopDualsUpdater.hpp:412-431
        out = osl.take_step(it, d, ap, ad, primals=False, duals=True, out=trial)
        osl.adjust_duals_plh(self.full, out, mu, self.kappa_sigma)
        return out

    def n_refactorizations(self):
        return self.full.num_refact

    # ---- second-order correction
    def c_resid(self, c):
        return self.bounds[4] - c

    def d_resid(self, it, d):
        return it["d"] - d

    def soc_resid(self, resid, c_soc, d_soc):
        r = dict(resid)
        r["ryc"], r["ryd"] = c_soc, d_soc
        return r

    def directions_no_ir(self, resid):
        return self.full.compute_directions(resid)

    # ---- quasi-Newton variant
    def duals_lsq(self, it, grad_f):                                                   # hiopDualsUpdater.cpp:239-330
        self.full.it = it
        ok, yc, yd = osl.duals_lsq_update(self.full, it, grad_f)
        if ok:
            it["yc"], it["yd"] = yc, yd
        return ok

    def dual_norms_inf(self, it):
        return max(ho.infnorm(it["yc"]), ho.infnorm(it["yd"])), None

    def zero_eq_duals(self, it):
        it["yc"][:] = 0.0
        it["yd"][:] = 0.0

    def hess_update(self, it, ev):
        """hiopHessianLowRank::update(it_curr, grad_f, Jac_c, Jac_d); set by the test for the low-rank provider."""
        raise NotImplementedError


def _errors(ops, it, norms, o):
    """evalNlpAndLogErrors, hiopAlgFilterIPM.cpp:636-712."""
    eq, bou = ops.dual_norms(it)
    n, m = ops.n_complem, ops.m
    smax = o["smax"]
    sd = min(max(smax, (bou + eq) / (n + m)) / smax, 1e8)
    sc = 0.0 if n == 0 else min(max(smax, bou / n) / smax, 1e8)
    e = dict(optim=norms["nrmInf_nlp_optim"], feas=norms["nrmInf_nlp_feasib"], complem=norms["nrmInf_nlp_complem"],
             cons_violation=norms["nrmInf_cons_violation"])
    e["nlp"] = max(e["optim"] / sd, e["cons_violation"], e["complem"] / sc)
    e["log"] = max(norms["nrmInf_bar_optim"] / sd, e["cons_violation"], norms["nrmInf_bar_complem"] / sc)
    return e


def _accept(ops, o, filt, theta, theta_trial, ap, f_logbar, f_logbar_trial, theta_min, gpd, it, dr, ev, mu):
    """accept_line_search_conditions, hiopAlgFilterIPM.cpp:2852-2944; returns (lsStatus, grad_phi_dx or None if not computed)."""
    suff = theta_trial <= (1 - o["gamma_theta"]) * theta or f_logbar_trial <= f_logbar - o["gamma_phi"] * theta
    if theta >= theta_min:
        st = 1 if suff else 0
    else:
        if gpd is None:
            gpd = ops.grad_phi_dx(it, dr, ev[1], mu)
        if gpd < 0.0 and ap * (-gpd) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"]:
            st = 3 if f_logbar_trial <= f_logbar + o["eta_phi"] * ap * gpd else 0
        else:
            st = 2 if suff else 0
    if st > 0 and filt.contains(theta_trial, f_logbar_trial):
        st = 0
    return st, gpd


def solve(ops, x0, on_kkt=None, table=None, quasi_newton=False, lsq_duals=None, **options):
    """quasi_newton=True: hiopAlgFilterIPMQuasiNewton::run (hiopAlgFilterIPM.cpp:960-1480) — the same loop with the secant
    update of the Hessian before every KKT update (:1212), `duals_init lsq` at the start and the LSQ duals update after the
    line search (hiopDualsLsqUpdate::go) — ops must provide hess_update(it, ev) and duals_lsq(it, grad_f).
    lsq_duals (default: = quasi_newton): `duals_init lsq` + `duals_update_type lsq`; False with quasi_newton=True is the option set
    of the reference's dense C interface (chiopInterface.cpp:133-135: quasi-Newton Hessian, linear duals, zero initial duals).
    Returns dict(x, obj, iters, status, n_fact).  `on_kkt(iter_num, it, mu, resid)` is called after every successful
    kkt update (the point at which the reference writes kkt_linsys_<iter>.iajaaa, hiopKKTLinSysCompressedMDSXYcYd via
    hiopKKTLinSys.cpp `write_linsys_counter_`)."""
    o = dict(DEFAULTS)
    o.update(options)
    eps_tol = o["tolerance"]
    mu = o["mu0"]
    tau = max(o["tau_min"], 1.0 - mu)                                   # hiopAlgFilterIPM.cpp:255 reload_options
    it = ops.start(x0, mu, o["kappa1"], o["kappa2"])
    ev = ops.evaluate(it)
    if lsq_duals is None:
        lsq_duals = quasi_newton
    if lsq_duals:                                                       # compute_initial_duals_eq, hiopDualsUpdater.hpp:154-186
        ok = ops.duals_lsq(it, ev[1])
        eq, _ = ops.dual_norms_inf(it)
        if not ok or eq > o["duals_lsq_ini_max"]:
            ops.zero_eq_duals(it)
    f_logbar = ops.logbar(it, ev[0], mu)                                # :2145
    resid, norms = ops.residual(it, ev, mu)                             # :2148
    theta_max = o["theta_max_fact"] * max(1.0, norms["nrmOne_nlp_feasib"])   # :2157-2158
    theta_min = o["theta_min_fact"] * max(1.0, norms["nrmOne_nlp_feasib"])
    filt = Filter()                                                     # cleared, :287; (re)initialized only at mu updates
    iter_num = 0
    ap = ad = 0.0
    e0 = None
    n_accep = 0
    n_fact = 0
    ls_status, ls_num, use_soc = -1, 0, 0
    status = "pending"
    while True:
        e = _errors(ops, it, norms, o)                                  # :2219
        if table is not None:
            table.append(dict(iter=iter_num, objective=float(ev[0]), inf_pr=float(e["feas"]), inf_du=float(e["optim"]), mu=float(mu),
                              alpha_du=float(ad), alpha_pr=float(ap), ls=ls_status, ls_num=ls_num, use_soc=use_soc))
        if e0 is None:
            e0 = e
        # ---- checkTermination, :814-845
        if e["nlp"] <= eps_tol and e["optim"] <= o["dual_tol"] and e["cons_violation"] <= o["cons_tol"] and e["complem"] <= o["comp_tol"]:
            status = "Solve_Success"
            break
        if iter_num >= o["max_iter"]:
            status = "Max_Iter_Exceeded"
            break
        rt = o["rel_tolerance"]
        if rt > 0 and e["optim"] <= rt * e0["optim"] and e["feas"] <= rt * e0["feas"] and \
                e["complem"] <= max(rt, 1e-6) * min(1.0, e0["complem"]):
            status = "Solve_Success_RelTol"
            break
        n_accep = n_accep + 1 if e["nlp"] <= o["acceptable_tolerance"] else 0
        if n_accep >= o["acceptable_iterations"]:
            status = "Solve_Acceptable_Level"
            break
        # ---- barrier update, :2291-2328 with update_log_barrier_params :556-567
        while e["log"] <= o["kappa_eps"] * mu:
            new_mu = max(0.0, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))
            new_mu = max(new_mu, min(eps_tol, o["comp_tol"]) / (10.0 + 1.0))
            if abs(new_mu - mu) < 1e-16:
                break
            mu = new_mu
            tau = max(o["tau_min"], 1.0 - mu)
            f_logbar = ops.logbar(it, ev[0], mu)
            resid, norms = ops.residual(it, ev, mu)
            e = _errors(ops, it, norms, o)
            filt.initialize(theta_max)                                  # :2321
        # ---- search direction, :2333-2462 (linsol_mode = stable semantics of the layer: no mode switch restated)
        if quasi_newton:
            ops.hess_update(it, ev)                                     # :1212
        if not ops.kkt_update(it, mu):
            raise RuntimeError("KKT update failed (inertia correction exhausted)")
        n_fact += 1 + ops.n_refactorizations()
        if on_kkt is not None:
            on_kkt(iter_num, it, mu, resid)
        ok, dr = ops.directions(resid)
        if not ok:
            raise RuntimeError("compute_directions_w_IR failed")
        # ---- backtracking line search, :2477-2588
        ap, ad = ops.fraction_to_the_bdry(it, dr, tau)
        theta = norms["nrmOne_nlp_feasib"]                              # resid->get_theta()
        ls_status, ls_num, use_soc = 0, 0, 0
        gpd = None
        ini_step = True
        while True:
            if not ini_step and ap < o["min_step_size"]:
                if quasi_newton:                                        # :1289-1297: solver_status_ = Steplength_Too_Small, the run ends
                    status = "Steplength_Too_Small"                     # (:1442-1444) at the current iterate
                    break
                raise NotImplementedError("minimum step size reached (feasibility restoration is not restated)")
            trial, nadj = ops.trial_primals(it, dr, ap, ad, mu)
            ev_t = ops.evaluate(trial)                                  # functions only in the reference
            f_logbar_trial = ops.logbar(trial, ev_t[0], mu)
            theta_trial = ops.theta(trial, ev_t[2], ev_t[3])
            ls_num += 1
            ls_status, gpd = _accept(ops, o, filt, theta, theta_trial, ap, f_logbar, f_logbar_trial, theta_min, gpd, it, dr, ev, mu)
            if ls_status > 0:
                break
            if ini_step and theta <= theta_trial and o["max_soc_iter"] > 0:
                # ---- apply_second_order_correction, :2949-3038
                theta_last, th, ap_soc, num_soc, st = 0.0, theta_trial, ap, 0, 0
                gpd_soc = None
                c_soc, d_soc = ops.c_resid(ev[2]), ops.d_resid(it, ev[3])
                while num_soc < o["max_soc_iter"] and (num_soc == 0 or th <= o["kappa_soc"] * theta_last):
                    theta_last = th
                    c_soc = ap_soc * c_soc + ops.c_resid(ev_t[2])
                    d_soc = ap_soc * d_soc + ops.d_resid(trial, ev_t[3])
                    r_soc = ops.soc_resid(resid, c_soc, d_soc)                   # hiopResidual::update_soc, hiopResidual.cpp:425-600
                    ok, dr_soc = ops.directions_no_ir(r_soc)                     # kkt->computeDirections
                    if not ok:
                        raise RuntimeError("computeDirections failed in the second-order correction")
                    ap_soc, _ = ops.fraction_to_the_bdry(it, dr_soc, tau)
                    trial, nadj = ops.trial_primals(it, dr_soc, ap_soc, ap_soc, mu)
                    ev_t = ops.evaluate(trial)
                    f_logbar_trial = ops.logbar(trial, ev_t[0], mu)
                    th = ops.theta(trial, ev_t[2], ev_t[3])
                    st, gpd_soc = _accept(ops, o, filt, theta, th, ap, f_logbar, f_logbar_trial, theta_min, gpd_soc, it, dr, ev, mu)
                    if st > 0:
                        break
                    num_soc += 1
                if st > 0:
                    ls_status, ap, dr, resid, gpd, use_soc = st, ap_soc, dr_soc, r_soc, gpd_soc, 1
                    if o["soc_theta_corrected"]:
                        theta_trial = th
                    break
            ap *= 0.5
            ini_step = False
        if status == "Steplength_Too_Small":
            break
        if nadj > 0:
            raise NotImplementedError("adjust_bounds after small slacks")                   # :2592-2603
        # ---- filter augmentation, :2616-2653
        if ls_status == 1:
            if gpd is None:
                gpd = ops.grad_phi_dx(it, dr, ev[1], mu)
            if gpd < 0 and ap * (-gpd) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"]:
                if not (f_logbar_trial <= f_logbar + o["eta_phi"] * ap * gpd):
                    filt.add(theta_trial, f_logbar_trial)
            else:
                filt.add(theta_trial, f_logbar_trial)
        elif ls_status == 2:
            filt.add(theta_trial, f_logbar_trial)
        iter_num += 1
        # ---- duals, then the accepted trial becomes the iterate, :2714-2754
        it = ops.duals_update(it, trial, dr, ap, ad, mu)
        if lsq_duals and theta_trial <= o["recalc_lsq_duals_tol"]:   # hiopDualsUpdater.cpp:118-149: with the gradient and the
            if not ops.duals_lsq(it, ev[1]):                            # Jacobians of the PREVIOUS iterate (they are re-evaluated
                raise RuntimeError("dual lsq update failed")            # only after `go`, hiopAlgFilterIPM.cpp:1448-1456)
        ev = ops.evaluate(it)
        f_logbar = ops.logbar(it, ev[0], mu)
        resid, norms = ops.residual(it, ev, mu)
    return dict(x=ops.primal(it), obj=float(ev[0]), iters=iter_num, status=status, n_fact=n_fact, mu=mu, err=e["nlp"], it=it)
