"""TEST INFRASTRUCTURE ONLY — numpy restatement of the hiopIterate / hiopResidual steps either side of the KKT solve
(SURVEY.md §8 f1), written with the oracle's hiopVector functions, one reference line per statement.

  hiopResidual::update                        src/Optimization/hiopResidual.cpp:154-365
  hiopIterate::determineSlacks                src/Optimization/hiopIterate.cpp:274-291
  hiopIterate::adjust_small_slacks            :414-505
  hiopIterate::determineDualsBounds_d         :314-327
  hiopIterate::fractionToTheBdry              :330-362
  hiopIterate::takeStep_primals / _duals      :367-390
  hiopIterate::adjustDuals_primalLogHessian   :507-521
  hiopIterate::evalLogBarrier / linearDampingTerm   :523-566
Iterates and residuals are dicts keyed like oracle/kkt_full.py (ITER_PARTS / RESID_PARTS).
Parity pin: these are compositions of hiopVector methods that are individually pinned by the reference's LinAlg unit
tests (tests/golden/reference_unit_tests.json); the composition is checked through the oracle IPM's selfcheck run.
"""
import numpy as np

from . import hiop_oracle as ho


def residual_update(full, it, c, d, grad_f, bounds, mu, kappa_d):
    """full: oracle.kkt_full.KKTLinSysFull (patterns + Jacobian products); bounds = (xl, xu, dl, du, crhs).
    Returns (resid dict, norms dict)."""
    xl, xu, dl, du, crhs = bounds
    p, ixl, ixu, idl, idu = full.p, full.ixl, full.ixu, full.idl, full.idu
    n = {k: 0.0 for k in ("nrmInf_nlp_optim", "nrmInf_nlp_feasib", "nrmInf_nlp_complem", "nrmInf_bar_optim",
                          "nrmInf_bar_feasib", "nrmInf_bar_complem", "nrmOne_nlp_feasib", "nrmOne_bar_feasib",
                          "nrmOne_nlp_optim", "nrmOne_bar_optim", "nrmInf_cons_violation")}
    r = {}
    rx = grad_f.copy()                                                             # :181
    rx += p.jac_trans_times_vec("c", it["yc"])                                     # :182
    rx += p.jac_trans_times_vec("d", it["yd"])                                     # :183
    rx += -it["zl"] + it["zu"]                                                     # :185-186
    n["nrmInf_nlp_optim"] = max(n["nrmInf_nlp_optim"], ho.infnorm(rx))             # :187-188
    n["nrmOne_nlp_optim"] += ho.onenorm(rx)                                        # :189
    if kappa_d > 0:
        ho.add_linear_damping_term(rx, ixl, ixu, 1.0, kappa_d * mu * 1.0)          # :192 (hiopIterate.cpp:568-577)
    rx = -rx                                                                       # :193
    n["nrmInf_bar_optim"] = max(n["nrmInf_bar_optim"], ho.infnorm(rx))
    n["nrmOne_bar_optim"] += ho.onenorm(rx)
    r["rx"] = rx
    rd = it["yd"] + it["vl"] - it["vu"]                                            # :203-205
    n["nrmInf_nlp_optim"] = max(n["nrmInf_nlp_optim"], ho.infnorm(rd))
    n["nrmOne_nlp_optim"] += ho.onenorm(rd)
    if kappa_d > 0:
        ho.add_linear_damping_term(rd, idl, idu, 1.0, kappa_d * mu * -1.0)         # :212
    n["nrmInf_bar_optim"] = max(n["nrmInf_bar_optim"], ho.infnorm(rd))
    n["nrmOne_bar_optim"] += ho.onenorm(rd)
    r["rd"] = rd
    ryc = crhs - c                                                                 # :214-215
    n["nrmInf_nlp_feasib"] = max(n["nrmInf_nlp_feasib"], ho.infnorm(ryc))
    n["nrmOne_nlp_feasib"] += ho.onenorm(ryc)
    n["nrmInf_cons_violation"] = max(n["nrmInf_cons_violation"], ho.infnorm(ryc))
    r["ryc"] = ryc
    if d.size:
        if np.any(idl == 1.0):
            a = ho.vmin_w_pattern(d - dl, idl)                                     # :219-223
            n["nrmInf_cons_violation"] = max(n["nrmInf_cons_violation"], -a if a < 0 else 0.0)
        if np.any(idu == 1.0):
            a = ho.vmin_w_pattern(du - d, idu)                                     # :224-227
            n["nrmInf_cons_violation"] = max(n["nrmInf_cons_violation"], -a if a < 0 else 0.0)
    ryd = it["d"] - d                                                              # :229-230
    n["nrmInf_nlp_feasib"] = max(n["nrmInf_nlp_feasib"], ho.infnorm(ryd))
    n["nrmOne_nlp_feasib"] += ho.onenorm(ryd)
    r["ryd"] = ryd
    sel = lambda v, pat: np.where(pat == 0.0, 0.0, v)
    r["rxl"] = sel(it["x"] - it["sxl"] - xl, ixl)                                  # :236-243
    r["rxu"] = sel(xu - it["x"] - it["sxu"], ixu)                                  # :248-254
    r["rdl"] = sel(it["d"] - it["sdl"] - dl, idl)                                  # :260-262
    r["rdu"] = sel(du - it["sdu"] - it["d"], idu)                                  # :268-272
    n["nrmInf_bar_feasib"] = n["nrmInf_nlp_feasib"]                                # :279
    n["nrmOne_bar_feasib"] = n["nrmOne_nlp_feasib"]                                # :280
    for key, s, z, pat in (("rszl", "sxl", "zl", ixl), ("rszu", "sxu", "zu", ixu), ("rsvl", "sdl", "vl", idl),
                           ("rsvu", "sdu", "vu", idu)):                            # :283-345
        v = sel(-it[s] * it[z], pat)
        n["nrmInf_nlp_complem"] = max(n["nrmInf_nlp_complem"], ho.infnorm(v))
        v = v + np.where(pat == 1.0, mu, 0.0)
        n["nrmInf_bar_complem"] = max(n["nrmInf_bar_complem"], ho.infnorm(v))
        r[key] = v
    return r, n


NORM_ORDER = ("nrmInf_nlp_optim", "nrmInf_nlp_feasib", "nrmInf_nlp_complem", "nrmInf_bar_optim", "nrmInf_bar_feasib",
              "nrmInf_bar_complem", "nrmOne_nlp_feasib", "nrmOne_bar_feasib", "nrmOne_nlp_optim", "nrmOne_bar_optim",
              "nrmInf_cons_violation")


def fraction_to_the_bdry(full, it, dr, tau):                                       # :330-362
    ap = ad = 10.0
    for s, pat in (("sxl", full.ixl), ("sxu", full.ixu), ("sdl", full.idl), ("sdu", full.idu)):
        ap = min(ap, ho.fraction_to_the_bdry_w_pattern(it[s], dr[s], tau, pat))
    for s, pat in (("zl", full.ixl), ("zu", full.ixu), ("vl", full.idl), ("vu", full.idu)):
        ad = min(ad, ho.fraction_to_the_bdry_w_pattern(it[s], dr[s], tau, pat))
    return ap, ad


def take_step(it, dr, alpha_primal, alpha_dual, primals=True, duals=True, out=None):   # :367-390
    out = {k: v.copy() for k, v in (out if out is not None else it).items()}
    if primals:
        for k in ("x", "d"):
            out[k] = it[k] + alpha_primal * dr[k]
    if duals:
        for k in ("yc", "yd"):
            out[k] = it[k] + alpha_primal * dr[k]
        for k in ("zl", "zu", "vl", "vu"):
            out[k] = it[k] + alpha_dual * dr[k]
    return out


def determine_slacks(full, it, bounds):                                            # :274-291
    xl, xu, dl, du, _ = bounds
    sel = lambda v, pat: np.where(pat == 0.0, 0.0, v)
    it["sxl"] = sel(it["x"] - xl, full.ixl)
    it["sxu"] = sel(xu - it["x"], full.ixu)
    it["sdl"] = sel(it["d"] - dl, full.idl)
    it["sdu"] = sel(du - it["d"], full.idu)


def _adjust_small_slacks_one(slack, bound, slack_dual, select, mu):               # :414-480
    if slack.size == 0:
        return 0
    eps = np.finfo(np.float64).eps
    small_val = eps * min(1.0, mu)
    scale_fact = eps ** 0.75
    if not (ho.vmin_w_pattern(slack, select) < small_val):
        return 0
    arg1 = slack.copy()
    arg1[select == 1.0] += -small_val                                              # addConstant_w_patternSelect
    arg1 = np.minimum(arg1, 0.0)                                                   # component_min(0)
    num = int(np.sum(arg1 < 0.0))
    ho.component_sgn(arg1)
    arg1 *= -1.0
    slack[:] = np.maximum(slack, 0.0)                                              # component_max(0)
    arg2 = np.empty_like(slack)
    ho.set_to_constant_w_pattern(arg2, mu, select)
    ho.component_div_w_pattern(arg2, slack_dual, select)
    arg3 = np.empty_like(slack)
    ho.set_to_constant_w_pattern(arg3, small_val, select)
    arg2 = np.maximum(arg2, arg3) - slack
    arg1 = arg1 * arg2 + slack
    ho.set_to_constant_w_pattern(arg2, 1.0, select)
    arg2 = np.maximum(arg2, np.abs(bound)) * scale_fact + slack
    slack[:] = np.minimum(arg1, arg2)
    return num


def adjust_small_slacks(full, it, it_curr, bounds, mu):                            # :483-505
    xl, xu, dl, du, _ = bounds
    n = 0
    n += _adjust_small_slacks_one(it["sxl"], xl, it_curr["zl"], full.ixl, mu)
    n += _adjust_small_slacks_one(it["sxu"], xu, it_curr["zu"], full.ixu, mu)
    n += _adjust_small_slacks_one(it["sdl"], dl, it_curr["vl"], full.idl, mu)
    n += _adjust_small_slacks_one(it["sdu"], du, it_curr["vu"], full.idu, mu)
    return n


def determine_duals_bounds_d(full, it, mu):                                        # :314-327
    with np.errstate(divide="ignore", invalid="ignore"):
        it["vl"] = np.where(full.idl == 0.0, 0.0, mu / it["sdl"])
        it["vu"] = np.where(full.idu == 0.0, 0.0, mu / it["sdu"])


def adjust_duals_plh(full, it, mu, kappa_sigma):                                   # :507-521
    ho.adjust_duals_plh(it["zl"], it["sxl"], full.ixl, mu, kappa_sigma)
    ho.adjust_duals_plh(it["zu"], it["sxu"], full.ixu, mu, kappa_sigma)
    ho.adjust_duals_plh(it["vl"], it["sdl"], full.idl, mu, kappa_sigma)
    ho.adjust_duals_plh(it["vu"], it["sdu"], full.idu, mu, kappa_sigma)


def eval_log_barrier(full, it):                                                    # :523-540
    return (ho.log_barrier(it["sxl"], full.ixl) + ho.log_barrier(it["sxu"], full.ixu) + ho.log_barrier(it["sdl"], full.idl) +
            ho.log_barrier(it["sdu"], full.idu))


def linear_damping_term(full, it, mu, kappa_d):                                    # :552-566
    return (ho.linear_damping_term(it["sxl"], full.ixl, full.ixu, mu, kappa_d) +
            ho.linear_damping_term(it["sxu"], full.ixu, full.ixl, mu, kappa_d) +
            ho.linear_damping_term(it["sdl"], full.idl, full.idu, mu, kappa_d) +
            ho.linear_damping_term(it["sdu"], full.idu, full.idl, mu, kappa_d))


def duals_lsq_update(full, it, grad_f):
    """hiopDualsLsqUpdateLinsysRedDense::do_lsq_update (src/Optimization/hiopDualsUpdater.cpp:239-330): returns
    (ok, yc, yd).  M = [Jc Jc^T, Jc Jd^T; Jd Jc^T, Jd Jd^T + I] (Cholesky, DPOTRF/DPOTRS in the reference)."""
    p = full.p
    me, mi = p.nyc, p.nyd
    m = me + mi
    # J as a dense matrix through the provider's products (columns of the identity) — small test sizes only
    J = np.zeros((m, p.nx))
    for r in range(m):
        e = np.zeros(m)
        e[r] = 1.0
        J[r] = p.jac_trans_times_vec("c", e[:me]) + p.jac_trans_times_vec("d", e[me:])
    M = J @ J.T                                                                    # :250-252
    M[me:, me:] += np.eye(mi)                                                      # :256
    vecx = grad_f - it["zl"] + it["zu"]                                            # :285-288
    vecd = it["vl"] - it["vu"]                                                     # :290-291
    rhs = -(J @ vecx)                                                              # :293-294
    rhs[me:] -= vecd                                                               # :295
    try:
        L = np.linalg.cholesky(M)
    except np.linalg.LinAlgError:
        return False, None, None
    sol = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
    return True, sol[:me].copy(), sol[me:].copy()
