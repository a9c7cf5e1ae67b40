"""ORACLE (test infrastructure only — never imported by hiop_amd/): the reference's gradient-based NLP scaling, restated.

hiopNlpFormulation::apply_scaling (src/Optimization/hiopNlpFormulation.cpp:671-714) decides, from the first evaluation at the
user's starting point, whether to scale at all (any |grad f|, |Jac_c| or |Jac_d| entry >= scaling_max_grad = 100); the factors are
those of hiopNLPObjGradScaling's constructor (src/Optimization/hiopNlpTransforms.cpp:423-499): one factor for the objective, one
per constraint ROW.  The scaled problem is what the algorithm sees (f, grad, c, d, Jacobian rows, constraint right-hand sides and
bounds, Hessian through obj_factor and the multipliers); the user gets the objective back unscaled (hiopNlpTransforms.hpp:387)."""
import numpy as np

SCALING_MAX_GRAD = 100.0   # hiopOptions.cpp: scaling_max_grad
SCALING_MIN_GRAD = 1e-8    # scaling_min_grad


def gradient_scaling(gradf, Jc, Jd, max_grad=SCALING_MAX_GRAD, min_grad=SCALING_MIN_GRAD):
    """(s_f, s_c, s_d) or None when apply_scaling returns false.  Jc, Jd: dense arrays (rows = constraints)."""
    Jc = np.asarray(Jc, dtype=np.float64).reshape(-1, gradf.size) if np.size(Jc) else np.zeros((0, gradf.size))
    Jd = np.asarray(Jd, dtype=np.float64).reshape(-1, gradf.size) if np.size(Jd) else np.zeros((0, gradf.size))
    g = float(np.abs(gradf).max()) if gradf.size else 0.0
    mc = float(np.abs(Jc).max()) if Jc.size else 0.0
    md = float(np.abs(Jd).max()) if Jd.size else 0.0
    if g < max_grad and mc < max_grad and md < max_grad:                      # :691-696
        return None
    s_f = max_grad / g if g > max_grad else 1.0                               # hiopNlpTransforms.cpp:441-446
    if min_grad > 0.0 and s_f < min_grad:
        s_f = min_grad

    def rows(J):                                                              # :473-490 (scaling_max_con_grad = 0)
        if J.shape[0] == 0:
            return np.zeros(0)
        r = np.abs(J).max(axis=1)
        if r.max() > max_grad:
            s = 1.0 / np.maximum(r / max_grad, 1.0)
        else:
            s = np.ones(J.shape[0])
        return np.maximum(s, min_grad) if min_grad > 0.0 else s               # :495-498
    return s_f, rows(Jc), rows(Jd)
