"""A compact primal-dual interior-point driver over the ORACLE's KKT classes — TEST INFRASTRUCTURE.

Purpose: pin the oracle's problem data + KKT algebra end to end against the optimal objective values stored in the
reference drivers' `-selfcheck` (src/Drivers/MDS/NlpMdsEx1Driver.cpp:149, src/Drivers/Dense/NlpDenseConsEx1Driver.cpp:139).
It is NOT a restatement of hiopAlgFilterIPM (no filter, no second-order correction, no restoration): a monotone
barrier method with fraction-to-the-boundary steps, which is enough for the convex example problems.  It therefore
reaches the same optimum by a different iteration path: the stored values are reproduced to the accuracy of the
reference's own termination tolerance (1e-5 relative), not to the last digit.

Formulation (HiOp's, src/Optimization/hiopNlpFormulation.hpp):  min f(x)  s.t. c(x) = 0, d(x) - d = 0,
xl <= x <= xu, dl <= d <= du; Newton step reduced to the XYcYd system
    [H+Dx  Jc^T  Jd^T ] [dx ]   [rx ]          Dx = zl/sxl + zu/sxu,  Dd = vl/sdl + vu/sdu
    [Jc    0     0    ] [dyc] = [ryc]          (src/Optimization/hiopKKTLinSys.cpp:543-650)
    [Jd    0    -Dd^-1] [dyd]   [ryd]
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import hiop_oracle as ho


def _project(x, lo, has_lo, hi, has_hi, k1=1e-2, k2=1e-2):
    x = x.copy()
    ho.project_into_bounds(x, lo, has_lo.astype(float), hi, has_hi.astype(float), k1, k2)
    return x


def solve_mds(p, mu0=0.1, tol=1e-8, max_iter=200, verbose=False, kkt=None, trace=None):
    """p: oracle.problems.MdsProblem (linear constraints, quadratic objective: Jacobians and Hessian constant)."""
    nxs, nxd, neq, nineq = p.nxs, p.nxd, p.neq, p.nineq
    n = nxs + nxd
    Jcs = sp.csr_matrix((p.Jcs_v, (p.Jcs_i, p.Jcs_j)), shape=(neq, nxs))
    Jds = sp.csr_matrix((p.Jds_v, (p.Jds_i, p.Jds_j)), shape=(nineq, nxs))
    Hs = np.zeros(nxs)
    ho.spsym_add_diag_to_vec(p.Hss_i, p.Hss_j, p.Hss_v, 1.0, Hs, 0)
    Hd = p.Hdd
    q_lin = getattr(p, "q_lin", None)
    if q_lin is None:          # MdsEx1: 0.5*x_i*(x_i-1) on the first nxs/2 sparse variables
        q_lin = np.zeros(n)
        q_lin[:nxs // 2] = -0.5

    def f(x):
        xs, y = x[:nxs], x[nxs:]
        return 0.5 * xs @ (Hs * xs) + 0.5 * y @ (Hd @ y) + q_lin @ x

    def grad(x):
        g = np.concatenate([Hs * x[:nxs], Hd @ x[nxs:]])
        return g + q_lin

    cons_c = lambda x: Jcs @ x[:nxs] + p.Jcd @ x[nxs:]
    cons_d = lambda x: Jds @ x[:nxs] + p.Jdd @ x[nxs:]
    JcT = lambda y: np.concatenate([Jcs.T @ y, p.Jcd.T @ y])
    JdT = lambda y: np.concatenate([Jds.T @ y, p.Jdd.T @ y])

    ixl, ixu = p.xl > -1e20, p.xu < 1e20
    idl, idu = p.dl > -1e20, p.du < 1e20
    x = _project(p.x0, p.xl, ixl, p.xu, ixu)
    d = _project(cons_d(x), p.dl, idl, p.du, idu)
    yc, yd = np.zeros(neq), np.zeros(nineq)
    mu = mu0
    sl = lambda v, lo, m: np.where(m, v - lo, 1.0)
    su = lambda v, hi, m: np.where(m, hi - v, 1.0)
    zl = np.where(ixl, mu / sl(x, p.xl, ixl), 0.0)
    zu = np.where(ixu, mu / su(x, p.xu, ixu), 0.0)
    vl = np.where(idl, mu / sl(d, p.dl, idl), 0.0)
    vu = np.where(idu, mu / su(d, p.du, idu), 0.0)

    # `kkt`: any object with the oracle class' interface (set_values / build_kkt_matrix / factorize_with_curv_check /
    # solve_compressed) -- the GPU parity test passes an adapter over the HIP implementation here
    if kkt is None:
        kkt = ho.KKTLinSysCompressedMDSXYcYd(nxs, nxd, neq, nineq, (p.Jcs_i, p.Jcs_j), (p.Jds_i, p.Jds_j), (p.Hss_i, p.Hss_j))
    n_fact = 0
    for it in range(max_iter):
        sxl, sxu, sdl, sdu = sl(x, p.xl, ixl), su(x, p.xu, ixu), sl(d, p.dl, idl), su(d, p.du, idu)
        g = grad(x)
        r_dual_x = g + JcT(yc) + JdT(yd) - zl + zu
        r_dual_d = -yd - vl + vu
        r_c, r_d = cons_c(x), cons_d(x) - d

        def err(m):
            comp = max(np.abs(zl * sxl - m)[ixl].max(initial=0), np.abs(zu * sxu - m)[ixu].max(initial=0),
                       np.abs(vl * sdl - m)[idl].max(initial=0), np.abs(vu * sdu - m)[idu].max(initial=0))
            return max(np.abs(r_dual_x).max(), np.abs(r_dual_d).max(initial=0), np.abs(r_c).max(initial=0),
                       np.abs(r_d).max(initial=0), comp)
        if verbose:
            print(f"{it:3d} f={f(x): .10e} err0={err(0.0):.2e} mu={mu:.1e}")
        if trace is not None:
            trace.append((f(x), err(0.0), mu))
        if err(0.0) < tol:
            break
        while err(mu) < 10 * mu and mu > tol / 10:
            mu = max(tol / 10, min(0.2 * mu, mu ** 1.5))
        Dx = np.where(ixl, zl / sxl, 0.0) + np.where(ixu, zu / sxu, 0.0)
        Dd = np.where(idl, vl / sdl, 0.0) + np.where(idu, vu / sdu, 0.0)
        rx = -(g + JcT(yc) + JdT(yd)) + np.where(ixl, mu / sxl, 0.0) - np.where(ixu, mu / sxu, 0.0)
        rdd = yd + np.where(idl, mu / sdl, 0.0) - np.where(idu, mu / sdu, 0.0)
        ryc = -r_c
        ryd = -r_d + rdd / Dd
        kkt.set_values(p.Jcs_v, p.Jds_v, p.Hss_v, p.Jcd, p.Jdd, p.Hdd, Dx, Dd)
        kkt.build_kkt_matrix(0.0, 0.0, 0.0, 0.0)
        nneg = kkt.factorize_with_curv_check()
        n_fact += 1
        assert nneg == neq + nineq, f"wrong inertia {nneg}"
        ok, dx, dyc, dyd = kkt.solve_compressed(rx, ryc, ryd)
        assert ok
        dd = (dyd + rdd) / Dd
        dzl = np.where(ixl, mu / sxl - zl - zl / sxl * dx, 0.0)
        dzu = np.where(ixu, mu / sxu - zu + zu / sxu * dx, 0.0)
        dvl = np.where(idl, mu / sdl - vl - vl / sdl * dd, 0.0)
        dvu = np.where(idu, mu / sdu - vu + vu / sdu * dd, 0.0)
        tau = max(0.99, 1.0 - mu)
        ap = min(ho.fraction_to_the_bdry_w_pattern(sxl, dx, tau, ixl.astype(float)),
                 ho.fraction_to_the_bdry_w_pattern(sxu, -dx, tau, ixu.astype(float)),
                 ho.fraction_to_the_bdry_w_pattern(sdl, dd, tau, idl.astype(float)),
                 ho.fraction_to_the_bdry_w_pattern(sdu, -dd, tau, idu.astype(float)))
        ad = min(ho.fraction_to_the_bdry_w_pattern(zl, dzl, tau, ixl.astype(float)),
                 ho.fraction_to_the_bdry_w_pattern(zu, dzu, tau, ixu.astype(float)),
                 ho.fraction_to_the_bdry_w_pattern(vl, dvl, tau, idl.astype(float)),
                 ho.fraction_to_the_bdry_w_pattern(vu, dvu, tau, idu.astype(float)))
        x = x + ap * dx
        d = d + ap * dd
        yc = yc + ad * dyc
        yd = yd + ad * dyd
        zl, zu, vl, vu = zl + ad * dzl, zu + ad * dzu, vl + ad * dvl, vu + ad * dvu
    return dict(x=x, obj=f(x), iters=it, n_fact=n_fact, err=err(0.0))
