"""Shared builders for the full-space KKT layer tests (oracle side, numpy only)."""
import numpy as np

from oracle import hiop_oracle as ho
from oracle import kkt_full as kf
from oracle import problems


def random_iterate(nx, nd, nyc, nyd, ixl, ixu, idl, idu, seed=3, mu=0.1):
    """An interior primal-dual iterate: slacks and bound duals positive where the bound exists, 0 elsewhere
    (what hiopIterate holds, src/Optimization/hiopIterate.cpp)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    it = {"x": rng.uniform(-1, 1, nx), "d": rng.uniform(-1, 1, nd), "yc": rng.uniform(-1, 1, nyc),
          "yd": rng.uniform(-1, 1, nyd)}
    for s, z, pat, n in (("sxl", "zl", ixl, nx), ("sxu", "zu", ixu, nx), ("sdl", "vl", idl, nd), ("sdu", "vu", idu, nd)):
        sv = rng.uniform(0.05, 2.0, n)
        it[s] = sv * pat
        it[z] = (mu / sv) * rng.uniform(0.5, 1.5, n) * pat
    return it


def random_resid(sizes, ixl, ixu, idl, idu, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    r = {k: rng.uniform(-1, 1, s) for k, s in zip(kf.RESID_PARTS, sizes)}
    for k, pat in (("rxl", ixl), ("rszl", ixl), ("rxu", ixu), ("rszu", ixu), ("rdl", idl), ("rsvl", idl),
                   ("rdu", idu), ("rsvu", idu)):
        r[k] *= pat
    return r


def negative_curvature_resid(sizes, idx=1):
    """rx = e_idx, everything else 0: pushes the Newton direction along the coordinate made nonconvex by dense_case."""
    r = {k: np.zeros(s) for k, s in zip(kf.RESID_PARTS, sizes)}
    r["rx"][idx] = 1.0
    return r


def patterns(p):
    f = lambda b: b.astype(np.float64)
    return f(p.xl > -1e20), f(p.xu < 1e20), f(p.dl > -1e20), f(p.du < 1e20)


def oracle_mds(p, nonconvex=False):
    k = ho.KKTLinSysCompressedMDSXYcYd(p.nxs, p.nxd, p.neq, p.nineq, (p.Jcs_i, p.Jcs_j), (p.Jds_i, p.Jds_j),
                                       (p.Hss_i, p.Hss_j))
    Hdd = p.Hdd.copy()
    if nonconvex:
        Hdd[0, 0] = -3.0          # one negative direction in the dense block -> wrong inertia at delta = 0
    k.set_values(p.Jcs_v, p.Jds_v, p.Hss_v, p.Jcd, p.Jdd, Hdd, None, None)
    return k


def zero_equality_row(k, row):
    k.Jcd = k.Jcd.copy()
    k.Jcd[row, :] = 0.0
    k.Jcs_val = np.where(k.Jcs_ij[0] == row, 0.0, k.Jcs_val)


def mds_case(ns=8, nd=6, neq=None, nonconvex=False, seed=3):
    p = problems.mds_ex1(ns, nd) if neq is None else problems.mds_ex1_g(ns, nd, neq)
    ixl, ixu, idl, idu = patterns(p)
    k = oracle_mds(p, nonconvex)
    prov = kf.MdsProvider(k)
    full = kf.KKTLinSysFull(prov, ixl, ixu, idl, idu)
    it = random_iterate(prov.nx, prov.nd, prov.nyc, prov.nyd, ixl, ixu, idl, idu, seed)
    return p, k, full, it


def dense_case(nx=12, neq=3, nineq=4, seed=11, nonconvex=False, xd_form=False, inertia_free=False, neg_value=-2.0,
               free_nonconvex=False):
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.uniform(-1, 1, (nx, nx))
    H = A @ A.T / nx + np.eye(nx)
    if nonconvex:
        H[1, 1] = neg_value
    Jc = rng.uniform(-1, 1, (neq, nx))
    Jd = rng.uniform(-1, 1, (nineq, nx))
    ixl = (rng.uniform(0, 1, nx) < 0.6).astype(np.float64)
    ixu = (rng.uniform(0, 1, nx) < 0.4).astype(np.float64)
    idl = np.ones(nineq)
    idu = (rng.uniform(0, 1, nineq) < 0.5).astype(np.float64)
    if free_nonconvex:            # no bounds on the negative-curvature variable: Dx[1] = 0
        ixl[1] = ixu[1] = 0.0
    prov = kf.DenseXDYcYdProvider(H, Jc, Jd) if xd_form else kf.DenseXYcYdProvider(H, Jc, Jd)
    full = kf.KKTLinSysFull(prov, ixl, ixu, idl, idu, inertia_free=inertia_free)
    it = random_iterate(nx, nineq, neq, nineq, ixl, ixu, idl, idu, seed)
    return (H, Jc, Jd, ixl, ixu, idl, idu), full, it
