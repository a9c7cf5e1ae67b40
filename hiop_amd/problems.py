"""Synthetic problem generators: the reference's example drivers as DATA (numpy, host side) — inputs for the bench and the
parity tests, nothing here computes what the library computes.

MdsEx1      : the reference's MDS example, src/Drivers/MDS/NlpMdsEx1.hpp (sizes :109-113, bounds
              :115-160, Jacobian :294-400, Hessian :403-440, Qd/Md :79-88).
MdsEx1G     : the same problem family with the number of equalities decoupled from the number of
              sparse variables, needed to reach BASELINE config 3 (n_sparse=1e5, n_dense=4096, m=4096)
              which the stock driver cannot express (m = ns+3, n_sparse = 2 ns).
DenseConsEx1/2 : src/Drivers/Dense/NlpDenseConsEx1.{hpp,cpp}, NlpDenseConsEx2.{hpp,cpp}.
The MDS generators return the constant Jacobian / Hessian blocks in the layout hiopInterfaceMDS::eval_Jac_cons /
eval_Hess_Lagr hand to the solver (src/Interface/hiopInterface.hpp:645-657,744-761): sparse blocks as
row-sorted COO with int32 indices, dense blocks row-major.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class MdsProblem:
    name: str
    nxs: int
    nxd: int
    neq: int
    nineq: int
    Jcs_i: np.ndarray
    Jcs_j: np.ndarray
    Jcs_v: np.ndarray
    Jds_i: np.ndarray
    Jds_j: np.ndarray
    Jds_v: np.ndarray
    Hss_i: np.ndarray
    Hss_j: np.ndarray
    Hss_v: np.ndarray
    Jcd: np.ndarray      # neq x nxd
    Jdd: np.ndarray      # nineq x nxd
    Hdd: np.ndarray      # nxd x nxd (symmetric, upper triangle is what the KKT uses)
    xl: np.ndarray
    xu: np.ndarray
    dl: np.ndarray       # inequality lower bounds
    du: np.ndarray
    x0: np.ndarray

    @property
    def N(self):
        return self.nxd + self.neq + self.nineq


def _Qd(nd):
    Q = np.full((nd, nd), 1e-8)
    Q[np.arange(nd), np.arange(nd)] += 2.0
    for i in range(1, nd - 1):          # NlpMdsEx1.hpp:83-88 (note: starts at 1)
        Q[i, i + 1] = 1.0
        Q[i + 1, i] = 1.0
    return Q


def _ineq_rows(ns, empty_sp_row=False):
    # NlpMdsEx1.hpp:317-337: row0 = x_1 + e^T s ; row1 = x_2 (unless empty) ; row2 = x_3
    ii = [0] + [0] * ns
    jj = [0] + [ns + i for i in range(ns)]
    if not empty_sp_row:
        ii.append(1)
        jj.append(1)
    ii.append(2)
    jj.append(2)
    return np.array(ii, np.int32), np.array(jj, np.int32)


def mds_ex1(ns: int, nd: int, empty_sp_row: bool = False) -> MdsProblem:
    if ns % 4 != 0:
        ns = 4 * ((4 + ns) // 4)        # NlpMdsEx1.hpp:67-72
    assert ns >= 4
    ci = np.repeat(np.arange(ns, dtype=np.int32), 2)
    cj = np.empty(2 * ns, np.int32)
    cj[0::2] = np.arange(ns)
    cj[1::2] = np.arange(ns) + ns
    di, dj = _ineq_rows(ns, empty_sp_row)
    n = 2 * ns + nd
    xl = np.full(n, -1e20)
    xu = np.full(n, 1e20)
    xu[:ns] = 3.0
    xl[ns:2 * ns] = 0.0
    xl[2 * ns] = -4.0
    xu[2 * ns] = 4.0
    return MdsProblem(
        name=f"MdsEx1(ns={ns},nd={nd})", nxs=2 * ns, nxd=nd, neq=ns, nineq=3,
        Jcs_i=ci, Jcs_j=cj, Jcs_v=np.ones(2 * ns),
        Jds_i=di, Jds_j=dj, Jds_v=np.ones(di.size),
        Hss_i=np.arange(2 * ns, dtype=np.int32), Hss_j=np.arange(2 * ns, dtype=np.int32), Hss_v=np.ones(2 * ns),
        Jcd=np.full((ns, nd), -1.0), Jdd=np.ones((3, nd)), Hdd=_Qd(nd),
        xl=xl, xu=xu, dl=np.array([-2.0, -1e20, -2.0]), du=np.array([2.0, 2.0, 1e20]), x0=np.ones(n))


def mds_ex1_g(ns: int, nd: int, neq: int) -> MdsProblem:
    """Generalised MdsEx1: `ns` x-variables and `ns` s-variables (n_sparse = 2 ns), `nd` dense variables,
    `neq` <= ns equalities.  Equality j:  sum_{i = j (mod neq)} (x_i + s_i) + 0.5 x_{(j+1) mod ns} + (Md y)_j = 0,
    so consecutive rows overlap in one column and the Schur block Jcs Hxs^-1 Jcs^T is tridiagonal-like
    instead of diagonal.  Inequalities, bounds, objective as in MdsEx1."""
    assert 4 <= neq <= ns
    rows, cols, vals = [], [], []
    for j in range(neq):
        xs = np.arange(j, ns, neq)
        c = {int(i): 1.0 for i in xs}
        extra = (j + 1) % ns
        c[extra] = c.get(extra, 0.0) + 0.5
        for i in xs:
            c[ns + int(i)] = 1.0
        keys = sorted(c)
        rows.extend([j] * len(keys))
        cols.extend(keys)
        vals.extend(c[k] for k in keys)
    di, dj = _ineq_rows(ns)
    n = 2 * ns + nd
    xl = np.full(n, -1e20)
    xu = np.full(n, 1e20)
    xu[:ns] = 3.0
    xl[ns:2 * ns] = 0.0
    xl[2 * ns] = -4.0
    xu[2 * ns] = 4.0
    return MdsProblem(
        name=f"MdsEx1G(ns={ns},nd={nd},neq={neq})", nxs=2 * ns, nxd=nd, neq=neq, nineq=3,
        Jcs_i=np.array(rows, np.int32), Jcs_j=np.array(cols, np.int32), Jcs_v=np.array(vals, np.float64),
        Jds_i=di, Jds_j=dj, Jds_v=np.ones(di.size),
        Hss_i=np.arange(2 * ns, dtype=np.int32), Hss_j=np.arange(2 * ns, dtype=np.int32), Hss_v=np.ones(2 * ns),
        Jcd=np.full((neq, nd), -1.0), Jdd=np.ones((3, nd)), Hdd=_Qd(nd),
        xl=xl, xu=xu, dl=np.array([-2.0, -1e20, -2.0]), du=np.array([2.0, 2.0, 1e20]), x0=np.ones(n))


def barrier_diagonals(p: MdsProblem, seed: int = 20240916, mu: float = 0.1):
    """Synthetic log-barrier diagonals of an interior iterate (what hiopKKTLinSysCompressedMDSXYcYd::update
    computes at hiopKKTLinSysMDS.cpp:155-157 and :280-282):  Dx = zl/sxl + zu/sxu on bounded variables
    (0 on free ones), Dd = vl/sdl + vu/sdu > 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = p.nxs + p.nxd
    has_l = p.xl > -1e20
    has_u = p.xu < 1e20
    sl = rng.uniform(0.05, 2.0, n)
    su = rng.uniform(0.05, 2.0, n)
    Dx = np.where(has_l, mu / sl / sl, 0.0) + np.where(has_u, mu / su / su, 0.0)
    hl = p.dl > -1e20
    hu = p.du < 1e20
    tl = rng.uniform(0.05, 2.0, p.nineq)
    tu = rng.uniform(0.05, 2.0, p.nineq)
    Dd = np.where(hl, mu / tl / tl, 0.0) + np.where(hu, mu / tu / tu, 0.0)
    return Dx, Dd


def random_rhs(p: MdsProblem, seed: int = 7):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.uniform(-1, 1, p.nxs + p.nxd), rng.uniform(-1, 1, p.neq), rng.uniform(-1, 1, p.nineq))


def dense_ex2(n: int):
    """The reference's DenseConsEx2 (src/Drivers/Dense/NlpDenseConsEx2.hpp:18-30, .cpp:60-330):
        min sum 1/4 (x_i - 1)^4   s.t.  sum x_i = n+1;  5 <= 2 x_1 + sum_{i>=2} x_i;
        1 <= 2 x_1 + 0.5 x_2 + sum_{i>=3} x_i <= 2n;   4 x_1 + 2 x_2 + 2 x_3 + sum_{i>=4} x_i <= 4n;
        x_1 free, x_2 >= 0, 1.5 <= x_3 <= 10, x_i >= 0.5 (i >= 4);  x0 = 0.
    Returned in HiOp's split form: equality Jacobian Jc (1 x n) with rhs, inequality Jacobian Jd (3 x n) with dl/du."""
    assert n >= 4
    Jc = np.ones((1, n))
    Jd = np.ones((3, n))
    Jd[0, 0] = 2.0
    Jd[1, 0], Jd[1, 1] = 2.0, 0.5
    Jd[2, 0], Jd[2, 1], Jd[2, 2] = 4.0, 2.0, 2.0
    xl = np.full(n, 0.5)
    xu = np.full(n, 1e20)
    xl[0] = -1e20
    xl[1] = 0.0
    xl[2], xu[2] = 1.5, 10.0
    return dict(n=n, Jc=Jc, Jd=Jd, crhs=np.array([n + 1.0]), dl=np.array([5.0, 1.0, -1e20]),
                du=np.array([1e20, 2.0 * n, 4.0 * n]), xl=xl, xu=xu, x0=np.zeros(n),
                f=lambda x: 0.25 * float(np.sum((x - 1.0) ** 4)), grad=lambda x: (x - 1.0) ** 3,
                hess_diag=lambda x: 3.0 * (x - 1.0) ** 2)


def dense_ex1(n: int, r: float = 1.0):
    """The reference's DenseConsEx1 (src/Drivers/Dense/NlpDenseConsEx1.hpp:21-40,136-222, .cpp:17-54,90-104,246-259): the
    discretised QP  min <c, x> + 1/2 <x, x>  s.t.  integral(x) = 0.5,  0.1 <= x <= 1  on a (distorted) mesh of [0, 1] with
    element lengths m_k = m1 + k h, h = 2(1-r)/((1+r) n (n-1)), m1 = 2r/((1+r) n); inner products are mass-weighted
    (<a, b> = sum m_k a_k b_k), c(t) = -1 + 10 t for t <= 0.1 and 0 after, t_k = the middle of element k; x0 = 0.5.
    No inequality constraints (Jd is 0 x n).  `exact` = the optimum of the discrete problem (x = clip(lambda - c, 0.1, 1))."""
    k = np.arange(n, dtype=np.float64)
    m1 = 2.0 * r / ((1.0 + r) * n)
    h = 2.0 * (1.0 - r) / (1.0 + r) / (n - 1) / n
    mass = m1 + k * h
    t = 0.5 * ((2 * k + 1) * m1 + k * k * h)
    c = np.where(t <= 0.1, -1.0 + 10.0 * t, 0.0)

    def exact():
        lo, hi = -5.0, 5.0
        for _ in range(200):
            lam = 0.5 * (lo + hi)
            if float(np.sum(mass * np.clip(lam - c, 0.1, 1.0))) > 0.5:
                hi = lam
            else:
                lo = lam
        x = np.clip(0.5 * (lo + hi) - c, 0.1, 1.0)
        return float(np.sum(mass * (c * x + 0.5 * x * x))), x
    return dict(n=n, Jc=mass.reshape(1, n).copy(), Jd=np.zeros((0, n)), crhs=np.array([0.5]), dl=np.zeros(0), du=np.zeros(0),
                xl=np.full(n, 0.1), xu=np.full(n, 1.0), x0=np.full(n, 0.5), mass=mass, c=c,
                f=lambda x: float(np.sum(mass * (c * x + 0.5 * x * x))), grad=lambda x: mass * (x + c),
                hess_diag=lambda x: mass.copy(), exact=exact)


@dataclass
class SparseIneqProblem:
    """Data of an inequality-only sparse NLP at one iterate (the hiopNlpSparseIneq formulation the condensed sparse KKT needs,
    hiopKKTLinSysSparseCondensed.cpp:130-131)."""
    nx: int
    nineq: int
    Jd_i: np.ndarray
    Jd_j: np.ndarray
    Jd_v: np.ndarray
    H_i: np.ndarray
    H_j: np.ndarray
    H_v: np.ndarray


def sparse_ex2_ineq(n: int, x: np.ndarray | None = None, scal: float = 1.0) -> SparseIneqProblem:
    """The reference's SparseEx2 (src/Drivers/Sparse/NlpSparseEx2.hpp:30-52, convex objective) with its equality relaxed to a
    two-sided inequality — the form NlpSparseEx2Driver.cpp:289-296 runs with KKTLinsys = condensed:
        min  scal * sum 1/4 (x_i - 1)^4 + 1/2 x^T x
        s.t. 4 x_1 + 2 x_2 (two-sided),  2 x_1 + x_3 >= 5,  1 <= 2 x_1 + 0.5 x_i <= 2 n  (i = 4..n)
    Jacobian: n - 1 rows, each [2 or 4 at column 0, one more entry]; Hessian of the Lagrangian: diagonal 3 scal (x_i - 1)^2 + 1
    (the constraints are linear).  `x`: the iterate the Hessian is evaluated at (default: the driver's start point 0)."""
    assert n >= 3
    if x is None:
        x = np.zeros(n)
    m = n - 1
    Ji, Jj, Jv = [0, 0, 1, 1], [0, 1, 0, 2], [4.0, 2.0, 2.0, 1.0]
    for i in range(3, n):            # 0-based variable i, row i - 1
        Ji += [i - 1, i - 1]; Jj += [0, i]; Jv += [2.0, 0.5]
    Hi = np.arange(n, dtype=np.int32)
    Hv = 3.0 * scal * (x - 1.0) ** 2 + 1.0
    return SparseIneqProblem(n, m, np.array(Ji, np.int32), np.array(Jj, np.int32), np.array(Jv), Hi, Hi.copy(), Hv)


def sparse_chain_ineq(n: int, couple: int = 2, seed: int = 7) -> SparseIneqProblem:
    """A sparse inequality-only problem whose condensed matrix is BANDED (synthetic; the reference's sparse examples are all arrowheads):
    rows d_i = sum_{q < couple} a_iq x_{i+q}, i = 0 .. n - couple, diagonal Hessian — M = H + Dx + Jd^T Dd Jd has bandwidth couple - 1.
    The pattern the general sparse LDL^T (csrc/sparse_ldl.hip) is for."""
    r = np.random.Generator(np.random.PCG64(seed))
    m = n - couple + 1
    Ji = np.repeat(np.arange(m), couple).astype(np.int32)
    Jj = (Ji + np.tile(np.arange(couple), m)).astype(np.int32)
    Hi = np.arange(n, dtype=np.int32)
    return SparseIneqProblem(n, m, Ji, Jj, r.uniform(0.5, 1.5, Ji.size), Hi, Hi.copy(), r.uniform(0.5, 2.0, n))


def sparse_ex2_nlp(n: int, convex_obj: bool = False, rankdefic_eq: bool = True, rankdefic_ineq: bool = True, scal_neg_obj: float = 0.1):
    """The reference's SparseEx2 as a whole NLP (src/Drivers/Sparse/NlpSparseEx2.{hpp,cpp}: objective :126-143, constraints :156-186,
    bounds :52-110, start :319-326), with the driver's settings as defaults (NlpSparseEx2Driver.cpp:219-222):
        min sum (2 convex - 1) scal 1/4 (x_i - 1)^4 + 1/2 x_i^2
        s.t. 4 x_1 + 2 x_2 = 10;  2 x_1 + x_3 >= 5;  1 <= 2 x_1 + 0.5 x_i <= 2 n (i >= 4);  [4 x_1 + 2 x_3 <= 19];  [4 x_1 + 2 x_2 = 10]
        x_1 free, x_2 >= 0, 1 <= x_3 <= 10, x_i >= 0.5;  x0 = 0.
    The two bracketed rows (a dependent inequality, a DUPLICATED equality) make the Jacobians rank deficient; with convex_obj = False
    the Hessian diagonal -3 scal (x_i - 1)^2 + 1 goes negative away from 1: the inertia-correction loop has work to do.
    Returns the constraint rows as triplets (row-sorted) with [clow, cupp] in the user's order, plus callables."""
    sgn = 2 * int(convex_obj) - 1
    Ji, Jj, Jv, cl, cu = [0, 0, 1, 1], [0, 1, 0, 2], [4.0, 2.0, 2.0, 1.0], [10.0, 5.0], [10.0, 1e20]
    r = 2
    for i in range(3, n):
        Ji += [r, r]; Jj += [0, i]; Jv += [2.0, 0.5]; cl.append(1.0); cu.append(2.0 * n); r += 1
    if rankdefic_ineq:
        Ji += [r, r]; Jj += [0, 2]; Jv += [4.0, 2.0]; cl.append(-1e20); cu.append(19.0); r += 1
    if rankdefic_eq:
        Ji += [r, r]; Jj += [0, 1]; Jv += [4.0, 2.0]; cl.append(10.0); cu.append(10.0); r += 1
    xl = np.full(n, 0.5); xu = np.full(n, 1e20)
    xl[0] = -1e20; xl[1] = 0.0; xl[2] = 1.0; xu[2] = 10.0
    return dict(n=n, m=r, J_i=np.array(Ji, np.int32), J_j=np.array(Jj, np.int32), J_v=np.array(Jv), clow=np.array(cl), cupp=np.array(cu),
                xl=xl, xu=xu, x0=np.zeros(n),
                f=lambda x: float(np.sum(sgn * scal_neg_obj * 0.25 * (x - 1.0) ** 4 + 0.5 * x ** 2)),
                grad=lambda x: sgn * scal_neg_obj * (x - 1.0) ** 3 + x,
                hess_diag=lambda x: sgn * scal_neg_obj * 3.0 * (x - 1.0) ** 2 + 1.0)
