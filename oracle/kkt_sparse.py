"""CPU restatement of hiopKKTLinSysCondensedSparse — TEST INFRASTRUCTURE ONLY (see oracle/hiop_oracle.py's header).

reference: src/Optimization/hiopKKTLinSysSparseCondensed.cpp
  build_kkt_matrix        :105-335   M = Jd^T (Dd + delta_wd) Jd + H + Dx + delta_wx I   (CSR chain; here scipy.sparse)
  solve_compressed_direct :346-401
  solveCompressed         :403-449
The reference solves M by a sparse Cholesky: MA57 (COINHSL, no version pinned by cmake/FindHiopCOINHSL.cmake) on the CPU,
cuSOLVER sparse Cholesky on the GPU (:469-496).  Neither library nor its source is in the image: **parity is unpinned at the
solver boundary** (SURVEY.md section 8c).  What stands in here is the published algorithm the Cholesky implements on the
same matrix — LAPACK DPOTRF / DPOTRS on M as a dense matrix — i.e. the exact solution of the condensed system up to rounding,
with DPOTRF's "not positive definite" as the failure signal.  Anchors that ARE pinned: the XDYcYd identity (the solution of the
condensed path satisfies the uncondensed 4 x 4 block system the reference writes in its comment :364-368) and the reference's
own SpMV / CSR unit-test answers (tests/golden/reference_unit_tests.json).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.linalg import lapack

from . import hiop_oracle as ho


class KKTLinSysCondensedSparse:
    def __init__(self, nx, nineq, Jd_ij, H_ij):
        self.nx, self.nineq = nx, nineq
        self.Jd_ij, self.H_ij = Jd_ij, H_ij

    def set_values(self, Jd_val, H_val, Dx, Dd):
        self.Jd_val, self.H_val, self.Dx, self.Dd = Jd_val, H_val, Dx, Dd

    def set_diagonals(self, Dx, Dd):
        self.Dx, self.Dd = Dx, Dd

    def _J(self):
        return sp.csr_matrix((self.Jd_val, self.Jd_ij), shape=(self.nineq, self.nx))

    def _H(self):
        U = sp.coo_matrix((self.H_val, self.H_ij), shape=(self.nx, self.nx)).tocsr()
        return U + sp.triu(U, 1).T                                            # :259-300 (upper + its transpose, diagonal once)

    def build_kkt_matrix(self, delta_wx, delta_wd):                           # :105-335
        dv = lambda d, n: d if np.isscalar(d) else np.asarray(d, dtype=float)[:n]
        self.Hd = self.Dd + dv(delta_wd, self.nineq)                          # :150-151
        Dxp = self.Dx + dv(delta_wx, self.nx)                                 # :196
        J = self._J()
        JtDJ = (J.T @ sp.diags(self.Hd) @ J)                                  # :205-252
        self.M = (self._H() + sp.diags(Dxp) + JtDJ).tocsr()                   # :259-318
        return self.M

    def factorize(self):
        """Cholesky of M (what MA57 / cuSOLVER do): 0 negative eigenvalues on success, -1 when M is not positive definite."""
        c, info = lapack.dpotrf(self.M.toarray(), lower=0)
        self._chol = c if info == 0 else None
        return 0 if info == 0 else -1

    def solve_compressed(self, rx, rd, ryd):                                  # :346-401
        J = self._J()
        rhs = rx + J.T @ (self.Hd * ryd + rd)                                 # :370-379
        if self._chol is None:
            return False, None, None, None
        dx, info = lapack.dpotrs(self._chol, rhs, lower=0)                    # :384
        if info != 0:
            return False, None, None, None
        dd = J @ dx - ryd                                                     # :391-392
        dyd = self.Hd * dd - rd                                               # :394-396
        return True, dx, dd, dyd


def xdycyd_residual(k: KKTLinSysCondensedSparse, delta_wx, delta_wd, rx, rd, ryd, dx, dd, dyd):
    """Componentwise backward error of the UNcondensed system the condensed path stands for (the reference's comment :364-368,
    the XDYcYd order of hiopKKTLinSysCompressedSparseXDYcYd with no equalities and delta_cd = 0):
        [H + Dx + dwx   0          Jd^T ] [dx ]   [rx ]
        [0              Dd + dwd   -I   ] [dd ] = [rd ]
        [Jd             -I          0   ] [dyd]   [ryd]"""
    J, H = k._J(), k._H()
    dv = lambda d, n: d if np.isscalar(d) else np.asarray(d, dtype=float)[:n]
    D1 = k.Dx + dv(delta_wx, k.nx)
    D2 = k.Dd + dv(delta_wd, k.nineq)
    r1 = H @ dx + D1 * dx + J.T @ dyd - rx
    r2 = D2 * dd - dyd - rd
    r3 = J @ dx - dd - ryd
    a = np.abs
    s1 = a(H) @ a(dx) + a(D1) * a(dx) + a(J).T @ a(dyd) + a(rx)
    s2 = a(D2) * a(dd) + a(dyd) + a(rd)
    s3 = a(J) @ a(dx) + a(dd) + a(ryd)

    def be(r, s):
        s = np.where(s > 0, s, 1.0)
        return float(np.max(a(r) / s)) if r.size else 0.0
    return be(r1, s1), be(r2, s2), be(r3, s3)


class SparseCondensedProvider:
    """oracle/kkt_full.py provider: hiopKKTLinSysCondensedSparse as the compressed system of the full-space layer (an XDYcYd
    class, hiopKKTLinSysSparseCondensed.hpp:78)."""
    xd_form = True

    def __init__(self, k: KKTLinSysCondensedSparse):
        self.k = k
        self.nx, self.nd, self.nyc, self.nyd = k.nx, k.nineq, 0, k.nineq

    def set_diagonals(self, Dx, Dd):
        self.k.set_diagonals(Dx, Dd)

    def build(self, dwx, dwd, dcc, dcd):
        self.k.build_kkt_matrix(dwx, dwd)

    def factorize(self):
        n0 = self.k.factorize()
        return -1 if n0 < 0 else self.nyc + self.nyd      # Sylvester: M > 0  <=>  the full KKT has nyd negative eigenvalues

    def solve_xd(self, rx, rd, ryc, ryd):
        ok, dx, dd, dyd = self.k.solve_compressed(rx, rd, ryd)
        if not ok:
            z = np.zeros
            return False, z(self.nx), z(self.nd), z(0), z(self.nd)
        return True, dx, dd, np.zeros(0), dyd

    def hess_times_vec(self, x):
        return self.k._H() @ x

    def jac_times_vec(self, which, x):                            # which: "c" (no equalities here) / "d"
        return np.zeros(0) if which in ("c", 0) else self.k._J() @ x

    def jac_trans_times_vec(self, which, y):
        return np.zeros(self.nx) if which in ("c", 0) else self.k._J().T @ y
