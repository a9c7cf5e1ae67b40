"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's Krylov solvers.

pcg        hiopPCGSolver::solve, src/LinAlg/hiopKrylovSolver.cpp:152-373
bicgstab   hiopBiCGStabSolver::solve, :397-700 (restated in oracle/kkt_full.py, re-exported here)
krylov_test_matrix   the matrix of the reference's own tests/test_pcg.cpp and tests/test_bicgstab.cpp (:14-53)

Pin: the reference's tests only print the convergence message, they hold no numbers.  The restatement is checked in
tests/test_oracle_krylov.py against a textbook preconditioned CG written independently (identical iterates), against a
direct solve, and on every exit path of the state machine (flags 0/1/3/4, minimal-residual fallback)."""
import numpy as np

from oracle.kkt_full import bicgstab  # noqa: F401


def krylov_test_matrix(n):
    """Upper triangle (COO) of the symmetric matrix of tests/test_pcg.cpp:26-51: diag 5(i+1), first off-diagonal 2(i+1),
    second off-diagonal (i+1); and the diagonal of the Jacobi preconditioner 1/(5(i+1)) (is_diag_pred)."""
    ii, jj, vv = [], [], []
    for i in range(n):
        ii.append(i); jj.append(i); vv.append((i + 1.0) * 5.0)
        if i + 1 < n:
            ii.append(i); jj.append(i + 1); vv.append((i + 1.0) * 2.0)
        if i + 2 < n:
            ii.append(i); jj.append(i + 2); vv.append((i + 1.0) * 1.0)
    minv = 1.0 / ((np.arange(n) + 1.0) * 5.0)
    return np.array(ii, dtype=np.int32), np.array(jj, dtype=np.int32), np.array(vv), minv


def sym_times_vec(n, ii, jj, vv, x):
    """hiopMatrixSymSparseTriplet::timesVec (hiopMatrixSparseTriplet.cpp:924-958): upper-triangle triplets."""
    y = np.zeros(n)
    np.add.at(y, ii, vv * x[jj])
    off = ii != jj
    np.add.at(y, jj[off], vv[off] * x[ii[off]])
    return y


def pcg(A, ML, MR, b, tol=1e-9, maxit=8, x0=None):
    """Returns (x, converged, flag, iter, abs_resid, rel_resid, xk) — xk is what the solver's persistent start vector
    holds afterwards (xk_ aliases x0_, :183)."""
    nrm = np.linalg.norm
    n2b = nrm(b)
    if n2b == 0.0:                                                  # :154-159
        return b.copy(), True, 0, 0.0, -1.0, -1.0, (np.zeros_like(b) if x0 is None else x0)
    xk = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
    flag, imin = 1, 0
    tolb = tol * n2b
    xmin = xk.copy()
    res = -(A(xk) - b)                                              # :190-193
    normr = nrm(res)
    abs_resid = normr
    if normr <= tolb:                                               # :195-201
        return xk.copy(), True, 0, 0.0, normr, normr / n2b, xk
    normrmin, rho = normr, 1.0
    stagsteps = moresteps = 0
    eps = np.finfo(np.float64).eps
    maxmsteps, maxstagsteps = 100, 3
    it = -1.0
    pk = None
    ii = 0
    while ii < maxit:
        yk = ML(res) if ML is not None else res.copy()
        zk = MR(yk) if MR is not None else yk.copy()
        rho1 = rho
        rho = float(res @ zk)
        if rho == 0 or abs(rho) > 1e20:                             # :233-237
            flag, it = 4, ii + 1
            break
        if ii == 0:
            pk = zk.copy()
        else:
            beta = rho / rho1
            if beta == 0 or abs(beta) > 1e20:
                flag, it = 4, ii + 1
                break
            pk = pk * beta + zk
        qk = A(pk)
        pq = float(pk @ qk)
        if pq <= 0.0 or abs(pq) > 1e20:                             # :256-262
            flag, it = 4, ii + 1
            break
        alpha = rho / pq
        if abs(alpha) > 1e20:
            flag, it = 4, ii + 1
            break
        if nrm(pk) * abs(alpha) < eps * nrm(xk):                    # :270-274
            stagsteps += 1
        else:
            stagsteps = 0
        xk = xk + alpha * pk
        res = res - alpha * qk
        normr = nrm(res)
        abs_resid = normr
        if normr <= tolb or stagsteps >= maxstagsteps or moresteps:      # :284-309
            res = -(A(xk) - b)
            abs_resid = nrm(res)
            if abs_resid <= tolb:
                flag, it = 0, ii + 1
                break
            if stagsteps >= maxstagsteps and moresteps == 0:
                stagsteps = 0
            moresteps += 1
            if moresteps >= maxmsteps:
                flag, it = 3, ii + 1
                break
        if abs_resid < normrmin:                                    # :311-315
            normrmin = abs_resid
            xmin = xk.copy()
            imin = ii
        if stagsteps >= maxstagsteps:
            flag, it = 3, ii + 1
            break
        ii += 1
    if flag == 0:                                                   # :324-328
        return xk.copy(), True, 0, float(it), abs_resid, abs_resid / n2b, xk
    res = -(A(xmin) - b)                                            # :330-352
    normr_comp = nrm(res)
    if normr_comp <= abs_resid:
        return xmin.copy(), False, flag, float(imin + 1), normr_comp, normr_comp / n2b, xk
    return xk.copy(), False, flag, float(ii + 1), abs_resid, abs_resid / n2b, xk
