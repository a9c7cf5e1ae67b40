"""CPU oracle for the HiOp KKT hot path — TEST INFRASTRUCTURE ONLY.

This module restates, in numpy (fp64) + LAPACK, the algorithms of the reference's CPU path for the
hot path this repository accelerates.  It is the *checker*: only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import it.  Nothing under hiop_amd/ imports it, and the product
path has no CPU fallback.

Every function cites the reference file:line (LLNL/hiop v1.1.0) whose loop it follows.

Third-party dependency of the reference path: LAPACK/BLAS (Fortran ABI; the reference links
whatever `find_package(LAPACK)` finds, CMakeLists.txt:99-106).  Here the same published routines
(DSYTRF / DSYTRS / DPOSVX / DPOTRF / DPOTRS, Bunch-Kaufman as in LAPACK 3.9+) come from the
OpenBLAS build bundled with scipy 1.15.3 (`scipy.linalg.lapack`).

Pinning: the reference cannot be built in this image under the project
rules (it needs two CMake-generated headers, one of which requires a Fortran compiler), so the
oracle is pinned against the known answers the reference's own tests hold:
  * closed-form expected values of tests/LinAlg/{vectorTests,matrixTestsDense,matrixTestsSparse,
    matrixTestsSymSparse}.hpp  (96 cases in tests/golden/reference_unit_tests.json, generator beside it;
    tests/test_reference_known_answers.py runs them through this module AND through the HIP library),
  * the `-selfcheck` objective values of the Dense/MDS drivers
    (src/Drivers/MDS/NlpMdsEx1Driver.cpp:149, src/Drivers/Dense/NlpDenseConsEx1Driver.cpp:139-140,
    NlpDenseConsEx2Driver.cpp:124-125) reproduced by oracle/ipm.py driving THIS module's KKT path,
  * `write_kkt` triples (matrix, rhs, solution) of the reference's MdsEx1 run committed under
    tests/golden/ (provenance in tests/golden/README.md).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import lapack

# =====================================================================================
# hiopVector (reference: src/LinAlg/hiopVectorPar.cpp) — element-wise family
# Functions mutate `y` in place like the reference methods mutate `this`.
# =====================================================================================


def set_to_constant_w_pattern(y, c, select):          # :139
    y[:] = np.where(select == 1.0, c, 0.0)


def copy_from_w_pattern(y, x, select):                 # :179
    m = select == 1.0
    y[m] = x[m]


def copy_to_starting_at_w_pattern(x, dest, start, select):   # :322
    sel = x[select == 1.0]
    dest[start:start + sel.size] = sel
    return sel.size


def starting_at_copy_to_starting_at_w_pattern(src, start_src, dest, start_dest, select_dest, num_elems=-1):  # :431
    n_src_avail = src.size - start_src
    if num_elems < 0:
        num_elems = min(n_src_avail, dest.size - start_dest)
    else:
        num_elems = min(num_elems, n_src_avail, dest.size - start_dest)
    idx = np.nonzero(select_dest[start_dest:] == 1.0)[0][:num_elems] + start_dest
    dest[idx] = src[start_src:start_src + idx.size]


def copy_from_indexes(y, src, idx):                    # :194-206
    y[:] = src[idx]


def copy_from_starting(y, start_in_y, v):              # :222-237 (both overloads: a buffer of nv entries, a whole vector)
    assert start_in_y + v.size <= y.size
    y[start_in_y:start_in_y + v.size] = v


def starting_at_copy_from_starting_at(dest, start_dest, src, start_src):   # :241-251: everything left of `dest` is overwritten
    n = dest.size - start_dest
    assert n >= 0
    dest[start_dest:] = src[start_src:start_src + n]


def copy_to_starting(x, dest, start_in_dest):          # :295-316 (non-distributed case)
    assert start_in_dest + x.size <= dest.size
    if x.size > 0:
        dest[start_in_dest:start_in_dest + x.size] = x


def copy_from_two_vec_w_pattern(y, c, c_map, d, d_map):   # :338-359
    assert c.size + d.size == y.size
    y[c_map] = c
    y[d_map] = d


def copy_to_two_vec_w_pattern(y, c, c_map, d, d_map):     # :366-387
    assert c.size + d.size == y.size
    c[:] = y[c_map]
    d[:] = y[d_map]


def starting_at_copy_to_starting_at(src, start_src, dest, start_dest, num_elems=-1):   # :395-428
    assert 0 <= start_src <= src.size and 0 <= start_dest <= dest.size
    if num_elems < 0:
        num_elems = min(src.size - start_src, dest.size - start_dest)
    else:
        num_elems = min(num_elems, src.size - start_src, dest.size - start_dest)
    dest[start_dest:start_dest + num_elems] = src[start_src:start_src + num_elems]


def isnan_local(x):                                    # :1167-1171  (exists e: isnan(e))
    return int(np.any(np.isnan(x)))


def isinf_local(x):                                    # :1173-1177
    return int(np.any(np.isinf(x)))


def isfinite_local(x):                                 # :1179-1183  (for all e: isfinite(e))
    return int(np.all(np.isfinite(x)))


def component_div_w_pattern(y, x, select):             # :580
    with np.errstate(divide="ignore", invalid="ignore"):
        y[:] = np.where(select == 0.0, 0.0, y / x)


def component_sgn(y):                                  # :639
    y[:] = (0.0 < y).astype(np.float64) - (y < 0.0).astype(np.float64)


def axzpy(y, alpha, x, z):                             # :710
    if alpha == 1.0:
        y += x * z
    elif alpha == -1.0:
        y -= x * z
    elif alpha != 0.0:
        y += alpha * x * z


def axdzpy(y, alpha, x, z):                            # :736
    if alpha == 0.0:
        return
    if alpha == 1.0:
        y += x / z
    elif alpha == -1.0:
        y -= x / z
    else:
        y += x / z * alpha


def axdzpy_w_pattern(y, alpha, x, z, select):          # :767
    m = select == 1.0
    if alpha == 1.0:
        y[m] += x[m] / z[m]
    elif alpha == -1.0:
        y[m] -= x[m] / z[m]
    else:
        y[m] += alpha * x[m] / z[m]


def add_log_barrier_grad(y, alpha, x, select):         # :893
    m = select == 1.0
    y[m] += alpha / x[m]


def add_linear_damping_term(y, ixl, ixr, alpha, ct):   # :927
    y[:] = alpha * y + (ixl - ixr) * ct


def adjust_duals_plh(z, x, select, mu, kappa):         # :1117
    m = select == 1.0
    a = mu / x[m]
    b = a / kappa
    a = a * kappa
    zi = z[m].copy()
    out = zi.copy()
    c1 = zi < b
    out[c1] = b[c1]
    c2 = (~c1) & (a <= b)
    out[c2] = b[c2]
    c3 = (~c1) & (~c2) & (a < zi)
    out[c3] = a[c3]
    z[m] = out


def project_into_bounds(x0, xl, ixl, xu, ixu, kappa1, kappa2):   # :964
    small = np.finfo(np.float64).tiny * 100
    both = (ixl != 0) & (ixu != 0)
    if np.any(both & (xl > xu)):
        return False
    aux = kappa2 * (xu - xl) - small
    lo = xl + np.minimum(kappa1 * np.maximum(1.0, np.abs(xl)), aux)
    hi = xu - np.minimum(kappa1 * np.maximum(1.0, np.abs(xu)), aux)
    v = x0.copy()
    c_lo = both & (v < lo)
    c_hi = both & ~c_lo & (v > hi)
    v[c_lo] = lo[c_lo]
    v[c_hi] = hi[c_hi]
    only_l = (~both) & (ixl != 0)
    v[only_l] = np.maximum(v[only_l], xl[only_l] + kappa1 * np.maximum(1.0, np.abs(xl[only_l])) - small)
    only_u = (~both) & (ixl == 0) & (ixu != 0)
    v[only_u] = np.minimum(v[only_u], xu[only_u] - kappa1 * np.maximum(1.0, np.abs(xu[only_u])) - small)
    x0[:] = v
    return True


# ---- reductions ----
def infnorm(x):                                        # :501
    return float(np.max(np.abs(x))) if x.size else 0.0


def onenorm(x):                                        # :540
    return float(np.sum(np.abs(x)))


def vmin_w_pattern(x, select):                         # :821
    m = select == 1.0
    return float(np.min(x[m])) if np.any(m) else float(np.finfo(np.float64).max)


def log_barrier(x, select):                            # :863 (Kahan-compensated in the reference)
    m = select == 1.0
    return float(np.sum(np.log(x[m].astype(np.longdouble))))


def linear_damping_term(x, ixl, ixr, mu, kappa_d):     # :907
    m = (ixl == 1.0) & (ixr == 0.0)
    return float(np.sum(x[m])) * mu * kappa_d


def fraction_to_the_bdry(x, d, tau):                   # :1017
    m = d < 0
    if not np.any(m):
        return 1.0
    return float(min(1.0, np.min(-tau * x[m] / d[m])))


def fraction_to_the_bdry_w_pattern(x, d, tau, select):  # :1038
    m = (d < 0) & (select != 0)
    if not np.any(m):
        return 1.0
    return float(min(1.0, np.min(-tau * x[m] / d[m])))


def all_positive_w_pattern(x, select):                 # :1095
    return int(not np.any((select != 0.0) & (x <= 0.0)))


def matches_pattern(x, select):                        # :1073
    return int(not np.any((select == 0.0) & (x != 0.0)))


# =====================================================================================
# hiopMatrixDenseRowMajor (reference: src/LinAlg/hiopMatrixDenseRowMajor.cpp)
# =====================================================================================
def times_vec(A, beta, y, alpha, x):                   # :458
    y[:] = (0.0 if beta == 0.0 else beta * y) + alpha * (A @ x)


def trans_times_vec(A, beta, y, alpha, x):             # :510
    y[:] = (0.0 if beta == 0.0 else beta * y) + alpha * (A.T @ x)


def add_sub_diagonal(A, start, alpha, d, src_start=0, num=-1):      # :719,735
    if num < 0:
        num = d.size - src_start
    num = min(num, A.shape[0] - start)
    idx = np.arange(num) + start
    A[idx, idx] += alpha * d[src_start:src_start + num]


def trans_add_to_sym_upper(A, row_start, col_start, alpha, W):      # :779
    m, n = A.shape
    W[row_start:row_start + n, col_start:col_start + m] += alpha * A.T


def add_upper_to_sym_upper(A, diag_start, alpha, W):                # :810
    n = A.shape[0]
    W[diag_start:diag_start + n, diag_start:diag_start + n] += alpha * np.triu(A)


def times_mat(A, beta, W, alpha, X):                   # :578 (timesMat_local)
    W[:] = beta * W + alpha * (A @ X)


def trans_times_mat(A, beta, W, alpha, X):             # :616
    W[:] = beta * W + alpha * (A.T @ X)


def times_mat_trans(A, beta, W, alpha, X):             # :646 (local part; the caller all-reduces)
    W[:] = beta * W + alpha * (A @ X.T)


def add_diagonal(A, alpha, d=None):                    # :703 (vector), :715 (constant)
    i = np.arange(A.shape[0])
    A[i, i] += alpha * d if d is not None else alpha


def max_abs_value(A):                                  # :832
    return float(np.max(np.abs(A))) if A.size else 0.0


def row_max_abs_value(A):                              # :844
    return np.max(np.abs(A), axis=1)


def scale_row(A, x, inv):                              # :865
    A *= ((1.0 / x) if inv else x)[:, None]


def copy_rows_from(dst, src, num_rows, row_dest):      # :169
    dst[row_dest:row_dest + num_rows, :] = src[:num_rows, :]


def copy_rows_from_select(dst, src, rows):             # :182
    dst[:len(rows), :] = src[np.asarray(rows), :]


def copy_block_from_matrix(dst, i0, j0, src):          # :200
    dst[i0:i0 + src.shape[0], j0:j0 + src.shape[1]] = src


def copy_from_matrix_block(dst, src, i0, j0):          # :222
    dst[:, :] = src[i0:i0 + dst.shape[0], j0:j0 + dst.shape[1]]


def symmetrize(A):                                     # :912 (upper -> lower)
    iu = np.triu_indices(A.shape[0], 1)
    A[iu[1], iu[0]] = A[iu]


def shift_rows(A, shift):                              # :238
    m = A.shape[0]
    if shift == 0 or abs(shift) == m or m <= 1:
        return
    if shift < 0:
        A[:m + shift, :] = A[-shift:, :].copy()
    else:
        A[shift:, :] = A[:m - shift, :].copy()


# =====================================================================================
# hiopMatrixSparseTriplet (reference: src/LinAlg/hiopMatrixSparseTriplet.cpp)
# COO, row-sorted, int32 indices
# =====================================================================================
def sp_times_vec(nrows, iRow, jCol, val, beta, y, alpha, x):        # :73
    y *= beta
    np.add.at(y, iRow, alpha * x[jCol] * val)


def sp_trans_times_vec(ncols, iRow, jCol, val, beta, y, alpha, x):  # :110
    y *= beta
    np.add.at(y, jCol, alpha * x[iRow] * val)


def _to_dense(nrows, ncols, iRow, jCol, val):
    M = np.zeros((nrows, ncols))
    np.add.at(M, (iRow, jCol), val)
    return M


def sp_add_MDinvMtrans_diag_block(m, ncols, iRow, jCol, val, start, alpha, D, W):   # :390
    """W[start+i, start+j] += alpha * sum_c M[i,c] M[j,c] / D[c], j >= i (upper triangle only)."""
    import scipy.sparse as sp
    M = sp.csr_matrix((val, (iRow, jCol)), shape=(m, ncols))
    S = (M @ sp.diags(1.0 / D) @ M.T).toarray()
    W[start:start + m, start:start + m] += alpha * np.triu(S)


def sp_add_MDinvNtrans(m1, m2, ncols, i1, j1, v1, i2, j2, v2, row_start, col_start, alpha, D, W):   # :447
    import scipy.sparse as sp
    M1 = sp.csr_matrix((v1, (i1, j1)), shape=(m1, ncols))
    M2 = sp.csr_matrix((v2, (i2, j2)), shape=(m2, ncols))
    S = (M1 @ sp.diags(1.0 / D) @ M2.T).toarray()
    W[row_start:row_start + m1, col_start:col_start + m2] += alpha * S


def sp_add_MDinvMtrans_rowmerge(m, iRow, jCol, val, start, alpha, D, W):
    """Literal restatement of the reference's sorted-merge double loop (:406-437); small cases only."""
    rs = np.searchsorted(iRow, np.arange(m + 1))
    for i in range(m):
        acc = 0.0
        for k in range(rs[i], rs[i + 1]):
            acc += val[k] / D[jCol[k]] * val[k]
        W[i + start, i + start] += alpha * acc
        for j in range(i + 1, m):
            acc = 0.0
            ki, kj = rs[i], rs[j]
            while ki < rs[i + 1] and kj < rs[j + 1]:
                if jCol[ki] == jCol[kj]:
                    acc += val[ki] / D[jCol[ki]] * val[kj]
                    ki += 1
                    kj += 1
                elif jCol[ki] < jCol[kj]:
                    ki += 1
                else:
                    kj += 1
            W[i + start, j + start] += alpha * acc


def spsym_times_vec(n, iRow, jCol, val, beta, y, alpha, x):           # hiopMatrixSymSparseTriplet::timesVec :924-958
    y *= beta
    np.add.at(y, iRow, alpha * val * x[jCol])
    off = iRow != jCol
    np.add.at(y, jCol[off], alpha * val[off] * x[iRow[off]])


def spsym_add_upper_to_sym_upper(iRow, jCol, val, diag_start, alpha, W):   # :980
    np.add.at(W, (iRow + diag_start, jCol + diag_start), alpha * val)


def spsym_add_diag_to_vec(iRow, jCol, val, alpha, y, vec_start, diag_src_start=0, num_elems=-1):    # :1018
    if num_elems < 0:
        num_elems = y.size
    m = (iRow == jCol) & (iRow >= diag_src_start) & (iRow < diag_src_start + num_elems)
    np.add.at(y, vec_start + iRow[m], alpha * val[m])


def sp_trans_add_to_sym_upper(iRow, jCol, val, row_start, col_start, alpha, W):                    # :255
    """W[row_start + j, col_start + i] += alpha * M[i, j] (the destination must be inside the upper triangle)."""
    assert np.all(jCol + row_start <= iRow + col_start)
    np.add.at(W, (jCol + row_start, iRow + col_start), alpha * val)


def sp_row_max_abs(nrows, iRow, val):                                                               # :285
    ret = np.zeros(nrows)
    np.maximum.at(ret, iRow, np.abs(val))
    return ret


def sp_scale_rows(iRow, val, scal, inv):                                                            # :303
    val *= (1.0 / scal[iRow]) if inv else scal[iRow]


def sp_copy_to_dense(nrows, ncols, iRow, jCol, val):                                                # :363
    return _to_dense(nrows, ncols, iRow, jCol, val)


def sp_indexes_ordered(iRow, jCol):                                                                 # :377
    for k in range(1, iRow.size):
        if iRow[k] < iRow[k - 1] or (iRow[k] == iRow[k - 1] and jCol[k] < jCol[k - 1]):
            return False
    return True


def sp_times_mat_trans(m1, m2, ncols, i1, j1, v1, i2, j2, v2, beta, W, alpha):                      # :144
    """W = beta*W + alpha * M1 * M2^T (both sparse, W dense m1 x m2)."""
    import scipy.sparse as sp
    M1 = sp.csr_matrix((v1, (i1, j1)), shape=(m1, ncols))
    M2 = sp.csr_matrix((v2, (i2, j2)), shape=(m2, ncols))
    W *= beta
    W += alpha * (M1 @ M2.T).toarray()


# =====================================================================================
# hiopLinSolverSymDenseLapack (reference: src/LinAlg/hiopLinSolverSymDenseLapack.hpp:75-195)
# =====================================================================================
class LinSolverSymDenseLapack:
    """M holds the UPPER triangle, row-major (= LAPACK column-major lower, uplo='L')."""

    def __init__(self, n):
        self.n = n
        self.M = np.zeros((n, n))
        self.ldu = None
        self.ipiv = None

    def matrix_changed(self):
        n = self.n
        if n == 0:
            return 0
        # row-major upper == Fortran lower of the transposed view
        a = np.asfortranarray(self.M.T)
        ldu, ipiv, info = lapack.dsytrf(a, lower=1)
        if info != 0:
            return -1
        self.ldu, self.ipiv = ldu, ipiv
        # inertia, LINPACK dsidi style (:127-167); scipy returns 1-based LAPACK ipiv (negative = 2x2 block)
        neg = null = 0
        t = 0.0
        MM = ldu.T  # MM[k, k+1] is the sub-diagonal entry of the 2x2 block in the C view
        for k in range(n):
            d = MM[k, k]
            if ipiv[k] <= 0:
                if t == 0.0:
                    if k + 1 < n:
                        t = abs(MM[k, k + 1])
                        d = (d / t) * MM[k + 1, k + 1] - t
                else:
                    d = t
                    t = 0.0
            if d < -1e-14:
                neg += 1
            elif d < 1e-14:
                null += 1
        if null > 0:
            return -1
        return neg

    def solve(self, rhs):
        if self.n == 0:
            return True
        x, info = lapack.dsytrs(self.ldu, self.ipiv, rhs, lower=1)
        rhs[:] = x
        return info == 0


def ldlt_nopiv(M_upper):
    """Reference semantics of the no-pivot GPU solver (MAGMA dsytrf_nopiv, hiopLinSolverSymDenseMagma.cpp:349):
    A = U^T D U; returns (U unit upper, d).  Plain loops; small cases only."""
    n = M_upper.shape[0]
    A = np.triu(M_upper).copy()
    d = np.zeros(n)
    for k in range(n):
        d[k] = A[k, k]
        if k + 1 < n:
            v = A[k, k + 1:].copy()
            u = v / d[k]
            A[k + 1:, k + 1:] -= np.triu(np.outer(v, u))
            A[k, k + 1:] = u
    U = np.triu(A, 1) + np.eye(n)
    return U, d


# =====================================================================================
# hiopKKTLinSysCompressedMDSXYcYd (reference: src/Optimization/hiopKKTLinSysMDS.cpp)
# =====================================================================================
class KKTLinSysCompressedMDSXYcYd:
    def __init__(self, nxs, nxd, neq, nineq, Jcs, Jds, Hss):
        """Jcs/Jds/Hss: (iRow, jCol) int32 index pairs (row-sorted COO)."""
        self.nxs, self.nxd, self.neq, self.nineq = nxs, nxd, neq, nineq
        self.Jcs_ij, self.Jds_ij, self.Hss_ij = Jcs, Jds, Hss
        self.linsys = LinSolverSymDenseLapack(nxd + neq + nineq)

    def set_values(self, Jcs_val, Jds_val, Hss_val, Jcd, Jdd, Hdd, Dx, Dd):
        self.Jcs_val, self.Jds_val, self.Hss_val = Jcs_val, Jds_val, Hss_val
        self.Jcd, self.Jdd, self.Hdd, self.Dx, self.Dd = Jcd, Jdd, Hdd, Dx, Dd

    def build_kkt_matrix(self, delta_wx, delta_wd, delta_cc, delta_cd):       # :172-305
        nxs, nxd, neq, nineq = self.nxs, self.nxd, self.neq, self.nineq
        M = self.linsys.M
        M[:] = 0.0                                                           # :196
        add_upper_to_sym_upper(self.Hdd, 0, 1.0, M)                          # :204
        trans_add_to_sym_upper(self.Jcd, 0, nxd, 1.0, M)                     # :205
        trans_add_to_sym_upper(self.Jdd, 0, nxd + neq, 1.0, M)               # :206
        add_sub_diagonal(M, 0, 1.0, self.Dx, nxs, nxd)                       # :213
        # the perturbations are vectors in the reference (:178-181; delta_wx over [sparse; dense] x, delta_wd / delta_cd over the
        # inequalities, delta_cc over the equalities); a scalar stands for the constant vector
        dv = lambda d, lo, hi: d if np.isscalar(d) else np.asarray(d, dtype=float)[lo:hi]
        idx = np.arange(nxd)
        M[idx, idx] += dv(delta_wx, nxs, nxs + nxd)                          # :215
        Hxs = self.Dx[:nxs] + dv(delta_wx, 0, nxs)                           # :223-227
        spsym_add_diag_to_vec(self.Hss_ij[0], self.Hss_ij[1], self.Hss_val, 1.0, Hxs, 0)   # :231
        self.Hxs = Hxs
        ci, cj = self.Jcs_ij
        di, dj = self.Jds_ij
        sp_add_MDinvMtrans_diag_block(neq, nxs, ci, cj, self.Jcs_val, nxd, -1.0, Hxs, M)   # :239
        idx = np.arange(neq) + nxd
        M[idx, idx] += -dv(delta_cc, 0, neq)                                 # :245
        sp_add_MDinvMtrans_diag_block(nineq, nxs, di, dj, self.Jds_val, nxd + neq, -1.0, Hxs, M)   # :267
        sp_add_MDinvNtrans(neq, nineq, nxs, ci, cj, self.Jcs_val, di, dj, self.Jds_val, nxd, nxd + neq, -1.0, Hxs, M)  # :275
        self.Dd_inv = 1.0 / (dv(delta_wd, 0, nineq) + self.Dd)               # :280-286
        idx = np.arange(nineq) + nxd + neq
        M[idx, idx] += -self.Dd_inv                                          # :289
        M[idx, idx] += -dv(delta_cd, 0, nineq)                               # :290
        return M

    def factorize_with_curv_check(self):                                     # :78-110
        n_neg = self.linsys.matrix_changed()
        if n_neg >= 0:
            n_neg_xs = int(np.sum(self.Hxs < -1e-14))
            n_zero_xs = int(np.sum(np.abs(self.Hxs) < 1e-14))
            if n_zero_xs > 0:
                return -1
            n_neg += n_neg_xs
        return n_neg

    def solve_compressed(self, rx, ryc, ryd):                                # :307-403
        nxs, nxd, neq, nineq = self.nxs, self.nxd, self.neq, self.nineq
        ci, cj = self.Jcs_ij
        di, dj = self.Jds_ij
        rxs = rx[:nxs] / self.Hxs                                            # :337-338
        dyc = ryc.copy()
        sp_times_vec(neq, ci, cj, self.Jcs_val, 1.0, dyc, -1.0, rxs)         # :344
        ryd = ryd.copy()
        sp_times_vec(nineq, di, dj, self.Jds_val, 1.0, ryd, -1.0, rxs)       # :347
        rhs = np.concatenate([rx[nxs:], dyc, ryd])                           # :353-357
        self.last_rhs = rhs.copy()
        ok = self.linsys.solve(rhs)                                          # :367
        dx = np.empty(nxs + nxd)
        dx[nxs:] = rhs[:nxd]                                                 # :383
        dyc = rhs[nxd:nxd + neq].copy()
        dyd = rhs[nxd + neq:].copy()
        dxs = rx[:nxs].copy()                                                # :390
        sp_trans_times_vec(nxs, ci, cj, self.Jcs_val, 1.0, dxs, -1.0, dyc)   # :391
        sp_trans_times_vec(nxs, di, dj, self.Jds_val, 1.0, dxs, -1.0, dyd)   # :392
        dx[:nxs] = dxs / self.Hxs                                            # :393
        return ok, dx, dyc, dyd


def kkt_mds_full_residual(k: KKTLinSysCompressedMDSXYcYd, deltas, rx, ryc, ryd, dx, dyc, dyd):
    """Backward error of a solution of the UNcondensed XYcYd system
        [H+Dx+dwx  Jc^T  Jd^T            ] [dx ]   [rx ]
        [Jc       -dcc   0               ] [dyc] = [ryc]
        [Jd        0  -(Dd+dwd)^-1 - dcd ] [dyd]   [ryd]
    (the system whose residual the reference checks under HIOP_DEEPCHECKS, hiopKKTLinSys.cpp:695-740).
    Returns, per block row, the componentwise (Oettli-Prager) backward error
        max_i |K x - b|_i / (|K| |x| + |b|)_i
    which is scale-invariant: ~n*eps for a backward-stable solve whatever the conditioning."""
    import scipy.sparse as sp
    dwx, dwd, dcc, dcd = deltas
    nxs, nxd, neq, nineq = k.nxs, k.nxd, k.neq, k.nineq
    Jcs = sp.csr_matrix((k.Jcs_val, k.Jcs_ij), shape=(neq, nxs))
    Jds = sp.csr_matrix((k.Jds_val, k.Jds_ij), shape=(nineq, nxs))
    Hs_diag = np.zeros(nxs)
    spsym_add_diag_to_vec(k.Hss_ij[0], k.Hss_ij[1], k.Hss_val, 1.0, Hs_diag, 0)
    Hd = np.triu(k.Hdd) + np.triu(k.Hdd, 1).T
    dxs, dxd = dx[:nxs], dx[nxs:]
    if not np.isscalar(dwx):   # vector-valued perturbations (the reference's form): [sparse; dense] split
        dwx = np.asarray(dwx, dtype=float)
        dwx_s, dwx_d = dwx[:nxs], dwx[nxs:]
    else:
        dwx_s = dwx_d = dwx
    D1s = Hs_diag + k.Dx[:nxs] + dwx_s
    D1d = k.Dx[nxs:] + dwx_d
    D3 = 1.0 / (k.Dd + dwd) + dcd
    r1s = D1s * dxs + Jcs.T @ dyc + Jds.T @ dyd - rx[:nxs]
    r1d = Hd @ dxd + D1d * dxd + k.Jcd.T @ dyc + k.Jdd.T @ dyd - rx[nxs:]
    r2 = Jcs @ dxs + k.Jcd @ dxd - dcc * dyc - ryc
    r3 = Jds @ dxs + k.Jdd @ dxd - D3 * dyd - ryd
    a = np.abs
    s1s = a(D1s) * a(dxs) + a(Jcs).T @ a(dyc) + a(Jds).T @ a(dyd) + a(rx[:nxs])
    s1d = a(Hd) @ a(dxd) + a(D1d) * a(dxd) + a(k.Jcd).T @ a(dyc) + a(k.Jdd).T @ a(dyd) + a(rx[nxs:])
    s2 = a(Jcs) @ a(dxs) + a(k.Jcd) @ a(dxd) + np.abs(dcc) * a(dyc) + a(ryc)
    s3 = a(Jds) @ a(dxs) + a(k.Jdd) @ a(dxd) + a(D3) * a(dyd) + a(ryd)

    def be(r, sc):
        sc = np.where(sc > 0, sc, 1.0)
        return float(np.max(a(r) / sc)) if r.size else 0.0

    return max(be(r1s, s1s), be(r1d, s1d)), be(r2, s2), be(r3, s3)


# =====================================================================================
# hiopHessianLowRank (reference: src/Optimization/hiopHessianLowRank.cpp) — compact L-BFGS,
#   B = B0 + Dx - [B0 S, Y] (...)^-1 [...]^T ,  B0 = sigma I ;  inverse through the 2l x 2l matrix V.
# `allreduce(buf)` sums a numpy buffer over the column partition in place (identity for one rank) —
# the MPI_Allreduce call sites :459, :590-591.
# =====================================================================================
def _no_reduce(buf):
    return buf


def symm_mat_times_diag_times_mat_trans_local(beta, W, alpha, X, d):      # :1079  (writes BOTH triangles)
    G = (X * d) @ X.T
    Wn = beta * W + alpha * G
    iu = np.triu_indices(W.shape[0])
    W[iu] = Wn[iu]
    W.T[iu] = Wn[iu]


def mat_times_diag_times_mat_trans_local(W, S, d, X):                     # :1119
    W[:, :] = (S * d) @ X.T


class HessianLowRank:
    """State and methods of hiopHessianLowRank for ONE rank's column slice [il, iu)."""

    def __init__(self, n_local, l_max=6, sigma0=1.0, sigma_update_strategy="sigma0", rank=0, allreduce=_no_reduce):
        self.n, self.l_max = n_local, l_max
        self.sigma = self.sigma0 = sigma0
        self.strategy = sigma_update_strategy
        self.rank, self.allreduce = rank, allreduce
        self.l_curr = -1
        self.St = np.zeros((0, n_local))
        self.Yt = np.zeros((0, n_local))
        self.L = np.zeros((0, 0))
        self.D = np.zeros(0)
        self.DhInv = np.ones(n_local)
        self.Dx = np.zeros(n_local)
        self.matrix_changed = False
        self.prev = None

    # -- :197
    def update_log_barrier_diagonal(self, Dx):
        self.DhInv = 1.0 / (self.sigma + Dx)
        self.Dx = Dx.copy()
        self.matrix_changed = True

    def _dot(self, a, b):
        return float(self.allreduce(np.array([a @ b]))[0])

    # -- :262  (x, grad_f, Jc, Jd, yc, yd are the CURRENT iterate's; returns True if a secant pair was stored)
    def update(self, x, grad_f, Jc, Jd, yc, yd):
        stored = False
        if self.l_curr >= 0:
            px, pg, pJc, pJd = self.prev
            s_new = x - px
            s_inf = float(self.allreduce_max(np.array([infnorm(s_new)]))[0]) if hasattr(self, "allreduce_max") else infnorm(s_new)
            if s_inf >= 100 * np.finfo(np.float64).eps:
                y_new = grad_f - pg
                y_new += Jc.T @ yc - pJc.T @ yc + Jd.T @ yd - pJd.T @ yd
                sTy = self._dot(s_new, y_new)
                s_nrm2 = np.sqrt(self._dot(s_new, s_new))
                y_nrm2 = np.sqrt(self._dot(y_new, y_new))
                if sTy > s_nrm2 * y_nrm2 * np.sqrt(np.finfo(np.float64).eps):
                    if self.l_max > 0:
                        YTs = self.allreduce(self.Yt @ s_new) if self.Yt.shape[0] else np.zeros(0)
                        l = self.l_curr
                        if l < self.l_max:                       # growL / growD :779-823
                            self.St = np.vstack([self.St, s_new])
                            self.Yt = np.vstack([self.Yt, y_new])
                            Ln = np.zeros((l + 1, l + 1))
                            Ln[:l, :l] = self.L
                            Ln[l, :l] = YTs
                            self.L = Ln
                            self.D = np.concatenate([self.D, [sTy]])
                            self.l_curr += 1
                        else:                                    # shift; updateL / updateD :828-867
                            self.St = np.vstack([self.St[1:], s_new])
                            self.Yt = np.vstack([self.Yt[1:], y_new])
                            lm1 = l - 1
                            Lm = self.L
                            for i in range(1, lm1):
                                for j in range(i):
                                    Lm[i, j] = Lm[i + 1, j + 1]
                            Lm[lm1, :lm1] = YTs[1:]
                            Lm[lm1, lm1] = 0.0
                            self.D = np.concatenate([self.D[1:], [sTy]])
                    st = self.strategy
                    if st == "sty":
                        self.sigma = sTy / (s_nrm2 * s_nrm2)
                    elif st == "sty_inv":
                        self.sigma = y_nrm2 * y_nrm2 / sTy
                    elif st == "snrm_ynrm":
                        self.sigma = np.sqrt(s_nrm2 * s_nrm2 / y_nrm2 / y_nrm2)
                    elif st == "sty_srnm_ynrm":
                        self.sigma = 0.5 * (sTy / (s_nrm2 * s_nrm2) + y_nrm2 * y_nrm2 / sTy)
                    else:
                        self.sigma = self.sigma0
                    self.sigma = max(min(1e8, self.sigma), 1e-8)
                    stored = True
            self.prev = (x.copy(), grad_f.copy(), Jc.copy(), Jd.copy())
        else:
            self.prev = (x.copy(), grad_f.copy(), Jc.copy(), Jd.copy())
            self.l_curr += 1
        return stored

    # -- :400
    def update_internal_bfgs_representation(self):
        l = self.St.shape[0]
        DpYtDhInvY = np.zeros((l, l))
        symm_mat_times_diag_times_mat_trans_local(0.0, DpYtDhInvY, 1.0, self.Yt, self.DhInv)
        B0DhInv = self.DhInv * self.sigma
        StB0DhInvY = np.zeros((l, l))
        mat_times_diag_times_mat_trans_local(StB0DhInvY, self.St, B0DhInv, self.Yt)
        theDiag = (B0DhInv - 1.0) * self.sigma
        StDS = np.zeros((l, l))
        symm_mat_times_diag_times_mat_trans_local(0.0, StDS, 1.0, self.St, theDiag)
        buf = self.allreduce(np.concatenate([DpYtDhInvY.ravel(), StB0DhInvY.ravel(), StDS.ravel()]))   # :459
        DpYtDhInvY = buf[:l * l].reshape(l, l) + np.diag(self.D)
        StB0DhInvYmL = buf[l * l:2 * l * l].reshape(l, l) - self.L
        StDS = buf[2 * l * l:].reshape(l, l)
        V = np.zeros((2 * l, 2 * l))
        V[l:, l:] = DpYtDhInvY
        V[:l, l:] = StB0DhInvYmL
        V[:l, :l] = StDS
        self.V_upper = V
        self.Vfull = np.triu(V) + np.triu(V, 1).T
        if l > 0:                                                 # factorizeV :633 (DSYTRF, uplo='L' of the transpose)
            a = np.asfortranarray(V.T)
            self.V_ldu, self.V_ipiv, info = lapack.dsytrf(a, lower=1)
            assert info == 0
        self.matrix_changed = False

    def solve_with_V(self, rhs):                                  # :677,:729  rhs: (2l,) or (2l, nrhs)
        if self.St.shape[0] == 0:
            return rhs
        x, info = lapack.dsytrs(self.V_ldu, self.V_ipiv, rhs, lower=1)
        assert info == 0
        return x

    # -- :495
    def solve(self, rhsx):
        if self.matrix_changed:
            self.update_internal_bfgs_representation()
        x = rhsx * self.DhInv
        l = self.St.shape[0]
        ytx = self.allreduce(self.Yt @ x) if l else np.zeros(0)
        stx = self.allreduce(self.St @ (x * self.sigma)) if l else np.zeros(0)
        sol = self.solve_with_V(np.concatenate([stx, ytx]))
        spart, ypart = sol[:l], sol[l:]
        result = (self.St.T @ spart) * self.sigma + self.Yt.T @ ypart
        result *= self.DhInv
        return x - result

    # -- :549   W = beta*W + alpha*X*this^-1*X^T
    def sym_mat_times_inverse_times_mat_trans(self, beta, W, alpha, X):
        if self.matrix_changed:
            self.update_internal_bfgs_representation()
        l = self.St.shape[0]
        k = W.shape[0]
        symm_mat_times_diag_times_mat_trans_local(beta if self.rank == 0 else 0.0, W, alpha, X, self.DhInv)   # :568-571
        S1 = np.zeros((k, l))
        Y1 = np.zeros((k, l))
        mat_times_diag_times_mat_trans_local(S1, X, self.DhInv * self.sigma, self.St)
        mat_times_diag_times_mat_trans_local(Y1, X, self.DhInv, self.Yt)
        S1Y1 = self.allreduce(np.hstack([S1, Y1]).ravel()).reshape(k, 2 * l)       # :590
        W[:, :] = self.allreduce(W.ravel().copy()).reshape(k, k)                    # :591
        S1, Y1 = S1Y1[:, :l], S1Y1[:, l:]
        if l > 0:
            S2Y2 = self.solve_with_V(np.asfortranarray(S1Y1.T)).T                   # :606 (k x 2l)
            S2, Y2 = S2Y2[:, :l], S2Y2[:, l:]
            W -= alpha * (S1 @ S2.T)                                                # :614
            W -= alpha * (Y1 @ Y2.T)                                                # :618
        return W

    # -- :974  y = beta*y + alpha*(B0 + Dx + sum b b^T - a a^T) x
    def times_vec(self, beta, y, alpha, x, add_log_term=True):
        l = self.St.shape[0]
        a, b = [], []
        eps = np.finfo(np.float64).eps
        for k in range(l):
            yk, sk = self.Yt[k], self.St[k]
            skTyk = self._dot(yk, sk)
            if skTyk < eps:
                skTyk = eps
            bk = yk / np.sqrt(skTyk)
            ak = sk * self.sigma
            for i in range(k):
                ak = ak + self._dot(b[i], sk) * b[i]
                ak = ak - self._dot(a[i], sk) * a[i]
            ak = ak / np.sqrt(self._dot(ak, sk))
            a.append(ak)
            b.append(bk)
        out = beta * y
        if add_log_term:
            out = out + alpha * x * self.Dx
        out = out + alpha * self.sigma * x
        for k in range(l):
            out = out + alpha * self._dot(b[k], x) * b[k]
            out = out - alpha * self._dot(a[k], x) * a[k]
        y[:] = out

    def dense_matrix_local(self):
        """B (without Dx) as a dense n x n matrix through the compact formula — single-rank test helper
        (Byrd-Nocedal-Schnabel: B = B0 - [B0 S, Y] [[S^T B0 S, L],[L^T, -D]]^-1 [S^T B0; Y^T])."""
        n = self.n
        l = self.St.shape[0]
        B = np.eye(n) * self.sigma
        if l:
            S, Y = self.St.T, self.Yt.T
            Mid = np.block([[self.sigma * S.T @ S, self.L], [self.L.T, -np.diag(self.D)]])
            Wm = np.hstack([self.sigma * S, Y])
            B = B - Wm @ np.linalg.solve(Mid, Wm.T)
        return B


# =====================================================================================
# hiopKKTLinSysLowRank (reference: src/Optimization/hiopKKTLinSys.cpp:1057-1330)
# =====================================================================================
def solve_with_refin(N, rhs):                                     # :1192-1330
    """DPOSVX(FACT='E') + the reference's residual loop (inf-norm tol 1e-8, <= 3 Cholesky refinements).
    N is the full symmetric k x k matrix (both triangles, as symmMatTimesDiagTimesMatTrans_local leaves it)."""
    k = N.shape[0]
    if k == 0:
        return rhs.copy(), 0
    A = np.asfortranarray(N.T.copy())
    out = lapack.dposvx(A, rhs.reshape(-1, 1), fact="E", lower=1)
    X = out[6][:, 0].copy() if isinstance(out[6], np.ndarray) and out[6].ndim == 2 else None
    if X is None:   # scipy return order: a_s, lu, equed, s, b_s, x, rcond, ferr, berr, info (version dependent)
        for o in out:
            if isinstance(o, np.ndarray) and o.shape == (k, 1):
                X = o[:, 0].copy()
    info = out[-1]
    n_refin = 0
    x = X.copy()
    while True:
        x = X.copy()
        resid = rhs - N @ x
        if infnorm(resid) < 1e-8 or n_refin >= 3:
            break
        c, info2 = lapack.dpotrf(np.asfortranarray(N.T.copy()), lower=1)
        dxr, _ = lapack.dpotrs(c, resid, lower=1)
        x = x + dxr
        n_refin += 1
        X = x   # NOTE: the reference re-copies X (un-refined) at the top of its loop (:1248); refinement
        #         results are only kept from the LAST pass.  Keeping the refined x is mathematically the
        #         intended behaviour and differs from the reference only when DPOSVX misses 1e-8.
    return x, int(info)


class KKTLinSysLowRank:
    def __init__(self, hess: HessianLowRank, m_eq, m_ineq):
        self.H, self.m_eq, self.m_ineq = hess, m_eq, m_ineq

    def update(self, Dx, Dd, Jc, Jd):                             # :1057-1096 (Dx, Dd computed by the caller's vector ops)
        self.H.update_log_barrier_diagonal(Dx)
        self.Dd_inv = 1.0 / Dd
        self.J = np.vstack([Jc, Jd])                              # :1127-1128

    def solve_compressed(self, rx, ryc, ryd):                     # :1110-1187
        H, J = self.H, self.J
        k = J.shape[0]
        N = np.zeros((k, k))
        H.sym_mat_times_inverse_times_mat_trans(0.0, N, 1.0, J)   # :1132
        idx = np.arange(self.m_ineq) + self.m_eq
        N[idx, idx] += self.Dd_inv                                # :1135
        dx = H.solve(rx)                                          # :1147
        rhs = np.concatenate([ryc, ryd])
        Jdx = J @ dx
        if H.rank != 0:                                           # timesVec: only rank 0 applies beta (:466)
            rhs_loc = Jdx
        else:
            rhs_loc = -rhs + Jdx
        rhs = H.allreduce(rhs_loc.copy())                         # :1157 (Allreduce k)
        sol, ierr = solve_with_refin(N, rhs)                      # :1169
        self.last_N = N
        dyc, dyd = sol[:self.m_eq].copy(), sol[self.m_eq:].copy()
        rx2 = rx - J.T @ sol                                      # :1178
        dx = H.solve(rx2)                                         # :1180
        return ierr == 0, dx, dyc, dyd


# =====================================================================================
# hiopMatrixSparseTriplet — the ASSEMBLY surface (reference: src/LinAlg/hiopMatrixSparseTriplet.cpp).
# A triplet matrix here is a tuple of three numpy arrays (iRow, jCol, val) modified in place; sources are sorted by
# (row, column).  The loops follow the reference statement by statement (small sizes only).
# =====================================================================================
def sp_copy_sub_diagonal_from(T, start_on_dest_diag, num_elems, d, start_on_nnz_idx, scal=1.0):      # :216-233
    i, j, v = T
    for r in range(num_elems):
        i[r + start_on_nnz_idx] = j[r + start_on_nnz_idx] = r + start_on_dest_diag
        v[r + start_on_nnz_idx] = scal * d[r]


def sp_set_sub_diagonal_to(T, start_on_dest_diag, num_elems, c, start_on_nnz_idx):                    # :235-249
    i, j, v = T
    for r in range(num_elems):
        i[r + start_on_nnz_idx] = j[r + start_on_nnz_idx] = r + start_on_dest_diag
        v[r + start_on_nnz_idx] = c


def sp_copy_rows_from(T, S, rows_idxs):                                                               # :562-611
    i, j, v = T
    si, sj, sv = S
    its, itd = 0, 0
    for row_dest, row_src in enumerate(rows_idxs):
        while its < si.size and si[its] < row_src:
            its += 1
        while its < si.size and si[its] == row_src:
            i[itd], j[itd], v[itd] = row_dest, sj[its], sv[its]
            itd += 1
            its += 1
    return itd


def sp_copy_rows_block_from(T, S, rows_src_idx_st, n_rows, rows_dest_idx_st, dest_nnz_st):            # :619-669
    i, j, v = T
    si, sj, sv = S
    its, itd = 0, dest_nnz_st
    for row_add in range(n_rows):
        row_src, row_dest = rows_src_idx_st + row_add, rows_dest_idx_st + row_add
        while its < si.size and si[its] < row_src:
            its += 1
        while its < si.size and si[its] == row_src:
            i[itd], j[itd], v[itd] = row_dest, sj[its], sv[its]
            itd += 1
            its += 1
    return itd


def sp_copy_diag_matrix_to_subblock(T, src_val, dest_row_st, col_dest_st, dest_nnz_st, nnz_to_copy):  # :671-687
    i, j, v = T
    for e in range(nnz_to_copy):
        i[dest_nnz_st + e], j[dest_nnz_st + e], v[dest_nnz_st + e] = dest_row_st + e, col_dest_st + e, src_val


def sp_copy_diag_matrix_to_subblock_w_pattern(T, dx, dest_row_st, dest_col_st, dest_nnz_st, ix):      # :689-719
    i, j, v = T
    k, found = dest_nnz_st, 0
    for q in range(ix.size):
        if ix[q] != 0.0:
            i[k], j[k], v[k] = dest_row_st + found, dest_col_st + found, dx[q]
            k += 1
            found += 1
    return found


def sp_copy_submatrix_from(T, S, dest_row_st, dest_col_st, dest_nnz_st, offdiag_only=False, trans=False):   # :1042-1108
    i, j, v = T
    si, sj, sv = (S[1], S[0], S[2]) if trans else S
    k = dest_nnz_st
    for q in range(si.size):
        if offdiag_only and si[q] == sj[q]:
            continue
        i[k], j[k], v[k] = dest_row_st + si[q], dest_col_st + sj[q], sv[q]
        k += 1
    return k


def sp_set_submatrix_to_constant_diag_w_pattern(T, scalar, dest_row_st, dest_col_st, dest_nnz_st, ix, rowpattern):   # :1110-1168
    i, j, v = T
    k, found = dest_nnz_st, 0
    for q in range(ix.size):
        if ix[q] != 0.0:
            i[k] = dest_row_st + (found if rowpattern else q)
            j[k] = dest_col_st + (q if rowpattern else found)
            v[k] = scalar
            k += 1
            found += 1
    return found


def sp_set_jac_fr(T, n, Jc, m_c, Jd, m_d):                                                            # :790-922
    """this = [Jc -I I 0 0; Jd 0 0 -I I]; returns the number of entries written."""
    i, j, v = T
    k = 0
    for (si, sj, sv), m, row0, colp in ((Jc, m_c, 0, n), (Jd, m_d, m_c, n + 2 * m_c)):
        for r in range(m):
            for q in np.nonzero(si == r)[0]:
                i[k], j[k], v[k] = r + row0, sj[q], sv[q]
                k += 1
            i[k], j[k], v[k] = r + row0, colp + r, -1.0
            k += 1
            i[k], j[k], v[k] = r + row0, colp + m + r, 1.0
            k += 1
    return k


def spsym_set_hess_fr(T, H, m_h, add_diag):                                                           # :1374-1497
    i, j, v = T
    hi, hj, hv = H
    k = 0
    if m_h > 0:
        for r in range(m_h):
            rows = np.nonzero(hi == r)[0]
            i[k], j[k], v[k] = r, r, add_diag[r]
            q0 = 0
            if rows.size and hi[rows[0]] == hj[rows[0]]:
                v[k] += hv[rows[0]]
                q0 = 1
            k += 1
            for q in rows[q0:]:
                i[k], j[k], v[k] = r, hj[q], hv[q]
                k += 1
    else:
        for r in range(add_diag.size):
            i[k], j[k], v[k] = r, r, add_diag[r]
            k += 1
    return k
