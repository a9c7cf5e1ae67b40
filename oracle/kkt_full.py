"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the full-space layer of HiOp's KKT hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(libhiopamd.so) never does.  Every function cites the reference lines it restates:

  hiopKKTLinSys.cpp:218-314   hiopKKTLinSys::compute_directions_for_full_space
  hiopKKTLinSys.cpp:316-376   hiopKKTLinSysCurvCheck::factorize               (inertia-correction loop)
  hiopKKTLinSys.cpp:543-583   hiopKKTLinSysCompressedXYcYd::update
  hiopKKTLinSys.cpp:585-690   hiopKKTLinSysCompressedXYcYd::computeDirections
  hiopKKTLinSys.cpp:911-961   hiopKKTLinSys::compute_directions_w_IR
  hiopKKTLinSys.cpp:1619-1736 hiopMatVecKKTFullOpr::times_vec
  hiopKKTLinSys.cpp:1900-1909 hiopPrecondKKTOpr::times_vec
  hiopKrylovSolver.cpp:390-700 hiopBiCGStabSolver::solve
  hiopPDPerturbation.cpp:69-395 hiopPDPerturbation / hiopPDPerturbationPrimalFirstScalar
  hiopFactAcceptor.cpp:63-104 hiopFactAcceptorIC::requireReFactorization
  hiopKKTLinSysDense.hpp:84-212 hiopKKTLinSysDenseXYcYd::build_kkt_matrix / solveCompressed
  hiopVectorCompoundPD.cpp:99-210 part order of the 12-part compound vector

Parity pin: no golden vectors exist in the reference for these functions (its tests exercise them only through
whole-solver runs); the restatement is pinned by (i) algebraic identities checked in tests/test_oracle_kkt_full.py
(K * computeDirections(r) == r on the full 12-block system, BiCGStab fixed points) and (ii) the reference's
selfcheck objective values reproduced by oracle/ipm.py through this layer.
"""
import numpy as np
from . import hiop_oracle as ho

ITER_PARTS = ("x", "d", "yc", "yd", "sxl", "sxu", "sdl", "sdu", "zl", "zu", "vl", "vu")       # CompoundPD.cpp:99-153
RESID_PARTS = ("rx", "rd", "ryc", "ryd", "rxl", "rxu", "rdl", "rdu", "rszl", "rszu", "rsvl", "rsvu")  # :155-210


def part_sizes(nx, nd, nyc, nyd):
    return [nx, nd, nyc, nyd, nx, nx, nd, nd, nx, nx, nd, nd]


def pack(parts, names):
    return np.concatenate([np.asarray(parts[k], dtype=np.float64) for k in names])


def unpack(slab, names, sizes):
    out, o = {}, 0
    for k, s in zip(names, sizes):
        out[k] = slab[o:o + s].copy()
        o += s
    return out


# ---------------------------------------------------------------------------------------------------------
# hiopPDPerturbationPrimalFirstScalar (hiopPDPerturbation.cpp:69-395) — scalar state machine, host side
# ---------------------------------------------------------------------------------------------------------
class PDPerturbationPrimalFirstScalar:
    NOT_EST, NOT_DEG, DEG = 0, 1, 2
    NO_TEST, C0W0, CPOSW0, C0WPOS, CPOSWPOS = range(5)

    def __init__(self, delta_w_min_bar=1e-20, delta_w_max_bar=1e20, delta_0_bar=1e-4, kappa_w_minus=1. / 3,
                 kappa_w_plus_bar=100., kappa_w_plus=8., delta_c_bar=1e-8, kappa_c=0.25):   # hiopOptions.cpp:1080-1123
        self.delta_w_min_bar, self.delta_w_max_bar, self.delta_w_0_bar = delta_w_min_bar, delta_w_max_bar, delta_0_bar
        self.kappa_w_minus, self.kappa_w_plus_bar, self.kappa_w_plus = kappa_w_minus, kappa_w_plus_bar, kappa_w_plus
        self.delta_c_bar, self.kappa_c = delta_c_bar, kappa_c
        self.wx = self.wd = self.cc = self.cd = 0.0                 # *_curr_db_
        self.wx_last = self.wd_last = self.cc_last = self.cd_last = 0.0
        self.hess_degenerate = self.jac_degenerate = self.NOT_EST
        self.num_degen_iters, self.num_degen_max_iters = 0, 3
        self.test_type = self.NO_TEST
        self.mu = 1e-8

    def set_mu(self, mu):
        self.mu = mu

    def deltas(self):
        return self.wx, self.wd, self.cc, self.cd

    def compute_delta_c(self):                                      # :361
        return self.delta_c_bar * self.mu ** self.kappa_c

    def update_degeneracy_type(self):                               # :108-157
        t = self.test_type
        if t == self.NO_TEST:
            return
        if t == self.C0W0:
            if self.hess_degenerate == self.NOT_EST and self.jac_degenerate == self.NOT_EST:
                self.hess_degenerate = self.jac_degenerate = self.NOT_DEG
            elif self.hess_degenerate == self.NOT_EST:
                self.hess_degenerate = self.NOT_DEG
            elif self.jac_degenerate == self.NOT_EST:
                self.jac_degenerate = self.NOT_DEG
        elif t == self.CPOSW0:
            if self.hess_degenerate == self.NOT_EST:
                self.hess_degenerate = self.NOT_DEG
            if self.jac_degenerate == self.NOT_EST:
                self.num_degen_iters += 1
                if self.num_degen_iters >= self.num_degen_max_iters:
                    self.jac_degenerate = self.DEG
        elif t == self.C0WPOS:
            if self.jac_degenerate == self.NOT_EST:
                self.jac_degenerate = self.NOT_DEG
            if self.hess_degenerate == self.NOT_EST:
                self.num_degen_iters += 1
                if self.num_degen_iters >= self.num_degen_max_iters:
                    self.hess_degenerate = self.DEG
        elif t == self.CPOSWPOS:
            self.num_degen_iters += 1
            if self.num_degen_iters >= self.num_degen_max_iters:
                self.hess_degenerate = self.jac_degenerate = self.DEG

    def _guts_wrong_inertia(self):                                  # :331-358 (its two out-arguments are never written)
        if self.wx == 0.:
            if self.wx_last == 0.:
                self.wx = self.delta_w_0_bar
            else:
                self.wx = max(self.delta_w_min_bar, self.wx_last * self.kappa_w_minus)
        else:
            if self.wx_last == 0. or 1e5 * self.wx_last < self.wx:
                self.wx = self.kappa_w_plus_bar * self.wx
            else:
                self.wx = self.kappa_w_plus * self.wx
        self.wd = self.wx
        if self.wx > self.delta_w_max_bar:
            self.wx_last = self.wd_last = 0.
            return False
        return True

    def compute_initial_deltas(self):                               # :161-212
        delta_temp = delta_temp2 = 0.0
        self.update_degeneracy_type()
        if self.wx > 0.:
            self.wx_last = self.wx
        if self.wd > 0.:
            self.wd_last = self.wd
        if self.cc > 0.:
            self.cc_last = self.cc
        if self.cd > 0.:
            self.cd_last = self.cd
        if self.hess_degenerate == self.NOT_EST or self.jac_degenerate == self.NOT_EST:
            self.test_type = self.C0W0
        else:
            self.test_type = self.NO_TEST
        delta_temp = self.compute_delta_c() if self.jac_degenerate == self.DEG else 0.0
        self.cc = self.cd = delta_temp
        if self.hess_degenerate == self.DEG:
            self.wx = self.wd = 0.
            if not self._guts_wrong_inertia():
                return False
            # NOTE (:203-209): the reference then overwrites the just-computed values with its two locals, which
            # guts_of_compute_perturb_wrong_inertia never assigns: delta_wx := delta_c (or 0), delta_wd := 0.
        else:
            delta_temp = delta_temp2 = 0.
        self.wx, self.wd = delta_temp, delta_temp2
        return True

    def compute_perturb_wrong_inertia(self):                        # :215-243
        self.update_degeneracy_type()
        ret = self._guts_wrong_inertia()
        if not ret and self.cc == 0.:
            self.wx = self.wd = 0.
            self.cc = self.cd = self.compute_delta_c()
            self.test_type = self.NO_TEST
            if self.hess_degenerate == self.DEG:
                self.hess_degenerate = self.NOT_EST
            ret = self._guts_wrong_inertia()
        return ret

    def compute_perturb_singularity(self):                          # :248-325
        bret = True
        if self.hess_degenerate == self.NOT_EST or self.jac_degenerate == self.NOT_EST:
            t = self.test_type
            if t == self.C0W0:
                if self.jac_degenerate == self.NOT_EST:
                    self.cc = self.cd = self.compute_delta_c()
                    self.test_type = self.CPOSW0
                else:
                    if not self._guts_wrong_inertia():
                        bret = False
                    else:
                        self.test_type = self.C0WPOS
            elif t == self.CPOSW0:
                self.cd = self.cc = 0.
                if not self._guts_wrong_inertia():
                    bret = False
                else:
                    self.test_type = self.C0WPOS
            elif t == self.C0WPOS:
                self.cc = self.cd = self.compute_delta_c()
                if not self._guts_wrong_inertia():
                    bret = False
                else:
                    self.test_type = self.CPOSWPOS
            elif t == self.CPOSWPOS:
                if not self._guts_wrong_inertia():
                    bret = False
            else:
                raise AssertionError("something went wrong - should not get here")   # :302
        else:
            if self.cc > 0.:
                if not self._guts_wrong_inertia():
                    bret = False
            else:
                self.cd = self.cc = self.compute_delta_c()
        return bret


class PDPerturbationDualFirstScalar(PDPerturbationPrimalFirstScalar):
    """hiopPDPerturbationDualFirstScalar (hiopPDPerturbation.cpp:457-626): for the normal-equation KKT, where a wrong inertia
    means the condensed matrix is not positive definite — the dual regularisation is tried first, the primal one after."""

    def __init__(self, *a, delta_c_min_bar=1e-20, kappa_c_plus=10., **kw):         # :459-462
        super().__init__(*a, **kw)
        self.delta_c_min_bar, self.kappa_c_plus = delta_c_min_bar, kappa_c_plus

    def _dual_perturb_impl(self):                                                   # :558-589
        if self.cc == 0.:
            if self.cc_last == 0.:
                self.cc = max(self.delta_c_min_bar, self.delta_c_bar * self.mu ** self.kappa_c)
            else:
                self.cc = max(self.delta_c_min_bar, self.cc_last * self.kappa_w_minus)
        else:
            if self.cc_last == 0. or 1e5 * self.cc_last < self.cc:
                self.cc = self.kappa_w_plus_bar * self.cc
            else:
                self.cc = self.kappa_c_plus * self.cc
        self.cd = self.cc
        if self.cc > self.delta_w_max_bar:
            self.cc_last = self.cd_last = 0.
            return False
        return True

    def _primal_perturb_impl(self):                                                 # :591-620
        return self._guts_wrong_inertia()                                           # the same arithmetic as :331-358

    def compute_initial_deltas(self):                                               # :470-512
        self.update_degeneracy_type()
        if self.wx > 0.:
            self.wx_last = self.wx
        if self.wd > 0.:
            self.wd_last = self.wd
        if self.cc > 0.:
            self.cc_last = self.cc
        if self.cd > 0.:
            self.cd_last = self.cd
        if self.hess_degenerate == self.NOT_EST or self.jac_degenerate == self.NOT_EST:
            self.test_type = self.C0W0
        else:
            self.test_type = self.NO_TEST
        self.cc = self.cd = 0.
        if self.jac_degenerate == self.DEG:
            if not self._dual_perturb_impl():
                return False
        self.wx = self.wd = 0.
        if self.hess_degenerate == self.DEG:
            if not self._primal_perturb_impl():
                return False
        return True

    def compute_perturb_wrong_inertia(self):                                        # :514-547
        self.update_degeneracy_type()
        ret = self._dual_perturb_impl()
        if not ret and self.wx == 0.:
            self.cc = self.cd = 0.
            ret = self._primal_perturb_impl()
            if not ret:
                return ret
            self.test_type = self.NO_TEST
            if self.jac_degenerate == self.DEG:
                self.jac_degenerate = self.NOT_EST
            ret = self._dual_perturb_impl()
        return ret

    def compute_perturb_singularity(self):                                          # :549-556
        return self.compute_perturb_wrong_inertia()


def randomized(base):
    """hiopPDPerturbationPrimalFirstRand / DualFirstRand (hiopPDPerturbation.cpp:414-455, :670-711): the scalar machine of
    `base`; deltas() hands out VECTORS, uniform in [min_uniform_ratio, max_uniform_ratio] x scalar (0.9, 1.0;
    hiopPDPerturbation.hpp:53-54).  The draw itself comes from `source(name, n, lo, hi)` so that a test can feed the very
    vectors another implementation drew (the reference uses the host's std generator: there is no stream to reproduce)."""

    class Rand(base):
        min_uniform_ratio, max_uniform_ratio = 0.9, 1.0

        def __init__(self, sizes, source=None, **kw):
            super().__init__(**kw)
            self.sizes = sizes                                   # nx, nd, nyc, nyd
            rng = np.random.default_rng(0)
            self.source = source or (lambda name, n, lo, hi: rng.uniform(lo, hi, n) if hi > lo else np.full(n, lo))

        def deltas(self):
            """One draw per value of the scalars (set_delta_curr_vec runs when the machine changes them); callers between two
            changes (build, the 12-block operator, test_direction) see the same vectors."""
            key = (self.wx, self.wd, self.cc, self.cd)
            if getattr(self, "_key", None) != key:
                self._key = key
                self._vecs = tuple(self.source(name, n, self.min_uniform_ratio * sc, self.max_uniform_ratio * sc)
                                   for name, n, sc in zip(("wx", "wd", "cc", "cd"), self.sizes, key))
            return self._vecs
    Rand.__name__ = base.__name__.replace("Scalar", "Rand")
    return Rand


PDPerturbationPrimalFirstRand = randomized(PDPerturbationPrimalFirstScalar)
PDPerturbationDualFirstRand = randomized(PDPerturbationDualFirstScalar)


class PDPerturbationNull:
    """hiopPDPerturbationNull (hiopPDPerturbation.hpp:343-370): all deltas stay zero (quasi-Newton path)."""
    wx = wd = cc = cd = 0.0

    def set_mu(self, mu):
        pass

    def deltas(self):
        return 0.0, 0.0, 0.0, 0.0

    def compute_initial_deltas(self):
        return True

    def compute_perturb_wrong_inertia(self):
        return True

    def compute_perturb_singularity(self):
        return True


def require_refactorization(perturb, n_required_neg_eig, n_neg_eig):     # hiopFactAcceptor.cpp:63-104
    if n_required_neg_eig > 0:
        if n_neg_eig < 0:
            return 1 if perturb.compute_perturb_singularity() else -1
        if n_neg_eig != n_required_neg_eig:
            return 1 if perturb.compute_perturb_wrong_inertia() else -1
        return 0
    if n_neg_eig != 0:
        return 1 if perturb.compute_perturb_wrong_inertia() else -1
    return 0


def require_refactorization_inertia_free(perturb, n_required_neg_eig, n_neg_eig, force_reg=False):   # :106-155
    if n_required_neg_eig > 0:
        if n_neg_eig < 0:
            return 1 if perturb.compute_perturb_singularity() else -1
        if not force_reg:
            return 0
        return 1 if perturb.compute_perturb_wrong_inertia() else -1
    if n_neg_eig < 0:
        return 1 if perturb.compute_perturb_wrong_inertia() else -1
    if not force_reg:
        return 0
    return 1 if perturb.compute_perturb_wrong_inertia() else -1


# ---------------------------------------------------------------------------------------------------------
# providers: the compressed XYcYd systems the full-space layer sits on
# ---------------------------------------------------------------------------------------------------------
class MdsProvider:
    """hiopKKTLinSysCompressedMDSXYcYd + the MDS matrices' timesVec (hiopMatrixMDS.hpp:68-81,310-323)."""

    def __init__(self, k: ho.KKTLinSysCompressedMDSXYcYd):
        self.k = k
        self.nx, self.nd, self.nyc, self.nyd = k.nxs + k.nxd, k.nineq, k.neq, k.nineq

    def set_diagonals(self, Dx, Dd):
        self.k.Dx, self.k.Dd = Dx, Dd

    def build(self, dwx, dwd, dcc, dcd):
        self.k.build_kkt_matrix(dwx, dwd, dcc, dcd)

    def factorize(self):
        return self.k.factorize_with_curv_check()

    def solve(self, rx, ryc, ryd):
        return self.k.solve_compressed(rx, ryc, ryd)

    @property
    def Dd_inv(self):
        return self.k.Dd_inv

    def hess_times_vec(self, x):                                  # hiopMatrixMDS.hpp:310-318
        k = self.k
        y = np.zeros(self.nx)
        i, j = k.Hss_ij
        ys = y[:k.nxs]
        np.add.at(ys, i, k.Hss_val * x[j])                        # hiopMatrixSparseTriplet.cpp:941-958
        off = i != j
        np.add.at(ys, j[off], k.Hss_val[off] * x[i[off]])
        y[k.nxs:] = k.Hdd @ x[k.nxs:]
        return y

    def _jac(self, which):
        k = self.k
        return (k.Jcs_ij, k.Jcs_val, k.Jcd, k.neq) if which == "c" else (k.Jds_ij, k.Jds_val, k.Jdd, k.nineq)

    def jac_times_vec(self, which, x):                            # hiopMatrixMDS.hpp:68-74
        (i, j), v, De, m = self._jac(which)
        y = np.zeros(m)
        np.add.at(y, i, v * x[j])
        return y + De @ x[self.k.nxs:]

    def jac_trans_times_vec(self, which, yv):                     # hiopMatrixMDS.hpp:75-81
        (i, j), v, De, m = self._jac(which)
        out = np.zeros(self.nx)
        np.add.at(out[:self.k.nxs], j, v * yv[i])
        out[self.k.nxs:] = De.T @ yv
        return out


class DenseXYcYdProvider:
    """hiopKKTLinSysDenseXYcYd (hiopKKTLinSysDense.hpp:71-225): the whole XYcYd system as one dense matrix."""

    def __init__(self, H, Jc, Jd, linsolver=None):
        self.H, self.Jc, self.Jd = H, Jc, Jd
        self.nx, self.nyc, self.nyd = H.shape[0], Jc.shape[0], Jd.shape[0]
        self.nd = self.nyd
        n = self.nx + self.nyc + self.nyd
        self.linsys = linsolver if linsolver is not None else ho.LinSolverSymDenseLapack(n)

    def set_diagonals(self, Dx, Dd):
        self.Dx, self.Dd = Dx, Dd

    def build(self, dwx, dwd, dcc, dcd):                          # :84-170
        nx, neq, nineq = self.nx, self.nyc, self.nyd
        M = self.linsys.M
        M[:] = 0.0                                                # :134
        ho.add_upper_to_sym_upper(self.H, 0, 1.0, M)              # :137
        ho.trans_add_to_sym_upper(self.Jc, 0, nx, 1.0, M)         # :139
        ho.trans_add_to_sym_upper(self.Jd, 0, nx + neq, 1.0, M)   # :140
        idx = np.arange(nx)
        M[idx, idx] += self.Dx                                    # :142
        M[idx, idx] += dwx                                        # :143
        self.Dd_inv = 1.0 / (dwd + self.Dd)                       # :146-152
        idx = np.arange(nineq) + nx + neq
        M[idx, idx] -= self.Dd_inv                                # :155
        # :160 literally `Msys.addSubDiagonal(-1, nx, *delta_cd_)`: nineq entries starting at diagonal position nx
        idx = np.arange(nineq) + nx
        M[idx, idx] -= dcd
        return M

    def factorize(self):                                          # hiopKKTLinSys.cpp:310-313
        return self.linsys.matrix_changed()

    def solve(self, rx, ryc, ryd):                                # :172-212
        rhs = np.concatenate([rx, ryc, ryd])
        ok = self.linsys.solve(rhs)
        nx, nyc = self.nx, self.nyc
        return ok, rhs[:nx].copy(), rhs[nx:nx + nyc].copy(), rhs[nx + nyc:].copy()

    def hess_times_vec(self, x):
        return self.H @ x

    def jac_times_vec(self, which, x):
        return (self.Jc if which == "c" else self.Jd) @ x

    def jac_trans_times_vec(self, which, y):
        return (self.Jc if which == "c" else self.Jd).T @ y


class DenseXDYcYdProvider(DenseXYcYdProvider):
    """hiopKKTLinSysDenseXDYcYd (hiopKKTLinSysDense.hpp:229-380): unknowns [x | d | yc | yd]."""
    xd_form = True

    def __init__(self, H, Jc, Jd):
        n = H.shape[0] + Jc.shape[0] + 2 * Jd.shape[0]
        super().__init__(H, Jc, Jd, linsolver=ho.LinSolverSymDenseLapack(n))

    def build(self, dwx, dwd, dcc, dcd):                          # :249-328
        nx, neq, nineq = self.nx, self.nyc, self.nyd
        M = self.linsys.M
        M[:] = 0.0
        ho.add_upper_to_sym_upper(self.H, 0, 1.0, M)              # :286
        ho.trans_add_to_sym_upper(self.Jc, 0, nx + nineq, 1.0, M)          # :288
        ho.trans_add_to_sym_upper(self.Jd, 0, nx + nineq + neq, 1.0, M)    # :289
        idx = np.arange(nx)
        M[idx, idx] += self.Dx                                    # :292
        M[idx, idx] += dwx                                        # :293
        idx = np.arange(nineq) + nx
        M[idx, idx] += self.Dd                                    # :295
        M[idx, idx] += dwd                                        # :296
        M[idx, np.arange(nineq) + nx + nineq + neq] -= 1.0        # :299-307
        idx = np.arange(nineq) + nx + nineq                       # :312 literally addSubDiagonal(-1, nx+nineq, delta_cd)
        M[idx, idx] -= dcd
        return M

    def solve_xd(self, rx, rd, ryc, ryd):                         # :330-362
        rhs = np.concatenate([rx, rd, ryc, ryd])
        ok = self.linsys.solve(rhs)
        nx, nyc, nyd = self.nx, self.nyc, self.nyd
        return (ok, rhs[:nx].copy(), rhs[nx:nx + nyd].copy(), rhs[nx + nyd:nx + nyd + nyc].copy(),
                rhs[nx + nyd + nyc:].copy())


class LowRankProvider:
    """hiopKKTLinSysLowRank (hiopKKTLinSys.cpp:1030-1187) for ONE rank's column slice; its update does not
    factorize (N is rebuilt inside solveCompressed) and all perturbations are zero (hiopPDPerturbationNull)."""

    def __init__(self, K: ho.KKTLinSysLowRank, Jc, Jd):
        self.K, self.Jc, self.Jd = K, Jc, Jd
        self.nx, self.nyc, self.nyd = Jc.shape[1], Jc.shape[0], Jd.shape[0]
        self.nd = self.nyd

    def set_diagonals(self, Dx, Dd):
        self.K.update(Dx, Dd, self.Jc, self.Jd)

    def build(self, *deltas):
        pass

    def factorize(self):
        return self.nyc + self.nyd

    def solve(self, rx, ryc, ryd):
        return self.K.solve_compressed(rx.copy(), ryc, ryd)

    @property
    def Dd_inv(self):
        return self.K.Dd_inv

    def hess_times_vec(self, x):                                  # hiopHessianLowRank.cpp:1061 (no log-barrier term)
        y = np.zeros_like(x)
        self.K.H.times_vec(0.0, y, 1.0, x, add_log_term=False)
        return y

    def jac_times_vec(self, which, x):                            # all-reduced over the column partition
        J = self.Jc if which == "c" else self.Jd
        return self.K.H.allreduce(J @ x)

    def jac_trans_times_vec(self, which, y):
        return (self.Jc if which == "c" else self.Jd).T @ y


# ---------------------------------------------------------------------------------------------------------
# BiCGStab (hiopKrylovSolver.cpp:390-700) on flat numpy vectors
# ---------------------------------------------------------------------------------------------------------
def bicgstab(A, ML, b, tol, maxit, dot=None, MR=None, x0=None, ref_exit=True):
    """Returns (x, converged, flag, iter, abs_resid, rel_resid).  A, ML, MR: callables v -> matrix*v; the preconditioned
    direction is MR(ML(v)) (hiopKrylovSolver.cpp:504-511).  x0: start vector (default 0).
    ref_exit (default): the reference's 'tol is too small' exits copy xk over b BEFORE breaking (:561-566, :639-644), so the closing
    comparison of the minimal-residual iterate (:671-688) runs against that overwritten vector; False: against the original b."""
    if MR is not None:
        ML0 = ML
        ML = (lambda v: MR(ML0(v))) if ML0 is not None else MR
    if dot is None:
        dot = lambda u, v: float(u @ v)
    nrm = lambda u: np.sqrt(dot(u, u))
    n2b = nrm(b)
    if n2b == 0.0:                                                 # :405-413
        return np.zeros_like(b), True, 0, 0.0, 0.0, 0.0
    xk = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)   # x0 = 0 (set_x0(0.0), hiopKKTLinSys.cpp:941)
    flag = 1
    imin = 0.0
    tolb = tol * n2b
    xmin = xk.copy()
    res = b - A(xk)                                                # :446-449
    normr = nrm(res)
    abs_resid = normr
    if normr <= tolb:                                              # :453-461
        return xk, True, 0, 0.0, normr, normr / n2b
    rt = res.copy()
    normrmin = normr
    rho = omega = 1.0
    stagsteps = moresteps = 0
    eps = np.finfo(np.float64).eps
    maxmsteps, maxstagsteps = 100, 3
    alpha = 0.0
    it = 0.0
    pk = v = None
    ii = 0
    while ii < maxit:
        rho1 = rho
        rho = dot(rt, res)
        if rho == 0 or abs(rho) > 1e40:                            # :483-487
            flag, it = 4, ii + 1 - 0.5
            break
        if ii == 0:
            pk = res.copy()
        else:
            beta = rho / rho1 * (alpha / omega)
            if beta == 0 or abs(beta) > 1e40:
                flag, it = 4, ii + 1 - 0.5
                break
            pk = (pk - omega * v) * beta + res                     # :498-500
        ph = ML(pk) if ML is not None else pk.copy()
        v = A(ph)
        rtv = dot(rt, v)
        if rtv == 0.0 or abs(rtv) > 1e40:
            flag, it = 4, ii + 1 - 0.5
            break
        alpha = rho / rtv
        if abs(alpha) > 1e20:
            flag, it = 4, ii + 1 - 0.5
            break
        if nrm(ph) * abs(alpha) < eps * nrm(xk):                   # :531-535
            stagsteps += 1
        else:
            stagsteps = 0
        xk = xk + alpha * ph
        sk = res - alpha * v
        normr = nrm(sk)
        abs_resid = normr
        if normr <= tolb or stagsteps >= maxstagsteps or moresteps:     # :546-570
            sk = b - A(xk)
            abs_resid = nrm(sk)
            if abs_resid <= tolb:
                flag, it = 0, ii + 1 - 0.5
                break
            if stagsteps >= maxstagsteps and moresteps == 0:
                stagsteps = 0
            moresteps += 1
            if moresteps >= maxmsteps:
                if ref_exit:
                    b = xk.copy()                                      # :563 b->copyFrom(*xk_)
                flag, it = 3, ii + 1 - 0.5
                break
        if stagsteps >= maxstagsteps:
            flag, it = 3, ii + 1 - 0.5
            break
        if abs_resid < normrmin:
            normrmin = abs_resid
            xmin = xk.copy()
            imin = ii + 1 - 0.5
        ph = ML(sk) if ML is not None else sk.copy()
        t = A(ph)
        tt = dot(t, t)
        if tt == 0.0 or abs(tt) > 1e20:
            flag, it = 4, ii + 1
            break
        omega = dot(t, sk) / tt
        if abs(omega) > 1e20:
            flag, it = 4, ii + 1
            break
        if nrm(ph) * abs(omega) < eps * nrm(xk):
            stagsteps += 1
        else:
            stagsteps = 0
        xk = xk + omega * ph
        res = sk - omega * t
        normr = nrm(res)
        abs_resid = normr
        if normr <= tolb or stagsteps >= maxstagsteps or moresteps:     # :623-648
            res = b - A(xk)
            abs_resid = nrm(res)
            if abs_resid <= tolb:
                flag, it = 0, ii + 1
                break
            if stagsteps >= maxstagsteps and moresteps == 0:
                stagsteps = 0
            moresteps += 1
            if moresteps >= maxmsteps:
                if ref_exit:
                    b = xk.copy()                                      # :641
                flag, it = 3, ii + 1
                break
        if abs_resid < normrmin:
            normrmin = abs_resid
            xmin = xk.copy()
            imin = ii + 1
        if stagsteps >= maxstagsteps:
            flag, it = 3, ii + 1 - 0.5
            break
        ii += 1
    if flag == 0:                                                  # :665-669
        return xk, True, 0, it, abs_resid, abs_resid / n2b
    res = b - A(xmin)                                              # :671-688
    normr_comp = nrm(res)
    if normr_comp <= abs_resid:
        return xmin, False, flag, imin + 1, normr_comp, normr_comp / n2b
    return xk, False, flag, ii + 1, abs_resid, abs_resid / n2b


# ---------------------------------------------------------------------------------------------------------
# the full-space layer
# ---------------------------------------------------------------------------------------------------------
class KKTLinSysFull:
    """update / factorize / computeDirections / compute_directions_w_IR of hiopKKTLinSysCompressedXYcYd on top of
    a provider.  Iterates and residuals are dicts keyed by ITER_PARTS / RESID_PARTS."""

    def __init__(self, prov, ixl, ixu, idl, idu, perturb=None, n_required_neg_eig=None, inertia_free=False):
        self.p = prov
        self.inertia_free = inertia_free       # hiopFactAcceptorInertiaFreeDWD instead of hiopFactAcceptorIC
        self.ixl, self.ixu, self.idl, self.idu = ixl, ixu, idl, idu
        self.perturb = perturb if perturb is not None else PDPerturbationPrimalFirstScalar()
        self.n_req = prov.nyc + prov.nyd if n_required_neg_eig is None else n_required_neg_eig   # hiopAlgFilterIPM.cpp:2096
        self.sizes = part_sizes(prov.nx, prov.nd, prov.nyc, prov.nyd)
        self.num_refact = 0

    # -- :543-583
    def update(self, it):
        self.it = it
        Dx = np.zeros(self.p.nx)
        ho.axdzpy_w_pattern(Dx, 1.0, it["zl"], it["sxl"], self.ixl)
        ho.axdzpy_w_pattern(Dx, 1.0, it["zu"], it["sxu"], self.ixu)
        Dd = np.zeros(self.p.nd)
        ho.axdzpy_w_pattern(Dd, 1.0, it["vl"], it["sdl"], self.idl)
        ho.axdzpy_w_pattern(Dd, 1.0, it["vu"], it["sdu"], self.idu)
        self.Dx, self.Dd = Dx, Dd
        self.p.set_diagonals(Dx, Dd)
        return self.factorize()

    # -- :316-376
    def factorize(self):
        max_refact, self.num_refact = 10, 0
        if not self.perturb.compute_initial_deltas():
            return False
        while self.num_refact <= max_refact:
            self.p.build(*self.perturb.deltas())
            n_neg = self.p.factorize()
            cont = self._accept(n_neg)
            if cont == -1:
                return False
            if cont == 0:
                break
            self.num_refact += 1
        return self.num_refact <= max_refact

    def _accept(self, n_neg, force_reg=False):
        if self.inertia_free:
            return require_refactorization_inertia_free(self.perturb, self.n_req, n_neg, force_reg)
        return require_refactorization(self.perturb, self.n_req, n_neg)

    # -- :376-448
    def factorize_inertia_free(self):
        self._accept(1, True)                                       # :388 (return value unused)
        self.p.build(*self.perturb.deltas())
        solver_flag = self.p.factorize()
        max_refact, self.num_refact = 10, 0
        while self.num_refact <= max_refact and solver_flag < 0:
            if self._accept(solver_flag) == -1:
                return False
            self.p.build(*self.perturb.deltas())
            solver_flag = self.p.factorize()
            self.num_refact += 1
        return True

    # -- :455-513
    def test_direction(self, d, neg_curv_test_fact=1e-11, dot=None):
        dot = dot or (lambda u, v: float(u @ v))
        dwx, dwd, _, _ = self.perturb.deltas()
        x, dd = d["x"], d["d"]
        dWd = dot(self.p.hess_times_vec(x), x)
        dWd += dot(x * self.Dx + dwx * x, x)
        dWd += float((dd * self.Dd + dwd * dd) @ dd)
        xs_nrmsq = dot(x, x) + float(dd @ dd)
        self.last_dWd, self.last_xs_nrmsq = dWd, xs_nrmsq
        return not (dWd < xs_nrmsq * neg_curv_test_fact)

    # -- :585-690
    def compute_directions(self, r):
        it = self.it
        rx_tilde = r["rx"].copy()
        rl = r["rszl"] - it["zl"] * r["rxl"]
        ho.axdzpy_w_pattern(rx_tilde, 1.0, rl, it["sxl"], self.ixl)
        ru = r["rszu"] - it["zu"] * r["rxu"]
        ho.axdzpy_w_pattern(rx_tilde, -1.0, ru, it["sxu"], self.ixu)
        ryd2 = r["rd"].copy()
        rd2 = r["rsvl"] - it["vl"] * r["rdl"]
        ho.axdzpy_w_pattern(ryd2, 1.0, rd2, it["sdl"], self.idl)
        rd2 = r["rsvu"] - it["vu"] * r["rdu"]
        ho.axdzpy_w_pattern(ryd2, -1.0, rd2, it["sdu"], self.idu)
        if getattr(self.p, "xd_form", False):                       # XDYcYd::computeDirections (:810-905)
            ok, dx, dd, dyc, dyd = self.p.solve_xd(rx_tilde, ryd2, r["ryc"], r["ryd"])
            d = {"x": dx, "d": dd, "yc": dyc, "yd": dyd}
            if not ok:
                return False, d
            self.compute_directions_for_full_space(r, d)
            return True, d
        ryd_tilde = r["ryd"] + ryd2 * self.p.Dd_inv
        ok, dx, dyc, dyd = self.p.solve(rx_tilde, r["ryc"], ryd_tilde)
        d = {"x": dx, "yc": dyc, "yd": dyd}
        d["d"] = (ryd2 + dyd) * self.p.Dd_inv                       # :664-666
        if not ok:
            return False, d
        self.compute_directions_for_full_space(r, d)
        return True, d

    # -- :218-314
    def compute_directions_for_full_space(self, r, d):
        it = self.it
        d["sxl"] = (r["rxl"] + d["x"]) * self.ixl
        d["zl"] = _div_w_select(r["rszl"] - it["zl"] * d["sxl"], it["sxl"], self.ixl)
        d["sxu"] = (r["rxu"] - d["x"]) * self.ixu
        d["zu"] = _div_w_select((r["rszu"] - it["zu"] * d["sxu"]) * self.ixu, it["sxu"], self.ixu)
        d["sdl"] = (r["rdl"] + d["d"]) * self.idl
        d["vl"] = _div_w_select((r["rsvl"] - it["vl"] * d["sdl"]) * self.idl, it["sdl"], self.idl)
        d["sdu"] = (r["rdu"] - d["d"]) * self.idu
        d["vu"] = _div_w_select((r["rsvu"] - it["vu"] * d["sdu"]) * self.idu, it["sdu"], self.idu)

    # -- :1619-1736  y = KKT_full * x  (x keyed by ITER_PARTS, y keyed by RESID_PARTS)
    def times_vec(self, x):
        it, p = self.it, self.p
        dwx, dwd, dcc, dcd = self.perturb.deltas()
        y = {}
        y["rx"] = (p.hess_times_vec(x["x"]) + dwx * x["x"] + p.jac_trans_times_vec("c", x["yc"]) +
                   p.jac_trans_times_vec("d", x["yd"]) - x["zl"] + x["zu"])
        y["rd"] = -x["yd"] - x["vl"] + x["vu"] + dwd * x["d"]
        y["ryc"] = p.jac_times_vec("c", x["x"]) - dcc * x["yc"]
        y["ryd"] = p.jac_times_vec("d", x["x"]) - x["d"] - dcd * x["yd"]
        y["rxl"] = (x["sxl"] - x["x"]) * self.ixl
        y["rxu"] = (x["sxu"] + x["x"]) * self.ixu
        y["rdl"] = (x["sdl"] - x["d"]) * self.idl
        y["rdu"] = (x["sdu"] + x["d"]) * self.idu
        y["rszl"] = it["sxl"] * x["zl"] + it["zl"] * x["sxl"]
        y["rszu"] = it["sxu"] * x["zu"] + it["zu"] * x["sxu"]
        y["rsvl"] = it["sdl"] * x["vl"] + it["vl"] * x["sdl"]
        y["rsvu"] = it["sdu"] * x["vu"] + it["vu"] * x["sdu"]
        return y

    def times_vec_flat(self, xs):
        return pack(self.times_vec(unpack(xs, ITER_PARTS, self.sizes)), RESID_PARTS)

    def precond_flat(self, rs):                                     # :1900-1909
        ok, d = self.compute_directions(unpack(rs, RESID_PARTS, self.sizes))
        return pack(d, ITER_PARTS)

    # -- :911-961
    def compute_directions_w_IR(self, r, mu, ir_outer_tol_factor=1e-2, ir_outer_tol_min=1e-6, ir_outer_maxit=8,
                                dot=None):
        if ir_outer_maxit <= 0:
            return self.compute_directions(r) + (None,)
        tol = min(mu * ir_outer_tol_factor, ir_outer_tol_min)
        b = pack(r, RESID_PARTS)
        x, conv, flag, it, absr, relr = bicgstab(self.times_vec_flat, self.precond_flat, b, tol, ir_outer_maxit, dot)
        info = {"converged": conv, "flag": flag, "iter": it, "abs_resid": absr, "rel_resid": relr}
        return True, unpack(x, ITER_PARTS, self.sizes), info          # accepted even if not converged (:949-953)


def sharded_dot(sizes, allreduce):
    """Dot product of two local slabs on a column partition: the x-sized parts (x, sxl, sxu, zl, zu) are slices and
    their partial sums are all-reduced, the other parts are replicated (hiopVectorCompoundPD::dotProductWith sums the
    parts' own dotProductWith, each with its own communicator semantics)."""
    offs = np.concatenate([[0], np.cumsum(sizes)])
    dist_parts = (0, 4, 5, 8, 9)

    def dot(u, v):
        d = sum(float(u[offs[p]:offs[p + 1]] @ v[offs[p]:offs[p + 1]]) for p in dist_parts)
        rpl = sum(float(u[offs[p]:offs[p + 1]] @ v[offs[p]:offs[p + 1]]) for p in range(12) if p not in dist_parts)
        return float(allreduce(np.array([d]))[0]) + rpl
    return dot


def _div_w_select(num, den, pattern):                                  # hiopVectorPar.cpp:580
    out = np.zeros_like(num)
    m = pattern != 0.0
    out[m] = num[m] / den[m]
    return out
