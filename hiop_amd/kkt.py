"""Host-side handles of the stateful KKT objects of the C ABI.

`KKTLinSysCompressedMDSXYcYd` mirrors the reference class of the same name
(src/Optimization/hiopKKTLinSysMDS.hpp:97; update/build_kkt_matrix/factorizeWithCurvCheck/solveCompressed)
and `LinSolverSymDense` mirrors hiopLinSolverSymDense (src/LinAlg/hiopLinSolver.hpp:117:
sysMatrix / matrixChanged / solve).  All heavy lifting happens in libhiopamd.so on the device."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import MdsStructure, check, lib
from .runtime import Context, dev, dptr


class LinSolverSymDense:
    def __init__(self, ctx: Context, n: int):
        self.ctx, self.n = ctx, n
        self._L = lib()
        h = C.c_void_p()
        check(self._L.hiopamd_linsolver_create(C.byref(h), ctx.h, n), "hiopamd_linsolver_create")
        self.h = h
        ctx._register(self)

    def sys_matrix_ptr(self) -> int:
        return self._L.hiopamd_linsolver_sys_matrix(self.h)

    def set_sys_matrix(self, M: torch.Tensor):
        """Copy an n x n row-major device matrix (upper triangle significant) into sysMatrix()."""
        assert M.shape == (self.n, self.n) and M.dtype == torch.float64 and M.is_cuda
        torch.cuda.synchronize()
        check(self._L.hiopamd_copy_d2d(self.ctx.h, C.c_void_p(self.sys_matrix_ptr()), dptr(M.contiguous(), self.ctx),
                                       self.n * self.n * 8), "copy_d2d")
        self._assembled_from = M          # (a reference, not a copy: see matrix_changed)

    def get_sys_matrix(self) -> torch.Tensor:
        out = torch.empty((self.n, self.n), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        check(self._L.hiopamd_copy_d2d(self.ctx.h, dptr(out, self.ctx), C.c_void_p(self.sys_matrix_ptr()), self.n * self.n * 8),
              "copy_d2d")
        self.ctx.sync()
        return out

    # HIOPAMD_ERR_TIMEOUT (-6) only reaches a caller that switched the object's retry copy off (set_retry_copy(False): "I re-assemble and
    # call again", what the native KKT objects do).  With the copy off this wrapper is that caller when the matrix came through
    # set_sys_matrix: it copies the source again and calls once more (the library runs that call with the stepwise kernels).
    retry_after_timeout = True

    def set_retry_copy(self, enable: bool):
        check(self._L.hiopamd_linsolver_set_retry_copy(self.h, 1 if enable else 0), "hiopamd_linsolver_set_retry_copy")

    def timeouts(self) -> int:
        """bounded waits of the dataflow factorisation that expired over this object's life"""
        n = C.c_int64(0)
        check(self._L.hiopamd_linsolver_timeouts(self.h, C.byref(n)), "hiopamd_linsolver_timeouts")
        return n.value

    def matrix_changed(self) -> int:
        nneg = C.c_int(0)
        rc = self._L.hiopamd_linsolver_matrix_changed(self.h, C.byref(nneg))
        src = getattr(self, "_assembled_from", None)
        if rc == -6 and self.retry_after_timeout and src is not None:
            self.set_sys_matrix(src)
            self.ctx.sync()
            rc = self._L.hiopamd_linsolver_matrix_changed(self.h, C.byref(nneg))
        check(rc, "hiopamd_linsolver_matrix_changed")
        return nneg.value

    def solve(self, rhs: torch.Tensor, nrhs: int = 1) -> bool:
        rc = self._L.hiopamd_linsolver_solve(self.h, dptr(rhs, self.ctx), nrhs)
        check(rc, "hiopamd_linsolver_solve")
        return True

    def set_pivoting(self, enable: bool):
        """Bunch-Kaufman mode (the reference's safe solver, csrc/ldlt_bk.hip): exact inertia for any symmetric matrix"""
        check(self._L.hiopamd_linsolver_set_pivoting(self.h, 1 if enable else 0), "hiopamd_linsolver_set_pivoting")

    def set_safe_mode(self, enable: bool, n_pos_block: int):
        """static quasi-definite regularisation + iterative refinement (the reference's safe mode = Bunch-Kaufman solver)"""
        check(self._L.hiopamd_linsolver_set_safe_mode(self.h, 1 if enable else 0, n_pos_block), "hiopamd_linsolver_set_safe_mode")

    def growth(self):
        """(max |u_ij|, min |d_i|, max |d_i|) of the last factorisation"""
        u, dmin, dmax = C.c_double(0), C.c_double(0), C.c_double(0)
        check(self._L.hiopamd_linsolver_growth(self.h, C.byref(u), C.byref(dmin), C.byref(dmax)), "hiopamd_linsolver_growth")
        return u.value, dmin.value, dmax.value

    def solve_status(self) -> bool:
        ok = C.c_int(1)
        check(self._L.hiopamd_linsolver_solve_status(self.h, C.byref(ok)), "hiopamd_linsolver_solve_status")
        return bool(ok.value)

    def safe_mode_info(self):
        it, res = C.c_int(0), C.c_double(0)
        check(self._L.hiopamd_linsolver_safe_mode_info(self.h, C.byref(it), C.byref(res)), "hiopamd_linsolver_safe_mode_info")
        return it.value, res.value

    def set_dataflow(self, enable: bool):
        """dataflow factorisation (two persistent kernels) on / off (off = the stepwise kernels)."""
        check(self._L.hiopamd_linsolver_set_dataflow(self.h, 1 if enable else 0), "hiopamd_linsolver_set_dataflow")

    def inertia(self):
        p, n, z = C.c_int(), C.c_int(), C.c_int()
        check(self._L.hiopamd_linsolver_inertia(self.h, C.byref(p), C.byref(n), C.byref(z)), "inertia")
        return p.value, n.value, z.value

    def close(self):
        if self.h is not None:
            if self.ctx.h is not None:
                self._L.hiopamd_linsolver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KKTLinSysCompressedMDSXYcYd:
    """Device-resident condensed MDS KKT.  `prob` carries the (host, numpy) structure + constant blocks;
    they are uploaded once and stay in HBM."""

    def __init__(self, ctx: Context, nxs, nxd, neq, nineq, Jcs_ij, Jds_ij, Hss_ij):
        self.ctx = ctx
        self._L = lib()
        self.nxs, self.nxd, self.neq, self.nineq = nxs, nxd, neq, nineq
        self.N = nxd + neq + nineq
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self._host = [i32(Jcs_ij[0]), i32(Jcs_ij[1]), i32(Jds_ij[0]), i32(Jds_ij[1]), i32(Hss_ij[0]), i32(Hss_ij[1])]
        self._devi = [dev(a, torch.int32) for a in self._host]
        torch.cuda.synchronize()
        s = MdsStructure()
        s.nxs, s.nxd, s.neq, s.nineq = nxs, nxd, neq, nineq
        s.nnz_Jcs = self._host[0].size
        s.Jcs_i, s.Jcs_j = self._devi[0].data_ptr(), self._devi[1].data_ptr()
        s.Jcs_i_host, s.Jcs_j_host = self._host[0].ctypes.data, self._host[1].ctypes.data
        s.nnz_Jds = self._host[2].size
        s.Jds_i, s.Jds_j = self._devi[2].data_ptr(), self._devi[3].data_ptr()
        s.Jds_i_host, s.Jds_j_host = self._host[2].ctypes.data, self._host[3].ctypes.data
        s.nnz_Hss = self._host[4].size
        s.Hss_i, s.Hss_j = self._devi[4].data_ptr(), self._devi[5].data_ptr()
        h = C.c_void_p()
        check(self._L.hiopamd_kkt_mds_create(C.byref(h), ctx.h, C.byref(s)), "hiopamd_kkt_mds_create")
        self.h = h
        self._vals = None
        ctx._register(self)

    def set_values(self, Jcs_val, Jds_val, Hss_val, Jcd, Jdd, Hdd, Dx, Dd):
        """All arguments are device fp64 tensors; they are borrowed (kept alive here) until the next call."""
        self._vals = [Jcs_val, Jds_val, Hss_val, Jcd, Jdd, Hdd, Dx, Dd]
        torch.cuda.synchronize()
        check(self._L.hiopamd_kkt_mds_set_values(self.h, *[dptr(t, self.ctx) for t in self._vals]), "hiopamd_kkt_mds_set_values")

    def build_kkt_matrix(self, delta_wx=0.0, delta_wd=0.0, delta_cc=0.0, delta_cd=0.0):
        """Scalars -> hiopamd_kkt_mds_build; device tensors (or None = zero vector) -> hiopamd_kkt_mds_build_vec."""
        ds = (delta_wx, delta_wd, delta_cc, delta_cd)
        if any(isinstance(d, torch.Tensor) or d is None for d in ds):
            self._deltas = [d for d in ds]   # borrowed by the launch
            ptrs = [dptr(d, self.ctx) if isinstance(d, torch.Tensor) else None for d in ds]
            assert all(isinstance(d, torch.Tensor) or d is None for d in ds), "mix of scalars and vectors"
            check(self._L.hiopamd_kkt_mds_build_vec(self.h, *ptrs), "hiopamd_kkt_mds_build_vec")
            return
        check(self._L.hiopamd_kkt_mds_build(self.h, delta_wx, delta_wd, delta_cc, delta_cd), "hiopamd_kkt_mds_build")

    def factorize_with_curv_check(self) -> int:
        nneg = C.c_int(0)
        check(self._L.hiopamd_kkt_mds_factorize(self.h, C.byref(nneg)), "hiopamd_kkt_mds_factorize")
        return nneg.value

    def solve_compressed(self, rx, ryc, ryd, dx, dyc, dyd):
        check(self._L.hiopamd_kkt_mds_solve_compressed(self.h, dptr(rx, self.ctx), dptr(ryc, self.ctx), dptr(ryd, self.ctx), dptr(dx, self.ctx), dptr(dyc, self.ctx),
                                                       dptr(dyd, self.ctx)), "hiopamd_kkt_mds_solve_compressed")

    def sys_matrix(self) -> torch.Tensor:
        """Copy of the N x N system matrix (device tensor)."""
        ptr = self._L.hiopamd_kkt_mds_sys_matrix(self.h)
        out = torch.empty((self.N, self.N), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        check(self._L.hiopamd_copy_d2d(self.ctx.h, dptr(out, self.ctx), C.c_void_p(ptr), self.N * self.N * 8), "copy_d2d")
        self.ctx.sync()
        return out

    def Hxs(self) -> torch.Tensor:
        ptr = self._L.hiopamd_kkt_mds_Hxs(self.h)
        out = torch.empty(self.nxs, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        check(self._L.hiopamd_copy_d2d(self.ctx.h, dptr(out, self.ctx), C.c_void_p(ptr), self.nxs * 8), "copy_d2d")
        self.ctx.sync()
        return out

    def close(self):
        if self.h is not None:
            if self.ctx.h is not None:
                self._L.hiopamd_kkt_mds_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mds_from_problem(ctx: Context, p) -> tuple["KKTLinSysCompressedMDSXYcYd", dict]:
    """Upload an MDS problem description (numpy arrays: sizes, COO index/value arrays, dense blocks) to the device; returns the KKT handle and
    the dict of device tensors holding the constant blocks."""
    k = KKTLinSysCompressedMDSXYcYd(ctx, p.nxs, p.nxd, p.neq, p.nineq, (p.Jcs_i, p.Jcs_j), (p.Jds_i, p.Jds_j),
                                    (p.Hss_i, p.Hss_j))
    d = dict(Jcs_v=dev(p.Jcs_v), Jds_v=dev(p.Jds_v), Hss_v=dev(p.Hss_v), Jcd=dev(p.Jcd), Jdd=dev(p.Jdd), Hdd=dev(p.Hdd))
    return k, d


_SIGMA = {"sty": 1, "sty_inv": 2, "snrm_ynrm": 3, "sty_srnm_ynrm": 4, "sigma0": 5}


class HessianLowRank:
    """Device-resident compact L-BFGS Hessian (mirrors hiopHessianLowRank, src/Optimization/hiopHessianLowRank.hpp)."""

    def __init__(self, ctx: Context, n_local: int, m_eq: int, m_ineq: int, l_max: int = 6, sigma0: float = 1.0,
                 sigma_update_strategy: str = "sigma0"):
        self.ctx, self.n, self.m_eq, self.m_ineq, self.l_max = ctx, n_local, m_eq, m_ineq, l_max
        self._L = lib()
        h = C.c_void_p()
        check(self._L.hiopamd_hess_lowrank_create(C.byref(h), ctx.h, n_local, m_eq, m_ineq, l_max, sigma0,
                                                  _SIGMA[sigma_update_strategy]), "hiopamd_hess_lowrank_create")
        self.h = h
        ctx._register(self)

    def update(self, x, grad_f, Jc, Jd, yc, yd) -> bool:
        stored = C.c_int(0)
        check(self._L.hiopamd_hess_lowrank_update(self.h, dptr(x, self.ctx), dptr(grad_f, self.ctx), dptr(Jc, self.ctx), dptr(Jd, self.ctx), dptr(yc, self.ctx), dptr(yd, self.ctx),
                                                  C.byref(stored)), "hiopamd_hess_lowrank_update")
        return bool(stored.value)

    def update_log_barrier_diagonal(self, Dx):
        check(self._L.hiopamd_hess_lowrank_update_log_barrier_diagonal(self.h, dptr(Dx, self.ctx)), "update_log_barrier_diagonal")

    def solve(self, rhs, x):
        check(self._L.hiopamd_hess_lowrank_solve(self.h, dptr(rhs, self.ctx), dptr(x, self.ctx)), "hiopamd_hess_lowrank_solve")

    def sym_mat_times_inverse_times_mat_trans(self, beta, W, alpha, X):
        k = W.shape[0]
        work = torch.empty(k * (k + 2 * self.l_max) + 2 * k * self.l_max + 8, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        check(self._L.hiopamd_hess_lowrank_sym_mat_times_inverse_times_mat_trans(self.h, beta, dptr(W, self.ctx), k, alpha, dptr(X, self.ctx),
                                                                                 dptr(work, self.ctx)), "symMatTimesInverseTimesMatTrans")
        self.ctx.sync()

    def times_vec(self, beta, y, alpha, x, add_log_term=True):
        check(self._L.hiopamd_hess_lowrank_times_vec(self.h, beta, dptr(y, self.ctx), alpha, dptr(x, self.ctx), 1 if add_log_term else 0),
              "hiopamd_hess_lowrank_times_vec")

    @property
    def l_curr(self) -> int:
        return self._L.hiopamd_hess_lowrank_l_curr(self.h)

    @property
    def sigma(self) -> float:
        return self._L.hiopamd_hess_lowrank_sigma(self.h)

    def _rows(self, ptr):
        l = self.l_curr
        out = torch.empty((l, self.n), dtype=torch.float64, device="cuda")
        if l:
            torch.cuda.synchronize()
            check(self._L.hiopamd_copy_d2d(self.ctx.h, dptr(out, self.ctx), C.c_void_p(ptr), l * self.n * 8), "copy_d2d")
            self.ctx.sync()
        return out

    def St(self):
        return self._rows(self._L.hiopamd_hess_lowrank_St(self.h))

    def Yt(self):
        return self._rows(self._L.hiopamd_hess_lowrank_Yt(self.h))

    def close(self):
        if self.h is not None:
            if self.ctx.h is not None:
                self._L.hiopamd_hess_lowrank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KKTLinSysLowRank:
    """Mirrors hiopKKTLinSysLowRank (src/Optimization/hiopKKTLinSys.hpp:385): update / solveCompressed."""

    def __init__(self, ctx: Context, hess: HessianLowRank):
        self.ctx, self.hess = ctx, hess
        self._L = lib()
        h = C.c_void_p()
        check(self._L.hiopamd_kkt_lowrank_create(C.byref(h), ctx.h, hess.h), "hiopamd_kkt_lowrank_create")
        self.h = h
        self.k = hess.m_eq + hess.m_ineq
        ctx._register(self)

    def update(self, zl, sxl, ixl, zu, sxu, ixu, vl, sdl, idl, vu, sdu, idu, Jc, Jd):
        args = [zl, sxl, ixl, zu, sxu, ixu, vl, sdl, idl, vu, sdu, idu, Jc, Jd]
        self._jac = (Jc, Jd)      # the C object keeps these two pointers until the next update / set_jacobians
        check(self._L.hiopamd_kkt_lowrank_update(self.h, *[dptr(a, self.ctx) for a in args]), "hiopamd_kkt_lowrank_update")

    def update_diag(self, Dx, Dd, Jc, Jd):
        self._jac = (Jc, Jd)
        check(self._L.hiopamd_kkt_lowrank_update_diag(self.h, dptr(Dx, self.ctx), dptr(Dd, self.ctx), dptr(Jc, self.ctx), dptr(Jd, self.ctx)),
              "hiopamd_kkt_lowrank_update_diag")

    def solve_compressed(self, rx, ryc, ryd, dx, dyc, dyd) -> bool:
        ok = C.c_int(0)
        check(self._L.hiopamd_kkt_lowrank_solve_compressed(self.h, dptr(rx, self.ctx), dptr(ryc, self.ctx), dptr(ryd, self.ctx), dptr(dx, self.ctx), dptr(dyc, self.ctx),
                                                           dptr(dyd, self.ctx), C.byref(ok)), "hiopamd_kkt_lowrank_solve_compressed")
        return bool(ok.value)

    def set_cache(self, enable: bool):
        check(self._L.hiopamd_kkt_lowrank_set_cache(self.h, 1 if enable else 0), "hiopamd_kkt_lowrank_set_cache")

    def set_jacobians(self, Jc, Jd):
        self._jac = (Jc, Jd)
        check(self._L.hiopamd_kkt_lowrank_set_jacobians(self.h, dptr(Jc, self.ctx), dptr(Jd, self.ctx)), "hiopamd_kkt_lowrank_set_jacobians")

    def N(self) -> torch.Tensor:
        ptr = self._L.hiopamd_kkt_lowrank_N(self.h)
        out = torch.empty((self.k, self.k), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        check(self._L.hiopamd_copy_d2d(self.ctx.h, dptr(out, self.ctx), C.c_void_p(ptr), self.k * self.k * 8), "copy_d2d")
        self.ctx.sync()
        return out

    def close(self):
        if self.h is not None:
            if self.ctx.h is not None:
                self._L.hiopamd_kkt_lowrank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


ITER_PARTS = ("x", "d", "yc", "yd", "sxl", "sxu", "sdl", "sdu", "zl", "zu", "vl", "vu")
RESID_PARTS = ("rx", "rd", "ryc", "ryd", "rxl", "rxu", "rdl", "rdu", "rszl", "rszu", "rsvl", "rsvu")


class KKTLinSysSparseCondensed:
    """Mirrors hiopKKTLinSysCondensedSparse (src/Optimization/hiopKKTLinSysSparseCondensed.hpp:78): build_kkt_matrix /
    solveCompressed of the inequality-only sparse formulation; condensed matrix in CSR on the device, PCG + Jacobi inside."""

    def __init__(self, ctx: Context, nx, nineq, Jd_i, Jd_j, H_i, H_j):
        self.ctx, self.nx, self.nineq = ctx, nx, nineq
        self._L = lib()
        self._idx = [np.ascontiguousarray(a, dtype=np.int32) for a in (Jd_i, Jd_j, H_i, H_j)]
        h = C.c_void_p()
        check(self._L.hiopamd_kkt_sparse_condensed_create(C.byref(h), ctx.h, nx, nineq, self._idx[0].size, self._idx[0].ctypes.data,
                                                          self._idx[1].ctypes.data, self._idx[2].size, self._idx[2].ctypes.data,
                                                          self._idx[3].ctypes.data), "hiopamd_kkt_sparse_condensed_create")
        self.h = h
        self._vals = None
        ctx._register(self)

    def set_values(self, Jd_val, H_val, Dx, Dd):
        self._vals = [Jd_val, H_val, Dx, Dd]
        torch.cuda.synchronize()
        check(self._L.hiopamd_kkt_sparse_condensed_set_values(self.h, *[dptr(t, self.ctx) for t in self._vals]), "set_values")

    def build_kkt_matrix(self, delta_wx=0.0, delta_wd=0.0):
        if isinstance(delta_wx, torch.Tensor) or isinstance(delta_wd, torch.Tensor) or delta_wx is None or delta_wd is None:
            self._deltas = [delta_wx, delta_wd]
            check(self._L.hiopamd_kkt_sparse_condensed_build_vec(self.h, dptr(delta_wx, self.ctx), dptr(delta_wd, self.ctx)), "build_vec")
        else:
            check(self._L.hiopamd_kkt_sparse_condensed_build(self.h, C.c_double(delta_wx), C.c_double(delta_wd)), "build")

    def factorize(self) -> int:
        n = C.c_int(0)
        check(self._L.hiopamd_kkt_sparse_condensed_factorize(self.h, C.byref(n)), "factorize")
        return n.value

    def solve_compressed(self, rx, rd, ryd, dx, dd, dyd) -> bool:
        ok = C.c_int(0)
        check(self._L.hiopamd_kkt_sparse_condensed_solve_compressed(self.h, dptr(rx, self.ctx), dptr(rd, self.ctx), dptr(ryd, self.ctx),
                                                                    dptr(dx, self.ctx), dptr(dd, self.ctx), dptr(dyd, self.ctx),
                                                                    C.byref(ok)), "solve_compressed")
        return bool(ok.value)

    def inner_kind(self) -> str:
        """'dense' (LDL^T of the expanded matrix), 'bordered' (bordered-diagonal direct solver), 'pcg' or 'sparse_ldl' (nested
        dissection + multifrontal LDL^T + dense root)"""
        return ("dense", "bordered", "pcg", "sparse_ldl")[self.ctx._L.hiopamd_kkt_sparse_condensed_inner_kind(self.h)]

    def set_inner_solver(self, tol, max_iter):
        check(self._L.hiopamd_kkt_sparse_condensed_set_inner_solver(self.h, C.c_double(tol), int(max_iter)), "set_inner_solver")

    def last_solve(self):
        f, it, rel = C.c_int(0), C.c_double(0), C.c_double(0)
        check(self._L.hiopamd_kkt_sparse_condensed_last_solve(self.h, C.byref(f), C.byref(it), C.byref(rel)), "last_solve")
        return f.value, it.value, rel.value

    def close(self):
        if self.h is not None:
            if self.ctx.h is not None:
                self._L.hiopamd_kkt_sparse_condensed_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KKTLinSysXYcYd:
    """Full-space layer (mirrors hiopKKTLinSysCompressedXYcYd + hiopKKTLinSysCurvCheck::factorize +
    compute_directions_w_IR, src/Optimization/hiopKKTLinSys.hpp:226-330): update / computeDirections /
    compute_directions_w_IR on 12-part device slabs.  Built on an MDS, dense or low-rank condensed solver."""

    def __init__(self, ctx: Context, backend, ixl, ixu, idl, idu, dense_dims=None, xd_form=False):
        self.ctx, self.backend = ctx, backend
        self._L = lib()
        self._pat = [ixl, ixu, idl, idu]          # borrowed by the C object: keep alive
        torch.cuda.synchronize()
        h = C.c_void_p()
        p = [dptr(t, self.ctx) for t in self._pat]
        if isinstance(backend, KKTLinSysCompressedMDSXYcYd):
            check(self._L.hiopamd_kkt_xycyd_create_mds(C.byref(h), ctx.h, backend.h, *p), "hiopamd_kkt_xycyd_create_mds")
        elif isinstance(backend, KKTLinSysLowRank):
            check(self._L.hiopamd_kkt_xycyd_create_lowrank(C.byref(h), ctx.h, backend.h, *p),
                  "hiopamd_kkt_xycyd_create_lowrank")
        elif isinstance(backend, KKTLinSysSparseCondensed):
            check(self._L.hiopamd_kkt_xycyd_create_sparse_condensed(C.byref(h), ctx.h, backend.h, *p),
                  "hiopamd_kkt_xycyd_create_sparse_condensed")
        else:
            nx, neq, nineq = dense_dims
            create = self._L.hiopamd_kkt_xycyd_create_dense_xdycyd if xd_form else self._L.hiopamd_kkt_xycyd_create_dense
            check(create(C.byref(h), ctx.h, nx, neq, nineq, *p), "hiopamd_kkt_xycyd_create_dense")
        self.h = h
        self.dim = self._L.hiopamd_kkt_xycyd_dim(h)
        off = (C.c_int64 * 13)()
        check(self._L.hiopamd_kkt_xycyd_offsets(h, off), "offsets")
        self.off = list(off)
        self._keep = []
        ctx._register(self)

    # -- slab helpers (host numpy dict <-> one device tensor)
    def pack(self, parts: dict, names) -> torch.Tensor:
        return dev(np.concatenate([np.asarray(parts[k], dtype=np.float64) for k in names]))

    def unpack(self, slab: torch.Tensor, names) -> dict:
        a = slab.cpu().numpy()
        return {k: a[self.off[i]:self.off[i + 1]].copy() for i, k in enumerate(names)}

    def set_matrices(self, H, Jc, Jd):
        self._keep = [H, Jc, Jd]
        torch.cuda.synchronize()
        check(self._L.hiopamd_kkt_xycyd_set_matrices(self.h, dptr(H, self.ctx), dptr(Jc, self.ctx), dptr(Jd, self.ctx)), "set_matrices")

    def set_mu(self, mu: float):
        check(self._L.hiopamd_kkt_xycyd_set_mu(self.h, mu), "set_mu")

    def set_regularization(self, dual_first=False, randomized=False, seed=0x9E3779B97F4A7C15):
        """hiopPDPerturbation{PrimalFirst,DualFirst}{Scalar,Rand} (hiopAlgFilterIPM.cpp:2165-2177)."""
        check(self._L.hiopamd_kkt_xycyd_set_regularization(self.h, int(dual_first), int(randomized), C.c_uint64(seed)), "set_regularization")

    def delta_vectors(self):
        """Copies of the current regularisation vectors (randomized mode): delta_wx, delta_wd, delta_cc, delta_cd."""
        ptrs = [C.c_void_p() for _ in range(4)]
        check(self._L.hiopamd_kkt_xycyd_delta_vectors(self.h, *[C.byref(p) for p in ptrs]), "delta_vectors")
        sizes = [self.off[1] - self.off[0], self.off[2] - self.off[1], self.off[3] - self.off[2], self.off[4] - self.off[3]]
        out = []
        for p, n in zip(ptrs, sizes):
            t = torch.empty(n, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            if n:
                check(self._L.hiopamd_copy_d2d(self.ctx.h, dptr(t, self.ctx), p, n * 8), "copy_d2d")
            out.append(t)
        self.ctx.sync()
        return out

    def set_perturbation_options(self, opts8):
        arr = (C.c_double * 8)(*opts8)
        check(self._L.hiopamd_kkt_xycyd_set_perturbation_options(self.h, arr), "set_perturbation_options")

    def update(self, it: torch.Tensor) -> bool:
        self._iter = it
        ok = C.c_int(0)
        check(self._L.hiopamd_kkt_xycyd_update(self.h, dptr(it, self.ctx), C.byref(ok)), "hiopamd_kkt_xycyd_update")
        return bool(ok.value)

    def factorize(self) -> bool:
        ok = C.c_int(0)
        check(self._L.hiopamd_kkt_xycyd_factorize(self.h, C.byref(ok)), "hiopamd_kkt_xycyd_factorize")
        return bool(ok.value)

    def set_fact_acceptor(self, inertia_free: bool):
        check(self._L.hiopamd_kkt_xycyd_set_fact_acceptor(self.h, 1 if inertia_free else 0), "set_fact_acceptor")

    def factorize_inertia_free(self) -> bool:
        ok = C.c_int(0)
        check(self._L.hiopamd_kkt_xycyd_factorize_inertia_free(self.h, C.byref(ok)), "factorize_inertia_free")
        return bool(ok.value)

    def test_direction(self, dir_: torch.Tensor, neg_curv_test_fact: float = 1e-11):
        acc = C.c_int(0)
        dWd, nrm = C.c_double(0), C.c_double(0)
        check(self._L.hiopamd_kkt_xycyd_test_direction(self.h, dptr(dir_, self.ctx), neg_curv_test_fact, C.byref(acc),
                                                       C.byref(dWd), C.byref(nrm)), "hiopamd_kkt_xycyd_test_direction")
        return bool(acc.value), dWd.value, nrm.value

    def deltas(self):
        d = (C.c_double * 4)()
        check(self._L.hiopamd_kkt_xycyd_deltas(self.h, d), "deltas")
        return tuple(d)

    @property
    def num_refact(self) -> int:
        return self._L.hiopamd_kkt_xycyd_num_refactorizations(self.h)

    def compute_directions(self, resid: torch.Tensor, dir_: torch.Tensor) -> bool:
        ok = C.c_int(0)
        check(self._L.hiopamd_kkt_xycyd_compute_directions(self.h, dptr(resid, self.ctx), dptr(dir_, self.ctx), C.byref(ok)),
              "hiopamd_kkt_xycyd_compute_directions")
        return bool(ok.value)

    def times_vec(self, y: torch.Tensor, x: torch.Tensor):
        check(self._L.hiopamd_kkt_xycyd_times_vec(self.h, dptr(y, self.ctx), dptr(x, self.ctx)), "hiopamd_kkt_xycyd_times_vec")

    def compute_directions_w_IR(self, resid, dir_, ir_outer_tol_factor=1e-2, ir_outer_tol_min=1e-6, ir_outer_maxit=8):
        ok, conv = C.c_int(0), C.c_int(0)
        info = (C.c_double * 4)()
        check(self._L.hiopamd_kkt_xycyd_compute_directions_w_IR(self.h, dptr(resid, self.ctx), dptr(dir_, self.ctx), ir_outer_tol_factor,
                                                                ir_outer_tol_min, ir_outer_maxit, C.byref(ok),
                                                                C.byref(conv), info),
              "hiopamd_kkt_xycyd_compute_directions_w_IR")
        return bool(ok.value), {"converged": bool(conv.value), "flag": int(info[0]), "iter": info[1],
                                "abs_resid": info[2], "rel_resid": info[3]}

    def linsolver_sys_matrix(self, n) -> torch.Tensor:
        ls = self._L.hiopamd_kkt_xycyd_linsolver(self.h)
        ptr = self._L.hiopamd_linsolver_sys_matrix(C.c_void_p(ls))
        out = torch.empty((n, n), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        check(self._L.hiopamd_copy_d2d(self.ctx.h, dptr(out, self.ctx), C.c_void_p(ptr), n * n * 8), "copy_d2d")
        self.ctx.sync()
        return out

    def close(self):
        if self.h is not None:
            if self.ctx.h is not None:
                self._L.hiopamd_kkt_xycyd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IpmSlabOps:
    """hiopIterate / hiopResidual steps on the 12-part slabs of a KKTLinSysXYcYd (hiopamd_residual_update,
    hiopamd_iterate_*): mirrors src/Optimization/hiopResidual.cpp:154 and hiopIterate.cpp:274-566."""

    def __init__(self, full: KKTLinSysXYcYd, xl, xu, dl, du, crhs):
        self.full, self._L, self.ctx = full, lib(), full.ctx
        self._b = [xl, xu, dl, du, crhs]
        torch.cuda.synchronize()
        check(self._L.hiopamd_kkt_xycyd_set_bounds(full.h, *[dptr(t, self.ctx) for t in self._b]), "set_bounds")

    def residual_update(self, it, c, d, grad_f, mu, kappa_d, resid):
        n = (C.c_double * 11)()
        check(self._L.hiopamd_residual_update(self.full.h, dptr(it, self.ctx), dptr(c, self.ctx), dptr(d, self.ctx), dptr(grad_f, self.ctx), mu, kappa_d,
                                              dptr(resid, self.ctx), n), "hiopamd_residual_update")
        return list(n)

    def fraction_to_the_bdry(self, it, dir_, tau):
        ap, ad = C.c_double(0), C.c_double(0)
        check(self._L.hiopamd_iterate_fraction_to_the_bdry(self.full.h, dptr(it, self.ctx), dptr(dir_, self.ctx), tau, C.byref(ap), C.byref(ad)),
              "fraction_to_the_bdry")
        return ap.value, ad.value

    def take_step(self, out, it, dir_, alpha_primal, alpha_dual, primals=True, duals=True):
        check(self._L.hiopamd_iterate_take_step(self.full.h, dptr(out, self.ctx), dptr(it, self.ctx), dptr(dir_, self.ctx), alpha_primal, alpha_dual,
                                                int(primals), int(duals)), "take_step")

    def determine_slacks(self, it):
        check(self._L.hiopamd_iterate_determine_slacks(self.full.h, dptr(it, self.ctx)), "determine_slacks")

    def adjust_small_slacks(self, it, it_curr, mu) -> int:
        n = C.c_int(0)
        check(self._L.hiopamd_iterate_adjust_small_slacks(self.full.h, dptr(it, self.ctx), dptr(it_curr, self.ctx), mu, C.byref(n)),
              "adjust_small_slacks")
        return n.value

    def adjust_bounds(self, it, xl, xu, dl, du):
        """hiopNlpFormulation::adjust_bounds: the (device) bound arrays follow the slacks of `it`"""
        check(self._L.hiopamd_iterate_adjust_bounds(self.full.h, dptr(it, self.ctx), dptr(xl, self.ctx), dptr(xu, self.ctx), dptr(dl, self.ctx),
                                                    dptr(du, self.ctx)), "adjust_bounds")

    def determine_duals_bounds_d(self, it, mu):
        check(self._L.hiopamd_iterate_determine_duals_bounds_d(self.full.h, dptr(it, self.ctx), mu), "determine_duals_bounds_d")

    def adjust_duals_plh(self, it, mu, kappa_sigma):
        check(self._L.hiopamd_iterate_adjust_duals_plh(self.full.h, dptr(it, self.ctx), mu, kappa_sigma), "adjust_duals_plh")

    def eval_log_barrier(self, it) -> float:
        v = C.c_double(0)
        check(self._L.hiopamd_iterate_eval_log_barrier(self.full.h, dptr(it, self.ctx), C.byref(v)), "eval_log_barrier")
        return v.value

    def duals_lsq_update(self, it, grad_f) -> bool:
        ok = C.c_int(0)
        check(self._L.hiopamd_duals_lsq_update(self.full.h, dptr(it, self.ctx), dptr(grad_f, self.ctx), C.byref(ok)), "hiopamd_duals_lsq_update")
        return bool(ok.value)

    def linear_damping_term(self, it, mu, kappa_d) -> float:
        v = C.c_double(0)
        check(self._L.hiopamd_iterate_linear_damping_term(self.full.h, dptr(it, self.ctx), mu, kappa_d, C.byref(v)),
              "linear_damping_term")
        return v.value


class PDPerturbation:
    """hiopPDPerturbation's scalar state machines on their own (host only; include/hiop_amd.h hiopamd_pd_perturbation_*):
    kind "primal_first" (hiopPDPerturbationPrimalFirstScalar), "dual_first" (…DualFirstScalar) or "null"."""
    KINDS = {"primal_first": 0, "dual_first": 1, "null": 2}

    def __init__(self, kind="primal_first", options8=None):
        self._L = lib()
        h = C.c_void_p()
        check(self._L.hiopamd_pd_perturbation_create(C.byref(h), self.KINDS[kind]), "pd_perturbation_create")
        self.h = h
        if options8 is not None:
            check(self._L.hiopamd_pd_perturbation_set_options(self.h, (C.c_double * 8)(*options8)), "pd_perturbation_set_options")

    def set_mu(self, mu):
        check(self._L.hiopamd_pd_perturbation_set_mu(self.h, float(mu)), "set_mu")

    def _call(self, name):
        ok = C.c_int(0)
        check(getattr(self._L, name)(self.h, C.byref(ok)), name)
        return bool(ok.value)

    def compute_initial_deltas(self):
        return self._call("hiopamd_pd_perturbation_compute_initial_deltas")

    def compute_perturb_wrong_inertia(self):
        return self._call("hiopamd_pd_perturbation_compute_perturb_wrong_inertia")

    def compute_perturb_singularity(self):
        return self._call("hiopamd_pd_perturbation_compute_perturb_singularity")

    def state(self):
        c, l, s = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_int * 4)()
        check(self._L.hiopamd_pd_perturbation_get(self.h, c, l, s), "pd_perturbation_get")
        return tuple(c), tuple(l), tuple(s)

    def deltas(self):
        return self.state()[0]

    def close(self):
        if self.h is not None:
            self._L.hiopamd_pd_perturbation_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
