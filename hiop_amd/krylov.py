"""Host-side mirror of hiopKrylovSolver / hiopPCGSolver / hiopBiCGStabSolver (src/LinAlg/hiopKrylovSolver.hpp:80-258) over
the C ABI (hiopamd_krylov_*).  Linear operators are Python callables (x: device tensor view, y: device tensor view) ->
None that fill y; they are wrapped as C callbacks on raw device pointers."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import LINOP_FN, check, lib
from .runtime import Context, dptr


def _view(ptr: int, n: int) -> torch.Tensor:
    """A float64 tensor view of n doubles at a raw device address (no copy, no ownership)."""
    class _Cai:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3, "strides": None}
    return torch.as_tensor(_Cai(), device="cuda")


class KrylovSolver:
    PCG, BICGSTAB = 0, 1

    def __init__(self, ctx: Context, kind: int, n: int, A, Mleft=None, Mright=None):
        self.ctx, self.n = ctx, n
        self._L = lib()
        self._cbs = []

        def wrap(op):
            if op is None:
                return C.cast(None, LINOP_FN)

            def cb(user, x, y):
                try:
                    op(_view(x, n), _view(y, n))
                    return 0
                except Exception:      # an exception cannot cross the C frame
                    import traceback
                    traceback.print_exc()
                    return -2
            f = LINOP_FN(cb)
            self._cbs.append(f)      # keep the trampolines alive as long as the solver
            return f
        h = C.c_void_p()
        check(self._L.hiopamd_krylov_create(C.byref(h), ctx.h, kind, n, wrap(A), None, wrap(Mleft), None, wrap(Mright), None),
              "hiopamd_krylov_create")
        self.h = h
        ctx._register(self)

    def set_tol(self, tol: float):
        check(self._L.hiopamd_krylov_set_tol(self.h, tol), "set_tol")

    def set_max_num_iter(self, maxit: int):
        check(self._L.hiopamd_krylov_set_max_num_iter(self.h, maxit), "set_max_num_iter")

    def set_exit_mode(self, reference: bool):
        """BiCGStab 'tol is too small' exit: True (default) = the reference's closing comparison against the overwritten right-hand
        side (hiopKrylovSolver.cpp:561-566, :639-644), False = against the original one."""
        check(self._L.hiopamd_krylov_set_exit_mode(self.h, 1 if reference else 0), "set_exit_mode")

    def set_x0(self, xval: float):
        check(self._L.hiopamd_krylov_set_x0(self.h, xval), "set_x0")

    def x0(self) -> torch.Tensor:
        return _view(self._L.hiopamd_krylov_x0(self.h), self.n)

    def solve(self, b: torch.Tensor) -> bool:
        ok = C.c_int(0)
        check(self._L.hiopamd_krylov_solve(self.h, dptr(b, self.ctx), C.byref(ok)), "hiopamd_krylov_solve")
        return bool(ok.value)

    def get_convergence_flag(self) -> int:
        return self._L.hiopamd_krylov_get_convergence_flag(self.h)

    def get_sol_num_iter(self) -> float:
        return self._L.hiopamd_krylov_get_sol_num_iter(self.h)

    def get_sol_abs_resid(self) -> float:
        return self._L.hiopamd_krylov_get_sol_abs_resid(self.h)

    def get_sol_rel_resid(self) -> float:
        return self._L.hiopamd_krylov_get_sol_rel_resid(self.h)

    def close(self):
        if self.h:
            self._L.hiopamd_krylov_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
