"""Example problems for the parity tests — TEST INFRASTRUCTURE (numpy, host side).

The input generators (constant Jacobian / Hessian blocks, bounds, barrier diagonals, right-hand sides of MdsEx1, MdsEx1G,
DenseConsEx1/2) live in hiop_amd/problems.py, where the bench takes its synthetic inputs from; they are re-exported here.
What IS oracle material is below: the literal restatement of the example's callbacks, the checker of the device-resident
callbacks (csrc/example_mds.hip).
"""
from __future__ import annotations

import numpy as np

from hiop_amd.problems import *          # noqa: F401,F403
from hiop_amd.problems import _Qd, _ineq_rows   # noqa: F401


def mds_ex1_callbacks(ns: int, nd: int, x: np.ndarray, empty_sp_row: bool = False):
    """eval_f / eval_grad_f / eval_cons of MdsEx1 at x = (x, s, y), as the reference's loops compute them
    (NlpMdsEx1.hpp:186-209, :269-289, :211-266; ns already a multiple of four).  Returns (f, grad, cons[ns + 3])."""
    Q = _Qd(nd)
    xs, s, y = x[:ns], x[ns:2 * ns], x[2 * ns:]
    f = 0.0
    for i in range(ns):                          # :194-195
        f += xs[i] * (xs[i] - 1.0)
    f *= 0.5
    f += 0.5 * float((Q @ y) @ y)                # :197-201
    f += 0.5 * float(np.sum(s * s))              # :203-206
    grad = np.concatenate([xs - 0.5, s, Q @ y])  # :273-286
    ey = float(np.sum(y))
    cons = np.empty(ns + 3)
    cons[:ns] = xs + s - ey                      # :228-231, :262-264 (Md = -1)
    cons[ns] = xs[0] + float(np.sum(s)) + ey     # :238-241
    cons[ns + 1] = (0.0 if empty_sp_row else xs[1]) + ey   # :242-248
    cons[ns + 2] = xs[2] + ey                    # :249-251
    return f, grad, cons


