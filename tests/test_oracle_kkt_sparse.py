"""The oracle's restatement of hiopKKTLinSysCondensedSparse (oracle/kkt_sparse.py) on the reference's SparseEx2 shape: the
condensed path must solve the UNcondensed XDYcYd system it stands for (hiopKKTLinSysSparseCondensed.cpp:364-368), also with
vector-valued perturbations; a non-convex Hessian makes the Cholesky fail (-1) until delta_wx is large enough."""
import numpy as np
import pytest

from hiop_amd import problems as pr
from oracle import kkt_sparse as ks


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


@pytest.mark.parametrize("n", [3, 10, 500])
def test_condensed_path_solves_the_uncondensed_system(n):
    r = rng(n)
    p = pr.sparse_ex2_ineq(n, x=r.uniform(0.5, 2.0, n))
    k = ks.KKTLinSysCondensedSparse(p.nx, p.nineq, (p.Jd_i, p.Jd_j), (p.H_i, p.H_j))
    Dx, Dd = r.uniform(0, 3, n), r.uniform(0.1, 5, p.nineq)
    k.set_values(p.Jd_v, p.H_v, Dx, Dd)
    for deltas in ((0.0, 0.0), (1e-4, 1e-6), (r.uniform(0.5e-4, 1.5e-4, n), r.uniform(0.5e-6, 1.5e-6, p.nineq))):
        k.build_kkt_matrix(*deltas)
        assert k.factorize() == 0
        rx, rd, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, p.nineq), r.uniform(-1, 1, p.nineq)
        ok, dx, dd, dyd = k.solve_compressed(rx, rd, ryd)
        assert ok
        assert max(ks.xdycyd_residual(k, deltas[0], deltas[1], rx, rd, ryd, dx, dd, dyd)) < 1e-12


def test_non_convex_hessian_needs_regularisation():
    n = 40
    p = pr.sparse_ex2_ineq(n)
    k = ks.KKTLinSysCondensedSparse(p.nx, p.nineq, (p.Jd_i, p.Jd_j), (p.H_i, p.H_j))
    Hneg = p.H_v.copy(); Hneg[5] = -50.0
    k.set_values(p.Jd_v, Hneg, np.zeros(n), np.full(p.nineq, 0.5))
    k.build_kkt_matrix(0.0, 0.0)
    assert k.factorize() == -1
    k.build_kkt_matrix(60.0, 0.0)
    assert k.factorize() == 0
