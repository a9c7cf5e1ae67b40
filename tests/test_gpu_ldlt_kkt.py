"""GPU parity of the hiopLinSolverSymDense operator (blocked no-pivot LDL^T on fp64 MFMA) and of the
condensed MDS KKT (assemble -> factor+inertia -> solveCompressed) against the oracle (LAPACK
Bunch-Kaufman path of the reference) and the reference's own `write_kkt` dumps.

Tolerances (fp64): factors vs the unblocked no-pivot recurrence rtol 1e-9 (conditioning of the
quasi-definite test matrices ~1e4); solutions: relative KKT residual <= 1e-10 and agreement with the
LAPACK solution <= 1e-8 relative; inertia exact."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import hiop_oracle as ho
from oracle import problems as pr
from oracle.iajaaa import read_iajaaa

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def D(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def quasi_definite(n1, n2, seed):
    """[[H, J^T],[J, -C]] with H, C SPD: every symmetric permutation has an LDL^T (Vanderbei)."""
    r = rng(seed)
    n = n1 + n2
    G = r.uniform(-1, 1, (n1, n1)); H = G @ G.T / n1 + np.eye(n1) * 2.0
    J = r.uniform(-1, 1, (n2, n1))
    Cm = np.diag(r.uniform(0.5, 2.0, n2))
    A = np.zeros((n, n))
    A[:n1, :n1] = H; A[n1:, :n1] = J; A[:n1, n1:] = J.T; A[n1:, n1:] = -Cm
    return A


@pytest.mark.parametrize("n1,n2", [(1, 0), (3, 2), (40, 24), (64, 1), (65, 64), (100, 157), (300, 203), (700, 324)])
def test_ldlt_factor_matches_unblocked_recurrence(ctx, n1, n2):
    from hiop_amd.kkt import LinSolverSymDense
    A = quasi_definite(n1, n2, n1 * 13 + n2)
    n = n1 + n2
    ls = LinSolverSymDense(ctx, n)
    Mu = np.triu(A) + np.tril(rng(1).uniform(5, 6, (n, n)), -1)  # garbage below the diagonal must be ignored
    ls.set_sys_matrix(D(Mu))
    nneg = ls.matrix_changed()
    assert nneg == n2
    assert ls.inertia() == (n1, n2, 0)
    F = ls.get_sys_matrix().cpu().numpy()
    U, d = ho.ldlt_nopiv(A)
    np.testing.assert_allclose(np.diag(F), d, rtol=1e-9)
    np.testing.assert_allclose(np.triu(F, 1), np.triu(U, 1), rtol=1e-8, atol=1e-11)
    # lower triangle untouched
    np.testing.assert_array_equal(np.tril(F, -1), np.tril(Mu, -1))
    # solves
    r = rng(2)
    for nrhs in (1, 3):
        B = r.uniform(-1, 1, (nrhs, n))
        Bd = D(B)
        torch.cuda.synchronize()
        ls.solve(Bd, nrhs)
        ctx.sync()
        X = Bd.cpu().numpy()
        for q in range(nrhs):
            res = np.abs(A @ X[q] - B[q]).max() / (np.abs(A).max() * np.abs(X[q]).max())
            assert res < 1e-12
    ls.close()


@pytest.mark.parametrize("n", [768, 1024, 1124, 1280, 2049, 4096])
def test_dataflow_factorisation_equals_stepwise(ctx, n):
    """The factorisation as two persistent dataflow kernels (csrc/ldlt_dataflow.hpp) against the stepwise kernels on the
    same matrix: same inertia, factors equal to rounding (the tile products are summed in a different grouping), the
    unblocked recurrence to 1e-9, and repeatable (flags are re-initialised by every call)."""
    from hiop_amd.kkt import LinSolverSymDense
    n1 = (2 * n) // 3
    A = quasi_definite(n1, n - n1, n)
    ls = LinSolverSymDense(ctx, n)
    Mu = D(np.triu(A))
    out = {}
    for mode in (True, False, True):
        ls.set_dataflow(mode)
        ls.set_sys_matrix(Mu)
        assert ls.matrix_changed() == n - n1
        Fm = ls.get_sys_matrix().cpu().numpy()
        out.setdefault(mode, []).append(np.triu(Fm))
    a, b = out[True]
    assert np.array_equal(a, b)                      # run-to-run reproducible: fixed summation order inside every tile
    s = out[False][0]
    assert np.abs(a - s).max() <= 1e-11 * np.abs(s).max()
    U, d = ho.ldlt_nopiv(A)
    np.testing.assert_allclose(np.diag(a), d, rtol=1e-9)
    np.testing.assert_allclose(np.triu(a, 1), np.triu(U, 1), rtol=1e-8, atol=1e-11)
    # and the dataflow solve on top of it
    r = rng(4)
    B = r.uniform(-1, 1, n)
    Bd = D(B)
    torch.cuda.synchronize()
    ls.solve(Bd, 1)
    ctx.sync()
    X = Bd.cpu().numpy()
    assert np.abs(A @ X - B).max() / (np.abs(A).max() * np.abs(X).max()) < 1e-12
    ls.close()


def test_ldlt_full_size_property(ctx):
    """N=8192 (BASELINE config 3 size): residual + inertia properties, no O(N^3) host work."""
    from hiop_amd.kkt import LinSolverSymDense
    n1, n2 = 4096, 4096
    n = n1 + n2
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    J = torch.rand((n2, n1), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    Hd = torch.rand(n1, generator=g, device="cuda", dtype=torch.float64) + 50.0
    Hoff = (torch.rand((n1, n1), generator=g, device="cuda", dtype=torch.float64) - 0.5) * 0.01
    A = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    A[:n1, :n1] = torch.diag(Hd) + (Hoff + Hoff.T)
    A[:n1, n1:] = J.T
    A[n1:, :n1] = J
    A[n1:, n1:] = -torch.diag(torch.rand(n2, generator=g, device="cuda", dtype=torch.float64) + 0.5)
    ls = LinSolverSymDense(ctx, n)
    ls.set_sys_matrix(torch.triu(A))
    nneg = ls.matrix_changed()
    assert nneg == n2 and ls.inertia() == (n1, n2, 0)
    b = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    x = b.clone()
    torch.cuda.synchronize()
    ls.solve(x, 1); ctx.sync()
    res = (A @ x - b).abs().max().item() / (A.abs().max().item() * x.abs().max().item())
    assert res < 1e-11
    ls.close()


def test_singular_matrix_returns_minus_one(ctx):
    from hiop_amd.kkt import LinSolverSymDense
    n = 70
    A = quasi_definite(50, 20, 5)
    A[10, :] = 0; A[:, 10] = 0      # exactly zero pivot in no-pivot elimination order -> 'singular' (-1)
    ls = LinSolverSymDense(ctx, n)
    ls.set_sys_matrix(D(np.triu(A)))
    assert ls.matrix_changed() == -1
    # solve before a successful factorisation is a call-sequence error, not a silent garbage solve
    from hiop_amd import HiopAmdError
    with pytest.raises(HiopAmdError):
        ls.solve(D(np.ones(n)), 1)
    ls.close()


@pytest.mark.parametrize("it", [0, 5, 10])
def test_golden_reference_kkt_systems(ctx, it):
    """(matrix, rhs) -> solution triples written by the reference's LAPACK path (`write_kkt yes`)."""
    from hiop_amd.kkt import LinSolverSymDense
    g = read_iajaaa(GOLD / f"kkt_linsys_{it}.iajaaa")
    n = g["n"]
    ls = LinSolverSymDense(ctx, n)
    ls.set_sys_matrix(D(g["M_upper"]))
    assert ls.matrix_changed() == g["neq"] + g["nineq"]
    for rhs, sol in g["pairs"]:
        x = D(rhs)
        torch.cuda.synchronize()
        ls.solve(x, 1); ctx.sync()
        np.testing.assert_allclose(x.cpu().numpy(), sol, rtol=1e-9, atol=1e-9 * np.abs(sol).max())
    ls.close()


def _kkt_pair(ctx, p, seed=3):
    from hiop_amd.kkt import mds_from_problem
    Dx, Dd = pr.barrier_diagonals(p, seed=seed)
    ko = ho.KKTLinSysCompressedMDSXYcYd(p.nxs, p.nxd, p.neq, p.nineq, (p.Jcs_i, p.Jcs_j), (p.Jds_i, p.Jds_j),
                                        (p.Hss_i, p.Hss_j))
    ko.set_values(p.Jcs_v, p.Jds_v, p.Hss_v, p.Jcd, p.Jdd, p.Hdd, Dx, Dd)
    kg, dv = mds_from_problem(ctx, p)
    dv["Dx"], dv["Dd"] = D(Dx), D(Dd)
    kg.set_values(dv["Jcs_v"], dv["Jds_v"], dv["Hss_v"], dv["Jcd"], dv["Jdd"], dv["Hdd"], dv["Dx"], dv["Dd"])
    return ko, kg, dv


PROBLEMS = [
    lambda: pr.mds_ex1(4, 4),
    lambda: pr.mds_ex1(40, 12),
    lambda: pr.mds_ex1(40, 12, empty_sp_row=True),
    lambda: pr.mds_ex1(400, 100),
    lambda: pr.mds_ex1_g(600, 130, 257),
    lambda: pr.mds_ex1_g(900, 700, 500),   # N = 1203: the solver object works at a padded order, the KKT object assembles into its padded copy
]


@pytest.mark.parametrize("mk", PROBLEMS)
@pytest.mark.parametrize("deltas", [(0.0, 0.0, 0.0, 0.0), (1e-4, 1e-4, 1e-8, 1e-8)])
def test_kkt_mds_assemble_factor_solve(ctx, mk, deltas):
    p = mk()
    ko, kg, dv = _kkt_pair(ctx, p)
    Mo = ko.build_kkt_matrix(*deltas).copy()
    kg.build_kkt_matrix(*deltas)
    Mg = kg.sys_matrix().cpu().numpy()
    np.testing.assert_allclose(np.triu(Mg), np.triu(Mo), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(kg.Hxs().cpu().numpy(), ko.Hxs, rtol=1e-15)
    n_o = ko.factorize_with_curv_check()
    n_g = kg.factorize_with_curv_check()
    assert n_o == p.neq + p.nineq
    assert n_g == n_o
    rx, ryc, ryd = pr.random_rhs(p)
    ok, dx_o, dyc_o, dyd_o = ko.solve_compressed(rx, ryc, ryd)
    assert ok
    dx, dyc, dyd = D(np.zeros_like(rx)), D(np.zeros_like(ryc)), D(np.zeros_like(ryd))
    rxd, rycd, rydd = D(rx), D(ryc), D(ryd)
    torch.cuda.synchronize()
    kg.solve_compressed(rxd, rycd, rydd, dx, dyc, dyd)
    ctx.sync()
    dx, dyc, dyd = dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy()
    # fp64 tolerance on the KKT residual of the UNcondensed system (north_star's parity statement)
    res = ho.kkt_mds_full_residual(ko, deltas, rx, ryc, ryd, dx, dyc, dyd)
    res_o = ho.kkt_mds_full_residual(ko, deltas, rx, ryc, ryd, dx_o, dyc_o, dyd_o)
    assert max(res) < 1e-12, (res, res_o)   # componentwise backward error
    scale = max(np.abs(dx_o).max(), np.abs(dyc_o).max(), np.abs(dyd_o).max())
    assert np.abs(dx - dx_o).max() / scale < 1e-8
    assert np.abs(dyc - dyc_o).max() / scale < 1e-8
    assert np.abs(dyd - dyd_o).max() / scale < 1e-8
    # rx/ryc are inputs only (reference :341-343 keeps them intact)
    np.testing.assert_array_equal(rxd.cpu().numpy(), rx)
    np.testing.assert_array_equal(rycd.cpu().numpy(), ryc)
    kg.close()


@pytest.mark.parametrize("mk", PROBLEMS[1:])
def test_kkt_mds_vector_regularisation(ctx, mk):
    """hiopamd_kkt_mds_build_vec: the perturbations as VECTORS, entry by entry, like the reference adds them
    (hiopKKTLinSysMDS.cpp:178-181,213-215,223-227,245,280,289-290 — what hiopPDPerturbationPrimalFirstRand / DualFirstRand
    produce); a null pointer is a zero vector; constant vectors reproduce the scalar entry point bit for bit."""
    p = mk()
    ko, kg, dv = _kkt_pair(ctx, p)
    r = rng(77)
    dwx = r.uniform(0.5e-4, 1.5e-4, p.nxs + p.nxd)
    dwd = r.uniform(0.5e-4, 1.5e-4, p.nineq)
    dcc = r.uniform(0.5e-8, 1.5e-8, p.neq)
    dcd = r.uniform(0.5e-8, 1.5e-8, p.nineq)
    Mo = ko.build_kkt_matrix(dwx, dwd, dcc, dcd).copy()
    kg.build_kkt_matrix(D(dwx), D(dwd), D(dcc), D(dcd))
    Mg = kg.sys_matrix().cpu().numpy()
    np.testing.assert_allclose(np.triu(Mg), np.triu(Mo), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(kg.Hxs().cpu().numpy(), ko.Hxs, rtol=1e-15)
    assert kg.factorize_with_curv_check() == ko.factorize_with_curv_check() == p.neq + p.nineq
    rx, ryc, ryd = pr.random_rhs(p)
    dx, dyc, dyd = D(np.zeros_like(rx)), D(np.zeros_like(ryc)), D(np.zeros_like(ryd))
    rxd, rycd, rydd = D(rx), D(ryc), D(ryd)
    torch.cuda.synchronize()
    kg.solve_compressed(rxd, rycd, rydd, dx, dyc, dyd)
    ctx.sync()
    res = ho.kkt_mds_full_residual(ko, (dwx, dwd, dcc, dcd), rx, ryc, ryd, dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy())
    assert max(res) < 1e-12, res
    # constant vectors == scalars, bit for bit; null pointers == zeros
    kg.build_kkt_matrix(1e-4, 1e-4, 1e-8, 1e-8)
    Ms = kg.sys_matrix()
    kg.build_kkt_matrix(D(np.full(p.nxs + p.nxd, 1e-4)), D(np.full(p.nineq, 1e-4)), D(np.full(p.neq, 1e-8)), D(np.full(p.nineq, 1e-8)))
    assert torch.equal(torch.triu(kg.sys_matrix()), torch.triu(Ms))
    kg.build_kkt_matrix(0.0, 0.0, 0.0, 0.0)
    M0 = kg.sys_matrix()
    kg.build_kkt_matrix(None, None, None, None)
    assert torch.equal(torch.triu(kg.sys_matrix()), torch.triu(M0))
    kg.close()


def test_kkt_mds_inertia_correction_signal(ctx):
    """A negative entry in Hxs (non-convex sparse block) must show up in the inertia count through
    Haynsworth additivity (hiopKKTLinSysMDS.cpp:83-108); a zero entry must yield -1."""
    p = pr.mds_ex1(40, 12)
    ko, kg, dv = _kkt_pair(ctx, p)
    Hneg = p.Hss_v.copy(); Hneg[3] = -5.0
    dv["Hss_v2"] = D(Hneg)
    kg.set_values(dv["Jcs_v"], dv["Jds_v"], dv["Hss_v2"], dv["Jcd"], dv["Jdd"], dv["Hdd"], dv["Dx"], dv["Dd"])
    ko.set_values(p.Jcs_v, p.Jds_v, Hneg, p.Jcd, p.Jdd, p.Hdd, ko.Dx, ko.Dd)
    ko.build_kkt_matrix(0, 0, 0, 0); kg.build_kkt_matrix(0, 0, 0, 0)
    assert kg.factorize_with_curv_check() == ko.factorize_with_curv_check()
    Hz = p.Hss_v.copy(); Hz[5] = -ko.Dx[5]
    dv["Hss_v3"] = D(Hz)
    kg.set_values(dv["Jcs_v"], dv["Jds_v"], dv["Hss_v3"], dv["Jcd"], dv["Jdd"], dv["Hdd"], dv["Dx"], dv["Dd"])
    kg.build_kkt_matrix(0, 0, 0, 0)
    assert kg.factorize_with_curv_check() == -1
    kg.close()


def test_kkt_mds_mode_switches_between_assembly_and_factorisation_at_a_padded_order(ctx):
    """N = 1203: the solver object works at a padded order and the KKT object assembles straight into its padded copy
    (hiopamd_linsolver_assembly_matrix).  Safe mode and the pivoted solver work on the N x N matrix instead: switching one of them on
    AFTER an assembly and BEFORE its factorisation must carry the assembled matrix over; switching back must return to the padded copy."""
    p = pr.mds_ex1_g(900, 700, 500)
    ko, kg, dv = _kkt_pair(ctx, p)
    deltas = (1e-4, 1e-4, 1e-8, 1e-8)
    ko.build_kkt_matrix(*deltas)
    n_o = ko.factorize_with_curv_check()
    rx, ryc, ryd = pr.random_rhs(p)
    ok, dx_o, dyc_o, dyd_o = ko.solve_compressed(rx, ryc, ryd)
    assert ok and n_o == p.neq + p.nineq
    scale = max(np.abs(dx_o).max(), np.abs(dyc_o).max(), np.abs(dyd_o).max())
    L = kg._L

    def solve():
        dx, dyc, dyd = D(np.zeros_like(rx)), D(np.zeros_like(ryc)), D(np.zeros_like(ryd))
        rxd, rycd, rydd = D(rx), D(ryc), D(ryd)
        torch.cuda.synchronize()
        kg.solve_compressed(rxd, rycd, rydd, dx, dyc, dyd)
        ctx.sync()
        return dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy()

    for mode in (1, 2, 0, 1, 0):          # regularise-and-refine, Bunch-Kaufman, plain (padded copy), and again
        # the control: the mode set BEFORE the assembly (which then writes the N x N matrix for modes 1 and 2)
        assert L.hiopamd_kkt_mds_set_safe_mode(kg.h, mode) == 0
        kg.build_kkt_matrix(*deltas)
        assert kg.factorize_with_curv_check() == n_o, mode
        want = solve()
        # the case: assembled in plain mode (into the padded copy), the mode switched on afterwards
        assert L.hiopamd_kkt_mds_set_safe_mode(kg.h, 0) == 0
        kg.build_kkt_matrix(*deltas)
        assert L.hiopamd_kkt_mds_set_safe_mode(kg.h, mode) == 0
        assert kg.factorize_with_curv_check() == n_o, mode
        got = solve()
        for a, b in zip(got, want):
            assert np.array_equal(a, b), mode      # the same matrix through the same kernels: the same bits
        res = ho.kkt_mds_full_residual(ko, deltas, rx, ryc, ryd, *got)
        assert max(res) < (1e-9 if mode == 1 else 1e-11), (mode, res)   # (safe mode refines to 1e-13 (||K|| ||x|| + ||b||), not componentwise)
        assert np.abs(got[0] - dx_o).max() / scale < (1e-5 if mode == 1 else 1e-8), mode   # (forward error of the refined solve: x cond(K))
    kg.close()


def test_kkt_mds_full_size_roundtrip(ctx):
    """BASELINE config 3 (n_sparse=1e5, n_dense=4096, m=4096 -> N=8192): assemble -> factor -> solve, checked
    by the size-independent property 'residual of the uncondensed KKT system' (sparse mat-vecs on host)."""
    p = pr.mds_ex1_g(50000, 4096, 4093)
    assert p.N == 8192 and p.nxs == 100000
    ko, kg, dv = _kkt_pair(ctx, p)
    deltas = (0.0, 0.0, 0.0, 0.0)
    kg.build_kkt_matrix(*deltas)
    assert kg.factorize_with_curv_check() == p.neq + p.nineq
    rx, ryc, ryd = pr.random_rhs(p)
    dx, dyc, dyd = D(np.zeros_like(rx)), D(np.zeros_like(ryc)), D(np.zeros_like(ryd))
    rxd, rycd, rydd = D(rx), D(ryc), D(ryd)
    torch.cuda.synchronize()
    kg.solve_compressed(rxd, rycd, rydd, dx, dyc, dyd); ctx.sync()
    res = ho.kkt_mds_full_residual(ko, deltas, rx, ryc, ryd, dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy())
    assert max(res) < 1e-11, res   # componentwise backward error of the uncondensed system
    kg.close()


class _GpuKktAdapter:
    """oracle-class interface over the HIP condensed MDS KKT (numpy in / numpy out)."""

    def __init__(self, ctx, p):
        from hiop_amd.kkt import mds_from_problem
        self.ctx, self.p = ctx, p
        self.k, self.dv = mds_from_problem(ctx, p)

    def set_values(self, Jcs_v, Jds_v, Hss_v, Jcd, Jdd, Hdd, Dx, Dd):
        dv = self.dv
        dv["Dx"], dv["Dd"] = D(Dx), D(Dd)
        self.k.set_values(dv["Jcs_v"], dv["Jds_v"], dv["Hss_v"], dv["Jcd"], dv["Jdd"], dv["Hdd"], dv["Dx"], dv["Dd"])

    def build_kkt_matrix(self, *deltas):
        self.k.build_kkt_matrix(*deltas)

    def factorize_with_curv_check(self):
        return self.k.factorize_with_curv_check()

    def solve_compressed(self, rx, ryc, ryd):
        dx, dyc, dyd = D(np.zeros_like(rx)), D(np.zeros_like(ryc)), D(np.zeros_like(ryd))
        a, b, c = D(rx), D(ryc), D(ryd)
        torch.cuda.synchronize()
        self.k.solve_compressed(a, b, c, dx, dyc, dyd)
        self.ctx.sync()
        return True, dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy()


@pytest.mark.parametrize("ns,nd", [(40, 12), (400, 100)])
def test_ipm_iterations_on_the_hip_kkt_track_the_cpu_path(ctx, ns, nd):
    """Whole interior-point solves of the reference's MdsEx1 with every KKT assemble/factor/solve done by the HIP
    path: same number of iterations and the same objective / KKT-error / mu trajectory as with the oracle (LAPACK)
    KKT — the reference's own CPU-vs-GPU parity bar is 1e-5 absolute on the iteration table
    (tests/testMDS1CompareIterations.awk:13); here 1e-7 relative."""
    import json
    from oracle import ipm
    p = pr.mds_ex1(ns, nd)
    t_cpu, t_gpu = [], []
    r_cpu = ipm.solve_mds(p, mu0=0.1, tol=1e-5, trace=t_cpu)
    ad = _GpuKktAdapter(ctx, p)
    r_gpu = ipm.solve_mds(p, mu0=0.1, tol=1e-5, kkt=ad, trace=t_gpu)
    assert r_gpu["iters"] == r_cpu["iters"]
    a, b = np.array(t_cpu), np.array(t_gpu)
    np.testing.assert_allclose(b[:, 0], a[:, 0], rtol=1e-7, atol=1e-9)     # objective
    np.testing.assert_allclose(b[:, 2], a[:, 2], rtol=0, atol=0)            # mu schedule identical
    np.testing.assert_allclose(b[:, 1], a[:, 1], rtol=1e-4, atol=1e-9)     # KKT error
    if (ns, nd) == (400, 100):
        gold = json.loads((GOLD / "selfcheck_objectives.json").read_text())["MdsEx1"]
        assert abs(r_gpu["obj"] - gold["objective"]) < 1e-4
    ad.k.close()


@pytest.mark.parametrize("fname", ["kkt_linsys_0.iajaaa", "kkt_linsys_10.iajaaa"])
def test_iajaaa_writer_reproduces_reference_files(ctx, fname, tmp_path):
    """hiopamd_io_write_iajaaa_matrix / _append_iajaaa_vector (hiopCSR_IO.hpp:44-152): a dump of the reference's own
    matrix, rhs and solution from device memory must be token-for-token the file the reference wrote."""
    import ctypes as C
    import os
    from hiop_amd._lib import lib, check
    from oracle.iajaaa import read_iajaaa
    src = os.path.join(os.path.dirname(__file__), "golden", fname)
    g = read_iajaaa(src)
    n = g["n"]
    Md = torch.as_tensor(np.ascontiguousarray(g["M_upper"])).cuda()
    out = str(tmp_path / fname).encode()
    L = lib()
    torch.cuda.synchronize()
    check(L.hiopamd_io_write_iajaaa_matrix(ctx.h, C.c_char_p(out), n, C.c_void_p(Md.data_ptr()), n, g["nx"], g["neq"],
                                           g["nineq"]), "write_iajaaa_matrix")
    for rhs, sol in g["pairs"]:
        for v in (rhs, sol):
            vd = torch.as_tensor(np.ascontiguousarray(v)).cuda()
            torch.cuda.synchronize()
            check(L.hiopamd_io_append_iajaaa_vector(ctx.h, C.c_char_p(out), n, C.c_void_p(vd.data_ptr())), "append")
    assert open(out.decode()).read().split() == open(src).read().split()


@pytest.mark.parametrize("n", [1, 17, 255, 256, 257, 511, 513, 1023, 1280, 2049, 2048, 2560, 4096])
def test_dataflow_solve_block_boundaries_multi_rhs_and_repeats(ctx, n):
    """hiopamd_linsolver_solve around the 256-block boundaries of the dataflow kernel (ragged last block, one block, N = 1;
    multiples of 512 from 2048 on run the 512-block task graph with the inverted 512 x 512 diagonal blocks),
    several right-hand sides in one call, and repeated solves (the exchange buffers alternate by epoch parity and are
    re-poisoned by the previous launch): every solve must reproduce the dense solution and repeat bit for bit."""
    from hiop_amd.kkt import LinSolverSymDense
    g = torch.Generator(device="cuda"); g.manual_seed(100 + n)
    n1 = (n + 1) // 2
    M = torch.rand((n, n), generator=g, device="cuda", dtype=torch.float64) - 0.5
    M = M + M.T
    sgn = torch.ones(n, device="cuda", dtype=torch.float64); sgn[n1:] = -1.0
    M = M + torch.diag(sgn * (2.0 + 0.6 * n ** 0.5 * 4))          # quasi-definite: n1 positive, n - n1 negative pivots
    ls = LinSolverSymDense(ctx, n)
    ls.set_sys_matrix(torch.triu(M))
    assert ls.matrix_changed() == n - n1
    nrhs = 3
    B = torch.rand((nrhs, n), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    X = B.clone()
    ls.solve(X, nrhs); ctx.sync()
    ref = torch.linalg.solve(M, B.T).T
    assert (X - ref).abs().max().item() <= 1e-10 * max(1.0, ref.abs().max().item())
    first = None
    for rep in range(5):                                        # odd and even epochs
        x = B[1].clone()
        ls.solve(x, 1); ctx.sync()
        if first is None:
            first = x.clone()
        assert torch.equal(x, first)
    assert torch.equal(first, X[1])


# ((900, 437): order 1337 — odd and not a multiple of 256: the object factors and solves it at a padded order, two right-hand sides)
@pytest.mark.parametrize("n1,n2", [(300, 200), (700, 324), (900, 437)])
def test_safe_mode_tiny_leading_pivot(ctx, n1, n2):
    """Safe mode (the reference's switch to the Bunch-Kaufman solver, hiopKKTLinSysMDS.cpp:408-430): a quasi-definite matrix
    whose leading pivot is 1e-13 ||A||.  The plain no-pivot factor has element growth ~1e13 — reported by the growth
    monitor — and its solution is useless; with safe mode on (static quasi-definite regularisation + refinement against the
    saved matrix) the solution matches LAPACK's (numpy.linalg.solve = DGESV) to 1e-8, or solve_status says it failed."""
    from hiop_amd.kkt import LinSolverSymDense
    r = rng(n1 + 7 * n2)
    A = quasi_definite(n1, n2, 3 * n1 + n2)
    n = n1 + n2
    amax = np.abs(A).max()
    A[0, :n1] = 0.0; A[:n1, 0] = 0.0            # keep H positive definite: its first row/column is just the tiny diagonal
    A[0, 0] = 1e-13 * amax
    A[0, n1:] = r.uniform(0.5, 1.0, n2); A[n1:, 0] = A[0, n1:]   # ... but the variable is well determined by the constraints
    B = r.uniform(-1, 1, (2, n))
    want = np.linalg.solve(A, B.T).T
    ls = LinSolverSymDense(ctx, n)
    # ---- fast path: factor "succeeds", the growth monitor shows why it must not be trusted
    ls.set_sys_matrix(D(np.triu(A)))
    nneg = ls.matrix_changed()
    u, dmin, dmax = ls.growth()
    assert u > 1e10 * amax or nneg != n2
    # ---- safe mode
    ls.set_safe_mode(True, n1)
    ls.set_sys_matrix(D(np.triu(A)))
    nneg = ls.matrix_changed()
    assert nneg == n2                              # inertia of the regularised quasi-definite matrix
    u2, _, _ = ls.growth()
    assert u2 < 1e9 * amax                         # growth bounded by ~ ||A|| / delta, delta = sqrt(eps) ||A||
    Bd = D(B)
    torch.cuda.synchronize()
    ls.solve(Bd, 2)
    ctx.sync()
    ok = ls.solve_status()
    its, res = ls.safe_mode_info()
    X = Bd.cpu().numpy()
    if ok:
        assert res <= 1e-13 and its <= 10
        np.testing.assert_allclose(X, want, rtol=1e-8, atol=1e-8 * np.abs(want).max())
    # a well-conditioned matrix in safe mode: same answer as the fast path to refinement accuracy, 1-2 refinements
    A2 = quasi_definite(n1, n2, 11)
    ls.set_sys_matrix(D(np.triu(A2)))
    assert ls.matrix_changed() == n2
    Bd = D(B)
    torch.cuda.synchronize()
    ls.solve(Bd, 2); ctx.sync()
    assert ls.solve_status()
    np.testing.assert_allclose(Bd.cpu().numpy(), np.linalg.solve(A2, B.T).T, rtol=1e-10, atol=1e-11)
    assert ls.safe_mode_info()[0] <= 3
    ls.close()


def test_dataflow_kernels_under_reduced_residency(ctx):
    """Co-residency: the dataflow solve (one launch, tasks waiting for each other's flags) and the dataflow factorisation
    (two persistent kernels) must complete whatever share of the device they get — tickets / queues are handed out in a
    topological order, so a resident workgroup only ever waits for work that is already running.  Here another stream keeps
    the device busy with large fp64 GEMMs (torch) while factorisations and solves run; results must be bitwise those of the
    quiet device, and no bounded wait may fire."""
    from hiop_amd.kkt import LinSolverSymDense
    n = 2048
    A = quasi_definite(2 * n // 3, n - 2 * n // 3, 77)
    B = rng(5).uniform(-1, 1, (3, n))
    ls = LinSolverSymDense(ctx, n)
    Mu = D(np.triu(A))

    def run_once():
        ls.set_sys_matrix(Mu)
        nneg = ls.matrix_changed()
        Bd = D(B)
        torch.cuda.synchronize()
        ls.solve(Bd, 3)
        ctx.sync()
        assert ls.solve_status()
        return nneg, ls.get_sys_matrix().cpu().numpy(), Bd.cpu().numpy()

    quiet = run_once()
    side = torch.cuda.Stream()
    G = torch.rand(4096, 4096, dtype=torch.float64, device="cuda")
    out = torch.empty_like(G)
    with torch.cuda.stream(side):
        for _ in range(60):            # ~ 60 x 1.8 ms of device-filling work queued behind each other
            torch.mm(G, G, out=out)
    busy = [run_once() for _ in range(3)]
    side.synchronize()
    for nneg, F, X in busy:
        assert nneg == quiet[0]
        assert np.array_equal(F, quiet[1])
        assert np.array_equal(X, quiet[2])
    ls.close()
