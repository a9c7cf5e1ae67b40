"""Device-resident user callbacks of the reference's MDS example: `hiopamd_mdsex1_*` =
the hiopInterfaceMDS callbacks of class MdsEx1 (src/Drivers/MDS/NlpMdsEx1.hpp; RAJA twin NlpMdsRajaEx1.cpp) with device
pointers.  Checked against the numpy restatement of the same class (constant blocks: hiop_amd/problems.py; callbacks:
oracle/problems.py — the blocks are the ones the KKT parity tests and the
stored -selfcheck objectives already pin), including the layouts hiopamd_kkt_mds_set_values() consumes."""
import ctypes as C

import numpy as np
import pytest
import torch

from hiop_amd.runtime import dptr
from oracle import problems as op

pytestmark = pytest.mark.gpu


def dev(n, dtype=torch.float64):
    return torch.full((max(int(n), 1),), -777, dtype=dtype, device="cuda")


@pytest.mark.parametrize("ns_in,nd,empty", [(40, 12, False), (400, 100, False), (38, 7, True), (4, 1, False), (2000, 300, True)])
def test_callbacks_equal_the_reference_example(ctx, ns_in, nd, empty):
    L = ctx._L
    h = C.c_void_p()
    assert L.hiopamd_mdsex1_create(C.byref(h), ctx.h, ns_in, nd, int(empty)) == 0
    p = op.mds_ex1(ns_in, nd, empty)          # rounds ns up to a multiple of four like the reference
    ns = p.neq
    n, m = C.c_int64(), C.c_int64()
    assert L.hiopamd_mdsex1_get_prob_sizes(h, C.byref(n), C.byref(m)) == 0
    assert (n.value, m.value) == (2 * ns + nd, ns + 3)
    info = [C.c_int() for _ in range(6)]
    assert L.hiopamd_mdsex1_get_sparse_dense_blocks_info(h, *[C.byref(v) for v in info]) == 0
    assert [v.value for v in info] == [p.nxs, p.nxd, p.Jcs_i.size, p.Jds_i.size, p.Hss_i.size, 0]

    # bounds, starting point
    xl, xu, cl, cu, x0 = dev(n.value), dev(n.value), dev(m.value), dev(m.value), dev(n.value)
    assert L.hiopamd_mdsex1_get_vars_info(h, dptr(xl), dptr(xu)) == 0
    assert L.hiopamd_mdsex1_get_cons_info(h, dptr(cl), dptr(cu)) == 0
    assert L.hiopamd_mdsex1_get_starting_point(h, dptr(x0)) == 0
    ctx.sync()
    np.testing.assert_array_equal(xl.cpu().numpy(), p.xl)
    np.testing.assert_array_equal(xu.cpu().numpy(), p.xu)
    np.testing.assert_array_equal(cl.cpu().numpy(), np.concatenate([np.zeros(ns), p.dl]))
    np.testing.assert_array_equal(cu.cpu().numpy(), np.concatenate([np.zeros(ns), p.du]))
    np.testing.assert_array_equal(x0.cpu().numpy(), p.x0)

    # objective, gradient, constraints at a random point
    r = np.random.Generator(np.random.PCG64(ns + nd))
    xh = r.uniform(-1.5, 2.5, n.value)
    x = torch.as_tensor(xh).cuda()
    f = C.c_double()
    g, c = dev(n.value), dev(m.value)
    assert L.hiopamd_mdsex1_eval_f(h, dptr(x), C.byref(f)) == 0
    assert L.hiopamd_mdsex1_eval_grad_f(h, dptr(x), dptr(g)) == 0
    assert L.hiopamd_mdsex1_eval_cons(h, dptr(x), dptr(c)) == 0
    ctx.sync()
    fw, gw, cw = op.mds_ex1_callbacks(ns, nd, xh, empty)
    assert abs(f.value - fw) <= 1e-13 * max(1.0, abs(fw))      # fp64, summation order differs
    np.testing.assert_allclose(g.cpu().numpy(), gw, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(c.cpu().numpy(), cw, rtol=1e-13, atol=1e-12)

    # Jacobian: pattern call (values NULL), then values call (pattern NULL), as the solver does
    ie, je = dev(p.Jcs_i.size, torch.int32), dev(p.Jcs_i.size, torch.int32)
    ii, ji = dev(p.Jds_i.size, torch.int32), dev(p.Jds_i.size, torch.int32)
    assert L.hiopamd_mdsex1_eval_Jac_cons_eq(h, dptr(x), dptr(ie), dptr(je), None, None) == 0
    assert L.hiopamd_mdsex1_eval_Jac_cons_ineq(h, dptr(x), 0, dptr(ii), dptr(ji), None, None) == 0
    ve, vi = dev(p.Jcs_i.size), dev(p.Jds_i.size)
    Jcd, Jdd = dev(ns * nd), dev(3 * nd)
    assert L.hiopamd_mdsex1_eval_Jac_cons_eq(h, dptr(x), None, None, dptr(ve), dptr(Jcd)) == 0
    assert L.hiopamd_mdsex1_eval_Jac_cons_ineq(h, dptr(x), 0, None, None, dptr(vi), dptr(Jdd)) == 0
    ctx.sync()
    np.testing.assert_array_equal(ie.cpu().numpy()[:p.Jcs_i.size], p.Jcs_i)
    np.testing.assert_array_equal(je.cpu().numpy()[:p.Jcs_i.size], p.Jcs_j)
    np.testing.assert_array_equal(ii.cpu().numpy()[:p.Jds_i.size], p.Jds_i)
    np.testing.assert_array_equal(ji.cpu().numpy()[:p.Jds_i.size], p.Jds_j)
    np.testing.assert_array_equal(ve.cpu().numpy()[:p.Jcs_i.size], p.Jcs_v)
    np.testing.assert_array_equal(vi.cpu().numpy()[:p.Jds_i.size], p.Jds_v)
    np.testing.assert_array_equal(Jcd.cpu().numpy()[:ns * nd].reshape(ns, nd), p.Jcd)
    np.testing.assert_array_equal(Jdd.cpu().numpy()[:3 * nd].reshape(3, nd), p.Jdd)

    # the one-call layout of the interface (all m rows in one set of arrays): equalities first, the inequalities behind them
    # with their rows shifted by ns
    nnz = p.Jcs_i.size + p.Jds_i.size
    i1, j1, v1, JD = dev(nnz, torch.int32), dev(nnz, torch.int32), dev(nnz), dev((ns + 3) * nd)
    off = p.Jcs_i.size
    assert L.hiopamd_mdsex1_eval_Jac_cons_eq(h, dptr(x), dptr(i1), dptr(j1), dptr(v1), dptr(JD)) == 0
    assert L.hiopamd_mdsex1_eval_Jac_cons_ineq(h, dptr(x), ns, C.c_void_p(i1.data_ptr() + 4 * off), C.c_void_p(j1.data_ptr() + 4 * off),
                                               C.c_void_p(v1.data_ptr() + 8 * off), C.c_void_p(JD.data_ptr() + 8 * ns * nd)) == 0
    ctx.sync()
    np.testing.assert_array_equal(i1.cpu().numpy()[:nnz], np.concatenate([p.Jcs_i, p.Jds_i + ns]))
    np.testing.assert_array_equal(j1.cpu().numpy()[:nnz], np.concatenate([p.Jcs_j, p.Jds_j]))
    np.testing.assert_array_equal(v1.cpu().numpy()[:nnz], np.ones(nnz))
    np.testing.assert_array_equal(JD.cpu().numpy()[:(ns + 3) * nd].reshape(ns + 3, nd), np.vstack([p.Jcd, p.Jdd]))

    # Hessian of the Lagrangian with obj_factor = 0.75 (lambda is ignored: linear constraints)
    ih, jh, vh, Hdd = dev(2 * ns, torch.int32), dev(2 * ns, torch.int32), dev(2 * ns), dev(nd * nd)
    lam = torch.as_tensor(r.uniform(-1, 1, m.value)).cuda()
    assert L.hiopamd_mdsex1_eval_Hess_Lagr(h, dptr(x), C.c_double(0.75), dptr(lam), dptr(ih), dptr(jh), dptr(vh), dptr(Hdd)) == 0
    ctx.sync()
    np.testing.assert_array_equal(ih.cpu().numpy()[:2 * ns], p.Hss_i)
    np.testing.assert_array_equal(jh.cpu().numpy()[:2 * ns], p.Hss_j)
    np.testing.assert_array_equal(vh.cpu().numpy()[:2 * ns], 0.75 * p.Hss_v)
    np.testing.assert_array_equal(Hdd.cpu().numpy()[:nd * nd].reshape(nd, nd), 0.75 * p.Hdd)
    assert L.hiopamd_mdsex1_destroy(h) == 0


@pytest.mark.parametrize("n", [4, 1000, 100003])
def test_dense_cons_ex2_callbacks_single_rank(ctx, n):
    """`hiopamd_denseex2_*` = the hiopInterfaceDenseConstraints callbacks of DenseConsEx2
    (src/Drivers/Dense/NlpDenseConsEx2.cpp) on device pointers, against dense_ex2 of hiop_amd/problems.py (re-exported by oracle/problems.py)."""
    L = ctx._L
    h = C.c_void_p()
    assert L.hiopamd_denseex2_create(C.byref(h), ctx.h, n, 0) == 0
    p = op.dense_ex2(n)
    nn, mm = C.c_int64(), C.c_int64()
    assert L.hiopamd_denseex2_get_prob_sizes(h, C.byref(nn), C.byref(mm)) == 0 and (nn.value, mm.value) == (n, 4)
    cols = (C.c_int64 * 2)()
    assert L.hiopamd_denseex2_get_vecdistrib_info(h, cols) == 0 and list(cols) == [0, n]
    cl, cu = (C.c_double * 4)(), (C.c_double * 4)()
    assert L.hiopamd_denseex2_get_cons_info(h, cl, cu) == 0
    np.testing.assert_array_equal(np.array(cl), np.concatenate([p["crhs"], p["dl"]]))
    np.testing.assert_array_equal(np.array(cu), np.concatenate([p["crhs"], p["du"]]))
    xl, xu, x0 = dev(n), dev(n), dev(n)
    assert L.hiopamd_denseex2_get_vars_info(h, dptr(xl), dptr(xu)) == 0
    assert L.hiopamd_denseex2_get_starting_point(h, dptr(x0)) == 0
    r = np.random.Generator(np.random.PCG64(n))
    xh = r.uniform(0.0, 3.0, n)
    x = torch.as_tensor(xh).cuda()
    f = C.c_double()
    g, c, J = dev(n), dev(4), dev(4 * n)
    assert L.hiopamd_denseex2_eval_f(h, dptr(x), C.byref(f)) == 0
    assert L.hiopamd_denseex2_eval_grad_f(h, dptr(x), dptr(g)) == 0
    assert L.hiopamd_denseex2_eval_cons(h, dptr(x), dptr(c)) == 0
    assert L.hiopamd_denseex2_eval_Jac_cons(h, dptr(x), dptr(J)) == 0
    ctx.sync()
    np.testing.assert_array_equal(xl.cpu().numpy(), p["xl"])
    np.testing.assert_array_equal(xu.cpu().numpy(), p["xu"])
    np.testing.assert_array_equal(x0.cpu().numpy(), p["x0"])
    Jw = np.vstack([p["Jc"], p["Jd"]])
    np.testing.assert_array_equal(J.cpu().numpy().reshape(4, n), Jw)
    assert abs(f.value - p["f"](xh)) <= 1e-12 * max(1.0, abs(p["f"](xh)))
    np.testing.assert_allclose(g.cpu().numpy(), p["grad"](xh), rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(c.cpu().numpy(), Jw @ xh, rtol=1e-12)
    assert L.hiopamd_denseex2_destroy(h) == 0


def test_dense_cons_ex2_local_part_of_a_three_rank_partition():
    """rank 1 of 3 (n = 10: columns [4, 7), none of the three special variables is local): partition, bounds, Jacobian
    columns and the LOCAL contributions to the objective / constraint bodies (a hook that does not reduce)."""
    from hiop_amd._lib import ALLREDUCE_FN
    from hiop_amd.runtime import Context
    c2 = Context(0)
    L = c2._L
    cb = ALLREDUCE_FN(lambda user, buf, count, op, stream: 0)
    assert L.hiopamd_ctx_set_allreduce(c2.h, cb, None, 1, 3) == 0
    n = 10
    h = C.c_void_p()
    assert L.hiopamd_denseex2_create(C.byref(h), c2.h, n, 0) == 0
    cols = (C.c_int64 * 4)()
    assert L.hiopamd_denseex2_get_vecdistrib_info(h, cols) == 0 and list(cols) == [0, 4, 7, 10]   # quotient 3, remainder 1
    p = op.dense_ex2(n)
    lo, hi = 4, 7
    xl, xu, J, cc = dev(3), dev(3), dev(12), dev(4)
    xh = np.array([0.7, 1.9, 2.4])
    x = torch.as_tensor(xh).cuda()
    f = C.c_double()
    assert L.hiopamd_denseex2_get_vars_info(h, dptr(xl), dptr(xu)) == 0
    assert L.hiopamd_denseex2_eval_Jac_cons(h, dptr(x), dptr(J)) == 0
    assert L.hiopamd_denseex2_eval_cons(h, dptr(x), dptr(cc)) == 0
    assert L.hiopamd_denseex2_eval_f(h, dptr(x), C.byref(f)) == 0
    c2.sync()
    np.testing.assert_array_equal(xl.cpu().numpy(), p["xl"][lo:hi])
    np.testing.assert_array_equal(xu.cpu().numpy(), p["xu"][lo:hi])
    Jw = np.vstack([p["Jc"], p["Jd"]])[:, lo:hi]
    np.testing.assert_array_equal(J.cpu().numpy().reshape(4, 3), Jw)
    np.testing.assert_allclose(cc.cpu().numpy(), Jw @ xh, rtol=1e-14)
    assert abs(f.value - 0.25 * np.sum((xh - 1.0) ** 4)) < 1e-14
    assert L.hiopamd_denseex2_destroy(h) == 0
    c2.close()


def test_callbacks_feed_the_condensed_kkt_without_host_values(ctx):
    """The purpose of row f4: the values the example's eval_Jac_cons / eval_Hess_Lagr write on the device go straight into
    hiopamd_kkt_mds_set_values; only the sparsity pattern (once) crosses to the host, for the symbolic plan.  The assembled
    condensed KKT matrix, its inertia and a solve must equal the oracle's, which is fed from the numpy restatement."""
    from hiop_amd.kkt import KKTLinSysCompressedMDSXYcYd
    from oracle import hiop_oracle as ho
    L = ctx._L
    ns, nd = 40, 12
    p = op.mds_ex1(ns, nd)
    h = C.c_void_p()
    assert L.hiopamd_mdsex1_create(C.byref(h), ctx.h, ns, nd, 0) == 0
    x = torch.ones(2 * ns + nd, dtype=torch.float64, device="cuda")
    lam = torch.zeros(ns + 3, dtype=torch.float64, device="cuda")
    ie, je, ii, ji = (dev(k, torch.int32) for k in (2 * ns, 2 * ns, ns + 3, ns + 3))
    ih, jh = dev(2 * ns, torch.int32), dev(2 * ns, torch.int32)
    assert L.hiopamd_mdsex1_eval_Jac_cons_eq(h, dptr(x), dptr(ie), dptr(je), None, None) == 0
    assert L.hiopamd_mdsex1_eval_Jac_cons_ineq(h, dptr(x), 0, dptr(ii), dptr(ji), None, None) == 0
    assert L.hiopamd_mdsex1_eval_Hess_Lagr(h, dptr(x), C.c_double(1.0), dptr(lam), dptr(ih), dptr(jh), None, None) == 0
    ctx.sync()
    pat = [t.cpu().numpy() for t in (ie, je, ii, ji, ih, jh)]
    kg = KKTLinSysCompressedMDSXYcYd(ctx, 2 * ns, nd, ns, 3, (pat[0], pat[1]), (pat[2], pat[3]), (pat[4], pat[5]))
    # values: device to device
    ve, vi, vh = dev(2 * ns), dev(ns + 3), dev(2 * ns)
    Jcd, Jdd, Hdd = dev(ns * nd), dev(3 * nd), dev(nd * nd)
    assert L.hiopamd_mdsex1_eval_Jac_cons_eq(h, dptr(x), None, None, dptr(ve), dptr(Jcd)) == 0
    assert L.hiopamd_mdsex1_eval_Jac_cons_ineq(h, dptr(x), 0, None, None, dptr(vi), dptr(Jdd)) == 0
    assert L.hiopamd_mdsex1_eval_Hess_Lagr(h, dptr(x), C.c_double(1.0), dptr(lam), None, None, dptr(vh), dptr(Hdd)) == 0
    Dx, Dd = op.barrier_diagonals(p, seed=3)
    Dxd, Ddd = torch.as_tensor(Dx).cuda(), torch.as_tensor(Dd).cuda()
    kg.set_values(ve, vi, vh, Jcd, Jdd, Hdd, Dxd, Ddd)
    ko = ho.KKTLinSysCompressedMDSXYcYd(p.nxs, p.nxd, p.neq, p.nineq, (p.Jcs_i, p.Jcs_j), (p.Jds_i, p.Jds_j), (p.Hss_i, p.Hss_j))
    ko.set_values(p.Jcs_v, p.Jds_v, p.Hss_v, p.Jcd, p.Jdd, p.Hdd, Dx, Dd)
    Mo = ko.build_kkt_matrix(0.0, 0.0, 0.0, 0.0).copy()
    kg.build_kkt_matrix(0.0, 0.0, 0.0, 0.0)
    np.testing.assert_allclose(np.triu(kg.sys_matrix().cpu().numpy()), np.triu(Mo), rtol=1e-13, atol=1e-13)
    assert kg.factorize_with_curv_check() == ko.factorize_with_curv_check() == ns + 3
    rx, ryc, ryd = op.random_rhs(p)
    ok, dx_o, dyc_o, dyd_o = ko.solve_compressed(rx, ryc, ryd)
    assert ok
    dx, dyc, dyd = dev(rx.size), dev(ryc.size), dev(ryd.size)
    rxd, rycd, rydd = torch.as_tensor(rx).cuda(), torch.as_tensor(ryc).cuda(), torch.as_tensor(ryd).cuda()
    torch.cuda.synchronize()
    kg.solve_compressed(rxd, rycd, rydd, dx, dyc, dyd)
    ctx.sync()
    scale = max(np.abs(dx_o).max(), np.abs(dyc_o).max(), np.abs(dyd_o).max())
    assert np.abs(dx.cpu().numpy()[:rx.size] - dx_o).max() / scale < 1e-8
    assert np.abs(dyc.cpu().numpy()[:ryc.size] - dyc_o).max() / scale < 1e-8
    assert np.abs(dyd.cpu().numpy()[:ryd.size] - dyd_o).max() / scale < 1e-8
    kg.close()
    assert L.hiopamd_mdsex1_destroy(h) == 0
