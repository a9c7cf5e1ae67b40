"""Whole interior-point solves with EVERY per-iteration operation on the device (SURVEY §8 f1 + the a-rows):
objective / gradient / constraint bodies from the MDS matrices in HBM, hiopResidual::update, XYcYd::update with the
inertia-correction loop, compute_directions_w_IR (BiCGStab over the 12-part slab), fraction to the boundary, primal/dual
step, safe slacks, dual safeguard — driven by the same oracle/ipm_full.py loop that runs the numpy restatements.
The GPU run must follow the CPU run iteration by iteration and end at the reference driver's stored selfcheck objective."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import ipm_full
from oracle import kkt_full as kf
from oracle import problems as pr
from tests.test_gpu_kkt_xycyd import D
from tests.test_oracle_selfcheck import _full_layer_setup

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / "golden" / "selfcheck_objectives.json").read_text())


class DeviceOps:
    def __init__(self, ctx, p, full_o, bounds, q):
        from hiop_amd.kkt import IpmSlabOps, KKTLinSysXYcYd, mds_from_problem
        self.ctx, self.p = ctx, p
        self.kg, self._keep = mds_from_problem(ctx, p)
        d = self._keep
        self.kg.set_values(d["Jcs_v"], d["Jds_v"], d["Hss_v"], d["Jcd"], d["Jdd"], d["Hdd"], None, None)
        self.fg = KKTLinSysXYcYd(ctx, self.kg, D(full_o.ixl), D(full_o.ixu), D(full_o.idl), D(full_o.idu))
        self.ops = IpmSlabOps(self.fg, *[D(b) for b in bounds])
        self.q = D(q)
        self.nx, self.neq, self.nineq = p.nxs + p.nxd, p.neq, p.nineq
        z = lambda n: torch.zeros(n, dtype=torch.float64, device="cuda")
        self.hx, self.grad, self.c, self.d = z(self.nx), z(self.nx), z(self.neq), z(self.nineq)
        torch.cuda.synchronize()

    def from_host(self, it):
        return self.fg.pack(it, kf.ITER_PARTS)

    def primal(self, it):
        return it[:self.nx].cpu().numpy()

    def evaluate(self, it):
        L, kg, ctx = self.fg._L, self.kg, self.ctx
        x = it[:self.nx]
        ptr = lambda t: C.c_void_p(t.data_ptr())
        assert L.hiopamd_kkt_mds_hess_times_vec(kg.h, 0.0, ptr(self.hx), 1.0, ptr(x)) == 0
        ctx.call("hiopamd_vec_copy", self.nx, self.grad, self.hx)
        ctx.call("hiopamd_vec_axpy", self.nx, self.grad, 1.0, self.q)
        assert L.hiopamd_kkt_mds_jac_times_vec(kg.h, 0, 0.0, ptr(self.c), 1.0, ptr(x)) == 0
        assert L.hiopamd_kkt_mds_jac_times_vec(kg.h, 1, 0.0, ptr(self.d), 1.0, ptr(x)) == 0
        f = 0.5 * ctx.reduce_double("hiopamd_vec_dot", self.nx, x, self.hx) + ctx.reduce_double("hiopamd_vec_dot", self.nx, x, self.q)
        return f, self.grad, self.c, self.d

    def residual(self, it, ev, mu, kappa_d):
        resid = torch.empty_like(it)
        torch.cuda.synchronize()
        n = self.ops.residual_update(it, ev[2], ev[3], ev[1], mu, kappa_d, resid)
        return resid, n

    def kkt_update(self, it, mu):
        self.fg.set_mu(mu)
        return self.fg.update(it)

    def directions(self, resid):
        d = torch.empty_like(resid)
        torch.cuda.synchronize()
        ok, info = self.fg.compute_directions_w_IR(resid, d)
        return ok, d

    def fraction_to_the_bdry(self, it, d, tau):
        return self.ops.fraction_to_the_bdry(it, d, tau)

    def step(self, it, d, ap, ad, mu):
        trial = it.clone()
        torch.cuda.synchronize()
        self.ops.take_step(trial, it, d, ap, ad)
        self.ops.determine_slacks(trial)
        nadj = self.ops.adjust_small_slacks(trial, it, mu)
        self.ops.adjust_duals_plh(trial, mu, 1e10)
        self.ctx.sync()
        return trial, nadj

    def n_refactorizations(self):
        return self.fg.num_refact


@pytest.mark.parametrize("ns,nd", [(40, 12), (400, 100)])
def test_device_ipm_follows_the_oracle_and_reaches_the_selfcheck_objective(ctx, ns, nd):
    p = pr.mds_ex1(ns, nd)
    full, bounds, model, q = _full_layer_setup(p)
    mu0, tol = 0.1, 1e-5
    it0 = ipm_full.initial_iterate(full, bounds, p.x0, lambda x: model(x)[3], mu0)
    t_cpu, t_gpu = [], []
    r_cpu = ipm_full.solve(ipm_full.OracleOps(full, bounds, model), it0, mu0=mu0, tol=tol, trace=t_cpu)
    dev = DeviceOps(ctx, p, full, bounds, q)
    r_gpu = ipm_full.solve(dev, it0, mu0=mu0, tol=tol, trace=t_gpu)
    assert r_gpu["iters"] == r_cpu["iters"] and r_gpu["n_fact"] == r_cpu["n_fact"]
    a, b = np.array(t_cpu), np.array(t_gpu)
    np.testing.assert_allclose(b[:, 2], a[:, 2], rtol=0, atol=0)            # identical barrier schedule
    np.testing.assert_allclose(b[:, 0], a[:, 0], rtol=1e-7, atol=1e-9)     # objective per iteration
    np.testing.assert_allclose(b[:, 1], a[:, 1], rtol=1e-4, atol=1e-10)    # NLP error per iteration
    np.testing.assert_allclose(r_gpu["x"], r_cpu["x"], rtol=0, atol=1e-7 * max(1.0, np.abs(r_cpu["x"]).max()))
    if (ns, nd) == (400, 100):
        assert abs(r_gpu["obj"] - GOLD["MdsEx1"]["objective"]) < 2e-4
