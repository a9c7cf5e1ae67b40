"""The C interface of the MDS solver (include/hiop_amd_interface.h; reference: src/Interface/hiopInterface.h:63-98,
chiopInterface.cpp:64-95) exercised the way the reference exercises its own: a plain C program with a struct of callbacks
(tests/c/mds_c_interface.c, the counterpart of src/Drivers/MDS/NlpMdsEx1.c), compiled with gcc against the public headers and
linked to libhiopamd.so.

CPU: the headers compile as C11 with -Wall -Wextra, the three entry points and the additions resolve.
GPU: the program solves MdsEx1(400, 100) with HOST callbacks and passes the reference driver's own check (objective
-4.999509728895e+01 to 1e-6, NlpMdsEx1.c:376); the run equals the numpy run of the same loop (oracle/ipm_filter.py — itself pinned
on the reference's KKT dumps): same iteration count, objective to 1e-9.  With DEVICE callbacks (the library's device-resident
example problem) the same holds against the C++ example's data."""
import os
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "c" / "mds_c_interface.c"


def _compile(tmp_path, src=None):
    from hiop_amd.build import build
    lib = build()
    src = SRC if src is None else src
    exe = tmp_path / src.stem
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", f"-I{ROOT / 'include'}", str(src), "-o", str(exe),
           f"-L{lib.parent}", "-lhiopamd", "-lm", f"-Wl,-rpath,{lib.parent}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_program_compiles_and_links_against_the_public_headers(tmp_path):
    exe = _compile(tmp_path)
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True).stdout
    for s in ("hiop_mds_create_problem", "hiop_mds_solve_problem", "hiop_mds_destroy_problem", "hiopamd_mds_set_callback_mem_space",
              "hiopamd_mds_get_solve_info"):
        assert s in syms


def test_struct_layout_is_the_reference_s():
    """Member order of cHiopMDSProblem (hiopInterface.h:63-95): two library pointers, user_data, solution, obj_value, then the ten
    callbacks in the reference's order — a binding built against the reference's header must find every member where it expects it."""
    txt = (ROOT / "include" / "hiop_amd_interface.h").read_text()
    body = txt[txt.index("typedef struct cHiopMDSProblem {"):txt.index("} cHiopMDSProblem;")]
    names = re.findall(r"(?:\(\*|\s\*?)(refcppHiop|hiopinterface|user_data|solution|obj_value|get_starting_point|get_prob_sizes|"
                       r"get_vars_info|get_cons_info|eval_f|eval_grad_f|eval_cons|get_sparse_dense_blocks_info|eval_Jac_cons|"
                       r"eval_Hess_Lagr)\)?[;(]", body)
    names = list(dict.fromkeys(names))          # (`user_data` is also the last parameter of every callback)
    assert names == ["refcppHiop", "hiopinterface", "user_data", "solution", "obj_value", "get_starting_point", "get_prob_sizes",
                     "get_vars_info", "get_cons_info", "eval_f", "eval_grad_f", "eval_cons", "get_sparse_dense_blocks_info",
                     "eval_Jac_cons", "eval_Hess_Lagr"]


def _oracle_run(ns, nd, c_driver_q, tolerance=1e-8):
    from oracle import ipm_filter
    from tests.test_oracle_reference_trajectory import reference_setup
    p, k, full, bounds, model, q = reference_setup(ns, nd)
    if c_driver_q:      # NlpMdsEx1.c:311-316 ADDS 1 to the 1e-8 background on the off-diagonals; the C++ class sets them to 1
        Q = p.Hdd.copy()
        for i in range(1, nd - 1):
            Q[i, i + 1] = Q[i + 1, i] = 1.0 + 1e-8
        p.Hdd = Q
        p, k, full, bounds, model, q = reference_setup(ns, nd, p)
    return ipm_filter.solve(ipm_filter.FilterOracleOps(full, bounds, model), np.ones(2 * ns + nd), mu0=0.1, tolerance=tolerance)


def test_oracle_passes_the_c_driver_check():
    """The numpy run at the C interface's options (mu0 = 0.1, everything else default: tolerance 1e-8) against the objective the
    reference's C driver requires (NlpMdsEx1.c:376, 1e-6)."""
    r = _oracle_run(400, 100, True)
    assert r["status"] == "Solve_Success" and abs(r["obj"] - (-4.999509728895e+01)) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["host", "device"])
def test_c_program_solves_mds_ex1_like_the_reference_driver(tmp_path, mode):
    exe = _compile(tmp_path)
    r = subprocess.run([str(exe), mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = [l for l in r.stdout.strip().splitlines() if l.startswith("obj=")][-1]
    g = re.match(r"obj=(\S+) iters=(\d+) status=(-?\d+) nfact=(\d+) xsum=(\S+) rc=(-?\d+)", last)
    assert g, last
    obj, iters, status = float(g.group(1)), int(g.group(2)), int(g.group(3))
    assert status == 0 and abs(obj - (-4.999509728895e+01)) <= 1e-6      # the reference driver's own acceptance check
    o = _oracle_run(400, 100, mode == "host")
    assert iters == o["iters"] and abs(obj - o["obj"]) <= 1e-9
    assert abs(float(g.group(5)) - float(o["x"].sum())) <= 1e-6 * max(1.0, abs(o["x"].sum()))
    # the iteration table it printed is the reference's format (header every 10 iterations, one line per iteration)
    lines = [l for l in r.stdout.splitlines() if re.match(r"^\s*\d+\s+-?\d\.\d{7}e", l)]
    assert len(lines) == iters + 1 and r.stdout.count("iter    objective") == iters // 10 + 1


DENSE_SRC = ROOT / "tests" / "c" / "dense_c_interface.c"


def test_dense_c_program_compiles_and_links(tmp_path):
    exe = _compile(tmp_path, DENSE_SRC)
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True).stdout
    for s in ("hiop_dense_create_problem", "hiop_dense_solve_problem", "hiop_dense_destroy_problem"):
        assert s in syms


@pytest.mark.gpu
@pytest.mark.parametrize("n", [500, 5000])
def test_dense_c_program_solves_dense_cons_ex2_with_the_quasi_newton_path(tmp_path, n):
    """hiop_dense_create/solve/destroy_problem (hiopInterface.h:150-176) on DenseConsEx2 written as C callbacks: the quasi-Newton
    filter IPM of the reference's dense C interface (chiopInterface.cpp:129-159: secant Hessian, linear duals, zero initial duals)
    on the device's low-rank KKT path.  The run must end at the problem's optimum 1/64 (x_3 = 1.5, the rest 1; bounds relaxed by 1e-8)
    and equal the numpy run of the same loop: same iteration count (+-1: the secant recursion amplifies rounding late in the run),
    objective to 1e-9."""
    from oracle import ipm_filter
    from oracle import problems as pr
    from tests.test_oracle_reference_trajectory import quasi_newton_setup
    exe = _compile(tmp_path, DENSE_SRC)
    r = subprocess.run([str(exe), str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    g = re.match(r"obj=(\S+) iters=(\d+) status=(-?\d+) maxdev=(\S+) rc=(-?\d+)", r.stdout.strip().splitlines()[-1])
    assert g, r.stdout[-500:]
    obj, iters, status, maxdev = float(g.group(1)), int(g.group(2)), int(g.group(3)), float(g.group(4))
    assert status == 0 and abs(obj - 1.0 / 64) <= 2e-8 and maxdev <= 1e-2
    q = pr.dense_ex2(n)
    ops, full, bounds = quasi_newton_setup(q)
    o = ipm_filter.solve(ops, q["x0"], quasi_newton=True, lsq_duals=False)
    assert o["status"] == "Solve_Success"
    assert abs(iters - o["iters"]) <= 1 and abs(obj - o["obj"]) <= 1e-9


@pytest.mark.gpu
def test_dense_c_program_with_a_fixed_variable(tmp_path):
    """xlow == xupp on one variable: the reference's dense C interface runs with fixed_var = relax (chiopInterface.cpp:138), i.e. the
    bounds relaxer opens the variable by bound_relax_perturb * max(1, |x|) on both sides (hiopNlpFormulation.cpp:342-347, 398-402).  The
    device run must equal the numpy run of the same loop with the same bounds."""
    from oracle import ipm_filter
    from oracle import problems as pr
    from tests.test_oracle_reference_trajectory import quasi_newton_setup
    n = 500
    exe = _compile(tmp_path, DENSE_SRC)
    r = subprocess.run([str(exe), str(n)], capture_output=True, text=True, timeout=600, env=dict(os.environ, DENSE_FIX_LAST="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    g = re.match(r"obj=(\S+) iters=(\d+) status=(-?\d+) maxdev=(\S+) rc=(-?\d+)", r.stdout.strip().splitlines()[-1])
    assert g, r.stdout[-500:]
    obj, iters, status, maxdev = float(g.group(1)), int(g.group(2)), int(g.group(3)), float(g.group(4))
    q = pr.dense_ex2(n)
    q["xl"] = q["xl"].copy(); q["xu"] = q["xu"].copy()
    q["xl"][n - 1] = q["xu"][n - 1] = 1.0
    ops, full, bounds = quasi_newton_setup(q)
    o = ipm_filter.solve(ops, q["x0"], quasi_newton=True, lsq_duals=False)
    assert o["status"] == "Solve_Success" and status == 0 and maxdev <= 1e-2
    assert abs(float(o["x"][n - 1]) - 1.0) <= 2e-8
    assert abs(iters - o["iters"]) <= 1 and abs(obj - o["obj"]) <= 1e-9


def _dense_user_problem(n, kobj, krow):
    """DenseConsEx2 as the C driver presents it with DENSE_KOBJ / DENSE_KROW: objective x kobj, constraint row 2 (the second inequality)
    x krow, body and bounds."""
    from oracle import problems as pr
    q = pr.dense_ex2(n)
    u = dict(q)
    f0, g0 = q["f"], q["grad"]
    u["f"] = lambda x: kobj * f0(x)
    u["grad"] = lambda x: kobj * g0(x)
    u["Jd"] = q["Jd"].copy(); u["Jd"][1] *= krow
    u["dl"] = q["dl"].copy(); u["dl"][1] *= krow
    u["du"] = q["du"].copy(); u["du"][1] *= krow
    return u


def _oracle_run_with_reference_scaling(u):
    """The reference's order of operations: bounds relaxed in the user's space (hiopNlpFormulation.cpp:398-402), scaling decided at the
    user's starting point and applied to the functions AND to the (relaxed) constraint bounds (:671-714), the loop on the scaled problem,
    the objective handed back unscaled."""
    from oracle import hiop_oracle as ho
    from oracle import ipm_filter
    from oracle import kkt_full as kf
    from oracle.nlp_scaling import gradient_scaling
    n = u["n"]
    s = gradient_scaling(u["grad"](u["x0"]), u["Jc"], u["Jd"])
    assert s is not None
    s_f, s_c, s_d = s
    xl, xu, dl, du = ipm_filter.relax_bounds(u["xl"], u["xu"], u["dl"], u["du"], ipm_filter.DEFAULTS["bound_relax_perturb"])
    Jc, Jd = s_c[:, None] * u["Jc"], s_d[:, None] * u["Jd"]
    bounds = (xl, xu, s_d * dl, s_d * du, s_c * u["crhs"])
    f64 = lambda b: b.astype(np.float64)
    ixl, ixu, idl, idu = f64(u["xl"] > -1e20), f64(u["xu"] < 1e20), f64(u["dl"] > -1e20), f64(u["du"] < 1e20)
    H = ho.HessianLowRank(n, l_max=6, sigma0=1.0, sigma_update_strategy="sty")
    K = ho.KKTLinSysLowRank(H, Jc.shape[0], Jd.shape[0])
    full = kf.KKTLinSysFull(kf.LowRankProvider(K, Jc, Jd), ixl, ixu, idl, idu, perturb=kf.PDPerturbationNull())
    model = lambda x: (s_f * u["f"](x), s_f * u["grad"](x), Jc @ x, Jd @ x)

    class Ops(ipm_filter.FilterOracleOps):
        def hess_update(self, it, ev):
            H.update(it["x"], ev[1], Jc, Jd, it["yc"], it["yd"])
    table = []
    o = ipm_filter.solve(Ops(full, bounds, model), u["x0"], quasi_newton=True, lsq_duals=False, table=table)
    o["obj_user"] = o["obj"] / s_f
    o["table_obj_user"] = [row["objective"] / s_f for row in table]
    return o, s


def test_oracle_scaling_of_a_badly_scaled_dense_problem():
    """The restated apply_scaling on the badly scaled variant: objective factor 100 / max|grad f(x0)|, only the offending row scaled, and
    the scaled run ends at the same minimiser (objective = kobj / 64)."""
    u = _dense_user_problem(200, 1.0e4, 1.0e3)
    o, (s_f, s_c, s_d) = _oracle_run_with_reference_scaling(u)
    g0 = np.abs(u["grad"](u["x0"])).max()
    assert s_f == pytest.approx(100.0 / g0) and np.all(s_c == 1.0)
    assert s_d[1] == pytest.approx(100.0 / np.abs(u["Jd"][1]).max()) and s_d[0] == 1.0 and s_d[2] == 1.0
    assert o["status"] == "Solve_Success" and abs(o["obj_user"] - 1.0e4 / 64) <= 1e-3


@pytest.mark.gpu
def test_dense_c_program_on_a_badly_scaled_problem(tmp_path):
    """Gradient-based NLP scaling (the reference's default scaling_type = gradient; hiopNlpFormulation.cpp:671-714,
    hiopNlpTransforms.cpp:423-499) behind hiop_dense_solve_problem: objective x 1e4 and one constraint row x 1e3 in the user's
    callbacks.  The device run must equal the numpy run of the restated scaling + loop: iterations (+-1), reported (unscaled) objective."""
    n = 500
    exe = _compile(tmp_path, DENSE_SRC)
    r = subprocess.run([str(exe), str(n)], capture_output=True, text=True, timeout=600, env=dict(os.environ, DENSE_KOBJ="1e4", DENSE_KROW="1e3"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "gradient-based scaling on" in r.stdout
    g = re.match(r"obj=(\S+) iters=(\d+) status=(-?\d+) maxdev=(\S+) rc=(-?\d+)", r.stdout.strip().splitlines()[-1])
    assert g, r.stdout[-500:]
    obj, iters, status, maxdev = float(g.group(1)), int(g.group(2)), int(g.group(3)), float(g.group(4))
    o, _ = _oracle_run_with_reference_scaling(_dense_user_problem(n, 1.0e4, 1.0e3))
    # The row scaled by 1e3 puts the constraint residuals' rounding floor at ~1e-10, and the secant recursion amplifies rounding: late in
    # the run the two implementations part (one may stop at the acceptable level, 10 iterations below 1e-6, where the other goes on to
    # 1e-8).  Required: the same trajectory while it is well above that floor, the same optimum.
    assert o["status"] in ("Solve_Success", "Solve_Acceptable_Level") and status in (0, 2) and maxdev <= 1e-2
    assert abs(obj - o["obj_user"]) <= 1e-9 * abs(o["obj_user"])
    dev_obj = [float(l.split()[1]) for l in r.stdout.splitlines() if re.match(r"^\s*\d+\s+-?\d\.\d{7}e", l)]
    assert len(dev_obj) == iters + 1
    for i in range(20):
        assert abs(dev_obj[i] - o["table_obj_user"][i]) <= 2e-7 * abs(o["table_obj_user"][i]), i   # (the table prints 8 digits)


def _mds_oracle_run_with_reference_scaling(ns, nd, kobj, krow):
    """MdsEx1 as the C driver presents it with MDS_KOBJ / MDS_KROW (objective x kobj; the first inequality x krow), then the
    reference's scaling (decided at x0 = 1, hiopNlpFormulation.cpp:671-714) applied to the problem data — the constraints are linear and
    the Hessian comes from the objective alone, so scaling the data IS evaluating the scaled functions —, bounds relaxed BEFORE they are
    scaled, the Newton loop at the C interface's options, the objective handed back unscaled."""
    import copy
    from oracle import hiop_oracle as ho
    from oracle import ipm_filter, ipm_full
    from oracle import kkt_full as kf
    from oracle import problems as pr
    from oracle.nlp_scaling import gradient_scaling
    p = pr.mds_ex1(ns, nd)
    Q = p.Hdd.copy()
    for i in range(1, nd - 1):          # the C driver's Q (see _oracle_run)
        Q[i, i + 1] = Q[i + 1, i] = 1.0 + 1e-8
    p.Hdd = Q
    n = p.nxs + p.nxd
    q = np.zeros(n); q[:p.nxs // 2] = -0.5
    # the user's badly scaled problem
    u = copy.copy(p)
    u.Hss_v, u.Hdd, u.q_lin = kobj * p.Hss_v, kobj * p.Hdd, kobj * q
    u.Jds_v = np.where(p.Jds_i == 0, krow, 1.0) * p.Jds_v
    u.Jdd = p.Jdd.copy(); u.Jdd[0] *= krow
    u.dl = p.dl.copy(); u.du = p.du.copy(); u.dl[0] *= krow; u.du[0] *= krow
    x0 = np.ones(n)
    model_u, _ = ipm_full.mds_model(u)
    dense = lambda I, J, V, D, m: np.concatenate([np.asarray(__import__("scipy.sparse").sparse.csr_matrix((V, (I, J)), shape=(m, u.nxs)).todense()), D], axis=1)
    s = gradient_scaling(model_u(x0)[1], dense(u.Jcs_i, u.Jcs_j, u.Jcs_v, u.Jcd, u.neq), dense(u.Jds_i, u.Jds_j, u.Jds_v, u.Jdd, u.nineq))
    assert s is not None
    s_f, s_c, s_d = s
    w = copy.copy(u)
    w.Hss_v, w.Hdd, w.q_lin = s_f * u.Hss_v, s_f * u.Hdd, s_f * u.q_lin
    w.Jcs_v, w.Jcd = s_c[u.Jcs_i] * u.Jcs_v, s_c[:, None] * u.Jcd
    w.Jds_v, w.Jdd = s_d[u.Jds_i] * u.Jds_v, s_d[:, None] * u.Jdd
    k = ho.KKTLinSysCompressedMDSXYcYd(w.nxs, w.nxd, w.neq, w.nineq, (w.Jcs_i, w.Jcs_j), (w.Jds_i, w.Jds_j), (w.Hss_i, w.Hss_j))
    k.set_values(w.Jcs_v, w.Jds_v, w.Hss_v, w.Jcd, w.Jdd, w.Hdd, None, None)
    f64 = lambda b: b.astype(np.float64)
    full = kf.KKTLinSysFull(kf.MdsProvider(k), f64(u.xl > -1e20), f64(u.xu < 1e20), f64(u.dl > -1e20), f64(u.du < 1e20))
    xl, xu, dl, du = ipm_filter.relax_bounds(u.xl, u.xu, u.dl, u.du, ipm_filter.DEFAULTS["bound_relax_perturb"])
    bounds = (xl, xu, s_d * dl, s_d * du, np.zeros(u.neq))
    model, _ = ipm_full.mds_model(w)
    o = ipm_filter.solve(ipm_filter.FilterOracleOps(full, bounds, model), x0, mu0=0.1, tolerance=1e-8)
    o["obj_user"] = o["obj"] / s_f
    return o, s


def test_oracle_scaling_of_a_badly_scaled_mds_problem():
    o, (s_f, s_c, s_d) = _mds_oracle_run_with_reference_scaling(40, 12, 1.0e4, 1.0e3)
    assert s_f < 1.0 and np.all(s_c == 1.0) and s_d[0] == pytest.approx(0.1) and np.all(s_d[1:] == 1.0)
    r = _oracle_run(40, 12, True)
    assert o["status"] == "Solve_Success" and abs(o["obj_user"] - 1.0e4 * r["obj"]) <= 1e-5 * abs(1.0e4 * r["obj"])


@pytest.mark.gpu
def test_mds_c_program_on_a_badly_scaled_problem(tmp_path):
    """Gradient-based NLP scaling behind hiop_mds_solve_problem (host callbacks): objective x 1e4 (the user's Hessian gets obj_factor =
    the objective's scale factor), the first inequality x 1e3 (sparse and dense Jacobian entries, bounds).  Equal to the numpy run of the
    restated scaling + Newton loop: iterations, the reported (unscaled) objective to 1e-9 relative."""
    exe = _compile(tmp_path)
    r = subprocess.run([str(exe), "host"], capture_output=True, text=True, timeout=600, env=dict(os.environ, MDS_KOBJ="1e4", MDS_KROW="1e3"))
    last = [l for l in r.stdout.strip().splitlines() if l.startswith("obj=")][-1]
    g = re.match(r"obj=(\S+) iters=(\d+) status=(-?\d+) nfact=(\d+) xsum=(\S+) rc=(-?\d+)", last)
    assert g, r.stdout[-1500:] + r.stderr[-1500:]
    assert "gradient-based scaling on" in r.stdout
    obj, iters, status = float(g.group(1)), int(g.group(2)), int(g.group(3))
    o, _ = _mds_oracle_run_with_reference_scaling(400, 100, 1.0e4, 1.0e3)
    assert status == 0 and o["status"] == "Solve_Success"
    assert iters == o["iters"] and abs(obj - o["obj_user"]) <= 1e-9 * abs(o["obj_user"])
    assert abs(float(g.group(5)) - float(o["x"].sum())) <= 1e-6 * max(1.0, abs(o["x"].sum()))


def test_gradient_scaling_factors_follow_the_reference_formulas():
    """oracle/nlp_scaling.py against hand-evaluated cases of hiopNlpFormulation::apply_scaling (hiopNlpFormulation.cpp:671-714) and
    hiopNLPObjGradScaling's constructor (hiopNlpTransforms.cpp:423-499) at the default options (scaling_max_grad 100, scaling_min_grad
    1e-8, scaling_max_obj_grad = scaling_max_con_grad = 0)."""
    from oracle.nlp_scaling import gradient_scaling
    g = np.array([3.0, -99.9])
    Jc = np.array([[1.0, -2.0]])
    Jd = np.array([[0.5, 99.0], [1.0, 1.0]])
    assert gradient_scaling(g, Jc, Jd) is None                                  # every entry below 100: apply_scaling returns false
    s_f, s_c, s_d = gradient_scaling(np.array([3.0, -100.0]), Jc, Jd)           # ">= max_grad" triggers, but 100 is not "> max_grad":
    assert s_f == 1.0 and np.all(s_c == 1.0) and np.all(s_d == 1.0)             # the transformation exists with unit factors
    s_f, s_c, s_d = gradient_scaling(np.array([3.0, -400.0]), Jc, np.array([[0.5, 250.0], [1.0, 1.0], [-1000.0, 2.0]]))
    assert s_f == 0.25 and np.all(s_c == 1.0)                                   # 100 / 400; the equality block has no row above 100
    np.testing.assert_allclose(s_d, [0.4, 1.0, 0.1], rtol=0, atol=0)            # 1 / max(1, rowmax / 100), row by row
    s_f, _, s_d = gradient_scaling(np.array([1e12]), np.zeros((0, 1)), np.array([[1e11]]))
    assert s_f == 1e-8 and s_d[0] == 1e-8                                       # the scaling_min_grad floor


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["mds-host", "mds-device", "dense"])
def test_a_problem_object_can_be_solved_twice(tmp_path, which):
    """chiopInterface.cpp:79-87 / :141-150: every hiop_*_solve_problem builds a fresh solver over the same problem object, so a binding may
    call it twice and gets the same answer twice.  Here: the second call on the same object starts over from the user's data and must
    return rc 0 with the same iteration count, the bit-identical objective and solution (the device path is deterministic)."""
    env = dict(os.environ, HIOPAMD_TEST_RESOLVE="1")
    if which == "dense":
        exe = _compile(tmp_path, DENSE_SRC)
        r = subprocess.run([str(exe), "500"], capture_output=True, text=True, timeout=600, env=env)
    else:
        exe = _compile(tmp_path)
        r = subprocess.run([str(exe), which.split("-")[1]], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("resolve:")]
    assert len(line) == 1, r.stdout[-2000:]
    assert "rc=0" in line[0] and "same_obj=1" in line[0] and "same_iters=1" in line[0] and "same_x=1" in line[0], line[0]
    assert "already solved" not in r.stderr
