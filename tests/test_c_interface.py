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
