#!/usr/bin/env python3
"""Writes tests/golden/iteration_table_mds_ex1_{40_12,400_100}.txt: the iteration table (format of the reference's
`outputIteration`, src/Optimization/hiopAlgFilterIPM.cpp:2783-2812, emitted through hiopamd_io_format_iteration) of the
ORACLE's restatement of the reference's Newton filter IPM (oracle/ipm_filter.py on the numpy restatements) on MdsEx1 with
the driver's options (NlpMdsEx1Driver.cpp:130-139: mu0 = 0.1, tolerance 1e-5, duals_init zero, linear duals update).  That
run follows the reference's own trajectory (tests/test_oracle_reference_trajectory.py: KKT systems of iterations 0, 5, 10
equal to the reference's dumps to 1e-12, 14 iterations and the stored -selfcheck objective on (400, 100)), so these tables
are the reference's iteration tables up to rounding of the printed digits.  The HIP run of the same loop must reproduce
these lines under the reference's own CPU-vs-GPU rule (tests/testMDS1CompareIterations.awk:13-40: every numeric column
within 1e-5 absolute, same line-search tag)."""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def table_lines(table):
    from hiop_amd._lib import lib
    L = lib()
    buf = C.create_string_buffer(256)
    L.hiopamd_io_iteration_header(buf, 256)
    out = [buf.value.decode()]
    for r in table:
        first = r["iter"] == 0
        L.hiopamd_io_format_iteration(buf, 256, 0, r["iter"], C.c_double(r["objective"]), C.c_double(r["inf_pr"]),
                                      C.c_double(r["inf_du"]), C.c_double(r["mu"]), C.c_double(r["alpha_du"]),
                                      C.c_double(r["alpha_pr"]), r.get("ls", -1 if first else 1),
                                      r.get("ls_num", 0 if first else 1), 0, 0)
        out.append(buf.value.decode())
    return out


def oracle_table(ns, nd):
    from oracle import ipm_filter
    from test_oracle_reference_trajectory import DRIVER_OPTIONS, reference_setup
    p, k, full, bounds, model, q = reference_setup(ns, nd)
    table = []
    ipm_filter.solve(ipm_filter.FilterOracleOps(full, bounds, model), p.x0, table=table, **DRIVER_OPTIONS)
    return table


if __name__ == "__main__":
    for ns, nd in ((40, 12), (400, 100)):
        lines = table_lines(oracle_table(ns, nd))
        path = os.path.join(HERE, f"iteration_table_mds_ex1_{ns}_{nd}.txt")
        with open(path, "w") as f:
            f.writelines(lines)
        print(path, len(lines) - 1, "iterations")
