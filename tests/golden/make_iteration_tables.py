#!/usr/bin/env python3
"""Writes tests/golden/iteration_table_mds_ex1_{40_12,400_100}.txt: the iteration table (format of the reference's
`outputIteration`, src/Optimization/hiopAlgFilterIPM.cpp:2783-2812, emitted through hiopamd_io_format_iteration) of the
ORACLE's full-space IPM (oracle/ipm_full.py on the numpy restatements) on MdsEx1 at the driver's settings (mu0 = 0.1,
tolerance 1e-5).  The HIP run of the same loop must reproduce these lines under the reference's own CPU-vs-GPU rule
(tests/testMDS1CompareIterations.awk:13-40: every numeric column within 1e-5 absolute, same line-search tag)."""
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
                                      C.c_double(r["alpha_pr"]), -1 if first else 1, 0 if first else 1, 0, 0)
        out.append(buf.value.decode())
    return out


def oracle_table(ns, nd):
    from oracle import ipm_full
    from oracle import problems as pr
    from test_oracle_selfcheck import _full_layer_setup
    p = pr.mds_ex1(ns, nd)
    full, bounds, model, q = _full_layer_setup(p)
    it0 = ipm_full.initial_iterate(full, bounds, p.x0, lambda x: model(x)[3], 0.1)
    table = []
    ipm_full.solve(ipm_full.OracleOps(full, bounds, model), it0, mu0=0.1, tol=1e-5, table=table)
    return table


if __name__ == "__main__":
    for ns, nd in ((40, 12), (400, 100)):
        lines = table_lines(oracle_table(ns, nd))
        path = os.path.join(HERE, f"iteration_table_mds_ex1_{ns}_{nd}.txt")
        with open(path, "w") as f:
            f.writelines(lines)
        print(path, len(lines) - 1, "iterations")
