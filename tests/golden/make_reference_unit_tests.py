#!/usr/bin/env python3
"""Transcribes the closed-form known answers of the reference's LinAlg unit tests into
tests/golden/reference_unit_tests.json (DATA: constant inputs + expected values; no reference code).

Sources (LLNL/hiop v1.1.0): tests/LinAlg/vectorTests.hpp, matrixTestsDense.hpp, matrixTestsSparse.hpp,
matrixTestsSymSparse.hpp; driver sizes tests/testVector.cpp:235 (Nlocal = 1000), tests/testMatrixDense.cpp:168-170
(M = 50, K = 100, N = 500 -> scaled to M = 10, N = 100 here), tests/testMatrixSparse.cpp:85-103 (M = 5, N = 50,
5 entries per row at columns 0, 10, 20, 30, 49 — matrixTestsSparseTriplet.cpp:303-330).
Constants of tests/LinAlg/testBase.hpp: zero, quarter, half, one, two, three.

Array encoding: {"n": N, "fill": c, "set": [[idx, val], ...]} (negative idx counts from the end); matrices
{"m": M, "n": N, "fill": c, "set": [[i, j, val], ...]}.
"""
import json
import math
import os

N = 1000
zero, quarter, half, one, two, three = 0.0, 0.25, 0.5, 1.0, 2.0, 3.0


def V(fill, sets=(), n=N):
    return {"n": n, "fill": fill, "set": [list(s) for s in sets]}


def Mx(m, n, fill, sets=()):
    return {"m": m, "n": n, "fill": fill, "set": [list(s) for s in sets]}


cases = []


def case(op, ref, args, expect):
    cases.append({"op": op, "ref": ref, "args": args, "expect": expect})


VT = "tests/LinAlg/vectorTests.hpp"
# ---- element-wise
case("setToConstant", f"{VT}:119", {"y": V(zero), "c": one}, {"y": V(one)})
case("setToConstant_w_patternSelect", f"{VT}:164", {"y": V(zero), "c": two, "select": V(one, [(-1, zero)])},
     {"y": V(two, [(-1, zero)])})
case("copyFrom", f"{VT}:198", {"y": V(zero), "x": V(one)}, {"y": V(one)})
case("selectPattern", f"{VT}:750", {"y": V(two), "select": V(one, [(-1, zero)])}, {"y": V(two, [(-1, zero)])})
case("scale", f"{VT}:781", {"y": V(half), "c": half}, {"y": V(quarter)})
case("componentMult", f"{VT}:795", {"y": V(half), "x": V(half)}, {"y": V(quarter)})
case("componentDiv", f"{VT}:814", {"y": V(one), "x": V(two)}, {"y": V(half)})
case("componentDiv_w_selectPattern", f"{VT}:833",
     {"y": V(half), "x": V(one, [(-1, zero)]), "select": V(one, [(-1, zero)])}, {"y": V(half, [(-1, zero)])})
case("component_min_c", f"{VT}:872", {"y": V(one), "c": half}, {"y": V(half)})
case("component_min_v", f"{VT}:887", {"y": V(one), "x": V(half)}, {"y": V(half)})
case("component_max_c", f"{VT}:906", {"y": V(one), "c": two}, {"y": V(two)})
case("component_max_v", f"{VT}:921", {"y": V(one), "x": V(two)}, {"y": V(two)})
case("component_abs", f"{VT}:940", {"y": V(-quarter, [(-1, half)])}, {"y": V(quarter, [(-1, half)])})
case("component_sqrt", f"{VT}:968", {"y": V(quarter, [(-1, two * two)])}, {"y": V(half, [(-1, two)])})
case("component_sgn", f"{VT}:996", {"y": V(-quarter, [(-1, half)])}, {"y": V(-one, [(-1, one)])})
case("axpy", f"{VT}:1077", {"y": V(two), "alpha": half, "x": V(two)}, {"y": V(two + half * two)})
case("axzpy", f"{VT}:1103", {"y": V(two), "alpha": half, "x": V(two), "z": V(two)}, {"y": V(two + half * two * two)})
case("axdzpy", f"{VT}:1135", {"y": V(two), "alpha": three, "x": V(half), "z": V(half)}, {"y": V(two + three * half / half)})
case("axdzpy_w_pattern", f"{VT}:1167",
     {"y": V(two), "alpha": three, "x": V(half), "z": V(half, [(-1, zero)]), "select": V(one, [(-1, zero)])},
     {"y": V(two + three * half / half, [(-1, two)])})
case("addConstant", f"{VT}:1211", {"y": V(zero), "c": two}, {"y": V(two)})
case("addConstant_w_patternSelect", f"{VT}:1228", {"y": V(zero), "c": half, "select": V(one, [(-1, zero)])},
     {"y": V(half, [(-1, zero)])})
case("negate", f"{VT}:1283", {"y": V(one)}, {"y": V(-one)})
case("invert", f"{VT}:1296", {"y": V(two)}, {"y": V(half)})
case("addLogBarrierGrad", f"{VT}:1372", {"y": V(two), "alpha": half, "x": V(two), "select": V(one, [(-1, zero)])},
     {"y": V(two + half / two, [(-1, two)])})
case("addLinearDampingTerm", f"{VT}:1461",
     {"y": V(one), "ixl": V(one, [(0, zero), (1, zero), (2, one), (3, one)]),
      "ixr": V(zero, [(0, one), (1, zero), (2, one), (3, zero)]), "alpha": quarter, "ct": two},
     {"y_first4": [quarter - two, quarter, quarter, quarter + two]})
# adjustDuals_plh: z = 1, x = 2, mu = kappa = 1/2: a = mu/x = 1/4, b = a/kappa = 1/2, a*kappa = 1/8; z=1 >= b, a<=b -> z = b
case("adjustDuals_plh", f"{VT}:1891", {"z": V(one), "x": V(two), "select": V(one), "mu": half, "kappa": half}, {"z": V(half)})
# ---- reductions
case("onenorm", f"{VT}:1024", {"x": V(-one)}, {"value": float(N)})
case("twonorm", f"{VT}:1040", {"x": V(-one)}, {"value": math.sqrt(N)})
case("infnorm", f"{VT}:1057", {"x": V(one, [(-1, -two)])}, {"value": two})
case("dotProductWith", f"{VT}:1259", {"x": V(one), "y": V(two)}, {"value": two * N})
case("logBarrier_local", f"{VT}:1309", {"x": V(one, [(-1, 1000 * three)]), "select": V(one, [(-1, zero)])},
     {"value": (N - 1) * math.log(one)})
case("logBarrier_local", f"{VT}:1331", {"x": V(zero, [(-1, one)]), "select": V(zero, [(-1, one)])}, {"value": math.log(one)})
case("sum_local", f"{VT}:1350", {"x": V(half, [(-1, two)])}, {"value": (N - 1) * half + two})
case("linearDampingTerm_local", f"{VT}:1417",
     {"x": V(one), "ixl": V(one, [(-1, two)]), "ixr": V(zero, [(-1, two)]), "mu": two, "kappa_d": two},
     {"value": (N - 1) * one * two * two})
case("allPositive", f"{VT}:1552", {"x": V(one)}, {"value": 1})
case("allPositive", f"{VT}:1560", {"x": V(one, [(-1, -one)])}, {"value": 0})
case("allPositive_w_patternSelect", f"{VT}:1574", {"x": V(one), "select": V(one)}, {"value": 1})
case("allPositive_w_patternSelect", f"{VT}:1584", {"x": V(-one), "select": V(one)}, {"value": 0})
case("allPositive_w_patternSelect", f"{VT}:1590", {"x": V(one, [(-1, -one)]), "select": V(one)}, {"value": 0})
case("min", f"{VT}:1606", {"x": V(two, [(-1, -one)])}, {"value": -one})
case("min", f"{VT}:1616", {"x": V(one, [(-1, two)])}, {"value": one})
case("min_w_pattern", f"{VT}:1630", {"x": V(one, [(-1, -one)]), "select": V(one)}, {"value": -one})
case("min_w_pattern", f"{VT}:1644", {"x": V(one, [(-1, -one)]), "select": V(one, [(-1, zero)])}, {"value": one})
case("fractionToTheBdry_local", f"{VT}:1773", {"x": V(one), "d": V(two), "tau": half}, {"value": one})
case("fractionToTheBdry_local", f"{VT}:1790", {"x": V(one), "d": V(-one, [(-1, -two)]), "tau": half}, {"value": quarter})
case("fractionToTheBdry_w_pattern_local", f"{VT}:1809", {"x": V(one), "d": V(one), "tau": half, "select": V(one)}, {"value": one})
case("fractionToTheBdry_w_pattern_local", f"{VT}:1828",
     {"x": V(one), "d": V(one, [(-1, -half)]), "tau": half, "select": V(one, [(-1, zero)])}, {"value": one})
case("fractionToTheBdry_w_pattern_local", f"{VT}:1840",
     {"x": V(one), "d": V(-one, [(-1, -two)]), "tau": half, "select": V(one)}, {"value": quarter})
case("matchesPattern", f"{VT}:1860", {"x": V(one), "select": V(one)}, {"value": 1})
case("matchesPattern", f"{VT}:1870", {"x": V(one), "select": V(one, [(-1, zero)])}, {"value": 0})
case("matchesPattern", f"{VT}:1876", {"x": V(one, [(-1, zero)]), "select": V(one)}, {"value": 1})
# projectIntoBounds (kappa1 = kappa2 = 1/2)
PB = f"{VT}:1656"
case("projectIntoBounds_local", PB, {"x": V(one), "xl": V(one), "ixl": V(one), "xu": V(-one), "ixu": V(one), "kappa1": half,
                                      "kappa2": half}, {"ok": 0})
case("projectIntoBounds_local", PB, {"x": V(one), "xl": V(-one), "ixl": V(one), "xu": V(one), "ixu": V(one), "kappa1": half,
                                      "kappa2": half}, {"ok": 1, "x": V(half)})
case("projectIntoBounds_local", PB, {"x": V(-two), "xl": V(zero), "ixl": V(one), "xu": V(two), "ixu": V(one), "kappa1": half,
                                      "kappa2": half}, {"ok": 1, "x": V(half)})
case("projectIntoBounds_local", PB, {"x": V(two), "xl": V(-two), "ixl": V(one), "xu": V(zero), "ixu": V(one), "kappa1": half,
                                      "kappa2": half}, {"ok": 1, "x": V(-half)})

# ---- dense matrices (row-major M x Nn), tests/LinAlg/matrixTestsDense.hpp
MD = "tests/LinAlg/matrixTestsDense.hpp"
M_, Nn = 10, 100
case("mat_timesVec", f"{MD}:173", {"A": Mx(M_, Nn, one), "beta": one, "y": V(three, n=M_), "alpha": one, "x": V(three, n=Nn)},
     {"y": V(three + three * Nn, n=M_)})
case("mat_transTimesVec", f"{MD}:208",
     {"A": Mx(M_, Nn, one, [(i, Nn - 1, zero) for i in range(M_)]), "beta": one, "y": V(three, n=Nn), "alpha": one,
      "x": V(three, n=M_)}, {"y": V(three + three * M_, [(-1, three)], n=Nn)})
case("mat_addSubDiagonal", f"{MD}:447", {"A": Mx(Nn, Nn, half), "alpha": half, "start": Nn - M_, "d": V(one, n=M_), "src_start": 0,
                                          "num": M_},
     {"A": Mx(Nn, Nn, half, [(i, i, half + half) for i in range(Nn - M_, Nn)])})
case("mat_addSubDiagonal", f"{MD}:475", {"A": Mx(Nn, Nn, half), "alpha": half, "start": Nn - M_ + 1, "d": V(one, n=M_),
                                          "src_start": 1, "num": M_ - 1},
     {"A": Mx(Nn, Nn, half, [(i, i, half + half) for i in range(Nn - M_ + 1, Nn)])})
case("mat_addSubDiagonal_const", f"{MD}:494", {"A": Mx(Nn, Nn, half), "start": 1, "num": Nn - 2, "c": two},
     {"A": Mx(Nn, Nn, half, [(i, i, half + two) for i in range(1, Nn - 1)])})
case("mat_addMatrix", f"{MD}:516", {"A": Mx(M_, Nn, half), "alpha": half, "B": Mx(M_, Nn, one)}, {"A": Mx(M_, Nn, half + half)})
case("mat_transAddToSymDenseMatrixUpperTriangle", f"{MD}:544",
     {"A": Mx(M_, 20, half), "row_start": 0, "col_start": Nn - M_, "alpha": half, "W": Mx(Nn, Nn, one)},
     {"W": Mx(Nn, Nn, one, [(i, j, one + half * half) for i in range(20) for j in range(Nn - M_, Nn)])})
case("mat_addUpperTriangleToSymDenseMatrixUpperTriangle", f"{MD}:587",
     {"A": Mx(M_, M_, half), "diag_start": 0, "alpha": half, "W": Mx(Nn, Nn, one)},
     {"W": Mx(Nn, Nn, one, [(i, j, one + half * half) for i in range(M_) for j in range(i, M_)])})

# ---- sparse triplet (matrixTestsSparse.hpp), M = 5, N = 50, entries at columns 0,10,20,30,49 of every row, values 1
MS = "tests/LinAlg/matrixTestsSparse.hpp"
sm, sn = 5, 50
cols = [0, 10, 20, 30, 49]
sp = {"m": sm, "n": sn, "iRow": [r for r in range(sm) for _ in cols], "jCol": [c for _ in range(sm) for c in cols],
      "val": [one] * (sm * len(cols))}
off = 2
Wn = sm + 2 * off
# W += alpha * A D^-1 A^T on the upper triangle of the diagonal block at `off`: alpha=1/2, A=1, d=1/2 -> 1/2*1*1/(1/2)*5 = 5
case("sp_addMDinvMtransToDiagBlockOfSymDeMatUTri", f"{MS}:415",
     {"A": sp, "offset": off, "alpha": half, "D": V(half, n=sn), "W": Mx(Wn, Wn, zero)},
     {"W": Mx(Wn, Wn, zero, [(i, j, half * one * one / half * len(cols)) for i in range(off, off + sm) for j in range(i, off + sm)])})
case("sp_timesVec", f"{MS}:128", {"A": sp, "beta": one, "y": V(two, n=sm), "alpha": one, "x": V(three, n=sn)},
     {"y": V(two + three * len(cols), n=sm)})
case("sp_transTimesVec", f"{MS}:162", {"A": sp, "beta": one, "y": V(two, n=sn), "alpha": one, "x": V(three, n=sm)},
     {"y": V(two, [(c, two + three * sm) for c in cols], n=sn)})
# sym sparse: y += alpha*diag(A) (matrixTestsSymSparse.hpp:180): A = 1/2 on its entries, W = 1, alpha = 1/2
ss = {"m": 6, "n": 6, "iRow": [0, 0, 1, 2, 3, 3, 5], "jCol": [0, 4, 1, 5, 3, 4, 5], "val": [half] * 7}
case("spsym_startingAtAddSubDiagonalToStartingAt", "tests/LinAlg/matrixTestsSymSparse.hpp:180",
     {"A": ss, "alpha": half, "y": V(one, n=6)}, {"y": V(one, [(0, one + quarter), (1, one + quarter), (3, one + quarter), (5, one + quarter)], n=6)})

# ---- round 2: the rest of the a3 / a14 surface --------------------------------------------------------------
Kk = 12    # inner dimension of the GEMM tests (driver: K = 100, scaled)
# W = beta W + alpha A X, all local (matrixTestsDense.hpp:270): A = 2, X = 3, W = 2, alpha = beta = 2
case("mat_timesMat", f"{MD}:270", {"A": Mx(M_, Kk, two), "X": Mx(Kk, Nn, three), "W": Mx(M_, Nn, two), "alpha": two, "beta": two},
     {"W": Mx(M_, Nn, two * two + two * two * three * Kk)})
# W = beta W + alpha A^T X (:308): last row of X zero -> K - 1 terms
case("mat_transTimesMat", f"{MD}:308",
     {"A": Mx(Kk, M_, two), "X": Mx(Kk, Nn, three, [(Kk - 1, j, zero) for j in range(Nn)]), "W": Mx(M_, Nn, two), "alpha": two,
      "beta": two}, {"W": Mx(M_, Nn, two * two + two * two * three * (Kk - 1))})
# W = beta W + alpha A X^T (:360): last row of X zero -> last column of W only beta W
case("mat_timesMatTrans", f"{MD}:360",
     {"A": Mx(M_, Nn, two), "X": Mx(Kk, Nn, three, [(Kk - 1, j, zero) for j in range(Nn)]), "W": Mx(M_, Kk, two), "alpha": two,
      "beta": two},
     {"W": Mx(M_, Kk, two * two + two * two * three * Nn, [(i, Kk - 1, two * two) for i in range(M_)])})
case("mat_addDiagonal", f"{MD}:401", {"A": Mx(Nn, Nn, quarter), "alpha": two, "d": V(half, n=Nn)},
     {"A": Mx(Nn, Nn, quarter, [(i, i, quarter + half * two) for i in range(Nn)])})
case("mat_addDiagonal_const", f"{MD}:433", {"A": Mx(Nn, Nn, quarter), "alpha": two},
     {"A": Mx(Nn, Nn, quarter, [(i, i, quarter + two) for i in range(Nn)])})
case("mat_maxAbsValue", f"{MD}:625", {"A": Mx(M_, Nn, zero, [(M_ - 1, Nn - 1, one)])}, {"value": one})
case("mat_maxAbsValue", f"{MD}:639", {"A": Mx(M_, Nn, zero, [(M_ - 1, Nn - 1, -one)])}, {"value": one})
case("mat_row_max_abs_value", f"{MD}:651", {"A": Mx(M_, Nn, one, [(M_ - 1, Nn - 1, -two)])}, {"y": V(one, [(-1, two)], n=M_)})
case("mat_scale_row", f"{MD}:682", {"A": Mx(M_, Nn, two), "x": V(three, n=M_), "inv": 0}, {"A": Mx(M_, Nn, two * three)})
case("mat_isFinite", f"{MD}:707", {"A": Mx(M_, Nn, zero)}, {"ok": 1})
case("mat_isFinite", f"{MD}:719", {"A": Mx(M_, Nn, zero, [(M_ - 1, Nn - 1, "inf")])}, {"ok": 0})
Ms = 4     # a smaller source / destination for the copy tests
case("mat_copyRowsFrom", f"{MD}:778", {"dst": Mx(M_, Nn, one), "src": Mx(Ms, Nn, two), "num_rows": Ms, "row_dest": M_ - Ms},
     {"dst": Mx(M_, Nn, one, [(i, j, two) for i in range(M_ - Ms, M_) for j in range(Nn)])})
case("mat_copyRowsFromSelect", f"{MD}:818",
     {"dst": Mx(Ms, Nn, one), "src": Mx(M_, Nn, two, [(Ms - 1, j, zero) for j in range(Nn)]),
      "rows": [Ms - 1] + list(range(1, Ms - 1)) + [0]},
     {"dst": Mx(Ms, Nn, two, [(0, j, zero) for j in range(Nn)])})
case("mat_copyBlockFromMatrix", f"{MD}:859", {"src": Mx(Ms, 20, one), "dst": Mx(M_, Nn, two), "i0": M_ - Ms, "j0": Nn - 20},
     {"dst": Mx(M_, Nn, two, [(i, j, one) for i in range(M_ - Ms, M_) for j in range(Nn - 20, Nn)])})
# dst = src[i0 : i0 + m, j0 : j0 + n] with i0 = src_m - dst_m - 1 (:897): the zero at the last element of src is NOT inside
case("mat_copyFromMatrixBlock", f"{MD}:897",
     {"src": Mx(M_, Nn, one, [(M_ - 1, Nn - 1, zero)]), "dst": Mx(Ms, 20, two), "i0": M_ - Ms - 1, "j0": Nn - 20 - 1},
     {"dst": Mx(Ms, 20, one)})
case("mat_shiftRows", f"{MD}:979", {"A": Mx(M_, Nn, one, [(0, j, two) for j in range(Nn)]), "shift": M_ - 1},
     {"A": Mx(M_, Nn, one, [(i, j, two) for i in (0, M_ - 1) for j in range(Nn)])})
case("mat_shiftRows", f"{MD}:1003", {"A": Mx(M_, Nn, one, [(M_ - 1, j, two) for j in range(Nn)]), "shift": -(M_ - 1)},
     {"A": Mx(M_, Nn, one, [(i, j, two) for i in (0, M_ - 1) for j in range(Nn)])})
case("mat_symmetrize", f"{MD}:1102", {"A": Mx(M_, M_, zero, [(i, j, one) for i in range(M_) for j in range(i + 1, M_)])},
     {"A": Mx(M_, M_, one, [(i, i, zero) for i in range(M_)])})
# ---- sparse (driver tests/testMatrixSparse.cpp:85-170: second matrix M2 = 2 M rows, same columns; offsets i = 1, j = M2 + 1)
sm2 = 2 * sm
sp2 = {"m": sm2, "n": sn, "iRow": [r for r in range(sm2) for _ in cols], "jCol": [c for _ in range(sm2) for c in cols],
       "val": [one] * (sm2 * len(cols))}
case("sp_maxAbsValue", f"{MS}:195", {"A": dict(sp, val=[zero] * (sm * len(cols) - 1) + [one])}, {"value": one})
case("sp_maxAbsValue", f"{MS}:209", {"A": dict(sp, val=[one] * (sm * len(cols) - 1) + [-two])}, {"value": two})
case("sp_row_max_abs_value", f"{MS}:223", {"A": dict(sp, val=[one] * (sm * len(cols) - 1) + [-two])}, {"y": V(one, [(-1, two)], n=sm)})
case("sp_scale_row", f"{MS}:255", {"A": dict(sp, val=[two] * (sm * len(cols))), "x": V(three, n=sm), "inv": 0},
     {"val": V(two * three, n=sm * len(cols))})
case("sp_isFinite", f"{MS}:277", {"A": dict(sp, val=[two] * (sm * len(cols)))}, {"ok": 1})
case("sp_isFinite", f"{MS}:286", {"A": dict(sp, val=[two] * (sm * len(cols) - 1) + ["inf"])}, {"ok": 0})
# W = beta W + alpha A B^T (:504): alpha = 1/2, beta = 2, A = B = 1, W = 0: every (i, j) shares all 5 columns
case("sp_timesMatTrans", f"{MS}:504", {"A": sp, "B": sp2, "W": Mx(sm, sm2, zero), "alpha": half, "beta": two},
     {"W": Mx(sm, sm2, two * zero + half * one * one * len(cols))})
# W += alpha A D^-1 B^T above the diagonal (:587): alpha = 1/2, d = 1/2, offsets (1, M2 + 1); W is (N + 10 M)^2 in the driver,
# here just large enough; only i <= j entries of the block are touched
Wb = 1 + sm2 + 1 + sm2 + 2
case("sp_addMDinvNtransToSymDeMatUTri", f"{MS}:587",
     {"A": sp, "B": sp2, "D": V(half, n=sn), "W": Mx(Wb, Wb, zero), "alpha": half, "i_offset": 1, "j_offset": sm2 + 1},
     {"W": Mx(Wb, Wb, zero, [(i, j, half * one * one / half * len(cols)) for i in range(1, 1 + sm)
                             for j in range(sm2 + 1, sm2 + 1 + sm2) if i <= j])})
# dense copy of the sparse matrix (:1218)
case("sp_copy_to", f"{MS}:1218", {"A": dict(sp, val=[two] * (sm * len(cols))), "W": Mx(sm, sn, one)},
     {"W": Mx(sm, sn, zero, [(r, c, two) for r in range(sm) for c in cols])})
# ---- symmetric sparse (matrixTestsSymSparse.hpp): upper-triangle triplets ss, A = 1/2
SS = "tests/LinAlg/matrixTestsSymSparse.hpp"
ssn = 6
full = {}
for r_, c_ in zip(ss["iRow"], ss["jCol"]):
    full[(r_, c_)] = 1
    full[(c_, r_)] = 1
per_row = [sum(1 for (r_, c_) in full if r_ == i) for i in range(ssn)]
# y = beta y + alpha A x (:87): alpha = 2, beta = 1/2, A = 1/2, y = 2, x = 3; a row counts its entries of the FULL symmetric matrix
case("spsym_timesVec", f"{SS}:87", {"A": ss, "beta": half, "y": V(two, n=ssn), "alpha": two, "x": V(three, n=ssn)},
     {"y": {"n": ssn, "fill": 0.0, "set": [[i, half * two + two * half * three * per_row[i]] for i in range(ssn)]}})
# W upper += alpha A at the diagonal offset (:128): alpha = 1/2, A = 1/2, W = 1
Wd0 = 3
case("spsym_addUpperTriangleToSymDenseMatrixUpperTriangle", f"{SS}:128",
     {"A": ss, "diag_start": Wd0, "alpha": half, "W": Mx(ssn + Wd0, ssn + Wd0, one)},
     {"W": Mx(ssn + Wd0, ssn + Wd0, one, [(Wd0 + r_, Wd0 + c_, one + half * half) for r_, c_ in zip(ss["iRow"], ss["jCol"])])})

# ---- round 6: the copy family, the is* predicates, the remaining dense row / triangle methods -------------------------------
# driver sizes (tests/testVector.cpp:235-262): v has Nlocal = 1000 entries, v_smaller / v2_smaller Mlocal = 500
Ms_ = 500


def VR(fill, ranges, n=N):
    """vector with constant ranges: [[lo, hi, val], ...] (hi exclusive)"""
    return {"n": n, "fill": fill, "set": [], "range": [list(r) for r in ranges]}


def LIN(n, i0, di=1):
    return {"n": n, "i0": i0, "di": di}


case("setToZero", f"{VT}:98", {"y": V(one)}, {"y": V(zero)})
# v = 3 (size M), from = 1 with from[M-1] = 2 (size N), idxs = 0, 1, ..., M-1: v becomes the first M entries of from
case("copy_from_indexes", f"{VT}:221", {"y": V(three, n=Ms_), "src": V(one, [(Ms_ - 1, two)]), "idx": LIN(Ms_, 0)},
     {"y": V(one, [(-1, two)], n=Ms_)})
case("copyFromStarting", f"{VT}:274", {"y": V(two), "start": 1, "src": V(one, n=N - 1)}, {"y": V(one, [(0, two)])})
case("copyFromStarting", f"{VT}:284", {"y": V(two), "start": N - Ms_, "src": V(one, n=Ms_)}, {"y": VR(two, [(N - Ms_, N, one)])})
case("copyFromStarting", f"{VT}:293", {"y": V(two), "start": 1, "src": V(one, n=Ms_)}, {"y": VR(two, [(1, Ms_ + 1, one)])})
case("copyFromStarting", f"{VT}:301", {"y": V(two), "start": 0, "src": V(one, n=0)}, {"y": V(two)})
# dest = 1 (size M), src = 2 (size N): one element (start_dest = M - 1, start_src = N / 2), then all of dest from src's tail
case("startingAtCopyFromStartingAt", f"{VT}:346", {"y": V(one, n=Ms_), "start_dest": Ms_ - 1, "src": V(two), "start_src": N // 2},
     {"y": V(one, [(-1, two)], n=Ms_)})
case("startingAtCopyFromStartingAt", f"{VT}:358", {"y": V(one, n=Ms_), "start_dest": 0, "src": V(two), "start_src": N - Ms_},
     {"y": V(two, n=Ms_)})
case("copyTo", f"{VT}:373", {"x": V(two), "dest": V(one)}, {"dest": V(two)})
case("copyToStarting", f"{VT}:397", {"x": V(one, n=Ms_), "dest": V(two), "start": N - Ms_}, {"dest": VR(two, [(N - Ms_, N, one)])})
case("copyToStarting", f"{VT}:430", {"x": V(one, n=0), "dest": V(two), "start": 0}, {"dest": V(two)})
# from = 1 with 3 at both ends, pattern = 0 with 1 at both ends: the two selected entries land at start, start + 1
case("copyToStartingAt_w_pattern", f"{VT}:449",
     {"x": V(one, [(0, three), (-1, three)], n=Ms_), "dest": V(two), "start": N - Ms_, "select": V(zero, [(0, one), (-1, one)], n=Ms_)},
     {"dest": V(two, [(N - Ms_, three), (N - Ms_ + 1, three)]), "nnz": 2})
case("copy_from_two_vec_w_pattern", f"{VT}:496",
     {"y": V(zero), "c": V(one, n=Ms_), "c_map": LIN(Ms_, 0), "d": V(two, n=Ms_), "d_map": LIN(Ms_, Ms_)}, {"y": VR(one, [(Ms_, N, two)])})
case("copy_to_two_vec_w_pattern", f"{VT}:540",
     {"y": V(two), "c": V(zero, n=Ms_), "c_map": LIN(Ms_, 0), "d": V(zero, n=Ms_), "d_map": LIN(Ms_, Ms_)},
     {"c": V(two, n=Ms_), "d": V(two, n=Ms_)})
# startingAtCopyToStartingAt (from = 1, size M; to = 2, size N): the edge cases of :580-745 in the test's order
SA = f"{VT}:580"
case("startingAtCopyToStartingAt", SA, {"src": V(one, n=0), "start_src": 0, "dest": V(two), "start_dest": 0, "num": -1}, {"dest": V(two)})
case("startingAtCopyToStartingAt", SA, {"src": V(one, n=Ms_), "start_src": 0, "dest": V(two), "start_dest": 0, "num": 0}, {"dest": V(two)})
case("startingAtCopyToStartingAt", SA, {"src": V(one, n=Ms_), "start_src": 0, "dest": V(two), "start_dest": 0, "num": Ms_},
     {"dest": VR(two, [(0, Ms_, one)])})
case("startingAtCopyToStartingAt", SA, {"src": V(one, n=Ms_), "start_src": 0, "dest": V(two), "start_dest": N - Ms_, "num": Ms_},
     {"dest": VR(two, [(N - Ms_, N, one)])})
case("startingAtCopyToStartingAt", SA, {"src": V(one, n=Ms_), "start_src": 0, "dest": V(two), "start_dest": N - Ms_, "num": Ms_ // 2},
     {"dest": VR(two, [(N - Ms_, N - Ms_ + Ms_ // 2, one)])})
case("startingAtCopyToStartingAt", SA, {"src": V(one, n=Ms_), "start_src": 0, "dest": V(two), "start_dest": N - Ms_, "num": -1},
     {"dest": VR(two, [(N - Ms_, N, one)])})
case("startingAtCopyToStartingAt", SA, {"src": V(one, n=Ms_), "start_src": 0, "dest": V(two), "start_dest": N, "num": -1}, {"dest": V(two)})
# the predicates (:1951-2036): 0, 1/0, 0/0, and a single special value at the end of a vector of ones
case("isnan", f"{VT}:1955", {"x": V(zero)}, {"value": 0})
case("isnan", f"{VT}:1959", {"x": V("inf")}, {"value": 0})
case("isnan", f"{VT}:1963", {"x": V("nan")}, {"value": 1})
case("isnan", f"{VT}:1967", {"x": V(one, [(-1, "nan")])}, {"value": 1})
case("isinf", f"{VT}:1986", {"x": V(zero)}, {"value": 0})
case("isinf", f"{VT}:1990", {"x": V("nan")}, {"value": 0})
case("isinf", f"{VT}:1994", {"x": V("inf")}, {"value": 1})
case("isinf", f"{VT}:1998", {"x": V(one, [(-1, "inf")])}, {"value": 1})
case("isfinite", f"{VT}:2017", {"x": V(zero)}, {"value": 1})
case("isfinite", f"{VT}:2021", {"x": V("nan")}, {"value": 0})
case("isfinite", f"{VT}:2025", {"x": V(one, [(-1, "inf")])}, {"value": 0})
# ---- dense, the rest (matrixTestsDense.hpp)
case("mat_setToZero", f"{MD}:99", {"A": Mx(M_, Nn, one)}, {"A": Mx(M_, Nn, zero)})
case("mat_setToConstant", f"{MD}:108", {"A": Mx(M_, Nn, zero), "c": one}, {"A": Mx(M_, Nn, one)})
case("mat_copyFrom", f"{MD}:118", {"dst": Mx(M_, Nn, zero), "src": Mx(M_, Nn, one)}, {"dst": Mx(M_, Nn, one)})
case("mat_copy_to", f"{MD}:147", {"src": Mx(M_, Nn, one), "dst": Mx(M_, Nn, zero)}, {"dst": Mx(M_, Nn, one)})
# appendRow: A = 1 with room for one more row, vec = 2: the new last row is the vector, the row count grows by one
case("mat_appendRow", f"{MD}:736", {"A": Mx(M_, Nn, one), "vec": V(two, n=Nn)},
     {"A": Mx(M_ + 1, Nn, one, [(M_, j, two) for j in range(Nn)]), "m": M_ + 1})
case("mat_replaceRow", f"{MD}:1027", {"A": Mx(M_, Nn, one), "row": M_ - 1, "vec": V(two, [(1, zero)], n=Nn)},
     {"A": Mx(M_, Nn, one, [(M_ - 1, j, (zero if j == 1 else two)) for j in range(Nn)])})
case("mat_getRow", f"{MD}:1068", {"A": Mx(M_, Nn, one, [(M_ - 1, Nn - 1, zero)]), "row": M_ - 1, "vec": V(two, n=Nn)},
     {"vec": V(one, [(-1, zero)], n=Nn)})
# rows set to their index (A[i][j] = i), then one triangle copied over the other (:1169, :1189)
rowidx = [(i, j, float(i)) for i in range(M_) for j in range(M_)]
case("mat_overwriteUpperTriangleWithLower", f"{MD}:1169", {"A": Mx(M_, M_, zero, rowidx)},
     {"A": Mx(M_, M_, zero, [(i, j, float(i if j <= i else j)) for i in range(M_) for j in range(M_)])})
case("mat_overwriteLowerTriangleWithUpper", f"{MD}:1189", {"A": Mx(M_, M_, zero, rowidx)},
     {"A": Mx(M_, M_, zero, [(i, j, float(j if j < i else i)) for i in range(M_) for j in range(M_)])})

# ---- sparse triplet, the assembly surface (matrixTestsSparse.hpp; driver arguments of tests/testMatrixSparse.cpp:150-215) ----------
# test matrices: initializeMatrix (matrixTestsSparseTriplet.cpp:303-330): `epr` entries per row at columns 0, n/epr, 2n/epr, ..., the
# last one in column n - 1
def pattern(m, n, epr, val):
    iR, jC = [], []
    for r_ in range(m):
        for e_ in range(epr - 1):
            iR.append(r_); jC.append(e_ * (n // epr))
        iR.append(r_); jC.append(n - 1)
    return {"m": m, "n": n, "iRow": iR, "jCol": jC, "val": [val] * len(iR)}


def image_rule(A, nnz_st, nnz_cp, A_val, B_entries, B_val):
    """The expected dense copy as the reference's tests state it (e.g. :1100-1125): an entry that is both in the part of A that was not
    replaced and among the new entries shows their sum, one that is only among the new entries the new value, one that is only in A
    A's value, everything else zero."""
    kept = set(zip(A["iRow"][:nnz_st], A["jCol"][:nnz_st])) | set(zip(A["iRow"][nnz_st + nnz_cp:], A["jCol"][nnz_st + nnz_cp:]))
    out = {}
    for ij in kept | set(B_entries):
        in_kept, in_B = ij in kept, ij in B_entries
        out[ij] = (B_val + A_val) if (in_kept and in_B) else (B_val if in_B else A_val)
    return [(i, j, v) for (i, j), v in sorted(out.items())]


mxn = pattern(sm, sn, 5, one)                      # 5 x 50
m2xn = pattern(2 * sm, sn, 5, two)                 # 10 x 50
n4 = 2 * sm + sn                                   # 60
nnz1, nnz4 = sm * 5, n4 * 5
# copyRowsFrom (:939): row i of A = row 2 i of B; A = 1, B = 2 -> every value of A is 2, its pattern is unchanged (the rows share it)
case("sp_copyRowsFrom", f"{MS}:939", {"A": mxn, "B": m2xn, "select": [2 * i for i in range(sm)]},
     {"val": V(two, n=nnz1), "iRow": mxn["iRow"], "jCol": mxn["jCol"], "nnz": nnz1})
# copyRowsBlockFrom (:1009; driver :179-180): row 0 of the source becomes row M - 1 of the destination at nnz position nnz - 5
case("sp_copyRowsBlockFrom", f"{MS}:1009", {"dest": dict(m2xn, val=[half] * (2 * nnz1)), "src": mxn, "src_row_st": 0, "n_rows": 1,
                                             "dest_row_st": sm - 1, "dest_nnz_st": nnz1 - 5},
     {"val": VR(half, [(nnz1 - 5, nnz1, one)], n=2 * nnz1)})
case("sp_copyRowsBlockFrom", f"{MS}:1009", {"dest": dict(m2xn, val=[half] * (2 * nnz1)), "src": pattern(sm, sn, 5, one) | {"iRow": [], "jCol": [], "val": []},
                                             "src_row_st": 0, "n_rows": 1, "dest_row_st": sm - 1, "dest_nnz_st": nnz1},
     {"val": V(half, n=2 * nnz1)})
# copySubmatrixFrom / Trans (:1070, :1147; driver :188-193): B = the 5 x 50 matrix (value 2) into the 60 x 60 one (value 1) at (M, 2M),
# written over the last nnz(B) entries
A4 = pattern(n4, n4, 5, one)
Bsub = dict(mxn, val=[two] * nnz1)
Bent = {(sm + i, 2 * sm + j) for i, j in zip(Bsub["iRow"], Bsub["jCol"])}
case("sp_copySubmatrixFrom", f"{MS}:1070", {"A": A4, "B": Bsub, "row_st": sm, "col_st": 2 * sm, "nnz_st": nnz4 - nnz1, "trans": 0},
     {"W": Mx(n4, n4, zero, image_rule(A4, nnz4 - nnz1, nnz1, one, Bent, two))})
BentT = {(sm + j, 2 * sm + i) for i, j in zip(Bsub["iRow"], Bsub["jCol"])}
case("sp_copySubmatrixFrom", f"{MS}:1147", {"A": A4, "B": Bsub, "row_st": sm, "col_st": 2 * sm, "nnz_st": nnz4 - nnz1, "trans": 1},
     {"W": Mx(n4, n4, zero, image_rule(A4, nnz4 - nnz1, nnz1, one, BentT, two))})
# copyDiagMatrixToSubblock (:1256; driver :201): 2 I of order nnz(B) = 25 at (M, 2M) over the entries [nnz4 - 2 nnz, nnz4 - nnz); A = 1/2
A4h = pattern(n4, n4, 5, half)
Dent = {(sm + e, 2 * sm + e) for e in range(nnz1)}
case("sp_copyDiagMatrixToSubblock", f"{MS}:1256", {"A": A4h, "src_val": two, "row_st": sm, "col_st": 2 * sm, "nnz_st": nnz4 - 2 * nnz1, "nnz_to_copy": nnz1},
     {"W": Mx(n4, n4, zero, image_rule(A4h, nnz4 - 2 * nnz1, nnz1, half, Dent, two))})
# ..._w_pattern (:1327; driver :205): D = 2 (length N = 50), the LAST M entries of the pattern selected: M diagonal entries
patt = V(zero, [(sn - 1 - i, one) for i in range(sm)], n=sn)
Dent5 = {(sm + e, 2 * sm + e) for e in range(sm)}
case("sp_copyDiagMatrixToSubblock_w_pattern", f"{MS}:1327",
     {"A": A4h, "D": V(two, n=sn), "pattern": patt, "row_st": sm, "col_st": 2 * sm, "nnz_st": nnz4 - 2 * nnz1, "nnz_to_copy": sm},
     {"W": Mx(n4, n4, zero, image_rule(A4h, nnz4 - 2 * nnz1, sm, half, Dent5, two))})
# setSubmatrixToConstantDiag_w_colpattern / _rowpattern (:1408, :1492; driver :209-210): entry q of the pattern (q = N - 5 .. N - 1 set)
# gives (row_st + q, col_st + found) / (row_st + found, col_st + q)
colent = {(sm + (sn - sm + f), 2 * sm + f) for f in range(sm)}
rowent = {(sm + f, 2 * sm + (sn - sm + f)) for f in range(sm)}
case("sp_setSubmatrixToConstantDiag_w_pattern", f"{MS}:1408",
     {"A": A4h, "scalar": two, "pattern": patt, "row_st": sm, "col_st": 2 * sm, "nnz_st": nnz4 - 2 * nnz1, "nnz_to_copy": sm, "rowpattern": 0},
     {"W": Mx(n4, n4, zero, image_rule(A4h, nnz4 - 2 * nnz1, sm, half, colent, two))})
case("sp_setSubmatrixToConstantDiag_w_pattern", f"{MS}:1492",
     {"A": A4h, "scalar": two, "pattern": patt, "row_st": sm, "col_st": 2 * sm, "nnz_st": nnz4 - 2 * nnz1, "nnz_to_copy": sm, "rowpattern": 1},
     {"W": Mx(n4, n4, zero, image_rule(A4h, nnz4 - 2 * nnz1, sm, half, rowent, two))})
# transAddToSymDenseMatrixUpperTriangle (:846; driver :139): W (100 x 100) = 1, A (5 x 50) = 1/2, alpha = 1/2 at (0, 100 - 5):
# W[j][95 + i] += alpha A[i][j] for the entries of A (all inside the upper triangle)
Wn = sn + 10 * sm
case("sp_transAddToSymDenseMatrixUpperTriangle", f"{MS}:846",
     {"A": dict(mxn, val=[half] * nnz1), "row_st": 0, "col_st": Wn - sm, "alpha": half, "W": Mx(Wn, Wn, one)},
     {"W": Mx(Wn, Wn, one, [(j, Wn - sm + i, one + half * half) for i, j in zip(mxn["iRow"], mxn["jCol"])])})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_unit_tests.json")
with open(out, "w") as f:
    json.dump({"source": "LLNL/hiop v1.1.0 tests/LinAlg (transcribed constants and expected values)", "cases": cases}, f, indent=0)
print(f"{out}: {len(cases)} cases")
