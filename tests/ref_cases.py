"""Runs the known-answer cases of tests/golden/reference_unit_tests.json through (a) the oracle and (b) the HIP
kernels (C ABI).  The comparison tolerance is the reference's own `isEqual`: relative 10*eps
(tests/LinAlg/testBase.hpp:76-77,106-109)."""
import ctypes as C
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).parent / "golden" / "reference_unit_tests.json"
EPS10 = 10 * np.finfo(np.float64).eps


def load_cases():
    return json.loads(GOLD.read_text())["cases"]


def _num(v):
    return float("inf") if v == "inf" else float("nan") if v == "nan" else v


def vec(d):
    a = np.full(d["n"], _num(d["fill"]), dtype=np.float64)
    for lo, hi, v in d.get("range", []):
        a[lo:hi] = _num(v)
    for i, v in d["set"]:
        a[i] = _num(v)
    return a


def lin(d):
    """hiopVectorInt::linspace(i0, di) of n entries"""
    return (d["i0"] + d["di"] * np.arange(d["n"])).astype(np.int32)


def mat(d):
    a = np.full((d["m"], d["n"]), d["fill"], dtype=np.float64)
    for i, j, v in d["set"]:
        a[i, j] = _num(v)
    return a


def coo(d):
    return (np.array(d["iRow"], np.int32), np.array(d["jCol"], np.int32), np.array([_num(v) for v in d["val"]], np.float64),
            d["m"], d["n"])


def close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= EPS10 * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b))))


def check(case, out):
    """out: dict name -> value produced; compares with case['expect']."""
    for k, e in case["expect"].items():
        if k in ("value", "ok", "nnz", "m"):
            assert close(out[k], e), (case["op"], case["ref"], k, out[k], e)
        elif k == "y_first4":
            assert close(out["y"][:4], e), (case["op"], case["ref"], out["y"][:4], e)
        elif isinstance(e, list):
            assert np.array_equal(np.asarray(out[k]), np.asarray(e)), (case["op"], case["ref"], k)
        elif isinstance(e, dict) and "m" in e:
            assert close(out[k], mat(e)), (case["op"], case["ref"], k)
        else:
            assert close(out[k], vec(e)), (case["op"], case["ref"], k, out[k][:5], vec(e)[:5])


# ------------------------------------------------------------------ sparse assembly (both sides)
_SP_ASSEMBLY = ("sp_copyRowsFrom", "sp_copyRowsBlockFrom", "sp_copySubmatrixFrom", "sp_copyDiagMatrixToSubblock",
                "sp_copyDiagMatrixToSubblock_w_pattern", "sp_setSubmatrixToConstantDiag_w_pattern", "sp_transAddToSymDenseMatrixUpperTriangle")


def _trip(d):
    i, j, v, m, n = coo(d)
    return [i.copy(), j.copy(), v.copy()], m, n


def _sp_assembly_oracle(op, a):
    from oracle import hiop_oracle as ho
    if op == "sp_copyRowsFrom":
        T, m, n = _trip(a["A"]); S, _, _ = _trip(a["B"])
        nnz = ho.sp_copy_rows_from(T, S, np.array(a["select"], np.int32))
        return {"val": T[2], "iRow": T[0], "jCol": T[1], "nnz": nnz}
    if op == "sp_copyRowsBlockFrom":
        T, m, n = _trip(a["dest"]); S, _, _ = _trip(a["src"])
        ho.sp_copy_rows_block_from(T, S, a["src_row_st"], a["n_rows"], a["dest_row_st"], a["dest_nnz_st"])
        return {"val": T[2]}
    if op == "sp_transAddToSymDenseMatrixUpperTriangle":
        i, j, v, m, n = coo(a["A"]); W = mat(a["W"])
        ho.sp_trans_add_to_sym_upper(i, j, v, a["row_st"], a["col_st"], a["alpha"], W)
        return {"W": W}
    T, m, n = _trip(a["A"])
    if op == "sp_copySubmatrixFrom":
        S, _, _ = _trip(a["B"])
        ho.sp_copy_submatrix_from(T, S, a["row_st"], a["col_st"], a["nnz_st"], False, bool(a["trans"]))
    elif op == "sp_copyDiagMatrixToSubblock":
        ho.sp_copy_diag_matrix_to_subblock(T, a["src_val"], a["row_st"], a["col_st"], a["nnz_st"], a["nnz_to_copy"])
    elif op == "sp_copyDiagMatrixToSubblock_w_pattern":
        assert ho.sp_copy_diag_matrix_to_subblock_w_pattern(T, vec(a["D"]), a["row_st"], a["col_st"], a["nnz_st"], vec(a["pattern"])) == a["nnz_to_copy"]
    else:
        assert ho.sp_set_submatrix_to_constant_diag_w_pattern(T, a["scalar"], a["row_st"], a["col_st"], a["nnz_st"], vec(a["pattern"]),
                                                              bool(a["rowpattern"])) == a["nnz_to_copy"]
    return {"W": ho.sp_copy_to_dense(m, n, T[0], T[1], T[2])}     # copy_to: duplicates add up (what the tests' "B_val + A_val" relies on)


def _sp_assembly_gpu(ctx, op, a):
    import torch
    D = lambda arr, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(arr)).to(dt).cuda()
    I32 = lambda arr: D(arr, torch.int32)
    P = lambda t: C.c_void_p(t.data_ptr() if t.numel() else 0)

    def dev_trip(d):
        i, j, v, m, n = coo(d)
        return [I32(i), I32(j), D(v)], m, n

    def run(name, *args):
        torch.cuda.synchronize(); ctx.call(name, *args); ctx.sync()

    def image(T, m, n):
        W = D(np.full((m, n), 7.0))
        run("hiopamd_sp_copy_to_dense", m, n, T[2].numel(), T[0], T[1], T[2], W, n)
        return W.cpu().numpy()
    if op == "sp_copyRowsFrom":
        T, m, n = dev_trip(a["A"]); S, _, _ = dev_trip(a["B"])
        run("hiopamd_sp_copy_rows_from", T[0], T[1], T[2], S[2].numel(), S[0], S[1], S[2], I32(np.array(a["select"], np.int32)), len(a["select"]))
        return {"val": T[2].cpu().numpy(), "iRow": T[0].cpu().numpy(), "jCol": T[1].cpu().numpy(), "nnz": T[2].numel()}
    if op == "sp_copyRowsBlockFrom":
        T, m, n = dev_trip(a["dest"]); S, _, _ = dev_trip(a["src"])
        run("hiopamd_sp_copy_rows_block_from", T[0], T[1], T[2], S[2].numel(), P(S[0]), P(S[1]), P(S[2]), a["src_row_st"], a["n_rows"], a["dest_row_st"],
            a["dest_nnz_st"])
        return {"val": T[2].cpu().numpy()}
    if op == "sp_transAddToSymDenseMatrixUpperTriangle":
        T, m, n = dev_trip(a["A"]); W = D(mat(a["W"]))
        run("hiopamd_sp_trans_add_to_sym_upper", T[2].numel(), T[0], T[1], T[2], a["row_st"], a["col_st"], a["alpha"], W, W.shape[1])
        return {"W": W.cpu().numpy()}
    T, m, n = dev_trip(a["A"])
    if op == "sp_copySubmatrixFrom":
        S, _, _ = dev_trip(a["B"])
        run("hiopamd_sp_copy_submatrix_from", T[0], T[1], T[2], S[2].numel(), S[0], S[1], S[2], a["row_st"], a["col_st"], a["nnz_st"], 0, a["trans"])
    elif op == "sp_copyDiagMatrixToSubblock":
        run("hiopamd_sp_copy_diag_matrix_to_subblock", T[0], T[1], T[2], a["src_val"], a["row_st"], a["col_st"], a["nnz_st"], a["nnz_to_copy"])
    elif op == "sp_copyDiagMatrixToSubblock_w_pattern":
        found = C.c_int(-1); patt = vec(a["pattern"])
        run("hiopamd_sp_copy_diag_matrix_to_subblock_w_pattern", T[0], T[1], T[2], D(vec(a["D"])), a["row_st"], a["col_st"], a["nnz_st"], patt.size, D(patt),
            C.byref(found))
        assert found.value == a["nnz_to_copy"]
    else:
        found = C.c_int(-1); patt = vec(a["pattern"])
        run("hiopamd_sp_set_submatrix_to_constant_diag_w_pattern", T[0], T[1], T[2], a["scalar"], a["row_st"], a["col_st"], a["nnz_st"], patt.size, D(patt),
            a["rowpattern"], C.byref(found))
        assert found.value == a["nnz_to_copy"]
    return {"W": image(T, m, n)}


# ------------------------------------------------------------------ oracle
def run_oracle(case):
    from oracle import hiop_oracle as ho
    op, a = case["op"], case["args"]
    g = lambda k: vec(a[k])
    if op == "setToConstant":
        y = g("y"); y[:] = a["c"]; return {"y": y}
    if op == "setToZero":
        y = g("y"); y[:] = 0.0; return {"y": y}
    if op == "copy_from_indexes":
        y = g("y"); ho.copy_from_indexes(y, g("src"), lin(a["idx"])); return {"y": y}
    if op == "copyFromStarting":
        y = g("y"); ho.copy_from_starting(y, a["start"], g("src")); return {"y": y}
    if op == "startingAtCopyFromStartingAt":
        y = g("y"); ho.starting_at_copy_from_starting_at(y, a["start_dest"], g("src"), a["start_src"]); return {"y": y}
    if op == "copyTo":
        dest = g("dest"); dest[:] = g("x"); return {"dest": dest}
    if op == "copyToStarting":
        dest = g("dest"); ho.copy_to_starting(g("x"), dest, a["start"]); return {"dest": dest}
    if op == "copyToStartingAt_w_pattern":
        dest = g("dest"); nnz = ho.copy_to_starting_at_w_pattern(g("x"), dest, a["start"], g("select")); return {"dest": dest, "nnz": nnz}
    if op == "copy_from_two_vec_w_pattern":
        y = g("y"); ho.copy_from_two_vec_w_pattern(y, g("c"), lin(a["c_map"]), g("d"), lin(a["d_map"])); return {"y": y}
    if op == "copy_to_two_vec_w_pattern":
        c, d = g("c"), g("d"); ho.copy_to_two_vec_w_pattern(g("y"), c, lin(a["c_map"]), d, lin(a["d_map"])); return {"c": c, "d": d}
    if op == "startingAtCopyToStartingAt":
        dest = g("dest"); ho.starting_at_copy_to_starting_at(g("src"), a["start_src"], dest, a["start_dest"], a["num"]); return {"dest": dest}
    if op == "isnan":
        return {"value": ho.isnan_local(g("x"))}
    if op == "isinf":
        return {"value": ho.isinf_local(g("x"))}
    if op == "isfinite":
        return {"value": ho.isfinite_local(g("x"))}
    if op == "mat_setToZero":
        A = mat(a["A"]); A[:] = 0.0; return {"A": A}
    if op == "mat_setToConstant":
        A = mat(a["A"]); A[:] = a["c"]; return {"A": A}
    if op in ("mat_copyFrom", "mat_copy_to"):
        dst = mat(a["dst"]); dst[:] = mat(a["src"]); return {"dst": dst}
    if op == "mat_appendRow":      # hiopMatrixDenseRowMajor.cpp:150-161: the row behind the last one, then m_local_++
        A = mat(a["A"]); A2 = np.vstack([A, g("vec")[None, :]]); return {"A": A2, "m": A2.shape[0]}
    if op == "mat_replaceRow":     # :263-268
        A = mat(a["A"]); A[a["row"], :] = g("vec"); return {"A": A}
    if op == "mat_getRow":         # :270-276
        A = mat(a["A"]); v = g("vec"); v[:] = A[a["row"], :]; return {"vec": v}
    if op == "mat_overwriteUpperTriangleWithLower":   # :1057-1064
        A = mat(a["A"]); il = np.tril_indices(A.shape[0], -1); A[il[1], il[0]] = A[il]; return {"A": A}
    if op == "mat_overwriteLowerTriangleWithUpper":   # :1066-1073
        A = mat(a["A"]); ho.symmetrize(A); return {"A": A}
    if op == "setToConstant_w_patternSelect":
        y = g("y"); ho.set_to_constant_w_pattern(y, a["c"], g("select")); return {"y": y}
    if op == "copyFrom":
        y = g("y"); y[:] = g("x"); return {"y": y}
    if op == "selectPattern":
        y = g("y"); y[g("select") == 0.0] = 0.0; return {"y": y}
    if op == "scale":
        return {"y": g("y") * a["c"]}
    if op == "componentMult":
        return {"y": g("y") * g("x")}
    if op == "componentDiv":
        return {"y": g("y") / g("x")}
    if op == "componentDiv_w_selectPattern":
        y = g("y"); ho.component_div_w_pattern(y, g("x"), g("select")); return {"y": y}
    if op == "component_min_c":
        return {"y": np.minimum(g("y"), a["c"])}
    if op == "component_min_v":
        return {"y": np.minimum(g("y"), g("x"))}
    if op == "component_max_c":
        return {"y": np.maximum(g("y"), a["c"])}
    if op == "component_max_v":
        return {"y": np.maximum(g("y"), g("x"))}
    if op == "component_abs":
        return {"y": np.abs(g("y"))}
    if op == "component_sqrt":
        return {"y": np.sqrt(g("y"))}
    if op == "component_sgn":
        y = g("y"); ho.component_sgn(y); return {"y": y}
    if op == "axpy":
        return {"y": g("y") + a["alpha"] * g("x")}
    if op == "axzpy":
        y = g("y"); ho.axzpy(y, a["alpha"], g("x"), g("z")); return {"y": y}
    if op == "axdzpy":
        y = g("y"); ho.axdzpy(y, a["alpha"], g("x"), g("z")); return {"y": y}
    if op == "axdzpy_w_pattern":
        y = g("y")
        with np.errstate(divide="ignore"):
            ho.axdzpy_w_pattern(y, a["alpha"], g("x"), g("z"), g("select"))
        return {"y": y}
    if op == "addConstant":
        return {"y": g("y") + a["c"]}
    if op == "addConstant_w_patternSelect":
        y = g("y"); y[g("select") == 1.0] += a["c"]; return {"y": y}
    if op == "negate":
        return {"y": -g("y")}
    if op == "invert":
        return {"y": 1.0 / g("y")}
    if op == "addLogBarrierGrad":
        y = g("y"); ho.add_log_barrier_grad(y, a["alpha"], g("x"), g("select")); return {"y": y}
    if op == "addLinearDampingTerm":
        y = g("y"); ho.add_linear_damping_term(y, g("ixl"), g("ixr"), a["alpha"], a["ct"]); return {"y": y}
    if op == "adjustDuals_plh":
        z = g("z"); ho.adjust_duals_plh(z, g("x"), g("select"), a["mu"], a["kappa"]); return {"z": z}
    if op == "onenorm":
        return {"value": ho.onenorm(g("x"))}
    if op == "twonorm":
        return {"value": float(np.linalg.norm(g("x")))}
    if op == "infnorm":
        return {"value": ho.infnorm(g("x"))}
    if op == "dotProductWith":
        return {"value": float(g("x") @ g("y"))}
    if op == "logBarrier_local":
        return {"value": ho.log_barrier(g("x"), g("select"))}
    if op == "sum_local":
        return {"value": float(g("x").sum())}
    if op == "linearDampingTerm_local":
        return {"value": ho.linear_damping_term(g("x"), g("ixl"), g("ixr"), a["mu"], a["kappa_d"])}
    if op == "allPositive":
        return {"value": int(not np.any(g("x") <= 0))}
    if op == "allPositive_w_patternSelect":
        return {"value": ho.all_positive_w_pattern(g("x"), g("select"))}
    if op == "min":
        return {"value": float(g("x").min())}
    if op == "min_w_pattern":
        return {"value": ho.vmin_w_pattern(g("x"), g("select"))}
    if op == "fractionToTheBdry_local":
        return {"value": ho.fraction_to_the_bdry(g("x"), g("d"), a["tau"])}
    if op == "fractionToTheBdry_w_pattern_local":
        return {"value": ho.fraction_to_the_bdry_w_pattern(g("x"), g("d"), a["tau"], g("select"))}
    if op == "matchesPattern":
        return {"value": ho.matches_pattern(g("x"), g("select"))}
    if op == "projectIntoBounds_local":
        x = g("x"); ok = ho.project_into_bounds(x, g("xl"), g("ixl"), g("xu"), g("ixu"), a["kappa1"], a["kappa2"])
        return {"ok": int(ok), "x": x}
    if op == "mat_timesVec":
        y = g("y"); ho.times_vec(mat(a["A"]), a["beta"], y, a["alpha"], g("x")); return {"y": y}
    if op == "mat_transTimesVec":
        y = g("y"); ho.trans_times_vec(mat(a["A"]), a["beta"], y, a["alpha"], g("x")); return {"y": y}
    if op == "mat_addSubDiagonal":
        A = mat(a["A"]); ho.add_sub_diagonal(A, a["start"], a["alpha"], g("d"), a["src_start"], a["num"]); return {"A": A}
    if op == "mat_addSubDiagonal_const":
        A = mat(a["A"]); i = np.arange(a["num"]) + a["start"]; A[i, i] += a["c"]; return {"A": A}
    if op == "mat_addMatrix":
        return {"A": mat(a["A"]) + a["alpha"] * mat(a["B"])}
    if op == "mat_transAddToSymDenseMatrixUpperTriangle":
        W = mat(a["W"]); ho.trans_add_to_sym_upper(mat(a["A"]), a["row_start"], a["col_start"], a["alpha"], W); return {"W": W}
    if op == "mat_addUpperTriangleToSymDenseMatrixUpperTriangle":
        W = mat(a["W"]); ho.add_upper_to_sym_upper(mat(a["A"]), a["diag_start"], a["alpha"], W); return {"W": W}
    if op == "mat_timesMat":
        W = mat(a["W"]); ho.times_mat(mat(a["A"]), a["beta"], W, a["alpha"], mat(a["X"])); return {"W": W}
    if op == "mat_transTimesMat":
        W = mat(a["W"]); ho.trans_times_mat(mat(a["A"]), a["beta"], W, a["alpha"], mat(a["X"])); return {"W": W}
    if op == "mat_timesMatTrans":
        W = mat(a["W"]); ho.times_mat_trans(mat(a["A"]), a["beta"], W, a["alpha"], mat(a["X"])); return {"W": W}
    if op == "mat_addDiagonal":
        A = mat(a["A"]); ho.add_diagonal(A, a["alpha"], g("d")); return {"A": A}
    if op == "mat_addDiagonal_const":
        A = mat(a["A"]); ho.add_diagonal(A, a["alpha"]); return {"A": A}
    if op == "mat_maxAbsValue":
        return {"value": ho.max_abs_value(mat(a["A"]))}
    if op == "mat_row_max_abs_value":
        return {"y": ho.row_max_abs_value(mat(a["A"]))}
    if op == "mat_scale_row":
        A = mat(a["A"]); ho.scale_row(A, g("x"), a["inv"]); return {"A": A}
    if op == "mat_isFinite":
        return {"ok": int(np.all(np.isfinite(mat(a["A"]))))}
    if op == "mat_copyRowsFrom":
        dst = mat(a["dst"]); ho.copy_rows_from(dst, mat(a["src"]), a["num_rows"], a["row_dest"]); return {"dst": dst}
    if op == "mat_copyRowsFromSelect":
        dst = mat(a["dst"]); ho.copy_rows_from_select(dst, mat(a["src"]), a["rows"]); return {"dst": dst}
    if op == "mat_copyBlockFromMatrix":
        dst = mat(a["dst"]); ho.copy_block_from_matrix(dst, a["i0"], a["j0"], mat(a["src"])); return {"dst": dst}
    if op == "mat_copyFromMatrixBlock":
        dst = mat(a["dst"]); ho.copy_from_matrix_block(dst, mat(a["src"]), a["i0"], a["j0"]); return {"dst": dst}
    if op == "mat_shiftRows":
        A = mat(a["A"]); ho.shift_rows(A, a["shift"]); return {"A": A}
    if op == "mat_symmetrize":
        A = mat(a["A"]); ho.symmetrize(A); return {"A": A}
    if op in _SP_ASSEMBLY:
        return _sp_assembly_oracle(op, a)
    if op == "sp_maxAbsValue":
        return {"value": float(np.max(np.abs(coo(a["A"])[2])))}
    if op == "sp_row_max_abs_value":
        i, j, v, m, n = coo(a["A"]); return {"y": ho.sp_row_max_abs(m, i, v)}
    if op == "sp_scale_row":
        i, j, v, m, n = coo(a["A"]); ho.sp_scale_rows(i, v, g("x"), a["inv"]); return {"val": v}
    if op == "sp_isFinite":
        return {"ok": int(np.all(np.isfinite(coo(a["A"])[2])))}
    if op == "sp_timesMatTrans":
        i, j, v, m, n = coo(a["A"]); i2, j2, v2, m2, _ = coo(a["B"]); W = mat(a["W"])
        ho.sp_times_mat_trans(m, m2, n, i, j, v, i2, j2, v2, a["beta"], W, a["alpha"]); return {"W": W}
    if op == "sp_addMDinvNtransToSymDeMatUTri":
        i, j, v, m, n = coo(a["A"]); i2, j2, v2, m2, _ = coo(a["B"]); W = mat(a["W"])
        ho.sp_add_MDinvNtrans(m, m2, n, i, j, v, i2, j2, v2, a["i_offset"], a["j_offset"], a["alpha"], g("D"), W); return {"W": W}
    if op == "sp_copy_to":
        i, j, v, m, n = coo(a["A"]); return {"W": ho.sp_copy_to_dense(m, n, i, j, v)}
    if op == "spsym_timesVec":
        i, j, v, m, n = coo(a["A"]); y = g("y"); ho.spsym_times_vec(n, i, j, v, a["beta"], y, a["alpha"], g("x")); return {"y": y}
    if op == "spsym_addUpperTriangleToSymDenseMatrixUpperTriangle":
        i, j, v, m, n = coo(a["A"]); W = mat(a["W"]); ho.spsym_add_upper_to_sym_upper(i, j, v, a["diag_start"], a["alpha"], W); return {"W": W}
    if op == "sp_addMDinvMtransToDiagBlockOfSymDeMatUTri":
        i, j, v, m, n = coo(a["A"]); W = mat(a["W"])
        ho.sp_add_MDinvMtrans_rowmerge(m, i, j, v, a["offset"], a["alpha"], g("D"), W); return {"W": W}
    if op == "sp_timesVec":
        i, j, v, m, n = coo(a["A"]); y = g("y"); ho.sp_times_vec(m, i, j, v, a["beta"], y, a["alpha"], g("x")); return {"y": y}
    if op == "sp_transTimesVec":
        i, j, v, m, n = coo(a["A"]); y = g("y"); ho.sp_trans_times_vec(n, i, j, v, a["beta"], y, a["alpha"], g("x")); return {"y": y}
    if op == "spsym_startingAtAddSubDiagonalToStartingAt":
        i, j, v, m, n = coo(a["A"]); y = g("y"); ho.spsym_add_diag_to_vec(i, j, v, a["alpha"], y, 0); return {"y": y}
    raise KeyError(op)


# ------------------------------------------------------------------ HIP (C ABI)
_EW = {
    "setToConstant": ("hiopamd_vec_set_to_constant", lambda a, D, g: [a["c"]]),
    "setToConstant_w_patternSelect": ("hiopamd_vec_set_to_constant_w_pattern", lambda a, D, g: [a["c"], D(g("select"))]),
    "copyFrom": ("hiopamd_vec_copy", lambda a, D, g: [D(g("x"))]),
    "selectPattern": ("hiopamd_vec_select_pattern", lambda a, D, g: [D(g("select"))]),
    "scale": ("hiopamd_vec_scale", lambda a, D, g: [a["c"]]),
    "componentMult": ("hiopamd_vec_component_mult", lambda a, D, g: [D(g("x"))]),
    "componentDiv": ("hiopamd_vec_component_div", lambda a, D, g: [D(g("x"))]),
    "componentDiv_w_selectPattern": ("hiopamd_vec_component_div_w_pattern", lambda a, D, g: [D(g("x")), D(g("select"))]),
    "component_min_c": ("hiopamd_vec_component_min_c", lambda a, D, g: [a["c"]]),
    "component_min_v": ("hiopamd_vec_component_min_v", lambda a, D, g: [D(g("x"))]),
    "component_max_c": ("hiopamd_vec_component_max_c", lambda a, D, g: [a["c"]]),
    "component_max_v": ("hiopamd_vec_component_max_v", lambda a, D, g: [D(g("x"))]),
    "component_abs": ("hiopamd_vec_component_abs", lambda a, D, g: []),
    "component_sqrt": ("hiopamd_vec_component_sqrt", lambda a, D, g: []),
    "component_sgn": ("hiopamd_vec_component_sgn", lambda a, D, g: []),
    "axpy": ("hiopamd_vec_axpy", lambda a, D, g: [a["alpha"], D(g("x"))]),
    "axzpy": ("hiopamd_vec_axzpy", lambda a, D, g: [a["alpha"], D(g("x")), D(g("z"))]),
    "axdzpy": ("hiopamd_vec_axdzpy", lambda a, D, g: [a["alpha"], D(g("x")), D(g("z"))]),
    "axdzpy_w_pattern": ("hiopamd_vec_axdzpy_w_pattern", lambda a, D, g: [a["alpha"], D(g("x")), D(g("z")), D(g("select"))]),
    "addConstant": ("hiopamd_vec_add_constant", lambda a, D, g: [a["c"]]),
    "addConstant_w_patternSelect": ("hiopamd_vec_add_constant_w_pattern", lambda a, D, g: [a["c"], D(g("select"))]),
    "negate": ("hiopamd_vec_negate", lambda a, D, g: []),
    "invert": ("hiopamd_vec_invert", lambda a, D, g: []),
    "addLogBarrierGrad": ("hiopamd_vec_add_log_barrier_grad", lambda a, D, g: [a["alpha"], D(g("x")), D(g("select"))]),
    "addLinearDampingTerm": ("hiopamd_vec_add_linear_damping_term", lambda a, D, g: [D(g("ixl")), D(g("ixr")), a["alpha"], a["ct"]]),
}
_RED_D = {
    "onenorm": ("hiopamd_vec_onenorm", lambda a, D, g: [D(g("x"))]),
    "twonorm": ("hiopamd_vec_twonorm", lambda a, D, g: [D(g("x"))]),
    "infnorm": ("hiopamd_vec_infnorm", lambda a, D, g: [D(g("x"))]),
    "dotProductWith": ("hiopamd_vec_dot", lambda a, D, g: [D(g("x")), D(g("y"))]),
    "logBarrier_local": ("hiopamd_vec_log_barrier", lambda a, D, g: [D(g("x")), D(g("select"))]),
    "sum_local": ("hiopamd_vec_sum", lambda a, D, g: [D(g("x"))]),
    "linearDampingTerm_local": ("hiopamd_vec_linear_damping_term", lambda a, D, g: [D(g("x")), D(g("ixl")), D(g("ixr")), a["mu"], a["kappa_d"]]),
    "min": ("hiopamd_vec_min", lambda a, D, g: [D(g("x"))]),
    "min_w_pattern": ("hiopamd_vec_min_w_pattern", lambda a, D, g: [D(g("x")), D(g("select"))]),
    "fractionToTheBdry_local": ("hiopamd_vec_fraction_to_the_bdry", lambda a, D, g: [D(g("x")), D(g("d")), a["tau"]]),
    "fractionToTheBdry_w_pattern_local": ("hiopamd_vec_fraction_to_the_bdry_w_pattern", lambda a, D, g: [D(g("x")), D(g("d")), a["tau"], D(g("select"))]),
}
_RED_I = {
    "allPositive": ("hiopamd_vec_all_positive", lambda a, D, g: [D(g("x"))]),
    "allPositive_w_patternSelect": ("hiopamd_vec_all_positive_w_pattern", lambda a, D, g: [D(g("x")), D(g("select"))]),
    "matchesPattern": ("hiopamd_vec_matches_pattern", lambda a, D, g: [D(g("x")), D(g("select"))]),
}


def run_gpu(ctx, case):
    import torch
    op, a = case["op"], case["args"]
    g = lambda k: vec(a[k])
    D = lambda arr, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(arr)).to(dt).cuda()

    def run(name, *args):
        torch.cuda.synchronize(); ctx.call(name, *args); ctx.sync()
    I32 = lambda arr: D(arr, torch.int32)
    P = lambda t: C.c_void_p(t.data_ptr() if t.numel() else 0)
    if op == "setToZero":
        y = D(g("y")); run("hiopamd_vec_set_to_constant", y.numel(), y, 0.0); return {"y": y.cpu().numpy()}
    if op == "copy_from_indexes":
        y = D(g("y")); run("hiopamd_vec_copy_from_indexes", y.numel(), y, D(g("src")), I32(lin(a["idx"]))); return {"y": y.cpu().numpy()}
    if op == "copyFromStarting":      # dest[start : start + n_src] = src: the vector adapter's recipe (startingAtCopyFromStartingAt with the count of src)
        y, src = D(g("y")), D(g("src"))
        run("hiopamd_vec_starting_at_copy_to_starting_at", P(src), src.numel(), 0, y, y.numel(), a["start"], src.numel()); return {"y": y.cpu().numpy()}
    if op == "startingAtCopyFromStartingAt":
        y, src = D(g("y")), D(g("src"))
        run("hiopamd_vec_starting_at_copy_from_starting_at", y, y.numel(), a["start_dest"], src, src.numel(), a["start_src"]); return {"y": y.cpu().numpy()}
    if op == "copyTo":
        dest = D(g("dest")); run("hiopamd_vec_copy", dest.numel(), dest, D(g("x"))); return {"dest": dest.cpu().numpy()}
    if op == "copyToStarting":
        x, dest = D(g("x")), D(g("dest"))
        run("hiopamd_vec_starting_at_copy_to_starting_at", P(x), x.numel(), 0, dest, dest.numel(), a["start"], x.numel()); return {"dest": dest.cpu().numpy()}
    if op == "copyToStartingAt_w_pattern":
        x, dest = D(g("x")), D(g("dest")); nnz = C.c_int64(-1)
        run("hiopamd_vec_copy_to_starting_at_w_pattern", x.numel(), x, dest, a["start"], D(g("select")), C.byref(nnz))
        return {"dest": dest.cpu().numpy(), "nnz": nnz.value}
    if op == "copy_from_two_vec_w_pattern":
        y, c, d = D(g("y")), D(g("c")), D(g("d"))
        run("hiopamd_vec_copy_from_two_vec_w_pattern", y, c, I32(lin(a["c_map"])), c.numel(), d, I32(lin(a["d_map"])), d.numel()); return {"y": y.cpu().numpy()}
    if op == "copy_to_two_vec_w_pattern":
        y, c, d = D(g("y")), D(g("c")), D(g("d"))
        run("hiopamd_vec_copy_to_two_vec_w_pattern", y, c, I32(lin(a["c_map"])), c.numel(), d, I32(lin(a["d_map"])), d.numel())
        return {"c": c.cpu().numpy(), "d": d.cpu().numpy()}
    if op == "startingAtCopyToStartingAt":
        src, dest = D(g("src")), D(g("dest"))
        run("hiopamd_vec_starting_at_copy_to_starting_at", P(src), src.numel(), a["start_src"], dest, dest.numel(), a["start_dest"], a["num"])
        return {"dest": dest.cpu().numpy()}
    if op in ("isnan", "isinf", "isfinite"):
        x = D(g("x")); torch.cuda.synchronize(); return {"value": ctx.reduce_int("hiopamd_vec_" + op, x.numel(), x)}
    if op in ("mat_setToZero", "mat_setToConstant"):
        A = D(mat(a["A"])); run("hiopamd_mat_set_to_constant", A.shape[0], A.shape[1], A, A.shape[1], a.get("c", 0.0)); return {"A": A.cpu().numpy()}
    if op in ("mat_copyFrom", "mat_copy_to"):   # (copy_to: the adapter's device-to-device copy of the whole storage)
        dst, src = D(mat(a["dst"])), D(mat(a["src"]))
        if op == "mat_copyFrom":
            run("hiopamd_mat_copy_block", src.shape[0], src.shape[1], dst, dst.shape[1], src, src.shape[1])
        else:
            run("hiopamd_copy_d2d", dst, src, src.numel() * 8)
        return {"dst": dst.cpu().numpy()}
    if op == "mat_appendRow":     # storage with room for one more row (the reference allocates max_rows): adapters/hiopMatrixDenseHipNative.cpp appendRow
        A0 = mat(a["A"]); m, n = A0.shape
        S = D(np.vstack([A0, np.full((1, n), -7.0)])); v = D(g("vec"))
        run("hiopamd_copy_d2d", C.c_void_p(S.data_ptr() + 8 * m * n), v, 8 * n); return {"A": S.cpu().numpy(), "m": m + 1}
    if op == "mat_replaceRow":
        A = D(mat(a["A"])); v = D(g("vec")); n = A.shape[1]
        run("hiopamd_copy_d2d", C.c_void_p(A.data_ptr() + 8 * a["row"] * n), v, 8 * n); return {"A": A.cpu().numpy()}
    if op == "mat_getRow":
        A = D(mat(a["A"])); v = D(g("vec")); n = A.shape[1]
        run("hiopamd_copy_d2d", v, C.c_void_p(A.data_ptr() + 8 * a["row"] * n), 8 * n); return {"vec": v.cpu().numpy()}
    if op == "mat_overwriteLowerTriangleWithUpper":
        A = D(mat(a["A"])); run("hiopamd_mat_symmetrize", A.shape[0], A, A.shape[1]); return {"A": A.cpu().numpy()}
    if op == "mat_overwriteUpperTriangleWithLower":   # no device kernel: a DEEPCHECKS-only method of the reference, done on the host by the adapter
        import pytest
        pytest.skip("overwriteUpperTriangleWithLower has no device entry point (the adapter does it on the host)")
    if op in _EW:
        name, extra = _EW[op]
        y = D(g("y")); run(name, y.numel(), y, *extra(a, D, g)); return {"y": y.cpu().numpy()}
    if op in _RED_D:
        name, extra = _RED_D[op]
        args = extra(a, D, g); torch.cuda.synchronize()
        return {"value": ctx.reduce_double(name, args[0].numel(), *args)}
    if op in _RED_I:
        name, extra = _RED_I[op]
        args = extra(a, D, g); torch.cuda.synchronize()
        return {"value": ctx.reduce_int(name, args[0].numel(), *args)}
    if op == "adjustDuals_plh":
        z = D(g("z")); run("hiopamd_vec_adjust_duals_plh", z.numel(), z, D(g("x")), D(g("select")), a["mu"], a["kappa"])
        return {"z": z.cpu().numpy()}
    if op == "projectIntoBounds_local":
        x = D(g("x")); ok = C.c_int(-1)
        run("hiopamd_vec_project_into_bounds", x.numel(), x, D(g("xl")), D(g("ixl")), D(g("xu")), D(g("ixu")), a["kappa1"],
            a["kappa2"], C.byref(ok))
        return {"ok": ok.value, "x": x.cpu().numpy()}
    if op == "mat_timesVec":
        A = mat(a["A"]); y = D(g("y")); run("hiopamd_mat_times_vec", A.shape[0], A.shape[1], D(A), A.shape[1], a["beta"], y, a["alpha"], D(g("x")))
        return {"y": y.cpu().numpy()}
    if op == "mat_transTimesVec":
        A = mat(a["A"]); y = D(g("y")); run("hiopamd_mat_trans_times_vec", A.shape[0], A.shape[1], D(A), A.shape[1], a["beta"], y, a["alpha"], D(g("x")))
        return {"y": y.cpu().numpy()}
    if op == "mat_addSubDiagonal":
        A = D(mat(a["A"])); run("hiopamd_mat_add_sub_diagonal", A, A.shape[1], a["start"], a["alpha"], D(g("d")), a["src_start"], a["num"])
        return {"A": A.cpu().numpy()}
    if op == "mat_addSubDiagonal_const":
        A = D(mat(a["A"])); run("hiopamd_mat_add_sub_diagonal_const", A, A.shape[1], a["start"], a["num"], a["c"])
        return {"A": A.cpu().numpy()}
    if op == "mat_addMatrix":
        A = D(mat(a["A"])); B = D(mat(a["B"])); run("hiopamd_mat_add_matrix", A.shape[0], A.shape[1], A, A.shape[1], a["alpha"], B, B.shape[1])
        return {"A": A.cpu().numpy()}
    if op == "mat_transAddToSymDenseMatrixUpperTriangle":
        A = mat(a["A"]); W = D(mat(a["W"]))
        run("hiopamd_mat_trans_add_to_sym_upper", A.shape[0], A.shape[1], D(A), A.shape[1], a["row_start"], a["col_start"], a["alpha"], W, W.shape[1])
        return {"W": W.cpu().numpy()}
    if op == "mat_addUpperTriangleToSymDenseMatrixUpperTriangle":
        A = mat(a["A"]); W = D(mat(a["W"]))
        run("hiopamd_mat_add_upper_to_sym_upper", A.shape[0], D(A), A.shape[1], a["diag_start"], a["alpha"], W, W.shape[1])
        return {"W": W.cpu().numpy()}
    if op in ("mat_timesMat", "mat_transTimesMat", "mat_timesMatTrans"):
        A, X, W = mat(a["A"]), mat(a["X"]), D(mat(a["W"]))
        if op == "mat_timesMat":
            run("hiopamd_mat_times_mat", A.shape[0], A.shape[1], X.shape[1], D(A), A.shape[1], a["beta"], W, W.shape[1], a["alpha"], D(X), X.shape[1])
        elif op == "mat_transTimesMat":
            run("hiopamd_mat_trans_times_mat", A.shape[0], A.shape[1], X.shape[1], D(A), A.shape[1], a["beta"], W, W.shape[1], a["alpha"], D(X), X.shape[1])
        else:
            run("hiopamd_mat_times_mat_trans", A.shape[0], A.shape[1], X.shape[0], D(A), A.shape[1], a["beta"], W, W.shape[1], a["alpha"], D(X), X.shape[1])
        return {"W": W.cpu().numpy()}
    if op == "mat_addDiagonal":
        A = D(mat(a["A"])); run("hiopamd_mat_add_diagonal_vec", A.shape[0], A, A.shape[1], a["alpha"], D(g("d"))); return {"A": A.cpu().numpy()}
    if op == "mat_addDiagonal_const":
        A = D(mat(a["A"])); run("hiopamd_mat_add_diagonal_const", A.shape[0], A, A.shape[1], a["alpha"]); return {"A": A.cpu().numpy()}
    if op == "mat_maxAbsValue":
        A = D(mat(a["A"])); torch.cuda.synchronize()
        return {"value": ctx.reduce_double("hiopamd_mat_max_abs", A.shape[0], A.shape[1], A, A.shape[1])}
    if op == "mat_row_max_abs_value":
        A = D(mat(a["A"])); y = D(np.zeros(A.shape[0])); run("hiopamd_mat_row_max_abs", A.shape[0], A.shape[1], A, A.shape[1], y)
        return {"y": y.cpu().numpy()}
    if op == "mat_scale_row":
        A = D(mat(a["A"])); run("hiopamd_mat_scale_rows", A.shape[0], A.shape[1], A, A.shape[1], D(g("x")), a["inv"]); return {"A": A.cpu().numpy()}
    if op == "mat_isFinite":
        A = D(mat(a["A"])); torch.cuda.synchronize()
        return {"ok": ctx.reduce_int("hiopamd_mat_is_finite", A.shape[0], A.shape[1], A, A.shape[1])}
    if op == "mat_copyRowsFrom":
        dst, src = D(mat(a["dst"])), D(mat(a["src"]))
        run("hiopamd_mat_copy_rows_from", a["num_rows"], dst.shape[1], dst, dst.shape[1], a["row_dest"], src, src.shape[1]); return {"dst": dst.cpu().numpy()}
    if op == "mat_copyRowsFromSelect":
        dst, src = D(mat(a["dst"])), D(mat(a["src"]))
        run("hiopamd_mat_copy_rows_from_idx", len(a["rows"]), dst.shape[1], dst, dst.shape[1], src, src.shape[1], D(np.array(a["rows"], np.int32), torch.int32))
        return {"dst": dst.cpu().numpy()}
    if op == "mat_copyBlockFromMatrix":
        dst, src = D(mat(a["dst"])), D(mat(a["src"]))
        sub = dst[a["i0"]:, a["j0"]:]      # (a view: its data pointer is the block's first element, ld = dst's)
        torch.cuda.synchronize()
        ctx.call("hiopamd_mat_copy_block", src.shape[0], src.shape[1], C.c_void_p(sub.data_ptr()), dst.shape[1], src, src.shape[1]); ctx.sync()
        return {"dst": dst.cpu().numpy()}
    if op == "mat_copyFromMatrixBlock":
        dst, src = D(mat(a["dst"])), D(mat(a["src"]))
        sub = src[a["i0"]:, a["j0"]:]
        torch.cuda.synchronize()
        ctx.call("hiopamd_mat_copy_block", dst.shape[0], dst.shape[1], dst, dst.shape[1], C.c_void_p(sub.data_ptr()), src.shape[1]); ctx.sync()
        return {"dst": dst.cpu().numpy()}
    if op == "mat_shiftRows":
        A = D(mat(a["A"])); run("hiopamd_mat_shift_rows", A.shape[0], A.shape[1], A, A.shape[1], a["shift"]); return {"A": A.cpu().numpy()}
    if op == "mat_symmetrize":
        A = D(mat(a["A"])); run("hiopamd_mat_symmetrize", A.shape[0], A, A.shape[1]); return {"A": A.cpu().numpy()}
    if op in _SP_ASSEMBLY:
        return _sp_assembly_gpu(ctx, op, a)
    if op.startswith("sp"):
        i, j, v, m, n = coo(a["A"])
        id_, jd, vd = D(i, torch.int32), D(j, torch.int32), D(v)
        if op == "sp_maxAbsValue":
            torch.cuda.synchronize(); return {"value": ctx.reduce_double("hiopamd_vec_infnorm", v.size, vd)}
        if op == "sp_row_max_abs_value":
            y = D(np.full(m, 9.0)); run("hiopamd_sp_row_max_abs", m, v.size, id_, vd, y); return {"y": y.cpu().numpy()}
        if op == "sp_scale_row":
            run("hiopamd_sp_scale_rows", v.size, id_, vd, D(g("x")), a["inv"]); return {"val": vd.cpu().numpy()}
        if op == "sp_isFinite":
            torch.cuda.synchronize(); return {"ok": ctx.reduce_int("hiopamd_vec_isfinite", v.size, vd)}
        if op in ("sp_timesMatTrans", "sp_addMDinvNtransToSymDeMatUTri"):
            from hiop_amd._lib import lib
            i2, j2, v2, m2, _ = coo(a["B"])
            L = lib(); plan = C.c_void_p()
            assert L.hiopamd_sp_plan_create(C.byref(plan), m, m2, n, v.size, i.ctypes.data, j.ctypes.data, v2.size, i2.ctypes.data, j2.ctypes.data, 0) == 0
            W = D(mat(a["W"]))
            if op == "sp_timesMatTrans":      # = scale W, then the row-build with D = ones (what the HiOp-side adapter does)
                run("hiopamd_vec_scale", W.numel(), W, a["beta"])
                run("hiopamd_sp_add_MDinvNt", plan, vd, D(v2), D(np.ones(n)), a["alpha"], W, W.shape[1], 0, 0)
            else:
                run("hiopamd_sp_add_MDinvNt", plan, vd, D(v2), D(g("D")), a["alpha"], W, W.shape[1], a["i_offset"], a["j_offset"])
            L.hiopamd_sp_plan_destroy(plan)
            return {"W": W.cpu().numpy()}
        if op == "sp_copy_to":
            W = D(mat(a["W"])); run("hiopamd_sp_copy_to_dense", m, n, v.size, id_, jd, vd, W, W.shape[1]); return {"W": W.cpu().numpy()}
        if op == "spsym_timesVec":
            y = D(g("y")); run("hiopamd_spsym_times_vec", n, v.size, id_, jd, vd, a["beta"], y, a["alpha"], D(g("x"))); return {"y": y.cpu().numpy()}
        if op == "spsym_addUpperTriangleToSymDenseMatrixUpperTriangle":
            W = D(mat(a["W"])); run("hiopamd_spsym_add_upper_to_sym_upper", v.size, id_, jd, vd, a["diag_start"], a["alpha"], W, W.shape[1]); return {"W": W.cpu().numpy()}
        if op == "sp_timesVec":
            y = D(g("y")); run("hiopamd_sp_times_vec", m, n, v.size, id_, jd, vd, a["beta"], y, a["alpha"], D(g("x"))); return {"y": y.cpu().numpy()}
        if op == "sp_transTimesVec":
            y = D(g("y")); run("hiopamd_sp_trans_times_vec", m, n, v.size, id_, jd, vd, a["beta"], y, a["alpha"], D(g("x"))); return {"y": y.cpu().numpy()}
        if op == "spsym_startingAtAddSubDiagonalToStartingAt":
            y = D(g("y")); run("hiopamd_spsym_add_diag_to_vec", v.size, id_, jd, vd, a["alpha"], y, 0, y.numel(), 0, y.numel()); return {"y": y.cpu().numpy()}
        if op == "sp_addMDinvMtransToDiagBlockOfSymDeMatUTri":
            from hiop_amd._lib import lib
            L = lib(); plan = C.c_void_p()
            assert L.hiopamd_sp_plan_create(C.byref(plan), m, m, n, v.size, i.ctypes.data, j.ctypes.data, v.size, i.ctypes.data, j.ctypes.data, 1) == 0
            W = D(mat(a["W"]))
            run("hiopamd_sp_add_MDinvNt", plan, vd, vd, D(g("D")), a["alpha"], W, W.shape[1], a["offset"], a["offset"])
            L.hiopamd_sp_plan_destroy(plan)
            return {"W": W.cpu().numpy()}
    raise KeyError(op)
