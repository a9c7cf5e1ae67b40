"""GPU parity of the hiopVector kernels (HIP, through the C ABI) against the numpy oracle.

Mirrors the structure of the reference's tests/LinAlg/vectorTests.hpp (one test per public method,
sizes around Nlocal=1000 of tests/testVector.cpp:235) plus ragged / empty sizes.
Tolerances: element-wise results are bit-exact except where an FMA contraction can differ by 1 ulp
(rtol 4*eps); reductions are compared at rtol 1e-13 (different association)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import hiop_oracle as ho

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 63, 64, 65, 257, 1000, 100003]
EPS = np.finfo(np.float64).eps


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def mk(n, seed, lo=-1.0, hi=1.0):
    return rng(seed).uniform(lo, hi, n)


def pattern(n, seed):
    return (rng(seed).uniform(0, 1, n) < 0.6).astype(np.float64)


def D(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def run(ctx, name, *args):
    torch.cuda.synchronize()
    ctx.call(name, *args)
    ctx.sync()


EW_CASES = {
    # name: (builder(n) -> (abi_args(list after n), numpy reference function producing expected y))
}


def _ew(ctx, n, name, y, extra_dev, expected, rtol=4 * EPS):
    yd = D(y)
    run(ctx, name, n, yd, *extra_dev)
    got = yd.cpu().numpy()
    np.testing.assert_allclose(got, expected, rtol=rtol, atol=1e-16)


@pytest.mark.parametrize("n", SIZES)
def test_elementwise_family(ctx, n):
    y, x, z = mk(n, 1), mk(n, 2, 0.5, 2.0), mk(n, 3, 0.5, 2.0)
    s = pattern(n, 4)
    xd, zd, sd = D(x), D(z), D(s)
    _ew(ctx, n, "hiopamd_vec_set_to_constant", y, [2.5], np.full(n, 2.5))
    e = np.where(s == 1.0, 3.0, 0.0)
    _ew(ctx, n, "hiopamd_vec_set_to_constant_w_pattern", y, [3.0, sd], e)
    e = y.copy(); ho.copy_from_w_pattern(e, x, s)
    _ew(ctx, n, "hiopamd_vec_copy_from_w_pattern", y, [xd, sd], e)
    _ew(ctx, n, "hiopamd_vec_copy", y, [xd], x)
    _ew(ctx, n, "hiopamd_vec_component_mult", y, [xd], y * x)
    _ew(ctx, n, "hiopamd_vec_component_div", y, [xd], y / x)
    e = y.copy(); ho.component_div_w_pattern(e, x, s)
    _ew(ctx, n, "hiopamd_vec_component_div_w_pattern", y, [xd, sd], e)
    _ew(ctx, n, "hiopamd_vec_component_min_c", y, [0.25], np.minimum(y, 0.25))
    _ew(ctx, n, "hiopamd_vec_component_max_c", y, [0.25], np.maximum(y, 0.25))
    _ew(ctx, n, "hiopamd_vec_component_min_v", y, [D(x - 1)], np.minimum(y, x - 1))
    _ew(ctx, n, "hiopamd_vec_component_max_v", y, [D(x - 1)], np.maximum(y, x - 1))
    _ew(ctx, n, "hiopamd_vec_component_abs", y, [], np.abs(y))
    e = y.copy(); ho.component_sgn(e)
    _ew(ctx, n, "hiopamd_vec_component_sgn", y, [], e)
    _ew(ctx, n, "hiopamd_vec_component_sqrt", x, [], np.sqrt(x))
    _ew(ctx, n, "hiopamd_vec_scale", y, [0.75], y * 0.75)
    _ew(ctx, n, "hiopamd_vec_axpy", y, [0.3, xd], y + 0.3 * x)
    e = y.copy(); e[s == 1.0] += 0.3 * x[s == 1.0]
    _ew(ctx, n, "hiopamd_vec_axpy_w_pattern", y, [0.3, xd, sd], e)
    for alpha in (1.0, -1.0, 0.5, 0.0):
        e = y.copy(); ho.axzpy(e, alpha, x, z)
        _ew(ctx, n, "hiopamd_vec_axzpy", y, [alpha, xd, zd], e)
        e = y.copy(); ho.axdzpy(e, alpha, x, z)
        _ew(ctx, n, "hiopamd_vec_axdzpy", y, [alpha, xd, zd], e)
        e = y.copy(); ho.axdzpy_w_pattern(e, alpha, x, z, s)
        _ew(ctx, n, "hiopamd_vec_axdzpy_w_pattern", y, [alpha, xd, zd, sd], e)
    _ew(ctx, n, "hiopamd_vec_add_constant", y, [1.5], y + 1.5)
    e = y.copy(); e[s == 1.0] += 1.5
    _ew(ctx, n, "hiopamd_vec_add_constant_w_pattern", y, [1.5, sd], e)
    _ew(ctx, n, "hiopamd_vec_negate", y, [], -y)
    _ew(ctx, n, "hiopamd_vec_invert", x, [], 1.0 / x)
    e = y.copy(); ho.add_log_barrier_grad(e, 0.7, x, s)
    _ew(ctx, n, "hiopamd_vec_add_log_barrier_grad", y, [0.7, xd, sd], e)
    s2 = pattern(n, 5)
    e = y.copy(); ho.add_linear_damping_term(e, s, s2, 0.9, 1e-3)
    _ew(ctx, n, "hiopamd_vec_add_linear_damping_term", y, [sd, D(s2), 0.9, 1e-3], e)
    e = y.copy(); e[s == 0.0] = 0.0
    _ew(ctx, n, "hiopamd_vec_select_pattern", y, [sd], e)
    zz = mk(n, 6, 0.01, 3.0)
    e = zz.copy(); ho.adjust_duals_plh(e, x, s, 0.1, 1e10)
    _ew(ctx, n, "hiopamd_vec_adjust_duals_plh", zz, [xd, sd, 0.1, 1e10], e)
    e = zz.copy(); ho.adjust_duals_plh(e, x, s, 0.1, 1.5)
    _ew(ctx, n, "hiopamd_vec_adjust_duals_plh", zz, [xd, sd, 0.1, 1.5], e)
    _ew(ctx, n, "hiopamd_vec_set_to_linspace", y, [0.5, 0.25], 0.5 + 0.25 * np.arange(n), rtol=1e-15)


@pytest.mark.parametrize("n", SIZES)
def test_index_and_pattern_copies(ctx, n):
    x = mk(n, 11)
    s = pattern(n, 12)
    nsel = int(s.sum())
    # copy_from_indexes
    idx = rng(13).integers(0, max(n, 1), n).astype(np.int32)
    yd = D(np.zeros(n))
    run(ctx, "hiopamd_vec_copy_from_indexes", n, yd, D(x), D(idx, torch.int32))
    np.testing.assert_array_equal(yd.cpu().numpy(), x[idx] if n else x)
    # copyToStartingAt_w_pattern (order preserving compaction)
    dest = np.full(nsel + 5, -7.0)
    dd = D(dest)
    out = C.c_int64(-1)
    run(ctx, "hiopamd_vec_copy_to_starting_at_w_pattern", n, D(x), dd, 3, D(s), C.byref(out))
    e = dest.copy(); cnt = ho.copy_to_starting_at_w_pattern(x, e, 3, s)
    assert out.value == cnt
    np.testing.assert_array_equal(dd.cpu().numpy(), e)
    # startingAtCopyToStartingAt_w_pattern (scatter to selected)
    src = mk(nsel + 4, 14)
    for num in (-1, max(nsel // 2, 0)):
        dest = np.full(n, 9.0)
        dd = D(dest)
        run(ctx, "hiopamd_vec_starting_at_copy_to_starting_at_w_pattern", D(src), 2, dd, n, 0, D(s), num)
        e = dest.copy(); ho.starting_at_copy_to_starting_at_w_pattern(src, 2, e, 0, s, num)
        np.testing.assert_array_equal(dd.cpu().numpy(), e)
    # two-vector maps
    perm = rng(15).permutation(n).astype(np.int32)
    nc = n // 3
    cmap, dmap = perm[:nc].copy(), perm[nc:].copy()
    c, d = mk(nc, 16), mk(n - nc, 17)
    yd = D(np.zeros(n))
    run(ctx, "hiopamd_vec_copy_from_two_vec_w_pattern", yd, D(c), D(cmap, torch.int32), nc, D(d), D(dmap, torch.int32),
        n - nc)
    e = np.zeros(n); e[cmap] = c; e[dmap] = d
    np.testing.assert_array_equal(yd.cpu().numpy(), e)
    cd, ddv = D(np.zeros(nc)), D(np.zeros(n - nc))
    run(ctx, "hiopamd_vec_copy_to_two_vec_w_pattern", yd, cd, D(cmap, torch.int32), nc, ddv, D(dmap, torch.int32), n - nc)
    np.testing.assert_array_equal(cd.cpu().numpy(), c)
    np.testing.assert_array_equal(ddv.cpu().numpy(), d)
    # axpy with index map
    m = n // 2
    imap = perm[:m].copy()
    xs = mk(m, 18)
    yv = mk(n, 19)
    yd = D(yv)
    run(ctx, "hiopamd_vec_axpy_w_map", m, yd, 0.5, D(xs), D(imap, torch.int32))
    e = yv.copy(); e[imap] += 0.5 * xs
    np.testing.assert_allclose(yd.cpu().numpy(), e, rtol=4 * EPS)


@pytest.mark.parametrize("n", SIZES)
def test_reductions(ctx, n):
    x, y = mk(n, 21), mk(n, 22)
    xp = mk(n, 23, 0.01, 5.0)
    s = pattern(n, 24)
    s2 = pattern(n, 25)
    xd, yd, xpd, sd, s2d = D(x), D(y), D(xp), D(s), D(s2)
    torch.cuda.synchronize()
    R = 1e-13
    assert ctx.reduce_double("hiopamd_vec_dot", n, xd, yd) == pytest.approx(float(x @ y), rel=R, abs=1e-13)
    assert ctx.reduce_double("hiopamd_vec_twonorm", n, xd) == pytest.approx(float(np.linalg.norm(x)), rel=R)
    assert ctx.reduce_double("hiopamd_vec_infnorm", n, xd) == ho.infnorm(x)
    assert ctx.reduce_double("hiopamd_vec_onenorm", n, xd) == pytest.approx(ho.onenorm(x), rel=R)
    assert ctx.reduce_double("hiopamd_vec_sum", n, xd) == pytest.approx(float(x.sum()), rel=R, abs=1e-12)
    assert ctx.reduce_double("hiopamd_vec_min", n, xd) == (float(x.min()) if n else np.finfo(np.float64).max)
    assert ctx.reduce_double("hiopamd_vec_min_w_pattern", n, xd, sd) == ho.vmin_w_pattern(x, s)
    assert ctx.reduce_double("hiopamd_vec_log_barrier", n, xpd, sd) == pytest.approx(ho.log_barrier(xp, s), rel=1e-13, abs=1e-10)
    assert ctx.reduce_double("hiopamd_vec_linear_damping_term", n, xd, sd, s2d, 0.1, 1e-5) == pytest.approx(
        ho.linear_damping_term(x, s, s2, 0.1, 1e-5), rel=R, abs=1e-18)
    assert ctx.reduce_double("hiopamd_vec_fraction_to_the_bdry", n, xpd, yd, 0.99) == pytest.approx(
        ho.fraction_to_the_bdry(xp, y, 0.99), rel=4 * EPS)
    assert ctx.reduce_double("hiopamd_vec_fraction_to_the_bdry_w_pattern", n, xpd, yd, 0.99, sd) == pytest.approx(
        ho.fraction_to_the_bdry_w_pattern(xp, y, 0.99, s), rel=4 * EPS)
    assert ctx.reduce_int("hiopamd_vec_all_positive", n, xpd) == 1
    assert ctx.reduce_int("hiopamd_vec_all_positive", n, xd) == int(not np.any(x <= 0))
    assert ctx.reduce_int("hiopamd_vec_all_positive_w_pattern", n, xd, sd) == ho.all_positive_w_pattern(x, s)
    assert ctx.reduce_int("hiopamd_vec_matches_pattern", n, xd, sd) == ho.matches_pattern(x, s)
    xm = x * s
    assert ctx.reduce_int("hiopamd_vec_matches_pattern", n, D(xm), sd) == 1
    assert ctx.reduce_int("hiopamd_vec_is_zero", n, xd) == int(not np.any(x != 0))
    assert ctx.reduce_int("hiopamd_vec_is_zero", n, D(np.zeros(n))) == 1
    assert ctx.reduce_int("hiopamd_vec_isnan", n, xd) == 0
    assert ctx.reduce_int("hiopamd_vec_isinf", n, xd) == 0
    assert ctx.reduce_int("hiopamd_vec_isfinite", n, xd) == 1
    if n > 2:
        bad = x.copy(); bad[n // 2] = np.nan; bad[0] = np.inf
        bd = D(bad)
        assert ctx.reduce_int("hiopamd_vec_isnan", n, bd) == 1
        assert ctx.reduce_int("hiopamd_vec_isinf", n, bd) == 1
        assert ctx.reduce_int("hiopamd_vec_isfinite", n, bd) == 0
    assert ctx.reduce_int64("hiopamd_vec_num_elems_less_than", n, xd, 0.1) == int((x < 0.1).sum())
    assert ctx.reduce_int64("hiopamd_vec_num_elems_abs_less_than", n, xd, 0.1) == int((np.abs(x) < 0.1).sum())
    assert ctx.reduce_int("hiopamd_vec_is_equal", n, xd, D(x.copy())) == 1
    if n:
        assert ctx.reduce_int("hiopamd_vec_is_equal", n, xd, yd) == 0


def test_reduction_is_run_to_run_deterministic(ctx):
    n = 1_000_003
    x, y = D(mk(n, 31)), D(mk(n, 32))
    torch.cuda.synchronize()
    vals = {ctx.reduce_double("hiopamd_vec_dot", n, x, y) for _ in range(5)}
    assert len(vals) == 1


def test_batched_reductions_one_round_trip_same_bits(ctx):
    """hiopamd_ctx_reduce_begin / _end: the scalar-returning reductions launched inside the bracket deliver their host results when the
    bracket closes (one stream synchronisation), bitwise those of the unbatched calls; more than 64 pending results flush in between;
    a reduction the library needs for its own control flow inside a caller's bracket (here: the Krylov solver's norms) is not deferred."""
    from hiop_amd._lib import lib
    L = lib()
    n = 300_007
    x, y = mk(n, 41), mk(n, 42)
    xp, s, s2 = mk(n, 43, 0.01, 5.0), pattern(n, 44), pattern(n, 45)
    xd, yd, xpd, sd, s2d = D(x), D(y), D(xp), D(s), D(s2)
    torch.cuda.synchronize()
    calls = [("hiopamd_vec_dot", (n, xd, yd)), ("hiopamd_vec_twonorm", (n, xd)), ("hiopamd_vec_infnorm", (n, xd)),
             ("hiopamd_vec_onenorm", (n, xd)), ("hiopamd_vec_sum", (n, xd)), ("hiopamd_vec_min", (n, xd)),
             ("hiopamd_vec_min_w_pattern", (n, xd, sd)), ("hiopamd_vec_log_barrier", (n, xpd, sd)),
             ("hiopamd_vec_linear_damping_term", (n, xd, sd, s2d, 0.1, 1e-5)), ("hiopamd_vec_fraction_to_the_bdry", (n, xpd, yd, 0.99)),
             ("hiopamd_vec_fraction_to_the_bdry_w_pattern", (n, xpd, yd, 0.99, sd))]
    single = [ctx.reduce_double(name, *a) for name, a in calls]
    outs = [C.c_double(-7.0) for _ in calls]
    assert L.hiopamd_ctx_reduce_begin(ctx.h) == 0
    for (name, a), o in zip(calls, outs):
        ctx.call(name, *a, C.byref(o))
    assert all(o.value == -7.0 for o in outs)          # nothing has been written yet: no synchronisation happened
    assert L.hiopamd_ctx_reduce_end(ctx.h) == 0
    assert [o.value for o in outs] == single
    # more results than pinned slots: flushed on the way, all correct at the end; nested brackets close on the outermost end
    many = [C.c_double(0.0) for _ in range(150)]
    assert L.hiopamd_ctx_reduce_begin(ctx.h) == 0 and L.hiopamd_ctx_reduce_begin(ctx.h) == 0
    for q, o in enumerate(many):
        ctx.call("hiopamd_vec_dot", n - q, xd, yd, C.byref(o))
    assert L.hiopamd_ctx_reduce_end(ctx.h) == 0
    assert many[-1].value == 0.0                       # (the inner end does not flush)
    assert L.hiopamd_ctx_reduce_end(ctx.h) == 0
    for q in (0, 1, 63, 64, 65, 149):
        assert many[q].value == ctx.reduce_double("hiopamd_vec_dot", n - q, xd, yd)
    assert L.hiopamd_ctx_reduce_end(ctx.h) != 0        # unbalanced end is an error, not a crash
    # the library's own reductions inside a caller's bracket: the log-barrier objective of an iterate (four sums, one round trip of its own)
    assert L.hiopamd_ctx_reduce_begin(ctx.h) == 0
    v0 = C.c_double(0.0)
    ctx.call("hiopamd_vec_dot", n, xd, yd, C.byref(v0))
    t = ctx.reduce_int("hiopamd_vec_all_positive", n, xpd)    # an immediate reduction in between keeps the pending one pending
    assert t == 1 and v0.value == 0.0
    assert L.hiopamd_ctx_reduce_end(ctx.h) == 0
    assert v0.value == single[0]


def test_fraction_to_the_bdry_multi(ctx):
    ns = [1000, 37, 5000, 3]
    xs = [mk(n, 40 + i, 0.01, 2.0) for i, n in enumerate(ns)]
    ds = [mk(n, 50 + i) for i, n in enumerate(ns)]
    ss = [pattern(n, 60 + i) for i, n in enumerate(ns)]
    xd, dd, sd = [D(a) for a in xs], [D(a) for a in ds], [D(a) for a in ss]
    torch.cuda.synchronize()
    k = len(ns)
    narr = (C.c_int64 * k)(*ns)
    P = C.c_void_p * k
    out = C.c_double(0)
    ctx.call("hiopamd_vec_fraction_to_the_bdry_multi", k, C.cast(narr, C.c_void_p), C.cast(P(*[t.data_ptr() for t in xd]), C.c_void_p),
             C.cast(P(*[t.data_ptr() for t in dd]), C.c_void_p), C.cast(P(*[t.data_ptr() for t in sd]), C.c_void_p), 0.995,
             C.byref(out))
    e = min(ho.fraction_to_the_bdry_w_pattern(x, d, 0.995, s) for x, d, s in zip(xs, ds, ss))
    assert out.value == pytest.approx(e, rel=4 * EPS)


@pytest.mark.parametrize("n", [0, 5, 1000, 65537])
def test_project_into_bounds(ctx, n):
    x0 = mk(n, 71, -3, 3)
    xl = mk(n, 72, -2, 0)
    xu = xl + mk(n, 73, 1e-3, 3)
    ixl, ixu = pattern(n, 74), pattern(n, 75)
    xd = D(x0)
    ok = C.c_int(-1)
    run(ctx, "hiopamd_vec_project_into_bounds", n, xd, D(xl), D(ixl), D(xu), D(ixu), 1e-2, 1e-2, C.byref(ok))
    e = x0.copy()
    assert ho.project_into_bounds(e, xl, ixl, xu, ixu, 1e-2, 1e-2)
    assert ok.value == 1
    np.testing.assert_allclose(xd.cpu().numpy(), e, rtol=4 * EPS)
    if n > 3:
        xl2 = xl.copy(); ixl2 = ixl.copy(); ixu2 = ixu.copy()
        xl2[2] = xu[2] + 1; ixl2[2] = ixu2[2] = 1.0
        run(ctx, "hiopamd_vec_project_into_bounds", n, D(x0), D(xl2), D(ixl2), D(xu), D(ixu2), 1e-2, 1e-2, C.byref(ok))
        assert ok.value == 0
