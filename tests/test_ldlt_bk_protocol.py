"""CPU replay of the device protocol of the pivoted factorisation (hiop_amd/csrc/ldlt_bk.hip), thread by thread.

The HIP file claims that none of its three launches per column has an internal ordering requirement: "everything a thread writes
depends only on its own row plus a handful of scalars the deciding workgroup saved".  This test restates the three kernels at the
granularity of one thread = one Python call that reads and writes the SHARED arrays in place, and runs the threads of a launch in
RANDOM order.  If a thread read a location another thread of the same launch writes, the result would depend on the order; it must
instead equal the oracle's factor (oracle/bunch_kaufman.py: pivots, permutation, L, D) for every order tried.  The saved scalars, the
"last workgroup decides" points and the two index spaces of the apply launch (rows i >= k, previous columns j < k) are those of the
HIP code, in the same places."""
import numpy as np
import pytest

from oracle import bunch_kaufman as bk
from tests.test_oracle_bunch_kaufman import make

ALPHA = bk.ALPHA
NB = 64


class State:
    def __init__(self):
        self.next_k = 0
        self.info = 0


def column_launch(a, W, st, n, k, k0, second, order):
    """bk_column_kernel<SECOND>: one call of `thread(i)` per row i >= k, in the given order; then the deciding block."""
    if st.next_k != k or (second and not st.need2):
        return
    kw = k - k0
    col = kw + 1 if second else kw
    src = st.imax if second else k
    coef = W[src, :kw].copy()                      # (loaded into LDS before any thread writes column `col`: columns < kw only)
    best, bidx = -1.0, 1 << 30

    def thread(i):
        nonlocal best, bidx
        v = a[i, src] if (not second or i >= src) else a[src, i]
        for p in range(kw):
            v -= a[i, k0 + p] * coef[p]
        W[i, col] = v
        cand = (i != src) if second else (i > k)
        if cand:
            av = abs(v)
            if av > best or (av == best and i < bidx):
                best, bidx = av, i

    for i in order:
        thread(i)
    # ---- the last workgroup decides (reads what the others wrote: behind the fence + counter of the HIP code)
    if not second:
        wkk = W[k, kw]
        st.absakk, st.colmax = abs(wkk), (best if best >= 0 else 0.0)
        st.imax = bidx if best >= 0 else k
        st.c0_k = wkk
        st.need2 = False
        if not (max(st.absakk, st.colmax) > 0.0):
            if st.info == 0:
                st.info = k + 1
        elif not (st.absakk >= ALPHA * st.colmax):
            st.need2 = True
        if not st.need2:
            st.kp, st.kstep, st.use_c1 = k, 1, False
            st.c0_kk = st.c0_kp = wkk
            st.c1_kk = st.c1_kp = 0.0
            st.akk_old = a[k, k]
    else:
        imax, rowmax = src, (best if best >= 0 else 0.0)
        wii = abs(W[imax, kw + 1])
        st.use_c1 = False
        if st.absakk >= ALPHA * st.colmax * (st.colmax / rowmax):
            st.kp, st.kstep = k, 1
        elif wii >= ALPHA * rowmax:
            st.kp, st.kstep, st.use_c1 = imax, 1, True
        else:
            st.kp, st.kstep = imax, 2
        kk = k + st.kstep - 1
        st.c0_kk, st.c0_kp = W[kk, kw], W[st.kp, kw]
        st.c1_kk, st.c1_kp = W[kk, kw + 1], W[st.kp, kw + 1]
        st.akk_old = a[kk, kk]
        st.need2 = False


def apply_launch(a, W, e, ipiv, perm, st, n, k, k0, order):
    """bk_apply_kernel: thread t owns row i = k + t of the trailing part AND previous column j = t (and panel column t of W)."""
    if st.next_k != k:
        return
    kp, kstep, use_c1 = st.kp, st.kstep, st.use_c1
    kk, kw = k + kstep - 1, k - k0
    swp = kp != kk

    def thread(t):
        if swp:
            if t < k:
                a[kk, t], a[kp, t] = a[kp, t], a[kk, t]
            if t < kw:
                W[kk, t], W[kp, t] = W[kp, t], W[kk, t]
        i = k + t
        if i >= n:
            return
        is_kk, is_kp = swp and i == kk, swp and i == kp
        w1 = 0.0
        if is_kk:
            w0, w1 = (st.c1_kp if use_c1 else st.c0_kp), st.c1_kp
        elif is_kp:
            w0, w1 = (st.c1_kk if use_c1 else st.c0_kk), st.c1_kk
        else:
            w0 = W[i, kw + 1] if use_c1 else W[i, kw]
            if kstep == 2:
                w1 = W[i, kw + 1]
        if use_c1 or is_kk or is_kp:
            W[i, kw] = w0
        if kstep == 2 and (is_kk or is_kp):
            W[i, kw + 1] = w1
        if swp:
            if i == kp:
                a[kp, kp] = st.akk_old
            elif kk < i < kp:
                a[kp, i] = a[i, kk]
            elif i > kp:
                a[i, kp] = a[i, kk]
        if kstep == 1:
            dk = ((st.c1_kp if use_c1 else st.c0_kp) if swp else st.c0_k)
            if i == k:
                a[k, k] = dk
            else:
                a[i, k] = w0 * (1.0 / dk) if dk != 0.0 else w0
        else:
            wk0 = st.c0_k
            wk10 = st.c0_kp if swp else st.c0_kk
            wk11 = st.c1_kp if swp else st.c1_kk
            if i == k:
                a[k, k] = wk0
            elif i == k + 1:
                a[k + 1, k] = 0.0
                e[k] = wk10
                a[k + 1, k + 1] = wk11
            else:
                d21 = wk10
                d11, d22 = wk11 / d21, wk0 / d21
                tt = 1.0 / (d11 * d22 - 1.0)
                d21 = tt / d21
                a[i, k] = d21 * (d11 * w0 - w1)
                a[i, k + 1] = d21 * (d22 * w1 - w0)

    for t in order:
        thread(t)
    # ---- the last workgroup
    if kstep == 1:
        ipiv[k] = kp + 1
    else:
        ipiv[k] = ipiv[k + 1] = -(kp + 1)
    if swp:
        perm[kk], perm[kp] = perm[kp], perm[kk]
    st.next_k = k + kstep


def replay(A, rng):
    n = A.shape[0]
    a = np.tril(np.array(A, dtype=np.float64))      # a[i, j], i >= j: what the device addresses as A[j * lda + i]
    a[np.triu_indices(n, 1)] = np.nan                # the other triangle is never read
    e, ipiv, perm = np.zeros(n + 1), np.zeros(n, dtype=np.int64), np.arange(n)
    st = State()
    k0 = 0
    while k0 < n:
        last = (n - k0) <= NB
        kcap = n if last else k0 + NB - 1
        W = np.full((n, NB), np.nan)
        for k in range(k0, kcap):
            rows = np.arange(k, n)
            column_launch(a, W, st, n, k, k0, False, rng.permutation(rows))
            column_launch(a, W, st, n, k, k0, True, rng.permutation(rows))
            apply_launch(a, W, e, ipiv, perm, st, n, k, k0, rng.permutation(max(n - k, k)))
        kend = st.next_k
        if last:
            assert kend == n
            break
        kb = kend - k0
        assert kcap <= kend <= k0 + NB
        upd = a[kend:, k0:kend] @ W[kend:, :kb].T       # ldlt_rankk_update: a(c, r) -= sum_p a(c, k0 + p) W(r, p), c >= r >= kend
        a[kend:, kend:] -= np.tril(upd)
        k0 = kend
    return a, e[:n], ipiv, perm, st.info


@pytest.mark.parametrize("kind,n", [("rand", 1), ("rand", 2), ("rand", 9), ("rand", 64), ("rand", 65), ("rand", 150), ("zero_diag", 70),
                                    ("kkt", 96), ("arrow", 90), ("graded", 100)])
def test_threads_in_any_order_give_the_oracle_factor(kind, n):
    A = make(kind, n)
    fo = bk.factor(A)
    for seed in range(3):
        a, e, ipiv, perm, info = replay(A, np.random.default_rng(seed))
        assert info == 0
        np.testing.assert_array_equal(ipiv, fo.ipiv)
        np.testing.assert_array_equal(perm, fo.perm)
        L = np.tril(a, -1) + np.eye(n)
        g = max(1.0, np.abs(fo.L).max())
        np.testing.assert_allclose(np.diag(a), fo.d, rtol=1e-10, atol=1e-12 * np.abs(A).max() * g * g)
        np.testing.assert_allclose(e, fo.e, rtol=1e-10, atol=1e-12 * np.abs(A).max() * g * g)
        np.testing.assert_allclose(L, fo.L, rtol=1e-9, atol=1e-11 * g * g)


def test_exactly_zero_column_sets_info_and_goes_on():
    A = np.zeros((6, 6))
    A[0, 0] = 2.0
    a, e, ipiv, perm, info = replay(A, np.random.default_rng(0))
    assert info == 2 == bk.factor(A).info


# ---- the panel kernel (round 5): G workgroups, ONE grid barrier per column, tagged granules for what crosses workgroups -------------------
# bk_panel_kernel runs the three phases of every column of a panel inside ONE launch.  What the launches' boundaries used to order is now
# ordered by (a) a grid barrier at the END of every column step and (b) tagged granules: every scalar that crosses workgroups — the partial
# maxima of phases A and B, the values W(k / k+1 / imax, kw / kw+1) and the two diagonal entries, published by the threads that own those
# rows — is written as 8-byte words {32-bit payload, 32-bit tag}, tag = 4 k + 1 (phase A of column k) or 4 k + 2 (phase B); the deciding
# wave of a workgroup waits until every granule it needs carries the expected tag.  Nothing else orders phases A, B and C of a column.
# The model runs G workgroups as coroutines that yield wherever the device code lets another workgroup overtake (between any two granule
# stores, at a wait, at the barrier) under a random scheduler; every workgroup keeps its OWN copy of the decision (the d_* words in
# LDS).  Two sabotaged variants must be caught, otherwise the model proves nothing: `publish=False` (the deciding wave reads W and a
# directly, as the first form of the kernel did behind a barrier that no longer exists) and `check_all_tags=False` (the reader looks at
# the tag of ONE granule of a multi-granule value).
GMAX = 16          # BK_G of the HIP file: the granule layout is sized for it whatever the number of workgroups of a launch


def _halves(v):
    b = int(np.float64(v).view(np.uint64))
    return b & 0xffffffff, b >> 32


def _join(lo, hi):
    return float(np.uint64((int(hi) << 32) | int(lo)).view(np.float64))


def panel_kernel(a, W, e, ipiv, perm, st, n, k0, kcap, G, T, rng, publish=True, check_all_tags=True):
    gp, gt = np.zeros(6 * GMAX + 16, dtype=np.uint64), np.zeros(6 * GMAX + 16, dtype=np.int64)     # granules: payload, tag (zeroed before the launch)

    def rows_of(g, lo):
        return [i for i in range(lo, n) if (i // T) % G == g]

    def gput_f64(idx, v, tag):
        lo, hi = _halves(v)
        gp[idx], gt[idx] = lo, tag
        yield "run"
        gp[idx + 1], gt[idx + 1] = hi, tag

    def gget_f64(idx):
        return _join(gp[idx], gp[idx + 1])

    PUB = 6 * GMAX

    def workgroup(g):
        k = st.next_k
        d = State()
        while k < kcap:
            kw = k - k0
            tagA, tagB = 4 * k + 1, 4 * k + 2

            def column_phase(second, src):
                col = kw + 1 if second else kw
                tag = tagB if second else tagA
                coef = W[src, :kw].copy()
                best, bidx = -1.0, 1 << 30
                for i in rng.permutation(rows_of(g, k)):
                    v = a[i, src] if (not second or i >= src) else a[src, i]
                    c0own = W[i, kw] if second else None
                    for p in range(kw):
                        v -= a[i, k0 + p] * coef[p]
                    W[i, col] = v
                    if not second:
                        if i == k:
                            yield from gput_f64(PUB + 0, v, tag)
                            yield from gput_f64(PUB + 12, a[i, i], tag)
                        if i == k + 1:
                            yield from gput_f64(PUB + 2, v, tag)
                            yield from gput_f64(PUB + 14, a[i, i], tag)
                    else:
                        if i == k:
                            yield from gput_f64(PUB + 6, v, tag)
                        if i == k + 1:
                            yield from gput_f64(PUB + 8, v, tag)
                        if i == src:
                            yield from gput_f64(PUB + 4, c0own, tag)
                            yield from gput_f64(PUB + 10, v, tag)
                    if (i != src) if second else (i > k):
                        av = abs(v)
                        if av > best or (av == best and i < bidx):
                            best, bidx = av, i
                base = ((GMAX if second else 0) + g) * 3
                yield from gput_f64(base, best, tag)
                yield "run"
                gp[base + 2], gt[base + 2] = bidx, tag

            def needed(second):
                idx = [((GMAX if second else 0) * 3 + q, tagB if second else tagA) for q in range(3 * G)]
                for q in ((0, 1, 2, 3, 4, 5, 6, 7) if second else (0, 6)):
                    t = tagA if q in (0, 1, 6, 7) else tagB
                    idx += [(PUB + 2 * q, t), (PUB + 2 * q + 1, t)]
                if not check_all_tags:      # sabotage: one granule per value
                    idx = [x for x in idx if (x[0] < PUB and x[0] % 3 == 0) or (x[0] >= PUB and x[0] % 2 == 0)]
                return idx

            def ready(second):
                want = needed(second)
                return lambda: all(gt[i] == t for i, t in want)

            def fold(second):
                best, bidx = -1.0, 1 << 30
                for b in range(G):
                    base = ((GMAX if second else 0) + b) * 3
                    v, ix = gget_f64(base), int(gp[base + 2])
                    if v > best or (v == best and ix < bidx):
                        best, bidx = v, ix
                return best, bidx

            yield from column_phase(False, k)
            yield ("wait", ready(False))
            best, bidx = fold(False)
            wkk = gget_f64(PUB) if publish else W[k, kw]
            d.absakk, d.colmax = abs(wkk), (best if best >= 0 else 0.0)
            d.imax = bidx if best >= 0 else k
            d.c0_k, d.need2 = wkk, False
            if not (max(d.absakk, d.colmax) > 0.0):
                if g == 0 and st.info == 0:
                    st.info = k + 1
            elif not (d.absakk >= ALPHA * d.colmax):
                d.need2 = True
            if not d.need2:
                d.kp, d.kstep, d.use_c1 = k, 1, False
                d.c0_kk = d.c0_kp = wkk
                d.c1_kk = d.c1_kp = 0.0
                d.akk_old = gget_f64(PUB + 12) if publish else a[k, k]
            yield "run"
            if d.need2:
                imax = d.imax
                yield from column_phase(True, imax)
                yield ("wait", ready(True))
                best, bidx = fold(True)
                rowmax = best if best >= 0 else 0.0
                if publish:
                    c0_k, c0_k1, c0_im, c1_k, c1_k1, c1_im, a_k, a_k1 = (gget_f64(PUB + 2 * q) for q in range(8))
                else:
                    k1 = min(k + 1, n - 1)
                    c0_k, c0_k1, c0_im, c1_k, c1_k1, c1_im, a_k, a_k1 = (W[k, kw], W[k1, kw], W[imax, kw], W[k, kw + 1], W[k1, kw + 1],
                                                                         W[imax, kw + 1], a[k, k], a[k1, k1])
                wii = abs(c1_im)
                d.use_c1 = False
                if d.absakk >= ALPHA * d.colmax * (d.colmax / rowmax):
                    d.kp, d.kstep = k, 1
                elif wii >= ALPHA * rowmax:
                    d.kp, d.kstep, d.use_c1 = imax, 1, True
                else:
                    d.kp, d.kstep = imax, 2
                two = d.kstep == 2
                d.c0_kk, d.c1_kk = (c0_k1, c1_k1) if two else (c0_k, c1_k)
                d.c0_kp, d.c1_kp = (c0_k, c1_k) if d.kp == k else (c0_im, c1_im)
                d.akk_old = a_k1 if two else a_k
                yield "run"
            # ---- phase C: thread j owns previous column j, panel column j of W and row i = j
            kp, kstep, use_c1 = d.kp, d.kstep, d.use_c1
            kk = k + kstep - 1
            swp = kp != kk
            for j in rng.permutation(rows_of(g, 0)):
                if swp and k0 <= j < k:          # (the columns in front of the panel: once per panel, defer_swaps below)
                    a[kk, j], a[kp, j] = a[kp, j], a[kk, j]
                if swp and j < kw:
                    W[kk, j], W[kp, j] = W[kp, j], W[kk, j]
                i = j
                if i < k:
                    continue
                is_kk, is_kp = swp and i == kk, swp and i == kp
                wc0, wc1 = W[i, kw], (W[i, kw + 1] if (use_c1 or kstep == 2) else 0.0)
                akki = a[i, kk] if (swp and i > kk and i != kp) else 0.0
                w1 = 0.0
                if is_kk:
                    w0, w1 = (d.c1_kp if use_c1 else d.c0_kp), d.c1_kp
                elif is_kp:
                    w0, w1 = (d.c1_kk if use_c1 else d.c0_kk), d.c1_kk
                else:
                    w0 = wc1 if use_c1 else wc0
                    if kstep == 2:
                        w1 = wc1
                if use_c1 or is_kk or is_kp:
                    W[i, kw] = w0
                if kstep == 2 and (is_kk or is_kp):
                    W[i, kw + 1] = w1
                if swp:
                    if i == kp:
                        a[kp, kp] = d.akk_old
                    elif kk < i < kp:
                        a[kp, i] = akki
                    elif i > kp:
                        a[i, kp] = akki
                if kstep == 1:
                    dk = ((d.c1_kp if use_c1 else d.c0_kp) if swp else d.c0_k)
                    if i == k:
                        a[k, k] = dk
                    else:
                        a[i, k] = w0 * (1.0 / dk) if dk != 0.0 else w0
                else:
                    wk0, wk10, wk11 = d.c0_k, (d.c0_kp if swp else d.c0_kk), (d.c1_kp if swp else d.c1_kk)
                    if i == k:
                        a[k, k] = wk0
                    elif i == k + 1:
                        a[k + 1, k] = 0.0
                        e[k] = wk10
                        a[k + 1, k + 1] = wk11
                    else:
                        d21 = wk10
                        d11, d22 = wk11 / d21, wk0 / d21
                        tt = 1.0 / (d11 * d22 - 1.0)
                        d21 = tt / d21
                        a[i, k] = d21 * (d11 * w0 - w1)
                        a[i, k + 1] = d21 * (d22 * w1 - w0)
            if g == 0:
                if kstep == 1:
                    ipiv[k] = kp + 1
                else:
                    ipiv[k] = ipiv[k + 1] = -(kp + 1)
                if swp:
                    perm[kk], perm[kp] = perm[kp], perm[kk]
            yield "barrier"
            k += kstep
        if g == 0:
            st.next_k = k

    progs = [workgroup(g) for g in range(G)]
    at_barrier, done, waits = set(), set(), {}
    while len(done) < G:
        runnable = [g for g in range(G) if g not in at_barrier and g not in done and (g not in waits or waits[g]())]
        if not runnable:
            blocked = [g for g in range(G) if g in waits and g not in at_barrier and g not in done]
            assert not blocked, "deadlock: a workgroup waits for granules nobody will write"
            assert len(at_barrier) + len(done) == G and at_barrier     # everybody is at the barrier: it opens
            at_barrier.clear()
            continue
        g = int(rng.choice(runnable))
        waits.pop(g, None)
        try:
            what = next(progs[g])
            if what == "barrier":
                at_barrier.add(g)
            elif isinstance(what, tuple):
                waits[g] = what[1]
        except StopIteration:
            done.add(g)
            assert not at_barrier, "a workgroup left the kernel while others wait at a barrier"


def defer_swaps(a, ipiv, n, k0, kend, rng):
    """bk_defer_swaps_kernel: one thread per column j < k0; the panel's interchanges, read back from ipiv in pivot order, applied to the
    column's entries gathered in a private buffer (64 panel rows + one slot per distinct row further down)"""
    swaps, k = [], k0
    while k < kend:
        pv = int(ipiv[k])
        kstep = 1 if pv > 0 else 2
        kp, kk = abs(pv) - 1, k + kstep - 1
        if kp != kk:
            swaps.append((kk, kp))
        k += kstep
    if not swaps:
        return
    slot = []
    for t, (kk, kp) in enumerate(swaps):
        sl = kp - k0
        if sl >= NB:
            first = next(u for u in range(t + 1) if swaps[u][1] == kp)
            sl = NB + first
        slot.append(sl)
    nrow = min(NB, n - k0)
    for j in rng.permutation(k0):
        buf = np.full(2 * NB, np.nan)
        buf[:nrow] = a[k0:k0 + nrow, j]
        for t, (kk, kp) in enumerate(swaps):
            if slot[t] == NB + t:
                buf[NB + t] = a[kp, j]
        for t, (kk, kp) in enumerate(swaps):
            x, y = kk - k0, slot[t]
            buf[x], buf[y] = buf[y], buf[x]
        a[k0:k0 + nrow, j] = buf[:nrow]
        for t, (kk, kp) in enumerate(swaps):
            if slot[t] == NB + t:
                a[kp, j] = buf[NB + t]


def replay_panels(A, rng, T, Gmax, publish=True, check_all_tags=True):
    n = A.shape[0]
    a = np.tril(np.array(A, dtype=np.float64))
    a[np.triu_indices(n, 1)] = np.nan
    e, ipiv, perm = np.zeros(n + 1), np.zeros(n, dtype=np.int64), np.arange(n)
    st = State()
    k0 = 0
    while k0 < n:
        last = (n - k0) <= NB
        kcap = n if last else k0 + NB - 1
        W = np.full((n, NB), np.nan)
        panel_kernel(a, W, e, ipiv, perm, st, n, k0, kcap, min(Gmax, (n + T - 1) // T), T, rng, publish, check_all_tags)
        kend = st.next_k
        if k0 > 0 and kend > k0:
            defer_swaps(a, ipiv, n, k0, kend, rng)
        if last:
            assert kend == n
            break
        kb = kend - k0
        assert kcap <= kend <= k0 + NB
        upd = a[kend:, k0:kend] @ W[kend:, :kb].T
        a[kend:, kend:] -= np.tril(upd)
        k0 = kend
    return a, e[:n], ipiv, perm, st.info


def matches_oracle(A, out):
    n = A.shape[0]
    fo = bk.factor(A)
    a, e, ipiv, perm, info = out
    if info != fo.info or not np.array_equal(ipiv, fo.ipiv) or not np.array_equal(perm, fo.perm):
        return False
    L = np.tril(a, -1) + np.eye(n)
    g = max(1.0, np.abs(fo.L).max())
    with np.errstate(invalid="ignore"):
        return bool(np.allclose(np.diag(a), fo.d, rtol=1e-10, atol=1e-12 * np.abs(A).max() * g * g) and
                    np.allclose(e, fo.e, rtol=1e-10, atol=1e-12 * np.abs(A).max() * g * g) and np.allclose(L, fo.L, rtol=1e-9, atol=1e-11 * g * g))


@pytest.mark.parametrize("kind,n,T,G", [("rand", 1, 4, 8), ("rand", 2, 1, 8), ("rand", 9, 2, 3), ("rand", 65, 4, 16), ("rand", 150, 8, 8), ("rand", 130, 2, 16),
                                        ("zero_diag", 70, 4, 5), ("kkt", 96, 4, 8), ("arrow", 90, 16, 4), ("graded", 100, 8, 2)])
def test_panel_kernel_workgroups_at_their_own_pace_give_the_oracle_factor(kind, n, T, G):
    A = make(kind, n)
    for seed in range(3):
        assert matches_oracle(A, replay_panels(A, np.random.default_rng(seed), T, G))


def _caught(**kw):
    A = make("rand", 65)
    bad = 0
    for seed in range(12):
        try:
            ok = matches_oracle(A, replay_panels(A, np.random.default_rng(seed), 4, 8, **kw))
        except (AssertionError, ZeroDivisionError, FloatingPointError, IndexError, OverflowError):
            ok = False
        bad += 0 if ok else 1
    return bad


def test_panel_kernel_without_the_published_scalars_is_caught():
    """the deciding waves read W / a directly (nothing orders that against another workgroup's phase C any more): some interleaving lets
    a workgroup decide on values another one has already interchanged — the model must notice, otherwise it proves nothing"""
    assert _caught(publish=False) > 0


def test_panel_kernel_reader_that_checks_one_tag_per_value_is_caught():
    """a value of two or three granules is complete only when ALL of them carry the tag: a reader that looks at the first one combines
    the new low word with an old high word (or an old row index) under some interleaving"""
    assert _caught(check_all_tags=False) > 0
