"""The sparse LDL^T of row f2 (csrc/sparse_ldl.hip) on the CPU: its symbolic phase is host code behind two host-only entry points
(hiopamd_sparse_ldl_analyse / _plan); this test REPLAYS the numeric phase the device kernels execute — gather every front from the
plan's sources in list order, LDL^T of its pivot columns, L panel + update matrix to the pools, root gather, forward / backward sweeps
by levels — in numpy, exactly as the plan prescribes, and compares with dense linear algebra:
  * the ordering is a permutation, every front has <= 128 rows, the root is the only dense part;
  * M x = b solved through the replayed factors agrees with numpy.linalg.solve;
  * (#negative pivots, #zero pivots) equals the inertia from the eigenvalues (Sylvester) on positive definite AND indefinite
    quasi-definite matrices — the right-hand-side-independent verdict the reference gets from its sparse Cholesky
    (hiopKKTLinSysSparseCondensed.cpp:386-388, :469-496).
The GPU tests (tests/test_gpu_sparse_ldl.py) run the same matrices through the device kernels."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp


def _lib():
    from hiop_amd._lib import lib
    return lib()


def csr_full(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def get_plan(n, rp, ci):
    L = _lib()
    ip = lambda a: a.ctypes.data_as(C.c_void_p)
    sizes = np.zeros(16, dtype=np.int64)
    nul = C.c_void_p(0)
    args0 = [nul] * 24
    rc = L.hiopamd_sparse_ldl_plan(n, ip(rp), ip(ci), ip(sizes), *args0)
    if rc != 0:
        return rc, None
    nf, nlev, nidx, mr, ms, vr, vs, rmr, rms, rvr, rvs, r, lsz, usz, vsz, _ = [int(v) for v in sizes]
    i32 = lambda k: np.zeros(max(k, 1), dtype=np.int32)
    i64 = lambda k: np.zeros(max(k, 1), dtype=np.int64)
    P = dict(level_ptr=i32(nlev + 1), f_nc=i32(nf), f_nr=i32(nf), f_lofs=i64(nf), f_uofs=i64(nf), f_vofs=i64(nf), f_iofs=i64(nf),
             fidx=i32(nidx), mat_dest=i32(mr), mat_ptr=i64(mr + 1), mat_src=i64(ms), mat_front=i64(nf + 1), vec_dest=i32(vr),
             vec_ptr=i64(vr + 1), vec_src=i64(vs), vec_front=i64(nf + 1), rmat_dest=i32(rmr), rmat_ptr=i64(rmr + 1), rmat_src=i64(rms),
             rvec_dest=i32(rvr), rvec_ptr=i64(rvr + 1), rvec_src=i64(rvs), root_old=i32(r))
    order = ["level_ptr", "f_nc", "f_nr", "f_lofs", "f_uofs", "f_vofs", "f_iofs", "fidx", "mat_dest", "mat_ptr", "mat_src", "mat_front",
             "vec_dest", "vec_ptr", "vec_src", "vec_front", "rmat_dest", "rmat_ptr", "rmat_src", "rvec_dest", "rvec_ptr", "rvec_src",
             "root_old"]
    rc = L.hiopamd_sparse_ldl_plan(n, ip(rp), ip(ci), ip(sizes), *[ip(P[k]) for k in order])
    assert rc == 0
    P.update(nf=nf, nlev=nlev, r=r, lsz=lsz, usz=usz, vsz=vsz, n_rmat=rmr, n_rvec=rvr)
    return 0, P


def gather(src, b, e, vals, pool):
    s = 0.0
    for q in range(b, e):
        i = int(src[q])
        s += vals[i] if i >= 0 else pool[-(i + 1)]
    return s


def replay_factor(P, vals):
    """what sl_factor_level_kernel + sl_root_gather_kernel do; returns (Lpool, root matrix (upper), n_neg, n_zero)"""
    lpool, upool = np.zeros(max(P["lsz"], 1)), np.zeros(max(P["usz"], 1))
    nneg = nzero = 0
    for lev in range(P["nlev"]):
        for q in range(P["level_ptr"][lev], P["level_ptr"][lev + 1]):
            nc, nr = int(P["f_nc"][q]), int(P["f_nr"][q])
            f = nc + nr
            assert f <= 128 and nc <= 48
            F = np.zeros((f, f))
            for t in range(P["mat_front"][q], P["mat_front"][q + 1]):
                d = int(P["mat_dest"][t])
                assert d // f >= d % f                                  # lower triangle only
                F[d // f, d % f] = gather(P["mat_src"], P["mat_ptr"][t], P["mat_ptr"][t + 1], vals, upool)
            for k in range(nc):
                d = F[k, k]
                bad = not (abs(d) >= 1e-14) or not np.isfinite(d)
                nzero += bad
                nneg += (not bad) and d < 0
                di = 0.0 if bad else 1.0 / d
                v = F[k + 1:, k].copy()
                lk = v * di
                F[k + 1:, k + 1:] -= np.tril(np.outer(lk, v))
                F[k + 1:, k] = lk
            Lp = np.tril(F[:, :nc])                                       # row i, column k: i >= k (diagonal = d)
            lpool[P["f_lofs"][q]:P["f_lofs"][q] + f * nc] = Lp.reshape(-1)
            upool[P["f_uofs"][q]:P["f_uofs"][q] + nr * nr] = np.tril(F[nc:, nc:]).reshape(-1)
    r = P["r"]
    R = np.zeros((r, r))
    for t in range(P["n_rmat"]):
        d = int(P["rmat_dest"][t])
        assert d // r <= d % r                                          # upper triangle
        R[d // r, d % r] = gather(P["rmat_src"], P["rmat_ptr"][t], P["rmat_ptr"][t + 1], vals, upool)
    return lpool, R, int(nneg), int(nzero)


def replay_solve(P, lpool, R, b):
    x = b.copy()
    vpool = np.zeros(max(P["vsz"], 1))
    for lev in range(P["nlev"]):
        for q in range(P["level_ptr"][lev], P["level_ptr"][lev + 1]):
            nc, nr = int(P["f_nc"][q]), int(P["f_nr"][q])
            f = nc + nr
            idx = P["fidx"][P["f_iofs"][q]:P["f_iofs"][q] + f]
            w = np.zeros(f)
            w[:nc] = x[idx[:nc]]
            for t in range(P["vec_front"][q], P["vec_front"][q + 1]):
                w[P["vec_dest"][t]] += gather(P["vec_src"], P["vec_ptr"][t], P["vec_ptr"][t + 1], vpool, vpool)
            Lq = lpool[P["f_lofs"][q]:P["f_lofs"][q] + f * nc].reshape(f, nc)
            for k in range(nc):
                w[k + 1:] -= Lq[k + 1:, k] * w[k]
            x[idx[:nc]] = w[:nc]
            vpool[P["f_vofs"][q]:P["f_vofs"][q] + nr] = w[nc:]
    r = P["r"]
    if r:
        xr = x[P["root_old"]].copy()
        for t in range(P["n_rvec"]):
            xr[P["rvec_dest"][t]] += gather(P["rvec_src"], P["rvec_ptr"][t], P["rvec_ptr"][t + 1], vpool, vpool)
        Rs = np.triu(R) + np.triu(R, 1).T
        x[P["root_old"]] = np.linalg.solve(Rs, xr)
    for lev in range(P["nlev"] - 1, -1, -1):
        for q in range(P["level_ptr"][lev], P["level_ptr"][lev + 1]):
            nc, nr = int(P["f_nc"][q]), int(P["f_nr"][q])
            f = nc + nr
            idx = P["fidx"][P["f_iofs"][q]:P["f_iofs"][q] + f]
            Lq = lpool[P["f_lofs"][q]:P["f_lofs"][q] + f * nc].reshape(f, nc)
            w = x[idx].copy()
            w[:nc] /= np.diag(Lq[:nc, :nc])
            for k in range(nc - 1, -1, -1):
                w[k] -= Lq[k + 1:, k] @ w[k + 1:]
            x[idx[:nc]] = w[:nc]
    return x


# ---- test matrices ---------------------------------------------------------------------------------------------------------------
def banded(n, bw, seed, shift=None):
    rng = np.random.default_rng(seed)
    diags = [rng.uniform(-1, 1, n - k) for k in range(1, bw + 1)]
    A = sp.diags(diags, list(range(1, bw + 1)), shape=(n, n))
    A = A + A.T
    d = np.asarray(abs(A).sum(axis=1)).ravel() + 1.0 if shift is None else shift
    return (A + sp.diags(d)).tocsr()


def block_arrow(nblocks, bs, border, seed):
    """block-diagonal (dense bs x bs SPD blocks) + `border` dense rows coupling every block"""
    rng = np.random.default_rng(seed)
    n = nblocks * bs + border
    blocks = []
    for _ in range(nblocks):
        B = rng.uniform(-1, 1, (bs, bs))
        blocks.append(B @ B.T + bs * np.eye(bs))
    D = sp.block_diag(blocks)
    E = sp.random(nblocks * bs, border, density=min(1.0, 3.0 / border), random_state=seed, data_rvs=lambda k: rng.uniform(-1, 1, k))
    Cb = rng.uniform(-1, 1, (border, border))
    Cb = Cb @ Cb.T + (border + 4.0 * E.power(2).sum()) * np.eye(border) / border + 50 * np.eye(border)
    return sp.bmat([[D, E], [E.T, sp.csr_matrix(Cb)]]).tocsr()


def random_fill(n, deg, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=deg / n, random_state=seed, data_rvs=lambda k: rng.uniform(-1, 1, k))
    A = A + A.T
    d = np.asarray(abs(A).sum(axis=1)).ravel() + 1.0
    return (A + sp.diags(d)).tocsr()


def quasi_definite(A, nneg, seed):
    """flip the sign of the rows / columns of `nneg` variables' diagonal dominance: K = [A11 B; B^T -A22]-like, inertia known from eig"""
    rng = np.random.default_rng(seed)
    A = sp.lil_matrix(A)
    pick = rng.choice(A.shape[0], nneg, replace=False)
    for i in pick:
        A[i, i] = -A[i, i]
    return A.tocsr()


CASES = [
    ("tridiagonal", lambda: banded(700, 1, 1)),
    ("banded-7", lambda: banded(1200, 7, 2)),
    ("banded-40", lambda: banded(900, 40, 3)),
    ("block-arrow-33", lambda: block_arrow(40, 6, 33, 4)),
    ("block-arrow-200", lambda: block_arrow(30, 8, 200, 5)),
    ("random-fill", lambda: random_fill(500, 2.2, 6)),
    ("diagonal", lambda: sp.diags(np.linspace(1, 2, 300)).tocsr()),
    ("one-by-one", lambda: sp.csr_matrix(np.array([[3.0]]))),
    ("two-components", lambda: sp.block_diag([banded(130, 3, 11), banded(75, 2, 12)]).tocsr()),
    ("many-tiny-components", lambda: sp.block_diag([banded(3, 1, 20 + q) for q in range(90)] + [sp.csr_matrix(np.array([[2.0 + q]])) for q in range(40)]).tocsr()),
    ("banded-indefinite", lambda: quasi_definite(banded(800, 5, 7), 300, 8)),
    ("arrow-indefinite", lambda: quasi_definite(block_arrow(25, 6, 40, 9), 60, 10)),
]


@pytest.mark.parametrize("name,make", CASES, ids=[c[0] for c in CASES])
def test_replayed_plan_factors_and_solves(name, make):
    A = make()
    n = A.shape[0]
    rp, ci, vals = csr_full(A)
    info = np.zeros(8, dtype=np.int64)
    perm = np.zeros(n, dtype=np.int32)
    L = _lib()
    ip = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.hiopamd_sparse_ldl_analyse(n, ip(rp), ip(ci), ip(info), ip(perm)) == 0
    assert sorted(perm.tolist()) == list(range(n))
    assert info[5] <= 128
    rc, P = get_plan(n, rp, ci)
    assert rc == 0 and P["r"] == info[3] and P["nf"] == info[1]
    lpool, R, nneg, nzero = replay_factor(P, vals)
    Ad = A.toarray()
    ev = np.linalg.eigvalsh(Ad)
    if P["r"]:
        Rs = np.triu(R) + np.triu(R, 1).T
        evr = np.linalg.eigvalsh(Rs)
        nneg += int((evr < 0).sum())
        assert np.abs(evr).min() > 1e-10
    assert nzero == 0
    assert nneg == int((ev < 0).sum()), (name, nneg, int((ev < 0).sum()))
    b = np.random.default_rng(11).uniform(-1, 1, n)
    x = replay_solve(P, lpool, R, b)
    xs = np.linalg.solve(Ad, b)
    assert np.abs(x - xs).max() <= 1e-9 * max(1.0, np.abs(xs).max()), name
    # tree parallelism: a banded matrix must come out with O(log n) levels, not a path
    # (a band of 40 has separators of 40 and leaf fronts of 48 + 80 rows: most of it is handed to the dense root, which is allowed)
    if name in ("banded-7", "banded-indefinite", "tridiagonal"):
        assert P["nlev"] <= 4 * int(np.ceil(np.log2(n))) and P["r"] <= 128


def test_patterns_with_large_separators_are_refused_not_mishandled():
    """a 2-D grid of 200 x 200: separators of ~200 vertices at every level of the dissection, the dense root would be larger than the
    solver's limit only for much bigger grids; here it must either work (root <= 20480) or be refused with HIOPAMD_ERR_STATE = -5"""
    k = 60
    T = sp.diags([np.ones(k - 1), np.ones(k - 1)], [-1, 1])
    A = (sp.kron(sp.eye(k), T) + sp.kron(T, sp.eye(k)) + 5 * sp.eye(k * k)).tocsr()
    rp, ci, vals = csr_full(A)
    info = np.zeros(8, dtype=np.int64)
    L = _lib()
    ip = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.hiopamd_sparse_ldl_analyse(k * k, ip(rp), ip(ci), ip(info), C.c_void_p(0))
    assert rc in (0, -5)
    if rc == 0:
        assert info[3] <= 20480


def test_missing_diagonal_is_refused():
    A = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [2.0, 0.0, 1.0], [0.0, 1.0, 3.0]]))
    A.eliminate_zeros()
    rp, ci, _ = csr_full(A)
    info = np.zeros(8, dtype=np.int64)
    L = _lib()
    ip = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.hiopamd_sparse_ldl_analyse(3, ip(rp), ip(ci), ip(info), C.c_void_p(0)) == -5


# ---- a seeded sweep over pattern families the fixed cases do not have -----------------------------------------------------------
def _random_structure(seed):
    """one of: grid (2-D 5-point, narrow), tree-like (random recursive tree + a few chords), star-of-chains (a hub row), block chain with
    dense coupling blocks, banded with random gaps + isolated vertices, union of two banded matrices under a random permutation;
    strictly diagonally dominant with a random sign pattern on the diagonal (quasi-definite: the inertia is the diagonal's)"""
    rng = np.random.default_rng(seed)
    kind = seed % 6
    n = int(rng.integers(40, 420))
    rows, cols = [], []

    def edge(i, j):
        if i != j:
            rows.append(i); cols.append(j)

    if kind == 0:                                   # grid w x h, w small (separators of size w)
        w = int(rng.integers(3, 9)); h = max(2, n // w); n = w * h
        for y in range(h):
            for x in range(w):
                if x + 1 < w: edge(y * w + x, y * w + x + 1)
                if y + 1 < h: edge(y * w + x, (y + 1) * w + x)
    elif kind == 1:                                 # random recursive tree + chords
        for i in range(1, n): edge(i, int(rng.integers(0, i)))
        for _ in range(n // 10): edge(int(rng.integers(0, n)), int(rng.integers(0, n)))
    elif kind == 2:                                 # chains hanging off one hub (a dense row)
        for i in range(1, n):
            edge(i, 0)
            if i % 7: edge(i, i - 1)
    elif kind == 3:                                 # chain of dense blocks with dense couplings
        bs = int(rng.integers(3, 10)); nb = max(2, n // bs); n = nb * bs
        for b in range(nb):
            for i in range(bs):
                for j in range(i):
                    edge(b * bs + i, b * bs + j)
                if b + 1 < nb:
                    for j in range(bs):
                        if rng.random() < 0.6: edge(b * bs + i, (b + 1) * bs + j)
    elif kind == 4:                                 # banded with gaps, some isolated vertices
        bw = int(rng.integers(1, 6))
        for i in range(n):
            if i % 11 == 5: continue                # isolated
            for k in range(1, bw + 1):
                if i + k < n and (i + k) % 11 != 5 and rng.random() < 0.8: edge(i, i + k)
    else:                                           # two banded matrices on a shuffled numbering
        p = rng.permutation(n)
        for i in range(n - 1): edge(int(p[i]), int(p[i + 1]))
        q = rng.permutation(n)
        for i in range(0, n - 3, 2): edge(int(q[i]), int(q[i + 3]))
    v = rng.uniform(-1, 1, len(rows))
    A = sp.coo_matrix((v, (rows, cols)), shape=(n, n)).tocsr()
    A = A + A.T
    d = np.asarray(abs(A).sum(axis=1)).ravel() + rng.uniform(0.5, 2.0, n)
    sign = np.where(rng.random(n) < 0.3, -1.0, 1.0)
    return (A + sp.diags(d * sign)).tocsr(), int((sign < 0).sum())


@pytest.mark.parametrize("seed", range(48))
def test_seeded_sweep_of_pattern_families(seed):
    A, nneg_expected = _random_structure(seed)
    n = A.shape[0]
    rp, ci, vals = csr_full(A)
    rc, P = get_plan(n, rp, ci)
    assert rc == 0
    # every variable is eliminated exactly once: the fronts' pivot columns and the root partition 0 .. n-1
    piv = []
    for q in range(P["nf"]):
        piv += list(P["fidx"][P["f_iofs"][q]:P["f_iofs"][q] + P["f_nc"][q]])
    piv += list(P["root_old"][:P["r"]])
    assert sorted(int(i) for i in piv) == list(range(n))
    lpool, R, nneg, nzero = replay_factor(P, vals)
    Rs = np.triu(R) + np.triu(R, 1).T
    ev_root = np.linalg.eigvalsh(Rs) if P["r"] else np.zeros(0)
    assert nzero == 0 and nneg + int((ev_root < 0).sum()) == nneg_expected
    b = np.random.default_rng(seed + 1000).uniform(-1, 1, n)
    x = replay_solve(P, lpool, R, b)
    assert np.abs(A @ x - b).max() <= 1e-10 * max(1.0, np.abs(x).max())


def test_the_analysis_does_not_depend_on_the_number_of_host_threads():
    """Round 6: the nested dissection forks at its top levels and the plan lists are sorted on several threads.  The ordering, the
    supernodes and every gather plan must be what the sequential analysis produces (so that every sum of the numeric phase — and every bit
    of the factor — is independent of the host the analysis ran on).  n = 3e5 banded: sides of >= 20 000 vertices fork three levels deep."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from test_sparse_ldl_plan import banded, csr_full, get_plan\n"
        "A = banded(300000, 4, seed=11); rp, ci, v = csr_full(A)\n"
        "rc, P = get_plan(300000, rp, ci); assert rc == 0\n"
        "h = hashlib.sha256()\n"
        "for k in sorted(P):\n"
        "    h.update(k.encode()); h.update(np.ascontiguousarray(P[k]).tobytes() if isinstance(P[k], np.ndarray) else str(P[k]).encode())\n"
        "print(h.hexdigest())\n"
    ) % (str(__import__("pathlib").Path(__file__).resolve().parent.parent), str(__import__("pathlib").Path(__file__).resolve().parent))
    digests = []
    for threads in ("1", "8"):
        env = dict(os.environ, HIOPAMD_HOST_THREADS=threads)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(r.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1]
