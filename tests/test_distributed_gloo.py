"""Column-sharded (memory-distributed) dense-constraint path, world_size = 2 over gloo on CPU.

The reference distributes every O(n) object by columns and all-reduces the small blocks (SURVEY.md §2.2:
hiopHessianLowRank.cpp:459,590-591; hiopMatrixDenseRowMajor.cpp:466,487 "only rank 0 applies beta").
These tests run the oracle's restatement on 2 ranks (each holding a column slice, reductions through
torch.distributed/gloo) and require the result to equal the single-rank computation — i.e. they check the
host-side sharding rules the HIP path implements with the same hook points (partition arithmetic, which
blocks are reduced, rank-0-only terms).  No GPU needed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def col_partition(n, P):
    """quotient/remainder split used by the reference drivers (src/Drivers/Dense/NlpDenseConsEx2.cpp:25-39)."""
    q, rem = divmod(n, P)
    cols = [0]
    for r in range(P):
        cols.append(cols[-1] + q + (1 if r < rem else 0))
    return cols


def _problem(n, me, mi, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    q = r.uniform(0.5, 3.0, n)
    Jc = r.uniform(-1, 1, (me, n)); Jd = r.uniform(-1, 1, (mi, n))
    xs = [r.uniform(-1, 1, n)]
    for _ in range(8):
        xs.append(xs[-1] + r.uniform(-0.2, 0.2, n))
    ycs = [r.uniform(-0.1, 0.1, me) for _ in range(9)]
    yds = [r.uniform(-0.1, 0.1, mi) for _ in range(9)]
    Dx = r.uniform(0, 2, n) * (r.uniform(0, 1, n) < 0.5)
    Dd = r.uniform(0.5, 2, mi)
    rx, ryc, ryd = r.uniform(-1, 1, n), r.uniform(-1, 1, me), r.uniform(-1, 1, mi)
    return q, Jc, Jd, xs, ycs, yds, Dx, Dd, rx, ryc, ryd


def _run(H, K, sl, prob):
    q, Jc, Jd, xs, ycs, yds, Dx, Dd, rx, ryc, ryd = prob
    for it, x in enumerate(xs):
        H.update(x[sl], (q * x)[sl], Jc[:, sl], Jd[:, sl], ycs[it], yds[it])
    K.update(Dx[sl], Dd, Jc[:, sl], Jd[:, sl])
    ok, dx, dyc, dyd = K.solve_compressed(rx[sl].copy(), ryc, ryd)
    y = np.zeros(dx.size)
    H.times_vec(0.0, y, 1.0, rx[sl])
    return ok, dx, dyc, dyd, y, H.sigma, K.last_N


def _worker(rank, world, port, n, me, mi, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import hiop_oracle as ho

    def allreduce(buf):
        t = torch.from_numpy(np.ascontiguousarray(buf))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    prob = _problem(n, me, mi, seed)
    cols = col_partition(n, world)
    sl = slice(cols[rank], cols[rank + 1])
    H = ho.HessianLowRank(sl.stop - sl.start, l_max=6, sigma0=1.0, sigma_update_strategy="sty", rank=rank,
                          allreduce=allreduce)

    def allreduce_max(buf):
        t = torch.from_numpy(np.ascontiguousarray(buf))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.numpy()
    H.allreduce_max = allreduce_max
    K = ho.KKTLinSysLowRank(H, me, mi)
    ok, dx, dyc, dyd, y, sigma, N = _run(H, K, sl, prob)
    # gather the distributed pieces on rank 0
    parts = [None] * world
    dist.all_gather_object(parts, (dx, y))
    if rank == 0:
        q.put((ok, np.concatenate([p[0] for p in parts]), dyc, dyd, np.concatenate([p[1] for p in parts]), sigma, N))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,me,mi", [(101, 2, 3), (64, 1, 0)])
def test_sharded_lowrank_kkt_equals_single_rank(n, me, mi):
    from oracle import hiop_oracle as ho
    seed = 11
    prob = _problem(n, me, mi, seed)
    H1 = ho.HessianLowRank(n, l_max=6, sigma0=1.0, sigma_update_strategy="sty")
    K1 = ho.KKTLinSysLowRank(H1, me, mi)
    ref = _run(H1, K1, slice(0, n), prob)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, me, mi, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ok, dx, dyc, dyd, y, sigma, N = got
    assert ok and ref[0]
    assert sigma == pytest.approx(ref[5], rel=1e-12)
    np.testing.assert_allclose(N, ref[6], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dx, ref[1], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dyc, ref[2], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dyd, ref[3], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(y, ref[4], rtol=1e-9, atol=1e-10)


def _full_space_run(H, K, sl, prob, rank_allreduce=None):
    """Full-space layer (update -> compute_directions_w_IR) on one rank's slice; returns the direction parts."""
    from oracle import kkt_full as kf
    from tests import kkt_full_cases as cases
    q, Jc, Jd, xs, ycs, yds, Dx, Dd, rx, ryc, ryd = prob
    n, me, mi = Jc.shape[1], Jc.shape[0], Jd.shape[0]
    for it, x in enumerate(xs):
        H.update(x[sl], (q * x)[sl], Jc[:, sl], Jd[:, sl], ycs[it], yds[it])
    r = np.random.Generator(np.random.PCG64(99))
    ixl = (r.uniform(0, 1, n) < 0.7).astype(np.float64)
    ixu = (r.uniform(0, 1, n) < 0.3).astype(np.float64)
    idl = np.ones(mi)
    idu = (r.uniform(0, 1, mi) < 0.5).astype(np.float64)
    it_full = cases.random_iterate(n, mi, me, mi, ixl, ixu, idl, idu, seed=3)
    res_full = cases.random_resid(kf.part_sizes(n, mi, me, mi), ixl, ixu, idl, idu, seed=5)
    xparts = ("x", "sxl", "sxu", "zl", "zu", "rx", "rxl", "rxu", "rszl", "rszu")
    loc = lambda d: {k: (v[sl] if k in xparts else v) for k, v in d.items()}
    prov = kf.LowRankProvider(K, Jc[:, sl], Jd[:, sl])
    full = kf.KKTLinSysFull(prov, ixl[sl], ixu[sl], idl, idu, perturb=kf.PDPerturbationNull())
    assert full.update(loc(it_full))
    dot = kf.sharded_dot(full.sizes, rank_allreduce) if rank_allreduce else None
    ok, d, info = full.compute_directions_w_IR(loc(res_full), mu=1e-8, dot=dot)
    y = full.times_vec(d)
    return ok, d, info, y


def _worker_full(rank, world, port, n, me, mi, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import hiop_oracle as ho

    def allreduce(buf):
        t = torch.from_numpy(np.ascontiguousarray(buf))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def allreduce_max(buf):
        t = torch.from_numpy(np.ascontiguousarray(buf))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.numpy()

    prob = _problem(n, me, mi, seed)
    cols = col_partition(n, world)
    sl = slice(cols[rank], cols[rank + 1])
    H = ho.HessianLowRank(sl.stop - sl.start, l_max=6, sigma0=1.0, sigma_update_strategy="sty", rank=rank,
                          allreduce=allreduce)
    H.allreduce_max = allreduce_max
    K = ho.KKTLinSysLowRank(H, me, mi)
    ok, d, info, y = _full_space_run(H, K, sl, prob, allreduce)
    parts = [None] * world
    dist.all_gather_object(parts, (d, y))
    if rank == 0:
        q.put((ok, parts, info))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,me,mi", [(101, 2, 3)])
def test_sharded_full_space_ir_equals_single_rank(n, me, mi):
    """12-part compound vectors on a column partition: x-sized parts sliced, dual-sized parts replicated; the
    BiCGStab scalars must come out identical on every rank (hiopVectorCompoundPD dot-product semantics)."""
    from oracle import hiop_oracle as ho
    from oracle import kkt_full as kf
    seed = 11
    prob = _problem(n, me, mi, seed)
    H1 = ho.HessianLowRank(n, l_max=6, sigma0=1.0, sigma_update_strategy="sty")
    K1 = ho.KKTLinSysLowRank(H1, me, mi)
    ok1, d1, info1, y1 = _full_space_run(H1, K1, slice(0, n), prob)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_full, args=(r, 2, port, n, me, mi, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, parts, info = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and ok1 and info["converged"] and info1["converged"] and info["iter"] == info1["iter"]
    xparts = ("x", "sxl", "sxu", "zl", "zu")
    for k in kf.ITER_PARTS:
        got = np.concatenate([p[0][k] for p in parts]) if k in xparts else parts[0][0][k]
        np.testing.assert_allclose(got, d1[k], rtol=1e-8, atol=1e-10, err_msg=k)
        if k not in xparts:     # replicated parts are identical on both ranks
            np.testing.assert_array_equal(parts[0][0][k], parts[1][0][k])
    rx = ("rx", "rxl", "rxu", "rszl", "rszu")
    for k in kf.RESID_PARTS:
        got = np.concatenate([p[1][k] for p in parts]) if k in rx else parts[0][1][k]
        np.testing.assert_allclose(got, y1[k], rtol=1e-8, atol=1e-10, err_msg=k)


def test_col_partition_matches_reference_rule():
    assert col_partition(10, 3) == [0, 4, 7, 10]
    assert col_partition(8, 8) == list(range(9))
    assert col_partition(10_000_000, 8)[-1] == 10_000_000
