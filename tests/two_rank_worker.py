"""Worker of tests/test_gpu_two_rank.py: one rank of a 2-rank column partition, BOTH ranks on the same MI355X.

The product library runs with `comm_size = 2`: the context's all-reduce hook (hiopamd_ctx_set_allreduce) is a host
callback that stages the small device buffer through the host and reduces it with torch.distributed / gloo — RCCL
refuses two ranks on one device, the hook is the same plug point RCCL uses on a real node (csrc/context.hip).
Everything that is computed here goes through the C ABI; the oracle is not involved."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

X_PARTS_IT = ("x", "sxl", "sxu", "zl", "zu")
X_PARTS_RES = ("rx", "rxl", "rxu", "rszl", "rszu")


def col_partition(n, P):
    """quotient/remainder split of the reference drivers (src/Drivers/Dense/NlpDenseConsEx2.cpp:25-39)."""
    q, rem = divmod(n, P)
    cols = [0]
    for r in range(P):
        cols.append(cols[-1] + q + (1 if r < rem else 0))
    return cols


def make_problem(n, me, mi, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    p = {}
    p["q"] = r.uniform(0.5, 3.0, n)
    p["Jc"] = r.uniform(-1, 1, (me, n)); p["Jd"] = r.uniform(-1, 1, (mi, n))
    xs = [r.uniform(-1, 1, n)]
    for _ in range(8):
        xs.append(xs[-1] + r.uniform(-0.2, 0.2, n))
    p["xs"] = xs
    p["ycs"] = [r.uniform(-0.1, 0.1, me) for _ in range(9)]
    p["yds"] = [r.uniform(-0.1, 0.1, mi) for _ in range(9)]
    p["Dx"] = r.uniform(0, 2, n) * (r.uniform(0, 1, n) < 0.5)
    p["Dd"] = r.uniform(0.5, 2, mi)
    p["rx"], p["ryc"], p["ryd"] = r.uniform(-1, 1, n), r.uniform(-1, 1, me), r.uniform(-1, 1, mi)
    p["ixl"] = (r.uniform(0, 1, n) < 0.7).astype(np.float64)
    p["ixu"] = (r.uniform(0, 1, n) < 0.3).astype(np.float64)
    p["idl"] = np.ones(mi)
    p["idu"] = (np.arange(mi) % 2 == 0).astype(np.float64)
    p["xl"] = np.where(p["ixl"] == 1.0, r.uniform(-3, -2, n), -1e20)
    p["xu"] = np.where(p["ixu"] == 1.0, r.uniform(2, 3, n), 1e20)
    p["dl"] = np.where(p["idl"] == 1.0, r.uniform(-3, -2, mi), -1e20)
    p["du"] = np.where(p["idu"] == 1.0, r.uniform(2, 3, mi), 1e20)
    p["crhs"] = r.uniform(-1, 1, me)
    return p


def slice_parts(parts, sl, xnames):
    return {k: (v[sl] if k in xnames else v) for k, v in parts.items()}


def run_partition(ctx, prob, sl, tiny_slack_at=None):
    """The sharded sequence on the slice `sl` of the columns: secant updates -> KKT update -> solveCompressed -> full-space
    update -> compute_directions_w_IR -> adjust_small_slacks.  Returns host numpy results (local slices of distributed
    vectors, whole replicated ones)."""
    import torch
    from hiop_amd.kkt import HessianLowRank, KKTLinSysLowRank, KKTLinSysXYcYd, IpmSlabOps, ITER_PARTS, RESID_PARTS
    from tests import kkt_full_cases as cases

    def D(a):
        return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float64).cuda()

    n = prob["q"].size
    me, mi = prob["Jc"].shape[0], prob["Jd"].shape[0]
    nl = sl.stop - sl.start
    J = D(np.vstack([prob["Jc"][:, sl], prob["Jd"][:, sl]]))
    Jc, Jd = J[:me], J[me:]
    H = HessianLowRank(ctx, nl, me, mi, l_max=6, sigma0=1.0, sigma_update_strategy="sty")
    stored = []
    for it, x in enumerate(prob["xs"]):
        args = [D(x[sl]), D((prob["q"] * x)[sl]), Jc, Jd, D(prob["ycs"][it]), D(prob["yds"][it])]
        torch.cuda.synchronize()
        stored.append(H.update(*args))
        ctx.sync()
        del args
    K = KKTLinSysLowRank(ctx, H)
    # NOTE: the C ABI is asynchronous on the context's stream.  Device temporaries handed to a call must outlive the kernels
    # that read them: keep them in variables and synchronise before they are released (torch's caching allocator would
    # recycle the block for the next upload while the kernels of the call are still queued — with two processes sharing
    # the GPU that window is wide, and 1 / Dd was once computed from the bytes of the next test vector).
    Dx_d, Dd_d = D(prob["Dx"][sl]), D(prob["Dd"])
    torch.cuda.synchronize()
    K.update_diag(Dx_d, Dd_d, Jc, Jd)
    ctx.sync()
    rx = D(prob["rx"][sl])
    dx, dyc, dyd = D(np.zeros(nl)), D(np.zeros(me)), D(np.zeros(mi))
    ryc_d, ryd_d = D(prob["ryc"]), D(prob["ryd"])
    torch.cuda.synchronize()
    ok = K.solve_compressed(rx, ryc_d, ryd_d, dx, dyc, dyd)
    ctx.sync()
    out = dict(stored=stored, sigma=H.sigma, ok=ok, dx=dx.cpu().numpy(), dyc=dyc.cpu().numpy(), dyd=dyd.cpu().numpy(),
               N=K.N().cpu().numpy())
    # full-space layer on the same partition
    ixl, ixu, idl, idu = prob["ixl"], prob["ixu"], prob["idl"], prob["idu"]
    pats = [D(ixl[sl]), D(ixu[sl]), D(idl), D(idu)]
    fg = KKTLinSysXYcYd(ctx, K, *pats)
    fg.set_matrices(None, Jc, Jd)
    it_full = cases.random_iterate(n, mi, me, mi, ixl, ixu, idl, idu, seed=3)
    sizes = [n, mi, me, mi, n, n, mi, mi, n, n, mi, mi]
    res_full = cases.random_resid(sizes, ixl, ixu, idl, idu)
    it_g = fg.pack(slice_parts(it_full, sl, X_PARTS_IT), ITER_PARTS)
    r_g = fg.pack(slice_parts(res_full, sl, X_PARTS_RES), RESID_PARTS)
    torch.cuda.synchronize()
    assert fg.update(it_g)
    d_g = torch.zeros(fg.dim, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ok_ir, info = fg.compute_directions_w_IR(r_g, d_g)
    ctx.sync()
    out["ir_ok"], out["ir_info"] = ok_ir, info
    out["dir"] = fg.unpack(d_g, ITER_PARTS)
    acc, dWd, nrm = fg.test_direction(d_g)
    out["test_direction"] = (acc, dWd, nrm)
    # hiopIterate::adjust_small_slacks with ONE tiny slack (global index tiny_slack_at): min_w_pattern and the adjusted
    # count are all-reduced in the reference (hiopVectorPar.cpp:833-836, :1231-1236) -> every rank must return 1
    bnds = [D(prob["xl"][sl]), D(prob["xu"][sl]), D(prob["dl"]), D(prob["du"]), D(prob["crhs"])]
    ops = IpmSlabOps(fg, *bnds)
    it2 = {k: v.copy() for k, v in it_full.items()}
    if tiny_slack_at is not None:
        assert ixl[tiny_slack_at] == 1.0
        it2["sxl"][tiny_slack_at] = 1e-30
    it2_g = fg.pack(slice_parts(it2, sl, X_PARTS_IT), ITER_PARTS)
    torch.cuda.synchronize()
    out["num_adjusted"] = ops.adjust_small_slacks(it2_g, it_g, 1e-3)
    ctx.sync()
    out["sxl_adjusted"] = fg.unpack(it2_g, ITER_PARTS)["sxl"]
    # reductions of the step routines on the partition
    out["log_barrier"] = ops.eval_log_barrier(it_g)
    out["ftb"] = ops.fraction_to_the_bdry(it_g, d_g, 0.995)
    fg.close(); K.close(); H.close()
    return out


def install_gloo_hook(ctx, rank, world):
    """all-reduce hook = device buffer -> host -> gloo -> device, on the context's stream order (the copies synchronise
    the stream).  Returns the ctypes callback object (the caller keeps it alive)."""
    import torch
    import torch.distributed as dist
    from hiop_amd._lib import ALLREDUCE_FN, lib, check
    L = lib()
    calls = {"n": 0, "doubles": 0, "log": []}

    def hook(user, buf, count, op, stream):
        try:
            host = np.empty(int(count), dtype=np.float64)
            if L.hiopamd_copy_d2h(ctx.h, C.c_void_p(host.ctypes.data), C.c_void_p(buf), int(count) * 8) != 0:
                return -1
            t = torch.from_numpy(host)
            rop = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[int(op)]
            before = float(np.abs(host).sum())
            dist.all_reduce(t, op=rop)
            calls["log"].append((int(count), int(op), before, float(np.abs(host).sum())))
            if L.hiopamd_copy_h2d(ctx.h, C.c_void_p(buf), C.c_void_p(host.ctypes.data), int(count) * 8) != 0:
                return -1
            calls["n"] += 1
            calls["doubles"] += int(count)
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            print("allreduce hook failed:", e, file=sys.stderr)
            return -1

    cb = ALLREDUCE_FN(hook)
    check(L.hiopamd_ctx_set_allreduce(ctx.h, cb, None, rank, world), "hiopamd_ctx_set_allreduce")
    return cb, calls


def worker(rank, world, port, n, me, mi, seed, tiny_at, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hiop_amd.runtime import Context
        ctx = Context(0)
        cb, calls = install_gloo_hook(ctx, rank, world)
        prob = make_problem(n, me, mi, seed)
        cols = col_partition(n, world)
        sl = slice(cols[rank], cols[rank + 1])
        out = run_partition(ctx, prob, sl, tiny_slack_at=tiny_at)
        out["allreduce_calls"] = dict(calls)
        gathered = [None] * world
        dist.all_gather_object(gathered, out)
        if rank == 0:
            outq.put(gathered)
        dist.barrier()
        ctx.close()
        del cb
    finally:
        dist.destroy_process_group()
