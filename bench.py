#!/usr/bin/env python3
"""bench.py — KKT assemble+factor+solve iterations/sec (fp64) on MI355X.

Workload at N=1 (BASELINE.json `metric`, configs[2] — the configuration the metric is quoted on):
  "NlpMDS_ex4 Newton, n_sparse=1e5 n_dense=4096 m=4096": the condensed mixed-dense-sparse KKT system
  of hiopKKTLinSysCompressedMDSXYcYd (N = n_dense + m = 8192) on the generalised MdsEx1 problem
  (hiop_amd/problems.py::mds_ex1_g, SURVEY.md §8d "MdsEx1-g").
One "step" = one IPM iteration's worth of KKT work, inputs already resident in HBM:
  1 x build_kkt_matrix  (zero N^2, scatter dense blocks, 3 sparse Schur row-builds, diagonals)
  1 x factorizeWithCurvCheck (blocked no-pivot LDL^T on fp64 MFMA + inertia, returned to the host)
  3 x solveCompressed  (rhs reduction, forward/backward substitution, recovery of the sparse part)
(the 1+1+3 split is SURVEY.md §8d's "ideal C3 iteration": initial solve + one BiCGStab refinement
iteration = 2 preconditioner solves).

Multi-GPU (`--gpus N`, launched by torch.distributed.run): the MDS path does not shard (the reference's
MDS interface is explicitly single-rank, src/Interface/hiopInterface.hpp:582-584) -> "replicas only":
every rank runs the same workload on its own GPU, value = N * steps / max-over-ranks time, scaling weak.
The path that DOES shard — the memory-distributed dense-constraint quasi-Newton KKT (variables split by
columns, RCCL all-reduce of the small blocks) — is measured in the same run at every N and reported in the
`dense_sharded` object of the same JSON line (weak scaling: n_local per GPU fixed).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak (dense), SURVEY.md §8d
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ns", type=int, default=50000, help="x (and s) sparse variables: n_sparse = 2*ns")
    ap.add_argument("--nd", type=int, default=4096)
    ap.add_argument("--neq", type=int, default=4093)
    ap.add_argument("--solves", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--dense-nlocal", type=int, default=1_250_000, help="columns per GPU of the sharded dense case")
    ap.add_argument("--dense-k", type=int, default=200, help="number of dense constraints m")
    ap.add_argument("--no-dense", action="store_true")
    return ap.parse_args()


def cpu_baseline(p, Dx, Dd, rhs, nsolves, steps):
    """The oracle's restatement of the same step (numpy + scipy-OpenBLAS LAPACK DSYTRF/DSYTRS, i.e. the
    reference's CPU path, src/LinAlg/hiopLinSolverSymDenseLapack.hpp) on the host cores, bounded sample."""
    from oracle import hiop_oracle as ho
    k = ho.KKTLinSysCompressedMDSXYcYd(p.nxs, p.nxd, p.neq, p.nineq, (p.Jcs_i, p.Jcs_j), (p.Jds_i, p.Jds_j),
                                       (p.Hss_i, p.Hss_j))
    k.set_values(p.Jcs_v, p.Jds_v, p.Hss_v, p.Jcd, p.Jdd, p.Hdd, Dx, Dd)
    rx, ryc, ryd = rhs

    def one_step():
        t0 = time.perf_counter()
        k.build_kkt_matrix(0.0, 0.0, 0.0, 0.0)
        nneg = k.factorize_with_curv_check()
        assert nneg == p.neq + p.nineq
        for _ in range(nsolves):
            k.solve_compressed(rx, ryc, ryd)
        return time.perf_counter() - t0

    # DSYTRF in OpenBLAS does not scale to every hardware thread of the box (128 threads were SLOWER than the reference's own
    # 8-core run): one step at each of a few thread counts, then the remaining steps at the best one
    tried = {}
    try:
        from threadpoolctl import threadpool_limits
        ncpu = os.cpu_count() or 1
        for thr in sorted({t for t in (16, 32, 64) if t <= ncpu} or {ncpu}):
            with threadpool_limits(limits=thr):
                tried[thr] = one_step()
        best = min(tried, key=tried.get)
        ctxm = threadpool_limits(limits=best)
    except Exception:
        import contextlib
        best, ctxm = os.cpu_count() or 1, contextlib.nullcontext()
    with ctxm:
        times = [one_step() for _ in range(max(1, steps - 1))]
    dt = sum(times)
    return dict(value=len(times) / dt, unit="KKT iterations/s", cores=int(best), kind="port",
                threads_tried={str(t): round(v, 3) for t, v in tried.items()},
                sample=f"{len(times)} steps of the same workload (N={p.N}; build + DSYTRF + {nsolves}x DSYTRS) at the best of the "
                       f"tried OpenBLAS thread counts ({best}), {dt:.1f} s (+ {sum(tried.values()):.1f} s of one-step trials), "
                       f"on the host via oracle/hiop_oracle.py (numpy + scipy-OpenBLAS LAPACK)")


def _fake_multi():
    """HIOPAMD_BENCH_FAKE_MULTI=1 (test aid, never set by the driver): run the N > 1 code path on ONE GPU — every rank on device
    0, torch.distributed over gloo, the library's all-reduce hook staged through the host instead of RCCL.  Timings are
    meaningless; it exists so that the multi-rank control flow can be exercised where only one GPU is available."""
    on = os.environ.get("HIOPAMD_BENCH_FAKE_MULTI", "0") == "1"
    if on:
        # Two PROCESSES on one device cannot both run the dataflow factorisation: each needs its 16 chain workgroups resident
        # at the same time on the same 16 reserved CUs (one workgroup per CU: 147 KB of LDS), and the dispatcher may give each
        # process half of them — both pairs of persistent kernels then wait for roles that can never start (the bounded waits
        # turn that into an error after 3 s, which is how the rehearsal found it).  One process per GPU, the real
        # configuration, does not have the problem; the rehearsal uses the stepwise kernels.
        os.environ["HIOPAMD_DF"] = "0"
    return on


def _install_host_allreduce(ctx, dist):
    """generic all-reduce hook of the library (hiopamd_ctx_set_allreduce) through host staging + gloo"""
    import ctypes as C
    import numpy as np
    import torch
    from hiop_amd._lib import ALLREDUCE_FN
    L = ctx._L

    def hook(user, buf, count, op, stream):
        try:
            host = np.empty(int(count), dtype=np.float64)
            if L.hiopamd_copy_d2h(ctx.h, C.c_void_p(host.ctypes.data), C.c_void_p(buf), int(count) * 8) != 0:
                return -1
            rop = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[int(op)]
            dist.all_reduce(torch.from_numpy(host), op=rop)
            if L.hiopamd_copy_h2d(ctx.h, C.c_void_p(buf), C.c_void_p(host.ctypes.data), int(count) * 8) != 0:
                return -1
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            print("allreduce hook failed:", e, file=sys.stderr)
            return -1
    ctx._allreduce_cb = ALLREDUCE_FN(hook)     # keep the trampoline alive
    rc = L.hiopamd_ctx_set_allreduce(ctx.h, ctx._allreduce_cb, None, dist.get_rank(), dist.get_world_size())
    if rc != 0:
        raise RuntimeError(f"hiopamd_ctx_set_allreduce failed: {rc}")


def dense_lowrank_bench(ctx, world, rank, a, dist, hooked=True, rooflines=False):
    """Memory-distributed dense-constraint case (BASELINE configs[1]/[3]): quasi-Newton low-rank KKT with the
    variables column-sharded over the GPUs of the node, n_local per GPU fixed (weak scaling), k = m constraints,
    l = 6 secant pairs; small blocks all-reduced with RCCL over xGMI through the context hook.
    One step = hiopHessianLowRank::update + hiopKKTLinSysLowRank::update + `solves` x solveCompressed."""
    import torch
    import ctypes as C
    from hiop_amd.kkt import HessianLowRank, KKTLinSysLowRank
    from hiop_amd.runtime import dptr
    n, me, mi, l = a.dense_nlocal, a.dense_k // 2, a.dense_k - a.dense_k // 2, 6
    hooked = hooked and world > 1
    if hooked:
        if _fake_multi():
            _install_host_allreduce(ctx, dist)
        else:
            ctx.init_rccl_from_torch_distributed()
    gl = torch.Generator(device="cuda"); gl.manual_seed(1000 + rank)     # local (sharded) data
    gr = torch.Generator(device="cuda"); gr.manual_seed(7)               # replicated data
    U = lambda g, *shape, lo=-1.0, hi=1.0: torch.rand(*shape, generator=g, device="cuda", dtype=torch.float64) * (hi - lo) + lo
    J = U(gl, me + mi, n)                    # [Jc; Jd] stored as one block: the KKT object borrows it (no 2 GB copy per update)
    Jc, Jd = J[:me], J[me:]
    q = U(gl, n, lo=0.5, hi=3.0)
    x = U(gl, n)
    H = HessianLowRank(ctx, n, me, mi, l_max=l, sigma0=1.0, sigma_update_strategy="sty")
    K = KKTLinSysLowRank(ctx, H)
    Dx = U(gl, n, lo=0.0, hi=2.0); Dd = U(gr, mi, lo=0.5, hi=2.0)
    rx0 = U(gl, n); ryc, ryd = U(gr, me), U(gr, mi)
    rx = rx0.clone()
    dx, dyc, dyd = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros_like(ryc), torch.zeros_like(ryd)
    yc, yd = U(gr, me, lo=-0.1, hi=0.1), U(gr, mi, lo=-0.1, hi=0.1)
    steps_x = [U(gl, n, lo=-0.05, hi=0.05) for _ in range(4)]
    torch.cuda.synchronize()

    xs = [x.clone(), x.clone()]           # the iterate alternates between two buffers (no allocation inside the step)
    g = torch.empty_like(x)

    # the C ABI called like a C++ caller does: device addresses resolved once (persistent buffers), one ctypes call per entry point
    # (see the same remark at the MDS step below)
    Lh = ctx._L
    PP = lambda t: C.c_void_p(t.data_ptr())
    p_xs = [PP(xs[0]), PP(xs[1])]
    p_g, p_Jc, p_Jd, p_yc, p_yd, p_Dx, p_Dd = PP(g), PP(Jc), PP(Jd) if mi > 0 else C.c_void_p(0), PP(yc), PP(yd), PP(Dx), PP(Dd)
    p_rx, p_rx0, p_ryc, p_ryd, p_dx, p_dyc, p_dyd = PP(rx), PP(rx0), PP(ryc), PP(ryd), PP(dx), PP(dyc), PP(dyd)
    stored, okc = C.c_int(0), C.c_int(0)
    nbytes_rx = rx.numel() * 8

    def step(i):
        # the stand-in for the NLP side (new iterate, its gradient) runs on the CONTEXT's stream like everything else: no
        # cross-stream hand-off, no host synchronisation besides the ones the reference's API implies (update() returns
        # whether the pair was stored, solveCompressed() whether the reduced system was positive definite)
        xn, xo = xs[(i + 1) & 1], xs[i & 1]
        torch.add(xo, steps_x[i % 4], out=xn)
        torch.mul(q, xn, out=g)
        rc = Lh.hiopamd_hess_lowrank_update(H.h, p_xs[(i + 1) & 1], p_g, p_Jc, p_Jd, p_yc, p_yd, C.byref(stored))
        rc = rc or Lh.hiopamd_kkt_lowrank_update_diag(K.h, p_Dx, p_Dd, p_Jc, p_Jd)
        for _ in range(a.solves):
            rc = rc or Lh.hiopamd_copy_d2d(ctx.h, p_rx, p_rx0, nbytes_rx)
            rc = rc or Lh.hiopamd_kkt_lowrank_solve_compressed(K.h, p_rx, p_ryc, p_ryd, p_dx, p_dyc, p_dyd, C.byref(okc))
            if rc != 0 or not okc.value:
                raise RuntimeError(f"status {rc}; reduced system SPD: {okc.value}")

    def barrier():
        ctx.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    with torch.cuda.stream(ctx.torch_stream):
        for i in range(8 + a.warmup):      # fill the secant memory, then warm up
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(8 + a.warmup + i)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if _fake_multi() else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    coll = None
    if hooked:
        # a second pass of the same steps with the hook's calls counted and bracketed by HIP events on the context's stream (not the
        # timed pass: the events cost a few microseconds per collective): how many collectives a step makes and what they cost
        import ctypes as C2
        Lc = ctx._L
        with torch.cuda.stream(ctx.torch_stream):
            Lc.hiopamd_ctx_collective_stats_begin(ctx.h, 1)
            for i in range(a.steps):
                step(8 + a.warmup + a.steps + i)
            cnt, cms, nr = C2.c_int64(0), C2.c_double(0.0), C2.c_int(0)
            Lc.hiopamd_ctx_collective_stats_read(ctx.h, C2.byref(cnt), C2.byref(cms))
            Lc.hiopamd_ctx_rccl_ranks(ctx.h, C2.byref(nr))
            barrier()
        coll = dict(collectives_per_step=cnt.value / a.steps, collective_ms_per_step=cms.value / a.steps, rccl_ranks=nr.value)
    k = me + mi
    out = dict(value=a.steps / dt, unit="KKT iterations/s", ms_per_step=1e3 * dt / a.steps, scaling="weak",
               workload=f"NlpDenseCons quasi-Newton low-rank KKT, n_local={n} per GPU (n={n * world}), m={k}, l={l}; "
                        f"step = Hessian secant update + KKT update + {a.solves} solveCompressed "
                        f"(N = J (H+Dx)^-1 J^T formed once per step, cached for the other solves; collectives per step: 3 in the secant update, "
                        f"1 for N, 1 per solveCompressed)",
               collective=(("host-staged gloo all-reduce (HIOPAMD_BENCH_FAKE_MULTI rehearsal), %d ranks" if _fake_multi()
                            else "RCCL all-reduce (ncclAllReduce on the context stream), %d ranks") % world) if hooked
               else "none (single rank, no hook)",
               hbm_gb_J_per_gpu=8.0 * k * n / 1e9)
    if coll is not None:
        # rccl_ranks = ncclCommCount of the communicator behind the hook (0 in the host-staged rehearsal); collective_ms_per_step = device
        # time between HIP events around every hook call of a step, summed (max over ranks is NOT taken: rank 0's view)
        out.update(coll)
    if rooflines:
        # the two kernels that carry the step, timed alone with HIP events on the context's stream (20 calls each):
        #  * weighted stacked Gram  G = J DhInv [J; S; Y]^T  (fp64 MFMA): algorithmic flops = the unique entries only,
        #    2 n (k(k+1)/2 + 2 l k)   ("symmetric half counted")
        #  * GEMV  y = J x  (HBM): algorithmic bytes = 8 n (k + 1) + 8 k
        L = ctx._L
        kw = k + 2 * l
        G = torch.zeros(k * kw, dtype=torch.float64, device="cuda")
        St, Yt = C.c_void_p(L.hiopamd_hess_lowrank_St(H.h)), C.c_void_p(L.hiopamd_hess_lowrank_Yt(H.h))
        lc = H.l_curr
        yk = torch.zeros(k, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()

        def gram():
            L.hiopamd_gram_weighted_stacked(ctx.h, k, n, dptr(J), n, k, dptr(J), n, lc, St, n, lc, Yt, n, dptr(q), 0.0,
                                            dptr(G), kw, 1.0)

        def gemv():
            L.hiopamd_mat_times_vec(ctx.h, k, n, dptr(J), n, 0.0, dptr(yk), 1.0, dptr(x))

        def gemvt():
            L.hiopamd_mat_trans_times_vec(ctx.h, k, n, dptr(J), n, 0.0, dptr(dx), 1.0, dptr(yk))

        t_gram, t_gemv, t_gemvt = time_on_ctx_stream(ctx, gram), time_on_ctx_stream(ctx, gemv), time_on_ctx_stream(ctx, gemvt)
        gram_flops = 2.0 * n * (k * (k + 1) / 2 + 2 * lc * k)
        gemv_bytes = 8.0 * n * (k + 1) + 8.0 * k
        # HBM-side bytes per launch of the three kernels at THIS shape: committed PMC summary (scripts/calls/r05_pmc.sh part 2: the
        # kernels alone at bench.py's two shapes, FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, separate passes); null for any other shape
        tr, tr_src = {}, None
        for pmc_dir in ("r06_pmc_dense", "r05_pmc_dense"):   # (the newest committed summary)
            try:
                pd = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_dir, "summary.json")))
                tr = {kk: vv.get("hbm_bytes_per_launch") for kk, vv in pd.get(f"k{k}_n{n}", {}).items() if isinstance(vv, dict)}
                tr_src = f"profiles/{pmc_dir}/summary.json [k{k}_n{n}]" if tr else None
                break
            except Exception:
                continue
        out["roofline"] = [
            dict(kernel="gram_weighted_stacked (J DhInv [J;S;Y]^T, v_mfma_f64_16x16x4_f64)", bound="mfma",
                 achieved=gram_flops / (t_gram * 1e-3) / 1e12, peak=PEAK_FP64_MFMA_TFLOPS, unit="TFLOP/s",
                 frac=gram_flops / (t_gram * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS, avg_launch_ms=t_gram,
                 algorithmic_flops_per_launch=gram_flops, algorithmic_bytes_per_launch=8.0 * n * (kw + 1), hbm_gbs=8.0 * n * (kw + 1) / (t_gram * 1e-3) / 1e9,
                 traffic=tr.get("gram_weighted_stacked"), traffic_source=tr_src),
            dict(kernel="mat_times_vec (y = J x, row-major k x n_local)", bound="hbm",
                 achieved=gemv_bytes / (t_gemv * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                 frac=gemv_bytes / (t_gemv * 1e-3) / 1e9 / PEAK_HBM_GBS, avg_launch_ms=t_gemv,
                 algorithmic_bytes_per_launch=gemv_bytes, traffic=tr.get("mat_times_vec"), traffic_source=tr_src),
            dict(kernel="mat_trans_times_vec (x = J^T y)", bound="hbm",
                 achieved=gemv_bytes / (t_gemvt * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                 frac=gemv_bytes / (t_gemvt * 1e-3) / 1e9 / PEAK_HBM_GBS, avg_launch_ms=t_gemvt,
                 algorithmic_bytes_per_launch=gemv_bytes, traffic=tr.get("mat_trans_times_vec"), traffic_source=tr_src),
        ]
    K.close(); H.close()
    return out


def sparse_condensed_bench(ctx, a, n=1_000_000, pattern="sparse_ex2"):
    """BASELINE configs[4] (sparse condensed KKT + Krylov, the SpMV path): hiopKKTLinSysCondensedSparse on the SparseEx2 pattern
    (hiop_amd/problems.py::sparse_ex2_ineq: n variables, n - 1 two-entry inequality rows, diagonal Hessian), n = 1e6.
    One step = new barrier diagonals -> build_kkt_matrix (CSR J^T D J + H + Dx, numeric phase on the cached symbolic analysis) ->
    factorize -> `solves` x solveCompressed.  Inner solver: the bordered-diagonal direct factorisation (csrc/arrow_ldl.hip: SparseEx2's
    condensed matrix is diagonal + one dense row / column) when the pattern allows it, PCG + Jacobi (tol 1e-12) otherwise."""
    import torch
    import numpy as np
    from hiop_amd import problems as pr
    from hiop_amd.kkt import KKTLinSysSparseCondensed
    from hiop_amd.runtime import dev
    rng = np.random.Generator(np.random.PCG64(5))
    # pattern "chain": a banded condensed matrix (hiop_amd/problems.py::sparse_chain_ineq) — the general sparse LDL^T's case
    p = pr.sparse_ex2_ineq(n, x=rng.uniform(0.5, 2.0, n)) if pattern == "sparse_ex2" else pr.sparse_chain_ineq(n, couple=3)
    t0 = time.perf_counter()
    K = KKTLinSysSparseCondensed(ctx, p.nx, p.nineq, p.Jd_i, p.Jd_j, p.H_i, p.H_j)
    t_symbolic = time.perf_counter() - t0
    Jv, Hv = dev(p.Jd_v), dev(p.H_v)
    Dxs = [dev(rng.uniform(0, 3, n)) for _ in range(2)]
    Dds = [dev(rng.uniform(0.1, 5, p.nineq)) for _ in range(2)]
    rx, rd, ryd = dev(rng.uniform(-1, 1, n)), dev(rng.uniform(-1, 1, p.nineq)), dev(rng.uniform(-1, 1, p.nineq))
    dx, dd, dyd = torch.zeros_like(rx), torch.zeros_like(rd), torch.zeros_like(ryd)
    torch.cuda.synchronize()
    its = []

    # direct C ABI calls on resolved addresses (see the remark at the MDS step)
    import ctypes as C
    Ls = ctx._L
    PP = lambda t: C.c_void_p(t.data_ptr())
    p_Jv, p_Hv, p_Dxs, p_Dds = PP(Jv), PP(Hv), [PP(t) for t in Dxs], [PP(t) for t in Dds]
    p_rx, p_rd, p_ryd, p_dx, p_dd, p_dyd = PP(rx), PP(rd), PP(ryd), PP(dx), PP(dd), PP(dyd)
    nneg, okc, flag = C.c_int(0), C.c_int(0), C.c_int(0)
    itd, reld = C.c_double(0.0), C.c_double(0.0)
    z = C.c_double(0.0)

    def step(i):
        rc = Ls.hiopamd_kkt_sparse_condensed_set_values(K.h, p_Jv, p_Hv, p_Dxs[i & 1], p_Dds[i & 1])
        rc = rc or Ls.hiopamd_kkt_sparse_condensed_build(K.h, z, z)
        rc = rc or Ls.hiopamd_kkt_sparse_condensed_factorize(K.h, C.byref(nneg))
        if rc != 0 or nneg.value != 0:
            raise RuntimeError(f"status {rc}; condensed matrix positive definite: {nneg.value == 0}")
        for _ in range(a.solves):
            rc = Ls.hiopamd_kkt_sparse_condensed_solve_compressed(K.h, p_rx, p_rd, p_ryd, p_dx, p_dd, p_dyd, C.byref(okc))
            if rc != 0 or not okc.value:
                raise RuntimeError("inner solve failed")
            Ls.hiopamd_kkt_sparse_condensed_last_solve(K.h, C.byref(flag), C.byref(itd), C.byref(reld))
            its.append(itd.value)

    for i in range(max(a.warmup, 1)):
        step(i)
    ctx.sync(); torch.cuda.synchronize()
    its.clear()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nnzJ = int(p.Jd_v.size)
    kind = K.inner_kind()
    inner = {"bordered": "bordered-diagonal direct LDL^T (diagonal + a border of <= 32 variables: exact inertia, no iteration)",
             "pcg": "PCG + Jacobi, tol 1e-12", "dense": "dense LDL^T of the expanded matrix",
             "sparse_ldl": "sparse LDL^T: nested dissection, multifrontal by tree levels, dense root (exact inertia)"}[kind]
    out = dict(value=a.steps / dt, unit="KKT iterations/s", ms_per_step=1e3 * dt / a.steps,
               workload=f"NlpSparse condensed KKT ({'SparseEx2 pattern' if pattern == 'sparse_ex2' else 'chain constraints of 3 variables: banded condensed matrix'}, inequality-only form): n={n}, m={p.nineq}, nnz(Jd)={nnzJ}; step = "
                        f"build (CSR J^T D J + H + Dx, numeric) + factorize + {a.solves} solveCompressed ({inner})",
               inner_solver=kind, pcg_iterations_per_solve=float(np.mean(its)) if its else None, symbolic_analysis_s=t_symbolic,
               note="the sparse DIRECT solver of the reference's condensed path (MA57 / cuSOLVER Cholesky) is not in the image and not in the "
                    "reference tree (SURVEY 8c); its role is taken by the bordered-diagonal factorisation for arrowhead patterns, by the general sparse "
                    "LDL^T (nested dissection + multifrontal + dense root, csrc/sparse_ldl.hip) for other patterns, and by PCG only when "
                    "that solver's dense root would exceed its limit — a measured number, not a parity claim")
    if kind == "sparse_ldl":
        # what bounds it: the L panels are written once per factorisation and read twice per solve (forward, backward sweep) — HBM bytes —
        # but the kernels that do it are one launch per level of the supernode tree (16 levels: 16 + 3 x 32 launches per step) with a
        # dependent chain of pivot steps inside every front: the entry is latency / launch bound, and the fraction says so
        i8 = (C.c_int64 * 8)()
        if Ls.hiopamd_kkt_sparse_condensed_ldl_info(K.h, i8) == 0:
            nnzL = int(i8[4])
            bytes_step = 8.0 * nnzL * (1 + 2 * a.solves)
            out["roofline"] = dict(bound="hbm", kernel="sl_factor_regs (leaf-like levels: one wave per front, the front in registers) / sl_factor_level / sl_fwd_level / sl_bwd_level (one launch per tree level)",
                                   achieved=bytes_step / (dt / a.steps) / 1e9, peak=8000.0, unit="GB/s",
                                   frac=bytes_step / (dt / a.steps) / 8e12, algorithmic_bytes_per_step=bytes_step,
                                   nnz_L=nnzL, supernodes=int(i8[0]), levels=int(i8[2]), root_order=int(i8[3]), traffic=None,
                                   note="algorithmic bytes = 8 nnz(L) (1 write + 2 reads per solve); whole step time, launches and host "
                                        "round trip of the inertia included")
            # HBM-side traffic of the level kernels of one step (PMC: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes), measured on
            # THIS pattern (banded n = 1e6): scripts/calls/r06_pmc_sparse.sh -> profiles/r06_pmc_sparse/summary.json
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_sparse", "summary.json")))["per_step"]
                if pattern == "chain" and n == 1_000_000:
                    tr = pmc["factor_fetch_bytes"] + pmc["factor_write_bytes"] + a.solves * (pmc["solve_fetch_bytes"] + pmc["solve_write_bytes"])
                    out["roofline"].update(traffic=tr, traffic_over_algorithmic=tr / bytes_step,
                                           traffic_source="profiles/r06_pmc_sparse/summary.json: the level kernels of 1 factorisation + "
                                                          f"{a.solves} solves; the L panels are stored as full nc x f rectangles (zeros above the "
                                                          "diagonal of the pivot block) and the sweeps also read index lists and gather plans")
            except Exception:
                pass
    K.close()
    return out


def mds_other_orders_bench(ctx, a, dims=((4000, 4000), (4096, 4092), (3500, 3000))):
    """The headline step — assemble + factor + `solves` solveCompressed of the MDS condensed KKT — at orders N = nd + neq + 3 that are
    NOT the headline's 8192 (which is even, a multiple of 256 and of 512: the best case of the tile form, of the dataflow's task graph and
    of the solve's block size at once).  The solver object factors and solves such an order at a padded one ("orders the
    fast forms do not take"); round 5 ran N = 8003 at 99 it/s and 8191 at 98 against 165 at 8192."""
    import ctypes as C
    import torch
    from hiop_amd.runtime import dev
    from hiop_amd.kkt import mds_from_problem
    from hiop_amd._lib import lib
    from hiop_amd import problems as pr
    L = lib()
    out = []
    for nd, neq in dims:
        p = pr.mds_ex1_g(a.ns, nd, neq)
        Dx, Dd = pr.barrier_diagonals(p)
        rhs = pr.random_rhs(p)
        kg, dv = mds_from_problem(ctx, p)
        dv["Dx"], dv["Dd"] = dev(Dx), dev(Dd)
        kg.set_values(dv["Jcs_v"], dv["Jds_v"], dv["Hss_v"], dv["Jcd"], dv["Jdd"], dv["Hdd"], dv["Dx"], dv["Dd"])
        rx, ryc, ryd0 = dev(rhs[0]), dev(rhs[1]), dev(rhs[2])
        ryd = ryd0.clone()
        dx, dyc, dyd = torch.zeros_like(rx), torch.zeros_like(ryc), torch.zeros_like(ryd)
        torch.cuda.synchronize()
        P = lambda t: C.c_void_p(t.data_ptr())
        n_neg, z = C.c_int(0), C.c_double(0.0)
        expected_neg = p.neq + p.nineq

        def step():
            rc = L.hiopamd_kkt_mds_build(kg.h, z, z, z, z)
            rc = rc or L.hiopamd_kkt_mds_factorize(kg.h, C.byref(n_neg))
            if rc != 0 or n_neg.value != expected_neg:
                raise RuntimeError(f"status {rc}, inertia {n_neg.value} != {expected_neg}")
            for _ in range(a.solves):
                rc = L.hiopamd_copy_d2d(ctx.h, P(ryd), P(ryd0), ryd.numel() * 8)
                rc = rc or L.hiopamd_kkt_mds_solve_compressed(kg.h, P(rx), P(ryc), P(ryd), P(dx), P(dyc), P(dyd))
                if rc != 0:
                    raise RuntimeError(f"solve_compressed status {rc}")

        for _ in range(a.warmup):
            step()
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        ctx.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append({"N": p.N, "n_dense": p.nxd, "m": p.neq + p.nineq, "value": a.steps / dt, "unit": "KKT iterations/s",
                    "ms_per_step": 1e3 * dt / a.steps})
        kg.close()
    return {"workload": "the headline step at other orders (same n_sparse, other n_dense / m): assemble + factor + "
                        f"{a.solves} solveCompressed; the solver object works at a padded order", "orders": out}


def ipm_end_to_end_bench(ns=4092, nd=4097):
    """The reference's metric measured the reference's way: a REAL interior-point run — hiop_mds_create/solve/destroy_problem of this
    library (include/hiop_amd_interface.h) on the stock MdsEx1 problem at the headline order (ns = 4092 sparse pairs, nd = 4097 dense
    variables, m = ns + 3: N = nd + m = 8192), callbacks on device arrays — and iterations / (time inside the KKT span), the span
    hiopRunKKTStats::tmTotal covers (hiopAlgFilterIPM.cpp:2339-2461).  The driver is the C program of the tests, compiled here with gcc."""
    import re
    import shutil
    import subprocess
    import tempfile
    from pathlib import Path
    from hiop_amd.build import build
    root = Path(__file__).resolve().parent
    if shutil.which("gcc") is None:
        return {"error": "no gcc on this box"}
    lib = build()
    with tempfile.TemporaryDirectory() as td:
        exe = Path(td) / "mds_c_interface"
        cmd = ["gcc", "-std=c11", "-O1", f"-I{root / 'include'}", str(root / "tests" / "c" / "mds_c_interface.c"), "-o", str(exe),
               f"-L{lib.parent}", "-lhiopamd", "-lm", f"-Wl,-rpath,{lib.parent}"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return {"error": "gcc: " + r.stderr[-300:]}
        out = {}
        for mode in ("device", "host"):
            r = subprocess.run([str(exe), mode, str(ns), str(nd)], capture_output=True, text=True, timeout=600)
            g = re.search(r"obj=(\S+) iters=(\d+) status=(-?\d+) nfact=(\d+)", r.stdout)
            t = re.search(r"times: total=(\S+) kkt=(\S+)", r.stdout)
            if r.returncode != 0 or not g or not t:
                out[mode] = {"error": (r.stdout[-200:] + r.stderr[-200:])}
                continue
            iters, total, kkt = int(g.group(2)), float(t.group(1)), float(t.group(2))
            out[mode] = dict(objective=float(g.group(1)), iterations=iters, status=int(g.group(3)), factorizations=int(g.group(4)),
                             solve_seconds=total, kkt_span_seconds=kkt, kkt_iterations_per_s=iters / kkt if kkt > 0 else None,
                             ipm_iterations_per_s=iters / total if total > 0 else None)
    out["workload"] = (f"hiop_mds_solve_problem on stock MdsEx1(ns={ns}, nd={nd}): n={2 * ns + nd}, m={ns + 3}, N={nd + ns + 3}; Newton filter IPM, "
                       "tolerance 1e-8, mu0 0.1; 'device' = callbacks on device arrays (hiopamd_mdsex1_*), 'host' = callbacks on host arrays "
                       "(Jacobian / Hessian blocks cross PCIe every iteration: the PCIe-inclusive rate)")
    return out


def time_on_ctx_stream(ctx, fn, reps=20):
    """average milliseconds of `fn` (a C-ABI call that launches on the context's stream), HIP events on that stream."""
    import torch
    s = ctx.torch_stream
    fn(); fn()
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, LOCAL_RANK -> device) on
    127.0.0.1 and relay rank 0's JSON line.  Fails loudly when the node has fewer than N devices."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if _fake_multi() and have >= 1:
        have = a.gpus
    if have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} HIP device(s) visible on this node")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"bench.py: rank exit codes {rcs}")


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: no HIP device visible (there is no CPU product path)")
    if _fake_multi():
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if _fake_multi() else "nccl", rank=rank, world_size=world)

    from hiop_amd.runtime import Context, dev
    from hiop_amd.kkt import mds_from_problem
    from hiop_amd._lib import lib
    import ctypes as C
    from hiop_amd import problems as pr

    p = pr.mds_ex1_g(a.ns, a.nd, a.neq)
    Dx, Dd = pr.barrier_diagonals(p)
    rhs = pr.random_rhs(p)
    ctx = Context(local_rank)
    kg, dv = mds_from_problem(ctx, p)
    dv["Dx"], dv["Dd"] = dev(Dx), dev(Dd)
    kg.set_values(dv["Jcs_v"], dv["Jds_v"], dv["Hss_v"], dv["Jcd"], dv["Jdd"], dv["Hdd"], dv["Dx"], dv["Dd"])
    rx, ryc, ryd0 = dev(rhs[0]), dev(rhs[1]), dev(rhs[2])
    ryd = ryd0.clone()
    dx, dyc, dyd = torch.zeros_like(rx), torch.zeros_like(ryc), torch.zeros_like(ryd)
    torch.cuda.synchronize()
    L = lib()
    expected_neg = p.neq + p.nineq

    # The step calls the C ABI the way a C++ caller (the HiOp-side adapter) does: raw device addresses resolved ONCE — every operand is a
    # persistent buffer written before the timed region and synchronised —, one ctypes call per entry point.  (The test wrappers of
    # hiop_amd/kkt.py re-resolve every tensor per call and order the context's stream behind torch's current stream each time: six event
    # records + waits per solveCompressed, ~0.15 ms of harness per step that is not part of the path being measured.)
    P = lambda t: C.c_void_p(t.data_ptr())
    p_rx, p_ryc, p_ryd, p_ryd0, p_dx, p_dyc, p_dyd = P(rx), P(ryc), P(ryd), P(ryd0), P(dx), P(dyc), P(dyd)
    nbytes_ryd = ryd.numel() * 8
    n_neg = C.c_int(0)
    z = C.c_double(0.0)

    def step():
        rc = L.hiopamd_kkt_mds_build(kg.h, z, z, z, z)
        rc = rc or L.hiopamd_kkt_mds_factorize(kg.h, C.byref(n_neg))
        if rc != 0 or n_neg.value != expected_neg:
            raise RuntimeError(f"status {rc}, inertia {n_neg.value} != {expected_neg}")
        for _ in range(a.solves):
            rc = L.hiopamd_copy_d2d(ctx.h, p_ryd, p_ryd0, nbytes_ryd)
            rc = rc or L.hiopamd_kkt_mds_solve_compressed(kg.h, p_rx, p_ryc, p_ryd, p_dx, p_dyc, p_dyd)
            if rc != 0:
                raise RuntimeError(f"solve_compressed status {rc}")

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if _fake_multi() else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline pass: the same K steps with HIP events around every launch of the dominant kernel
    ls = C.c_void_p(L.hiopamd_kkt_mds_linsolver(kg.h))
    L.hiopamd_linsolver_profile(ls, 1)
    for _ in range(a.steps):
        step()
    ctx.sync()
    ums, ufl, ul = C.c_double(0), C.c_double(0), C.c_int64(0)
    L.hiopamd_linsolver_profile_read(ls, C.byref(ums), C.byref(ufl), C.byref(ul))
    L.hiopamd_linsolver_profile(ls, 0)
    launches = max(int(ul.value), 1)
    flops_per_launch = ufl.value / launches
    ms_per_launch = ums.value / launches
    achieved = (flops_per_launch / (ms_per_launch * 1e-3)) / 1e12 if ms_per_launch > 0 else 0.0
    # HBM-side traffic of that kernel per launch: PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate rocprofv3 --pmc
    # passes with --kernel-trace only) of the SAME task graph run as one dispatch (HIOPAMD_DF_ONE=1: rocprofv3 serialises
    # dispatches, and the production form is a pair of kernels that wait for each other) — scripts/calls/r06_pmc.sh, committed
    # summary profiles/r06_pmc/summary.json (older rounds' summaries as fall-back).  Algorithmic bytes of the same launch: every trailing tile read + written once
    # per super-panel, the two operand row panels read once, the row-panel substitution (read A, write V and U).
    traffic, traffic_src, alg_bytes = None, "n/a (no committed PMC summary found)", None
    for pmc_dir in ("r06_pmc", "r05_pmc", "r04_pmc", "r03_pmc"):
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_dir, "summary.json")))
            if p.N == 8192:
                traffic = pmc["ldlt_df_one_kernel"]["hbm_bytes_per_launch"]
                traffic_src = (f"profiles/{pmc_dir}/summary.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch of "
                               "ldlt_df_one_kernel (the dataflow LDL^T's chain + wide roles as ONE dispatch, N = 8192); L2 memory-side "
                               "requests, Infinity-Cache hits included")
            break
        except Exception:
            continue
    nsp_ = (p.N + 255) // 256
    alg_bytes = 0.0
    for j_ in range(nsp_ - 1):
        m_ = p.N - 256 * (j_ + 1)
        alg_bytes += 8.0 * 2.0 * (m_ * m_ / 2.0 + 64.0 * m_)      # C tiles of the upper triangle: read + write
        alg_bytes += 8.0 * 2.0 * 256.0 * m_                       # V and U row panels, once
        alg_bytes += 8.0 * 3.0 * 256.0 * max(m_ - 256, 0)         # substitution of the row panel's tail: read A, write V and U
    # the dominant kernel is the persistent wide kernel of the dataflow factorisation: `achieved` = the algorithmic flops of its
    # trailing-update tiles (2 K per updated element of the upper triangle) / its WHOLE duration, which also contains the
    # row-panel substitution tasks and every wait for the chain kernel — a lower bound on the tile rate, by construction
    kname = "ldlt_wide_kernel<2,false> (four waves, ONE workgroup per CU)"
    roofline = dict(bound="mfma", kernel=kname + " — dataflow LDL^T: row-panel substitution tasks + 128x128x256 / x512 trailing-update tiles, v_mfma_f64_16x16x4_f64",
                    achieved=achieved, peak=PEAK_FP64_MFMA_TFLOPS, unit="TFLOP/s", frac=achieved / PEAK_FP64_MFMA_TFLOPS,
                    traffic=traffic, traffic_unit="HBM bytes per launch (PMC)", traffic_source=traffic_src,
                    launches_per_step=launches / a.steps, avg_launch_ms=ms_per_launch,
                    algorithmic_flops_per_launch=flops_per_launch, algorithmic_bytes_per_launch=alg_bytes,
                    traffic_over_algorithmic=(traffic / alg_bytes) if traffic and alg_bytes else None,
                    update_ms_per_step=ums.value / a.steps)

    # ---- run-stats spans (the reference's hiopRunStatsKKT sub-spans): a third pass of the same K steps with the context's
    # span timers on (HIP events, no host sync inside a span)
    L.hiopamd_ctx_spans_enable(ctx.h, 1)
    t0s = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.sync()
    dts = time.perf_counter() - t0s
    sp_ms, sp_cnt = (C.c_double * 8)(), (C.c_int64 * 8)()
    L.hiopamd_ctx_spans_read(ctx.h, sp_ms, sp_cnt)
    L.hiopamd_ctx_spans_enable(ctx.h, 0)
    L.hiopamd_span_name.restype = C.c_char_p
    spans = {L.hiopamd_span_name(i).decode(): {"ms_per_step": sp_ms[i] / a.steps, "calls_per_step": sp_cnt[i] / a.steps}
             for i in range(8)}
    ff, ft = C.c_double(0), C.c_double(0)
    L.hiopamd_linsolver_flops(ls, C.byref(ff), C.byref(ft))
    fact_ms = spans["linsolv.tmFactTime"]["ms_per_step"]
    spans["linsolv.flopsFact_per_step"] = float(p.N) ** 3 / 3.0
    spans["linsolv.fact_tflops"] = (float(p.N) ** 3 / 3.0) / (fact_ms * 1e-3) / 1e12 if fact_ms > 0 else None
    spans["ms_per_step_with_span_events"] = 1e3 * dts / a.steps

    # ---- parity spot check of the last solve against the oracle's residual definition (not timed)
    from oracle import hiop_oracle as ho
    ko = ho.KKTLinSysCompressedMDSXYcYd(p.nxs, p.nxd, p.neq, p.nineq, (p.Jcs_i, p.Jcs_j), (p.Jds_i, p.Jds_j),
                                        (p.Hss_i, p.Hss_j))
    ko.set_values(p.Jcs_v, p.Jds_v, p.Hss_v, p.Jcd, p.Jdd, p.Hdd, Dx, Dd)
    res = ho.kkt_mds_full_residual(ko, (0.0, 0.0, 0.0, 0.0), rhs[0], rhs[1], rhs[2], dx.cpu().numpy(), dyc.cpu().numpy(),
                                   dyd.cpu().numpy())

    dense = None
    dense_c2 = None
    if not a.no_dense:
        dense = dense_lowrank_bench(ctx, world, rank, a, dist, rooflines=(world == 1))
        if world > 1:
            # the same shard on every rank WITHOUT the collective (a second context, no hook): what one rank does alone in the
            # same time window -> the sharded run's overhead is (ms_per_step / single_rank_ms_per_step - 1)
            ctx1 = Context(local_rank)
            alone = dense_lowrank_bench(ctx1, world, rank, a, dist, hooked=False)
            dense["single_rank_ms_per_step"] = alone["ms_per_step"]
            dense["n_ranks"] = world
            ctx1.close()
        if world == 1:
            # BASELINE configs[1]: NlpDenseCons_ex2 shape on ONE GPU, n = 1e6, m = 100 (the single-GPU dense low-rank case)
            import copy
            a2 = copy.copy(a)
            a2.dense_nlocal, a2.dense_k = 1_000_000, 100
            dense_c2 = dense_lowrank_bench(ctx, world, rank, a2, dist, rooflines=True)

    sparse_c5 = None
    sparse_banded = None
    if world == 1 and not a.no_dense:
        try:
            sparse_c5 = sparse_condensed_bench(ctx, a)
        except Exception as e:      # an auxiliary entry must not take the headline line down
            sparse_c5 = {"error": repr(e)}
        try:
            sparse_banded = sparse_condensed_bench(ctx, a, pattern="chain")
        except Exception as e:
            sparse_banded = {"error": repr(e)}

    other_orders = None
    if world == 1 and not a.no_dense and p.N == 8192:
        try:
            other_orders = mds_other_orders_bench(ctx, a)
        except Exception as e:
            other_orders = {"error": repr(e)}

    ipm_e2e = None
    if world == 1 and not a.no_dense:
        try:
            ctx.sync()
            ipm_e2e = ipm_end_to_end_bench()
        except Exception as e:
            ipm_e2e = {"error": repr(e)}

    out = None
    if rank == 0:
        value = world * a.steps / dt
        out = {
            "metric": "KKT assemble+factor+solve iters/sec (fp64)",
            "value": value, "unit": "KKT iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"NlpMDS condensed KKT (MdsEx1-g): n_sparse={p.nxs} n_dense={p.nxd} m={p.neq + p.nineq} "
                                   f"N={p.N}; step = 1 assemble + 1 LDL^T factor(+inertia) + {a.solves} solveCompressed",
                       "parallelism": "single GPU" if world == 1 else f"replicas only x{world} (MDS path does not shard)"},
            "roofline": roofline,
            "kkt_spans": spans,
            "check": {"kkt_backward_error": max(res), "inertia_neg": expected_neg},
        }
        if dense is not None:
            out["dense_sharded"] = dense
        if dense_c2 is not None:
            out["dense_n1e6_m100"] = dense_c2
        if sparse_c5 is not None:
            out["sparse_condensed_n1e6"] = sparse_c5
        if sparse_banded is not None:
            out["sparse_condensed_banded_n1e6"] = sparse_banded
        if other_orders is not None:
            out["mds_other_orders"] = other_orders
        if ipm_e2e is not None:
            out["ipm_end_to_end_N8192"] = ipm_e2e
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, Dx, Dd, rhs, a.solves, a.cpu_steps)
        print(json.dumps(out), flush=True)
    kg.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
