"""The column-sharded (memory-distributed) dense-constraint path of the PRODUCT LIBRARY on two ranks.

Two processes share cuda:0; each owns a hiopamd context with `comm_size = 2` and an all-reduce hook that stages the
small device buffers through the host and gloo (tests/two_rank_worker.py).  The sequence
  hiopamd_hess_lowrank_update x 9 -> hiopamd_kkt_lowrank_update_diag -> hiopamd_kkt_lowrank_solve_compressed ->
  hiopamd_kkt_xycyd_update -> hiopamd_kkt_xycyd_compute_directions_w_IR -> hiopamd_iterate_adjust_small_slacks
runs on both ranks and is compared with the same sequence on ONE rank (this process, no hook):
  * replicated outputs (dyc, dyd, N, the d/yc/yd/sd*/v* parts of the direction, every reduced scalar) must be
    BIT-IDENTICAL across the two ranks (they come out of the same all-reduced buffers);
  * against the single-rank HIP result: 1e-10 relative (the summation order over the columns differs).
reference semantics: rank-0-only beta terms hiopMatrixDenseRowMajor.cpp:466-487, the fused all-reduce of
hiopHessianLowRank.cpp:568-591, hiopVectorPar's all-reduced min_w_pattern / numOfElemsLessThan (:833-836, :1231-1236)."""
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import two_rank_worker as tw

pytestmark = pytest.mark.gpu

REPL_DIR_PARTS = ("d", "yc", "yd", "sdl", "sdu", "vl", "vu")
X_DIR_PARTS = ("x", "sxl", "sxu", "zl", "zu")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max(initial=0.0) / max(np.abs(b).max(initial=0.0), 1e-300))


@pytest.mark.parametrize("n,me,mi", [(4001, 3, 4), (1000, 1, 2), (2003, 100, 100)])   # last: k = 200 (the shape of the sharded bench case)
def test_two_ranks_through_the_library_equal_one_rank(ctx, n, me, mi):
    seed = 11
    prob = tw.make_problem(n, me, mi, seed)
    cols = tw.col_partition(n, 2)
    # one tiny slack, owned by rank 1 only
    tiny_at = next(i for i in range(cols[1] + 5, n) if prob["ixl"][i] == 1.0)
    one = tw.run_partition(ctx, prob, slice(0, n), tiny_slack_at=tiny_at)

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=tw.worker, args=(r, 2, port, n, me, mi, seed, tiny_at, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r0, r1 = got
    try:   # diagnostic summary (kept under gpurun_out/ on the GPU box)
        import json, os
        os.makedirs("gpurun_out", exist_ok=True)
        dbg = {"dyc": [r0["dyc"].tolist(), r1["dyc"].tolist(), one["dyc"].tolist()],
               "N_r0_vs_r1": _rel(r0["N"], r1["N"]), "N_r0_vs_one": _rel(r0["N"], one["N"]), "N_r1_vs_one": _rel(r1["N"], one["N"]),
               "dx_norms": [float(np.linalg.norm(r0["dx"])), float(np.linalg.norm(r1["dx"])), float(np.linalg.norm(one["dx"]))],
               "log0": r0["allreduce_calls"]["log"][:80], "log1": r1["allreduce_calls"]["log"][:80]}
        json.dump(dbg, open(f"gpurun_out/two_rank_debug_{n}.json", "w"), indent=1)
    except Exception:
        pass
    # the hook was actually used, the same number of times on both ranks (a diverging count would hang a real collective)
    l0, l1 = r0["allreduce_calls"]["log"], r1["allreduce_calls"]["log"]
    # same sequence of (count, op) on both ranks and the same reduced payload: a diverging sequence would hang or corrupt a
    # real collective
    assert [(c, o) for c, o, _, _ in l0] == [(c, o) for c, o, _, _ in l1]
    assert [a for _, _, _, a in l0] == [a for _, _, _, a in l1]
    assert r0["allreduce_calls"]["n"] > 20 and r0["allreduce_calls"]["n"] == r1["allreduce_calls"]["n"]
    # ---- replicated results: bit-identical across ranks
    assert r0["stored"] == r1["stored"] == one["stored"]
    assert r0["sigma"] == r1["sigma"]
    for key in ("dyc", "dyd", "N"):
        assert np.array_equal(r0[key], r1[key]), key
    for part in REPL_DIR_PARTS:
        assert np.array_equal(r0["dir"][part], r1["dir"][part]), part
    assert r0["ir_info"] == r1["ir_info"] and r0["ir_ok"] and r1["ir_ok"]
    assert r0["test_direction"] == r1["test_direction"]
    assert r0["log_barrier"] == r1["log_barrier"] and r0["ftb"] == r1["ftb"]
    assert r0["num_adjusted"] == r1["num_adjusted"] == one["num_adjusted"] == 1
    # ---- against the single-rank run of the same library
    assert r0["ok"] and r1["ok"] and one["ok"]
    assert r0["sigma"] == pytest.approx(one["sigma"], rel=1e-12)
    assert _rel(r0["N"], one["N"]) < 1e-10
    dx2 = np.concatenate([r0["dx"], r1["dx"]])
    assert _rel(dx2, one["dx"]) < 1e-10 and _rel(r0["dyc"], one["dyc"]) < 1e-10 and _rel(r0["dyd"], one["dyd"]) < 1e-10
    assert one["ir_ok"] and one["ir_info"]["converged"] and r0["ir_info"]["converged"]
    for part in X_DIR_PARTS:
        both = np.concatenate([r0["dir"][part], r1["dir"][part]])
        assert _rel(both, one["dir"][part]) < 1e-8, part      # through the BiCGStab refinement (tol 1e-6 on the residual)
    for part in REPL_DIR_PARTS:
        assert _rel(r0["dir"][part], one["dir"][part]) < 1e-8, part
    assert r0["test_direction"][0] == one["test_direction"][0]
    assert r0["test_direction"][1] == pytest.approx(one["test_direction"][1], rel=1e-7)
    assert r0["log_barrier"] == pytest.approx(one["log_barrier"], rel=1e-12)
    assert r0["ftb"] == pytest.approx(one["ftb"], rel=1e-12)
    sxl2 = np.concatenate([r0["sxl_adjusted"], r1["sxl_adjusted"]])
    np.testing.assert_allclose(sxl2, one["sxl_adjusted"], rtol=1e-14, atol=0)
    assert sxl2[tiny_at] > 1e-20     # the tiny slack was pushed back inside


def test_cached_N_is_bit_identical(ctx):
    """hiopamd_kkt_lowrank caches N = J (H+Dx)^-1 J^T + Dd^-1 and its factor between the solveCompressed calls of one
    outer iteration; every invalidating call must rebuild it, and the cached solves must equal the rebuilt ones bit for
    bit (same kernels, same order)."""
    from hiop_amd.kkt import HessianLowRank, KKTLinSysLowRank
    n, me, mi = 3000, 3, 5
    prob = tw.make_problem(n, me, mi, 5)

    def D(a):
        return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float64).cuda()

    def run(cache):
        J = D(np.vstack([prob["Jc"], prob["Jd"]]))
        Jc, Jd = J[:me], J[me:]
        H = HessianLowRank(ctx, n, me, mi, l_max=6, sigma0=1.0, sigma_update_strategy="sty")
        K = KKTLinSysLowRank(ctx, H)
        K.set_cache(cache)
        outs = []
        for it in range(5):
            x = prob["xs"][it]
            # device temporaries must outlive the asynchronous calls that read them (see tests/two_rank_worker.py)
            args = [D(x), D(prob["q"] * x), Jc, Jd, D(prob["ycs"][it]), D(prob["yds"][it])]
            diag = [D(prob["Dx"] * (1.0 + 0.1 * it)), D(prob["Dd"])]
            torch.cuda.synchronize()
            H.update(*args)
            K.update_diag(diag[0], diag[1], Jc, Jd)
            ctx.sync()
            for s in range(3):
                rx = D(prob["rx"] * (s + 1.0))
                dx, dyc, dyd = D(np.zeros(n)), D(np.zeros(me)), D(np.zeros(mi))
                rr = [D(prob["ryc"]), D(prob["ryd"] + s)]
                torch.cuda.synchronize()
                assert K.solve_compressed(rx, rr[0], rr[1], dx, dyc, dyd)
                ctx.sync()
                outs.append((dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy(), K.N().cpu().numpy()))
            if it == 2:   # Jacobian values changed in place + set_jacobians: the cache must not survive
                J.mul_(1.01)
                torch.cuda.synchronize()
                K.set_jacobians(Jc, Jd)
                dx, dyc, dyd = D(np.zeros(n)), D(np.zeros(me)), D(np.zeros(mi))
                rr = [D(prob["rx"]), D(prob["ryc"]), D(prob["ryd"])]
                torch.cuda.synchronize()
                assert K.solve_compressed(rr[0], rr[1], rr[2], dx, dyc, dyd)
                ctx.sync()
                outs.append((dx.cpu().numpy(), dyc.cpu().numpy(), dyd.cpu().numpy(), K.N().cpu().numpy()))
        K.close(); H.close()
        return outs

    a, b = run(True), run(False)
    assert len(a) == len(b)
    for i, (u, v) in enumerate(zip(a, b)):
        for x, y in zip(u, v):
            assert np.array_equal(x, y), i


def test_bench_two_rank_control_flow_on_one_gpu():
    """`bench.py --gpus 2` end to end where only one GPU exists: HIOPAMD_BENCH_FAKE_MULTI=1 puts both ranks on device 0,
    runs torch.distributed over gloo and routes the library's collectives through the generic all-reduce hook.  Checks the
    contract's shape at N > 1 (one JSON line from rank 0, whole-job value, replicas + the sharded dense case); the timings
    themselves mean nothing on a shared device."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIOPAMD_BENCH_FAKE_MULTI="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dense-nlocal", "100000"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]    # gloo prints its own banner lines to stdout
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    ds = d["dense_sharded"]
    assert ds["n_ranks"] == 2 and ds["ms_per_step"] > 0 and "2 ranks" in ds["collective"]
    # the N > 1 line explains itself: collectives per step counted at the hook (2 in the secant update, 1 for the l x l blocks, 1 for N,
    # 1 per solveCompressed: 4 + solves; the first steps after a memory shift may differ by the update that was skipped), their device
    # time between HIP events, the communicator's size (0 here: the rehearsal's hook is host-staged gloo, not RCCL), one rank alone
    assert 4 <= ds["collectives_per_step"] <= 4 + 3 + 1 and ds["collective_ms_per_step"] > 0 and ds["rccl_ranks"] == 0
    assert ds["single_rank_ms_per_step"] > 0
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only
