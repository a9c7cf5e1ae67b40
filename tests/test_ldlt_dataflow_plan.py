"""CPU check of the static schedule of the dataflow LDL^T (csrc/ldlt_dataflow.hpp) — no GPU needed.

`hiopamd_ldlt_dataflow_plan` (host-only entry point of libhiopamd.so) returns the task tables the two persistent kernels
execute: the chain kernel's per-role lists of F / T / U tile tasks and the wide kernel's per-super-panel TR / UP queues
(`hiopamd_ldlt_dataflow_queues`).  This test replays them with numpy under the SAME protocol the kernels use (tile version
counters, cdone / hdone / updone, tr[j][J], ver[I][J]; a wide worker TAKES a task only when the take-conditions of
ldlt_wide_kernel hold — TR(j): C_j factored, first two tile rows of UP(j-1) taken, UP(j-2) complete; UP(j): every TR(j)
taken — and every task may only start when its wait conditions hold) with the 16 chain roles and W wide workers
advancing in an arbitrary (seeded, adversarial) interleaving, and checks
  * liveness: every task completes — no role or worker waits forever, for several worker counts (1 .. 64) and orders;
  * soundness: the factor that comes out equals the unblocked U^T D U recurrence of the same matrix to 1e-10 — i.e. every
    task saw exactly the operands the algorithm needs (a missing wait shows up as a wrong factor under some interleaving);
  * the ragged case (N not a multiple of 256) hands over to the stepwise path with a consistent state.
reference semantics: the no-pivot factorisation of hiopLinSolverSymDenseMagmaNopiv (src/LinAlg/hiopLinSolverSymDenseMagma.cpp:324-480)."""
import ctypes as C
import random

import numpy as np
import pytest

F, T, U, SP, RC, CS, END = 1, 2, 3, 4, 5, 6, 0   # SP / RC / CS: fused spine step S(p), its companion R(p), column step C(p, c)
NVB = 4                  # row-panel workspaces used round robin: the MINIMUM the library allocates (DF_NVB_MIN of csrc/ldlt_dataflow.hpp;
                         # it takes one per super-panel when that fits 2 GB — fewer buffers is the stricter protocol, which is what is replayed)
TR, UP, UPH, UP2 = 1, 2, 3, 4   # UPH: head tile of the update (follows the super-panel block row by block row); UP2: K = 512,
                                # super-panels j - 1 and j applied together to a tile behind super-panel j + 1 (a task of queue j)


def get_plan(n):
    from hiop_amd._lib import lib
    L = lib()
    dims = (C.c_int * 8)()
    assert L.hiopamd_ldlt_dataflow_plan(n, dims, None, None, 0) == 0
    nsp, nt, nchain, last_has_next, nwide, roles, maxt, nw = list(dims)
    ct = (C.c_int * (2 * roles * maxt * 4))()
    wt = (C.c_int * (4 * max(nw, 1)))()
    assert L.hiopamd_ldlt_dataflow_plan(n, dims, ct, wt, nw) == 0
    ct = np.array(ct, dtype=np.int64).reshape(2, roles, maxt, 4)
    wt = np.array(wt, dtype=np.int64).reshape(-1, 4)[:nw]
    q = (C.c_int * (5 * max(nwide, 1)))()
    assert L.hiopamd_ldlt_dataflow_queues(n, q, nwide) == 0
    q = np.array(q, dtype=np.int64).reshape(-1, 5)[:nwide]
    fq = (C.c_int * (4 * max(nwide, 1)))()
    assert L.hiopamd_ldlt_dataflow_far_queues(n, fq, nwide) == 0
    fq = np.array(fq, dtype=np.int64).reshape(-1, 4)[:nwide]
    return dict(far=fq, nsp=nsp, nt=nt, nchain=nchain, last_has_next=last_has_next, nwide=nwide, roles=roles, ctasks=ct, wtasks=wt,
                queues=q)


def ldl_nopiv(A):
    """unblocked A = U^T D U (upper, unit diagonal): the recurrence the tests of the GPU factor compare against."""
    n = A.shape[0]
    S = np.triu(A).copy()
    for k in range(n):
        d = S[k, k]
        u = S[k, k + 1:] / d
        S[k + 1:, k + 1:] -= np.triu(np.outer(S[k, k + 1:], u))
        S[k, k + 1:] = u
    return S


class Sim:
    """numpy model of the two kernels.  The matrix is kept as one N x N array (upper part significant); the compact
    diagonal blocks are modelled by a per-block 'compact' copy exactly where the kernels use one."""

    def __init__(self, A, plan):
        self.N = A.shape[0]
        self.P = plan
        self.A = np.triu(A).copy()                 # trailing matrix + H tiles + U of the tails
        nsp = plan["nsp"]
        self.Cd = {0: self.A[:256, :256].copy()}   # compact copies: block 0 packed up front, others created by Nn updates
        # the V workspaces are modelled as the kernels use them: keyed by j % NVB, so that a super-panel that writes its V
        # before update j - NVB has read the buffer completely corrupts that update (and the factor check below sees it)
        self.V = {}                                # (j % NVB, p, c) -> un-scaled 64 x 64 tile
        self.dinv = np.zeros(self.N)
        self.cv = np.zeros((nsp + 1, 4, 4), dtype=int)
        self.cv[0] = 4
        self.hv = np.zeros((nsp + 1, 4, 4), dtype=int)
        self.cdone = np.zeros(nsp + 1, dtype=int)
        self.hdone = np.zeros(nsp + 1, dtype=int)
        self.updone = np.zeros(nsp + 1, dtype=int)
        nt = plan["nt"]
        self.tr = np.zeros((nsp + 1, nt), dtype=int)
        self.ver = np.zeros((nt, nt), dtype=int)
        self.upcnt = np.zeros(nsp + 1, dtype=int)
        for t in plan["wtasks"]:
            if t[0] in (UP, UPH, UP2):
                self.upcnt[t[1]] += 1
            if t[0] == UP2:                     # it reads the row panel of super-panel j - 1 as well
                self.upcnt[t[1] - 1] += 1
        self.Vtail = {}

    # ---- tile access (window of super-panel j)
    def tile_ref(self, j, r, c):
        if r < 4 and c < 4:
            return self.Cd[j], 64 * r, 64 * c
        if r < 4:
            return self.A, 256 * j + 64 * r, 256 * (j + 1) + 64 * (c - 4)
        return self.Cd.setdefault(j + 1, np.zeros((256, 256))), 64 * (r - 4), 64 * (c - 4)

    def get(self, j, r, c):
        M, a, b = self.tile_ref(j, r, c)
        return M[a:a + 64, b:b + 64]

    def ver_of(self, j, r, c):
        if r < 4 and c < 4:
            return self.cv[j], (r, c), 4
        if r < 4:
            return self.hv[j], (r, c - 4), 0
        return self.cv[j + 1], (r - 4, c - 4), 0

    def wide_ver(self, j, r, c):
        return self.ver[2 * j + r // 2, 2 * j + c // 2]

    # ---- chain tasks: ready? / run
    def fused_parts(self, tk):
        ty, p, a, b = tk
        if ty == SP:    # F(p) -> T(p, p+1) -> U(p; p+1, p+1)
            return [(F, p, p, p)] + ([(T, p, p, p + 1), (U, p, p + 1, p + 1)] if a else [])
        if ty == RC:    # (a = 1: the update of the diagonal tile (p+2, p+2) is a separate task of role 2)
            return [(T, p, p, p + 2), (U, p, p + 1, p + 2)] + ([(U, p, p + 2, p + 2)] if a == 0 else [])
        c = a                                                                    # CS: T(p, c) + U(p; a', c), a' = p+1 .. amax
        amax = b if b > 0 else min(3, c)
        return [(T, p, p, c)] + [(U, p, a2, c) for a2 in range(p + 1, amax + 1)]

    def chain_ready(self, j, tk):
        ty, p, a, b = tk
        if ty in (SP, RC, CS):
            # every wait of the fused task on OTHER roles' work, required up front (stricter than the kernel, which waits for
            # the later parts' conditions only when it reaches them): conditions on the task's own earlier parts are dropped
            parts = self.fused_parts(tk)
            own = set()
            for q in parts:
                if not self.part_ready(j, q, own):
                    return False
                own.add((q[0], q[1], q[2], q[3]))
            return True
        return self.part_ready(j, tk, set())

    def part_ready(self, j, tk, own):
        ty, p, a, b = tk
        done_f = (F, p, p, p) in own

        def solved(c):      # T(p, c) part of the same fused task already counted?
            return (T, p, p, c) in own
        if ty == F:
            arr, ix, base = self.ver_of(j, p, p)
            return arr[ix] >= base + p
        if ty == T:
            c = b
            arr, ix, base = self.ver_of(j, p, c)
            app, ipp, bpp = self.ver_of(j, p, p)
            ok = (done_f or app[ipp] >= bpp + p + 1) and arr[ix] >= base + p
            if c >= 4 and p == 0:
                ok = ok and self.wide_ver(j, p, c) >= j
            if j >= NVB:
                ok = ok and self.updone[j - NVB] >= self.upcnt[j - NVB]
            return ok
        arr, ix, base = self.ver_of(j, a, b)
        aa, ia, ba = self.ver_of(j, p, a)
        ab, ib, bb = self.ver_of(j, p, b)
        ok = (solved(a) or aa[ia] >= ba + p + 1) and (solved(b) or ab[ib] >= bb + p + 1) and arr[ix] >= base + p
        if b >= 4 and p == 0:
            ok = ok and self.wide_ver(j, a, b) >= j
        return ok

    def chain_run(self, j, tk):
        ty, p, a, b = tk
        if ty in (SP, RC, CS):
            for q in self.fused_parts(tk):
                self.chain_run(j, q)
            return
        k0 = 256 * j + 64 * p
        if ty == F:
            t = self.get(j, p, p)
            S = ldl_nopiv(t)
            t[:] = np.triu(S)
            self.dinv[k0:k0 + 64] = 1.0 / np.diag(S)
            arr, ix, _ = self.ver_of(j, p, p)
            arr[ix] += 1
            self.cdone[j] += 1
        elif ty == T:
            c = b
            Upp = self.get(j, p, p)
            Lpp = np.triu(Upp, 1).T + np.eye(64)             # unit lower
            x = self.get(j, p, c)
            v = np.linalg.solve(Lpp, x)
            self.V[(j % NVB, p, c)] = v.copy()
            x[:] = v * self.dinv[k0:k0 + 64][:, None]
            arr, ix, _ = self.ver_of(j, p, c)
            arr[ix] += 1
            if c < 4:
                self.cdone[j] += 1
            else:
                self.hdone[j] += 1
        else:
            va = self.V[(j % NVB, p, a)]
            ub = self.get(j, p, b)
            if a >= 4 and p == 0:      # first update of a next-diagonal-block tile: read from the matrix, write the compact copy
                r0 = 256 * (j + 1) + 64 * (a - 4)
                c0 = 256 * (j + 1) + 64 * (b - 4)
                src = self.A[r0:r0 + 64, c0:c0 + 64]
            else:
                src = self.get(j, a, b)
            res = src - va.T @ ub
            self.get(j, a, b)[:] = res
            arr, ix, _ = self.ver_of(j, a, b)
            arr[ix] += 1

    # ---- wide tasks
    def groups(self, B):
        rem = self.N - 128 * B
        return 8 if rem >= 128 else (rem + 15) // 16

    def wide_ready(self, tk):
        ty, j, x, y = tk
        if ty == TR:
            J = x // 128
            ok = self.cdone[j] >= 10 and self.ver[2 * j, J] >= j and self.ver[2 * j + 1, J] >= j
            if j >= NVB:
                ok = ok and self.updone[j - NVB] >= self.upcnt[j - NVB]
            return ok
        I, J = x, y
        if ty == UP2:
            assert j >= 1 and I >= 2 * j + 4 and J >= I
            return (self.ver[I, J] >= j - 1 and self.tr[j - 1, I] >= self.groups(I) and self.tr[j - 1, J] >= self.groups(J)
                    and self.tr[j, I] >= self.groups(I) and self.tr[j, J] >= self.groups(J))
        ok = self.ver[I, J] >= j
        if ty == UPH and 128 * (I + 1) <= self.N and 128 * (J + 1) <= self.N:
            # the kernel's gates, all four block rows at once (the model runs the tile atomically): the chain's tile solves
            # T(P, c) of the two H columns under tile row I — NOT all sixteen — and the substitution tasks of column block J
            assert 2 * j + 2 <= I < 2 * j + 4 and 2 * j + 4 <= J < 2 * j + 6
            cI = 2 * (I - 2 * j - 2)
            for P in range(4):
                ok = ok and self.hv[j][P, cI] >= P + 1 and self.hv[j][P, cI + 1] >= P + 1
            return ok and self.tr[j, J] >= self.groups(J)
        ok = ok and (self.hdone[j] >= 16 if I < 2 * j + 4 else self.tr[j, I] >= self.groups(I))
        return ok and self.tr[j, J] >= self.groups(J)

    def wide_run(self, tk):
        ty, j, x, y = tk
        N = self.N
        K0 = 256 * j
        if ty == TR:
            c0, c1 = x, min(N, x + y)     # (y = columns per substitution task)
            Ujj = self.Cd[j]
            L = np.triu(Ujj, 1).T + np.eye(256)
            blk = self.A[K0:K0 + 256, c0:c1]
            v = np.linalg.solve(L, blk)
            self.Vtail.setdefault(j % NVB, np.zeros((256, N)))[:, c0:c1] = v
            blk[:] = v * self.dinv[K0:K0 + 256][:, None]
            self.tr[j, x // 128] += (c1 - c0 + 15) // 16      # counted in 16-column groups
            return
        I, J = x, y
        r0, r1 = 128 * I, min(N, 128 * I + 128)
        c0, c1 = 128 * J, min(N, 128 * J + 128)
        if ty == UP2:
            blk = self.A[r0:r1, c0:c1]
            for jj in (j - 1, j):                   # the same order of accumulation as the two separate tasks
                upd = self.Vtail[jj % NVB][:, r0:r1].T @ self.A[256 * jj:256 * jj + 256, c0:c1]
                blk -= np.triu(upd) if I == J else upd
                self.updone[jj] += 1
            self.ver[I, J] = j + 1
            return
        s = 256 * (j + 1)
        if I < 2 * j + 4:      # rows in the head: V from the chain's T tasks on H tiles
            Vr = np.zeros((256, r1 - r0))
            for p in range(4):
                for q in range(4):
                    cc0 = s + 64 * q
                    lo, hi = max(cc0, r0), min(cc0 + 64, r1)
                    if lo < hi:
                        Vr[64 * p:64 * p + 64, lo - r0:hi - r0] = self.V[(j % NVB, p, 4 + q)][:, lo - cc0:hi - cc0]
        else:
            Vr = self.Vtail[j % NVB][:, r0:r1]
        Uc = self.A[K0:K0 + 256, c0:c1]
        upd = Vr.T @ Uc
        blk = self.A[r0:r1, c0:c1]
        if I == J:
            blk -= np.triu(upd)
        else:
            blk -= upd
        self.ver[I, J] = j + 1
        self.updone[j] += 1


def replay(A, plan, n_workers, seed, retire_at=None, ahead=False):
    """retire_at: the super-panel at which every second worker (the \"second workgroup of its CU\") leaves the wide kernel
    (HIOPAMD_DF_RETIRE in csrc/ldlt.hip); None: nobody leaves early.
    ahead: the eight-wave wide kernel (csrc/ldlt_wide8_body.inc) selects its NEXT task during the last stages of an ordinary update
    tile — a worker whose update task has started (its inputs are there) may take one more task, never an early substitution task,
    and holds it until the current one is done"""
    rnd = random.Random(seed)
    sim = Sim(A, plan)
    roles = plan["roles"]
    nchain = plan["nchain"]
    # chain role cursors: (j, index in list)
    cur = [[0, 0] for _ in range(roles)]

    def role_task(r):
        while cur[r][0] < nchain:
            j, it = cur[r]
            has_next = (j + 1 < nchain) or bool(plan["last_has_next"])
            lst = plan["ctasks"][0 if has_next else 1][r]
            if it < len(lst) and lst[it][0] != END:
                return j, tuple(int(v) for v in lst[it])
            cur[r] = [j + 1, 0]
        return None

    wt = [tuple(int(v) for v in t) for t in plan["wtasks"]]
    Q = plan["queues"]          # per super-panel: first TR, #TR, first NEAR, #NEAR, #NEAR of the first two tile rows
    FQ = plan["far"]            # per super-panel: first FAR, #FAR, feeding FAR queue (-1: none), tasks of it that must be taken
    nw = plan["nwide"]
    trq = [0] * (nw + 1)        # tasks handed out per queue (the DF_TRQ / DF_UPQ / DF_UPQF words)
    upq = [0] * (nw + 1)
    farq = [0] * (nw + 1)
    held = [None] * n_workers   # task index held by each wide worker
    ptr = [[0, 0, 0] for _ in range(n_workers)]   # (jtr, jn, jf) of each worker
    done_w = 0

    nxt = [None] * n_workers      # task selected ahead by each worker
    started = [False] * n_workers

    def take(w, allow_early=True):
        """the selection loop of ldlt_wide_kernel (one pass): returns a task index, None (nothing eligible), or 'done'"""
        while True:
            jtr, jn, jf = ptr[w]
            if jtr >= nw and jn >= nw and jf >= nw:
                return "done"
            if retire_at is not None and (w & 1) and jtr >= retire_at and jn >= retire_at and jf >= retire_at:
                return "done"      # (a worker only leaves between tasks: whatever it took, it finished)
            if jtr < nw:
                if trq[jtr] >= Q[jtr][1]:
                    ptr[w][0] += 1
                    continue
                ok = sim.cdone[jtr] >= 10 and (jtr < 1 or upq[jtr - 1] >= Q[jtr - 1][4]) and \
                    (jtr < NVB or sim.updone[jtr - NVB] >= sim.upcnt[jtr - NVB])
                if ok:
                    i = trq[jtr]; trq[jtr] += 1
                    return int(Q[jtr][0] + i)
            if jn < nw:
                if upq[jn] >= Q[jn][3]:
                    ptr[w][1] += 1
                    continue
                dq, need = int(FQ[jn][2]), int(FQ[jn][3])
                if trq[jn] >= Q[jn][1] and (dq < 0 or farq[dq] >= need):
                    i = upq[jn]; upq[jn] += 1
                    return int(Q[jn][2] + i)
            if jf < nw:
                if farq[jf] >= FQ[jf][1]:
                    ptr[w][2] += 1
                    continue
                if trq[jf] >= Q[jf][1]:
                    i = farq[jf]; farq[jf] += 1
                    return int(FQ[jf][0] + i)
            # nothing eligible: an EARLY substitution task (the chain has only started C_jtr); it is held until C_jtr is
            # complete (the kernel advances it block row by block row — here it simply blocks its worker, which is stricter)
            if allow_early and jtr < nw and trq[jtr] < Q[jtr][1] and 1 <= sim.cdone[jtr] < 10 and \
                    (jtr < 1 or upq[jtr - 1] >= Q[jtr - 1][3]) and (jtr < NVB or sim.updone[jtr - NVB] >= sim.upcnt[jtr - NVB]):
                i = trq[jtr]; trq[jtr] += 1
                return int(Q[jtr][0] + i)
            return None

    finished = [False] * n_workers
    while True:
        agents = []
        for r in range(roles):
            t = role_task(r)
            if t is not None:
                agents.append(("c", r, t))
        for w in range(n_workers):
            if not finished[w]:
                agents.append(("w", w, None))
        if not agents:
            break
        rnd.shuffle(agents)
        progressed = False
        for kind, who, t in agents:
            if kind == "c":
                j, tk = t
                if sim.chain_ready(j, tk):
                    sim.chain_run(j, tk)
                    cur[who][1] += 1
                    progressed = True
                    break            # one step, then re-shuffle: many different interleavings
            else:
                if held[who] is None and nxt[who] is not None:
                    held[who], nxt[who] = nxt[who], None
                    progressed = True
                    break
                if held[who] is None:
                    got = take(who)
                    if got == "done":
                        finished[who] = True
                        progressed = True
                        break
                    if got is None:
                        continue
                    held[who] = got
                    progressed = True   # taking a task is a step of its own: the run may have to wait
                    break
                if sim.wide_ready(wt[held[who]]):
                    tk = wt[held[who]]
                    if ahead and not started[who] and tk[0] in (UP, UP2) and 128 * (tk[2] + 1) <= sim.N and 128 * (tk[3] + 1) <= sim.N \
                            and rnd.random() < 0.7:
                        started[who] = True      # in its tile loop: one non-blocking look at the queues
                        got = take(who, allow_early=False)
                        if got not in (None, "done"):
                            nxt[who] = got
                        progressed = True
                        break
                    sim.wide_run(tk)
                    started[who] = False
                    held[who] = None
                    done_w += 1
                    progressed = True
                    break
        assert progressed, f"deadlock: chain cursors {cur}, held {held}, selected ahead {nxt}, queues TR {trq} UP {upq}"
    assert done_w == len(wt)
    return sim


def quasi_definite(n, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    m = n // 2
    G = r.uniform(-1, 1, (n, n)) * 0.02
    A = G + G.T
    d = np.concatenate([r.uniform(2, 4, m), -r.uniform(2, 4, n - m)])   # [H  J^T; J  -D] flavour: quasi-definite
    return A + np.diag(d)


@pytest.mark.parametrize("n,workers,seed", [(1024, 1, 0), (1024, 7, 1), (1280, 64, 2), (768, 3, 3)])
def test_schedule_is_live_and_sound(n, workers, seed):
    plan = get_plan(n)
    assert plan["nchain"] == n // 256 and plan["last_has_next"] == 0
    A = quasi_definite(n, seed)
    sim = replay(A, plan, workers, seed)
    want = ldl_nopiv(A)
    got = np.triu(sim.A)
    for j in range(plan["nsp"]):      # the factored diagonal blocks live in the compact copies
        got[256 * j:256 * j + 256, 256 * j:256 * j + 256] = np.triu(sim.Cd[j])
    assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.parametrize("n,workers,seed,retire_at", [(1536, 8, 4, 3), (1536, 2, 5, 0), (2048, 6, 6, 4)])
def test_schedule_stays_live_and_sound_when_every_second_worker_leaves_early(n, workers, seed, retire_at):
    """Round 3: the second workgroup of every CU leaves the wide kernel at super-panel jretire (the chain-bound half runs faster with one
    workgroup per CU).  A worker leaves only between tasks, and the protocol never counts on a particular number of workers — replayed
    here with odd workers leaving at `retire_at` (0: before they take anything)."""
    plan = get_plan(n)
    A = quasi_definite(n, seed)
    sim = replay(A, plan, workers, seed, retire_at=retire_at)
    want = ldl_nopiv(A)
    got = np.triu(sim.A)
    for j in range(plan["nsp"]):
        got[256 * j:256 * j + 256, 256 * j:256 * j + 256] = np.triu(sim.Cd[j])
    assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.parametrize("n,workers,seed", [(1536, 1, 7), (1536, 5, 8), (2048, 16, 9), (1024 + 100, 3, 10)])
def test_schedule_stays_live_and_sound_when_tasks_are_selected_ahead(n, workers, seed):
    """Round 4: the eight-wave wide kernel takes its next task while the current update tile is still in its loop."""
    plan = get_plan(n)
    A = quasi_definite(n, seed)
    sim = replay(A, plan, workers, seed, ahead=True)
    want = ldl_nopiv(A)
    s = 256 * plan["nchain"]
    got = np.triu(sim.A)
    for j in range(plan["nchain"]):
        got[256 * j:256 * j + 256, 256 * j:256 * j + 256] = np.triu(sim.Cd[j])
    assert np.abs(got[:s] - want[:s]).max() < 1e-10 * np.abs(want).max()


def test_ragged_order_hands_over_a_consistent_state():
    n = 1024 + 100
    plan = get_plan(n)
    assert plan["nchain"] == 3 and plan["last_has_next"] == 1 and plan["nwide"] == 3
    A = quasi_definite(n, 5)
    sim = replay(A, plan, 5, 5)
    want = ldl_nopiv(A)
    s = 256 * plan["nchain"]
    got = np.triu(sim.A)
    for j in range(plan["nchain"]):
        got[256 * j:256 * j + 256, 256 * j:256 * j + 256] = np.triu(sim.Cd[j])
    # rows of the chained super-panels are final
    assert np.abs(got[:s] - want[:s]).max() < 1e-10 * np.abs(want).max()
    # what is left for the stepwise kernels: the trailing matrix with every update of the chained panels applied
    # (its leading 256 x 256 block is the compact copy of super-panel nchain); finishing it reproduces the factor
    rest = got[s:, s:].copy()
    rest[:256, :256] = np.triu(sim.Cd[plan["nchain"]])
    fin = ldl_nopiv(rest)
    assert np.abs(fin - want[s:, s:]).max() < 1e-10 * np.abs(want).max()


def test_plan_shapes():
    p = get_plan(8192)
    assert p["nsp"] == 32 and p["nt"] == 64 and p["nchain"] == 32 and p["nwide"] == 31
    ups = [t for t in p["wtasks"] if t[0] in (UP, UPH, UP2)]
    # tiles of 31 trailing updates minus the skipped diagonal blocks; a fused (K = 512) task stands for two of them
    assert sum(2 if t[0] == UP2 else 1 for t in ups) == sum(t * (t + 1) // 2 - 3 for t in range(62, 0, -2))
    # default pairing: the update-bound first half, pairs (0,1) ... (14,15); the fused tasks sit in the odd queue and cover exactly
    # the tile rows behind super-panel j + 1
    for j in range(31):
        fused = [t for t in ups if t[0] == UP2 and t[1] == j]
        if j % 2 == 1 and j < 16:
            assert sorted((t[2], t[3]) for t in fused) == [(I, J) for I in range(2 * j + 4, 64) for J in range(I, 64)]
            assert not [t for t in ups if t[0] == UP and t[1] == j and t[2] >= 2 * j + 4]
            assert not [t for t in ups if t[0] == UP and t[1] == j - 1 and t[2] >= 2 * j + 4]
        else:
            assert not fused
    # the head tiles: the (up to) four tiles of A[R_j+1, next 256 columns], first in their queue
    assert sum(1 for t in ups if t[0] == UPH) == 4 * 30
    # the lists: TR tasks grouped by super-panel, then the update tasks grouped by super-panel (head tiles, the two tile rows of
    # the next row panel — counted in Q[j][4] —, then the rest row-major); the FAR lists are empty by default (HIOPAMD_DF_SPLIT=1
    # moves the rest of every super-panel there; the replay tests above run under that setting as well)
    Q, FQ = p["queues"], p["far"]
    assert len(Q) == 31 and len(FQ) == 31
    pos = 0
    for j in range(31):
        assert Q[j][0] == pos and all(t[0] == TR and t[1] == j for t in p["wtasks"][pos:pos + Q[j][1]])
        pos += Q[j][1]
    for j in range(31):
        seg = p["wtasks"][Q[j][2]:Q[j][2] + Q[j][3]]
        assert Q[j][2] == pos and all(t[0] in (UP, UPH, UP2) and t[1] == j for t in seg)
        nh = 4 if j < 30 else 0
        assert all(t[0] == UPH and t[2] < 2 * j + 4 and 2 * j + 4 <= t[3] < 2 * j + 6 for t in seg[:nh])
        assert all(t[2] in (2 * j + 2, 2 * j + 3) for t in seg[:Q[j][4]]) and all(t[2] >= 2 * j + 4 for t in seg[Q[j][4]:])
        rows = [int(t[2]) for t in seg[Q[j][4]:]]
        assert rows == sorted(rows)
        assert FQ[j][1] == 0 and FQ[j][2] == -1
        pos += Q[j][3]
    assert pos == len(p["wtasks"])


def test_chain_role_assignment_keeps_the_spine_fed():
    """The static role lists are part of the critical path (every role runs its list in order): pin the assignment that the
    measurements of round 2 arrived at (profiles/r02_probes/README.md) so that a change of df_chain_tasks is a conscious one."""
    p = get_plan(2048)
    lists = [[tuple(int(v) for v in t) for t in p["ctasks"][0][r] if t[0] != END] for r in range(p["roles"])]
    # role 0: the four fused spine steps; role 1: their companions without the diagonal update (z = 1)
    assert lists[0] == [(SP, q, 1, 0) for q in range(4)]
    assert lists[1] == [(RC, q, 1, 0) for q in range(4)]
    # role 2: per pivot the diagonal update U(p; p+2, p+2) FIRST (the next spine step waits for it), then column 3 of C_j
    assert lists[2][0] == (U, 0, 2, 2) and (CS, 0, 3, 0) in lists[2]
    assert [t for t in lists[2] if t[0] == U] == [(U, q, q + 2, q + 2) for q in range(4)]
    assert len(lists[2]) <= 6
    # the first H column: its role keeps T(p, 4) + the update of the next tile only; the left-overs ride on roles 13-15
    assert lists[3] == [(CS, 0, 4, 1), (CS, 1, 4, 2)]      # (T(2,4) belongs to the companion R(2), T(3,4) to the spine S(3))
    assert lists[13][0] == (U, 0, 2, 4) and lists[14][0] == (U, 0, 3, 4)
    assert (U, 1, 3, 4) in lists[15] and lists[15].index((U, 1, 3, 4)) == min(i for i, t in enumerate(lists[15]) if t[1] == 1)
    # every tile task of a super-panel appears exactly once (fused parts expanded)
    sim = Sim(np.eye(2048), p)
    seen = {}
    for r, lst in enumerate(lists):
        for t in lst:
            parts = sim.fused_parts(t) if t[0] in (SP, RC, CS) else [t]
            for q in parts:
                assert q not in seen, (q, r, seen[q])
                seen[q] = r
    want = {(F, q, q, q) for q in range(4)}
    want |= {(T, q, q, c) for q in range(4) for c in range(q + 1, 8)}
    want |= {(U, q, a, b) for q in range(4) for a in range(q + 1, 8) for b in range(a, 8)}
    assert set(seen) == want
