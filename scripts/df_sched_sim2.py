"""Discrete-event model of the dataflow LDL^T schedule, second form: the MFMA pipe of a CU is the resource (two resident
workgroups per CU share it while both are in their tile loops; a workgroup that waits, substitutes or runs a prologue leaves
the pipe to its partner).  Design aid, CPU only; not part of the product path.

mode "old": round-2 schedule (right-looking, FIFO by panel, 3 V workspaces).
mode "new": round-3 schedule — tile rows of panel p are brought up to date at the stages s with p-1-s in R (dense near 0,
            sparse far away) by ONE long-K task per tile; workers serve substitution tasks first, then the head tiles, then the
            row panels in order of need.
Constants (us): profiles/r02_probes/README.md.
"""
import heapq
import sys
from collections import defaultdict

CHAIN = 105.0
F0 = 17.0          # F(0) published after the chain started on the panel
BROW = 25.0        # one more block row of C_j per 25 us
CDONE = 93.0
HDONE = 100.0
TR_ROW = 10.0      # a substitution task: per block row once its inputs are there
TR_TAIL = 5.0
PRE = 12.0         # tile task before its loop: ticket, decode, flags, C tile + first operand stage
POST = 8.0         # epilogue + publish
WORK = 32.6        # 16 stages with the pipe to itself (27.7 us of MFMA at the 85 % of the stage loop)
ALONE = 0.92       # a lone workgroup reaches 92 % of the two-workgroup rate
UPH_TAIL = 25.0
NCU = 240


def rset(nsp, d, growth):
    R = set(range(d))
    r, g = float(d), 1.0
    while r < nsp:
        R.add(int(r))
        g *= growth
        r += max(1.0, g)
    return R


class Sim:
    def __init__(self, nsp=32, mode="new", d=2, growth=1.5, nvb=3, early_all=True, ncu=NCU, cap=99):
        self.nsp, self.mode, self.nvb, self.early_all, self.ncu = nsp, mode, nvb, early_all, ncu
        self.nt = nt = 2 * nsp
        self.ver = defaultdict(int)
        self.trd = set()                 # (s, J, q) done
        self.trrow = defaultdict(int)    # (s, J, q) -> block rows done
        self.cstart = [None] * (nsp + 1)
        self.cstart[0] = 0.0
        self.uph_done = [None] * (nsp + 1)
        self.head = defaultdict(list)
        self.rowq = defaultdict(list)
        R = rset(nsp, d, growth) if mode == "new" else set(range(nsp))
        for I in range(2, nt):
            p = I // 2
            for J in range(I, nt):
                last = self.wide_last(I, J)
                j0 = 0
                for s in range(last):
                    if (p - 1 - s) in R or s == last - 1:
                        if s == p - 1 and 2 * p + 2 <= J < 2 * p + 4:
                            self.head[s].append((I, J))
                        else:
                            k = j0
                            while k < s + 1:
                                k1 = min(k + cap, s + 1)
                                self.rowq[p].append((s, I, J, k, k1))
                                k = k1
                        j0 = s + 1
        for p in self.rowq:
            self.rowq[p].sort(key=lambda t: (t[0], t[3], t[1], t[2]))
        self.fifo = []
        if mode == "old":
            for s in range(nsp - 1):
                st = [t for p in self.rowq for t in self.rowq[p] if t[0] == s]
                st.sort(key=lambda t: (t[1], t[2]))
                self.fifo.extend(st)
        self.fifo_before = defaultdict(int)
        for t in self.fifo:
            self.fifo_before[t[0] + 1] += 1
        for s in range(1, nsp + 1):
            self.fifo_before[s] += self.fifo_before[s - 1]
        self.stage_cnt = defaultdict(int)
        self.stage_done = defaultdict(int)
        for p in self.rowq:
            for t in self.rowq[p]:
                self.stage_cnt[t[0]] += 1
        for s in self.head:
            self.stage_cnt[s] += len(self.head[s])
        self.first_rows = {s: sum(1 for t in self.fifo if t[0] == s and t[1] // 2 == s + 1) for s in range(nsp - 1)}
        self.fifo_stage_taken = defaultdict(int)
        self.ntr = {s: max(0, (nt - 2 * s - 4) * 4) for s in range(nsp - 1)}
        self.tr_taken = defaultdict(int)
        self.head_taken = defaultdict(int)
        self.head_done = defaultdict(int)
        self.pos = defaultdict(int)
        self.stage_tr = self.stage_h = 0
        self.plo = 1
        self.fifo_pos = 0
        self.units = sum(t[4] - t[3] for p in self.rowq for t in self.rowq[p]) + sum(len(h) for h in self.head.values())
        self.ntasks = sum(len(q) for q in self.rowq.values()) + sum(len(h) for h in self.head.values())

    def wide_last(self, I, J):
        p = I // 2
        return max(0, p - 1) if J // 2 == p else p

    def chain_time(self, j, now):
        """start of the chain's work on panel j if it is known to have started by `now`"""
        if self.cstart[j] is None:
            prev = self.cstart[j - 1]
            if prev is None or self.uph_done[j - 1] is None:
                return None
            t = max(prev + CHAIN, self.uph_done[j - 1] - 45.0)
            if self.mode == "old" and j >= self.nvb:
                if self.stage_done[j - self.nvb] < self.stage_cnt[j - self.nvb]:
                    return None
                t = max(t, self.stage_last.get(j - self.nvb, 0.0))
            self.cstart[j] = t
        return self.cstart[j] if self.cstart[j] <= now else None

    stage_last = {}

    def rows_ready(self, s, J):
        for I in (2 * s, 2 * s + 1):
            if J >= I and self.ver[(I, J)] < self.wide_last(I, J):
                return False
        return True

    # ---- task selection: returns a task descriptor or None
    def select(self, now):
        nsp = self.nsp
        while self.stage_tr < nsp - 1 and self.tr_taken[self.stage_tr] >= self.ntr[self.stage_tr]:
            self.stage_tr += 1
        s = self.stage_tr
        tr_early = None
        if s < nsp - 1:
            cs = self.chain_time(s, now)
            if cs is not None:
                i = self.tr_taken[s]
                if self.mode == "new":
                    prev_ok = self.pos[s] >= len(self.rowq[s])
                else:
                    prev_ok = s == 0 or self.fifo_stage_taken[s - 1] >= self.first_rows[s - 1]
                if prev_ok:
                    if now >= cs + CDONE or (self.mode == "new" and (i < 16 or self.early_all) and now >= cs + F0):
                        self.tr_taken[s] += 1
                        return ("TR", s, 2 * s + 4 + i // 4, i % 4)
                    if now >= cs + F0:
                        tr_early = (s, i)
        while self.stage_h < nsp - 1 and self.head_taken[self.stage_h] >= len(self.head[self.stage_h]):
            if not self.head[self.stage_h] and self.uph_done[self.stage_h] is None:
                cs = self.chain_time(self.stage_h, now)
                if cs is None:
                    break
                self.uph_done[self.stage_h] = cs + HDONE
            self.stage_h += 1
        s = self.stage_h
        if s < nsp - 1 and self.head_taken[s] < len(self.head[s]):
            cs = self.chain_time(s, now)
            I, J = self.head[s][self.head_taken[s]]
            ok = cs is not None and self.ver[(I, J)] >= s and self.tr_taken[s] >= min(16, self.ntr[s])
            if self.mode == "old":
                ok = ok and self.tr_taken[s] >= self.ntr[s] and self.fifo_pos >= self.fifo_before[s]
            if ok:
                self.head_taken[s] += 1
                return ("HEAD", s, I, J)
        if self.mode == "old":
            if self.fifo_pos < len(self.fifo):
                t = self.fifo[self.fifo_pos]
                sN = t[0]
                if self.chain_time(sN, now) is not None and self.tr_taken[sN] >= self.ntr[sN] and self.head_taken[sN] >= len(self.head[sN]):
                    self.fifo_pos += 1
                    self.fifo_stage_taken[sN] += 1
                    return ("TILE",) + t
        else:
            while self.plo < nsp and self.pos[self.plo] >= len(self.rowq[self.plo]):
                self.plo += 1
            for p in range(self.plo, nsp):
                if self.pos[p] >= len(self.rowq[p]):
                    continue
                t = self.rowq[p][self.pos[p]]
                sN = t[0]
                if self.tr_taken[sN] < self.ntr[sN] or self.ver[(t[1], t[2])] < t[3]:
                    continue
                self.pos[p] += 1
                return ("TILE",) + t
        if tr_early is not None and (self.early_all or self.mode == "old"):
            s, i = tr_early
            self.tr_taken[s] += 1
            return ("TR", s, 2 * s + 4 + i // 4, i % 4)
        return None

    def all_out(self):
        if self.stage_tr < self.nsp - 1 or self.stage_h < self.nsp - 1:
            return False
        if self.mode == "old":
            return self.fifo_pos >= len(self.fifo)
        return all(self.pos[p] >= len(self.rowq[p]) for p in self.rowq)

    def tile_deps_ok(self, t, now):
        _, s, I, J, j0, j1 = t
        if self.ver[(I, J)] < j0:
            return False
        for B in (I, J):
            if B >= 2 * s + 4:
                if any((s, B, q) not in self.trd for q in range(4)):
                    return False
            else:
                cs = self.chain_time(s, now)
                if cs is None or now < cs + HDONE:
                    return False
        return True

    def run(self, verbose=False):
        nw = 2 * self.ncu
        # worker state
        st = ["idle"] * nw          # idle | wait | pre | loop | post | tr | head
        task = [None] * nw
        rem = [0.0] * nw            # remaining pipe work (loop phase)
        last = [0.0] * nw
        rate = [0.0] * nw
        gen = [0] * nw
        ev = []
        seq = 0
        pipe_busy = 0.0

        def push(t, w, kind):
            nonlocal seq
            seq += 1
            heapq.heappush(ev, (t, seq, w, kind, gen[w]))

        def settle(w, now):
            nonlocal pipe_busy
            if st[w] == "loop":
                done = (now - last[w]) * rate[w]
                rem[w] -= done
                pipe_busy += done
                last[w] = now

        def rerate(w, now):
            """recompute the loop rates of w's CU and reschedule loop ends"""
            a, b = w & ~1, w | 1
            for x in (a, b):
                settle(x, now)
            both = st[a] == "loop" and st[b] == "loop"
            for x in (a, b):
                if st[x] == "loop":
                    r = 0.5 if both else ALONE
                    rate[x] = r
                    gen[x] += 1
                    push(now + max(0.0, rem[x]) / r, x, "loopend")

        for w in range(nw):
            push(0.0, w, "poll")
        t_end = 0.0
        guard = 0
        while ev:
            guard += 1
            if guard > 60_000_000:
                raise RuntimeError("no termination")
            now, _, w, kind, g = heapq.heappop(ev)
            if kind == "loopend":
                if g != gen[w] or st[w] != "loop":
                    continue
                settle(w, now)
                st[w] = "post"
                rerate(w, now)
                push(now + POST, w, "done")
                continue
            if kind == "poll":
                if st[w] == "idle":
                    tk = self.select(now)
                    if tk is None:
                        if self.all_out():
                            continue
                        push(now + 2.0, w, "poll")
                        continue
                    task[w] = tk
                    st[w] = "wait"
                if st[w] == "wait":
                    tk = task[w]
                    if tk[0] == "TILE":
                        if self.tile_deps_ok(tk, now):
                            st[w] = "pre"
                            push(now + PRE, w, "prend")
                        else:
                            push(now + 1.0, w, "poll")
                    elif tk[0] == "TR":
                        _, s, J, q = tk
                        cs = self.chain_time(s, now)
                        P = self.trrow[(s, J, q)]
                        if cs is not None and now >= cs + F0 + BROW * P and self.rows_ready(s, J):
                            st[w] = "tr"
                            push(now + TR_ROW, w, "trrow")
                        else:
                            push(now + 1.0, w, "poll")
                    else:   # HEAD
                        s, I, J = tk[1], tk[2], tk[3]
                        cs = self.chain_time(s, now)
                        ready = cs is not None and now >= cs + HDONE and all((s, J, q) in self.trd for q in range(4))
                        tk_start = tk[4] if len(tk) > 4 else None
                        if tk_start is None:
                            task[w] = tk + (now,)
                            tk_start = now
                        if ready:
                            st[w] = "head"
                            push(max(now + UPH_TAIL, tk_start + PRE + WORK * 2 + POST), w, "done")
                        else:
                            push(now + 1.0, w, "poll")
                continue
            if kind == "prend":
                st[w] = "loop"
                rem[w] = WORK * (task[w][5] - task[w][4])
                last[w] = now
                rerate(w, now)
                continue
            if kind == "trrow":
                _, s, J, q = task[w]
                self.trrow[(s, J, q)] += 1
                if self.trrow[(s, J, q)] >= 4:
                    push(now + TR_TAIL, w, "done")
                else:
                    st[w] = "wait"
                    push(now, w, "poll")
                continue
            if kind == "done":
                tk = task[w]
                if tk[0] == "TILE":
                    _, s, I, J, j0, j1 = tk
                    self.ver[(I, J)] = j1
                    self.stage_done[s] += 1
                    self.stage_last[s] = max(self.stage_last.get(s, 0.0), now)
                elif tk[0] == "TR":
                    self.trd.add((tk[1], tk[2], tk[3]))
                else:
                    _, s, I, J, _t0 = tk
                    self.ver[(I, J)] = s + 1
                    self.head_done[s] += 1
                    self.stage_done[s] += 1
                    self.stage_last[s] = max(self.stage_last.get(s, 0.0), now)
                    if self.head_done[s] == len(self.head[s]):
                        self.uph_done[s] = now
                t_end = max(t_end, now)
                st[w] = "idle"
                task[w] = None
                push(now, w, "poll")
        for j in range(1, self.nsp):
            if self.uph_done[j - 1] is None:
                self.uph_done[j - 1] = (self.cstart[j - 1] or 0.0) + HDONE
            self.chain_time(j, 1e18)
        total = max(t_end, (self.cstart[self.nsp - 1] or t_end) + CDONE)
        if verbose:
            print("  chain starts:", " ".join("%.0f" % (c if c is not None else -1) for c in self.cstart[:self.nsp]))
            print("  %d tile tasks for %d tile-panel units (mean batch %.2f panels)" % (self.ntasks, self.units, self.units / self.ntasks))
        return total, pipe_busy / (self.ncu * total)


if __name__ == "__main__":
    nsp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    Sim.stage_last = {}
    t, u = Sim(nsp, "old").run(verbose=True)
    print("round-2 schedule (right-looking, 3 V workspaces): %.0f us, pipe utilisation %.2f" % (t, u))
    for early in (True, False):
        for d in (1, 2, 3):
            for g in (1.0, 1.5, 2.0):
                Sim.stage_last = {}
                t, u = Sim(nsp, "new", d=d, growth=g, early_all=early).run(verbose=(d == 2 and g == 1.5))
                print("early-TR-all=%d  d=%d, gap growth %.2f: %.0f us, pipe utilisation %.2f" % (early, d, g, t, u))


def diagnose(nsp=32, d=2, growth=1.5, early=True, cap=99):
    """per stage: when the chain started, when the head tiles were handed out / finished and what they waited for"""
    Sim.stage_last = {}
    sim = Sim(nsp, "new", d=d, growth=growth, early_all=early, cap=cap)
    rec = defaultdict(dict)
    orig_select = sim.select

    def select(now):
        tk = orig_select(now)
        if tk is not None:
            if tk[0] == "HEAD":
                rec[tk[1]].setdefault("head_taken", now)
            if tk[0] == "TR" and tk[3] == 0 and tk[2] == 2 * tk[1] + 4:
                rec[tk[1]]["tr0_taken"] = now
            if tk[0] == "TILE" and tk[2] // 2 == tk[1] + 1:
                rec[tk[1]].setdefault("r0_first_taken", now)
                rec[tk[1]]["r0_last_taken"] = now
        return tk
    sim.select = select
    total, u = sim.run()
    print("total %.0f us, pipe %.2f" % (total, u))
    print("stage | chain start | first TR taken | r=0 tiles of rows s+1 taken (first, last) | head tiles taken | head done | next chain start")
    for s in range(nsp - 1):
        r = rec[s]
        print("%3d | %7.0f | %7.0f | %7.0f %7.0f | %7.0f | %7.0f | %7.0f" % (s, sim.cstart[s] or -1, r.get("tr0_taken", -1) - (sim.cstart[s] or 0),
              r.get("r0_first_taken", -1) - (sim.cstart[s] or 0), r.get("r0_last_taken", -1) - (sim.cstart[s] or 0),
              r.get("head_taken", -1) - (sim.cstart[s] or 0), (sim.uph_done[s] or -1) - (sim.cstart[s] or 0), (sim.cstart[s + 1] or -1) - (sim.cstart[s] or 0)))
