"""Discrete-event model of the dataflow LDL^T schedule (design aid, CPU only; not part of the product path).

Compares the round-2 schedule (right-looking: every super-panel's rank-256 update applied to the whole trailing matrix, FIFO by
panel, V workspaces recycled) with the round-3 one: the chain never waits for far-away tiles.  A tile row of panel p is brought
up to date only at the stages s (= after row panel s is substituted) with p - 1 - s in a set R that is dense near 0 and sparse
far away, by ONE long-K task that applies every panel it is behind by (left-looking bulk, K = 256 x batch); workers serve the
row panels in order of need (earliest deadline first), the substitution tasks before everything else.
Constants (us) are the measured ones of profiles/r02_probes/README.md (N = 8192 time line and phase sums).
"""
import heapq
import sys
from collections import defaultdict

CHAIN = 105.0      # period of the chain kernel when nothing else holds it up (tail of the round-2 time line)
CDONE = 93.0       # F(3) published after F(0) started
HDONE = 100.0      # last head tile solve after F(0) started
TR_T = 45.0        # substitution task (32 columns), slot time
OVH = 24.0         # tile task: select + wait + prologue + epilogue + publish
LOOP = 52.0        # 16 stages of a 128 x 128 x 256 tile with the partner workgroup also in its loop
UPH_TAIL = 25.0    # head tile: what is left after its last gated block row
SLOTS = 480


def rset(nsp, d, growth):
    R = set(range(d))
    r, g = float(d), 1.0
    while r < nsp:
        R.add(int(r))
        g *= growth
        r += max(1.0, g)
    return R


def simulate(nsp=32, mode="new", d=2, growth=1.5, nvb=3, verbose=False, slots=SLOTS, loop=LOOP, ovh=OVH):
    nt = 2 * nsp
    ver_done = {}                      # (I, J) -> [(j1, done time)]
    tr_done = defaultdict(dict)        # s -> {(J, q): done time}
    cstart = [None] * (nsp + 1)
    uph_done = [None] * (nsp + 1)
    cstart[0] = 0.0

    def vtime(I, J, j):
        if j <= 0:
            return 0.0
        for (j1, t) in ver_done.get((I, J), []):
            if j1 >= j:
                return t
        return None

    def wide_last(I, J):
        p = I // 2
        return max(0, p - 1) if J // 2 == p else p   # the chain applies panel p-1 to its next diagonal block

    head = defaultdict(list)          # s -> gated head tiles (rows of panel s+1, the 256 columns behind its diagonal block)
    rowq = defaultdict(list)          # p -> [(avail stage, I, J, j0, j1)] in (batch, tile) order
    R = rset(nsp, d, growth) if mode == "new" else set(range(nsp))
    for I in range(2, nt):
        p = I // 2
        for J in range(I, nt):
            last = wide_last(I, J)
            j0 = 0
            for s in range(last):
                if (p - 1 - s) in R or s == last - 1:
                    if s == p - 1 and 2 * p + 2 <= J < 2 * p + 4:
                        head[s].append((I, J))
                    else:
                        rowq[p].append((s, I, J, j0, s + 1))
                    j0 = s + 1
    for p in rowq:
        rowq[p].sort(key=lambda t: (t[0], t[1], t[2]))
    units = sum(t[4] - t[3] for p in rowq for t in rowq[p]) + sum(len(h) for h in head.values())
    ntasks = sum(len(q) for q in rowq.values())
    # round-2 mode: one FIFO over the stages (first the rows of the next panel, then the rest)
    fifo = []
    if mode == "old":
        for s in range(nsp - 1):
            st = [t for p in rowq for t in rowq[p] if t[0] == s]
            st.sort(key=lambda t: (t[1], t[2]))
            fifo.extend(st)
    fifo_before = defaultdict(int)
    for t in fifo:
        for s2 in range(t[0] + 1, nsp):
            fifo_before[s2] += 1
    stage_cnt = defaultdict(int)
    stage_done = defaultdict(int)
    stage_last = defaultdict(float)
    for p in rowq:
        for t in rowq[p]:
            stage_cnt[t[0]] += 1
    for s in head:
        stage_cnt[s] += len(head[s])
    first_rows = {s: sum(1 for t in fifo if t[0] == s and t[1] // 2 == s + 1) for s in range(nsp - 1)} if mode == "old" else {}
    fifo_stage_taken = defaultdict(int)

    ntr = {s: max(0, (nt - 2 * s - 4) * 4) for s in range(nsp - 1)}
    tr_taken = defaultdict(int)
    head_taken = defaultdict(int)
    pos = defaultdict(int)
    state = {"stage_tr": 0, "stage_h": 0, "plo": 1, "fifo_pos": 0}

    def chain_times(j):
        if cstart[j] is None:
            prev = cstart[j - 1]
            if prev is None or uph_done[j - 1] is None:
                return None
            t = max(prev + CHAIN, uph_done[j - 1] - 45.0)
            if mode == "old" and j >= nvb:
                if stage_done[j - nvb] < stage_cnt[j - nvb]:
                    return None
                t = max(t, stage_last[j - nvb])
            cstart[j] = t
        return cstart[j]

    def rows_ready(s, J):
        ts = [0.0]
        for I in (2 * s, 2 * s + 1):
            if J < I:
                continue
            v = vtime(I, J, wide_last(I, J))
            if v is None:
                return None
            ts.append(v)
        return max(ts)

    def finish(stage, end):
        stage_done[stage] += 1
        stage_last[stage] = max(stage_last[stage], end)

    def run_tile(now, t, cs_of):
        s, I, J, j0, j1 = t
        deps = [vtime(I, J, j0)]
        for B in (I, J):
            if B >= 2 * s + 4:
                deps.append(max(tr_done[s].get((B, q), 0.0) for q in range(4)))
            else:
                deps.append(cs_of + HDONE)
        start = max(now, max(deps))
        dur = ovh + loop * (j1 - j0)
        end = start + dur
        ver_done.setdefault((I, J), []).append((j1, end))
        finish(s, end)
        return end, dur

    def try_take(now):
        # ---- substitution tasks first
        while state["stage_tr"] < nsp - 1 and tr_taken[state["stage_tr"]] >= ntr[state["stage_tr"]]:
            state["stage_tr"] += 1
        s = state["stage_tr"]
        if s < nsp - 1:
            cs = chain_times(s)
            if cs is not None:
                i = tr_taken[s]
                early = mode == "new" and i < 16
                elig_t = cs + 17.0 if early else cs + CDONE
                if mode == "new":
                    prev_ok = pos[s] >= len(rowq[s])          # every tile task of row panel s handed out
                else:
                    prev_ok = s == 0 or fifo_stage_taken[s - 1] >= first_rows[s - 1]
                if now >= elig_t and prev_ok:
                    J = 2 * s + 4 + i // 4
                    rr = rows_ready(s, J)
                    if rr is not None:
                        tr_taken[s] += 1
                        start = max(now, rr)
                        end = max(start + TR_T, cs + CHAIN) if now < cs + CDONE else start + TR_T
                        tr_done[s][(J, i % 4)] = end
                        return end, TR_T
        # ---- the gated head tiles of the stage the chain works on
        while state["stage_h"] < nsp - 1 and head_taken[state["stage_h"]] >= len(head[state["stage_h"]]):
            if uph_done[state["stage_h"]] is None:
                cs = chain_times(state["stage_h"])
                if cs is None:
                    break
                hs = head[state["stage_h"]]
                uph_done[state["stage_h"]] = max([vtime(*h, state["stage_h"] + 1) for h in hs]) if hs else cs + HDONE
            state["stage_h"] += 1
        s = state["stage_h"]
        if s < nsp - 1 and head_taken[s] < len(head[s]):
            cs = chain_times(s)
            I, J = head[s][head_taken[s]]
            v = vtime(I, J, s)
            ok = cs is not None and v is not None and tr_taken[s] >= min(16, ntr[s])
            if mode == "old":
                ok = ok and tr_taken[s] >= ntr[s] and state["fifo_pos"] >= fifo_before[s]
            if ok:
                head_taken[s] += 1
                trh = max([tr_done[s].get((J, q), 0.0) for q in range(4)] + [0.0])
                end = max(now + ovh + loop, max(cs + HDONE, trh, v) + UPH_TAIL)
                ver_done.setdefault((I, J), []).append((s + 1, end))
                finish(s, end)
                return end, ovh + loop
        # ---- tile tasks
        if mode == "old":
            if state["fifo_pos"] < len(fifo):
                t = fifo[state["fifo_pos"]]
                sN = t[0]
                cs = chain_times(sN)
                if cs is not None and tr_taken[sN] >= ntr[sN] and head_taken[sN] >= len(head[sN]) and vtime(t[1], t[2], t[3]) is not None:
                    state["fifo_pos"] += 1
                    fifo_stage_taken[sN] += 1
                    return run_tile(now, t, cs)
            return None
        while state["plo"] < nsp and pos[state["plo"]] >= len(rowq[state["plo"]]):
            state["plo"] += 1
        for p in range(state["plo"], nsp):
            if pos[p] >= len(rowq[p]):
                continue
            t = rowq[p][pos[p]]
            sN = t[0]
            if tr_taken[sN] < ntr[sN]:
                continue
            v = vtime(t[1], t[2], t[3])
            if v is None or v > now:
                continue
            cs = chain_times(sN)
            if cs is None:
                continue
            pos[p] += 1
            return run_tile(now, t, cs)
        return None

    free = [(0.0, w) for w in range(slots)]
    heapq.heapify(free)
    t_end, busy, guard = 0.0, 0.0, 0
    while True:
        guard += 1
        if guard > 20_000_000:
            raise RuntimeError("simulation does not terminate")
        t, w = heapq.heappop(free)
        r = try_take(t)
        if r is None:
            all_out = state["stage_tr"] >= nsp - 1 and state["stage_h"] >= nsp - 1 and (
                state["fifo_pos"] >= len(fifo) if mode == "old" else all(pos[p] >= len(rowq[p]) for p in rowq))
            if all_out:
                break
            heapq.heappush(free, (t + 2.0, w))
            continue
        end, dur = r
        busy += dur
        t_end = max(t_end, end)
        heapq.heappush(free, (end, w))
    for j in range(1, nsp):
        if uph_done[j - 1] is None:
            hs = head[j - 1]
            uph_done[j - 1] = max([vtime(*h, j) or 0.0 for h in hs]) if hs else (cstart[j - 1] or 0.0) + HDONE
        chain_times(j)
    last_chain = (cstart[nsp - 1] if cstart[nsp - 1] is not None else t_end) + CDONE
    total = max(t_end, last_chain)
    if verbose:
        print("  chain starts:", " ".join("%.0f" % (c if c is not None else -1) for c in cstart[:nsp]))
        print("  tile tasks %d for %d tile-panel units (mean batch %.2f panels)" % (ntasks, units, units / max(1, ntasks + sum(len(h) for h in head.values()))))
    return total, busy / (slots * total)


if __name__ == "__main__":
    nsp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    t, u = simulate(nsp, "old", verbose=True)
    print("round-2 schedule (right-looking, 3 V workspaces): %.0f us, slot utilisation %.2f" % (t, u))
    for d in (1, 2, 3, 4):
        for g in (1.0, 1.25, 1.5, 2.0):
            t, u = simulate(nsp, "new", d=d, growth=g, verbose=(d == 2 and g == 1.5))
            print("every stage for the last d=%d, gap growth %.2f: %.0f us, slot utilisation %.2f" % (d, g, t, u))
