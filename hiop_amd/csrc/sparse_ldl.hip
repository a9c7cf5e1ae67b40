// A general sparse symmetric DIRECT solver, M = P^T L D L^T P without numerical pivoting — the inner solver of the condensed sparse KKT
//  for patterns that are neither small (dense LDL^T, nx <= 4096) nor bordered diagonals (arrow_ldl.hip).
//
// reference: hiopKKTLinSysCondensedSparse factors M = H + Dx + delta_wx + Jd^T (Dd + delta_wd) Jd with a sparse Cholesky (MA57 /
// cuSOLVER, src/Optimization/hiopKKTLinSysSparseCondensed.cpp:469-496); "the factorisation exists with positive pivots" IS the
// positive-definiteness verdict its inertia-correction loop branches on (:386-388), and that verdict must not depend on a right-hand side.
// Neither library is in the image; this file is the MI355X-native counterpart of that role:
//
//   symbolic (host, once per pattern)
//     * rows with more than max(64, min(10 sqrt(n), 20 x the average row)) entries are hubs: set aside, ordered last;
//     * the rest is ordered by NESTED DISSECTION on BFS level structures (George's automatic nested dissection: pseudo-peripheral root,
//       the cheapest middle level is the separator, recurse on the two sides; regions of <= SL_LEAF vertices are leaves).  Minimum degree
//       would give less fill but, on the banded / chain-like patterns this is for, an elimination tree that is a PATH (n dependent
//       steps); dissection gives a tree of depth O(log n) whose levels are what the device runs in parallel;
//     * every leaf region and every separator (in chunks of <= SL_LEAF columns) is ONE supernode = one dense frontal matrix (relaxed:
//       structural zeros inside a leaf are carried along); row structures by one symbolic pass in elimination order;
//     * supernodes whose front has more than SL_T rows, and everything above them in the tree, are not eliminated sparsely: their
//       columns form the ROOT, one dense matrix (order r) that receives the Schur complement of everything below and is factored by
//       this library's dense LDL^T (row a15).  r > SL_ROOT_MAX: HIOPAMD_ERR_STATE, the caller keeps its other inner solvers;
//     * gather plans: every entry of a front (and of the root) is the sum, in a FIXED order, of entries of M and of entries of its
//       children's update matrices — no atomics, bitwise reproducible.
//   numeric (device): one launch per level of the supernode tree, one workgroup per front: gather the front into LDS (lower triangle),
//     right-looking LDL^T of its pivot columns, store the L panel and the update matrix; then gather + factor the root.  Round 6: levels
//     of small fronts with many pivots (<= 40 rows: the leaves of a banded pattern) by ONE WAVE per front with the front in registers.
//     Non-positive pivots are COUNTED (integer atomics), the factorisation goes on: inertia = (#negative, #zero) of the sparse fronts +
//     the root's, exact by Sylvester's law whenever no pivot is zero (no pivoting: a matrix that is not quasi-definite may break down —
//     reported as n_zero > 0, which the caller treats like the reference treats a failed Cholesky).
//   solve: forward by levels (front vectors gathered like the fronts), dense root solve, backward by levels.  No host round trip.
#include "device_utils.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <future>
#include <numeric>
#include <thread>
#include <vector>

#define RC(x)                         \
  do {                                \
    const int rc__ = (x);             \
    if(rc__ != HIOPAMD_OK) return rc__; \
  } while(0)

using namespace hiopamd;

#include <chrono>
namespace {
int sl_host_threads(int64_t work_items);   // (below, with the gather plans)
// HIOPAMD_SL_TIMING=1: wall time of the phases of the symbolic analysis on stderr (host code; profiling aid)
struct SlStopwatch {
  const bool on = std::getenv("HIOPAMD_SL_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what)
  {
    if(!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[hiop_amd] sparse analysis: %-34s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
    t = now;
  }
};
}  // namespace

#ifndef HIOPAMD_SL_LEAF
#define HIOPAMD_SL_LEAF 32
#endif
// columns per supernode (leaf regions, separator chunks); < 64: a front's pivot rows fit one register per lane.  32 since round 6 (48
// before): a leaf of a narrow-band pattern then has at most 32 + 2 x bandwidth rows and is factored in registers (sl_factor_regs_kernel,
// <= 40 rows) whatever the order — with 48 the leaves of n = 2e5 / 7e5, bandwidth 5 came out at 53 rows (LDS kernel: 0.31 / 0.67 ms per
// factorisation against 0.21 / 0.40), those of n = 1e6 at 39 by luck; the solves pay 0-5 % for the extra level
// (profiles/r06_probes/call25_*).
constexpr int SL_LEAF = HIOPAMD_SL_LEAF;
constexpr int SL_PACK = 8;           // columns of a supernode that packs several tiny independent components
constexpr int SL_T = 128;            // rows of a front that is eliminated in LDS (128 x 129 doubles = 132 KB)
constexpr int SL_ROOT_MAX = 20480;   // order of the dense root (3.4 GB)

struct SlSymbolic {
  int n = 0;
  std::vector<int> perm, iperm;          // new -> old, old -> new
  int ns = 0;                            // supernodes (sparse and root), in elimination order
  std::vector<int> sfirst, snc;          // first column (new numbering) and number of columns
  std::vector<int> srow_ptr, srows;      // row structure (new numbering, ascending, all > last column)
  std::vector<int> sparent;              // parent supernode or -1
  std::vector<char> sroot;               // 1: its columns belong to the dense root
  std::vector<int> slevel;               // sparse supernodes: level in the tree (leaves 0)
  int r = 0;                             // order of the root
  std::vector<int> root_cols;            // new indices of the root's columns, ascending
  std::vector<int> root_pos;             // new index -> position in the root or -1
  int nlevels = 0;
  int64_t nnzL = 0;                      // entries of the sparse L panels + r (r + 1) / 2
};

namespace {

// ---- nested dissection ------------------------------------------------------------------------------------------------------
// Round 6: the two sides of a cut are independent sub-problems (no edge joins them once the separator is out), so the top levels of the
// recursion run on separate host threads: a sub-dissector shares the vertex-indexed work arrays (`mark`, `level`: the sets are disjoint)
// and owns its queue, its output and a RANGE of set ids (ids are only ever compared for equality).  Its output is spliced into the
// parent's in the order the sequential recursion would have produced it (side A, side B, separator last): the ordering — and with it
// every plan, every sum, every bit of the factor — does not depend on the number of threads or on their timing.
struct Dissector {
  int n;
  const int* rp;
  const int* ci;
  std::vector<char> removed;        // dense rows
  std::vector<int> mark_store, level_store;   // (owned by the top-level dissector only)
  int* mark = nullptr;
  int* level = nullptr;
  std::vector<int> queue, order;    // order: new -> old, filled in elimination order
  std::vector<int> sn_first, sn_nc;
  int stamp = 0;
  std::atomic<int>* next_stamp = nullptr;   // where sub-dissectors get their id ranges
  int par_depth = 0;                        // levels of the recursion that may still fork
  static constexpr int PAR_MIN = 20000;     // vertices on a side below which forking is not worth a thread

  void emit_supernodes(const std::vector<int>& verts)
  {
    for(size_t o = 0; o < verts.size(); o += SL_LEAF) {
      const int nc = (int)std::min<size_t>(SL_LEAF, verts.size() - o);
      sn_first.push_back((int)order.size());
      sn_nc.push_back(nc);
      for(int q = 0; q < nc; ++q) order.push_back(verts[o + q]);
    }
  }
  // BFS inside the vertex set whose members carry mark == id; returns the number of levels, `queue` holds the vertices in BFS order
  // and level[v] their levels
  int bfs(int root, int id)
  {
    queue.clear();
    queue.push_back(root);
    level[root] = 0;
    mark[root] = -id;   // visited
    int nl = 1;
    for(size_t h = 0; h < queue.size(); ++h) {
      const int v = queue[h];
      for(int q = rp[v]; q < rp[v + 1]; ++q) {
        const int w = ci[q];
        if(w != v && mark[w] == id) {
          mark[w] = -id;
          level[w] = level[v] + 1;
          nl = level[w] + 1;
          queue.push_back(w);
        }
      }
    }
    for(int v : queue) mark[v] = id;   // un-visit
    return nl;
  }
  void dissect(std::vector<int> verts)
  {
    // explicit work stack of (vertex set, "emit as separator chain") — sets are moved, never copied
    struct Item {
      std::vector<int> v;
      bool chain;
      bool connected;
    };
    std::vector<Item> stack;
    stack.push_back({std::move(verts), false, false});
    // post-order emission needs "parts first, separator last": a separator is pushed BEFORE its parts (LIFO)
    while(!stack.empty()) {
      Item it = std::move(stack.back());
      stack.pop_back();
      std::vector<int>& V = it.v;
      if(V.empty()) continue;
      if(it.chain || (int)V.size() <= SL_LEAF) {
        emit_supernodes(V);
        continue;
      }
      int id = ++stamp;
      for(int v : V) mark[v] = id;
      // connected components first, all of them in ONE sweep (a bordered pattern without its border is a million singletons)
      if(!it.connected) {
        std::vector<std::vector<int>> comps;
        for(int v0 : V) {
          if(mark[v0] != id) continue;
          const int cid = ++stamp;
          std::vector<int> comp;
          comp.push_back(v0);
          mark[v0] = cid;
          for(size_t h = 0; h < comp.size(); ++h) {
            const int v = comp[h];
            for(int q = rp[v]; q < rp[v + 1]; ++q) {
              const int w = ci[q];
              if(mark[w] == id) {
                mark[w] = cid;
                comp.push_back(w);
              }
            }
          }
          if(comp.size() == V.size()) break;   // connected: dissect it below
          comps.push_back(std::move(comp));
        }
        if(!comps.empty()) {
          // independent sub-problems.  Small ones are packed: consecutive components are merged while the group stays <= SL_LEAF columns
          // (one supernode with a block-diagonal pivot block; its front carries the union of the groups' boundaries)
          std::vector<int> pack;
          for(auto& c : comps) {
            if((int)c.size() > SL_LEAF) {
              stack.push_back({std::move(c), false, true});
              continue;
            }
            if(pack.size() + c.size() > (size_t)SL_PACK) {
              emit_supernodes(pack);
              pack.clear();
            }
            pack.insert(pack.end(), c.begin(), c.end());
          }
          if(!pack.empty()) emit_supernodes(pack);
          continue;
        }
        id = ++stamp;
        for(int v : V) mark[v] = id;
      }
      // pseudo-peripheral root: a few sweeps from the last vertex of the deepest level with the smallest degree
      int root = V[0], nl = 0;
      for(int sweep = 0; sweep < 4; ++sweep) {
        const int l2 = bfs(root, id);
        if(l2 <= nl) break;
        nl = l2;
        int best = queue.back(), bdeg = rp[best + 1] - rp[best];
        for(size_t h = queue.size(); h-- > 0 && level[queue[h]] == nl - 1;) {
          const int d = rp[queue[h] + 1] - rp[queue[h]];
          if(d < bdeg) {
            bdeg = d;
            best = queue[h];
          }
        }
        root = best;
      }
      nl = bfs(root, id);
      if(nl < 3) {   // no middle level: a clique-like set, eliminated as a chain of supernodes (its fronts decide whether it is root)
        emit_supernodes(V);
        continue;
      }
      std::vector<int64_t> cnt((size_t)nl, 0);
      for(int v : queue) cnt[(size_t)level[v]] += 1;
      // the separator: the smallest level that leaves at least 15 % of the set on either side.  No such level, or a separator of more
      // than a third of the set (graphs of small diameter: expanders, cliques): dissection buys nothing — the set is eliminated as a
      // chain of supernodes, i.e. it ends up in the dense root unless it is small.  (Unbalanced cuts would also make the recursion as
      // deep as the set is large.)
      const int64_t tot = (int64_t)V.size();
      int64_t below = cnt[0];
      int ms = -1;
      int64_t best = tot;
      double best_cost = 1e300;
      for(int m = 1; m + 1 < nl; ++m) {
        const int64_t above = tot - below - cnt[(size_t)m];
        const double cost = (double)cnt[(size_t)m] * (1.0 + std::fabs((double)(below - above)) / (double)tot);   // equal sizes: the balanced one
        if(20 * below >= 3 * tot && 20 * above >= 3 * tot && cost < best_cost) {
          best_cost = cost;
          best = cnt[(size_t)m];
          ms = m;
        }
        below += cnt[(size_t)m];
      }
      if(ms < 0 || 3 * best > tot) {
        emit_supernodes(V);
        continue;
      }
      std::vector<int> A, B, S;
      for(int v : queue) {
        if(level[v] < ms) A.push_back(v);
        else if(level[v] > ms) B.push_back(v);
        else S.push_back(v);
      }
      stack.push_back({std::move(S), true, false});    // emitted last
      if(par_depth > 0 && next_stamp && (int)A.size() >= PAR_MIN && (int)B.size() >= PAR_MIN) {
        // both sides at once; their supernodes go in front of whatever this dissector emits next (S is on top of the stack)
        auto sub = [this](std::vector<int> verts) {
          Dissector d;
          d.n = n; d.rp = rp; d.ci = ci;
          d.mark = mark; d.level = level;
          d.next_stamp = next_stamp;
          d.par_depth = par_depth - 1;
          d.stamp = next_stamp->fetch_add(4 * (int)verts.size() + 64);   // (a set of m vertices uses at most ~2 m ids: components + cuts)
          d.dissect(std::move(verts));
          return d;
        };
        std::future<Dissector> fa = std::async(std::launch::async, sub, std::move(A));
        Dissector db = sub(std::move(B));
        Dissector da = fa.get();
        for(const Dissector* d : {&da, &db}) {
          const int base = (int)order.size();
          for(size_t q = 0; q < d->sn_first.size(); ++q) {
            sn_first.push_back(base + d->sn_first[q]);
            sn_nc.push_back(d->sn_nc[q]);
          }
          order.insert(order.end(), d->order.begin(), d->order.end());
        }
        continue;
      }
      stack.push_back({std::move(B), false, false});
      stack.push_back({std::move(A), false, false});
    }
  }
};

// rows with more off-diagonal entries than this are "hubs": set aside, ordered last (they end up in the dense root).  AMD's rule is
// 10 sqrt(n); the border rows of a bordered pattern are often shorter than that and still 100x the typical row, hence 20x the average.
int sl_dense_threshold(int n, const int* rp)
{
  const double avg = n > 0 ? (double)rp[n] / (double)n : 0.0;
  return std::max(64, (int)std::min(10.0 * std::sqrt((double)n), 20.0 * avg));
}

int symbolic_analysis(int n, const int* rp, const int* ci, SlSymbolic& Y)
{
  Y.n = n;
  Dissector D;
  D.n = n;
  D.rp = rp;
  D.ci = ci;
  D.removed.assign((size_t)n, 0);
  D.mark_store.assign((size_t)n, 0);
  D.level_store.assign((size_t)n, 0);
  D.mark = D.mark_store.data();
  D.level = D.level_store.data();
  // ids 1 .. 4 n + 64 belong to the top-level dissector, sub-dissectors take theirs from the counter (4 |set| + 64 each; the sets of one
  // level of the recursion are disjoint, PAR_LEVELS levels fork: < 2^31 for every n this solver takes)
  std::atomic<int> next_stamp(4 * n + 65);
  constexpr int PAR_LEVELS = 3;
  if(sl_host_threads(n) > 1 && (int64_t)n * 4 * (PAR_LEVELS + 2) < (int64_t)2000000000) {
    D.next_stamp = &next_stamp;
    D.par_depth = PAR_LEVELS;
  }
  const int dense_thr = sl_dense_threshold(n, rp);
  std::vector<int> keep, dense;
  for(int v = 0; v < n; ++v) {
    bool has_diag = false;
    for(int q = rp[v]; q < rp[v + 1]; ++q) {
      if(ci[q] < 0 || ci[q] >= n) return HIOPAMD_ERR_ARG;
      if(q > rp[v] && ci[q] <= ci[q - 1]) return HIOPAMD_ERR_ARG;   // columns must ascend inside a row
      has_diag = has_diag || ci[q] == v;
    }
    if(!has_diag) return HIOPAMD_ERR_STATE;   // a structurally zero diagonal entry: not this solver's matrix
    if(rp[v + 1] - rp[v] - 1 > dense_thr) {
      D.removed[(size_t)v] = 1;
      dense.push_back(v);
    } else keep.push_back(v);
  }
  // the dissector only walks vertices whose mark equals the id of the current set: dense rows never get one
  SlStopwatch sw;
  D.order.reserve((size_t)n);
  D.dissect(std::move(keep));
  sw.lap("nested dissection");
  const int n_sparse_cols = (int)D.order.size();
  if(!dense.empty()) D.emit_supernodes(dense);
  if((int)D.order.size() != n) return HIOPAMD_ERR_STATE;
  Y.perm = D.order;
  Y.iperm.assign((size_t)n, -1);
  for(int k = 0; k < n; ++k) Y.iperm[(size_t)Y.perm[(size_t)k]] = k;
  Y.ns = (int)D.sn_first.size();
  Y.sfirst = D.sn_first;
  Y.snc = D.sn_nc;
  // ---- row structures, in elimination order
  std::vector<int> sn_of((size_t)n);
  for(int s = 0; s < Y.ns; ++s)
    for(int k = 0; k < Y.snc[(size_t)s]; ++k) sn_of[(size_t)(Y.sfirst[(size_t)s] + k)] = s;
  std::vector<std::vector<int>> rows((size_t)Y.ns), children((size_t)Y.ns);
  Y.sparent.assign((size_t)Y.ns, -1);
  std::vector<int> seen((size_t)n, -1);
  for(int s = 0; s < Y.ns; ++s) {
    const int first = Y.sfirst[(size_t)s], last = first + Y.snc[(size_t)s] - 1;
    std::vector<int>& R = rows[(size_t)s];
    for(int k = first; k <= last; ++k) {
      const int v = Y.perm[(size_t)k];
      for(int q = rp[v]; q < rp[v + 1]; ++q) {
        const int j = Y.iperm[(size_t)ci[q]];
        if(j > last && seen[(size_t)j] != s) {
          seen[(size_t)j] = s;
          R.push_back(j);
        }
      }
    }
    for(int c : children[(size_t)s])
      for(int j : rows[(size_t)c])
        if(j > last && seen[(size_t)j] != s) {
          seen[(size_t)j] = s;
          R.push_back(j);
        }
    std::sort(R.begin(), R.end());
    if(!R.empty()) {
      const int p = sn_of[(size_t)R[0]];
      Y.sparent[(size_t)s] = p;
      children[(size_t)p].push_back(s);
    }
    (void)n_sparse_cols;
  }
  sw.lap("row structures");
  // ---- sparse fronts / root
  Y.sroot.assign((size_t)Y.ns, 0);
  Y.slevel.assign((size_t)Y.ns, 0);
  for(int s = 0; s < Y.ns; ++s) {
    const int f = Y.snc[(size_t)s] + (int)rows[(size_t)s].size();
    bool root = f > SL_T || D.removed[(size_t)Y.perm[(size_t)Y.sfirst[(size_t)s]]] != 0;
    int lev = 0;
    for(int c : children[(size_t)s]) {
      root = root || Y.sroot[(size_t)c] != 0;
      lev = std::max(lev, Y.slevel[(size_t)c] + 1);
    }
    Y.sroot[(size_t)s] = root ? 1 : 0;
    Y.slevel[(size_t)s] = lev;
  }
  Y.root_pos.assign((size_t)n, -1);
  Y.root_cols.clear();
  Y.nlevels = 0;
  Y.nnzL = 0;
  for(int s = 0; s < Y.ns; ++s) {
    if(Y.sroot[(size_t)s]) {
      for(int k = 0; k < Y.snc[(size_t)s]; ++k) {
        Y.root_pos[(size_t)(Y.sfirst[(size_t)s] + k)] = (int)Y.root_cols.size();
        Y.root_cols.push_back(Y.sfirst[(size_t)s] + k);
      }
    } else {
      Y.nlevels = std::max(Y.nlevels, Y.slevel[(size_t)s] + 1);
      const int64_t nc = Y.snc[(size_t)s], f = nc + (int64_t)rows[(size_t)s].size();
      Y.nnzL += nc * f - nc * (nc - 1) / 2;
    }
  }
  Y.r = (int)Y.root_cols.size();
  Y.nnzL += (int64_t)Y.r * (Y.r + 1) / 2;
  // a root column must not appear BELOW a sparse column in the order of elimination of its own tree — guaranteed by construction (the root
  // is upward closed); what can happen is a sparse supernode whose rows are partly root columns: those rows go to the root's gather plan
  Y.srow_ptr.assign((size_t)Y.ns + 1, 0);
  for(int s = 0; s < Y.ns; ++s) Y.srow_ptr[(size_t)s + 1] = Y.srow_ptr[(size_t)s] + (int)rows[(size_t)s].size();
  Y.srows.resize((size_t)Y.srow_ptr[(size_t)Y.ns]);
  for(int s = 0; s < Y.ns; ++s) std::copy(rows[(size_t)s].begin(), rows[(size_t)s].end(), Y.srows.begin() + Y.srow_ptr[(size_t)s]);
  sw.lap("levels, root, copies");
  if(Y.r > SL_ROOT_MAX) return HIOPAMD_ERR_STATE;
  return HIOPAMD_OK;
}

// ---- gather plans -----------------------------------------------------------------------------------------------------------
// A plan = runs of sources per destination.  Source index >= 0: entry of the CSR value array of M; < 0: entry -(idx + 1) of the pool of
// update matrices (matrix plans) or of update vectors (vector plans).
struct SlPlanHost {
  std::vector<int> run_dest;      // destination (position inside the front: i * f + j, lower triangle; root: i * r + j, upper)
  std::vector<int64_t> run_ptr;   // runs + 1
  std::vector<int64_t> src;
  std::vector<int64_t> front_run; // fronts + 1: runs of front q are front_run[q] .. front_run[q + 1]
};
struct SlHostLayout {
  // sparse fronts in level order
  std::vector<int> fs;                  // front q -> supernode
  std::vector<int> level_ptr;           // levels + 1
  std::vector<int> f_nc, f_nr;          // per front
  std::vector<int64_t> f_lofs, f_uofs;  // L panel / update matrix offsets (doubles)
  std::vector<int64_t> f_vofs;          // update vector offsets
  std::vector<int64_t> f_iofs;          // offset into fidx
  std::vector<int> fidx;                // per front: ORIGINAL indices of its f variables (columns, then rows)
  std::vector<int> fmax_level;          // largest front per level
  std::vector<int> ncsum_level;         // pivot columns per level (which factor kernel a level gets)
  int64_t lsize = 0, usize = 0, vsize = 0;
  SlPlanHost mat, vec;                  // fronts
  SlPlanHost rmat, rvec;                // root (one "front")
  std::vector<int> root_old;            // root position -> original index
};

struct Contribution {
  int64_t dest;
  int64_t src;
};
void sort_contributions(std::vector<Contribution>& C)
{
  std::stable_sort(C.begin(), C.end(), [](const Contribution& a, const Contribution& b) { return a.dest < b.dest; });
}
// threads of the host-side analysis (round 6: it was single-threaded, 0.67 s at n = 1e6): the hardware's, at most 16; 1 for small patterns
int sl_host_threads(int64_t work_items)
{
  if(work_items < 100000) return 1;
  // HIOPAMD_HOST_THREADS caps it (1: the sequential analysis; the result is the same by construction — tests/test_sparse_ldl_plan.py checks)
  if(const char* e = std::getenv("HIOPAMD_HOST_THREADS")) return std::max(1, std::min(std::atoi(e), 64));
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(hw ? hw : 1u, 16u));
}
// f(q) for q in [0, count), dealt to the threads in contiguous blocks of about equal WEIGHT (weight(q): e.g. the entries to sort)
template <class W, class F>
void sl_parallel_for(int count, int threads, W weight, F f)
{
  if(threads <= 1 || count < 2 * threads) {
    for(int q = 0; q < count; ++q) f(q);
    return;
  }
  double total = 0.0;
  for(int q = 0; q < count; ++q) total += (double)weight(q) + 1.0;
  std::vector<int> cut((size_t)threads + 1, count);
  cut[0] = 0;
  double acc = 0.0;
  int t = 1;
  for(int q = 0; q < count && t < threads; ++q) {
    acc += (double)weight(q) + 1.0;
    if(acc >= total * t / threads) cut[(size_t)t++] = q + 1;
  }
  std::vector<std::thread> pool;
  for(int w = 1; w < threads; ++w)
    pool.emplace_back([&, w]() {
      for(int q = cut[(size_t)w]; q < cut[(size_t)w + 1]; ++q) f(q);
    });
  for(int q = cut[0]; q < cut[1]; ++q) f(q);
  for(auto& th : pool) th.join();
}
// (C sorted by destination: sort_contributions)
void build_runs(std::vector<Contribution>& C, SlPlanHost& P)
{
  for(size_t q = 0; q < C.size(); ++q) {
    if(q == 0 || C[q].dest != C[q - 1].dest) {
      P.run_dest.push_back((int)C[q].dest);
      P.run_ptr.push_back((int64_t)P.src.size());
    }
    P.src.push_back(C[q].src);
  }
}

int build_layout(int n, const int* rp, const int* ci, const SlSymbolic& Y, SlHostLayout& H)
{
  SlStopwatch sw;
  // fronts by level
  std::vector<int> order;
  for(int s = 0; s < Y.ns; ++s)
    if(!Y.sroot[(size_t)s]) order.push_back(s);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return Y.slevel[(size_t)a] < Y.slevel[(size_t)b]; });
  H.fs = order;
  const int nf = (int)order.size();
  std::vector<int> front_of((size_t)Y.ns, -1);
  H.level_ptr.assign((size_t)Y.nlevels + 1, 0);
  H.fmax_level.assign((size_t)std::max(Y.nlevels, 1), 0);
  H.ncsum_level.assign((size_t)std::max(Y.nlevels, 1), 0);
  for(int q = 0; q < nf; ++q) {
    front_of[(size_t)order[(size_t)q]] = q;
    H.level_ptr[(size_t)Y.slevel[(size_t)order[(size_t)q]] + 1] += 1;
  }
  for(int l = 0; l < Y.nlevels; ++l) H.level_ptr[(size_t)l + 1] += H.level_ptr[(size_t)l];
  H.f_nc.resize((size_t)nf); H.f_nr.resize((size_t)nf); H.f_lofs.resize((size_t)nf); H.f_uofs.resize((size_t)nf);
  H.f_vofs.resize((size_t)nf); H.f_iofs.resize((size_t)nf);
  for(int q = 0; q < nf; ++q) {
    const int s = order[(size_t)q];
    const int nc = Y.snc[(size_t)s], nr = Y.srow_ptr[(size_t)s + 1] - Y.srow_ptr[(size_t)s];
    H.f_nc[(size_t)q] = nc; H.f_nr[(size_t)q] = nr;
    H.f_lofs[(size_t)q] = H.lsize; H.lsize += (int64_t)(nc + nr) * nc;
    H.f_uofs[(size_t)q] = H.usize; H.usize += (int64_t)nr * nr;
    H.f_vofs[(size_t)q] = H.vsize; H.vsize += nr;
    H.f_iofs[(size_t)q] = (int64_t)H.fidx.size();
    for(int k = 0; k < nc; ++k) H.fidx.push_back(Y.perm[(size_t)(Y.sfirst[(size_t)s] + k)]);
    for(int t = Y.srow_ptr[(size_t)s]; t < Y.srow_ptr[(size_t)s + 1]; ++t) H.fidx.push_back(Y.perm[(size_t)Y.srows[(size_t)t]]);
    H.fmax_level[(size_t)Y.slevel[(size_t)s]] = std::max(H.fmax_level[(size_t)Y.slevel[(size_t)s]], nc + nr);
    H.ncsum_level[(size_t)Y.slevel[(size_t)s]] += nc;
  }
  if(H.usize > (int64_t)1 << 40) return HIOPAMD_ERR_STATE;
  H.root_old.resize((size_t)Y.r);
  for(int t = 0; t < Y.r; ++t) H.root_old[(size_t)t] = Y.perm[(size_t)Y.root_cols[(size_t)t]];
  // position of a new index inside a front: columns first, then the rows (both ascending): binary search in the row list
  auto pos_in_front = [&](int s, int j) -> int {
    const int first = Y.sfirst[(size_t)s], nc = Y.snc[(size_t)s];
    if(j >= first && j < first + nc) return j - first;
    const int* b = Y.srows.data() + Y.srow_ptr[(size_t)s];
    const int* e = Y.srows.data() + Y.srow_ptr[(size_t)s + 1];
    const int* it = std::lower_bound(b, e, j);
    if(it == e || *it != j) return -1;
    return nc + (int)(it - b);
  };
  std::vector<int> sn_of((size_t)n);
  for(int s = 0; s < Y.ns; ++s)
    for(int k = 0; k < Y.snc[(size_t)s]; ++k) sn_of[(size_t)(Y.sfirst[(size_t)s] + k)] = s;
  // contributions per sparse front / root: (1) entries of M whose EARLIER index (new numbering) is a column of the front,
  // (2) the update matrices of the children
  sw.lap("layout: offsets, index lists");
  std::vector<std::vector<Contribution>> Cm((size_t)nf), Cv((size_t)nf);
  std::vector<Contribution> Rm, Rv;
  for(int v = 0; v < n; ++v) {
    const int iv = Y.iperm[(size_t)v];
    for(int q = rp[v]; q < rp[v + 1]; ++q) {
      const int jw = Y.iperm[(size_t)ci[q]];
      if(jw > iv) continue;   // the lower triangle in the new numbering: row iv >= column jw (the symmetric twin is skipped)
      const int s = sn_of[(size_t)jw];
      if(Y.sroot[(size_t)s]) {
        const int a = Y.root_pos[(size_t)jw], b = Y.root_pos[(size_t)iv];
        if(a < 0 || b < 0) return HIOPAMD_ERR_STATE;   // (a root column's later neighbours are root columns: the root is upward closed)
        Rm.push_back({(int64_t)std::min(a, b) * Y.r + std::max(a, b), (int64_t)q});
      } else {
        const int fq = front_of[(size_t)s];
        const int f = H.f_nc[(size_t)fq] + H.f_nr[(size_t)fq];
        const int pi = pos_in_front(s, iv), pj = jw - Y.sfirst[(size_t)s];
        if(pi < 0) return HIOPAMD_ERR_STATE;
        Cm[(size_t)fq].push_back({(int64_t)pi * f + pj, (int64_t)q});
      }
    }
  }
  sw.lap("layout: matrix entries");
  for(int q = 0; q < nf; ++q) {   // child q -> its parent (front or root); children are visited in front order: a fixed order of additions
    const int s = order[(size_t)q];
    const int nr = H.f_nr[(size_t)q];
    if(nr == 0) continue;
    const int p = Y.sparent[(size_t)s];
    if(p < 0) return HIOPAMD_ERR_STATE;
    const int* rws = Y.srows.data() + Y.srow_ptr[(size_t)s];
    if(Y.sroot[(size_t)p]) {
      std::vector<int> rel((size_t)nr);
      for(int t = 0; t < nr; ++t) {
        rel[(size_t)t] = Y.root_pos[(size_t)rws[t]];
        if(rel[(size_t)t] < 0) return HIOPAMD_ERR_STATE;
      }
      for(int i = 0; i < nr; ++i) {
        for(int j = 0; j <= i; ++j)
          Rm.push_back({(int64_t)std::min(rel[(size_t)i], rel[(size_t)j]) * Y.r + std::max(rel[(size_t)i], rel[(size_t)j]),
                        -(H.f_uofs[(size_t)q] + (int64_t)i * nr + j) - 1});
        Rv.push_back({(int64_t)rel[(size_t)i], -(H.f_vofs[(size_t)q] + i) - 1});
      }
    } else {
      const int pq = front_of[(size_t)p];
      const int pf = H.f_nc[(size_t)pq] + H.f_nr[(size_t)pq];
      std::vector<int> rel((size_t)nr);
      for(int t = 0; t < nr; ++t) {
        rel[(size_t)t] = pos_in_front(p, rws[t]);
        if(rel[(size_t)t] < 0) return HIOPAMD_ERR_STATE;
      }
      for(int i = 0; i < nr; ++i) {
        for(int j = 0; j <= i; ++j) Cm[(size_t)pq].push_back({(int64_t)rel[(size_t)i] * pf + rel[(size_t)j], -(H.f_uofs[(size_t)q] + (int64_t)i * nr + j) - 1});
        Cv[(size_t)pq].push_back({(int64_t)rel[(size_t)i], -(H.f_vofs[(size_t)q] + i) - 1});
      }
    }
  }
  sw.lap("layout: children's updates");
  H.mat.front_run.assign(1, 0);
  H.vec.front_run.assign(1, 0);
  {
    // the sorts (by destination, stable: the order of the additions inside a destination is the order of the lists above) are independent
    // per front: all host threads; the runs are then appended front by front in one sequential pass
    int64_t ncontrib = 0;
    for(int q = 0; q < nf; ++q) ncontrib += (int64_t)Cm[(size_t)q].size() + (int64_t)Cv[(size_t)q].size();
    sl_parallel_for(nf, sl_host_threads(ncontrib), [&](int q) { return Cm[(size_t)q].size() + Cv[(size_t)q].size(); },
                    [&](int q) {
                      sort_contributions(Cm[(size_t)q]);
                      sort_contributions(Cv[(size_t)q]);
                    });
    int64_t nm = 0, nv = 0;
    for(int q = 0; q < nf; ++q) {
      nm += (int64_t)Cm[(size_t)q].size();
      nv += (int64_t)Cv[(size_t)q].size();
    }
    H.mat.src.reserve((size_t)nm); H.mat.run_dest.reserve((size_t)nm); H.mat.run_ptr.reserve((size_t)nm + 1);
    H.vec.src.reserve((size_t)nv); H.vec.run_dest.reserve((size_t)nv); H.vec.run_ptr.reserve((size_t)nv + 1);
  }
  for(int q = 0; q < nf; ++q) {
    build_runs(Cm[(size_t)q], H.mat);
    H.mat.front_run.push_back((int64_t)H.mat.run_dest.size());
    std::vector<Contribution>().swap(Cm[(size_t)q]);
    build_runs(Cv[(size_t)q], H.vec);
    H.vec.front_run.push_back((int64_t)H.vec.run_dest.size());
  }
  H.mat.run_ptr.push_back((int64_t)H.mat.src.size());
  H.vec.run_ptr.push_back((int64_t)H.vec.src.size());
  sort_contributions(Rm);
  sort_contributions(Rv);
  build_runs(Rm, H.rmat);
  H.rmat.run_ptr.push_back((int64_t)H.rmat.src.size());
  build_runs(Rv, H.rvec);
  H.rvec.run_ptr.push_back((int64_t)H.rvec.src.size());
  sw.lap("layout: runs");
  return HIOPAMD_OK;
}

// ---- device side ------------------------------------------------------------------------------------------------------------
struct SlPlanDev {
  int* run_dest = nullptr;
  int64_t* run_ptr = nullptr;
  int64_t* src = nullptr;
  int64_t* front_run = nullptr;
  int64_t nruns = 0;
};
template <class T>
int up(T** d, const std::vector<T>& h)
{
  *d = nullptr;
  HIOPAMD_CHECK(hipMalloc((void**)d, sizeof(T) * std::max<size_t>(h.size(), 1)));
  if(!h.empty()) HIOPAMD_CHECK(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return HIOPAMD_OK;
}
int up_plan(SlPlanDev& D, const SlPlanHost& H)
{
  RC(up(&D.run_dest, H.run_dest));
  RC(up(&D.run_ptr, H.run_ptr));
  RC(up(&D.src, H.src));
  RC(up(&D.front_run, H.front_run));
  D.nruns = (int64_t)H.run_dest.size();
  return HIOPAMD_OK;
}
void free_plan(SlPlanDev& D)
{
  (void)hipFree(D.run_dest); (void)hipFree(D.run_ptr); (void)hipFree(D.src); (void)hipFree(D.front_run);
}

// (vector plans have pool sources only: they pass the pool for `vals` as well — a literal nullptr there crashes the inliner of ROCm 7.2's clang)
__device__ __forceinline__ double sl_gather(const int64_t* __restrict__ src, int64_t b, int64_t e, const double* __restrict__ vals,
                                             const double* __restrict__ pool)
{
  double s = 0.0;
  for(int64_t q = b; q < e; ++q) {
    const int64_t i = src[q];
    s += (i >= 0) ? vals[i] : pool[-(i + 1)];
  }
  return s;
}

// one workgroup per front of a level: gather -> LDL^T of the pivot columns (lower triangle, right-looking, in LDS) -> L panel + update matrix
__global__ void sl_factor_level_kernel(int q0, const int* __restrict__ f_nc, const int* __restrict__ f_nr, const int64_t* __restrict__ f_lofs,
                                       const int64_t* __restrict__ f_uofs, const int* __restrict__ run_dest, const int64_t* __restrict__ run_ptr,
                                       const int64_t* __restrict__ srcs, const int64_t* __restrict__ front_run, const double* __restrict__ vals,
                                       double* __restrict__ upool, double* __restrict__ lpool, int* __restrict__ counts, int ldf)
{
  extern __shared__ double sl_lds[];
  const int q = q0 + blockIdx.x;
  const int nc = f_nc[q], nr = f_nr[q], f = nc + nr;
  const int tid = threadIdx.x, nt = blockDim.x;
  // The front's LOWER TRIANGLE, packed by rows: entry (i, j), j <= i, at i (i + 1) / 2 + j.  (Rounds 4-5 kept the full f x f square at an
  // odd pitch: 28 KB for the 58-row fronts of the banded n = 1e6 case, i.e. five fronts per CU — the leaf level, 32 768 fronts of ~15 us
  // of dependent pivot steps each, took 26 rounds of resident workgroups.  Half the LDS is twice the fronts in flight.)
  double* F = sl_lds;
  double* lk = F + (size_t)ldf * (ldf + 1) / 2;   // multipliers of the current pivot
  double* vk = lk + ldf;                          // the unscaled column of the current pivot
#define SL_TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
  for(int e = tid; e < f * (f + 1) / 2; e += nt) F[e] = 0.0;
  __syncthreads();
  for(int64_t t = front_run[q] + tid; t < front_run[q + 1]; t += nt) {
    const int d = run_dest[t];
    const int di = d / f, dj = d % f;
    F[(di >= dj) ? SL_TRI(di, dj) : SL_TRI(dj, di)] = sl_gather(srcs, run_ptr[t], run_ptr[t + 1], vals, upool);
  }
  __syncthreads();
  int nneg = 0, nzero = 0;
  for(int k = 0; k < nc; ++k) {
    const double d = F[SL_TRI(k, k)];
    const bool bad = !(fabs(d) >= 1e-14) || !isfinite(d);   // thresholds of the reference's dense solver class (hiopLinSolverSymDenseLapack.hpp:154-161)
    if(tid == 0) {
      nzero += bad ? 1 : 0;
      nneg += (!bad && d < 0.0) ? 1 : 0;
    }
    const double di = bad ? 0.0 : 1.0 / d;   // a zero pivot: the column is dropped (the verdict is "not factorisable" anyway)
    for(int i = k + 1 + tid; i < f; i += nt) {
      const double v = F[SL_TRI(i, k)];
      vk[i] = v;
      lk[i] = v * di;
    }
    __syncthreads();
    // F[i][j] -= lk[i] * vk[j],  f > i >= j > k: rows dealt to threads in interleaved pairs (row i and row f - 1 - (i - k - 1)) would balance
    // better; the plain row-cyclic form is enough for fronts of <= 128 rows
    // (rows dealt to threads, the threads of a row split its columns: no integer division per element — the element-per-thread form
    //  spent more time on e / m, e % m than on the update, and half its elements were above the diagonal)
    const int m = f - k - 1;   // rows below the pivot (0 at the last pivot of a front without rows: nothing to update, and no nt / m)
    if(m > 0) {
      const int tpr = (nt >= 2 * m) ? (nt / m < 8 ? nt / m : 8) : 1;   // threads per row
      const int rgrp = nt / tpr;                                       // rows in flight
      const int sub = tid % tpr, r0 = tid / tpr;
      if(r0 < rgrp)
        for(int ii = r0; ii < m; ii += rgrp) {
          const double li = lk[k + 1 + ii];
          double* Fr = F + SL_TRI(k + 1 + ii, k + 1);
          for(int jj = sub; jj <= ii; jj += tpr) Fr[jj] -= li * vk[k + 1 + jj];
        }
    }
    for(int i = k + 1 + tid; i < f; i += nt) F[SL_TRI(i, k)] = lk[i];
    __syncthreads();
  }
  if(tid == 0 && (nneg || nzero)) {
    if(nneg) atomicAdd(counts, nneg);
    if(nzero) atomicAdd(counts + 1, nzero);
  }
  // the L panel COLUMN by column (column k contiguous: L[k * f + i]): the solve sweeps walk a column with consecutive lanes.  (Row by
  // row, as first written, every column step of a sweep touched one 64-byte sector per row for 8 bytes of it: 8 x the panel's bytes from L2
  // per sweep — the leaf level of the n = 1e6 banded case took 159 / 150 of the forward / backward sweep's 275 / 241 us.)
  double* L = lpool + f_lofs[q];
  for(int e = tid; e < f * nc; e += nt) {
    const int k = e / f, i = e % f;
    L[e] = (i >= k) ? F[SL_TRI(i, k)] : 0.0;   // diagonal: d_k; below: L; above (inside the pivot block): unused
  }
  double* U = upool + f_uofs[q];
  for(int e = tid; e < nr * nr; e += nt) {
    const int i = e / nr, j = e % nr;
    U[e] = (j <= i) ? F[SL_TRI(nc + i, nc + j)] : 0.0;
  }
#undef SL_TRI
}

__global__ __launch_bounds__(kBlock) void sl_root_gather_kernel(int64_t nruns, int r, int64_t lda, const int* __restrict__ run_dest,
                                                                const int64_t* __restrict__ run_ptr, const int64_t* __restrict__ srcs,
                                                                const double* __restrict__ vals, const double* __restrict__ upool,
                                                                double* __restrict__ M)
{
  for(int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < nruns; t += (int64_t)gridDim.x * kBlock) {
    const int d = run_dest[t];
    M[(int64_t)(d / r) * lda + (d % r)] = sl_gather(srcs, run_ptr[t], run_ptr[t + 1], vals, upool);
  }
}

// ---- one wave per front (the sweeps; round 6: the factorisation of the levels with small fronts and many pivots) ------------------------
// Measured and NOT kept in round 6 (profiles/r06_probes/call15-17_*): several levels in ONE launch — a workgroup per subtree, a wave per
// front, a workgroup barrier between the levels.  With device-scope fences at the hand-over a solve took 8.7 ms instead of 0.34 (every
// fence writes back and invalidates the XCD's L2); with the workgroup-scope hand-over that is sufficient (the waves of a workgroup share
// their CU's L1) the five widest levels took 227 us in one launch against 135 us in five (a workgroup idles through the upper steps of its
// subtree while the wide levels need every CU's full occupancy), and the narrow levels cost 4.3 / 6 us per step inside a workgroup against
// 4.7 / 7 us per launch: a level of few fronts is a chain of ~8 dependent memory round trips per front, not launch overhead, and the
// look-up of the front list adds two more.  Requesting every independent load of a front up front (48 loads in flight, 144-224
// registers) made the wide levels slower (occupancy) and the narrow ones no faster.
struct SlFronts {
  const int *nc, *nr;
  const int64_t *lofs, *uofs, *vofs, *iofs;
  const int* idx;
};
struct SlRuns {
  const int* dest;
  const int64_t *ptr, *src, *front;
};
// value of lane `src` (wave-uniform) in every lane: two v_readlane_b32 — the pivot-by-pivot chains of the sweeps broadcast one value per
// step, and __shfl is a ds_bpermute (an LDS round trip on the critical path of every step)
__device__ __forceinline__ double sl_bcast(double v, int src)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// forward: w = [b(cols); 0] + children; L11 y = w_c; w_r -= L21 y; y -> b(cols); w_r -> vector pool.  One wave per front; w: SL_T doubles
// of LDS owned by the wave (LDS operations of one wave are executed in order: no barrier between its phases).
__device__ __forceinline__ void sl_fwd_front(int q, int lane, double* __restrict__ w, const SlFronts F, const SlRuns R,
                                             const double* __restrict__ lpool, double* __restrict__ vpool, double* __restrict__ b)
{
  const int nc = F.nc[q], nr = F.nr[q], f = nc + nr;
  const int* idx = F.idx + F.iofs[q];
  for(int i = lane; i < f; i += 64) w[i] = (i < nc) ? b[idx[i]] : 0.0;
  __builtin_amdgcn_wave_barrier();
  for(int64_t t = R.front[q] + lane; t < R.front[q + 1]; t += 64) w[R.dest[t]] += sl_gather(R.src, R.ptr[t], R.ptr[t + 1], vpool, vpool);
  __builtin_amdgcn_wave_barrier();
  const double* L = lpool + F.lofs[q];
  // the column sweep in REGISTERS: rows lane and lane + 64 of the front's vector live in this lane (f <= 128), the pivot entry of a step is
  // broadcast from the first register (nc <= 48 < 64) — no LDS round trip and no barrier per column
  double w0 = (lane < f) ? w[lane] : 0.0, w1 = (lane + 64 < f) ? w[lane + 64] : 0.0;
  // (sixteen columns' entries are requested together: the loop itself is a chain of dependent steps, and one load per step would put a
  //  memory round trip into each of them.  Requesting ALL of a front's columns — and its index list and plan entries — before anything is
  //  used was measured in round 6: 48 loads in flight cost 144 registers, the leaf level of the banded n = 1e6 pattern, which is bound by
  //  HBM and needs the occupancy, went from 68 to 73 us and level 1 from 32 to 44, and the narrow levels did not get faster at all.)
  const bool two = f > 64;   // uniform
  for(int k0 = 0; k0 < nc; k0 += 16) {
    double l0[16], l1[16];
#pragma unroll
    for(int u = 0; u < 16; ++u) {
      const int k = k0 + u;
      l0[u] = (k < nc && lane > k && lane < f) ? L[(int64_t)k * f + lane] : 0.0;
      l1[u] = (two && k < nc && lane + 64 < f) ? L[(int64_t)k * f + lane + 64] : 0.0;
    }
#pragma unroll
    for(int u = 0; u < 16; ++u) {
      const int k = k0 + u;
      if(k < nc) {   // uniform
        const double yk = sl_bcast(w0, k);
        w0 -= l0[u] * yk;      // (zero for the rows at or above the pivot)
        if(two) w1 -= l1[u] * yk;
      }
    }
  }
  if(lane < f) {
    if(lane < nc) b[idx[lane]] = w0;
    else vpool[F.vofs[q] + (lane - nc)] = w0;
  }
  if(lane + 64 < f) {
    if(lane + 64 < nc) b[idx[lane + 64]] = w1;
    else vpool[F.vofs[q] + (lane + 64 - nc)] = w1;
  }
  __builtin_amdgcn_wave_barrier();   // (the wave's next front reuses w)
}
__global__ __launch_bounds__(64) void sl_fwd_level_kernel(int q0, const SlFronts F, const SlRuns R, const double* __restrict__ lpool,
                                                          double* __restrict__ vpool, double* __restrict__ b)
{
  __shared__ double w[SL_T];
  sl_fwd_front(q0 + blockIdx.x, threadIdx.x, w, F, R, lpool, vpool, b);
}

__global__ __launch_bounds__(kBlock) void sl_root_rhs_kernel(int r, const int* __restrict__ root_old, int64_t nruns, const int* __restrict__ run_dest,
                                                             const int64_t* __restrict__ run_ptr, const int64_t* __restrict__ srcs,
                                                             const double* __restrict__ vpool, const double* __restrict__ b, double* __restrict__ xr,
                                                             int phase)
{
  // phase 0: xr = b(root); phase 1: xr[dest] += children (every destination once: a run per destination)
  const int64_t n = phase == 0 ? (int64_t)r : nruns;
  for(int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (int64_t)gridDim.x * kBlock) {
    if(phase == 0) xr[t] = b[root_old[t]];
    else xr[run_dest[t]] += sl_gather(srcs, run_ptr[t], run_ptr[t + 1], vpool, vpool);
  }
}
__global__ __launch_bounds__(kBlock) void sl_root_scatter_kernel(int r, const int* __restrict__ root_old, const double* __restrict__ xr,
                                                                 double* __restrict__ x)
{
  for(int t = blockIdx.x * kBlock + threadIdx.x; t < r; t += gridDim.x * kBlock) x[root_old[t]] = xr[t];
}

// backward: z = y ./ d; L11^T x_c = z - L21^T x_r; one wave per front
__device__ __forceinline__ void sl_bwd_front(int q, int lane, const SlFronts F, const double* __restrict__ lpool, double* __restrict__ x)
{
  const int nc = F.nc[q], nr = F.nr[q], f = nc + nr;
  const int* idx = F.idx + F.iofs[q];
  const double* L = lpool + F.lofs[q];
  // Rows lane and lane + 64 of the front's vector in registers (f <= 128; the pivot rows, nc <= 48, in the first one).  Round 6: the sweep
  // runs in AXPY form over the ROWS of L, from the last row up: when x_i is final (every row below it has been applied) it is broadcast
  // and lane k < min(i, nc) subtracts L_ik x_i from its own entry — no reduction.  (Rounds 4-5 took the columns from the last pivot down:
  // a dot product over the lanes per pivot, six dependent shuffles + a broadcast each, 140 us for the 32 768 leaf fronts of the banded
  // n = 1e6 case against 62 us for the forward sweep, which always was an axpy.)  Lane k reads ITS OWN column k of the panel (contiguous,
  // column-major storage) at row i: sixteen rows' entries are requested together.
  double w0 = 0.0, w1 = 0.0;
  if(lane < f) {
    const double v = x[idx[lane]];
    w0 = (lane < nc) ? v / L[(int64_t)lane * f + lane] : v;
  }
  if(lane + 64 < f) w1 = x[idx[lane + 64]];
  const double* Lk = L + (int64_t)((lane < nc) ? lane : 0) * f;   // this lane's column (lanes >= nc hold no pivot: their entries are masked)
  for(int i0 = f - 1; i0 >= 1; i0 -= 16) {
    double lv[16];
#pragma unroll
    for(int u = 0; u < 16; ++u) {
      const int i = i0 - u;
      lv[u] = (i >= 1 && lane < nc && i > lane) ? Lk[i] : 0.0;
    }
#pragma unroll
    for(int u = 0; u < 16; ++u) {
      const int i = i0 - u;
      if(i >= 1) {   // uniform
        const double xi = (i < 64) ? sl_bcast(w0, i) : sl_bcast(w1, i - 64);
        w0 -= lv[u] * xi;      // (zero for the rows at or below the lane's pivot and for the non-pivot lanes)
      }
    }
  }
  if(lane < nc) x[idx[lane]] = w0;
}
__global__ __launch_bounds__(64) void sl_bwd_level_kernel(int q0, const SlFronts F, const double* __restrict__ lpool, double* __restrict__ x)
{
  sl_bwd_front(q0 + blockIdx.x, threadIdx.x, F, lpool, x);
}

// The factorisation of a front of at most FMAX rows by ONE wave with the front in REGISTERS: lane i holds row i of the lower triangle
// (a[j] = F_ij, j <= i), the pivot loop is unrolled so that every register index is static.  Pivot k: its column — entry a[k] of every
// lane — goes to 64 doubles of LDS, every lane reads it back as broadcasts (all lanes one address: no bank conflicts) and updates its row
// with one fused multiply-add per entry: no barrier, no LDS traffic for the front itself, no integer arithmetic.  The LDS form of rounds
// 4-5 (sl_factor_level_kernel, still used above FMAX rows) spent two workgroup barriers and three dependent LDS phases per pivot and held
// 14 KB of LDS per 58-row front: the leaf level of the banded n = 1e6 pattern (32 768 fronts of ~30 pivots) took 450 of the
// factorisation's 610 us.  Fw: staging for the gather and for the update matrix (f (f + 1) / 2 doubles), col: 64 doubles; both the wave's own.
constexpr int SL_REGS_F = 40;
#ifndef HIOPAMD_SL_BCAST_SPLIT
#define HIOPAMD_SL_BCAST_SPLIT 1
#endif
#define SL_TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
template <int FMAX>
__device__ __forceinline__ void sl_factor_front_regs(int q, int lane, double* __restrict__ Fw, double* __restrict__ col, const SlFronts F,
                                                     const SlRuns R, const double* __restrict__ vals, double* __restrict__ upool,
                                                     double* __restrict__ lpool, int* __restrict__ counts)
{
  constexpr int KMAX = FMAX < SL_LEAF ? FMAX : SL_LEAF;
  const int nc = F.nc[q], nr = F.nr[q], f = nc + nr;
  const int ntri = f * (f + 1) / 2;
  for(int e = lane; e < ntri; e += 64) Fw[e] = 0.0;
  __builtin_amdgcn_wave_barrier();
  const float rf = 1.0f / (float)f;   // d / f for d < 4096, f <= 64: (d + 0.5) / f is at least 0.5 / 64 away from an integer, the float error is < 1e-5
  for(int64_t t = R.front[q] + lane; t < R.front[q + 1]; t += 64) {
    const int d = R.dest[t];
    const int di = (int)(((float)d + 0.5f) * rf), dj = d - di * f;
    Fw[(di >= dj) ? SL_TRI(di, dj) : SL_TRI(dj, di)] = sl_gather(R.src, R.ptr[t], R.ptr[t + 1], vals, upool);
  }
  __builtin_amdgcn_wave_barrier();
  double a[FMAX];
  const int rowbase = lane * (lane + 1) / 2;
#pragma unroll
  for(int j = 0; j < FMAX; ++j) a[j] = (j <= lane && lane < f) ? Fw[rowbase + j] : 0.0;
  int nneg = 0, nzero = 0;
#pragma unroll
  for(int k = 0; k < KMAX; ++k) {
    if(k < nc) {   // uniform
      const double d = __shfl(a[k], k, 64);
      const bool bad = !(fabs(d) >= 1e-14) || !isfinite(d);   // thresholds of the reference's dense solver class (hiopLinSolverSymDenseLapack.hpp:154-161)
      nzero += bad ? 1 : 0;
      nneg += (!bad && d < 0.0) ? 1 : 0;
      const double di = bad ? 0.0 : 1.0 / d;   // a zero pivot: the column is dropped (the verdict is "not factorisable" anyway)
      col[lane] = a[k];                        // column k, unscaled (zero in the lanes above the pivot and in the lanes behind the front)
      __builtin_amdgcn_wave_barrier();
      const double ak = a[k];
      const double l = (lane > k) ? ak * di : 0.0;
      // (the entries right of the diagonal get the symmetric value: never stored.)  The broadcast of column k is what bounds this loop:
      // through LDS it moves 64 x 8 bytes per entry into the wave's registers, and the CU's waves share one 128-byte-per-clock LDS
      // pipe; every other pair of entries therefore comes as a wave-uniform value from the vector unit (two v_readlane_b32) instead.
#pragma unroll
      for(int j = k + 1; j < FMAX; ++j) {
        if(HIOPAMD_SL_BCAST_SPLIT == 2 || (HIOPAMD_SL_BCAST_SPLIT == 1 && ((j >> 1) & 1))) a[j] -= l * sl_bcast(ak, j);
        else a[j] -= l * col[j];
      }
      if(lane > k) a[k] = l;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if(lane == 0 && (nneg || nzero)) {
    if(nneg) atomicAdd(counts, nneg);
    if(nzero) atomicAdd(counts + 1, nzero);
  }
  // the L panel column by column (column k contiguous: L[k * f + i], diagonal: d_k, above it inside the pivot block: zero)
  double* L = lpool + F.lofs[q];
#pragma unroll
  for(int k = 0; k < KMAX; ++k)
    if(k < nc && lane < f) L[(int64_t)k * f + lane] = (lane >= k) ? a[k] : 0.0;
  // the update matrix: rows / columns behind the pivots, row-major nr x nr with zeros right of the diagonal — through the staging triangle
  if(nr > 0) {
#pragma unroll
    for(int c = 0; c < FMAX; ++c)
      if(c >= nc && lane >= c && lane < f) Fw[SL_TRI(lane - nc, c - nc)] = a[c];
    __builtin_amdgcn_wave_barrier();
    double* U = upool + F.uofs[q];
    const float rn = 1.0f / (float)nr;
    for(int e = lane; e < nr * nr; e += 64) {
      const int i = (int)(((float)e + 0.5f) * rn), j = e - i * nr;
      U[e] = (j <= i) ? Fw[SL_TRI(i, j)] : 0.0;
    }
  }
  __builtin_amdgcn_wave_barrier();   // (the wave's next front reuses Fw / col)
}
#undef SL_TRI
template <int FMAX>
__global__ __launch_bounds__(64) void sl_factor_regs_kernel(int q0, const SlFronts F, const SlRuns R, const double* __restrict__ vals,
                                                            double* __restrict__ upool, double* __restrict__ lpool, int* __restrict__ counts)
{
  extern __shared__ double sl_lds[];   // the level's largest triangle + 64
  sl_factor_front_regs<FMAX>(q0 + blockIdx.x, threadIdx.x, sl_lds + 64, sl_lds, F, R, vals, upool, lpool, counts);
}

}  // namespace

struct hiopamd_sparse_ldl {
  hiopamd_ctx* ctx = nullptr;
  int n = 0;
  SlSymbolic Y;
  SlHostLayout H;
  int nf = 0;
  int *f_nc = nullptr, *f_nr = nullptr, *fidx = nullptr, *root_old = nullptr, *counts = nullptr;
  int64_t *f_lofs = nullptr, *f_uofs = nullptr, *f_vofs = nullptr, *f_iofs = nullptr;
  SlPlanDev mat, vec, rmat, rvec;
  bool reg_fronts = true;   // levels of small fronts with many pivots: one wave per front, the front in registers (HIOPAMD_SL_REGS=0: the LDS kernel everywhere)
  double *lpool = nullptr, *upool = nullptr, *vpool = nullptr, *xr = nullptr;
  hiopamd_linsolver* root = nullptr;
  int n_neg = 0, n_zero = 0;
  bool factored = false;
};

static SlFronts sl_fronts(const hiopamd_sparse_ldl* s) { return {s->f_nc, s->f_nr, s->f_lofs, s->f_uofs, s->f_vofs, s->f_iofs, s->fidx}; }

extern "C" {

int hiopamd_sparse_ldl_destroy(hiopamd_sparse_ldl* s)
{
  if(!s) return HIOPAMD_OK;
  if(s->ctx) (void)hipStreamSynchronize(s->ctx->stream);
  if(s->root) hiopamd_linsolver_destroy(s->root);
  (void)hipFree(s->f_nc); (void)hipFree(s->f_nr); (void)hipFree(s->fidx); (void)hipFree(s->root_old); (void)hipFree(s->counts);
  (void)hipFree(s->f_lofs); (void)hipFree(s->f_uofs); (void)hipFree(s->f_vofs); (void)hipFree(s->f_iofs);
  free_plan(s->mat); free_plan(s->vec); free_plan(s->rmat); free_plan(s->rvec);
  (void)hipFree(s->lpool); (void)hipFree(s->upool); (void)hipFree(s->vpool); (void)hipFree(s->xr);
  delete s;
  return HIOPAMD_OK;
}

// host only (no device needed): the analysis of a pattern.  info8 = {supernodes, sparse fronts, levels, root order, nnz(L) incl. the dense
// root's triangle, largest sparse front, dense rows set aside, 0}; perm_host (n, new -> old) may be NULL.  HIOPAMD_ERR_STATE: the root
// would exceed SL_ROOT_MAX or the diagonal is not structurally full.
int hiopamd_sparse_ldl_analyse(int n, const int* rowptr_host, const int* colidx_host, int64_t* info8_host, int* perm_host)
{
  if(n < 0 || !rowptr_host || (rowptr_host[n] > 0 && !colidx_host) || !info8_host) return HIOPAMD_ERR_ARG;
  SlSymbolic Y;
  const int rc = symbolic_analysis(n, rowptr_host, colidx_host, Y);
  int nfr = 0, fmax = 0;
  for(int s = 0; s < Y.ns; ++s)
    if(!Y.sroot.empty() && !Y.sroot[(size_t)s]) {
      nfr += 1;
      fmax = std::max(fmax, Y.snc[(size_t)s] + Y.srow_ptr[(size_t)s + 1] - Y.srow_ptr[(size_t)s]);
    }
  const int dense_thr = sl_dense_threshold(n, rowptr_host);
  int ndense = 0;
  for(int v = 0; v < n; ++v) ndense += (rowptr_host[v + 1] - rowptr_host[v] - 1 > dense_thr) ? 1 : 0;
  info8_host[0] = Y.ns; info8_host[1] = nfr; info8_host[2] = Y.nlevels; info8_host[3] = Y.r; info8_host[4] = Y.nnzL; info8_host[5] = fmax;
  info8_host[6] = ndense; info8_host[7] = 0;
  if(perm_host && (int)Y.perm.size() == n) std::copy(Y.perm.begin(), Y.perm.end(), perm_host);
  return rc;
}

// host only: the gather plans of the analysis, for an independent replay of the numeric phase (tests/test_sparse_ldl_plan.py does it in
// numpy).  Call with every pointer NULL to get the sizes in sizes16; then with buffers of those sizes.
//   sizes16 = {fronts, levels, len(fidx), mat runs, mat sources, vec runs, vec sources, root mat runs, root mat sources, root vec runs,
//              root vec sources, root order, L pool, update pool, vector pool, 0}
int hiopamd_sparse_ldl_plan(int n, const int* rowptr_host, const int* colidx_host, int64_t* sizes16, int* level_ptr, int* f_nc, int* f_nr,
                            int64_t* f_lofs, int64_t* f_uofs, int64_t* f_vofs, int64_t* f_iofs, int* fidx, int* mat_dest, int64_t* mat_ptr,
                            int64_t* mat_src, int64_t* mat_front, int* vec_dest, int64_t* vec_ptr, int64_t* vec_src, int64_t* vec_front,
                            int* rmat_dest, int64_t* rmat_ptr, int64_t* rmat_src, int* rvec_dest, int64_t* rvec_ptr, int64_t* rvec_src,
                            int* root_old)
{
  if(n < 0 || !rowptr_host || !sizes16) return HIOPAMD_ERR_ARG;
  SlSymbolic Y;
  RC(symbolic_analysis(n, rowptr_host, colidx_host, Y));
  SlHostLayout H;
  RC(build_layout(n, rowptr_host, colidx_host, Y, H));
  const int64_t sz[16] = {(int64_t)H.fs.size(), Y.nlevels, (int64_t)H.fidx.size(), (int64_t)H.mat.run_dest.size(), (int64_t)H.mat.src.size(),
                          (int64_t)H.vec.run_dest.size(), (int64_t)H.vec.src.size(), (int64_t)H.rmat.run_dest.size(), (int64_t)H.rmat.src.size(),
                          (int64_t)H.rvec.run_dest.size(), (int64_t)H.rvec.src.size(), Y.r, H.lsize, H.usize, H.vsize, 0};
  std::copy(sz, sz + 16, sizes16);
  auto cp = [](auto* dst, const auto& v) { if(dst) std::copy(v.begin(), v.end(), dst); };
  cp(level_ptr, H.level_ptr); cp(f_nc, H.f_nc); cp(f_nr, H.f_nr); cp(f_lofs, H.f_lofs); cp(f_uofs, H.f_uofs); cp(f_vofs, H.f_vofs);
  cp(f_iofs, H.f_iofs); cp(fidx, H.fidx);
  cp(mat_dest, H.mat.run_dest); cp(mat_ptr, H.mat.run_ptr); cp(mat_src, H.mat.src); cp(mat_front, H.mat.front_run);
  cp(vec_dest, H.vec.run_dest); cp(vec_ptr, H.vec.run_ptr); cp(vec_src, H.vec.src); cp(vec_front, H.vec.front_run);
  cp(rmat_dest, H.rmat.run_dest); cp(rmat_ptr, H.rmat.run_ptr); cp(rmat_src, H.rmat.src);
  cp(rvec_dest, H.rvec.run_dest); cp(rvec_ptr, H.rvec.run_ptr); cp(rvec_src, H.rvec.src);
  cp(root_old, H.root_old);
  return HIOPAMD_OK;
}

// pattern of the full symmetric matrix in CSR (host arrays, columns ascending inside a row, diagonal structurally full).
// HIOPAMD_ERR_STATE: not this solver's pattern (the dense root would exceed SL_ROOT_MAX) — nothing is created.
int hiopamd_sparse_ldl_create(hiopamd_sparse_ldl** out, hiopamd_ctx* ctx, int n, const int* rowptr_host, const int* colidx_host)
{
  if(!out || !ctx || n < 0 || !rowptr_host || (rowptr_host[n] > 0 && !colidx_host)) return HIOPAMD_ERR_ARG;
  *out = nullptr;
  auto* s = new hiopamd_sparse_ldl();
  s->ctx = ctx;
  s->n = n;
  int rc = symbolic_analysis(n, rowptr_host, colidx_host, s->Y);
  if(rc == HIOPAMD_OK) rc = build_layout(n, rowptr_host, colidx_host, s->Y, s->H);
  const SlHostLayout& H = s->H;
  s->nf = (int)H.fs.size();
  if(rc == HIOPAMD_OK) rc = up(&s->f_nc, H.f_nc);
  if(rc == HIOPAMD_OK) rc = up(&s->f_nr, H.f_nr);
  if(rc == HIOPAMD_OK) rc = up(&s->f_lofs, H.f_lofs);
  if(rc == HIOPAMD_OK) rc = up(&s->f_uofs, H.f_uofs);
  if(rc == HIOPAMD_OK) rc = up(&s->f_vofs, H.f_vofs);
  if(rc == HIOPAMD_OK) rc = up(&s->f_iofs, H.f_iofs);
  if(rc == HIOPAMD_OK) rc = up(&s->fidx, H.fidx);
  if(rc == HIOPAMD_OK) rc = up(&s->root_old, H.root_old);
  if(rc == HIOPAMD_OK) rc = up_plan(s->mat, H.mat);
  if(rc == HIOPAMD_OK) rc = up_plan(s->vec, H.vec);
  if(rc == HIOPAMD_OK) rc = up_plan(s->rmat, H.rmat);
  if(rc == HIOPAMD_OK) rc = up_plan(s->rvec, H.rvec);
  if(const char* e = std::getenv("HIOPAMD_SL_REGS")) s->reg_fronts = std::atoi(e) != 0;
  auto dalloc = [](double** p, int64_t cnt) { return hipMalloc((void**)p, sizeof(double) * (size_t)std::max<int64_t>(cnt, 1)) == hipSuccess ? HIOPAMD_OK : HIOPAMD_ERR_HIP; };
  if(rc == HIOPAMD_OK) rc = dalloc(&s->lpool, H.lsize);
  if(rc == HIOPAMD_OK) rc = dalloc(&s->upool, H.usize);
  if(rc == HIOPAMD_OK) rc = dalloc(&s->vpool, H.vsize);
  if(rc == HIOPAMD_OK) rc = dalloc(&s->xr, s->Y.r);
  if(rc == HIOPAMD_OK && hipMalloc((void**)&s->counts, 4 * sizeof(int)) != hipSuccess) rc = HIOPAMD_ERR_HIP;
  if(rc == HIOPAMD_OK && s->Y.r > 0) rc = hiopamd_linsolver_create(&s->root, ctx, s->Y.r);
  if(rc == HIOPAMD_OK && s->root) rc = hiopamd_linsolver_set_retry_copy(s->root, 0);   // (an expired wait: the root is gathered again, see factorize)
  // the host copies of the plans are no longer needed
  s->H.mat = SlPlanHost(); s->H.vec = SlPlanHost(); s->H.rmat = SlPlanHost(); s->H.rvec = SlPlanHost();
  std::vector<int>().swap(s->H.fidx);
  if(rc != HIOPAMD_OK) {
    hiopamd_sparse_ldl_destroy(s);
    return rc;
  }
  *out = s;
  return HIOPAMD_OK;
}

// Which factor kernel a level gets: fronts of at most 40 rows with many pivot columns — the leaves of a banded pattern — one wave each
// with the front in registers (sl_factor_regs_kernel; the banded n = 1e6 leaf level: 265 us against 391); fronts with few pivots —
// separators — and larger fronts in LDS by a workgroup each (the register form's cost follows the template's row count, not the pivots:
// 80 against 39 us on level 1 of that pattern).  A 64-row instantiation was measured as well: no gain on the 62-row leaves of bandwidth 7,
// 1.02 against 0.90 ms on the 20-column separators of the n = 2e5, bandwidth 20 pattern — not built.
static bool sl_level_in_registers(const hiopamd_sparse_ldl* s, int l)
{
  const SlHostLayout& H = s->H;
  const int cnt = H.level_ptr[(size_t)l + 1] - H.level_ptr[(size_t)l];
  return s->reg_fronts && cnt > 0 && H.fmax_level[(size_t)l] <= SL_REGS_F && (int64_t)H.ncsum_level[(size_t)l] >= 16 * (int64_t)cnt;
}

// {supernodes, sparse fronts, levels, root order, nnz(L), levels factored in registers, 0, 0}
int hiopamd_sparse_ldl_info(const hiopamd_sparse_ldl* s, int64_t* info8_host)
{
  if(!s || !info8_host) return HIOPAMD_ERR_ARG;
  info8_host[0] = s->Y.ns; info8_host[1] = s->nf; info8_host[2] = s->Y.nlevels; info8_host[3] = s->Y.r; info8_host[4] = s->Y.nnzL;
  info8_host[5] = 0;   // levels whose fronts are factored in registers (one wave per front; see hiopamd_sparse_ldl_factorize)
  for(int l = 0; l < s->Y.nlevels; ++l) info8_host[5] += sl_level_in_registers(s, l) ? 1 : 0;
  info8_host[6] = info8_host[7] = 0;
  return HIOPAMD_OK;
}

static int sl_gather_root(hiopamd_sparse_ldl* s, const double* vals)
{
  hiopamd_ctx* ctx = s->ctx;
  const int r = s->Y.r;
  double* M = hiopamd_linsolver_sys_matrix(s->root);
  HIOPAMD_CHECK(hipMemsetAsync(M, 0, sizeof(double) * (size_t)r * r, ctx->stream));
  if(s->rmat.nruns > 0)
    hipLaunchKernelGGL(sl_root_gather_kernel, dim3(grid_for(s->rmat.nruns)), dim3(kBlock), 0, ctx->stream, s->rmat.nruns, r,
                       (int64_t)r, s->rmat.run_dest, s->rmat.run_ptr, s->rmat.src, vals, s->upool, M);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

// csr_values: device array aligned with the pattern given at create.  n_neg / n_zero: pivots below -1e-14 / of magnitude below 1e-14
// (or non-finite).  n_zero > 0: no factorisation (solve refuses).  M is positive definite  <=>  n_neg == 0 && n_zero == 0.
int hiopamd_sparse_ldl_factorize(hiopamd_sparse_ldl* s, const double* vals, int* n_neg_host, int* n_zero_host)
{
  if(!s || (s->n > 0 && !vals) || !n_neg_host || !n_zero_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = s->ctx;
  s->factored = false;
  HIOPAMD_CHECK(hipMemsetAsync(s->counts, 0, 4 * sizeof(int), ctx->stream));
  const SlHostLayout& H = s->H;
  const SlFronts F = sl_fronts(s);
  const SlRuns R = {s->mat.run_dest, s->mat.run_ptr, s->mat.src, s->mat.front_run};
  for(int l = 0; l < s->Y.nlevels; ++l) {
    const int q0 = H.level_ptr[(size_t)l], cnt = H.level_ptr[(size_t)l + 1] - q0;
    if(cnt <= 0) continue;
    const int fmax = H.fmax_level[(size_t)l];
    if(sl_level_in_registers(s, l)) {
      const size_t lds = sizeof(double) * ((size_t)fmax * (fmax + 1) / 2 + 64);
      hipLaunchKernelGGL(sl_factor_regs_kernel<SL_REGS_F>, dim3((unsigned)cnt), dim3(64), lds, ctx->stream, q0, F, R, vals, s->upool, s->lpool,
                         s->counts);
      HIOPAMD_CHECK(hipGetLastError());
      continue;
    }
    const int ldf = fmax;   // (rows of the largest front of the level: the packed triangle has ldf (ldf + 1) / 2 entries)
    const size_t lds = sizeof(double) * ((size_t)ldf * (ldf + 1) / 2 + 2 * (size_t)ldf);
#ifndef HIOPAMD_SL_T128
#define HIOPAMD_SL_T128 48
#endif
    const int threads = fmax <= 16 ? 64 : (fmax <= HIOPAMD_SL_T128 ? 128 : 256);
    if(lds > 64 * 1024) HIOPAMD_CHECK(hipFuncSetAttribute((const void*)sl_factor_level_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sl_factor_level_kernel, dim3((unsigned)cnt), dim3(threads), lds, ctx->stream, q0, s->f_nc, s->f_nr, s->f_lofs, s->f_uofs,
                       s->mat.run_dest, s->mat.run_ptr, s->mat.src, s->mat.front_run, vals, s->upool, s->lpool, s->counts, ldf);
    HIOPAMD_CHECK(hipGetLastError());
  }
  int nneg_root = 0;
  bool root_singular = false;
  if(s->root) {
    RC(sl_gather_root(s, vals));
    int rc = hiopamd_linsolver_matrix_changed(s->root, &nneg_root);
    if(rc == HIOPAMD_ERR_TIMEOUT) {   // the dataflow factorisation gave up and left the matrix overwritten: gather again, stepwise kernels
      RC(sl_gather_root(s, vals));
      rc = hiopamd_linsolver_matrix_changed(s->root, &nneg_root);
    }
    RC(rc);
    if(nneg_root < 0) {
      root_singular = true;
      nneg_root = 0;
    }
  }
  int h[4] = {0, 0, 0, 0};
  HIOPAMD_CHECK(hipMemcpyAsync(h, s->counts, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  s->n_neg = h[0] + nneg_root;
  s->n_zero = h[1] + (root_singular ? 1 : 0);
  *n_neg_host = s->n_neg;
  *n_zero_host = s->n_zero;
  s->factored = s->n_zero == 0;
  return HIOPAMD_OK;
}

// x <- M^-1 x (device vector of length n) with the factors of the last hiopamd_sparse_ldl_factorize; no host round trip
int hiopamd_sparse_ldl_solve(hiopamd_sparse_ldl* s, double* x)
{
  if(!s || (s->n > 0 && !x)) return HIOPAMD_ERR_ARG;
  if(!s->factored) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = s->ctx;
  const SlHostLayout& H = s->H;
  const SlFronts F = sl_fronts(s);
  const SlRuns R = {s->vec.run_dest, s->vec.run_ptr, s->vec.src, s->vec.front_run};
  for(int l = 0; l < s->Y.nlevels; ++l) {
    const int q0 = H.level_ptr[(size_t)l], cnt = H.level_ptr[(size_t)l + 1] - q0;
    if(cnt <= 0) continue;
    hipLaunchKernelGGL(sl_fwd_level_kernel, dim3((unsigned)cnt), dim3(64), 0, ctx->stream, q0, F, R, s->lpool, s->vpool, x);
  }
  const int r = s->Y.r;
  if(s->root) {
    hipLaunchKernelGGL(sl_root_rhs_kernel, dim3(grid_for(r)), dim3(kBlock), 0, ctx->stream, r, s->root_old, s->rvec.nruns,
                       s->rvec.run_dest, s->rvec.run_ptr, s->rvec.src, s->vpool, x, s->xr, 0);
    if(s->rvec.nruns > 0)
      hipLaunchKernelGGL(sl_root_rhs_kernel, dim3(grid_for(s->rvec.nruns)), dim3(kBlock), 0, ctx->stream, r, s->root_old,
                         s->rvec.nruns, s->rvec.run_dest, s->rvec.run_ptr, s->rvec.src, s->vpool, x, s->xr, 1);
    HIOPAMD_CHECK(hipGetLastError());
    RC(hiopamd_linsolver_solve(s->root, s->xr, 1));
    hipLaunchKernelGGL(sl_root_scatter_kernel, dim3(grid_for(r)), dim3(kBlock), 0, ctx->stream, r, s->root_old, s->xr, x);
  }
  for(int l = s->Y.nlevels - 1; l >= 0; --l) {
    const int q0 = H.level_ptr[(size_t)l], cnt = H.level_ptr[(size_t)l + 1] - q0;
    if(cnt <= 0) continue;
    hipLaunchKernelGGL(sl_bwd_level_kernel, dim3((unsigned)cnt), dim3(64), 0, ctx->stream, q0, F, s->lpool, x);
  }
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

}  // extern "C"
