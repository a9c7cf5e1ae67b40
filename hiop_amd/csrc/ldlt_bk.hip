// Pivoted symmetric-indefinite factorisation (Bunch-Kaufman partial pivoting) on the device: the SAFE solver of the
// hiopLinSolverSymDense operator.
//
// reference: hiopLinSolverSymDenseMagmaBuKa (src/LinAlg/hiopLinSolverSymDenseMagma.cpp:120-250: magma_dsytrf + LINPACK-dsidi inertia +
// magma_dsytrs) and hiopLinSolverSymDenseLapack (hiopLinSolverSymDenseLapack.hpp:75-195: DSYTRF / DSYTRS) -- what the KKT classes
// switch to when the no-pivot factor misbehaves (hiopKKTLinSysMDS.cpp:408-430, hiopAlgFilterIPM.cpp:2400-2427).  The algorithm is
// LAPACK's DSYTRF, UPLO = 'L' (the row-major upper triangle of the KKT matrix IS the column-major lower one: a(i, j), i >= j, lives at
// A[j * lda + i]), i.e. panels of DLASYF: alpha = (1 + sqrt(17)) / 8, 1 x 1 and 2 x 2 pivots, the updated pivot columns kept in a panel
// W = L D (n x 64), one rank-64 update of the trailing matrix per panel.  (The CPU restatement the tests compare against, itself pinned on scipy's DSYTRF /
// DSYTRS, uses the same convention: tests/test_gpu_ldlt_bk.py.)
//
// Convention: every row interchange is applied to ALL previous columns of L when it happens, so P A P^T = L D L^T with ONE
// permutation; L is unit lower triangular and stored in the strict lower part, d on the diagonal, the off-diagonal entry of a
// 2 x 2 block in e[k] (the matrix position a(k+1, k) is zeroed).  Pivots (LAPACK's IPIV) and D are LAPACK's.
//
// Device mapping (round 5).  A panel of 64 columns is ONE launch (bk_panel_kernel: up to 16 workgroups of 512 threads, a row of the matrix
// owned by one thread for the whole panel); a column step is three PHASES and the decisions never leave the device:
//   phase A   updated column k into W(:, kw), |.| maximum below the diagonal per workgroup; every workgroup folds the partial maxima and
//             takes the first decision (1 x 1 without interchange, or "look at row imax")
//   phase B   (if asked for) updated column imax of the symmetric matrix into W(:, kw + 1), its off-diagonal maximum, the final
//             decision: pivot position kp, 1 x 1 or 2 x 2
//   phase C   interchange kk <-> kp (rows of L in the panel's previous columns, rows of W, the not yet updated column kk of A to position
//             kp), the column(s) of L and the block of D from W; workgroup 0 records IPIV / P; then the step's ONE grid barrier
// What crosses workgroups inside a step (partial maxima, the handful of scalars a decision needs) travels as tagged 8-byte granules that
// the deciding wave of every workgroup polls — see the comment in the kernel; the rows kk / kp of the columns IN FRONT of the panel are
// interchanged once per panel (bk_defer_swaps_kernel).  Everything a thread writes in a phase depends only on its own row plus the
// decision's scalars (tests/test_ldlt_bk_protocol.py replays the phases one thread at a time in random order, and the panel kernel as
// workgroups that run at their own pace between barriers).  Once per panel the host reads the panel's end (it depends on where 2 x 2
// pivots fell) and launches the trailing update: the stepwise LDL^T's fp64-MFMA rank-K kernel (ldlt.hip) with V = W, U = the L rows.
// Rounds 1-4 ran the three phases as three LAUNCHES per column (3 x 8192 launches at N = 8192: 173 ms;
// scripts/probes/retired/ldlt_bk_three_launch_kernels.hip.txt).  A solve is 2 x N / 256 block steps of two GEMVs each, on pre-inverted diagonal
// blocks, replayed as a HIP graph.  This is the exceptional path: the fast path stays the no-pivot dataflow factorisation.
#include "device_utils.hpp"

#include <climits>
#include <vector>

namespace hiopamd {

int ldlt_rankk_update(hiopamd_ctx* ctx, double* A, int64_t lda, int N, const double* V, int64_t ldv, int urow0, int K, int s, int kreal);   // ldlt.hip

constexpr int BK_NB = 64;                  // panel width (columns of W)
constexpr int BK_RPT = 4;                  // rows per thread of the column kernels
constexpr int BK_ROWS = kBlock * BK_RPT;   // rows per workgroup
constexpr double BK_ALPHA = 0.6403882032022076;   // (1 + sqrt(17)) / 8

struct BkState {
  int next_k;      // the column the factorisation is at
  int info;        // k + 1 of the first exactly zero pivot column (DSYTRF's INFO), 0 otherwise
  int kstep, kp;   // decision of the current step
  int imax;        // row of the largest off-diagonal entry of column k
  int need2;       // the second column kernel has work to do
  int use_c1;      // 1 x 1 pivot taken from row / column imax: the pivot column is W(:, kw + 1)
  int cnt[3];      // workgroups that finished (per kernel kind)
  double absakk, colmax;
  double c0_kk, c0_kp, c1_kk, c1_kp;   // W(kk, kw), W(kp, kw), W(kk, kw + 1), W(kp, kw + 1) before the interchange
  double c0_k;                          // W(k, kw)
  double akk_old;                       // a(kk, kk) before it is overwritten
  int inertia[3];                       // pos, neg, null (bk_inertia_kernel)
};

__device__ __forceinline__ void bk_argmax_combine(double& v, int& i, double v2, int i2)
{
  if(v2 > v || (v2 == v && i2 < i)) {   // first index among equal maxima, like IDAMAX
    v = v2;
    i = i2;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the whole panel in ONE launch.  History of the form, N = 8192, random symmetric matrix (scripts/bk_time.py; 5229 column steps,
// 4746 of them with phase B; profiles/r05_probes/README.md):
//   173 ms  three launches per column (rounds 1-4)
//   157 ms  one launch per panel, grid barriers behind every phase (4 per two-phase step), decisions loaded by one lane
//   136 ms  decision scalars published by the rows' owners and fetched in one round trip by the deciding wave; 3 barriers per step
//   128 ms  participants chosen at run time on ONE XCD, stores kept in that XCD's L2
//   121 ms  tagged granules instead of the barriers behind phases A and B: one barrier per step
//   117 ms  interchanges of the columns in front of the panel once per panel
//    94 ms  16 workgroups of 512 threads instead of 8 x 1024 (shorter workgroup barriers and reductions; 32 row entries in flight instead
//           of 16 is slower again: 102 ms)
// Per column step now (workgroup 0's clock, HIOPAMD_BK_TIMING build): phase A 3.2 + wait / decide 4.2, phase B 3.4 + 4.9, phase C +
// barrier 4.4 us; what is left is ~12 dependent L2 round trips per step.  Tried and not kept: 32 workgroups of 256 threads (a barrier
// costs one atomic per workgroup on one word), the rows of the panel's L columns kept in registers (spills) or in LDS (needs 32
// workgroups again).
#ifndef HIOPAMD_BK_TIMING
#define HIOPAMD_BK_TIMING 0
#endif
#ifndef HIOPAMD_BK_G
#define HIOPAMD_BK_G 16
#define HIOPAMD_BK_T 512
#define HIOPAMD_BK_DEPTH 16
#endif
constexpr int BK_G = HIOPAMD_BK_G;        // workgroups of the panel kernel (<= 16: the deciding wave polls 3 BK_G + 16 granules, one per lane)
constexpr int BK_T = HIOPAMD_BK_T;        // threads per workgroup: one row per thread up to n = BK_G * BK_T
constexpr int BK_DEPTH = HIOPAMD_BK_DEPTH;   // row entries in flight per thread in the column phases
constexpr int BK_GR_PUB = 6 * BK_G;       // first granule of the published scalars (behind 2 phases x BK_G workgroups x 3 granules)
constexpr int BK_GR_N = BK_GR_PUB + 16;   // granules
static_assert(3 * BK_G + 16 <= 64, "the deciding wave has one lane per granule");
constexpr long long BK_BAR_TIMEOUT = 200000000ll;   // 2 s of the 100 MHz clock

__device__ __forceinline__ double bk_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bk_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// XCD-local forms (bk_panel_kernel<true>: every participating workgroup runs on ONE XCD, established at run time): a store that stays in
// that XCD's L2 (no sc1: the line is kept, MI355X_MICROARCH.md "stores of each flavour") — the readers' sc1 loads bypass their L1 and are
// served by the same L2 —, and read-modify-write atomics executed in that L2 instead of at the memory side.
__device__ __forceinline__ void bk_st_l2(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void bk_sti_l2(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int bk_ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bk_sti(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// false: the wait expired (some workgroup of the grid is not running): every workgroup leaves, the host reports an error
template <bool LOCAL>
__device__ __forceinline__ bool bk_grid_barrier(unsigned* bar, unsigned& target, unsigned G, int* sh_ok)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if(threadIdx.x == 0) {
    target += G;
    const unsigned before = LOCAL ? __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                  : __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    unsigned spins = 0;
    long long t0 = 0;
    while(before + 1u < target && __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {   // (the last one to arrive does not poll)
      __builtin_amdgcn_s_sleep(2);
      if((++spins & 1023u) == 0) {
        const long long now = (long long)wall_clock64();
        if(t0 == 0) t0 = now;
        if(now - t0 > BK_BAR_TIMEOUT || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *sh_ok = ok;
  }
  __syncthreads();
  return *sh_ok != 0;
}

// LOCAL (the form the host launches for n > BK_T): 8 x as many workgroups as the panel needs are launched; each one reports the XCD it runs on
// (s_getreg XCC_ID), all wait for each other ONCE, and the (up to BK_G) workgroups of the XCD that got the most — the dispatcher deals a
// grid round-robin over the eight XCDs, so normally 8 of 64 — do the panel while the others leave.  Whatever the dispatcher did, the
// participants are on one XCD by construction, which is what allows the L2-resident stores and L2-executed atomics above: every
// cross-workgroup dependency of a column step then costs an L2 round trip instead of one to the memory side.
// bar[0] barrier counter, bar[1] abort, bar[2] arrivals of the rendezvous, bar[4 + x] workgroups that reported XCD x.
template <bool LOCAL>
__global__ __launch_bounds__(BK_T) void bk_panel_kernel(int n, int k0, int kcap, double* __restrict__ A, int64_t lda, double* __restrict__ Wb,
                                                          int64_t ldw, BkState* __restrict__ st, double* __restrict__ pval,
                                                          int* __restrict__ pidx, int* __restrict__ ipiv, int* __restrict__ perm,
                                                          double* __restrict__ e, unsigned* __restrict__ bar,
                                                          unsigned long long* __restrict__ gran)
{
  __shared__ double coef[BK_NB];
  __shared__ double rv[BK_T / 64];
  __shared__ int ri[BK_T / 64];
  __shared__ int sh_ok;
  __shared__ int sh_rank, sh_G;
  // the decision of the current column, identical in every workgroup
  __shared__ int d_need2, d_imax, d_kp, d_kstep, d_use_c1;
  __shared__ double d_absakk, d_colmax, d_c0_k, d_c0_kk, d_c0_kp, d_c1_kk, d_c1_kp, d_akk_old;
  const int tid = threadIdx.x;
  unsigned G = gridDim.x, g = blockIdx.x;
  if constexpr(LOCAL) {
    if(tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 7u;
      const unsigned mine = __hip_atomic_fetch_add(bar + 4 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the count above is at the memory side before the arrival is)
      (void)__hip_atomic_fetch_add(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int rank = -1, Gact = 0;
      unsigned spins = 0;
      long long t0 = 0;
      bool ok = true;
      while(__hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        __builtin_amdgcn_s_sleep(2);
        if((++spins & 1023u) == 0) {
          const long long now = (long long)wall_clock64();
          if(t0 == 0) t0 = now;
          if(now - t0 > BK_BAR_TIMEOUT || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = false;
            break;
          }
        }
      }
      if(ok) {
        unsigned cnt[8];
#pragma unroll
        for(int x = 0; x < 8; ++x) cnt[x] = __hip_atomic_load(bar + 4 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned xs = 0;
#pragma unroll
        for(unsigned x = 1; x < 8; ++x)
          if(cnt[x] > cnt[xs]) xs = x;
        Gact = (int)(cnt[xs] < (unsigned)BK_G ? cnt[xs] : (unsigned)BK_G);
        rank = (xcc == xs && mine < (unsigned)Gact) ? (int)mine : -1;
      }
      sh_rank = rank;
      sh_G = Gact;
    }
    __syncthreads();
    if(sh_rank < 0) return;   // not on the chosen XCD (or the rendezvous expired: the abort word is set)
    g = (unsigned)sh_rank;
    G = (unsigned)sh_G;
  }
  // stores of everything another workgroup of the panel reads
  auto pst = [&](double* p, double v) {
    if constexpr(LOCAL) bk_st_l2(p, v);
    else bk_st(p, v);
  };
  auto psti = [&](int* p, int v) {
    if constexpr(LOCAL) bk_sti_l2(p, v);
    else bk_sti(p, v);
  };
  unsigned target = 0u;
  int k = st->next_k;   // (written before this launch)
  // Scalars that cross workgroups travel as GRANULES: naturally aligned 8-byte words {32-bit payload, 32-bit tag} written by one store each
  // (never torn), zeroed before the launch; a value is complete for a reader when every one of its granules carries the tag it expects,
  // whatever the order in which the stores arrive.  tag = 4 k + 1 for what phase A of column k publishes, 4 k + 2 for phase B (k only
  // grows inside a launch, 0 is never a tag).
  //   gran[(phase * BK_G + g) * 3 + {0, 1, 2}]   partial maximum of workgroup g: low word, high word of the value, row index
  //   gran[6 BK_G + 2 q + {0, 1}]                 published by the OWNERS of the rows in question (low, high word):
  //                                           q = 0 W(k, kw)   1 W(k+1, kw)   6 a(k, k)   7 a(k+1, k+1)              (phase A)
  //                                               2 W(imax, kw)   3 W(k, kw+1)   4 W(k+1, kw+1)   5 W(imax, kw+1)    (phase B)
  // The deciding wave of every workgroup polls the granules it needs — that IS the synchronisation behind phases A and B: nothing phase B
  // or C reads of another workgroup's rows was written in phase A or B of the same column (a row's panel entries, the rows of W and the
  // interchanges all date from earlier columns), only the decision depends on everybody.  So a column step has ONE grid barrier, at its
  // end (phase C's interchanges write rows that other workgroups' threads read in the next column).  A granule is rewritten one column
  // later at the earliest, i.e. behind that barrier, which every reader of the old value has passed.
  auto gput = [&](int idx, unsigned payload, unsigned tag) {
    const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)payload;
    if constexpr(LOCAL) __hip_atomic_store(gran + idx, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(gran + idx, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto gput_f64 = [&](int idx, double v, unsigned tag) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    gput(idx, (unsigned)(b & 0xffffffffull), tag);
    gput(idx + 1, (unsigned)(b >> 32), tag);
  };
  // one phase of the column kernels for the rows this workgroup owns; the workgroup's partial maximum goes to its granules
  auto column_phase = [&](bool second, int src) {
    const int kw = k - k0, col = second ? kw + 1 : kw;
    const unsigned tag = 4u * (unsigned)k + (second ? 2u : 1u);
    __syncthreads();
    for(int p = tid; p < kw; p += BK_T) coef[p] = bk_ld(Wb + (int64_t)p * ldw + src);
    __syncthreads();
    double best = -1.0;
    int bidx = INT_MAX;
    for(int blk = k / BK_T; blk * BK_T < n; ++blk) {
      if((unsigned)blk % G != g) continue;
      const int i = blk * BK_T + tid;
      if(i >= k && i < n) {
        double v = (!second || i >= src) ? bk_ld(A + (int64_t)src * lda + i) : bk_ld(A + (int64_t)i * lda + src);
        const bool special = i == k || i == k + 1 || (second && i == src);
        double diag = 0.0, c0own = 0.0;
        if(special) {   // (three threads of the grid)
          if(!second) diag = bk_ld(A + (int64_t)i * lda + i);
          else c0own = bk_ld(Wb + (int64_t)kw * ldw + i);   // this thread's own result of phase A
        }
        const double* Ap = A + (int64_t)k0 * lda + i;
        // BK_DEPTH loads in flight at a time (the compiler keeps atomic loads in program order and would otherwise wait for each one
        // before the multiply-add that consumes it: up to 63 dependent round trips per row)
        int p = 0;
        for(; p + BK_DEPTH <= kw; p += BK_DEPTH) {
          double t16[BK_DEPTH];
#pragma unroll
          for(int q = 0; q < BK_DEPTH; ++q) t16[q] = bk_ld(Ap + (int64_t)(p + q) * lda);
#pragma unroll
          for(int q = 0; q < BK_DEPTH; ++q) v -= t16[q] * coef[p + q];
        }
        if(p < kw) {   // the rest, padded with zero products: the same fused multiply-adds as above, whatever BK_DEPTH is
          double t16[BK_DEPTH];
#pragma unroll
          for(int q = 0; q < BK_DEPTH; ++q) t16[q] = (p + q < kw) ? bk_ld(Ap + (int64_t)(p + q) * lda) : 0.0;
#pragma unroll
          for(int q = 0; q < BK_DEPTH; ++q) v -= t16[q] * ((p + q < kw) ? coef[p + q] : 0.0);
        }
        pst(Wb + (int64_t)col * ldw + i, v);
        if(special) {
          if(!second) {
            if(i == k) {
              gput_f64(BK_GR_PUB + 2 * 0, v, tag);
              gput_f64(BK_GR_PUB + 2 * 6, diag, tag);
            }
            if(i == k + 1) {
              gput_f64(BK_GR_PUB + 2 * 1, v, tag);
              gput_f64(BK_GR_PUB + 2 * 7, diag, tag);
            }
          } else {
            if(i == k) gput_f64(BK_GR_PUB + 2 * 3, v, tag);
            if(i == k + 1) gput_f64(BK_GR_PUB + 2 * 4, v, tag);
            if(i == src) {
              gput_f64(BK_GR_PUB + 2 * 2, c0own, tag);
              gput_f64(BK_GR_PUB + 2 * 5, v, tag);
            }
          }
        }
        const bool cand = second ? (i != src) : (i > k);
        if(cand) bk_argmax_combine(best, bidx, fabs(v), i);
      }
    }
    for(int off = 32; off > 0; off >>= 1) {
      const double v2 = __shfl_down(best, off, 64);
      const int i2 = __shfl_down(bidx, off, 64);
      bk_argmax_combine(best, bidx, v2, i2);
    }
    if((tid & 63) == 0) {
      rv[tid >> 6] = best;
      ri[tid >> 6] = bidx;
    }
    __syncthreads();
    if(tid == 0) {
      for(int w = 1; w < BK_T / 64; ++w) bk_argmax_combine(best, bidx, rv[w], ri[w]);
      const int base = ((second ? BK_G : 0) + (int)g) * 3;
      gput_f64(base, best, tag);
      gput(base + 2, (unsigned)bidx, tag);
    }
  };
  // the deciding wave (wave 0 of every workgroup): lane L < 3 G polls granule L of the phase's partial maxima, lanes 3 BK_G .. 3 BK_G + 15 the sixteen
  // granules of the published scalars; when every needed granule carries its tag the wave folds the maxima.  Every lane returns with the
  // folded maximum, `pubv(q)` hands out scalar q.  false: the wait expired (abort word set, every workgroup leaves).
  unsigned dw_pl = 0u;
  auto fold = [&](bool second, double& best, int& bidx) -> bool {
    const int lane = tid & 63;
    const unsigned tagA = 4u * (unsigned)k + 1u, tagB = tagA + 1u;
    int idx = -1;
    unsigned want = 0u;
    if(lane < 3 * (int)G) {
      idx = (second ? 3 * BK_G : 0) + lane;
      want = second ? tagB : tagA;
    } else if(lane >= 3 * BK_G && lane < 3 * BK_G + 16) {
      const int q = (lane - 3 * BK_G) >> 1;
      const bool fromA = q == 0 || q == 1 || q == 6 || q == 7;
      // phase A's decision reads scalars 0 and 6 only (row k + 1 may not exist; when it does not, no second phase follows)
      const bool needed = second ? true : (q == 0 || q == 6);
      if(needed) {
        idx = BK_GR_PUB + (lane - 3 * BK_G);
        want = fromA ? tagA : tagB;
      }
    }
    unsigned spins = 0;
    long long t0 = 0;
    bool ok = true;
    for(;;) {
      unsigned long long w = 0ull;
      if(idx >= 0) w = __hip_atomic_load(gran + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool have = idx < 0 || (unsigned)(w >> 32) == want;
      dw_pl = (unsigned)(w & 0xffffffffull);
      if(__all(have)) break;
      __builtin_amdgcn_s_sleep(1);
      if((++spins & 1023u) == 0) {
        const long long now = (long long)wall_clock64();
        if(t0 == 0) t0 = now;
        if(now - t0 > BK_BAR_TIMEOUT || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          if(lane == 0) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
      }
    }
    // lane j < G: workgroup j's partial maximum from granules 3 j, 3 j + 1, 3 j + 2
    const int j3 = (lane < (int)G) ? 3 * lane : 0;
    const unsigned lo = __shfl(dw_pl, j3, 64), hi = __shfl(dw_pl, j3 + 1, 64), ix = __shfl(dw_pl, j3 + 2, 64);
    best = -1.0;
    bidx = INT_MAX;
    if(lane < (int)G) {
      best = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
      bidx = (int)ix;
    }
    for(int off = 8; off > 0; off >>= 1) {   // (G <= 16)
      const double v2 = __shfl_down(best, off, 64);
      const int i2 = __shfl_down(bidx, off, 64);
      bk_argmax_combine(best, bidx, v2, i2);
    }
    best = __shfl(best, 0, 64);
    bidx = __shfl(bidx, 0, 64);
    return ok;
  };
  auto pubv = [&](int q) {
    const unsigned lo = __shfl(dw_pl, 3 * BK_G + 2 * q, 64), hi = __shfl(dw_pl, 3 * BK_G + 1 + 2 * q, 64);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
  };
#if HIOPAMD_BK_TIMING
  long long tm_last = (long long)wall_clock64(), tm_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BK_LAP(q)                                   \
  if(tid == 0) {                                    \
    const long long now__ = (long long)wall_clock64(); \
    tm_acc[q] += now__ - tm_last;                   \
    tm_last = now__;                                \
  }
#else
#define BK_LAP(q)
#endif
  while(k < kcap) {
    const int kw = k - k0;
    // ---- phase A
    column_phase(false, k);
    BK_LAP(0)
    if(tid < 64) {
      double best;
      int bidx;
      const bool ok = fold(false, best, bidx);
      const double wkk = pubv(0), akk = pubv(6);
      if(tid == 0) {
        sh_ok = ok ? 1 : 0;
        const double absakk = fabs(wkk);
        const double colmax = (best >= 0.0) ? best : 0.0;
        d_imax = (best >= 0.0) ? bidx : k;
        d_absakk = absakk;
        d_colmax = colmax;
        d_c0_k = wkk;
        int need2 = 0;
        if(!(fmax(absakk, colmax) > 0.0)) {   // exactly zero column (or NaN): DSYTRF's INFO = k + 1, no interchange
          if(g == 0 && st->info == 0) st->info = k + 1;
        } else if(!(absakk >= BK_ALPHA * colmax)) {
          need2 = 1;
        }
        d_need2 = need2;
        if(!need2) {
          d_kp = k;
          d_kstep = 1;
          d_use_c1 = 0;
          d_c0_kk = d_c0_kp = wkk;
          d_c1_kk = d_c1_kp = 0.0;
          d_akk_old = akk;
        }
      }
    }
    __syncthreads();
    if(!sh_ok) return;
    BK_LAP(1)
    // ---- phase B
    if(d_need2) {
      const int imax = d_imax;
      column_phase(true, imax);
      BK_LAP(2)
      if(tid < 64) {
        double best;
        int bidx;
        const bool ok = fold(true, best, bidx);
        const double c0_k = pubv(0), c0_k1 = pubv(1), c0_im = pubv(2), c1_k = pubv(3), c1_k1 = pubv(4), c1_im = pubv(5), a_k = pubv(6),
                     a_k1 = pubv(7);
        if(tid == 0) {
          sh_ok = ok ? 1 : 0;
          const double rowmax = (best >= 0.0) ? best : 0.0;
          const double absakk = d_absakk, colmax = d_colmax;
          const double wii = fabs(c1_im);
          int kp, kstep, use_c1 = 0;
          if(absakk >= BK_ALPHA * colmax * (colmax / rowmax)) {
            kp = k;
            kstep = 1;
          } else if(wii >= BK_ALPHA * rowmax) {
            kp = imax;
            kstep = 1;
            use_c1 = 1;
          } else {
            kp = imax;
            kstep = 2;
          }
          // row kk = k (kstep 1) or k + 1 (kstep 2); row kp = k or imax (imax may be k + 1: then both published copies are the same row's)
          d_kp = kp;
          d_kstep = kstep;
          d_use_c1 = use_c1;
          d_c0_kk = (kstep == 2) ? c0_k1 : c0_k;
          d_c1_kk = (kstep == 2) ? c1_k1 : c1_k;
          d_c0_kp = (kp == k) ? c0_k : c0_im;
          d_c1_kp = (kp == k) ? c1_k : c1_im;
          d_akk_old = (kstep == 2) ? a_k1 : a_k;
        }
      }
      __syncthreads();
      if(!sh_ok) return;
      BK_LAP(3)
    }
    // ---- phase C
    const int kp = d_kp, kstep = d_kstep, use_c1 = d_use_c1, kk = k + kstep - 1;
    const double c0_kk = d_c0_kk, c0_kp = d_c0_kp, c1_kk = d_c1_kk, c1_kp = d_c1_kp, c0_k = d_c0_k, akk_old = d_akk_old;
    const bool swp = kp != kk;
    for(int blk = 0; blk * BK_T < n; ++blk) {
      if((unsigned)blk % G != g) continue;
      const int64_t j = (int64_t)blk * BK_T + tid;
      if(j >= n) continue;
      if(swp && j >= k0 && j < k) {   // rows kk and kp of L, the panel's previous columns (the columns in front of the panel: once per
                                      // panel, bk_defer_swaps_kernel)
        double* pa = A + j * lda;
        const double u = bk_ld(pa + kk);
        pst(pa + kk, bk_ld(pa + kp));
        pst(pa + kp, u);
      }
      if(swp && j < kw) {   // rows kk and kp of W, the panel's previous columns
        double* pw = Wb + j * ldw;
        const double u = bk_ld(pw + kk);
        pst(pw + kk, bk_ld(pw + kp));
        pst(pw + kp, u);
      }
      const int64_t i = j;
      if(i >= k) {
        const bool is_kk = swp && i == kk, is_kp = swp && i == kp;
        // everything this row reads, in flight together (one memory round trip; the loads are this thread's own row of W and its entry of
        // column kk, which nobody else writes in this phase)
        const bool need1 = use_c1 || kstep == 2;   // (uniform)
        const double wc0 = bk_ld(Wb + (int64_t)kw * ldw + i);
        const double wc1 = need1 ? bk_ld(Wb + (int64_t)(kw + 1) * ldw + i) : 0.0;
        const double akki = (swp && i > kk && i != kp) ? bk_ld(A + (int64_t)kk * lda + i) : 0.0;
        double w0, w1 = 0.0;
        if(is_kk) {
          w0 = use_c1 ? c1_kp : c0_kp;
          w1 = c1_kp;
        } else if(is_kp) {
          w0 = use_c1 ? c1_kk : c0_kk;
          w1 = c1_kk;
        } else {
          w0 = use_c1 ? wc1 : wc0;
          if(kstep == 2) w1 = wc1;
        }
        if(use_c1 || is_kk || is_kp) pst(Wb + (int64_t)kw * ldw + i, w0);
        if(kstep == 2 && (is_kk || is_kp)) pst(Wb + (int64_t)(kw + 1) * ldw + i, w1);
        if(swp) {
          if(i == kp) pst(A + (int64_t)kp * lda + kp, akk_old);
          else if(i > kk && i < kp) pst(A + i * lda + kp, akki);
          else if(i > kp) pst(A + (int64_t)kp * lda + i, akki);
        }
        if(kstep == 1) {
          const double dk = swp ? (use_c1 ? c1_kp : c0_kp) : c0_k;
          if(i == k) pst(A + (int64_t)k * lda + k, dk);
          else pst(A + (int64_t)k * lda + i, (dk != 0.0) ? w0 * (1.0 / dk) : w0);
        } else {
          const double wk0 = c0_k;
          const double wk10 = swp ? c0_kp : c0_kk;
          const double wk11 = swp ? c1_kp : c1_kk;
          if(i == k) {
            pst(A + (int64_t)k * lda + k, wk0);
          } else if(i == k + 1) {
            pst(A + (int64_t)k * lda + k + 1, 0.0);
            pst(e + k, wk10);
            pst(A + (int64_t)(k + 1) * lda + k + 1, wk11);
          } else {
            double d21 = wk10;
            const double d11 = wk11 / d21, d22 = wk0 / d21;
            const double tt = 1.0 / (d11 * d22 - 1.0);
            d21 = tt / d21;
            pst(A + (int64_t)k * lda + i, d21 * (d11 * w0 - w1));
            pst(A + (int64_t)(k + 1) * lda + i, d21 * (d22 * w1 - w0));
          }
        }
      }
    }
    if(g == 0 && tid == 0) {
      if(kstep == 1) {
        psti(ipiv + k, kp + 1);
      } else {
        psti(ipiv + k, -(kp + 1));
        psti(ipiv + k + 1, -(kp + 1));
      }
      if(swp) {
        const int u = bk_ldi(perm + kk);
        psti(perm + kk, bk_ldi(perm + kp));
        psti(perm + kp, u);
      }
    }
    __syncthreads();
    BK_LAP(4)
    if(!bk_grid_barrier<LOCAL>(bar, target, G, &sh_ok)) return;
    BK_LAP(5)
    k += kstep;
  }
  if(g == 0 && tid == 0) st->next_k = k;
#if HIOPAMD_BK_TIMING
  if(g == 0 && tid == 0)
    for(int q = 0; q < 6; ++q) atomicAdd(gran + BK_GR_N + q, (unsigned long long)tm_acc[q]);
#endif
}

// The row interchanges of the panel [k0, kend) applied to the columns IN FRONT of the panel (j < k0), once per panel.  DLASYF interchanges
// rows kk and kp of all earlier columns in every column step; the columns in front of the panel are neither read nor written while the
// panel is factored, so applying the same interchanges in the same order afterwards gives the same matrix — and takes ~k uncoalesced
// two-line accesses out of every column step (they were what phase C waited for: 4.8 of a step's 21.5 us at N = 8192).
// One thread per column, 64 per workgroup: the column's entries in the panel's 64 rows and in the (at most 64) rows further down that an
// interchange names are gathered into LDS, interchanged there in pivot order, and written back: ~8 batches of loads per panel instead of
// a dependent round trip per interchange.
constexpr int BK_DS_PITCH = 2 * BK_NB + 1;
__global__ __launch_bounds__(64) void bk_defer_swaps_kernel(int n, int k0, int kend, double* __restrict__ A, int64_t lda,
                                                            const int* __restrict__ ipiv)
{
  extern __shared__ double ds_buf[];   // 64 x BK_DS_PITCH
  __shared__ int s_kk[BK_NB], s_kp[BK_NB], s_b[BK_NB], s_far[BK_NB];
  __shared__ int s_m, s_nfar;
  const int tid = threadIdx.x;
  if(tid == 0) {   // the panel's interchanges in pivot order
    int m = 0;
    for(int k = k0; k < kend;) {
      const int pv = ipiv[k];
      const int kstep = pv > 0 ? 1 : 2;
      const int kp = (pv > 0 ? pv : -pv) - 1, kk = k + kstep - 1;
      if(kp != kk) {
        s_kk[m] = kk;
        s_kp[m] = kp;
        ++m;
      }
      k += kstep;
    }
    s_m = m;
  }
  __syncthreads();
  const int m = s_m;
  if(m == 0) return;
  // slot of row kp: inside the panel's 64 rows its offset, otherwise 64 + (first interchange that names the same row)
  if(tid < m) {
    const int kp = s_kp[tid];
    int slot = kp - k0;
    if(slot >= BK_NB) {
      int first = tid;
      for(int t = 0; t < tid; ++t)
        if(s_kp[t] == kp) {
          first = t;
          break;
        }
      slot = BK_NB + first;
    }
    s_b[tid] = slot;
    s_far[tid] = (slot == BK_NB + tid) ? kp : -1;   // this interchange owns far slot 64 + tid
  }
  __syncthreads();
  const int j = blockIdx.x * 64 + tid;
  if(j >= k0) return;
  double* col = A + (int64_t)j * lda;
  double* buf = ds_buf + (size_t)tid * BK_DS_PITCH;
  const int nrow = (n - k0 < BK_NB) ? n - k0 : BK_NB;   // (all 64 rows behind k0: an interchange may name row k0 + 63 of a 63-column panel)
#pragma unroll 4
  for(int r0 = 0; r0 < BK_NB; r0 += 16) {
    double t16[16];
#pragma unroll
    for(int q = 0; q < 16; ++q) t16[q] = (r0 + q < nrow) ? col[k0 + r0 + q] : 0.0;
#pragma unroll
    for(int q = 0; q < 16; ++q) buf[r0 + q] = t16[q];
  }
  for(int r0 = 0; r0 < m; r0 += 16) {
    double t16[16];
#pragma unroll
    for(int q = 0; q < 16; ++q) t16[q] = (r0 + q < m && s_far[r0 + q] >= 0) ? col[s_far[r0 + q]] : 0.0;
#pragma unroll
    for(int q = 0; q < 16; ++q)
      if(r0 + q < m) buf[BK_NB + r0 + q] = t16[q];
  }
  for(int t = 0; t < m; ++t) {
    const int a = s_kk[t] - k0, b = s_b[t];
    const double u = buf[a];
    buf[a] = buf[b];
    buf[b] = u;
  }
  for(int r = 0; r < nrow; ++r) col[k0 + r] = buf[r];
  for(int t = 0; t < m; ++t)
    if(s_far[t] >= 0) col[s_far[t]] = buf[BK_NB + t];
}

__global__ __launch_bounds__(kBlock) void bk_iota_kernel(int n, int* __restrict__ perm)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if(i < n) perm[i] = i;
}

// inertia from D with the reference's rule and thresholds (hiopLinSolverSymDenseLapack.hpp:127-167, LINPACK dsidi): a 2 x 2 block
// counts the sign of (d_k / t) d_k+1 - t, t = |e_k|, then one positive
__global__ __launch_bounds__(kBlock) void bk_inertia_kernel(int n, const double* __restrict__ A, int64_t lda, const double* __restrict__ e,
                                                            BkState* __restrict__ st)
{
  __shared__ int cnt[3];
  if(threadIdx.x < 3) cnt[threadIdx.x] = 0;
  __syncthreads();
  for(int k = threadIdx.x; k < n; k += kBlock) {
    double d = A[(int64_t)k * lda + k];
    if(e[k] != 0.0) {                         // first row of a 2 x 2 block
      const double t = fabs(e[k]);
      d = (d / t) * A[(int64_t)(k + 1) * lda + k + 1] - t;
    } else if(k > 0 && e[k - 1] != 0.0) {     // its second row
      d = fabs(e[k - 1]);
    }
    atomicAdd(&cnt[d < -1e-14 ? 1 : (d < 1e-14 ? 2 : 0)], 1);
  }
  __syncthreads();
  if(threadIdx.x < 3) st->inertia[threadIdx.x] = cnt[threadIdx.x];
}

// ---- solve ----
__global__ __launch_bounds__(kBlock) void bk_gather_kernel(int n, const int* __restrict__ perm, const double* __restrict__ x,
                                                           double* __restrict__ y, int scatter)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if(i >= n) return;
  if(scatter) y[perm[i]] = x[i];
  else y[i] = x[perm[i]];
}

// z = D^-1 y with the 2 x 2 formula of DSYTRS (dsytrs.f, lower branch)
__global__ __launch_bounds__(kBlock) void bk_dsolve_kernel(int n, const double* __restrict__ A, int64_t lda, const double* __restrict__ e,
                                                           double* __restrict__ y)
{
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if(k >= n) return;
  if(k > 0 && e[k - 1] != 0.0) return;   // second row of a block: done by the thread of the first
  const double dk = A[(int64_t)k * lda + k];
  if(e[k] != 0.0) {
    const double akm1k = e[k];
    const double akm1 = dk / akm1k, ak = A[(int64_t)(k + 1) * lda + k + 1] / akm1k;
    const double denom = akm1 * ak - 1.0;
    const double bkm1 = y[k] / akm1k, bk = y[k + 1] / akm1k;
    y[k] = (ak * bkm1 - bk) / denom;
    y[k + 1] = (akm1 * bk - bkm1) / denom;
  } else {
    y[k] = y[k] / dk;
  }
}

}  // namespace hiopamd

using namespace hiopamd;

// The solve works on 256-row block steps with PRE-INVERTED diagonal blocks (round 5; before: 2 x n / 64 steps of a one-wave 64 x 64
// triangular solve + a skinny GEMV — 1024 kernels of ~8 us at n = 8192, 8.7 ms per solve, and a HIP graph of them still 8.0: the kernels
// themselves are latency).  X_J = L_JJ^-1 (unit lower triangular, 256 x 256) is built once per factorisation, column by column by
// forward substitution on e_c, into M_J[c][i] = X_J(i, c) (column c of the inverse contiguous, zero above the diagonal); a block step is
// then two GEMVs of shapes the library is fast at: y_J = M_J^T v_J and v_rest -= L(rest, J) y_J.
constexpr int BK_SB = 256;
constexpr int BK_INV_PITCH = BK_SB + 1;
// grid = 4 x (number of diagonal blocks), one wave each: lane c of strip q owns column 64 q + c of the block's inverse and keeps it in
// LDS (64 x 257 doubles).  Right-looking: eight columns of L at a time are staged into LDS (contiguous in memory: coalesced, one round
// trip per eight columns), then for each of them every lane subtracts L(i, k) x(k) from the rest of its column — the same (i, k) for
// every lane (entries above a column's diagonal are zeros that are multiplied along), so the staged column is an LDS broadcast.
// (The first form — one thread per column, the column in global memory — took 4.5 ms at n = 8192: a dependent L2 round trip per term.)
constexpr int BK_INV_NC = 8;
__global__ __launch_bounds__(64) void bk_invert_diag_kernel(int n, const double* __restrict__ A, int64_t lda, double* __restrict__ Minv)
{
  extern __shared__ double bk_inv_lds[];   // 64 x BK_INV_PITCH (the lanes' columns) + BK_INV_NC x BK_SB (staged columns of L)
  const int blk = blockIdx.x >> 2, c0 = (blockIdx.x & 3) * 64, lane = threadIdx.x, c = c0 + lane;
  const int J0 = blk * BK_SB;
  const int bs = (n - J0 < BK_SB) ? (n - J0) : BK_SB;
  if(c0 >= bs) return;
  double* x = bk_inv_lds + lane * BK_INV_PITCH;
  double* Ls = bk_inv_lds + 64 * BK_INV_PITCH;
  for(int i = c0; i < bs; ++i) x[i] = (i == c) ? 1.0 : 0.0;
  const double* L = A + (int64_t)J0 * lda + J0;   // L(J0 + i, J0 + k) = L[k * lda + i]
  for(int k0 = c0; k0 < bs; k0 += BK_INV_NC) {
    __syncthreads();   // (one wave: orders the LDS traffic of the previous round against the staging below)
#pragma unroll
    for(int q = 0; q < BK_INV_NC; ++q) {
      const int k = k0 + q;
      for(int i = k0 + lane; i < bs; i += 64) Ls[q * BK_SB + i] = (k < bs && i > k) ? L[(int64_t)k * lda + i] : 0.0;
    }
    __syncthreads();
    for(int q = 0; q < BK_INV_NC; ++q) {
      const int k = k0 + q;
      if(k >= bs) break;
      const double xk = x[k];   // final: every earlier column has been applied (0 for the lanes whose column starts below k)
      const double* lq = Ls + q * BK_SB;
      int i = k + 1;
      for(; i + 8 <= bs; i += 8) {   // (reads first, writes last: the compiler cannot know that x and the staged column do not overlap)
        double l8[8], x8[8];
#pragma unroll
        for(int u = 0; u < 8; ++u) {
          l8[u] = lq[i + u];
          x8[u] = x[i + u];
        }
#pragma unroll
        for(int u = 0; u < 8; ++u) x[i + u] = x8[u] - l8[u] * xk;
      }
      for(; i < bs; ++i) x[i] -= lq[i] * xk;
    }
  }
  if(c < bs) {
    double* col = Minv + (int64_t)blk * BK_SB * BK_SB + (int64_t)c * BK_SB;   // M[c][i] = X(i, c)
    for(int i = c; i < bs; ++i) col[i] = x[i];
  }
}

struct hiopamd_ldlt_bk {
  hiopamd_ctx* ctx = nullptr;
  int n = 0;
  double* Wb = nullptr;      // 64 x n panel W = L D (column p of the panel contiguous)
  double* e = nullptr;       // n: off-diagonals of the 2 x 2 blocks of D
  double* tmp = nullptr;     // n: permuted right-hand side
  double* tmp2 = nullptr;    // n: second vector of the sweeps
  double* Minv = nullptr;    // ceil(n / 256) inverted diagonal blocks of L, 256 x 256 each (bk_invert_diag_kernel)
  double* pval = nullptr;    // partial maxima of the column kernels
  int* pidx = nullptr;
  int* ipiv = nullptr;       // n: LAPACK's IPIV (1-based, negative for 2 x 2)
  int* perm = nullptr;       // n: (P A P^T)[i][j] = A[perm[i]][perm[j]]
  BkState* st = nullptr;
  unsigned* bar = nullptr;   // grid-barrier counter + abort word of the panel kernel
  unsigned long long* gran = nullptr;   // 64 tagged granules of the panel kernel (partial maxima, published scalars)
  bool factored = false;
  // the two triangular sweeps of a solve as an instantiated HIP graph (~260 short kernels at n = 8192): captured on the second
  // solve with the same matrix address (the first one runs eagerly and sizes the context's workspace), replayed from then on
  hipGraphExec_t sweeps = nullptr;
  const double* sw_A = nullptr;
  int64_t sw_lda = 0;
  void* sw_work = nullptr;     // the context's workspace at capture time (the GEMV partial sums live there)
  bool sw_warm = false;
  // The multi-workgroup panel kernel needs its workgroups resident TOGETHER (one rendezvous, one grid barrier per column).  When a
  // barrier expires once (a shared or busy device), this object factors with ONE workgroup from then on: no other workgroup to wait for,
  // so nothing can expire — slow (one CU walks every column) but it always finishes.  The call that timed out still returns
  // HIOPAMD_ERR_TIMEOUT (the matrix is partly overwritten: the caller re-assembles and calls again, as for the dataflow LDL^T).
  bool single_workgroup = false;
};

// L y = v, D z = y, L^T w = z with v = B->tmp on entry and on exit (y, z in B->tmp2): 2 x n / 256 block steps, every one two GEMVs
static int bk_enqueue_sweeps(hiopamd_ldlt_bk* B, const double* A, int64_t lda)
{
  hiopamd_ctx* ctx = B->ctx;
  const int n = B->n;
  hipStream_t s = ctx->stream;
  double *v = B->tmp, *y = B->tmp2;
  const dim3 gn((n + kBlock - 1) / kBlock), bn(kBlock);
  for(int jb = 0; jb < n; jb += BK_SB) {   // L y = P b
    const int bs = (n - jb < BK_SB) ? (n - jb) : BK_SB;
    const double* M = B->Minv + (int64_t)(jb / BK_SB) * BK_SB * BK_SB;
    int rc = hiopamd_mat_trans_times_vec(ctx, bs, bs, M, BK_SB, 0.0, y + jb, 1.0, v + jb);   // y_J = X_J v_J = M_J^T v_J
    const int rest = n - jb - bs;
    if(rc == HIOPAMD_OK && rest > 0)
      rc = hiopamd_mat_trans_times_vec(ctx, bs, rest, A + (int64_t)jb * lda + jb + bs, lda, 1.0, v + jb + bs, -1.0, y + jb);
    if(rc != HIOPAMD_OK) return rc;
  }
  hipLaunchKernelGGL(bk_dsolve_kernel, gn, bn, 0, s, n, A, lda, B->e, y);
  for(int jb = ((n - 1) / BK_SB) * BK_SB; jb >= 0; jb -= BK_SB) {   // L^T w = z, w into v
    const int bs = (n - jb < BK_SB) ? (n - jb) : BK_SB;
    const double* M = B->Minv + (int64_t)(jb / BK_SB) * BK_SB * BK_SB;
    const int rest = n - jb - bs;
    int rc = HIOPAMD_OK;
    if(rest > 0) rc = hiopamd_mat_times_vec(ctx, bs, rest, A + (int64_t)jb * lda + jb + bs, lda, 1.0, y + jb, -1.0, v + jb + bs);
    if(rc == HIOPAMD_OK) rc = hiopamd_mat_times_vec(ctx, bs, bs, M, BK_SB, 0.0, v + jb, 1.0, y + jb);   // w_J = X_J^T z_J = M_J z_J
    if(rc != HIOPAMD_OK) return rc;
  }
  return HIOPAMD_OK;
}

extern "C" {

int hiopamd_ldlt_bk_create(hiopamd_ldlt_bk** out, hiopamd_ctx* ctx, int n)
{
  if(!out || !ctx || n < 0) return HIOPAMD_ERR_ARG;
  hiopamd_ldlt_bk* B = new hiopamd_ldlt_bk();
  B->ctx = ctx;
  B->n = n;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  const size_t nblk = std::max<size_t>((nn + BK_ROWS - 1) / BK_ROWS, 64);   // (>= the panel kernel's workgroups: one partial maximum each)
  bool ok = hipMalloc((void**)&B->Wb, sizeof(double) * nn * BK_NB) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->e, sizeof(double) * (nn + 1)) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->tmp, sizeof(double) * nn) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->tmp2, sizeof(double) * nn) == hipSuccess;
  {
    const size_t mb = sizeof(double) * ((nn + BK_SB - 1) / BK_SB) * BK_SB * BK_SB;
    ok = ok && hipMalloc((void**)&B->Minv, mb) == hipSuccess;
    if(ok) ok = hipMemset(B->Minv, 0, mb) == hipSuccess;   // (the part above the diagonal of every block stays zero)
  }
  ok = ok && hipMalloc((void**)&B->pval, sizeof(double) * nblk) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->pidx, sizeof(int) * nblk) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->ipiv, sizeof(int) * nn) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->perm, sizeof(int) * nn) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->st, sizeof(BkState)) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->bar, 16 * sizeof(unsigned)) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->gran, (BK_GR_N + 8) * sizeof(unsigned long long)) == hipSuccess;
  if(ok) (void)hipMemset(B->gran, 0, (BK_GR_N + 8) * sizeof(unsigned long long));
  if(!ok) {
    hiopamd_ldlt_bk_destroy(B);
    return HIOPAMD_ERR_HIP;
  }
  *out = B;
  return HIOPAMD_OK;
}

int hiopamd_ldlt_bk_destroy(hiopamd_ldlt_bk* B)
{
  if(!B) return HIOPAMD_OK;
  (void)hipStreamSynchronize(B->ctx->stream);
  if(B->sweeps) (void)hipGraphExecDestroy(B->sweeps);
  void* ps[] = {B->Wb, B->e, B->tmp, B->tmp2, B->Minv, B->pval, B->pidx, B->ipiv, B->perm, B->st, B->bar, B->gran};
  for(void* p : ps) (void)hipFree(p);
  delete B;
  return HIOPAMD_OK;
}

// A: n x n row-major, upper triangle (= column-major lower), overwritten by L (strict lower), d (diagonal); inertia3_host = pos, neg,
// null by the reference's rule; info_host = DSYTRF's INFO (k + 1 of the first exactly zero pivot column, 0 if none)
int hiopamd_ldlt_bk_factor(hiopamd_ldlt_bk* B, double* A, int64_t lda, int* inertia3_host, int* info_host)
{
  if(!B || (B->n > 0 && !A) || lda < B->n) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = B->ctx;
  const int n = B->n;
  B->factored = false;
  if(inertia3_host) inertia3_host[0] = inertia3_host[1] = inertia3_host[2] = 0;
  if(info_host) *info_host = 0;
  if(n == 0) {
    B->factored = true;
    return HIOPAMD_OK;
  }
  hipStream_t s = ctx->stream;
  const int64_t ldw = n;
  HIOPAMD_CHECK(hipMemsetAsync(B->st, 0, sizeof(BkState), s));
  HIOPAMD_CHECK(hipMemsetAsync(B->e, 0, sizeof(double) * ((size_t)n + 1), s));
  hipLaunchKernelGGL(bk_iota_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, B->perm);
  int k0 = 0;
  while(k0 < n) {
    const bool last = (n - k0) <= BK_NB;   // (DSYTRF factors the last block unblocked: the same recurrence with an empty trailing update)
    const int kcap = last ? n : k0 + BK_NB - 1;
    {
      // the panel's columns in ONE launch (bk_panel_kernel): its three phases per column, separated by grid barriers, are what
      // tests/test_ldlt_bk_protocol.py replays thread by thread in random order
      const unsigned G = B->single_workgroup ? 1u : (unsigned)std::min(BK_G, (n + BK_T - 1) / BK_T);
      HIOPAMD_CHECK(hipMemsetAsync(B->bar, 0, 16 * sizeof(unsigned), s));
      HIOPAMD_CHECK(hipMemsetAsync(B->gran, 0, BK_GR_N * sizeof(unsigned long long), s));
      if(G >= 2)   // eight times the workgroups: those that land on one XCD do the panel (see bk_panel_kernel)
        hipLaunchKernelGGL(bk_panel_kernel<true>, dim3(8 * G), dim3(BK_T), 0, s, n, k0, kcap, A, lda, B->Wb, ldw, B->st, B->pval, B->pidx,
                           B->ipiv, B->perm, B->e, B->bar, B->gran);
      else
        hipLaunchKernelGGL(bk_panel_kernel<false>, dim3(G), dim3(BK_T), 0, s, n, k0, kcap, A, lda, B->Wb, ldw, B->st, B->pval, B->pidx,
                           B->ipiv, B->perm, B->e, B->bar, B->gran);
    }
    HIOPAMD_CHECK(hipGetLastError());
    int kend = 0;   // where the panel ended: k0 + 63 or k0 + 64, depending on where the 2 x 2 pivots fell (the last panel: n)
    unsigned aborted[2] = {0u, 0u};
    HIOPAMD_CHECK(hipMemcpyAsync(&kend, &B->st->next_k, sizeof(int), hipMemcpyDeviceToHost, s));
    HIOPAMD_CHECK(hipMemcpyAsync(aborted, B->bar, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIOPAMD_CHECK(hipStreamSynchronize(s));
    if(aborted[1] != 0u) {
      std::fprintf(stderr, "[hiop_amd] pivoted LDL^T: a grid barrier of the panel kernel expired (its %d workgroups were not all running); the matrix "
                           "is partly overwritten — re-assemble and call again: this solver object uses the one-workgroup panel kernel from now on\n",
                   std::min(BK_G, (n + BK_T - 1) / BK_T));
      B->single_workgroup = true;
      return HIOPAMD_ERR_TIMEOUT;
    }
    if(k0 > 0 && kend > k0) {
      static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(bk_defer_swaps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(sizeof(double) * 64 * BK_DS_PITCH)) == hipSuccess;
      if(!lds_ok) return HIOPAMD_ERR_HIP;
      hipLaunchKernelGGL(bk_defer_swaps_kernel, dim3((k0 + 63) / 64), dim3(64), sizeof(double) * 64 * BK_DS_PITCH, s, n, k0, kend, A, lda, B->ipiv);
    }
    if(last) break;
    if(kend < kcap || kend > k0 + BK_NB) return HIOPAMD_ERR_STATE;
    const int kb = kend - k0;
    const int kpad = ((kb + 7) / 8) * 8;   // the update kernel walks K in steps of 8: the rows of W past kb are zero
    if(kpad > kb) HIOPAMD_CHECK(hipMemsetAsync(B->Wb + (int64_t)kb * ldw, 0, sizeof(double) * (size_t)(kpad - kb) * ldw, s));
    // A22 -= L21 D L21^T = L21 W21^T:   a(c, r) -= sum_p a(c, k0 + p) W(r, p),  c >= r >= kend
    const int rc = ldlt_rankk_update(ctx, A, lda, n, B->Wb, ldw, k0, kpad, kend, kb);   // (U rows kb..kpad-1 are masked: they alias columns this launch updates)
    if(rc != HIOPAMD_OK) return rc;
    k0 = kend;
  }
  {
    static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(bk_invert_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(sizeof(double) * (64 * BK_INV_PITCH + BK_INV_NC * BK_SB))) == hipSuccess;
    if(!lds_ok) return HIOPAMD_ERR_HIP;
    hipLaunchKernelGGL(bk_invert_diag_kernel, dim3(4 * ((n + BK_SB - 1) / BK_SB)), dim3(64), sizeof(double) * (64 * BK_INV_PITCH + BK_INV_NC * BK_SB), s, n,
                       A, lda, B->Minv);
  }
  hipLaunchKernelGGL(bk_inertia_kernel, dim3(1), dim3(kBlock), 0, s, n, A, lda, B->e, B->st);
  HIOPAMD_CHECK(hipGetLastError());
  BkState h;
  HIOPAMD_CHECK(hipMemcpyAsync(&h, B->st, sizeof(BkState), hipMemcpyDeviceToHost, s));
  HIOPAMD_CHECK(hipStreamSynchronize(s));
#if HIOPAMD_BK_TIMING
  {
    unsigned long long tm[6];
    (void)hipMemcpy(tm, B->gran + BK_GR_N, sizeof(tm), hipMemcpyDeviceToHost);
    (void)hipMemset(B->gran + 64, 0, sizeof(tm));
    std::fprintf(stderr, "[hiop_amd] pivoted panels, workgroup 0, ms: phase A %.2f | wait+decide A %.2f | phase B %.2f | wait+decide B %.2f | phase C %.2f | end barrier %.2f\n",
                 tm[0] * 1e-5, tm[1] * 1e-5, tm[2] * 1e-5, tm[3] * 1e-5, tm[4] * 1e-5, tm[5] * 1e-5);
  }
#endif
  if(h.next_k != n) return HIOPAMD_ERR_STATE;
  if(inertia3_host) {
    inertia3_host[0] = h.inertia[0];
    inertia3_host[1] = h.inertia[1];
    inertia3_host[2] = h.inertia[2];
  }
  if(info_host) *info_host = h.info;
  B->factored = true;
  return HIOPAMD_OK;
}

int hiopamd_ldlt_bk_set_single_workgroup(hiopamd_ldlt_bk* B, int enable)
{
  if(!B) return HIOPAMD_ERR_ARG;
  B->single_workgroup = enable != 0;
  return HIOPAMD_OK;
}

// x <- A^-1 x for nrhs vectors of n (stride n), with the factor the last hiopamd_ldlt_bk_factor left in A
int hiopamd_ldlt_bk_solve(hiopamd_ldlt_bk* B, const double* A, int64_t lda, double* x_inout, int nrhs)
{
  if(!B || nrhs < 0 || (B->n > 0 && nrhs > 0 && (!A || !x_inout))) return HIOPAMD_ERR_ARG;
  if(!B->factored) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = B->ctx;
  const int n = B->n;
  if(n == 0) return HIOPAMD_OK;
  hipStream_t s = ctx->stream;
  const dim3 gn((n + kBlock - 1) / kBlock), bn(kBlock);
  for(int q = 0; q < nrhs; ++q) {
    double* x = x_inout + (int64_t)q * n;
    double* v = B->tmp;
    hipLaunchKernelGGL(bk_gather_kernel, gn, bn, 0, s, n, B->perm, x, v, 0);
    const bool same = B->sw_A == A && B->sw_lda == lda && B->sw_work == ctx->d_work;
    if(n < 1024) {   // a few dozen launches: not worth a graph
      const int rc = bk_enqueue_sweeps(B, A, lda);
      if(rc != HIOPAMD_OK) return rc;
    } else if(B->sweeps && same) {
      HIOPAMD_CHECK(hipGraphLaunch(B->sweeps, s));
    } else if(B->sw_warm && same) {
      // second solve on this matrix address: capture the launch sequence (nothing executes), instantiate, replay
      if(B->sweeps) {
        (void)hipGraphExecDestroy(B->sweeps);
        B->sweeps = nullptr;
      }
      hipGraph_t g = nullptr;
      HIOPAMD_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      const int rc = bk_enqueue_sweeps(B, A, lda);
      const hipError_t ec = hipStreamEndCapture(s, &g);
      if(rc != HIOPAMD_OK || ec != hipSuccess || !g) {
        if(g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc != HIOPAMD_OK ? rc : HIOPAMD_ERR_HIP;
      }
      const hipError_t ei = hipGraphInstantiate(&B->sweeps, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if(ei != hipSuccess) {
        B->sweeps = nullptr;
        return HIOPAMD_ERR_HIP;
      }
      HIOPAMD_CHECK(hipGraphLaunch(B->sweeps, s));
    } else {
      if(B->sweeps) {
        (void)hipGraphExecDestroy(B->sweeps);
        B->sweeps = nullptr;
      }
      const int rc = bk_enqueue_sweeps(B, A, lda);
      if(rc != HIOPAMD_OK) return rc;
      B->sw_A = A;
      B->sw_lda = lda;
      B->sw_work = ctx->d_work;   // (read AFTER the eager run: it has grown the workspace to what the sweeps need)
      B->sw_warm = true;
    }
    hipLaunchKernelGGL(bk_gather_kernel, gn, bn, 0, s, n, B->perm, v, x, 1);
  }
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

// the pivots of the last factorisation, for tests: ipiv (LAPACK's IPIV, 1-based, negative for 2 x 2), perm, e -- host arrays of n
int hiopamd_ldlt_bk_pivots(hiopamd_ldlt_bk* B, int* ipiv_host, int* perm_host, double* e_host)
{
  if(!B) return HIOPAMD_ERR_ARG;
  if(!B->factored) return HIOPAMD_ERR_STATE;
  hipStream_t s = B->ctx->stream;
  const size_t n = (size_t)B->n;
  if(n == 0) return HIOPAMD_OK;
  if(ipiv_host) HIOPAMD_CHECK(hipMemcpyAsync(ipiv_host, B->ipiv, sizeof(int) * n, hipMemcpyDeviceToHost, s));
  if(perm_host) HIOPAMD_CHECK(hipMemcpyAsync(perm_host, B->perm, sizeof(int) * n, hipMemcpyDeviceToHost, s));
  if(e_host) HIOPAMD_CHECK(hipMemcpyAsync(e_host, B->e, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIOPAMD_CHECK(hipStreamSynchronize(s));
  return HIOPAMD_OK;
}

}  // extern "C"
