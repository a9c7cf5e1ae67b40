// hiopMatrixDense (row-major) kernels for gfx950.
//
// reference: src/LinAlg/hiopMatrixDenseRowMajor.cpp (method line numbers are cited on the entry
// points in include/hiop_amd.h).  The matrices on the KKT hot path are either tall-skinny
// (Jacobian k x n_local, k <= ~200, n_local ~ 1e6: GEMV is HBM-bound, one pass over A) or the
// N x N condensed KKT matrix (assembly = streaming scatter of dense blocks).  Layout is the
// reference's: row-major, rows contiguous, so every kernel walks a row with consecutive lanes
// (512 B / wave / instruction) and tiles rows so the x / y vector is re-used from registers/L2.
#include "device_utils.hpp"
#include "dense_internal.hpp"

namespace hiopamd {

// ------------------------------------------------------------------------------------------
// y = beta*y + alpha*A*x      A: m x n row-major.
// stage 1: block (cx, ry) owns ROWS_PER_BLOCK rows x COLS_PER_BLOCK columns, writes partial sums
//          part[cx][row];  stage 2 folds the column chunks in index order (deterministic).
// ------------------------------------------------------------------------------------------
constexpr int GEMV_ROWS = 8;
constexpr int GEMV_COLS_PER_THREAD = 8;
constexpr int GEMV_CHUNKS = 4;                                        // column chunks one block walks before it reduces
constexpr int GEMV_COLS = kBlock * GEMV_COLS_PER_THREAD * GEMV_CHUNKS;  // 8192 columns per block

// Round 3: the counters of the round-2 kernel (profiles/r03_pmc_dense: SQ_WAIT_INST_LDS = 72 % of its busy cycles) showed the
// cross-lane reduction, not the stream over A, as its largest cost: 8 rows x 6 shuffle steps x 2 ds_bpermute per thread after only
// 64 loads.  Now a block walks GEMV_CHUNKS column chunks before it reduces, rows are read with 16-byte loads when the layout
// allows, and the wave reduction halves the number of live sums at every step (lanes l and l ^ 32 split the 8 rows 4 / 4, then 2 / 2,
// then 1 / 1: 4 + 2 + 1 + 3 exchanges instead of 48).
// Row groups of one launch (dense_internal.hpp, gemv_n_groups): group g owns the row tiles [tile0[g], tile0[g+1]) and the rows
// row0[g] .. row0[g] + m[g] - 1 of the result.  hiopamd_mat_times_vec is the one-group case.
struct GemvGroups {
  const double* A[3];
  int m[3], tile0[3], row0[3];
  int ngroups, m_total;
};
// XS: x is multiplied by a second vector on its way in (the product is rounded like the stored vector it replaces)
template <bool VEC, bool XS>
__global__ __launch_bounds__(kBlock) void gemv_n_stage1(const GemvGroups G, int64_t n, int64_t lda, const double* __restrict__ x,
                                                        const double* __restrict__ xscale, double* __restrict__ part, int chunks, int nchunks,
                                                        int rtiles)
{
  // Block -> (column chunk group bx, row tile by), XCD-aware (round 5).  x is read by every row tile: with the grid (chunk groups, row
  // tiles) of rounds 2-4 the k / 8 row tiles of one chunk group ran far apart in time and each re-fetched its 64 KB of x — 250 MB of
  // 2 250 at k = 200, n = 1.25e6 (PMC: 1.12 x the algorithmic bytes, profiles/r05_pmc_dense/summary.json).  The dispatcher deals
  // consecutive workgroups round-robin over the 8 XCDs; here XCD q takes the chunk groups == q (mod 8) and walks the row tiles of one
  // chunk group back to back, so that piece of x is fetched once per XCD-resident pass and hit in L2 by the other row tiles.  The partial
  // sums and their order are those of the old grid: same bits.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bx = (slot / rtiles) * 8 + xcd, by = slot % rtiles;
  if(bx >= nchunks) return;
  int g = 0;
  if(G.ngroups > 1 && by >= G.tile0[1]) g = 1;
  if(G.ngroups > 2 && by >= G.tile0[2]) g = 2;
  const double* __restrict__ A = G.A[g];
  const int m = G.m[g];
  const int r0 = (by - G.tile0[g]) * GEMV_ROWS;
  double acc[GEMV_ROWS];
#pragma unroll
  for(int r = 0; r < GEMV_ROWS; ++r) acc[r] = 0.0;
  for(int ch = 0; ch < chunks; ++ch) {
    const int64_t c0 = ((int64_t)bx * chunks + ch) * (kBlock * GEMV_COLS_PER_THREAD);
    if(c0 >= n) break;
    if constexpr(VEC) {
      // Interior blocks (all GEMV_ROWS rows, a full chunk of columns): no guards, so the 4 + 32 sixteen-byte loads of a chunk are
      // issued back to back and are all in flight together.  (Round 4: with the guarded form below the compiler put every load in
      // its own exec-masked branch with an `s_waitcnt vmcnt(0)` behind it -- ONE load in flight per lane; 5.0 TB/s at k = 200,
      // n = 1.25e6 came from occupancy alone.)
      // (Round 6: a row tile at the end of its group with fewer than GEMV_ROWS rows takes this path as well — the missing rows are read
      //  from the group's last row and their sums are never stored — instead of the guarded form below: the l <= 8 rows of the secant
      //  blocks are ONE such tile.)
      if(c0 + (int64_t)kBlock * GEMV_COLS_PER_THREAD <= n) {
        const double* xb = x + c0 + 2 * threadIdx.x;
        const double* Ab = A + (int64_t)r0 * lda + c0 + 2 * threadIdx.x;
        const int rlast = m - 1 - r0;   // >= 0: the tile has at least one row
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v xv[GEMV_COLS_PER_THREAD / 2], av[GEMV_ROWS][GEMV_COLS_PER_THREAD / 2];
#pragma unroll
        for(int u = 0; u < GEMV_COLS_PER_THREAD / 2; ++u) xv[u] = *reinterpret_cast<const d2v*>(xb + u * 2 * kBlock);
        if constexpr(XS) {
          const double* sb = xscale + c0 + 2 * threadIdx.x;
#pragma unroll
          for(int u = 0; u < GEMV_COLS_PER_THREAD / 2; ++u) {
            const d2v sv = *reinterpret_cast<const d2v*>(sb + u * 2 * kBlock);
            xv[u] = xv[u] * sv;
          }
        }
#pragma unroll
        for(int r = 0; r < GEMV_ROWS; ++r) {
          const int rr = (r <= rlast) ? r : rlast;
#pragma unroll
          for(int u = 0; u < GEMV_COLS_PER_THREAD / 2; ++u)   // A is read once: non-temporal, out of the way of x in L2 (0.353 -> 0.334 ms)
            av[r][u] = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(Ab + (int64_t)rr * lda + u * 2 * kBlock));
        }
#pragma unroll
        for(int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
          for(int u = 0; u < GEMV_COLS_PER_THREAD / 2; ++u) {   // same order of the fused multiply-adds as the guarded form
            acc[r] = fma(av[r][u].x, xv[u].x, acc[r]);
            acc[r] = fma(av[r][u].y, xv[u].y, acc[r]);
          }
        continue;
      }
      // columns c0 + 2 t + 512 u (+0, +1): 16 bytes per lane, 1 KB per wave per instruction (n even or the pair test below)
      double2 xv[GEMV_COLS_PER_THREAD / 2];
#pragma unroll
      for(int u = 0; u < GEMV_COLS_PER_THREAD / 2; ++u) {
        const int64_t j = c0 + 2 * threadIdx.x + (int64_t)u * 2 * kBlock;
        xv[u] = (j + 1 < n) ? *reinterpret_cast<const double2*>(x + j) : double2{(j < n) ? x[j] : 0.0, 0.0};
        if constexpr(XS) {
          const double2 sv = (j + 1 < n) ? *reinterpret_cast<const double2*>(xscale + j) : double2{(j < n) ? xscale[j] : 0.0, 0.0};
          xv[u].x = xv[u].x * sv.x;
          xv[u].y = xv[u].y * sv.y;
        }
      }
#pragma unroll
      for(int r = 0; r < GEMV_ROWS; ++r) {
        const int row = r0 + r;
        if(row < m) {
          const double* Ar = A + (int64_t)row * lda;
#pragma unroll
          for(int u = 0; u < GEMV_COLS_PER_THREAD / 2; ++u) {
            const int64_t j = c0 + 2 * threadIdx.x + (int64_t)u * 2 * kBlock;
            if(j + 1 < n) {
              const double2 a = *reinterpret_cast<const double2*>(Ar + j);
              acc[r] = fma(a.x, xv[u].x, acc[r]);
              acc[r] = fma(a.y, xv[u].y, acc[r]);
            } else if(j < n) {
              acc[r] = fma(Ar[j], xv[u].x, acc[r]);
            }
          }
        }
      }
    } else {
      // the 8-byte form (odd leading dimension or unaligned operands -- the stock MdsEx1 at n_dense = 4097): interior blocks unguarded
      // as well, half the rows at a time (8 + 32 loads in flight); same order of the multiply-adds as the guarded form
      if(c0 + (int64_t)kBlock * GEMV_COLS_PER_THREAD <= n) {
        const double* xb = x + c0 + threadIdx.x;
        const double* Ab = A + (int64_t)r0 * lda + c0 + threadIdx.x;
        const int rlast = m - 1 - r0;
        double xs[GEMV_COLS_PER_THREAD], as[GEMV_ROWS / 2][GEMV_COLS_PER_THREAD];
#pragma unroll
        for(int u = 0; u < GEMV_COLS_PER_THREAD; ++u) xs[u] = xb[u * kBlock];
        if constexpr(XS) {
#pragma unroll
          for(int u = 0; u < GEMV_COLS_PER_THREAD; ++u) {
            const double sv = xscale[c0 + threadIdx.x + u * kBlock];
            xs[u] = xs[u] * sv;
          }
        }
#pragma unroll
        for(int rh = 0; rh < GEMV_ROWS; rh += GEMV_ROWS / 2) {
#pragma unroll
          for(int r = 0; r < GEMV_ROWS / 2; ++r) {
            const int rr = (rh + r <= rlast) ? (rh + r) : rlast;
#pragma unroll
            for(int u = 0; u < GEMV_COLS_PER_THREAD; ++u) as[r][u] = __builtin_nontemporal_load(Ab + (int64_t)rr * lda + u * kBlock);
          }
#pragma unroll
          for(int r = 0; r < GEMV_ROWS / 2; ++r)
#pragma unroll
            for(int u = 0; u < GEMV_COLS_PER_THREAD; ++u) acc[rh + r] = fma(as[r][u], xs[u], acc[rh + r]);
        }
        continue;
      }
      double xv[GEMV_COLS_PER_THREAD];
#pragma unroll
      for(int u = 0; u < GEMV_COLS_PER_THREAD; ++u) {
        const int64_t j = c0 + threadIdx.x + (int64_t)u * kBlock;
        xv[u] = (j < n) ? x[j] : 0.0;
        if constexpr(XS) {
          const double sv = (j < n) ? xscale[j] : 0.0;
          xv[u] = xv[u] * sv;
        }
      }
#pragma unroll
      for(int r = 0; r < GEMV_ROWS; ++r) {
        const int row = r0 + r;
        if(row < m) {
          const double* Ar = A + (int64_t)row * lda;
#pragma unroll
          for(int u = 0; u < GEMV_COLS_PER_THREAD; ++u) {
            const int64_t j = c0 + threadIdx.x + (int64_t)u * kBlock;
            if(j < n) acc[r] = fma(Ar[j], xv[u], acc[r]);
          }
        }
      }
    }
  }
  // wave reduction with halving: after the step with distance D a lane keeps the rows whose bit (log2 of the group) matches its side
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s4[4], s2[2], s1;
  {
    const bool hi = (lane & 32) != 0;
#pragma unroll
    for(int q = 0; q < 4; ++q) {
      const double keep = hi ? acc[4 + q] : acc[q], give = hi ? acc[q] : acc[4 + q];
      s4[q] = keep + __shfl_xor(give, 32, 64);
    }
  }
  {
    const bool hi = (lane & 16) != 0;
#pragma unroll
    for(int q = 0; q < 2; ++q) {
      const double keep = hi ? s4[2 + q] : s4[q], give = hi ? s4[q] : s4[2 + q];
      s2[q] = keep + __shfl_xor(give, 16, 64);
    }
  }
  {
    const bool hi = (lane & 8) != 0;
    const double keep = hi ? s2[1] : s2[0], give = hi ? s2[0] : s2[1];
    s1 = keep + __shfl_xor(give, 8, 64);
  }
  s1 += __shfl_xor(s1, 4, 64);
  s1 += __shfl_xor(s1, 2, 64);
  s1 += __shfl_xor(s1, 1, 64);
  // lane l now holds the wave's sum of row 4 * bit5 + 2 * bit4 + bit3 of l (the same value in its 8 lanes l & ~7 ... | 7)
  __shared__ double sm[GEMV_ROWS][kBlock / 64];
  if((lane & 7) == 0) sm[((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)][wave] = s1;
  __syncthreads();
  if(threadIdx.x < GEMV_ROWS) {
    const int row = r0 + threadIdx.x;
    if(row < m) {
      double v = ((sm[threadIdx.x][0] + sm[threadIdx.x][1]) + sm[threadIdx.x][2]) + sm[threadIdx.x][3];
      part[(int64_t)bx * G.m_total + G.row0[g] + row] = v;
    }
  }
}

// stage 2 of gemv_n_groups: one wave per row of the stacked result (see dense_internal.hpp for what it writes)
__global__ __launch_bounds__(kBlock) void gemv_n_stage2_groups(const GemvGroups G, int nchunks, const double* __restrict__ part,
                                                               double* __restrict__ y, double a0, double a1, double a2,
                                                               const double* __restrict__ sub0, int nsub0, const double* __restrict__ sub1)
{
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if(wave >= G.m_total) return;
  double v = 0.0;
  for(int c = lane; c < nchunks; c += 64) v += part[(int64_t)c * G.m_total + wave];
  for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if(lane == 0) {
    int g = 0;
    if(G.ngroups > 1 && wave >= G.row0[1]) g = 1;
    if(G.ngroups > 2 && wave >= G.row0[2]) g = 2;
    const double alpha = (g == 0) ? a0 : (g == 1) ? a1 : a2;
    double out = alpha * v;
    if(g == 0 && sub0) out = out - ((wave < nsub0) ? sub0[wave] : sub1[wave - nsub0]);
    y[wave] = out;
  }
}

// (Folding in the LAST stage-1 workgroup of a row tile to arrive -- one launch instead of two -- was measured in round 4 and is slower:
// with a release fence per workgroup 0.54 ms instead of 0.33 at 200 x 1.25e6 (every fence walks the L2), with agent-scope relaxed atomics
// for partials and counter 0.343 against 0.333, the dense step 5.69 against 5.60 ms; scripts/calls/r04_gpu_18.sh.)
__global__ __launch_bounds__(kBlock) void gemv_n_stage2(int m, int nchunks, const double* __restrict__ part,
                                                        double beta, double* __restrict__ y, double alpha)
{
  // one wave per row: lanes fold chunks lane, lane+64, ... then shuffle tree
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if(wave >= m) return;
  double v = 0.0;
  for(int c = lane; c < nchunks; c += 64) v += part[(int64_t)c * m + wave];
  for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if(lane == 0) {
    // reference (DGEMV semantics): beta==0 overwrites (no NaN propagation from y)
    y[wave] = (beta == 0.0 ? 0.0 : beta * y[wave]) + alpha * v;
  }
}

// ------------------------------------------------------------------------------------------
// y = beta*y + alpha*A^T*x    A: m x n row-major ; y has n entries.
// thread per column (coalesced along rows); rows split over blockIdx.y when m is large.
// ------------------------------------------------------------------------------------------
constexpr int GEMVT_ROWCHUNK = 64;

// CP: column pairs per thread (pair p of thread t = columns base + 512 p + 2 t, +1)
// the per-column epilogue of gemv_t_tail (dense_internal.hpp): yj is the finished y_j; sy_sh = the 2 l coefficients in LDS.  The 2 l
// entries of the secant rows are requested eight rows at a time (a loop over a run-time l keeps ONE load in flight per lane: the tail then
// cost 55 us of a 350 us pass at n = 1.25e6, l = 6).
__device__ __forceinline__ void gemvt_tail_store(const GemvtTail& T, const double* sy_sh, int64_t j, double yj)
{
  double s1 = 0.0, s2 = 0.0;
  for(int q0 = 0; q0 < T.l; q0 += 8) {
    double sv[8], yv[8];
#pragma unroll
    for(int u = 0; u < 8; ++u) {
      const int q = (q0 + u < T.l) ? (q0 + u) : (T.l - 1);   // (clamped: the loads stay unconditional, the extra ones are not used)
      sv[u] = T.S[(int64_t)q * T.ld + j];
      yv[u] = T.Y[(int64_t)q * T.ld + j];
    }
#pragma unroll
    for(int u = 0; u < 8; ++u)
      if(q0 + u < T.l) s1 = fma(sv[u], sy_sh[q0 + u], s1);
#pragma unroll
    for(int u = 0; u < 8; ++u)
      if(q0 + u < T.l) s2 = fma(yv[u], sy_sh[T.l + q0 + u], s2);
  }
  double res = T.sigma * s1;
  res = res + s2;
  const double di = T.DhInv[j];
  const double a = yj * di, b = res * di;
  T.dx[j] = a - b;
}
// two adjacent columns j, j + 1 (j even, every row 16-byte aligned): one 16-byte load per secant row
__device__ __forceinline__ void gemvt_tail_store2(const GemvtTail& T, const double* sy_sh, int64_t j, double y0, double y1)
{
  double s10 = 0.0, s11 = 0.0, s20 = 0.0, s21 = 0.0;
  for(int q0 = 0; q0 < T.l; q0 += 8) {
    double2 sv[8], yv[8];
#pragma unroll
    for(int u = 0; u < 8; ++u) {
      const int q = (q0 + u < T.l) ? (q0 + u) : (T.l - 1);
      sv[u] = *reinterpret_cast<const double2*>(T.S + (int64_t)q * T.ld + j);
      yv[u] = *reinterpret_cast<const double2*>(T.Y + (int64_t)q * T.ld + j);
    }
#pragma unroll
    for(int u = 0; u < 8; ++u)
      if(q0 + u < T.l) {
        s10 = fma(sv[u].x, sy_sh[q0 + u], s10);
        s11 = fma(sv[u].y, sy_sh[q0 + u], s11);
      }
#pragma unroll
    for(int u = 0; u < 8; ++u)
      if(q0 + u < T.l) {
        s20 = fma(yv[u].x, sy_sh[T.l + q0 + u], s20);
        s21 = fma(yv[u].y, sy_sh[T.l + q0 + u], s21);
      }
  }
  const double2 di = *reinterpret_cast<const double2*>(T.DhInv + j);
  double r0 = T.sigma * s10, r1 = T.sigma * s11;
  r0 = r0 + s20;
  r1 = r1 + s21;
  const double a0 = y0 * di.x, b0 = r0 * di.x, a1 = y1 * di.y, b1 = r1 * di.y;
  *reinterpret_cast<double2*>(T.dx + j) = double2{a0 - b0, a1 - b1};
}

template <int CP, bool TAIL = false>
__global__ __launch_bounds__(kBlock) void gemv_t_kernel(int m, int64_t n, const double* __restrict__ A, int64_t lda,
                                                        const double* __restrict__ x, int rows_per_split,
                                                        double* __restrict__ out, int64_t out_stride, double beta,
                                                        double alpha, int direct, const GemvtTail T = GemvtTail())
{
  __shared__ double sy_sh[TAIL ? 128 : 1];
  if constexpr(TAIL) {
    if(threadIdx.x < 2 * T.l) sy_sh[threadIdx.x] = T.sy[threadIdx.x];
    // (visible behind the first __syncthreads() of the row loop below: m > 0 on this path)
  }
  // 16-byte accesses to the secant rows, DhInv and dx (the columns of a pair are j0 = even, j0 + 1)
  const bool tail_vec = TAIL && ((T.ld & 1) == 0) && ((((uintptr_t)T.S) | ((uintptr_t)T.Y) | ((uintptr_t)T.DhInv) | ((uintptr_t)T.dx)) & 15) == 0;
  const int r_begin = blockIdx.y * rows_per_split;
  int r_end = r_begin + rows_per_split;
  if(r_end > m) r_end = m;
  __shared__ double xs[GEMVT_ROWCHUNK];
  const int64_t jb = (int64_t)blockIdx.x * (2 * kBlock * CP) + 2 * threadIdx.x;
  double a0[CP], a1[CP];
#pragma unroll
  for(int p = 0; p < CP; ++p) a0[p] = a1[p] = 0.0;
  const bool full = (jb + (int64_t)(CP - 1) * 2 * kBlock + 1 < n) && ((lda & 1) == 0) && ((((uintptr_t)A) & 15) == 0);
  for(int rb = r_begin; rb < r_end; rb += GEMVT_ROWCHUNK) {
    int rc = r_end - rb;
    if(rc > GEMVT_ROWCHUNK) rc = GEMVT_ROWCHUNK;
    __syncthreads();
    if(threadIdx.x < rc) xs[threadIdx.x] = x[rb + threadIdx.x];
    __syncthreads();
    if(full) {
      const double* Ap = A + (int64_t)rb * lda + jb;
#pragma unroll(8 / CP)
      for(int r = 0; r < rc; ++r) {
#pragma unroll
        for(int p = 0; p < CP; ++p) {
          const double2 v = *reinterpret_cast<const double2*>(Ap + (int64_t)r * lda + p * 2 * kBlock);
          a0[p] = fma(v.x, xs[r], a0[p]);
          a1[p] = fma(v.y, xs[r], a1[p]);
        }
      }
    } else {
#pragma unroll
      for(int p = 0; p < CP; ++p) {
        const int64_t j0 = jb + (int64_t)p * 2 * kBlock;
        if(j0 >= n) continue;
        const double* Ap = A + (int64_t)rb * lda + j0;
        if(j0 + 1 < n) {   // both columns inside: eight rows = sixteen 8-byte loads in flight (odd leading dimension: no 16-byte loads)
          int r = 0;
          for(; r + 8 <= rc; r += 8) {
            double u0[8], u1[8];
#pragma unroll
            for(int q = 0; q < 8; ++q) {
              u0[q] = Ap[(int64_t)(r + q) * lda];
              u1[q] = Ap[(int64_t)(r + q) * lda + 1];
            }
#pragma unroll
            for(int q = 0; q < 8; ++q) {
              a0[p] = fma(u0[q], xs[r + q], a0[p]);
              a1[p] = fma(u1[q], xs[r + q], a1[p]);
            }
          }
          for(; r < rc; ++r) {
            a0[p] = fma(Ap[(int64_t)r * lda], xs[r], a0[p]);
            a1[p] = fma(Ap[(int64_t)r * lda + 1], xs[r], a1[p]);
          }
        } else {
          for(int r = 0; r < rc; ++r) a0[p] = fma(Ap[(int64_t)r * lda], xs[r], a0[p]);
        }
      }
    }
  }
#pragma unroll
  for(int p = 0; p < CP; ++p) {
    const int64_t j0 = jb + (int64_t)p * 2 * kBlock;
    if(j0 >= n) continue;
    if(direct) {
      const double v0 = (beta == 0.0 ? 0.0 : beta * out[j0]) + alpha * a0[p];
      out[j0] = v0;
      double v1 = 0.0;
      if(j0 + 1 < n) {
        v1 = (beta == 0.0 ? 0.0 : beta * out[j0 + 1]) + alpha * a1[p];
        out[j0 + 1] = v1;
      }
      if constexpr(TAIL) {
        if(tail_vec && j0 + 1 < n) {
          gemvt_tail_store2(T, sy_sh, j0, v0, v1);
        } else {
          gemvt_tail_store(T, sy_sh, j0, v0);
          if(j0 + 1 < n) gemvt_tail_store(T, sy_sh, j0 + 1, v1);
        }
      }
    } else {
      double* o = out + (int64_t)blockIdx.y * out_stride;
      o[j0] = a0[p];
      if(j0 + 1 < n) o[j0 + 1] = a1[p];
    }
  }
}

template <bool TAIL = false>
__global__ __launch_bounds__(kBlock) void gemv_t_fold(int64_t n, int nsplit, const double* __restrict__ part,
                                                      int64_t stride, double beta, double* __restrict__ y, double alpha,
                                                      const GemvtTail T = GemvtTail())
{
  __shared__ double sy_sh[TAIL ? 128 : 1];
  if constexpr(TAIL) {
    if(threadIdx.x < 2 * T.l) sy_sh[threadIdx.x] = T.sy[threadIdx.x];
    __syncthreads();
  }
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if(j >= n) return;
  double v = 0.0;
  for(int s = 0; s < nsplit; ++s) v += part[(int64_t)s * stride + j];
  const double yj = (beta == 0.0 ? 0.0 : beta * y[j]) + alpha * v;
  y[j] = yj;
  if constexpr(TAIL) gemvt_tail_store(T, sy_sh, j, yj);
}

// ------------------------------------------------------------------------------------------
// small generic GEMM: C(M x N) = beta*C + alpha*opA(M x K)*opB(K x N), arbitrary element strides.
// Used for the l x l / k x 2l / 2l x k products of the low-rank KKT (reference DGEMM call sites
// hiopMatrixDenseRowMajor.cpp:601,640,673).  LDS-tiled 16x16, one output per thread.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void small_gemm(int M, int N, int K, const double* __restrict__ A, int64_t a_rs,
                                                  int64_t a_cs, const double* __restrict__ B, int64_t b_rs,
                                                  int64_t b_cs, double beta, double* __restrict__ C, int64_t ldc,
                                                  double alpha)
{
  __shared__ double As[16][17];
  __shared__ double Bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
  double acc = 0.0;
  for(int k0 = 0; k0 < K; k0 += 16) {
    As[ty][tx] = (row < M && k0 + tx < K) ? A[(int64_t)row * a_rs + (int64_t)(k0 + tx) * a_cs] : 0.0;
    Bs[ty][tx] = (k0 + ty < K && col < N) ? B[(int64_t)(k0 + ty) * b_rs + (int64_t)col * b_cs] : 0.0;
    __syncthreads();
#pragma unroll
    for(int kk = 0; kk < 16; ++kk) acc = fma(As[ty][kk], Bs[kk][tx], acc);
    __syncthreads();
  }
  if(row < M && col < N) {
    double* c = C + (int64_t)row * ldc + col;
    *c = (beta == 0.0 ? 0.0 : beta * (*c)) + alpha * acc;
  }
}

// The same product on the matrix pipe (round 4; the reference's DGEMM call sites :578-673 -- timesMat, transTimesMat, timesMatTrans):
// workgroup tile 64 x 64 (2 x 2 waves of 2 x 2 v_mfma_f64_16x16x4_f64 tiles), K in LDS stages of 16 (zero-filled past K, M, N), any
// element strides: a stage is read along whichever index of the operand is contiguous.  Lane map of the instruction:
// A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[row = (l >> 4) + 4 reg][col = l & 15].
typedef double gemm_double4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gemm_mfma_kernel(int M, int N, int K, const double* __restrict__ A, int64_t a_rs, int64_t a_cs,
                                                        const double* __restrict__ B, int64_t b_rs, int64_t b_cs, double beta,
                                                        double* __restrict__ C, int64_t ldc, double alpha)
{
  constexpr int T = 64, KT = 16, LD = T + 4;   // (row stride 68 doubles: the four k-rows of an operand read fall on different banks)
  __shared__ double As[KT][LD];   // As[k][i]
  __shared__ double Bs[KT][LD];   // Bs[k][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lk = lane >> 4, li = lane & 15;
  const int r0 = blockIdx.y * T, c0 = blockIdx.x * T;
  gemm_double4 acc[2][2];
#pragma unroll
  for(int i = 0; i < 2; ++i)
#pragma unroll
    for(int j = 0; j < 2; ++j) acc[i][j] = gemm_double4{0.0, 0.0, 0.0, 0.0};
  const bool a_k_fast = (a_cs == 1), b_k_fast = (b_rs == 1);
  // this thread's four elements of each operand stage: (ii, ka) of A, (kb, jj) of B; the loads are unconditional on clamped
  // addresses (a guarded load gets its own branch and its own wait from the compiler: one load in flight), zero-filled afterwards,
  // and the next stage is fetched into registers while the current one is multiplied
  int ai[4], ak[4], bj[4], bk[4];
#pragma unroll
  for(int q = 0; q < 4; ++q) {
    if(a_k_fast) { ak[q] = tid & 15; ai[q] = (tid >> 4) + 16 * q; } else { ai[q] = tid & 63; ak[q] = (tid >> 6) + 4 * q; }
    if(b_k_fast) { bk[q] = tid & 15; bj[q] = (tid >> 4) + 16 * q; } else { bj[q] = tid & 63; bk[q] = (tid >> 6) + 4 * q; }
  }
  double va[4], vb[4];
  auto issue = [&](int k0) {   // the stage's eight loads, all in flight
#pragma unroll
    for(int q = 0; q < 4; ++q) {
      const int ir = r0 + ai[q], kr = k0 + ak[q], jc = c0 + bj[q], kc = k0 + bk[q];
      va[q] = A[(int64_t)(ir < M ? ir : M - 1) * a_rs + (int64_t)(kr < K ? kr : K - 1) * a_cs];
      vb[q] = B[(int64_t)(kc < K ? kc : K - 1) * b_rs + (int64_t)(jc < N ? jc : N - 1) * b_cs];
    }
  };
  auto commit = [&](int k0) {   // zero fill + into LDS (the empty asm keeps the compiler from sinking a load into its select's branch)
#pragma unroll
    for(int q = 0; q < 4; ++q) {
      asm volatile("" : "+v"(va[q]), "+v"(vb[q]));
      As[ak[q]][ai[q]] = (r0 + ai[q] < M && k0 + ak[q] < K) ? va[q] : 0.0;
      Bs[bk[q]][bj[q]] = (c0 + bj[q] < N && k0 + bk[q] < K) ? vb[q] : 0.0;
    }
  };
  if(K > 0) issue(0);
  for(int k0 = 0; k0 < K; k0 += KT) {
    commit(k0);
    __syncthreads();
    if(k0 + KT < K) issue(k0 + KT);
#pragma unroll
    for(int k4 = 0; k4 < KT / 4; ++k4) {
      double a[2], b[2];
#pragma unroll
      for(int i = 0; i < 2; ++i) a[i] = As[k4 * 4 + lk][wr * 32 + i * 16 + li];
#pragma unroll
      for(int j = 0; j < 2; ++j) b[j] = Bs[k4 * 4 + lk][wc * 32 + j * 16 + li];
#pragma unroll
      for(int i = 0; i < 2; ++i)
#pragma unroll
        for(int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for(int i = 0; i < 2; ++i)
#pragma unroll
    for(int j = 0; j < 2; ++j)
#pragma unroll
      for(int reg = 0; reg < 4; ++reg) {
        const int row = r0 + wr * 32 + i * 16 + lk + 4 * reg, col = c0 + wc * 32 + j * 16 + li;
        if(row < M && col < N) {
          double* c = C + (int64_t)row * ldc + col;
          *c = (beta == 0.0 ? 0.0 : beta * (*c)) + alpha * acc[i][j][reg];
        }
      }
}

// ------------------------------------------------------------------------------------------
// assembly kernels
// ------------------------------------------------------------------------------------------
// W[row_start + jc][col_start + ir] += alpha*A[ir][jc]   (32x32 LDS transpose, coalesced both sides)
__global__ __launch_bounds__(256) void trans_add_kernel(int m, int n, const double* __restrict__ A, int64_t lda,
                                                        int row_start, int col_start, double alpha,
                                                        double* __restrict__ W, int64_t ldw)
{
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int ir0 = blockIdx.y * 32, jc0 = blockIdx.x * 32;
#pragma unroll
  for(int r = ty; r < 32; r += 8) {
    int ir = ir0 + r, jc = jc0 + tx;
    tile[r][tx] = (ir < m && jc < n) ? A[(int64_t)ir * lda + jc] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for(int r = ty; r < 32; r += 8) {
    int jc = jc0 + r, ir = ir0 + tx;  // W row = jc, W col = ir
    if(ir < m && jc < n) {
      double* w = W + (int64_t)(row_start + jc) * ldw + (col_start + ir);
      *w += alpha * tile[tx][r];
    }
  }
}

__global__ __launch_bounds__(kBlock) void add_upper_kernel(int n, const double* __restrict__ A, int64_t lda,
                                                           int diag_start, double alpha, double* __restrict__ W,
                                                           int64_t ldw)
{
  const int i = blockIdx.y;
  const double* Ar = A + (int64_t)i * lda;
  double* Wr = W + (int64_t)(i + diag_start) * ldw + diag_start;
  for(int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    if(j >= i) Wr[j] += alpha * Ar[j];
  }
}

template <class F>
__global__ __launch_bounds__(kBlock) void mat_ew_kernel(int m, int64_t n, F f)
{
  // 2-D element-wise: blockIdx.y strides rows, x strides columns
  for(int i = blockIdx.y; i < m; i += gridDim.y) {
    for(int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) f(i, j);
  }
}
template <class F>
static inline int launch_mat_ew(hiopamd_ctx* ctx, int m, int64_t n, F f)
{
  if(m < 0 || n < 0) return HIOPAMD_ERR_ARG;
  if(m == 0 || n == 0) return HIOPAMD_OK;
  int gx = (int)((n + kBlock - 1) / kBlock);
  if(gx > 1024) gx = 1024;
  int gy = m;
  while((int64_t)gx * gy > 16384 && gy > 1) gy = (gy + 1) / 2;
  hipLaunchKernelGGL(mat_ew_kernel<F>, dim3(gx, gy), dim3(kBlock), 0, ctx->stream, m, n, f);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

struct OpMatAbsMax {
  const double* A;
  int64_t lda, n;
  __device__ double identity() const { return 0.0; }
  __device__ double map(int64_t t) const
  {
    int64_t i = t / n, j = t - i * n;
    return fabs(A[i * lda + j]);
  }
  __device__ double combine(double a, double b) const { return (b > a) ? b : a; }
};
struct OpMatNotFinite {
  const double* A;
  int64_t lda, n;
  __device__ double identity() const { return 0.0; }
  __device__ double map(int64_t t) const
  {
    int64_t i = t / n, j = t - i * n;
    return isfinite(A[i * lda + j]) ? 0.0 : 1.0;
  }
  __device__ double combine(double a, double b) const { return a + b; }
};

// one block per row: max |A[i,:]|
__global__ __launch_bounds__(kBlock) void row_max_abs_kernel(int64_t n, const double* __restrict__ A, int64_t lda,
                                                             double* __restrict__ out)
{
  const double* Ar = A + (int64_t)blockIdx.x * lda;
  double v = 0.0;
  for(int64_t j = threadIdx.x; j < n; j += kBlock) {
    double a = fabs(Ar[j]);
    if(a > v) v = a;
  }
  for(int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    if(o > v) v = o;
  }
  __shared__ double sm[kBlock / 64];
  if((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if(threadIdx.x == 0) {
    for(int w = 1; w < kBlock / 64; ++w)
      if(sm[w] > v) v = sm[w];
    out[blockIdx.x] = v;
  }
}

}  // namespace hiopamd

using namespace hiopamd;

extern "C" {

int hiopamd_mat_set_to_constant(hiopamd_ctx* ctx, int m, int64_t n, double* A, int64_t lda, double c)
{
  if(lda == n) return hiopamd_vec_set_to_constant(ctx, (int64_t)m * n, A, c);
  return launch_mat_ew(ctx, m, n, [=] __device__(int i, int64_t j) { A[(int64_t)i * lda + j] = c; });
}

int hiopamd_mat_times_vec(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double beta, double* y,
                          double alpha, const double* x)
{
  if(m < 0 || n < 0) return HIOPAMD_ERR_ARG;
  if(m == 0) return HIOPAMD_OK;
  if(n == 0) return hiopamd_vec_scale(ctx, m, y, beta);
  // column chunks one block walks before it reduces: 4 for the tall-skinny Jacobians (k >= 100 rows: thousands of blocks anyway), fewer
  // when there are few row tiles (the l x n secant blocks, l <= 8: ONE row tile -- 153 blocks at n = 1.25e6 with 4 chunks, 611 with 1)
  const int rtiles = (m + GEMV_ROWS - 1) / GEMV_ROWS;
  int chunks = GEMV_CHUNKS;
  while(chunks > 1 && (int64_t)rtiles * ((n + (int64_t)kBlock * GEMV_COLS_PER_THREAD * chunks - 1) / ((int64_t)kBlock * GEMV_COLS_PER_THREAD * chunks)) < 2048)
    chunks >>= 1;
  const int64_t cols = (int64_t)kBlock * GEMV_COLS_PER_THREAD * chunks;
  const int nchunks = (int)((n + cols - 1) / cols);
  double* part = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nchunks * m);
  const bool vec = (lda % 2 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)x % 16 == 0);
  const dim3 g1((unsigned)(8 * ((nchunks + 7) / 8)) * (unsigned)rtiles), b1(kBlock);   // (see the block map in the kernel)
  GemvGroups G;
  G.A[0] = A; G.A[1] = G.A[2] = nullptr;
  G.m[0] = m; G.m[1] = G.m[2] = 0;
  G.tile0[0] = G.tile0[1] = G.tile0[2] = 0;
  G.row0[0] = G.row0[1] = G.row0[2] = 0;
  G.ngroups = 1;
  G.m_total = m;
  if(vec) hipLaunchKernelGGL((gemv_n_stage1<true, false>), g1, b1, 0, ctx->stream, G, n, lda, x, (const double*)nullptr, part, chunks, nchunks, rtiles);
  else hipLaunchKernelGGL((gemv_n_stage1<false, false>), g1, b1, 0, ctx->stream, G, n, lda, x, (const double*)nullptr, part, chunks, nchunks, rtiles);
  const int waves_per_block = kBlock / 64;
  hipLaunchKernelGGL(gemv_n_stage2, dim3((m + waves_per_block - 1) / waves_per_block), dim3(kBlock), 0, ctx->stream, m,
                     nchunks, part, beta, y, alpha);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

static int gemv_t_impl(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double beta, double* y, double alpha, const double* x,
                      const GemvtTail* tail);

int hiopamd_mat_trans_times_vec(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double beta,
                                double* y, double alpha, const double* x)
{
  if(m < 0 || n < 0) return HIOPAMD_ERR_ARG;
  if(n == 0) return HIOPAMD_OK;
  if(m == 0) return hiopamd_vec_scale(ctx, n, y, beta);
  return gemv_t_impl(ctx, m, n, A, lda, beta, y, alpha, x, nullptr);
}

}  // extern "C"  (the two library-internal forms below have C++ linkage)

namespace hiopamd {
int gemv_t_tail(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double* y, double alpha, const double* x, const GemvtTail& tail)
{
  if(m <= 0 || n < 0 || tail.l < 0 || 2 * tail.l > 128) return HIOPAMD_ERR_ARG;
  if(n == 0) return HIOPAMD_OK;
  return gemv_t_impl(ctx, m, n, A, lda, 1.0, y, alpha, x, &tail);
}

int gemv_n_groups(hiopamd_ctx* ctx, int64_t n, int64_t lda, int ngroups, const double* const* A, const int* m, const double* x,
                  const double* xscale, double* y, const double* alpha, const double* sub0, int nsub0, const double* sub1)
{
  if(ngroups < 1 || ngroups > 3 || n <= 0) return HIOPAMD_ERR_ARG;
  GemvGroups G;
  int rtiles = 0, mt = 0;
  bool vec = (lda % 2 == 0) && ((uintptr_t)x % 16 == 0) && (!xscale || (uintptr_t)xscale % 16 == 0);
  for(int g = 0; g < 3; ++g) {
    G.A[g] = (g < ngroups) ? A[g] : nullptr;
    G.m[g] = (g < ngroups) ? m[g] : 0;
    G.tile0[g] = rtiles;
    G.row0[g] = mt;
    if(g < ngroups) {
      if(m[g] <= 0) return HIOPAMD_ERR_ARG;   // (callers leave empty groups out)
      rtiles += (m[g] + GEMV_ROWS - 1) / GEMV_ROWS;
      mt += m[g];
      vec = vec && ((uintptr_t)A[g] % 16 == 0);
    }
  }
  G.ngroups = ngroups;
  G.m_total = mt;
  int chunks = GEMV_CHUNKS;
  while(chunks > 1 && (int64_t)rtiles * ((n + (int64_t)kBlock * GEMV_COLS_PER_THREAD * chunks - 1) / ((int64_t)kBlock * GEMV_COLS_PER_THREAD * chunks)) < 2048)
    chunks >>= 1;
  const int64_t cols = (int64_t)kBlock * GEMV_COLS_PER_THREAD * chunks;
  const int nchunks = (int)((n + cols - 1) / cols);
  double* part = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nchunks * mt);
  const dim3 g1((unsigned)(8 * ((nchunks + 7) / 8)) * (unsigned)rtiles), b1(kBlock);
  if(vec && xscale) hipLaunchKernelGGL((gemv_n_stage1<true, true>), g1, b1, 0, ctx->stream, G, n, lda, x, xscale, part, chunks, nchunks, rtiles);
  else if(vec) hipLaunchKernelGGL((gemv_n_stage1<true, false>), g1, b1, 0, ctx->stream, G, n, lda, x, xscale, part, chunks, nchunks, rtiles);
  else if(xscale) hipLaunchKernelGGL((gemv_n_stage1<false, true>), g1, b1, 0, ctx->stream, G, n, lda, x, xscale, part, chunks, nchunks, rtiles);
  else hipLaunchKernelGGL((gemv_n_stage1<false, false>), g1, b1, 0, ctx->stream, G, n, lda, x, xscale, part, chunks, nchunks, rtiles);
  const int waves_per_block = kBlock / 64;
  hipLaunchKernelGGL(gemv_n_stage2_groups, dim3((mt + waves_per_block - 1) / waves_per_block), dim3(kBlock), 0, ctx->stream, G, nchunks, part, y,
                     alpha[0], ngroups > 1 ? alpha[1] : 0.0, ngroups > 2 ? alpha[2] : 0.0, sub0, nsub0, sub1);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
}  // namespace hiopamd

extern "C" {

static int gemv_t_impl(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double beta, double* y, double alpha, const double* x,
                      const GemvtTail* tail)
{
  constexpr int cp = 1;   // column pairs per thread (two: 0.41 vs 0.36 ms at k = 200, n = 1.25e6 -- scripts/calls/r04_gpu_13.sh)
  const int gx = (int)((n + 2 * kBlock * cp - 1) / (2 * kBlock * cp));
  // split rows so that the launch has >= ~1024 workgroups when the matrix is not tall-skinny
  int nsplit = 1;
  if(gx < 1024) {
    nsplit = (1024 + gx - 1) / gx;
    int maxsplit = (m + GEMVT_ROWCHUNK - 1) / GEMVT_ROWCHUNK;
    if(nsplit > maxsplit) nsplit = maxsplit;
    if(nsplit < 1) nsplit = 1;
  }
  int rows_per_split = (m + nsplit - 1) / nsplit;
  rows_per_split = ((rows_per_split + GEMVT_ROWCHUNK - 1) / GEMVT_ROWCHUNK) * GEMVT_ROWCHUNK;
  nsplit = (m + rows_per_split - 1) / rows_per_split;
  auto kern = gemv_t_kernel<cp, false>;
  if(nsplit == 1) {
    if(tail) hipLaunchKernelGGL((gemv_t_kernel<cp, true>), dim3(gx, 1), dim3(kBlock), 0, ctx->stream, m, n, A, lda, x, rows_per_split, y,
                                (int64_t)0, beta, alpha, 1, *tail);
    else hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(kBlock), 0, ctx->stream, m, n, A, lda, x, rows_per_split, y,
                            (int64_t)0, beta, alpha, 1, GemvtTail());
  } else {
    double* part = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nsplit * n);
    hipLaunchKernelGGL(kern, dim3(gx, nsplit), dim3(kBlock), 0, ctx->stream, m, n, A, lda, x, rows_per_split,
                       part, n, 0.0, 1.0, 0, GemvtTail());
    if(tail) hipLaunchKernelGGL(gemv_t_fold<true>, dim3((int)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, n, nsplit,
                                part, n, beta, y, alpha, *tail);
    else hipLaunchKernelGGL(gemv_t_fold<false>, dim3((int)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, n, nsplit,
                            part, n, beta, y, alpha, GemvtTail());
  }
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

static int launch_small_gemm(hiopamd_ctx* ctx, int M, int N, int K, const double* A, int64_t a_rs, int64_t a_cs,
                             const double* B, int64_t b_rs, int64_t b_cs, double beta, double* C, int64_t ldc,
                             double alpha)
{
  if(M < 0 || N < 0 || K < 0) return HIOPAMD_ERR_ARG;
  if(M == 0 || N == 0) return HIOPAMD_OK;
  if(M >= 32 && N >= 32)   // fp64 MFMA tiles; outputs narrower than half a tile stay on the scalar kernel
    hipLaunchKernelGGL(gemm_mfma_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, ctx->stream, M, N, K, A, a_rs, a_cs, B, b_rs,
                       b_cs, beta, C, ldc, alpha);
  else
    hipLaunchKernelGGL(small_gemm, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, ctx->stream, M, N, K, A, a_rs, a_cs,
                       B, b_rs, b_cs, beta, C, ldc, alpha);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mat_times_mat(hiopamd_ctx* ctx, int m, int n, int k, const double* A, int64_t lda, double beta, double* W,
                          int64_t ldw, double alpha, const double* X, int64_t ldx)
{
  return launch_small_gemm(ctx, m, k, n, A, lda, 1, X, ldx, 1, beta, W, ldw, alpha);
}
int hiopamd_mat_trans_times_mat(hiopamd_ctx* ctx, int m, int n, int k, const double* A, int64_t lda, double beta,
                                double* W, int64_t ldw, double alpha, const double* X, int64_t ldx)
{
  return launch_small_gemm(ctx, n, k, m, A, 1, lda, X, ldx, 1, beta, W, ldw, alpha);
}
int hiopamd_mat_times_mat_trans(hiopamd_ctx* ctx, int m, int64_t n, int k, const double* A, int64_t lda, double beta,
                                double* W, int64_t ldw, double alpha, const double* X, int64_t ldx)
{
  // A(m x n) * X(k x n)^T over a long n is the (unweighted) Gram kernel on MFMA
  if(n > 4096) return hiopamd_gram_weighted(ctx, m, k, n, A, lda, X, ldx, nullptr, beta, W, ldw, alpha, 0);
  return launch_small_gemm(ctx, m, k, (int)n, A, lda, 1, X, 1, ldx, beta, W, ldw, alpha);
}

int hiopamd_mat_add_sub_diagonal(hiopamd_ctx* ctx, double* A, int64_t lda, int start, double alpha, const double* d,
                                 int src_start, int num)
{
  if(num < 0 || start < 0 || src_start < 0) return HIOPAMD_ERR_ARG;
  return launch_ew(ctx, num, [=] __device__(int64_t i) {
    A[(int64_t)(start + i) * lda + (start + i)] += alpha * d[src_start + i];
  });
}
int hiopamd_mat_add_sub_diagonal_const(hiopamd_ctx* ctx, double* A, int64_t lda, int start, int num, double c)
{
  if(num < 0 || start < 0) return HIOPAMD_ERR_ARG;
  return launch_ew(ctx, num, [=] __device__(int64_t i) { A[(int64_t)(start + i) * lda + (start + i)] += c; });
}
int hiopamd_mat_add_diagonal_vec(hiopamd_ctx* ctx, int n, double* A, int64_t lda, double alpha, const double* d)
{
  return hiopamd_mat_add_sub_diagonal(ctx, A, lda, 0, alpha, d, 0, n);
}
int hiopamd_mat_add_diagonal_const(hiopamd_ctx* ctx, int n, double* A, int64_t lda, double value)
{
  return hiopamd_mat_add_sub_diagonal_const(ctx, A, lda, 0, n, value);
}
int hiopamd_mat_add_matrix(hiopamd_ctx* ctx, int m, int64_t n, double* A, int64_t lda, double alpha, const double* X,
                           int64_t ldx)
{
  return launch_mat_ew(ctx, m, n,
                       [=] __device__(int i, int64_t j) { A[(int64_t)i * lda + j] += alpha * X[(int64_t)i * ldx + j]; });
}

int hiopamd_mat_trans_add_to_sym_upper(hiopamd_ctx* ctx, int m, int n, const double* A, int64_t lda, int row_start,
                                       int col_start, double alpha, double* W, int64_t ldw)
{
  if(m < 0 || n < 0 || row_start < 0 || col_start < 0) return HIOPAMD_ERR_ARG;
  if(m == 0 || n == 0) return HIOPAMD_OK;
  // precondition of the reference (assert iW<=jW): the block maps inside the upper triangle
  if(row_start + n - 1 > col_start) return HIOPAMD_ERR_ARG;
  hipLaunchKernelGGL(trans_add_kernel, dim3((n + 31) / 32, (m + 31) / 32), dim3(256), 0, ctx->stream, m, n, A, lda,
                     row_start, col_start, alpha, W, ldw);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mat_add_upper_to_sym_upper(hiopamd_ctx* ctx, int n, const double* A, int64_t lda, int diag_start,
                                       double alpha, double* W, int64_t ldw)
{
  if(n < 0 || diag_start < 0) return HIOPAMD_ERR_ARG;
  if(n == 0) return HIOPAMD_OK;
  int gx = (n + kBlock - 1) / kBlock;
  if(gx > 8) gx = 8;
  hipLaunchKernelGGL(add_upper_kernel, dim3(gx, n), dim3(kBlock), 0, ctx->stream, n, A, lda, diag_start, alpha, W, ldw);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mat_copy_rows_from(hiopamd_ctx* ctx, int num_rows, int64_t n, double* A, int64_t lda, int row_dest,
                               const double* src, int64_t ldsrc)
{
  if(num_rows <= 0 || n <= 0) return (num_rows < 0 || n < 0) ? HIOPAMD_ERR_ARG : HIOPAMD_OK;
  HIOPAMD_CHECK(hipMemcpy2DAsync(A + (int64_t)row_dest * lda, lda * sizeof(double), src, ldsrc * sizeof(double),
                                 n * sizeof(double), num_rows, hipMemcpyDeviceToDevice, ctx->stream));
  return HIOPAMD_OK;
}
int hiopamd_mat_copy_rows_from_idx(hiopamd_ctx* ctx, int num_rows, int64_t n, double* A, int64_t lda,
                                   const double* src, int64_t ldsrc, const int* rows_idxs)
{
  return launch_mat_ew(ctx, num_rows, n, [=] __device__(int i, int64_t j) {
    A[(int64_t)i * lda + j] = src[(int64_t)rows_idxs[i] * ldsrc + j];
  });
}
int hiopamd_mat_copy_block(hiopamd_ctx* ctx, int m, int n, double* dst, int64_t lddst, const double* src,
                           int64_t ldsrc)
{
  return hiopamd_mat_copy_rows_from(ctx, m, n, dst, lddst, 0, src, ldsrc);
}

int hiopamd_mat_shift_rows(hiopamd_ctx* ctx, int m, int64_t n, double* A, int64_t lda, int shift)
{
  // reference :238 — rows move by `shift` (|shift| < m); vacated rows keep their old content.
  if(shift == 0 || m == 0 || n == 0) return HIOPAMD_OK;
  if(abs(shift) >= m) return HIOPAMD_ERR_ARG;
  const int cnt = m - abs(shift);
  // overlapping row ranges: stage through the workspace (one extra pass over l x n doubles)
  double* tmp = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)cnt * n);
  const double* src = shift > 0 ? A : A + (int64_t)(-shift) * lda;
  double* dst = shift > 0 ? A + (int64_t)shift * lda : A;
  HIOPAMD_CHECK(hipMemcpy2DAsync(tmp, n * sizeof(double), src, lda * sizeof(double), n * sizeof(double), cnt,
                                 hipMemcpyDeviceToDevice, ctx->stream));
  HIOPAMD_CHECK(hipMemcpy2DAsync(dst, lda * sizeof(double), tmp, n * sizeof(double), n * sizeof(double), cnt,
                                 hipMemcpyDeviceToDevice, ctx->stream));
  return HIOPAMD_OK;
}

int hiopamd_mat_max_abs(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double* out)
{
  return launch_reduce<double>(ctx, (int64_t)m * n, OpMatAbsMax{A, lda, n}, out);
}
int hiopamd_mat_is_finite(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, int* out)
{
  double c = 0.0;
  int st = launch_reduce<double>(ctx, (int64_t)m * n, OpMatNotFinite{A, lda, n}, &c);
  *out = (c == 0.0) ? 1 : 0;
  return st;
}
int hiopamd_mat_row_max_abs(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double* ret_vec)
{
  if(m <= 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(row_max_abs_kernel, dim3(m), dim3(kBlock), 0, ctx->stream, n, A, lda, ret_vec);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
int hiopamd_mat_scale_rows(hiopamd_ctx* ctx, int m, int64_t n, double* A, int64_t lda, const double* scal, int inv)
{
  return launch_mat_ew(ctx, m, n, [=] __device__(int i, int64_t j) {
    double s = scal[i];
    if(inv) s = 1.0 / s;
    A[(int64_t)i * lda + j] *= s;
  });
}
int hiopamd_mat_symmetrize(hiopamd_ctx* ctx, int n, double* A, int64_t lda)
{
  // copy the upper triangle onto the lower (reference :912 symmetrize())
  return launch_mat_ew(ctx, n, n, [=] __device__(int i, int64_t j) {
    if(j > i) A[j * lda + i] = A[(int64_t)i * lda + j];
  });
}

}  // extern "C"
