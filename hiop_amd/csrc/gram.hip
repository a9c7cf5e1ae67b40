// Weighted Gram kernels  W = beta*W + alpha * A * diag(d) * B^T  on fp64 MFMA (gfx950).
//
// reference: hiopHessianLowRank::symmMatTimesDiagTimesMatTrans_local
// (src/Optimization/hiopHessianLowRank.cpp:1079, triple scalar loop, no blocking) and
// matTimesDiagTimesMatTrans_local (:1119); the un-weighted form is hiopMatrixDense::timesMatTrans
// (src/LinAlg/hiopMatrixDenseRowMajor.cpp:646).  These are the quasi-Newton hot kernels: A is the
// k x n_local constraint Jacobian (k <= ~256, n_local ~ 1e6), so the contraction runs over the LONG
// dimension.  Both operands are K-contiguous ("NT" GEMM): tiles are staged through LDS with coalesced
// 256-byte row pieces, the weight d is folded in while staging, the K range is split across workgroups
// (>= 2 per CU) and the per-split partial tiles are folded in a fixed order by a second kernel
// (deterministic, no atomics) which also applies alpha/beta and, for the symmetric product, mirrors the
// upper triangle onto the lower one.
#include "device_utils.hpp"

#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace hiopamd {

constexpr int GR_T = 128;        // tile edge (rows of A x rows of B per workgroup)
constexpr int GR_KT = 32;        // k-depth per LDS stage
constexpr int GR_LDS = GR_KT + 2;  // 34: 16 rows x 2 k land on 32 distinct 8-byte LDS slots

typedef double double4_t __attribute__((ext_vector_type(4)));

// B operand = up to three stacked row blocks (e.g. [X; S; Y] of the low-rank KKT: one pass over the long
// dimension yields X D X^T, X D S^T and X D Y^T together)
struct GramRows {
  const double* p[3];
  int64_t ld[3];
  int rows[3];   // cumulative END row of each segment
};

// stage a 128 x 32 tile of the stacked matrix (rows r0.., cols k0..) into LDS, optionally scaled by d[k].
// All 8 row loads are issued (unconditionally, clamped) before the first LDS store.
template <int T, int NT = kBlock>
__device__ __forceinline__ void gram_stage(const GramRows X, int nrows, int r0, int64_t k0, int64_t kend,
                                           const double* __restrict__ d, bool vec_ok, double (*Xs)[GR_LDS], int tid)
{
  const int kk = (tid & 15) * 2;
  const int64_t k = k0 + kk;
  double w0 = 1.0, w1 = 1.0;
  if(d) {
    w0 = (k < kend) ? d[k] : 0.0;
    w1 = (k + 1 < kend) ? d[k + 1] : 0.0;
  }
  const bool k0ok = k < kend, k1ok = k + 1 < kend;
  const int64_t kc0 = k0ok ? k : (kend - 1), kc1 = k1ok ? (k + 1) : (kend - 1);
  constexpr int RPP = NT / 16;   // rows per pass
#pragma unroll
  for(int hb = 0; hb < T / RPP; hb += 4) {   // batches of 4 rows: 4 (x2) independent loads in flight each
    double v0[4], v1[4];
#pragma unroll
    for(int p = 0; p < 4; ++p) {
      const int gr = r0 + (hb + p) * RPP + (tid >> 4);
      const int grc = (gr < nrows) ? gr : (nrows - 1);
      const int seg = (grc < X.rows[0]) ? 0 : ((grc < X.rows[1]) ? 1 : 2);
      const int base = (seg == 0) ? 0 : X.rows[seg - 1];
      const double* src = X.p[seg] + (int64_t)(grc - base) * X.ld[seg];
      if(vec_ok && k1ok) {
        const double2 t = *reinterpret_cast<const double2*>(src + k);
        v0[p] = t.x;
        v1[p] = t.y;
      } else {
        v0[p] = src[kc0];
        v1[p] = src[kc1];
      }
    }
#pragma unroll
    for(int p = 0; p < 4; ++p) {
      const int r = (hb + p) * RPP + (tid >> 4);
      const bool rok = (r0 + r) < nrows;
      Xs[r][kk] = (rok && k0ok) ? v0[p] * w0 : 0.0;
      Xs[r][kk + 1] = (rok && k1ok) ? v1[p] * w1 : 0.0;
    }
  }
}

// partial[split][tile][T][T] (dense T x T slabs; only the valid part is read back).  T = 128: 4 x 4 MFMA tiles per wave;
// T = 64: 2 x 2 — v_mfma_f64_16x16x4_f64 runs faster with few independent accumulators per wave (36 TFLOP/s with 16, 46 with
// 4-8 in the registers-only probe; the LDL^T trailing update went from 30 to 40 TFLOP/s with 64 x 64 tiles).
template <int T, int WC = 2>
__global__ __launch_bounds__(128 * WC, (T == 128) ? 2 : 4) void gram_partial_kernel(int ma, int mb, int64_t n, const GramRows A,
                                                                 const GramRows B, int same_ab, int vec_all,
                                                                 const double* __restrict__ d, int64_t kchunk,
                                                                 int tiles_b, int sym, int sym_cols,
                                                                 double* __restrict__ partial)
{
  const int tile = blockIdx.y;
  const int ta = tile / tiles_b, tb = tile % tiles_b;
  if(sym && tb < ta) return;
  // stacked product A D [A; B1; B2]^T: a tile below the diagonal whose columns all belong to the A block is the
  // transpose of a tile above it -> not computed, the fold kernel mirrors it
  if(sym_cols > 0 && tb < ta && (tb + 1) * T <= sym_cols) return;
  const int split = blockIdx.x;
  const int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  __shared__ double As[T][GR_LDS];
  __shared__ double Bs[T][GR_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int lk = lane >> 4, li = lane & 15;
  const bool a_vec = vec_all != 0, b_vec = vec_all != 0;
  const bool same = sym && (ta == tb) && same_ab;

  constexpr int WT = T / 32;          // MFMA tiles per wave, rows
  constexpr int WU = T / (16 * WC);   // MFMA tiles per wave, columns
  double4_t acc[WT][WU];
#pragma unroll
  for(int i = 0; i < WT; ++i)
#pragma unroll
    for(int j = 0; j < WU; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};

  for(int64_t k0 = kbeg; k0 < kend; k0 += GR_KT) {
    __syncthreads();
    // the weight goes on the A side only; the un-weighted symmetric diagonal tile re-uses As for B
    gram_stage<T, 128 * WC>(A, ma, ta * T, k0, kend, d, a_vec, As, tid);
    const bool reuse = same && (d == nullptr);
    if(!reuse) gram_stage<T, 128 * WC>(B, mb, tb * T, k0, kend, nullptr, b_vec, Bs, tid);
    __syncthreads();
    const double(*Bsrc)[GR_LDS] = reuse ? As : Bs;
#pragma unroll
    for(int kk = 0; kk < GR_KT / 4; ++kk) {
      double a[WT], b[WU];
#pragma unroll
      for(int i = 0; i < WT; ++i) a[i] = As[wr * (T / 2) + i * 16 + li][kk * 4 + lk];
#pragma unroll
      for(int j = 0; j < WU; ++j) b[j] = Bsrc[wc * (T / WC) + j * 16 + li][kk * 4 + lk];
#pragma unroll
      for(int i = 0; i < WT; ++i)
#pragma unroll
        for(int j = 0; j < WU; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // write the partial slab (row-major 128x128)
  double* P = partial + ((int64_t)split * gridDim.y + tile) * (T * T);
#pragma unroll
  for(int i = 0; i < WT; ++i)
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = wr * (T / 2) + i * 16 + lk + 4 * reg;
#pragma unroll
      for(int j = 0; j < WU; ++j) {
        const int col = wc * (T / WC) + j * 16 + li;
        P[row * T + col] = acc[i][j][reg];
      }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Strip Gram: the low-rank KKT's shape — A = the first ma rows of the stacked right factor [A; S; Y] (ma ~ 200, 2l ~ 12) —
// is a bad fit for 128 x 128 workgroup tiles: 212 columns pad to 256, the symmetric part is skipped only per 128-tile, and
// every live tile re-stages the long operands (3 live tiles = 98,304 n flops issued for 45,000 n useful).  Here ONE
// workgroup of 8 waves owns a K-chunk and the WHOLE output for it: the stacked rows are staged once per chunk (each byte of
// the operands is read from HBM exactly once), the output is cut in 16 x 16 MFMA tiles, tiles below the diagonal of the
// symmetric block are neither computed nor stored (104 tiles instead of 192 at ma = 200, mb = 212), and the tiles are dealt
// to the 8 waves in row-major runs (<= 16 accumulators per wave).  The weight d[k] goes onto the A operand in registers.
// LDS: one stage = 256 rows x 32 k (k permuted so that k and k + 4 are adjacent: one ds_read_b128 feeds two k-steps),
// double-buffered; 1 workgroup per CU, 256 workgroups.  Partials go to the [split][tile][16][16] slabs gram_fold16_kernel
// folds.
// ---------------------------------------------------------------------------------------------------------
constexpr int GS_KT = 32;            // k per stage
constexpr int GS_LD = GS_KT + 2;     // 34 doubles per row: 16 rows x 16 bytes land in 16 distinct 16-byte bank groups
constexpr int GS_ROWS = 256;         // stacked rows (padded to 16)
constexpr int GS_WAVES = 8;
constexpr int GS_TPW = 16;           // tiles (accumulators) per wave
struct GramStripTiles {
  unsigned char ti[GS_WAVES][GS_TPW];   // tile row / column (units of 16) of each accumulator of each wave
  unsigned char tj[GS_WAVES][GS_TPW];
  unsigned char cnt[GS_WAVES];
};
typedef double gs_double2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64 * GS_WAVES, 1) void gram_strip_kernel(int ma, int mb, int64_t n, const GramRows X, int vec_ok,
                                                                      const double* __restrict__ d, int64_t kchunk, int tiles_b,
                                                                      int ntiles, const GramStripTiles tl,
                                                                      double* __restrict__ partial)
{
  __shared__ __attribute__((aligned(16))) double Xs[2][GS_ROWS][GS_LD];
  __shared__ double ds[2][GS_KT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4, li = lane & 15;
  const int split = blockIdx.x;
  const int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  const int rows_pad = ((mb + 15) / 16) * 16;
  const int ntw = tl.cnt[wave];
  // staging: 16 threads x 16 bytes per 32-k row piece, 32 rows per pass, rows_pad / 32 (<= 8) passes
  const int srow = tid >> 4, skk = (tid & 15) * 2;
  // position of k inside its group of 8: (k % 4) * 2 + (k / 4) % 2  ->  k and k + 4 adjacent
  const int p0 = (skk & ~7) + ((skk & 3) << 1) + ((skk >> 2) & 1);
  const int p1 = (skk & ~7) + (((skk + 1) & 3) << 1) + (((skk + 1) >> 2) & 1);
  constexpr int NP = GS_ROWS / 32;
  double v0[NP], v1[NP];
  double dreg = 0.0;
  // row pointers of this thread's NP rows, resolved ONCE (segment look-ups are dependent loads: inside the K loop they would
  // serialise every stage's operand loads behind them)
  const double* rp[NP];
  bool rok[NP];
#pragma unroll
  for(int ps = 0; ps < NP; ++ps) {
    const int r = ps * 32 + srow;
    const int rc = (r < mb) ? r : (mb - 1);
    const int seg = (rc < X.rows[0]) ? 0 : ((rc < X.rows[1]) ? 1 : 2);
    const int base = (seg == 0) ? 0 : ((seg == 1) ? X.rows[0] : X.rows[1]);
    const double* p = (seg == 0) ? X.p[0] : ((seg == 1) ? X.p[1] : X.p[2]);
    const int64_t ld = (seg == 0) ? X.ld[0] : ((seg == 1) ? X.ld[1] : X.ld[2]);
    rp[ps] = p + (int64_t)(rc - base) * ld;
    rok[ps] = r < mb;
  }
  auto gload = [&](int64_t k0) {
    const int64_t k = k0 + skk;
    const bool k0ok = k < kend, k1ok = k + 1 < kend;
    const int64_t kc0 = k0ok ? k : (kend - 1), kc1 = k1ok ? (k + 1) : (kend - 1);
#pragma unroll
    for(int ps = 0; ps < NP; ++ps) {
      if(ps * 32 < rows_pad) {   // uniform
        double a0, a1;
        if(vec_ok && k1ok) {
          const double2 t = *reinterpret_cast<const double2*>(rp[ps] + k);
          a0 = t.x;
          a1 = t.y;
        } else {
          a0 = rp[ps][kc0];
          a1 = rp[ps][kc1];
        }
        v0[ps] = (rok[ps] && k0ok) ? a0 : 0.0;
        v1[ps] = (rok[ps] && k1ok) ? a1 : 0.0;
      }
    }
    if(tid < GS_KT) {
      const int64_t kd = k0 + tid;
      dreg = (kd < kend) ? (d ? d[kd] : 1.0) : 0.0;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for(int ps = 0; ps < NP; ++ps) {
      if(ps * 32 < rows_pad) {
        Xs[buf][ps * 32 + srow][p0] = v0[ps];
        Xs[buf][ps * 32 + srow][p1] = v1[ps];
      }
    }
    if(tid < GS_KT) ds[buf][tid] = dreg;
  };
  double4_t acc[GS_TPW];
#pragma unroll
  for(int t = 0; t < GS_TPW; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  // LDS address of this lane's operand element = tile offset (wave-uniform, kept scalar) + lane offset
  int aoff[GS_TPW], boff[GS_TPW];
#pragma unroll
  for(int t = 0; t < GS_TPW; ++t) {
    aoff[t] = __builtin_amdgcn_readfirstlane(16 * GS_LD * (int)tl.ti[wave][t]);
    boff[t] = __builtin_amdgcn_readfirstlane(16 * GS_LD * (int)tl.tj[wave][t]);
  }
  const int lane_off = li * GS_LD + 2 * lk;
  if(kbeg < kend) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for(int64_t k0 = kbeg; k0 < kend; k0 += GS_KT) {
    const bool more = k0 + GS_KT < kend;
    if(more) gload(k0 + GS_KT);
#pragma unroll
    for(int g8 = 0; g8 < GS_KT / 8; ++g8) {
      // two k-steps per LDS read: k = 8 g8 + lk and k + 4
      const double w0 = ds[buf][8 * g8 + lk], w1 = ds[buf][8 * g8 + 4 + lk];
      const double* xb = &Xs[buf][0][0] + lane_off + 8 * g8;
      gs_double2 a = gs_double2{0.0, 0.0};
#pragma unroll
      for(int t = 0; t < GS_TPW; ++t) {
        if(t < ntw) {   // wave-uniform
          // the tiles of a wave are a row-major run: the (weighted) A operand changes only when the tile row does
          if(t == 0 || aoff[t] != aoff[t - 1]) {   // scalar compare
            a = *reinterpret_cast<const gs_double2*>(xb + aoff[t]);
            a.x *= w0;
            a.y *= w1;
          }
          const gs_double2 b = *reinterpret_cast<const gs_double2*>(xb + boff[t]);
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc[t], 0, 0, 0);
        }
      }
    }
    if(more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial slabs [split][tile][16][16]
#pragma unroll
  for(int t = 0; t < GS_TPW; ++t) {
    if(t < ntw) {
      const int tile = tl.ti[wave][t] * tiles_b + tl.tj[wave][t];
      double* P = partial + ((int64_t)split * ntiles + tile) * 256;
#pragma unroll
      for(int reg = 0; reg < 4; ++reg) P[(lk + 4 * reg) * 16 + li] = acc[t][reg];
    }
  }
}

// Strip Gram, second form (round 3).  The first form's inner loop compiled to one scalar branch per tile (tile count, "same tile
// row as before?") with the operand read, its wait and the two dependent MFMAs strictly in sequence: 43 % of the MFMA peak on
// the issued tiles at k = 200 (profiles/r02: 1.97 ms), 0.18 at k = 100.  Here the loop over the wave's tiles is branch-free:
// every wave runs exactly TPW tiles (padding tiles repeat the wave's first tile and are not stored), the operands of a batch of
// tiles are read together and the two k-steps of a batch are issued as two passes over the batch, so consecutive MFMAs never
// depend on each other.  WAVES / TPW / ROWS are chosen by the tile count: 8 waves x <= 16 tiles x 256 rows (1 workgroup per
// CU) for the k = 200 shape, 4 waves x <= 8 tiles x 128 rows with TWO workgroups per CU for k <= ~112.
template <int WAVES, int TPW>
struct GramStripTiles2 {
  unsigned char ti[WAVES][TPW];
  unsigned char tj[WAVES][TPW];
  unsigned char cnt[WAVES];
};
// The stage buffer: rows padded to 34 doubles (48 % bank-conflict cycles and still faster than the conflict-free plane layout measured
// in round 3: 1.27 vs 1.36 ms at k = 200 — the kernel is not LDS-bound; that variant lives in scripts/probes/retired/)
template <int WAVES, int TPW, int ROWS, bool VEC>
__device__ __forceinline__ void gram_strip2_body_rm(int ma, int mb, int64_t n, const GramRows& X, const double* __restrict__ d,
                                                 int64_t kchunk, int tiles_b, int ntiles, const GramStripTiles2<WAVES, TPW>& tl,
                                                 double* __restrict__ partial, double (*Xs)[ROWS][GS_LD], double (*ds)[GS_KT])
{
  constexpr int NT = 64 * WAVES;
  constexpr int RPP = NT / 16;          // rows per staging pass
  constexpr int NP = ROWS / RPP;        // staging passes
  constexpr int TB = (TPW >= 8) ? 4 : TPW;   // tiles per operand batch (registers: 8 per tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4, li = lane & 15;
  const int split = blockIdx.x;
  const int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  const int npa = __builtin_amdgcn_readfirstlane((((mb + 15) / 16) * 16 + RPP - 1) / RPP);   // staging passes with live rows
  const int ntw = tl.cnt[wave];
  const int srow = tid >> 4, skk = (tid & 15) * 2;
  // position of k inside its group of 8: (k % 4) * 2 + (k / 4) % 2  ->  k and k + 4 adjacent (one ds_read_b128 = two k-steps)
  const int p0 = (skk & ~7) + ((skk & 3) << 1) + ((skk >> 2) & 1);
  const int p1 = (skk & ~7) + (((skk + 1) & 3) << 1) + (((skk + 1) >> 2) & 1);
  // Staging without a branch or a wait inside: the row pointers are resolved once (segment look-ups are dependent loads), a
  // row outside the matrix is clamped to the last row and masked when it is written to LDS, a k outside the chunk likewise;
  // all loads of a stage are issued back to back one stage ahead, the masks are applied at the LDS store.
  const double* rp[NP];
  bool rok[NP];
#pragma unroll
  for(int ps = 0; ps < NP; ++ps) {
    const int r = ps * RPP + srow;
    const int rc = (r < mb) ? r : (mb - 1);
    const int seg = (rc < X.rows[0]) ? 0 : ((rc < X.rows[1]) ? 1 : 2);
    const int base = (seg == 0) ? 0 : ((seg == 1) ? X.rows[0] : X.rows[1]);
    const double* p = (seg == 0) ? X.p[0] : ((seg == 1) ? X.p[1] : X.p[2]);
    const int64_t ld = (seg == 0) ? X.ld[0] : ((seg == 1) ? X.ld[1] : X.ld[2]);
    rp[ps] = p + (int64_t)(rc - base) * ld;
    rok[ps] = r < mb;
  }
  double v0[NP], v1[NP];
  double dreg = 0.0;
  bool m0 = false, m1 = false, md = false;   // k / k + 1 (/ the weight's k) of the stage in flight inside the chunk
  const double* dsrc = d ? d : X.p[0];
  auto gload = [&](int64_t k0) {
    const int64_t k = k0 + skk;
    m0 = k < kend;
    m1 = k + 1 < kend;
    if constexpr(VEC) {
      // 16-byte loads (rows and leading dimensions 16-byte aligned): k is even; at the ragged end of the chunk the pair is
      // clamped to the last even position, which is inside the row (an odd n has at least one padding element: ld is even)
      const int64_t kc = m0 ? k : ((kend - 1) & ~(int64_t)1);
#pragma unroll
      for(int ps = 0; ps < NP; ++ps) {
        const double2 t = *reinterpret_cast<const double2*>(rp[ps] + kc);
        v0[ps] = t.x;
        v1[ps] = t.y;
      }
    } else {
      const int64_t kc0 = m0 ? k : (kend - 1), kc1 = m1 ? (k + 1) : (kend - 1);
#pragma unroll
      for(int ps = 0; ps < NP; ++ps) {
        v0[ps] = rp[ps][kc0];
        v1[ps] = rp[ps][kc1];
      }
    }
    // the weight: always a load (from the matrix itself when there is no weight vector), value and mask applied at the LDS store —
    // a select on the loaded value here would put a vmcnt(0) right behind the stage's loads
    const int64_t kd = k0 + (tid & (GS_KT - 1));
    md = kd < kend;
    dreg = dsrc[md ? kd : (kend - 1)];
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for(int ps = 0; ps < NP; ++ps) {
      if(ps < npa) {   // scalar; nothing waits inside
        Xs[buf][ps * RPP + srow][p0] = (rok[ps] && m0) ? v0[ps] : 0.0;
        Xs[buf][ps * RPP + srow][p1] = (rok[ps] && m1) ? v1[ps] : 0.0;
      }
    }
    if(tid < GS_KT) ds[buf][tid] = md ? (d ? dreg : 1.0) : 0.0;
  };
  double4_t acc[TPW];
#pragma unroll
  for(int t = 0; t < TPW; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  // LDS offset of each tile's operands (wave-uniform: scalar registers); a padding slot repeats tile 0
  int aoff[TPW], boff[TPW];
#pragma unroll
  for(int t = 0; t < TPW; ++t) {
    const int tt = (t < ntw) ? t : 0;
    aoff[t] = __builtin_amdgcn_readfirstlane(16 * GS_LD * (int)tl.ti[wave][tt]);
    boff[t] = __builtin_amdgcn_readfirstlane(16 * GS_LD * (int)tl.tj[wave][tt]);
  }
  const int lane_off = li * GS_LD + 2 * lk;
  if(kbeg < kend) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for(int64_t k0 = kbeg; k0 < kend; k0 += GS_KT) {
    const bool more = k0 + GS_KT < kend;
    if(more) gload(k0 + GS_KT);
#pragma unroll
    for(int g8 = 0; g8 < GS_KT / 8; ++g8) {
      const double w0 = ds[buf][8 * g8 + lk], w1 = ds[buf][8 * g8 + 4 + lk];
      const double* xb = &Xs[buf][0][0] + lane_off + 8 * g8;
#pragma unroll
      for(int tb0 = 0; tb0 < TPW; tb0 += TB) {
        gs_double2 av[TB], bv[TB];
#pragma unroll
        for(int q = 0; q < TB; ++q)
          if(tb0 + q < TPW) {
            av[q] = *reinterpret_cast<const gs_double2*>(xb + aoff[tb0 + q]);
            bv[q] = *reinterpret_cast<const gs_double2*>(xb + boff[tb0 + q]);
          }
#pragma unroll
        for(int q = 0; q < TB; ++q)
          if(tb0 + q < TPW) {
            av[q].x *= w0;
            av[q].y *= w1;
          }
#pragma unroll
        for(int q = 0; q < TB; ++q)
          if(tb0 + q < TPW) acc[tb0 + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q].x, bv[q].x, acc[tb0 + q], 0, 0, 0);
#pragma unroll
        for(int q = 0; q < TB; ++q)
          if(tb0 + q < TPW) acc[tb0 + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q].y, bv[q].y, acc[tb0 + q], 0, 0, 0);
      }
    }
    if(more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for(int t = 0; t < TPW; ++t) {
    if(t < ntw) {
      const int tile = tl.ti[wave][t] * tiles_b + tl.tj[wave][t];
      double* P = partial + ((int64_t)split * ntiles + tile) * 256;
#pragma unroll
      for(int reg = 0; reg < 4; ++reg) P[(lk + 4 * reg) * 16 + li] = acc[t][reg];
    }
  }
}

template <int WAVES, int TPW, int ROWS>
__global__ __launch_bounds__(64 * WAVES, (ROWS <= 128) ? 2 : 1) void gram_strip2_kernel_rm(int ma, int mb, int64_t n, const GramRows X,
                                                                                       int vec_ok, const double* __restrict__ d,
                                                                                       int64_t kchunk, int tiles_b, int ntiles,
                                                                                       const GramStripTiles2<WAVES, TPW> tl,
                                                                                       double* __restrict__ partial)
{
  __shared__ __attribute__((aligned(16))) double Xs[2][ROWS][GS_LD];
  __shared__ double ds[2][GS_KT];
  if(vec_ok) gram_strip2_body_rm<WAVES, TPW, ROWS, true>(ma, mb, n, X, d, kchunk, tiles_b, ntiles, tl, partial, Xs, ds);
  else gram_strip2_body_rm<WAVES, TPW, ROWS, false>(ma, mb, n, X, d, kchunk, tiles_b, ntiles, tl, partial, Xs, ds);
}

template <int WAVES, int TPW, int ROWS>
static void gram_strip2_launch(hiopamd_ctx* ctx, const std::vector<std::pair<int, int>>& need, int ma, int mb, int64_t n, const GramRows& B,
                               int vec, const double* d, int64_t kchunk, int nsplit, int tb_n, int ntiles16, double* partial)
{
  GramStripTiles2<WAVES, TPW> tl;
  std::memset(&tl, 0, sizeof(tl));
  const int per = ((int)need.size() + WAVES - 1) / WAVES;
  for(size_t q = 0; q < need.size(); ++q) {
    const int w = (int)q / per, t = (int)q % per;
    tl.ti[w][t] = (unsigned char)need[q].first;
    tl.tj[w][t] = (unsigned char)need[q].second;
    tl.cnt[w] = (unsigned char)(t + 1);
  }
  hipLaunchKernelGGL((gram_strip2_kernel_rm<WAVES, TPW, ROWS>), dim3(nsplit), dim3(64 * WAVES), 0, ctx->stream, ma, mb, n, B, vec, d,
                     kchunk, tb_n, ntiles16, tl, partial);
}

template <int T>
__global__ __launch_bounds__(kBlock) void gram_fold_kernel(int ma, int mb, int nsplit, int ntiles, int tiles_b, int sym,
                                                           int sym_cols, const double* __restrict__ partial, double beta,
                                                           double* __restrict__ W, int64_t ldw, double alpha)
{
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if(e >= (int64_t)ma * mb) return;
  const int i = (int)(e / mb), j = (int)(e % mb);
  if(sym && j < i) return;
  // element of a skipped (mirrored) tile: read the transposed element's partials
  const bool mirrored = sym_cols > 0 && (j / T) < (i / T) && ((j / T) + 1) * T <= sym_cols;
  const int si = mirrored ? j : i, sj = mirrored ? i : j;
  const int tile = (si / T) * tiles_b + (sj / T);
  const int off = (si % T) * T + (sj % T);
  double s = 0.0;
  // sixteen partials in flight per lane, summed in split order (round 4: the plain loop was one dependent load per addition --
  // 256 splits x one memory latency = 125 us for 166 workgroups' worth of outputs)
  const double* pp = partial + (int64_t)tile * (T * T) + off;
  const int64_t ps = (int64_t)ntiles * (T * T);
  int sp = 0;
  for(; sp + 16 <= nsplit; sp += 16) {
    double t[16];
#pragma unroll
    for(int u = 0; u < 16; ++u) t[u] = pp[(int64_t)(sp + u) * ps];
#pragma unroll
    for(int u = 0; u < 16; ++u) s += t[u];
  }
  for(; sp < nsplit; ++sp) s += pp[(int64_t)sp * ps];
  double* w = W + (int64_t)i * ldw + j;
  const double v = (beta == 0.0 ? 0.0 : beta * (*w)) + alpha * s;
  *w = v;
  // symmetric product: mirror onto the lower triangle, exactly like the reference's
  // Wdata[i*k+j] = Wdata[j*k+i] = beta*Wdata[i*k+j] + alpha*acc  (hiopHessianLowRank.cpp:1107)
  if(sym && j > i) W[(int64_t)j * ldw + i] = v;
}

// Fold of the strip kernels' [split][tile][16][16] slabs, tile-driven (round 4): a workgroup owns a quarter of a 16 x 16 tile (64
// elements = 512 contiguous bytes per split), wave w sums the w-th quarter of the splits with 16 loads in flight, the four sums are
// combined ((g0 + g1) + g2) + g3 through LDS -- a fixed order.  gram_fold_kernel<16> walked the outputs row-major instead (four
// separate 128-byte lines per wave load, 213 KB apart per split, one output per thread): 81-110 us for 54 MB of partials.
// Skipped (mirrored) tiles -- the rule of the strip launch: tile (i, j) with j < i and (j + 1) * 16 <= scols -- are written from
// their transposed tile; sym != 0: full symmetric output (only j >= i is summed, both triangles written).
__global__ __launch_bounds__(kBlock) void gram_fold16_kernel(int ma, int mb, int nsplit, int ntiles, int tiles_b, int sym,
                                                             int sym_cols, const double* __restrict__ partial, double beta,
                                                             double* __restrict__ W, int64_t ldw, double alpha)
{
  const int tile = blockIdx.x >> 2, quarter = blockIdx.x & 3;
  const int ta = tile / tiles_b, tb = tile % tiles_b;
  const int scols = sym ? ma : sym_cols;
  if(tb < ta && (tb + 1) * 16 <= scols) return;   // not computed: written from tile (tb, ta)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int off = quarter * 64 + lane;
  const int q = (nsplit + 3) >> 2;
  const int s0 = wave * q;
  int s1 = s0 + q;
  if(s1 > nsplit) s1 = nsplit;
  const int64_t ps = (int64_t)ntiles * 256;
  const double* pp = partial + (int64_t)tile * 256 + off;
  double g = 0.0;
  int sp = s0;
  for(; sp + 16 <= s1; sp += 16) {
    double t[16];
#pragma unroll
    for(int u = 0; u < 16; ++u) t[u] = pp[(int64_t)(sp + u) * ps];
#pragma unroll
    for(int u = 0; u < 16; ++u) g += t[u];
  }
  for(; sp < s1; ++sp) g += pp[(int64_t)sp * ps];
  __shared__ double red[4][64];
  red[wave][lane] = g;
  __syncthreads();
  if(wave != 0) return;
  const double s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  const int si = ta * 16 + (off >> 4), sj = tb * 16 + (off & 15);
  if(si >= ma || sj >= mb) return;
  if(sym && sj < si) return;
  double* w = W + (int64_t)si * ldw + sj;
  const double v = (beta == 0.0 ? 0.0 : beta * (*w)) + alpha * s;
  *w = v;
  if(sym) {
    // symmetric product: mirror onto the lower triangle, exactly like the reference's
    // Wdata[i*k+j] = Wdata[j*k+i] = beta*Wdata[i*k+j] + alpha*acc  (hiopHessianLowRank.cpp:1107)
    if(sj > si) W[(int64_t)sj * ldw + si] = v;
  } else if(ta < tb && (ta + 1) * 16 <= sym_cols && sj < ma) {
    // element (sj, si) belongs to the skipped tile (tb, ta): same sum, its own beta term
    double* wm = W + (int64_t)sj * ldw + si;
    *wm = (beta == 0.0 ? 0.0 : beta * (*wm)) + alpha * s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// small Gram: ma, mb <= 8 (the l x l blocks of the compact L-BFGS representation, l <= 8).  The 128 x 128 MFMA tile
// above would do (128/l)^2 times the useful work (1.1 ms for a 6 x 6 block at n = 1.25e6); this one is a plain
// streaming reduction: every thread owns columns k = k0 + t, k0 + t + 256, ..., keeps the ma x mb running sums in
// registers, the workgroup reduces them through LDS in a fixed order, a second launch folds the workgroups' partials.
// HBM-bound: (ma + mb + 1) * n * 8 bytes.
// ---------------------------------------------------------------------------------------------------------
constexpr int GS_M = 8;

__global__ __launch_bounds__(kBlock) void gram_small_partial(int ma, int mb, int64_t n, const double* __restrict__ A,
                                                             int64_t lda, const double* __restrict__ B, int64_t ldb,
                                                             const double* __restrict__ d, int64_t kchunk,
                                                             double* __restrict__ partial)
{
  const int64_t kbeg = (int64_t)blockIdx.x * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  double acc[GS_M][GS_M];
#pragma unroll
  for(int i = 0; i < GS_M; ++i)
#pragma unroll
    for(int j = 0; j < GS_M; ++j) acc[i][j] = 0.0;
  for(int64_t k = kbeg + threadIdx.x; k < kend; k += kBlock) {
    const double w = d ? d[k] : 1.0;
    double a[GS_M], b[GS_M];
#pragma unroll
    for(int i = 0; i < GS_M; ++i) a[i] = (i < ma) ? A[(int64_t)i * lda + k] * w : 0.0;
#pragma unroll
    for(int j = 0; j < GS_M; ++j) b[j] = (j < mb) ? B[(int64_t)j * ldb + k] : 0.0;
#pragma unroll
    for(int i = 0; i < GS_M; ++i)
#pragma unroll
      for(int j = 0; j < GS_M; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
  // fixed-order reduction: wave shuffle tree, then the 4 waves through LDS
  __shared__ double red[kBlock / 64][GS_M * GS_M];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for(int i = 0; i < GS_M; ++i)
#pragma unroll
    for(int j = 0; j < GS_M; ++j) {
      double v = acc[i][j];
      for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if(lane == 0) red[wave][i * GS_M + j] = v;
    }
  __syncthreads();
  if(threadIdx.x < GS_M * GS_M) {
    const int e = threadIdx.x;
    partial[(int64_t)blockIdx.x * (GS_M * GS_M) + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
  }
}

// one wave per output element: lane-strided partial sums, shuffle tree (fixed order)
__global__ __launch_bounds__(64) void gram_small_fold(int ma, int mb, int nsplit, int sym, const double* __restrict__ partial,
                                                      double beta, double* __restrict__ W, int64_t ldw, double alpha)
{
  const int e = blockIdx.x;
  const int i = e / GS_M, j = e % GS_M;
  if(i >= ma || j >= mb) return;
  if(sym && j < i) return;
  double s = 0.0;
  for(int sp = threadIdx.x; sp < nsplit; sp += 64) s += partial[(int64_t)sp * (GS_M * GS_M) + e];
  for(int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if(threadIdx.x != 0) return;
  double* w = W + (int64_t)i * ldw + j;
  const double v = (beta == 0.0 ? 0.0 : beta * (*w)) + alpha * s;
  *w = v;
  if(sym && j > i) W[(int64_t)j * ldw + i] = v;
}


// ---------------------------------------------------------------------------------------------------------
// The four l x l blocks of hiopHessianLowRank::updateInternalBFGSRepresentation (hiopHessianLowRank.cpp:400-460) and of the compact
// direct form in ONE pass over S, Y and DhInv (round 4; before: four gram_small passes, each re-reading its rows and a weight vector
// written by an element-wise launch in between):
//   G0 = Y DhInv Y^T      G1 = S (sigma DhInv) Y^T      G2 = S (sigma (sigma DhInv - 1)) S^T      G3 = sigma S S^T
// Wave q of a workgroup accumulates block q over the workgroup's column chunk (the four waves read the same lines: HBM once, the
// repeats are L2 / L1 hits); the weights are formed per element exactly as the element-wise launches formed them.  Partials go
// to [split][4][64]; gram_quad_fold sums the splits lane-strided + shuffle tree (fixed order), mirrors the symmetric blocks.
// ---------------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(kBlock) void gram_quad_partial(int64_t n, const double* __restrict__ S, const double* __restrict__ Y,
                                                            int64_t ld, const double* __restrict__ dh, double sigma, int64_t kchunk,
                                                            double* __restrict__ partial)
{
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t kbeg = (int64_t)blockIdx.x * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  const double* __restrict__ A = (q == 0) ? Y : S;
  const double* __restrict__ B = (q <= 1) ? Y : S;
  double acc[L][L];
#pragma unroll
  for(int i = 0; i < L; ++i)
#pragma unroll
    for(int j = 0; j < L; ++j) acc[i][j] = 0.0;
  for(int64_t k = kbeg + lane; k < kend; k += 64) {
    double a[L], b[L];
    const double d = (q == 3) ? 1.0 : dh[k];
#pragma unroll
    for(int i = 0; i < L; ++i) a[i] = A[(int64_t)i * ld + k];
#pragma unroll
    for(int j = 0; j < L; ++j) b[j] = B[(int64_t)j * ld + k];
    const double w = (q == 0) ? d : (q == 1) ? d * sigma : (q == 2) ? (d * sigma - 1.0) * sigma : 1.0;
#pragma unroll
    for(int i = 0; i < L; ++i) a[i] *= w;
#pragma unroll
    for(int i = 0; i < L; ++i)
#pragma unroll
      for(int j = 0; j < L; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
  double* out = partial + ((int64_t)blockIdx.x * 4 + q) * (GS_M * GS_M);
#pragma unroll
  for(int i = 0; i < L; ++i)
#pragma unroll
    for(int j = 0; j < L; ++j) {
      double v = acc[i][j];
      for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if(lane == 0) out[i * GS_M + j] = v;
    }
}

__global__ __launch_bounds__(64) void gram_quad_fold(int l, int nsplit, const double* __restrict__ partial, double sigma,
                                                     double* __restrict__ G)
{
  const int q = blockIdx.x / (GS_M * GS_M), e = blockIdx.x % (GS_M * GS_M);
  const int i = e / GS_M, j = e % GS_M;
  if(i >= l || j >= l) return;
  const bool sym = q != 1;
  if(sym && j < i) return;
  double s = 0.0;
  for(int sp = threadIdx.x; sp < nsplit; sp += 64) s += partial[((int64_t)sp * 4 + q) * (GS_M * GS_M) + e];
  for(int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if(threadIdx.x != 0) return;
  const double v = (q == 3) ? sigma * s : s;
  double* W = G + (int64_t)q * l * l;
  W[i * l + j] = v;
  if(sym && j > i) W[j * l + i] = v;
}

// The same four blocks for a secant memory longer than GS_M pairs (the reference's secant_memory_len goes up to 256; hiopamd_hess_lowrank
// accepts l_max <= 32): the l x l blocks are cut in GS_M x GS_M sub-blocks, blockIdx.y = sub-block (bi, bj); rows beyond l read as zero.
// Partials: [split][sub-block][4][GS_M * GS_M].  S and Y are re-read once per sub-block column (L2 hits mostly): the rare shape.
__global__ __launch_bounds__(kBlock) void gram_quad_partial_blk(int l, int nb, int64_t n, const double* __restrict__ S,
                                                                const double* __restrict__ Y, int64_t ld, const double* __restrict__ dh,
                                                                double sigma, int64_t kchunk, double* __restrict__ partial)
{
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bi = blockIdx.y / nb, bj = blockIdx.y % nb;
  const int64_t kbeg = (int64_t)blockIdx.x * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  const double* __restrict__ A = (q == 0) ? Y : S;
  const double* __restrict__ B = (q <= 1) ? Y : S;
  double acc[GS_M][GS_M];
#pragma unroll
  for(int i = 0; i < GS_M; ++i)
#pragma unroll
    for(int j = 0; j < GS_M; ++j) acc[i][j] = 0.0;
  for(int64_t k = kbeg + lane; k < kend; k += 64) {
    double a[GS_M], b[GS_M];
    const double d = (q == 3) ? 1.0 : dh[k];
    const double w = (q == 0) ? d : (q == 1) ? d * sigma : (q == 2) ? (d * sigma - 1.0) * sigma : 1.0;
#pragma unroll
    for(int i = 0; i < GS_M; ++i) a[i] = (GS_M * bi + i < l) ? A[(int64_t)(GS_M * bi + i) * ld + k] * w : 0.0;
#pragma unroll
    for(int j = 0; j < GS_M; ++j) b[j] = (GS_M * bj + j < l) ? B[(int64_t)(GS_M * bj + j) * ld + k] : 0.0;
#pragma unroll
    for(int i = 0; i < GS_M; ++i)
#pragma unroll
      for(int j = 0; j < GS_M; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
  double* out = partial + (((int64_t)blockIdx.x * (nb * nb) + blockIdx.y) * 4 + q) * (GS_M * GS_M);
#pragma unroll
  for(int i = 0; i < GS_M; ++i)
#pragma unroll
    for(int j = 0; j < GS_M; ++j) {
      double v = acc[i][j];
      for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if(lane == 0) out[i * GS_M + j] = v;
    }
}

// one wave per entry (q, i, j) of the four l x l blocks, splits summed lane-strided + shuffle tree like gram_quad_fold; the symmetric
// blocks take their lower triangle from the upper one
__global__ __launch_bounds__(64) void gram_quad_fold_blk(int l, int nb, int nsplit, const double* __restrict__ partial, double sigma,
                                                         double* __restrict__ G)
{
  const int q = blockIdx.x / (l * l), e = blockIdx.x % (l * l);
  const int i = e / l, j = e % l;
  const bool sym = q != 1;
  if(sym && j < i) return;
  const int sb = (i / GS_M) * nb + (j / GS_M), es = (i % GS_M) * GS_M + (j % GS_M);
  double s = 0.0;
  for(int sp = threadIdx.x; sp < nsplit; sp += 64) s += partial[(((int64_t)sp * (nb * nb) + sb) * 4 + q) * (GS_M * GS_M) + es];
  for(int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if(threadIdx.x != 0) return;
  const double v = (q == 3) ? sigma * s : s;
  double* W = G + (int64_t)q * l * l;
  W[i * l + j] = v;
  if(sym && j > i) W[j * l + i] = v;
}

}  // namespace hiopamd

using namespace hiopamd;

static bool seg_vec_ok(const GramRows& R, int nseg)
{
  for(int q = 0; q < nseg; ++q)
    if((R.ld[q] & 1) != 0 || (((uintptr_t)R.p[q]) & 15) != 0) return false;
  return true;
}

// W(ma x mb) = beta*W + alpha * A diag(d) [B0;B1;B2]^T
static int gram_launch(hiopamd_ctx* ctx, int ma, int mb, int64_t n, const GramRows& A, int nsegA, const GramRows& B,
                       int nsegB, bool same_ab, const double* d, double beta, double* W, int64_t ldw, double alpha,
                       int sym, int sym_cols = 0)
{
  if(ma < 0 || mb < 0 || n < 0) return HIOPAMD_ERR_ARG;
  if(ma == 0 || mb == 0) return HIOPAMD_OK;
  const int symm = (sym && same_ab && ma == mb) ? 1 : 0;
  if(ma <= GS_M && mb <= GS_M && nsegA == 1 && nsegB == 1 && n > 0) {   // streaming reduction for the tiny blocks
    int nsplit = 512;
    int64_t kchunk = (n + nsplit - 1) / nsplit;
    if(kchunk < 4 * kBlock) kchunk = 4 * kBlock;
    nsplit = (int)((n + kchunk - 1) / kchunk);
    double* partial = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nsplit * GS_M * GS_M);
    hipLaunchKernelGGL(gram_small_partial, dim3(nsplit), dim3(kBlock), 0, ctx->stream, ma, mb, n, A.p[0], A.ld[0], B.p[0],
                       B.ld[0], d, kchunk, partial);
    hipLaunchKernelGGL(gram_small_fold, dim3(GS_M * GS_M), dim3(64), 0, ctx->stream, ma, mb, nsplit, symm, partial, beta, W,
                       ldw, alpha);
    HIOPAMD_CHECK(hipGetLastError());
    return HIOPAMD_OK;
  }
  // strip form: A = the leading rows of the (stacked) right factor, everything fits one workgroup's tile budget
  const bool a_leads_b = (sym_cols == ma && sym_cols > 0) || (symm != 0);
  if(a_leads_b && ma <= mb && mb <= GS_ROWS && n >= 64 * GS_KT) {
    const int ta_n = (ma + 15) / 16, tb_n = (mb + 15) / 16;
    const int scols = symm ? ma : sym_cols;
    std::vector<std::pair<int, int>> need;
    for(int i = 0; i < ta_n; ++i)
      for(int j = 0; j < tb_n; ++j) {
        if(j < i && (j + 1) * 16 <= scols) continue;   // mirrored by the fold kernel (same rule, T = 16)
        need.emplace_back(i, j);
      }
    if((int)need.size() <= GS_WAVES * GS_TPW) {
      const int ntiles16 = ta_n * tb_n;
      const int rows_pad = tb_n * 16;
      const bool small = (int)need.size() <= 32 && rows_pad <= 128;   // 4 waves, two workgroups per CU
      int nsplit = small ? 512 : 256;
      int64_t kchunk = (n + nsplit - 1) / nsplit;
      kchunk = ((kchunk + GS_KT - 1) / GS_KT) * GS_KT;
      if(kchunk < 8 * GS_KT) kchunk = 8 * GS_KT;
      nsplit = (int)((n + kchunk - 1) / kchunk);
      double* partial = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nsplit * ntiles16 * 256);
      const int vec = seg_vec_ok(B, nsegB) ? 1 : 0;
      const int per2 = ((int)need.size() + (small ? 4 : 8) - 1) / (small ? 4 : 8);
      if(per2 <= 13) {   // (14-16 tiles per wave would spill: first form)
        // second form: branch-free tile loop, (waves, tiles per wave, rows) by the tile count
        const int per = per2;
        if(small) {
          if(per <= 4) gram_strip2_launch<4, 4, 128>(ctx, need, ma, mb, n, B, vec, d, kchunk, nsplit, tb_n, ntiles16, partial);
          else if(per <= 6) gram_strip2_launch<4, 6, 128>(ctx, need, ma, mb, n, B, vec, d, kchunk, nsplit, tb_n, ntiles16, partial);
          else if(per <= 7) gram_strip2_launch<4, 7, 128>(ctx, need, ma, mb, n, B, vec, d, kchunk, nsplit, tb_n, ntiles16, partial);
          else gram_strip2_launch<4, 8, 128>(ctx, need, ma, mb, n, B, vec, d, kchunk, nsplit, tb_n, ntiles16, partial);
        } else {
          if(per <= 8) gram_strip2_launch<8, 8, 256>(ctx, need, ma, mb, n, B, vec, d, kchunk, nsplit, tb_n, ntiles16, partial);
          else if(per <= 11) gram_strip2_launch<8, 11, 256>(ctx, need, ma, mb, n, B, vec, d, kchunk, nsplit, tb_n, ntiles16, partial);
          else gram_strip2_launch<8, 13, 256>(ctx, need, ma, mb, n, B, vec, d, kchunk, nsplit, tb_n, ntiles16, partial);
        }
      } else {
        GramStripTiles tl;
        std::memset(&tl, 0, sizeof(tl));
        const int per = ((int)need.size() + GS_WAVES - 1) / GS_WAVES;
        for(size_t q = 0; q < need.size(); ++q) {
          const int w = (int)q / per, t = (int)q % per;
          tl.ti[w][t] = (unsigned char)need[q].first;
          tl.tj[w][t] = (unsigned char)need[q].second;
          tl.cnt[w] = (unsigned char)(t + 1);
        }
        hipLaunchKernelGGL(gram_strip_kernel, dim3(nsplit), dim3(64 * GS_WAVES), 0, ctx->stream, ma, mb, n, B, vec, d, kchunk, tb_n,
                           ntiles16, tl, partial);
      }
      hipLaunchKernelGGL(gram_fold16_kernel, dim3((unsigned)(4 * ntiles16)), dim3(kBlock), 0, ctx->stream, ma, mb, nsplit, ntiles16, tb_n,
                         symm, symm ? 0 : sym_cols, partial, beta, W, ldw, alpha);
      HIOPAMD_CHECK(hipGetLastError());
      return HIOPAMD_OK;
    }
  }
  // other shapes: 128 x 128 tiles, 4 x 4 MFMA tiles per wave, two workgroups per CU
  constexpr int T = 128;
  const int tiles_a = (ma + T - 1) / T, tiles_b = (mb + T - 1) / T;
  const int ntiles = tiles_a * tiles_b;
  // K split: aim at one round of workgroups over the tiles that are actually computed (symmetric / mirrored ones return at
  // once), chunk a multiple of the stage depth
  int live_tiles = 0;
  for(int ta = 0; ta < tiles_a; ++ta)
    for(int tb = 0; tb < tiles_b; ++tb) {
      if(symm && tb < ta) continue;
      if(sym_cols > 0 && tb < ta && (tb + 1) * T <= sym_cols) continue;
      ++live_tiles;
    }
  if(live_tiles < 1) live_tiles = 1;
  int nsplit = 480 / live_tiles;
  if(nsplit < 1) nsplit = 1;
  int64_t kchunk = (n + nsplit - 1) / nsplit;
  kchunk = ((kchunk + GR_KT - 1) / GR_KT) * GR_KT;
  if(kchunk < 8 * GR_KT) kchunk = 8 * GR_KT;
  nsplit = (int)((n + kchunk - 1) / kchunk);
  if(nsplit < 1) nsplit = 1;
  double* partial = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nsplit * ntiles * T * T);
  const int vec_all = (seg_vec_ok(A, nsegA) && seg_vec_ok(B, nsegB)) ? 1 : 0;
  const int64_t tot = (int64_t)ma * mb;
  const dim3 fgrid((unsigned)((tot + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(gram_partial_kernel<128>, dim3(nsplit, ntiles), dim3(kBlock), 0, ctx->stream, ma, mb, n, A, B, same_ab ? 1 : 0, vec_all,
                     d, kchunk, tiles_b, symm, sym_cols, partial);
  hipLaunchKernelGGL(gram_fold_kernel<128>, fgrid, dim3(kBlock), 0, ctx->stream, ma, mb, nsplit, ntiles, tiles_b, symm, sym_cols, partial,
                     beta, W, ldw, alpha);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

extern "C" int hiopamd_gram_weighted(hiopamd_ctx* ctx, int ma, int mb, int64_t n, const double* A, int64_t lda,
                                     const double* B, int64_t ldb, const double* d, double beta, double* W,
                                     int64_t ldw, double alpha, int sym)
{
  GramRows Ra{{A, A, A}, {lda, lda, lda}, {ma, ma, ma}};
  GramRows Rb{{B, B, B}, {ldb, ldb, ldb}, {mb, mb, mb}};
  return gram_launch(ctx, ma, mb, n, Ra, 1, Rb, 1, A == B && lda == ldb, d, beta, W, ldw, alpha, sym);
}

// W(ma x (m0+m1+m2)) = beta*W + alpha * A diag(d) [B0;B1;B2]^T in ONE pass over the long dimension
extern "C" int hiopamd_gram_weighted_stacked(hiopamd_ctx* ctx, int ma, int64_t n, const double* A, int64_t lda, int m0,
                                             const double* B0, int64_t ldb0, int m1, const double* B1, int64_t ldb1,
                                             int m2, const double* B2, int64_t ldb2, const double* d, double beta,
                                             double* W, int64_t ldw, double alpha)
{
  if(m0 < 0 || m1 < 0 || m2 < 0) return HIOPAMD_ERR_ARG;
  GramRows Ra{{A, A, A}, {lda, lda, lda}, {ma, ma, ma}};
  // empty segments borrow a valid pointer so the clamped loads stay in bounds
  const double* q1 = m1 > 0 ? B1 : B0;
  const double* q2 = m2 > 0 ? B2 : q1;
  GramRows Rb{{B0, q1, q2}, {ldb0, m1 > 0 ? ldb1 : ldb0, m2 > 0 ? ldb2 : (m1 > 0 ? ldb1 : ldb0)}, {m0, m0 + m1, m0 + m1 + m2}};
  // first column block = A itself -> its part of the result is symmetric: the tiles below the diagonal inside it are
  // mirrored instead of computed (one tile in four at k = 200)
  const int sym_cols = (B0 == A && ldb0 == lda && m0 == ma) ? m0 : 0;
  return gram_launch(ctx, ma, m0 + m1 + m2, n, Ra, 1, Rb, 3, false, d, beta, W, ldw, alpha, 0, sym_cols);
}

// G (4 l^2 doubles) = [ Y DhInv Y^T | S (sigma DhInv) Y^T | S (sigma (sigma DhInv - 1)) S^T | sigma S S^T ], l x l row-major each
extern "C" int hiopamd_gram_lowrank_blocks(hiopamd_ctx* ctx, int l, int64_t n, const double* St, const double* Yt, int64_t ld,
                                           const double* DhInv, double sigma, double* G)
{
  if(!ctx || l < 0 || l > 256 || n < 0 || (l > 0 && (!St || !Yt || !DhInv || !G))) return HIOPAMD_ERR_ARG;
  if(l == 0) return HIOPAMD_OK;
  if(n == 0) return hiopamd_vec_set_to_constant(ctx, 4 * (int64_t)l * l, G, 0.0);
  int nsplit = 512;
  int64_t kchunk = (n + nsplit - 1) / nsplit;
  if(kchunk < 16 * 64) kchunk = 16 * 64;
  nsplit = (int)((n + kchunk - 1) / kchunk);
  if(l > GS_M) {   // a secant memory beyond GS_M pairs: the blocks in GS_M x GS_M sub-blocks (same weights, same fold)
    const int nb = (l + GS_M - 1) / GS_M;
    if(nb * nb > 128) {   // (keeps the partial buffer small; l <= 88 stays with 512 splits)
      nsplit = 64;
      kchunk = (n + nsplit - 1) / nsplit;
      if(kchunk < 16 * 64) kchunk = 16 * 64;
      nsplit = (int)((n + kchunk - 1) / kchunk);
    }
    double* partial = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nsplit * nb * nb * 4 * GS_M * GS_M);
    if(!partial) return HIOPAMD_ERR_HIP;
    hipLaunchKernelGGL(gram_quad_partial_blk, dim3(nsplit, nb * nb), dim3(kBlock), 0, ctx->stream, l, nb, n, St, Yt, ld, DhInv, sigma,
                       kchunk, partial);
    hipLaunchKernelGGL(gram_quad_fold_blk, dim3(4 * l * l), dim3(64), 0, ctx->stream, l, nb, nsplit, partial, sigma, G);
    HIOPAMD_CHECK(hipGetLastError());
    return HIOPAMD_OK;
  }
  double* partial = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nsplit * 4 * GS_M * GS_M);
  const dim3 g(nsplit), b(kBlock);
  switch(l) {
    case 1: hipLaunchKernelGGL(gram_quad_partial<1>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
    case 2: hipLaunchKernelGGL(gram_quad_partial<2>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
    case 3: hipLaunchKernelGGL(gram_quad_partial<3>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
    case 4: hipLaunchKernelGGL(gram_quad_partial<4>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
    case 5: hipLaunchKernelGGL(gram_quad_partial<5>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
    case 6: hipLaunchKernelGGL(gram_quad_partial<6>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
    case 7: hipLaunchKernelGGL(gram_quad_partial<7>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
    default: hipLaunchKernelGGL(gram_quad_partial<8>, g, b, 0, ctx->stream, n, St, Yt, ld, DhInv, sigma, kchunk, partial); break;
  }
  hipLaunchKernelGGL(gram_quad_fold, dim3(4 * GS_M * GS_M), dim3(64), 0, ctx->stream, l, nsplit, partial, sigma, G);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
