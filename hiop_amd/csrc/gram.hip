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
  for(int sp = 0; sp < nsplit; ++sp) s += partial[((int64_t)sp * ntiles + tile) * (T * T) + off];
  double* w = W + (int64_t)i * ldw + j;
  const double v = (beta == 0.0 ? 0.0 : beta * (*w)) + alpha * s;
  *w = v;
  // symmetric product: mirror onto the lower triangle, exactly like the reference's
  // Wdata[i*k+j] = Wdata[j*k+i] = beta*Wdata[i*k+j] + alpha*acc  (hiopHessianLowRank.cpp:1107)
  if(sym && j > i) W[(int64_t)j * ldw + i] = v;
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
  // HIOPAMD_GRAM_T = 128: 128 x 128 tiles, 4 x 4 MFMA tiles per wave, two workgroups per CU; 64 (default): 64 x 64 tiles,
  // 2 x 2 MFMA tiles per wave, four workgroups per CU
  static int gt = -1;
  if(gt < 0) gt = std::getenv("HIOPAMD_GRAM_T") ? std::atoi(std::getenv("HIOPAMD_GRAM_T")) : 128;
  const int T = (gt == 64) ? 64 : 128;   // 1284: 128 x 128 tiles with 8 waves (4 x 2 MFMA tiles per wave)
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
  int nsplit = ((T == 128) ? 480 : 960) / live_tiles;
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
  if(gt == 1284) {
    hipLaunchKernelGGL((gram_partial_kernel<128, 4>), dim3(nsplit, ntiles), dim3(512), 0, ctx->stream, ma, mb, n, A, B,
                       same_ab ? 1 : 0, vec_all, d, kchunk, tiles_b, symm, sym_cols, partial);
    hipLaunchKernelGGL(gram_fold_kernel<128>, fgrid, dim3(kBlock), 0, ctx->stream, ma, mb, nsplit, ntiles, tiles_b, symm,
                       sym_cols, partial, beta, W, ldw, alpha);
  } else if(T == 128) {
    hipLaunchKernelGGL(gram_partial_kernel<128>, dim3(nsplit, ntiles), dim3(kBlock), 0, ctx->stream, ma, mb, n, A, B,
                       same_ab ? 1 : 0, vec_all, d, kchunk, tiles_b, symm, sym_cols, partial);
    hipLaunchKernelGGL(gram_fold_kernel<128>, fgrid, dim3(kBlock), 0, ctx->stream, ma, mb, nsplit, ntiles, tiles_b, symm,
                       sym_cols, partial, beta, W, ldw, alpha);
  } else {
    hipLaunchKernelGGL(gram_partial_kernel<64>, dim3(nsplit, ntiles), dim3(kBlock), 0, ctx->stream, ma, mb, n, A, B,
                       same_ab ? 1 : 0, vec_all, d, kchunk, tiles_b, symm, sym_cols, partial);
    hipLaunchKernelGGL(gram_fold_kernel<64>, fgrid, dim3(kBlock), 0, ctx->stream, ma, mb, nsplit, ntiles, tiles_b, symm,
                       sym_cols, partial, beta, W, ldw, alpha);
  }
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
