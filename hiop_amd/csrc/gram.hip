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
#pragma unroll
  for(int hb = 0; hb < 8; hb += 4) {   // two batches of 4 rows: 4 (x2) independent loads in flight each
    double v0[4], v1[4];
#pragma unroll
    for(int p = 0; p < 4; ++p) {
      const int gr = r0 + (hb + p) * 16 + (tid >> 4);
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
      const int r = (hb + p) * 16 + (tid >> 4);
      const bool rok = (r0 + r) < nrows;
      Xs[r][kk] = (rok && k0ok) ? v0[p] * w0 : 0.0;
      Xs[r][kk + 1] = (rok && k1ok) ? v1[p] * w1 : 0.0;
    }
  }
}

// partial[split][tile][128][128] (dense 128x128 slabs; only the valid part is read back)
__global__ __launch_bounds__(kBlock, 2) void gram_partial_kernel(int ma, int mb, int64_t n, const GramRows A,
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
  if(sym_cols > 0 && tb < ta && (tb + 1) * GR_T <= sym_cols) return;
  const int split = blockIdx.x;
  const int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  __shared__ double As[GR_T][GR_LDS];
  __shared__ double Bs[GR_T][GR_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lk = lane >> 4, li = lane & 15;
  const bool a_vec = vec_all != 0, b_vec = vec_all != 0;
  const bool same = sym && (ta == tb) && same_ab;

  double4_t acc[4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};

  for(int64_t k0 = kbeg; k0 < kend; k0 += GR_KT) {
    __syncthreads();
    // the weight goes on the A side only; the un-weighted symmetric diagonal tile re-uses As for B
    gram_stage(A, ma, ta * GR_T, k0, kend, d, a_vec, As, tid);
    const bool reuse = same && (d == nullptr);
    if(!reuse) gram_stage(B, mb, tb * GR_T, k0, kend, nullptr, b_vec, Bs, tid);
    __syncthreads();
    const double(*Bsrc)[GR_LDS] = reuse ? As : Bs;
#pragma unroll
    for(int kk = 0; kk < GR_KT / 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for(int i = 0; i < 4; ++i) a[i] = As[wr * 64 + i * 16 + li][kk * 4 + lk];
#pragma unroll
      for(int j = 0; j < 4; ++j) b[j] = Bsrc[wc * 64 + j * 16 + li][kk * 4 + lk];
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // write the partial slab (row-major 128x128)
  double* P = partial + ((int64_t)split * gridDim.y + tile) * (GR_T * GR_T);
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = wr * 64 + i * 16 + lk + 4 * reg;
#pragma unroll
      for(int j = 0; j < 4; ++j) {
        const int col = wc * 64 + j * 16 + li;
        P[row * GR_T + col] = acc[i][j][reg];
      }
    }
}


// ---------------------------------------------------------------------------------------------------------
// v3 of the tile kernel (default).  Same 128 x 128 tile per workgroup, 64 x 64 quadrant per wave, but
//  * every 16 x 16 x 4 block product is FOUR v_mfma_f64_4x4x4_4b_f64 (block-diagonal 4 x 4 products; the B operand is read
//    from LDS with its row index rotated by 0/4/8/12 inside the 16-row block): 75 TFLOP/s sustained on MI355X versus
//    36-46 for v_mfma_f64_16x16x4_f64 (scripts/probes/mfma_f64_peak.hip);
//  * NO workgroup barrier in the K loop: one s_barrier per 32-deep stage costs 1.6 us against 3.6 us of MFMA work
//    (scripts/probes/mfma_lds_feed.hip: 74.6 -> 51.5 TFLOP/s), so every wave stages its OWN 64 A-rows and 64 B-rows
//    in a private LDS region (2x the L2->LDS traffic, 62.6 TFLOP/s in the probe) and only orders its own LDS
//    stores/loads;
//  * one workgroup per CU (512 VGPRs per lane): the global loads of stage s+2 are in flight during the MFMAs of stage
//    s+1 (64 prefetch doubles per lane).
// acc[i][j][s] of lane l = element (A-row 4*((l>>2)&3) + (l>>4),  B-row ((l&15) + 4 s) & 15) of block (i, j).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock, 1) void gram_partial3_kernel(int ma, int mb, int64_t n, const GramRows A,
                                                                  const GramRows B, int vec_all,
                                                                  const double* __restrict__ d, int64_t kchunk,
                                                                  int tiles_b, int sym, int sym_cols,
                                                                  double* __restrict__ partial)
{
  const int tile = blockIdx.y;
  const int ta = tile / tiles_b, tb = tile % tiles_b;
  if(sym && tb < ta) return;
  if(sym_cols > 0 && tb < ta && (tb + 1) * GR_T <= sym_cols) return;
  const int split = blockIdx.x;
  const int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = kbeg + kchunk;
  if(kend > n) kend = n;
  __shared__ __attribute__((aligned(16))) double Ws[4][GR_T][GR_LDS];   // per wave: rows 0..63 A, 64..127 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lk = lane >> 4, li = lane & 15;
  const bool vec = vec_all != 0;
  double(*W)[GR_LDS] = Ws[wave];

  double acc[4][4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int sft = 0; sft < 4; ++sft) acc[i][j][sft] = 0.0;

  // this wave's rows: A rows ra0 + [0,64), B rows rb0 + [0,64); lane -> (row lk + 4 p, k pair 2 li)
  const int ra0 = ta * GR_T + wr * 64, rb0 = tb * GR_T + wc * 64;
  const double* srcA[16];
  const double* srcB[16];
#pragma unroll
  for(int p = 0; p < 16; ++p) {
    {
      const int gr = ra0 + 4 * p + lk;
      const int grc = (gr < ma) ? gr : (ma - 1);
      const int seg = (grc < A.rows[0]) ? 0 : ((grc < A.rows[1]) ? 1 : 2);
      const int base = (seg == 0) ? 0 : A.rows[seg - 1];
      srcA[p] = A.p[seg] + (int64_t)(grc - base) * A.ld[seg];
    }
    {
      const int gr = rb0 + 4 * p + lk;
      const int grc = (gr < mb) ? gr : (mb - 1);
      const int seg = (grc < B.rows[0]) ? 0 : ((grc < B.rows[1]) ? 1 : 2);
      const int base = (seg == 0) ? 0 : B.rows[seg - 1];
      srcB[p] = B.p[seg] + (int64_t)(grc - base) * B.ld[seg];
    }
  }
  double pa0[16], pa1[16], pb0[16], pb1[16], w0 = 1.0, w1 = 1.0;
  auto gload = [&](int64_t k0) {
    const int64_t k = k0 + 2 * li;
    const bool k1ok = k + 1 < kend;
    const int64_t kc0 = (k < kend) ? k : (kend - 1), kc1 = k1ok ? (k + 1) : (kend - 1);
#pragma unroll
    for(int p = 0; p < 16; ++p) {
      if(vec && k1ok) {
        const double2 t = *reinterpret_cast<const double2*>(srcA[p] + k);
        pa0[p] = t.x;
        pa1[p] = t.y;
        const double2 u = *reinterpret_cast<const double2*>(srcB[p] + k);
        pb0[p] = u.x;
        pb1[p] = u.y;
      } else {
        pa0[p] = srcA[p][kc0];
        pa1[p] = srcA[p][kc1];
        pb0[p] = srcB[p][kc0];
        pb1[p] = srcB[p][kc1];
      }
    }
    if(d) {
      w0 = d[kc0];
      w1 = d[kc1];
    }
  };
  auto lstore = [&](int64_t k0) {
    const int64_t k = k0 + 2 * li;
    const bool k0ok = k < kend, k1ok = k + 1 < kend;
#pragma unroll
    for(int p = 0; p < 16; ++p) {
      const int r = 4 * p + lk;
      const bool aok = (ra0 + r) < ma, bok = (rb0 + r) < mb;
      *reinterpret_cast<double2*>(&W[r][2 * li]) = double2{(aok && k0ok) ? pa0[p] * w0 : 0.0, (aok && k1ok) ? pa1[p] * w1 : 0.0};
      *reinterpret_cast<double2*>(&W[64 + r][2 * li]) = double2{(bok && k0ok) ? pb0[p] : 0.0, (bok && k1ok) ? pb1[p] : 0.0};
    }
  };
  auto compute = [&]() {
    // operand reads right before each k-step: the probe reaches 74.5 TFLOP/s this way, pipelining them one step ahead adds
    // 40 VGPRs for nothing (mfma_lds_feed.hip); two k-steps per loop trip bound what the scheduler may hoist
#pragma unroll 2
    for(int kk = 0; kk < GR_KT / 4; ++kk) {
      double a[4], b[4][4];
#pragma unroll
      for(int i = 0; i < 4; ++i) a[i] = W[i * 16 + li][kk * 4 + lk];
#pragma unroll
      for(int j = 0; j < 4; ++j)
#pragma unroll
        for(int sft = 0; sft < 4; ++sft) b[j][sft] = W[64 + j * 16 + ((li + 4 * sft) & 15)][kk * 4 + lk];
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int j = 0; j < 4; ++j)
#pragma unroll
          for(int sft = 0; sft < 4; ++sft)
            acc[i][j][sft] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j][sft], acc[i][j][sft], 0, 0, 0);
    }
  };
  if(kbeg < kend) {
    gload(kbeg);
    lstore(kbeg);
    if(kbeg + GR_KT < kend) gload(kbeg + GR_KT);
    for(int64_t k0 = kbeg; k0 < kend; k0 += GR_KT) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS stores are visible to its own reads
      compute();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its reads are done before the region is refilled
      if(k0 + GR_KT < kend) {
        lstore(k0 + GR_KT);
        if(k0 + 2 * GR_KT < kend) gload(k0 + 2 * GR_KT);
      }
    }
  }
  double* P = partial + ((int64_t)split * gridDim.y + tile) * (GR_T * GR_T);
  const int arow = 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int sft = 0; sft < 4; ++sft) {
        const int row = wr * 64 + i * 16 + arow;
        const int col = wc * 64 + j * 16 + ((li + 4 * sft) & 15);
        P[row * GR_T + col] = acc[i][j][sft];
      }
}

__global__ __launch_bounds__(kBlock) void gram_fold_kernel(int ma, int mb, int nsplit, int ntiles, int tiles_b, int sym,
                                                           int sym_cols, const double* __restrict__ partial, double beta,
                                                           double* __restrict__ W, int64_t ldw, double alpha)
{
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if(e >= (int64_t)ma * mb) return;
  const int i = (int)(e / mb), j = (int)(e % mb);
  if(sym && j < i) return;
  // element of a skipped (mirrored) tile: read the transposed element's partials
  const bool mirrored = sym_cols > 0 && (j / GR_T) < (i / GR_T) && ((j / GR_T) + 1) * GR_T <= sym_cols;
  const int si = mirrored ? j : i, sj = mirrored ? i : j;
  const int tile = (si / GR_T) * tiles_b + (sj / GR_T);
  const int off = (si % GR_T) * GR_T + (sj % GR_T);
  double s = 0.0;
  for(int sp = 0; sp < nsplit; ++sp) s += partial[((int64_t)sp * ntiles + tile) * (GR_T * GR_T) + off];
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
  const int tiles_a = (ma + GR_T - 1) / GR_T, tiles_b = (mb + GR_T - 1) / GR_T;
  const int ntiles = tiles_a * tiles_b;
  // K split: aim at ~2 workgroups per CU (512) over the tiles that are actually computed (symmetric / mirrored ones
  // return at once), chunk a multiple of the stage depth
  int live_tiles = 0;
  for(int ta = 0; ta < tiles_a; ++ta)
    for(int tb = 0; tb < tiles_b; ++tb) {
      if(symm && tb < ta) continue;
      if(sym_cols > 0 && tb < ta && (tb + 1) * GR_T <= sym_cols) continue;
      ++live_tiles;
    }
  if(live_tiles < 1) live_tiles = 1;
  // HIOPAMD_GRAM_V2=1 selects the experimental 4x4x4 / wave-private-LDS kernel (gram_partial3_kernel); it is slower
  // than the 16x16x4 kernel on the k = 200 stack (5.8 vs 4.4 ms, profiles/r01_probes/README.md) and stays opt-in
  static int v2 = -1;
  if(v2 < 0) v2 = std::getenv("HIOPAMD_GRAM_V2") ? std::atoi(std::getenv("HIOPAMD_GRAM_V2")) : 0;
  // the default kernel runs two workgroups per CU, the experimental one a single one: size the split so that all
  // live workgroups fit one round with some slack
  int nsplit = (v2 ? 248 : 480) / live_tiles;
  if(nsplit < 1) nsplit = 1;
  int64_t kchunk = (n + nsplit - 1) / nsplit;
  kchunk = ((kchunk + GR_KT - 1) / GR_KT) * GR_KT;
  if(kchunk < 8 * GR_KT) kchunk = 8 * GR_KT;
  nsplit = (int)((n + kchunk - 1) / kchunk);
  if(nsplit < 1) nsplit = 1;
  double* partial = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nsplit * ntiles * GR_T * GR_T);
  const int vec_all = (seg_vec_ok(A, nsegA) && seg_vec_ok(B, nsegB)) ? 1 : 0;
  if(v2)
    hipLaunchKernelGGL(gram_partial3_kernel, dim3(nsplit, ntiles), dim3(kBlock), 0, ctx->stream, ma, mb, n, A, B, vec_all, d,
                       kchunk, tiles_b, symm, sym_cols, partial);
  else
    hipLaunchKernelGGL(gram_partial_kernel, dim3(nsplit, ntiles), dim3(kBlock), 0, ctx->stream, ma, mb, n, A, B,
                       same_ab ? 1 : 0, vec_all, d, kchunk, tiles_b, symm, sym_cols, partial);
  const int64_t tot = (int64_t)ma * mb;
  hipLaunchKernelGGL(gram_fold_kernel, dim3((unsigned)((tot + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, ma,
                     mb, nsplit, ntiles, tiles_b, symm, sym_cols, partial, beta, W, ldw, alpha);
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
