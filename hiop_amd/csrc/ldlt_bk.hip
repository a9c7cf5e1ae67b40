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
// Device mapping.  A column step is three launches on the context's stream; the decisions stay on the device (BkState) and the host
// never waits for them -- a launch whose column turned out to be the second half of a 2 x 2 pivot returns at once:
//   bk_column_kernel<false>  updated column k into W(:, kw), |.| maximum below the diagonal per workgroup; the LAST workgroup to finish
//                            folds the partial maxima and takes the first decision (1 x 1 without interchange, or "look at row imax")
//   bk_column_kernel<true>   (if asked for) updated column imax of the symmetric matrix into W(:, kw + 1), its off-diagonal maximum,
//                            the final decision: pivot position kp, 1 x 1 or 2 x 2
//   bk_apply_kernel          interchange kk <-> kp (rows of L in all previous columns, rows of W, the not yet updated column kk of A to
//                            position kp), the column(s) of L and the block of D from W; the last workgroup records IPIV / P and moves
//                            to the next column
// Everything a thread writes depends only on its own row plus a handful of scalars the deciding workgroup saved in BkState, so no
// launch has an internal ordering requirement.  Once per panel the host reads the panel's end (it depends on where 2 x 2 pivots
// fell) and launches the trailing update: the stepwise LDL^T's fp64-MFMA rank-K kernel (ldlt.hip) with V = W, U = the L rows.
// Cost at N = 8192: ~ 3 x 8192 small launches (latency-bound, ~ 0.1-0.2 s) + 2 N^3 / 3 flops of MFMA updates; a solve is 2 x N / 64
// block steps (one-wave 64 x 64 triangular solve + a skinny GEMV).  This is the exceptional path: the fast path stays the no-pivot
// dataflow factorisation.
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

// W(k:n, col) = [column src of the symmetric matrix](k:n) - A(k:n, k0:k-1) W(src, 0:kw-1)^T;  col = kw (src = k) or kw + 1 (src = imax)
template <bool SECOND>
__global__ __launch_bounds__(kBlock) void bk_column_kernel(int n, int k, int k0, const double* __restrict__ A, int64_t lda,
                                                           double* __restrict__ Wb, int64_t ldw, BkState* __restrict__ st,
                                                           double* __restrict__ pval, int* __restrict__ pidx)
{
  __shared__ double coef[BK_NB];
  __shared__ int sh_go, sh_src, sh_last;
  __shared__ double rv[kBlock / 64];
  __shared__ int ri[kBlock / 64];
  const int tid = threadIdx.x;
  if(tid == 0) {
    sh_go = (st->next_k == k) && (!SECOND || st->need2);
    sh_src = SECOND ? st->imax : k;
  }
  __syncthreads();
  if(!sh_go) return;
  const int kw = k - k0, col = SECOND ? kw + 1 : kw, src = sh_src;
  for(int p = tid; p < kw; p += kBlock) coef[p] = Wb[(int64_t)p * ldw + src];
  __syncthreads();
  double best = -1.0;
  int bidx = INT_MAX;
#pragma unroll
  for(int q = 0; q < BK_RPT; ++q) {
    const int i = k + blockIdx.x * BK_ROWS + q * kBlock + tid;
    if(i < n) {
      double v = (!SECOND || i >= src) ? A[(int64_t)src * lda + i] : A[(int64_t)i * lda + src];
      const double* Ap = A + (int64_t)k0 * lda + i;
      for(int p = 0; p < kw; ++p) v -= Ap[(int64_t)p * lda] * coef[p];
      Wb[(int64_t)col * ldw + i] = v;
      const bool cand = SECOND ? (i != src) : (i > k);
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
    for(int w = 1; w < kBlock / 64; ++w) bk_argmax_combine(best, bidx, rv[w], ri[w]);
    pval[blockIdx.x] = best;
    pidx[blockIdx.x] = bidx;
    __threadfence();
    sh_last = (atomicAdd(&st->cnt[SECOND ? 1 : 0], 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if(!sh_last || tid != 0) return;
  // ---- the last workgroup decides ----
  __threadfence();
  st->cnt[SECOND ? 1 : 0] = 0;
  best = -1.0;
  bidx = INT_MAX;
  for(int b = 0; b < (int)gridDim.x; ++b) bk_argmax_combine(best, bidx, __builtin_nontemporal_load(pval + b), __builtin_nontemporal_load(pidx + b));
  if(!SECOND) {
    const double wkk = __builtin_nontemporal_load(Wb + (int64_t)kw * ldw + k);
    const double absakk = fabs(wkk);
    const double colmax = (best >= 0.0) ? best : 0.0;
    const int imax = (best >= 0.0) ? bidx : k;
    st->absakk = absakk;
    st->colmax = colmax;
    st->imax = imax;
    st->c0_k = wkk;
    int need2 = 0;
    if(!(fmax(absakk, colmax) > 0.0)) {   // exactly zero column (or NaN): DSYTRF's INFO = k + 1, no interchange
      if(st->info == 0) st->info = k + 1;
    } else if(!(absakk >= BK_ALPHA * colmax)) {
      need2 = 1;
    }
    st->need2 = need2;
    if(!need2) {
      st->kp = k;
      st->kstep = 1;
      st->use_c1 = 0;
      st->c0_kk = st->c0_kp = wkk;
      st->c1_kk = st->c1_kp = 0.0;
      st->akk_old = A[(int64_t)k * lda + k];
    }
  } else {
    const int imax = src;
    const double rowmax = (best >= 0.0) ? best : 0.0;
    const double absakk = st->absakk, colmax = st->colmax;
    const double wii = fabs(__builtin_nontemporal_load(Wb + (int64_t)(kw + 1) * ldw + imax));
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
    const int kk = k + kstep - 1;
    st->kp = kp;
    st->kstep = kstep;
    st->use_c1 = use_c1;
    st->c0_kk = __builtin_nontemporal_load(Wb + (int64_t)kw * ldw + kk);
    st->c0_kp = __builtin_nontemporal_load(Wb + (int64_t)kw * ldw + kp);
    st->c1_kk = __builtin_nontemporal_load(Wb + (int64_t)(kw + 1) * ldw + kk);
    st->c1_kp = __builtin_nontemporal_load(Wb + (int64_t)(kw + 1) * ldw + kp);
    st->akk_old = A[(int64_t)kk * lda + kk];
    st->need2 = 0;
  }
}

// interchange + the column(s) of L and the block of D; thread t owns row i = k + t of the trailing part and previous column j = t
__global__ __launch_bounds__(kBlock) void bk_apply_kernel(int n, int k, int k0, double* __restrict__ A, int64_t lda,
                                                          double* __restrict__ Wb, int64_t ldw, BkState* __restrict__ st,
                                                          int* __restrict__ ipiv, int* __restrict__ perm, double* __restrict__ e)
{
  __shared__ int sh_go, sh_kp, sh_kstep, sh_use_c1, sh_last;
  __shared__ double sh_s[6];
  const int tid = threadIdx.x;
  if(tid == 0) {
    sh_go = (st->next_k == k);
    sh_kp = st->kp;
    sh_kstep = st->kstep;
    sh_use_c1 = st->use_c1;
    sh_s[0] = st->c0_kk;
    sh_s[1] = st->c0_kp;
    sh_s[2] = st->c1_kk;
    sh_s[3] = st->c1_kp;
    sh_s[4] = st->c0_k;
    sh_s[5] = st->akk_old;
  }
  __syncthreads();
  if(!sh_go) return;
  const int kp = sh_kp, kstep = sh_kstep, use_c1 = sh_use_c1, kk = k + kstep - 1, kw = k - k0;
  const double c0_kk = sh_s[0], c0_kp = sh_s[1], c1_kk = sh_s[2], c1_kp = sh_s[3], c0_k = sh_s[4], akk_old = sh_s[5];
  const bool swp = kp != kk;
  const int64_t t = (int64_t)blockIdx.x * kBlock + tid;
  if(swp) {
    if(t < k) {   // rows kk and kp of L, ALL previous columns
      double* pa = A + t * lda;
      const double u = pa[kk];
      pa[kk] = pa[kp];
      pa[kp] = u;
    }
    if(t < kw) {   // rows kk and kp of W, the panel's previous columns
      double* pw = Wb + t * ldw;
      const double u = pw[kk];
      pw[kk] = pw[kp];
      pw[kp] = u;
    }
  }
  const int64_t i = (int64_t)k + t;
  if(i < n) {
    // values of the pivot column(s) at row i AFTER the interchange
    const bool is_kk = swp && i == kk, is_kp = swp && i == kp;
    double w0, w1 = 0.0;
    if(is_kk) {          // gets what row kp held
      w0 = use_c1 ? c1_kp : c0_kp;
      w1 = c1_kp;
    } else if(is_kp) {   // gets what row kk held
      w0 = use_c1 ? c1_kk : c0_kk;
      w1 = c1_kk;
    } else {
      w0 = use_c1 ? Wb[(int64_t)(kw + 1) * ldw + i] : Wb[(int64_t)kw * ldw + i];
      if(kstep == 2) w1 = Wb[(int64_t)(kw + 1) * ldw + i];
    }
    if(use_c1 || is_kk || is_kp) Wb[(int64_t)kw * ldw + i] = w0;
    if(kstep == 2 && (is_kk || is_kp)) Wb[(int64_t)(kw + 1) * ldw + i] = w1;
    // the not yet updated column kk of A moves to position kp (its updated form is in W)
    if(swp) {
      if(i == kp) A[(int64_t)kp * lda + kp] = akk_old;
      else if(i > kk && i < kp) A[i * lda + kp] = A[(int64_t)kk * lda + i];
      else if(i > kp) A[(int64_t)kp * lda + i] = A[(int64_t)kk * lda + i];
    }
    if(kstep == 1) {
      // row k after the interchange: kk == k
      const double dk = swp ? (use_c1 ? c1_kp : c0_kp) : c0_k;
      if(i == k) A[(int64_t)k * lda + k] = dk;
      else A[(int64_t)k * lda + i] = (dk != 0.0) ? w0 * (1.0 / dk) : w0;
    } else {
      // D = [[W(k,kw), .], [W(k+1,kw), W(k+1,kw+1)]] after the interchange (row k is not part of it: kk = k + 1)
      const double wk0 = c0_k;
      const double wk10 = swp ? c0_kp : c0_kk;
      const double wk11 = swp ? c1_kp : c1_kk;
      if(i == k) {
        A[(int64_t)k * lda + k] = wk0;
      } else if(i == k + 1) {
        A[(int64_t)k * lda + k + 1] = 0.0;   // (LAPACK keeps the off-diagonal of D here; it goes to e)
        e[k] = wk10;
        A[(int64_t)(k + 1) * lda + k + 1] = wk11;
      } else {
        double d21 = wk10;
        const double d11 = wk11 / d21, d22 = wk0 / d21;
        const double tt = 1.0 / (d11 * d22 - 1.0);
        d21 = tt / d21;
        A[(int64_t)k * lda + i] = d21 * (d11 * w0 - w1);
        A[(int64_t)(k + 1) * lda + i] = d21 * (d22 * w1 - w0);
      }
    }
  }
  __syncthreads();
  if(tid == 0) {
    __threadfence();
    sh_last = (atomicAdd(&st->cnt[2], 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if(!sh_last || tid != 0) return;
  st->cnt[2] = 0;
  if(kstep == 1) {
    ipiv[k] = kp + 1;
  } else {
    ipiv[k] = ipiv[k + 1] = -(kp + 1);
  }
  if(swp) {
    const int u = perm[kk];
    perm[kk] = perm[kp];
    perm[kp] = u;
  }
  __threadfence();
  st->next_k = k + kstep;
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

// one wave: the unit lower triangular 64 x 64 diagonal block of L at (jb, jb), forward (L y = v) or backward (L^T x = v), in place
template <bool BWD>
__global__ __launch_bounds__(64) void bk_block_solve_kernel(const double* __restrict__ A, int64_t lda, int jb, int bs, double* __restrict__ v)
{
  __shared__ double Ls[64][65];   // Ls[j][i] = L(jb + i, jb + j), i > j
  const int lane = threadIdx.x;
  for(int j = 0; j < bs; ++j) Ls[j][lane] = (lane < bs && lane > j) ? A[(int64_t)(jb + j) * lda + jb + lane] : 0.0;
  double x = (lane < bs) ? v[jb + lane] : 0.0;
  __syncthreads();
  if(!BWD) {
    for(int j = 0; j < bs; ++j) {
      const double yj = __shfl(x, j, 64);
      if(lane > j) x -= Ls[j][lane] * yj;
    }
  } else {
    for(int i = bs - 1; i >= 0; --i) {
      const double xi = __shfl(x, i, 64);
      if(lane < i) x -= Ls[lane][i] * xi;
    }
  }
  if(lane < bs) v[jb + lane] = x;
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

struct hiopamd_ldlt_bk {
  hiopamd_ctx* ctx = nullptr;
  int n = 0;
  double* Wb = nullptr;      // 64 x n panel W = L D (column p of the panel contiguous)
  double* e = nullptr;       // n: off-diagonals of the 2 x 2 blocks of D
  double* tmp = nullptr;     // n: permuted right-hand side
  double* pval = nullptr;    // partial maxima of the column kernels
  int* pidx = nullptr;
  int* ipiv = nullptr;       // n: LAPACK's IPIV (1-based, negative for 2 x 2)
  int* perm = nullptr;       // n: (P A P^T)[i][j] = A[perm[i]][perm[j]]
  BkState* st = nullptr;
  bool factored = false;
};

extern "C" {

int hiopamd_ldlt_bk_create(hiopamd_ldlt_bk** out, hiopamd_ctx* ctx, int n)
{
  if(!out || !ctx || n < 0) return HIOPAMD_ERR_ARG;
  hiopamd_ldlt_bk* B = new hiopamd_ldlt_bk();
  B->ctx = ctx;
  B->n = n;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  const size_t nblk = (nn + BK_ROWS - 1) / BK_ROWS;
  bool ok = hipMalloc((void**)&B->Wb, sizeof(double) * nn * BK_NB) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->e, sizeof(double) * (nn + 1)) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->tmp, sizeof(double) * nn) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->pval, sizeof(double) * nblk) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->pidx, sizeof(int) * nblk) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->ipiv, sizeof(int) * nn) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->perm, sizeof(int) * nn) == hipSuccess;
  ok = ok && hipMalloc((void**)&B->st, sizeof(BkState)) == hipSuccess;
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
  void* ps[] = {B->Wb, B->e, B->tmp, B->pval, B->pidx, B->ipiv, B->perm, B->st};
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
    for(int k = k0; k < kcap; ++k) {
      const unsigned g = (unsigned)((n - k + BK_ROWS - 1) / BK_ROWS);
      const int span = (n - k > k) ? (n - k) : k;
      const unsigned ga = (unsigned)((span + kBlock - 1) / kBlock);
      hipLaunchKernelGGL(bk_column_kernel<false>, dim3(g), dim3(kBlock), 0, s, n, k, k0, A, lda, B->Wb, ldw, B->st, B->pval, B->pidx);
      hipLaunchKernelGGL(bk_column_kernel<true>, dim3(g), dim3(kBlock), 0, s, n, k, k0, A, lda, B->Wb, ldw, B->st, B->pval, B->pidx);
      hipLaunchKernelGGL(bk_apply_kernel, dim3(ga), dim3(kBlock), 0, s, n, k, k0, A, lda, B->Wb, ldw, B->st, B->ipiv, B->perm, B->e);
    }
    HIOPAMD_CHECK(hipGetLastError());
    if(last) break;
    int kend = 0;   // where the panel ended: k0 + 63 or k0 + 64, depending on where the 2 x 2 pivots fell
    HIOPAMD_CHECK(hipMemcpyAsync(&kend, &B->st->next_k, sizeof(int), hipMemcpyDeviceToHost, s));
    HIOPAMD_CHECK(hipStreamSynchronize(s));
    if(kend < kcap || kend > k0 + BK_NB) return HIOPAMD_ERR_STATE;
    const int kb = kend - k0;
    const int kpad = ((kb + 7) / 8) * 8;   // the update kernel walks K in steps of 8: the rows of W past kb are zero
    if(kpad > kb) HIOPAMD_CHECK(hipMemsetAsync(B->Wb + (int64_t)kb * ldw, 0, sizeof(double) * (size_t)(kpad - kb) * ldw, s));
    // A22 -= L21 D L21^T = L21 W21^T:   a(c, r) -= sum_p a(c, k0 + p) W(r, p),  c >= r >= kend
    const int rc = ldlt_rankk_update(ctx, A, lda, n, B->Wb, ldw, k0, kpad, kend, kb);   // (U rows kb..kpad-1 are masked: they alias columns this launch updates)
    if(rc != HIOPAMD_OK) return rc;
    k0 = kend;
  }
  hipLaunchKernelGGL(bk_inertia_kernel, dim3(1), dim3(kBlock), 0, s, n, A, lda, B->e, B->st);
  HIOPAMD_CHECK(hipGetLastError());
  BkState h;
  HIOPAMD_CHECK(hipMemcpyAsync(&h, B->st, sizeof(BkState), hipMemcpyDeviceToHost, s));
  HIOPAMD_CHECK(hipStreamSynchronize(s));
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
    for(int jb = 0; jb < n; jb += 64) {   // L y = P b
      const int bs = (n - jb < 64) ? (n - jb) : 64;
      hipLaunchKernelGGL(bk_block_solve_kernel<false>, dim3(1), dim3(64), 0, s, A, lda, jb, bs, v);
      const int rest = n - jb - bs;
      if(rest > 0) {
        const int rc = hiopamd_mat_trans_times_vec(ctx, bs, rest, A + (int64_t)jb * lda + jb + bs, lda, 1.0, v + jb + bs, -1.0, v + jb);
        if(rc != HIOPAMD_OK) return rc;
      }
    }
    hipLaunchKernelGGL(bk_dsolve_kernel, gn, bn, 0, s, n, A, lda, B->e, v);
    for(int jb = ((n - 1) / 64) * 64; jb >= 0; jb -= 64) {   // L^T w = z
      const int bs = (n - jb < 64) ? (n - jb) : 64;
      const int rest = n - jb - bs;
      if(rest > 0) {
        const int rc = hiopamd_mat_times_vec(ctx, bs, rest, A + (int64_t)jb * lda + jb + bs, lda, 1.0, v + jb, -1.0, v + jb + bs);
        if(rc != HIOPAMD_OK) return rc;
      }
      hipLaunchKernelGGL(bk_block_solve_kernel<true>, dim3(1), dim3(64), 0, s, A, lda, jb, bs, v);
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
