// Quasi-Newton low-rank path on MI355X:
//   hiopHessianLowRank  (compact L-BFGS, B = sigma*I + Dx - [..] V^-1 [..]^T)  and
//   hiopKKTLinSysLowRank (k x k reduced system N = J (H+Dx)^-1 J^T + Dd^-1, SPD solve with refinement).
//
// reference: src/Optimization/hiopHessianLowRank.cpp (update :262, updateInternalBFGSRepresentation :400,
// solve :495, symMatTimesInverseTimesMatTrans :549, timesVecCmn :974) and
// src/Optimization/hiopKKTLinSys.cpp:1057-1330 (update, solveCompressed, solveWithRefin).
// The reference runs this path on the CPU only (it asserts mem_space == DEFAULT, hiopKKTLinSys.cpp:1037).
//
// MI355X design:
//  * every O(n) object (x, gradients, Jacobian rows, the S/Y multivectors, DhInv) is a column slice living
//    in HBM; every k x k / 2l x 2l / l-vector is tiny, replicated, and stays ON THE DEVICE between kernels;
//  * the three Gram passes of symMatTimesInverseTimesMatTrans (X D X^T, X D S^T, X D Y^T) are ONE pass over
//    X on fp64 MFMA (hiopamd_gram_weighted_stacked) — the reference streams X three times with scalar loops;
//  * all reductions of one phase are packed into one device buffer and all-reduced ONCE (RCCL over xGMI
//    through the context's hook): the reference's two MPI_Allreduce at :590-591 become one, the three
//    l x l blocks at :459 stay one;
//  * the 2l x 2l matrix V is indefinite with a possibly singular leading block (free variables have Dx = 0),
//    so it cannot be factored without pivoting: a single-workgroup LU with partial pivoting in LDS replaces
//    the reference's DSYTRF/DSYTRS on the host — no D2H round trip;
//  * only the scalars the IPM branches on (||s||, s^T y, ...) cross to the host.
#include "device_utils.hpp"
#include "dense_internal.hpp"

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

namespace hiopamd {

constexpr int kMaxV = 64;  // 2*l_max <= 64

// V X = B for the 2l x 2l middle matrices (V of solveWithV, hiopHessianLowRank.cpp:677; M of the compact direct form): LU with
// partial pivoting.  Round 4: the factorisation runs ONCE per secant update (small_lu_factor_kernel: one workgroup, factors and
// pivots to device memory); a solve loads them into LDS and substitutes (small_lu_apply_kernel: one thread per right-hand side,
// its vector in LDS).  Before, every solve re-factored V and kept its vector in a dynamically indexed private array (scratch
// memory): 25 us per call, seven calls per KKT step.  B is nrhs x nv row-major (each ROW is one right-hand side -- the
// "RHS_fortran" view of the reference, :601-606); X may be B (in place).
__global__ __launch_bounds__(kBlock) void small_lu_factor_kernel(int nv, const double* __restrict__ V, double* __restrict__ LUg,
                                                                 int* __restrict__ pivg, int* __restrict__ info)
{
  __shared__ double LU[kMaxV][kMaxV + 1];
  __shared__ int pr;
  const int tid = threadIdx.x;
  for(int e = tid; e < nv * nv; e += kBlock) LU[e / nv][e % nv] = V[e];
  if(tid == 0) *info = 0;
  __syncthreads();
  for(int k = 0; k < nv; ++k) {
    if(tid == 0) {
      int p = k;
      double best = fabs(LU[k][k]);
      for(int r = k + 1; r < nv; ++r) {
        const double a = fabs(LU[r][k]);
        if(a > best) {
          best = a;
          p = r;
        }
      }
      pr = p;
      pivg[k] = p;
      if((best == 0.0 || !isfinite(best)) && *info == 0) *info = k + 1;
    }
    __syncthreads();
    const int p = pr;
    if(p != k) {
      for(int c = tid; c < nv; c += kBlock) {
        const double t = LU[k][c];
        LU[k][c] = LU[p][c];
        LU[p][c] = t;
      }
    }
    __syncthreads();
    const double dinv = 1.0 / LU[k][k];
    for(int r = k + 1 + tid; r < nv; r += kBlock) LU[r][k] *= dinv;
    __syncthreads();
    const int m = nv - k - 1;
    for(int e = tid; e < m * m; e += kBlock) {
      const int r = k + 1 + e / m, c = k + 1 + e % m;
      LU[r][c] -= LU[r][k] * LU[k][c];
    }
    __syncthreads();
  }
  for(int e = tid; e < nv * nv; e += kBlock) LUg[e] = LU[e / nv][e % nv];
}

constexpr int kLuApplyBlock = 64;
__global__ __launch_bounds__(kLuApplyBlock) void small_lu_apply_kernel(int nv, const double* __restrict__ LUg,
                                                                       const int* __restrict__ pivg, int nrhs,
                                                                       const double* B, int64_t ldb, double* X, int64_t ldx)
{
  extern __shared__ double lu_sm[];   // LU (nv x nv) | x (nv x 64: element i of thread t at [i * 64 + t])
  double* LU = lu_sm;
  double* xs = lu_sm + nv * nv;
  const int tid = threadIdx.x;
  for(int e = tid; e < nv * nv; e += kLuApplyBlock) LU[e] = LUg[e];
  __syncthreads();
  const int j = blockIdx.x * kLuApplyBlock + tid;
  if(j >= nrhs) return;
  double* x = xs + tid;
  const double* b = B + (int64_t)j * ldb;
  for(int i = 0; i < nv; ++i) x[i * kLuApplyBlock] = b[i];
  for(int k = 0; k < nv; ++k) {
    const int p = pivg[k];
    if(p != k) {
      const double t = x[k * kLuApplyBlock];
      x[k * kLuApplyBlock] = x[p * kLuApplyBlock];
      x[p * kLuApplyBlock] = t;
    }
  }
  for(int i = 1; i < nv; ++i) {
    double acc = x[i * kLuApplyBlock];
    for(int c = 0; c < i; ++c) acc -= LU[i * nv + c] * x[c * kLuApplyBlock];
    x[i * kLuApplyBlock] = acc;
  }
  for(int i = nv - 1; i >= 0; --i) {
    double acc = x[i * kLuApplyBlock];
    for(int c = i + 1; c < nv; ++c) acc -= LU[i * nv + c] * x[c * kLuApplyBlock];
    x[i * kLuApplyBlock] = acc / LU[i * nv + i];
  }
  double* xo = X + (int64_t)j * ldx;
  for(int i = 0; i < nv; ++i) xo[i] = x[i * kLuApplyBlock];
}

// V (2l x 2l, full symmetric) from the three reduced Gram blocks G = [YtDhInvY | StB0DhInvY | StDS] (each l x l)
//   V = [ StDS              StB0DhInvY - L ]
//       [ (..)^T            D + YtDhInvY   ]      (reference :414-475)
__global__ void assemble_V_kernel(int l, const double* __restrict__ G, const double* __restrict__ Lm,
                                  const double* __restrict__ Dv, double* __restrict__ V)
{
  const int nv = 2 * l;
  for(int e = threadIdx.x; e < nv * nv; e += blockDim.x) {
    int i = e / nv, j = e % nv;
    if(i > j) {
      const int t = i;
      i = j;
      j = t;
    }  // value of the upper-triangular entry (i <= j), mirrored
    double v;
    if(j < l) {
      v = G[2 * l * l + i * l + j];                        // StDS (symmetric: both triangles written by the Gram)
    } else if(i < l) {
      v = G[l * l + i * l + (j - l)] - Lm[i * l + (j - l)];  // StB0DhInvY - L
    } else {
      v = G[(i - l) * l + (j - l)] + ((i == j) ? Dv[i - l] : 0.0);  // D + YtDhInvY
    }
    V[e] = v;
  }
}

// Secant update, Jacobian part in ONE pass (reference: four GEMVs over Jc, Jc_prev, Jd, Jd_prev followed by two full copies
// Jac_prev <- Jac, hiopHessianLowRank.cpp:293-299,366-371 — 8 x the Jacobian's bytes; here 3 x):
//   y_new[j] += sum_r (J[r][j] - Jprev[r][j]) * mult[r]      and      Jprev[r][j] = J[r][j]
// one thread per two adjacent columns, the multipliers staged in LDS per chunk of rows.
constexpr int SJ_ROWCHUNK = 256;
// CP: column pairs per thread (pair p of thread t = columns base + 512 p + 2 t, +1): a workgroup streams CP x 4 KB of every row
template <int CP>
__global__ __launch_bounds__(kBlock) void secant_jac_kernel(int m, int64_t n, const double* __restrict__ J, double* __restrict__ Jp,
                                                            const double* __restrict__ mult, double* __restrict__ y_new)
{
  __shared__ double ms[SJ_ROWCHUNK];
  const int64_t jb = (int64_t)blockIdx.x * (2 * kBlock * CP) + 2 * threadIdx.x;
  const bool vec_ok = ((n & 1) == 0) && ((((uintptr_t)J) & 15) == 0) && ((((uintptr_t)Jp) & 15) == 0);
  double a0[CP], a1[CP];
#pragma unroll
  for(int p = 0; p < CP; ++p) a0[p] = a1[p] = 0.0;
  const bool full = vec_ok && (jb + (int64_t)(CP - 1) * 2 * kBlock + 1 < n);   // every pair of this thread inside the row
  for(int rb = 0; rb < m; rb += SJ_ROWCHUNK) {
    const int rc = (m - rb < SJ_ROWCHUNK) ? (m - rb) : SJ_ROWCHUNK;
    __syncthreads();
    if((int)threadIdx.x < rc) ms[threadIdx.x] = mult[rb + threadIdx.x];
    __syncthreads();
    if(full) {
      // batches of RBATCH rows: all loads of a batch are issued before its first store (the compiler cannot move a load of J_prev
      // across a store to J_prev on its own: with the plain loop 2 loads were in flight per lane); same order of the multiply-adds
      constexpr int RBATCH = 8 / CP;
      const double* Jr = J + (int64_t)rb * n + jb;
      double* Pr = Jp + (int64_t)rb * n + jb;
      int r = 0;
      for(; r + RBATCH <= rc; r += RBATCH) {
        double2 v[RBATCH][CP], q[RBATCH][CP];
#pragma unroll
        for(int u = 0; u < RBATCH; ++u)
#pragma unroll
          for(int p = 0; p < CP; ++p) v[u][p] = *reinterpret_cast<const double2*>(Jr + (int64_t)(r + u) * n + p * 2 * kBlock);
#pragma unroll
        for(int u = 0; u < RBATCH; ++u)
#pragma unroll
          for(int p = 0; p < CP; ++p) q[u][p] = *reinterpret_cast<const double2*>(Pr + (int64_t)(r + u) * n + p * 2 * kBlock);
#pragma unroll
        for(int u = 0; u < RBATCH; ++u)
#pragma unroll
          for(int p = 0; p < CP; ++p) {
            a0[p] = fma(v[u][p].x - q[u][p].x, ms[r + u], a0[p]);
            a1[p] = fma(v[u][p].y - q[u][p].y, ms[r + u], a1[p]);
            *reinterpret_cast<double2*>(Pr + (int64_t)(r + u) * n + p * 2 * kBlock) = v[u][p];
          }
      }
      for(; r < rc; ++r)
#pragma unroll
        for(int p = 0; p < CP; ++p) {
          const double2 v = *reinterpret_cast<const double2*>(Jr + (int64_t)r * n + p * 2 * kBlock);
          const double2 q = *reinterpret_cast<const double2*>(Pr + (int64_t)r * n + p * 2 * kBlock);
          a0[p] = fma(v.x - q.x, ms[r], a0[p]);
          a1[p] = fma(v.y - q.y, ms[r], a1[p]);
          *reinterpret_cast<double2*>(Pr + (int64_t)r * n + p * 2 * kBlock) = v;
        }
    } else {
#pragma unroll
      for(int p = 0; p < CP; ++p) {
        const int64_t j0 = jb + (int64_t)p * 2 * kBlock;
        if(j0 >= n) continue;
        const double* Jr = J + (int64_t)rb * n + j0;
        double* Pr = Jp + (int64_t)rb * n + j0;
        for(int r = 0; r < rc; ++r) {
          const double v0 = Jr[(int64_t)r * n], q0 = Pr[(int64_t)r * n];
          a0[p] = fma(v0 - q0, ms[r], a0[p]);
          Pr[(int64_t)r * n] = v0;
          if(j0 + 1 < n) {
            const double v1 = Jr[(int64_t)r * n + 1], q1 = Pr[(int64_t)r * n + 1];
            a1[p] = fma(v1 - q1, ms[r], a1[p]);
            Pr[(int64_t)r * n + 1] = v1;
          }
        }
      }
    }
  }
#pragma unroll
  for(int p = 0; p < CP; ++p) {
    const int64_t j0 = jb + (int64_t)p * 2 * kBlock;
    if(j0 < n) {
      y_new[j0] += a0[p];
      if(j0 + 1 < n) y_new[j0 + 1] += a1[p];
    }
  }
}

// Secant memory full: rows move up by one and the new pair goes into the last row -- St, Yt in ONE launch, in place (a thread owns
// two adjacent columns and walks the rows: read row r + 1, write row r).  Round 3 staged each shift through the workspace with two
// hipMemcpy2DAsync and copied the new rows with two more (reference: shiftRows + copyRowsFrom, hiopHessianLowRank.cpp:322-329).
__global__ __launch_bounds__(kBlock) void secant_shift_append_kernel(int l, int64_t n, double* __restrict__ St, double* __restrict__ Yt,
                                                                     const double* __restrict__ s_new,
                                                                     const double* __restrict__ y_new)
{
  const int64_t j0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2;
  if(j0 >= n) return;
  const bool two = (j0 + 1 < n) && ((n & 1) == 0) && ((((uintptr_t)St) & 15) == 0) && ((((uintptr_t)Yt) & 15) == 0) &&
                   ((((uintptr_t)s_new) & 15) == 0) && ((((uintptr_t)y_new) & 15) == 0);
  for(int which = 0; which < 2; ++which) {
    double* M = which ? Yt : St;
    const double* nw = which ? y_new : s_new;
    if(two) {
      int r = 0;
      for(; r + 4 <= l - 1; r += 4) {
        double2 v[4];
#pragma unroll
        for(int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const double2*>(M + (int64_t)(r + 1 + u) * n + j0);
#pragma unroll
        for(int u = 0; u < 4; ++u) *reinterpret_cast<double2*>(M + (int64_t)(r + u) * n + j0) = v[u];
      }
      for(; r < l - 1; ++r) {
        const double2 v = *reinterpret_cast<const double2*>(M + (int64_t)(r + 1) * n + j0);
        *reinterpret_cast<double2*>(M + (int64_t)r * n + j0) = v;
      }
      *reinterpret_cast<double2*>(M + (int64_t)(l - 1) * n + j0) = *reinterpret_cast<const double2*>(nw + j0);
    } else {
      for(int c = 0; c < 2 && j0 + c < n; ++c) {
        for(int r = 0; r < l - 1; ++r) M[(int64_t)r * n + j0 + c] = M[(int64_t)(r + 1) * n + j0 + c];
        M[(int64_t)(l - 1) * n + j0 + c] = nw[j0 + c];
      }
    }
  }
}


// three dots in one pass: out = [x.y, x.x, y.y]
struct dot3_t {
  double a, b, c;
};
struct OpDot3 {
  const double *x, *y;
  __device__ dot3_t identity() const { return dot3_t{0.0, 0.0, 0.0}; }
  __device__ dot3_t map(int64_t i) const
  {
    const double xv = x[i], yv = y[i];
    return dot3_t{xv * yv, xv * xv, yv * yv};
  }
  __device__ dot3_t combine(dot3_t p, dot3_t q) const { return dot3_t{p.a + q.a, p.b + q.b, p.c + q.c}; }
};
// the secant update's four scalars in one pass: max |s_i| (comparison semantics of vec_infnorm: a NaN never replaces the running
// maximum) and the three dot products of OpDot3, summed in the same order as there
struct sec4_t {
  double a, b, c, smax;
};
struct OpSecant4 {
  const double *x, *y;
  __device__ sec4_t identity() const { return sec4_t{0.0, 0.0, 0.0, 0.0}; }
  __device__ sec4_t map(int64_t i) const
  {
    const double xv = x[i], yv = y[i];
    return sec4_t{xv * yv, xv * xv, yv * yv, fabs(xv)};
  }
  __device__ sec4_t combine(sec4_t p, sec4_t q) const
  {
    return sec4_t{p.a + q.a, p.b + q.b, p.c + q.c, (q.smax > p.smax) ? q.smax : p.smax};
  }
};

}  // namespace hiopamd

using namespace hiopamd;

struct hiopamd_hess_lowrank {
  hiopamd_ctx* ctx = nullptr;
  int64_t n = 0;     // local length
  int l_max = 0, l_curr = -1;
  double sigma = 1.0, sigma0 = 1.0;
  int strategy = 3;
  bool matrix_changed = false;
  unsigned long long version = 0;   // bumped whenever (B + Dx) changes: consumers cache products of its inverse against it
  int m_eq = 0, m_ineq = 0;
  // device
  double *St = nullptr, *Yt = nullptr;       // l_max x n
  double *DhInv = nullptr, *Dx = nullptr;    // n
  double *x_prev = nullptr, *g_prev = nullptr, *Jc_prev = nullptr, *Jd_prev = nullptr;
  double *nv1 = nullptr, *nv2 = nullptr;     // n work vectors
  double *dL = nullptr, *dD = nullptr;       // l_max^2, l_max
  double *dG = nullptr;                      // 3 l_max^2 Gram blocks
  double *dV = nullptr;                      // (2 l_max)^2 full symmetric
  double *dSS = nullptr;                     // [sigma S^T S | L ; L^T | -D] for the compact mat-vec
  double *dsmall = nullptr;                  // small vectors (8 * 2 l_max)
  int* dinfo = nullptr;
  double *dVlu = nullptr, *dSSlu = nullptr;  // LU factors of dV / dSS (once per secant update)
  int *dVpiv = nullptr, *dSSpiv = nullptr;   // their pivots (inside the dinfo allocation)
  // host mirrors of the tiny BFGS bookkeeping (reference keeps L_, D_ on the host as well)
  std::vector<double> L, D;
  bool have_prev = false;
};

static int allreduce_dev(hiopamd_ctx* ctx, double* buf, size_t count, int op)
{
  // a hook installed on a 1-rank partition is still called (lets the RCCL path run on a single GPU)
  if(!ctx->allreduce) return HIOPAMD_OK;
  return ctx_allreduce(ctx, buf, count, op) == 0 ? HIOPAMD_OK : HIOPAMD_ERR_HIP;
}
static int to_host(hiopamd_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  HIOPAMD_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  return HIOPAMD_OK;
}
static int to_dev(hiopamd_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  HIOPAMD_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));  // src is pageable host memory of the caller's frame
  return HIOPAMD_OK;
}

static int small_lu_factor(hiopamd_ctx* ctx, int nv, const double* V, double* LU, int* piv, int* dinfo)
{
  if(nv == 0) return HIOPAMD_OK;
  if(nv > kMaxV) return HIOPAMD_ERR_ARG;
  hipLaunchKernelGGL(small_lu_factor_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, nv, V, LU, piv, dinfo);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
// X (nrhs rows of nv, leading dimension ldx) = solutions of the factored system for the rows of B; X == B allowed
static int small_lu_apply(hiopamd_ctx* ctx, int nv, const double* LU, const int* piv, int nrhs, const double* B, int64_t ldb,
                          double* X, int64_t ldx)
{
  if(nv == 0 || nrhs == 0) return HIOPAMD_OK;
  if(nv > kMaxV) return HIOPAMD_ERR_ARG;
  const size_t lds = sizeof(double) * ((size_t)nv * nv + (size_t)nv * kLuApplyBlock);
  hipLaunchKernelGGL(small_lu_apply_kernel, dim3((nrhs + kLuApplyBlock - 1) / kLuApplyBlock), dim3(kLuApplyBlock), lds, ctx->stream,
                     nv, LU, piv, nrhs, B, ldb, X, ldx);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

extern "C" {

int hiopamd_hess_lowrank_create(hiopamd_hess_lowrank** out, hiopamd_ctx* ctx, int64_t n_local, int m_eq, int m_ineq,
                                int l_max, double sigma0, int sigma_update_strategy)
{
  if(!out || !ctx || n_local < 0 || l_max < 0 || 2 * l_max > kMaxV || m_eq < 0 || m_ineq < 0) return HIOPAMD_ERR_ARG;
  hiopamd_hess_lowrank* h = new hiopamd_hess_lowrank();
  h->ctx = ctx;
  h->n = n_local;
  h->l_max = l_max;
  h->sigma = h->sigma0 = sigma0;
  h->strategy = sigma_update_strategy;
  h->m_eq = m_eq;
  h->m_ineq = m_ineq;
  const size_t n = (size_t)(n_local > 0 ? n_local : 1), lm = (size_t)(l_max > 0 ? l_max : 1);
  auto A = [](double** p, size_t cnt) { return hipMalloc((void**)p, sizeof(double) * (cnt ? cnt : 1)) == hipSuccess; };
  bool ok = A(&h->St, lm * n) && A(&h->Yt, lm * n) && A(&h->DhInv, n) && A(&h->Dx, n) && A(&h->x_prev, n) &&
            A(&h->g_prev, n) && A(&h->Jc_prev, (size_t)m_eq * n) && A(&h->Jd_prev, (size_t)m_ineq * n) && A(&h->nv1, n) &&
            A(&h->nv2, n) && A(&h->dL, lm * lm) && A(&h->dD, lm) && A(&h->dG, 4 * lm * lm) && A(&h->dV, 4 * lm * lm) &&
            A(&h->dSS, 4 * lm * lm) && A(&h->dVlu, 4 * lm * lm) && A(&h->dSSlu, 4 * lm * lm) && A(&h->dsmall, 16 * 2 * lm + 64);
  ok = ok && hipMalloc((void**)&h->dinfo, 64 + sizeof(int) * 4 * (size_t)lm) == hipSuccess;
  if(ok) {
    h->dVpiv = h->dinfo + 16;
    h->dSSpiv = h->dVpiv + 2 * lm;
  }
  if(!ok) {
    hiopamd_hess_lowrank_destroy(h);
    return HIOPAMD_ERR_HIP;
  }
  RC(hiopamd_vec_set_to_constant(ctx, n_local, h->DhInv, 1.0 / sigma0));
  RC(hiopamd_vec_set_to_constant(ctx, n_local, h->Dx, 0.0));
  *out = h;
  return HIOPAMD_OK;
}

int hiopamd_hess_lowrank_destroy(hiopamd_hess_lowrank* h)
{
  if(!h) return HIOPAMD_OK;
  (void)hipStreamSynchronize(h->ctx->stream);
  double* ps[] = {h->St, h->Yt, h->DhInv, h->Dx, h->x_prev, h->g_prev, h->Jc_prev, h->Jd_prev, h->nv1, h->nv2,
                  h->dL, h->dD, h->dG,    h->dV, h->dSS,    h->dsmall, h->dVlu, h->dSSlu};
  for(double* p : ps) (void)hipFree(p);
  (void)hipFree(h->dinfo);
  delete h;
  return HIOPAMD_OK;
}

int hiopamd_hess_lowrank_l_curr(const hiopamd_hess_lowrank* h) { return h ? (h->l_curr < 0 ? 0 : h->l_curr) : -1; }
double hiopamd_hess_lowrank_sigma(const hiopamd_hess_lowrank* h) { return h ? h->sigma : 0.0; }
double* hiopamd_hess_lowrank_St(hiopamd_hess_lowrank* h) { return h ? h->St : nullptr; }
double* hiopamd_hess_lowrank_Yt(hiopamd_hess_lowrank* h) { return h ? h->Yt : nullptr; }

// reference :197
int hiopamd_hess_lowrank_update_log_barrier_diagonal(hiopamd_hess_lowrank* h, const double* Dx)
{
  if(!h) return HIOPAMD_ERR_ARG;
  const double sigma = h->sigma;
  double* DhInv = h->DhInv;
  double* Dxc = h->Dx;
  RC(launch_ew(h->ctx, h->n, [=] __device__(int64_t i) {
    const double d = Dx[i];
    Dxc[i] = d;
    DhInv[i] = 1.0 / (sigma + d);
  }));
  h->matrix_changed = true;
  h->version++;
  return HIOPAMD_OK;
}

// reference :262.  x, grad_f, Jc (m_eq x n), Jd (m_ineq x n), yc, yd of the CURRENT iterate (device).
int hiopamd_hess_lowrank_update(hiopamd_hess_lowrank* h, const double* x, const double* grad_f, const double* Jc,
                                const double* Jd, const double* yc, const double* yd, int* stored_host)
{
  if(!h) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t n = h->n;
  const int me = h->m_eq, mi = h->m_ineq;
  if(stored_host) *stored_host = 0;
  bool jac_saved = false;   // the fused secant pass below leaves Jc_prev / Jd_prev up to date
  auto save_prev = [&]() -> int {
    double* xp_ = h->x_prev;
    double* gp_ = h->g_prev;
    RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
      xp_[i] = x[i];
      gp_[i] = grad_f[i];
    }));
    if(!jac_saved) {
      RC(hiopamd_vec_copy(ctx, (int64_t)me * n, h->Jc_prev, Jc));
      RC(hiopamd_vec_copy(ctx, (int64_t)mi * n, h->Jd_prev, Jd));
    }
    return HIOPAMD_OK;
  };
  if(h->l_curr < 0) {  // first iterate: just remember it (:381-389)
    RC(save_prev());
    h->l_curr = 0;
    return HIOPAMD_OK;
  }
  double* s_new = h->nv1;
  double* y_new = h->nv2;
  const double* xp = h->x_prev;
  const double* gp = h->g_prev;
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
    s_new[i] = x[i] - xp[i];
    y_new[i] = grad_f[i] - gp[i];
  }));
  // One pass for the secant pair and ONE host round trip for everything the decisions below need (round 4; before: infnorm,
  // the three dot products and Yt*s were three round trips, three collectives).  The Jacobian pass runs before the step-length test
  // instead of behind it: for a zero step it adds (J - J_prev)^T y to a y_new nobody reads and refreshes J_prev, which
  // save_prev() would do anyway.
  //   y_new += (Jc - Jc_prev)^T yc + (Jd - Jd_prev)^T yd                      (:293-299)
  {
    // (two column pairs per thread -- 8 KB of a row per workgroup -- measured slower: 5.84 vs 5.56 ms per step, scripts/calls/r04_gpu_13.sh;
    //  software-pipelining the batches changed nothing: 1.17 vs 1.09-1.17 ms for the two launches, scripts/calls/r04_gpu_14.sh;
    //  non-temporal stores of J_prev: no change either, scripts/calls/r04_gpu_18.sh)
    const unsigned gx = (unsigned)((n + 2 * (int64_t)kBlock - 1) / (2 * (int64_t)kBlock));
    if(me > 0 && n > 0) hipLaunchKernelGGL(secant_jac_kernel<1>, dim3(gx), dim3(kBlock), 0, ctx->stream, me, n, Jc, h->Jc_prev, yc, y_new);
    if(mi > 0 && n > 0) hipLaunchKernelGGL(secant_jac_kernel<1>, dim3(gx), dim3(kBlock), 0, ctx->stream, mi, n, Jd, h->Jd_prev, yd, y_new);
    HIOPAMD_CHECK(hipGetLastError());
    jac_saved = true;
  }
  const int l0 = (h->l_max > 0 && h->l_curr > 0) ? h->l_curr : 0;
  std::vector<double> YTs(l0 > 0 ? l0 : 1, 0.0);
  if(l0 > 0) {  // YTs = Yt * s_new (:311), fetched together with the scalars
    RC(hiopamd_mat_times_vec(ctx, l0, n, h->Yt, n, 0.0, h->dsmall, 1.0, s_new));
    if(!ctx->allreduce) HIOPAMD_CHECK(hipMemcpyAsync(YTs.data(), h->dsmall, sizeof(double) * l0, hipMemcpyDeviceToHost, ctx->stream));
  }
  sec4_t d4{0, 0, 0, 0};   // [max |s|, s^T y, s^T s, y^T y]                     (:283, :301)
  RC(launch_reduce<sec4_t>(ctx, n, OpSecant4{s_new, y_new}, &d4));
  if(ctx->allreduce) {     // two collectives: SUM over [Yt s (l), s^T y, s^T s, y^T y], MAX over |s|
    RC(to_dev(ctx, h->dsmall + l0, &d4.a, sizeof(double) * 3));
    RC(allreduce_dev(ctx, h->dsmall, (size_t)l0 + 3, HIOPAMD_SUM));
    RC(to_dev(ctx, h->dsmall + l0 + 3, &d4.smax, sizeof(double)));
    RC(allreduce_dev(ctx, h->dsmall + l0 + 3, 1, HIOPAMD_MAX));
    std::vector<double> back((size_t)l0 + 4);
    RC(to_host(ctx, back.data(), h->dsmall, sizeof(double) * back.size()));
    for(int j = 0; j < l0; ++j) YTs[j] = back[j];
    d4.a = back[l0], d4.b = back[l0 + 1], d4.c = back[l0 + 2], d4.smax = back[l0 + 3];
  }
  const double s_inf = d4.smax;
  if(s_inf >= 100 * std::numeric_limits<double>::epsilon()) {
    const dot3_t d3{d4.a, d4.b, d4.c};
    const double sTy = d3.a, s_nrm2 = std::sqrt(d3.b), y_nrm2 = std::sqrt(d3.c);
    if(sTy > s_nrm2 * y_nrm2 * std::sqrt(std::numeric_limits<double>::epsilon())) {
      if(h->l_max > 0) {
        const int l = h->l_curr;
        if(l < h->l_max) {  // grow (:313-320, growL :779, growD :808)
          RC(hiopamd_vec_copy(ctx, n, h->St + (int64_t)l * n, s_new));
          RC(hiopamd_vec_copy(ctx, n, h->Yt + (int64_t)l * n, y_new));
          std::vector<double> Ln((size_t)(l + 1) * (l + 1), 0.0);
          for(int i = 0; i < l; ++i)
            for(int j = 0; j < l; ++j) Ln[(size_t)i * (l + 1) + j] = h->L[(size_t)i * l + j];
          for(int j = 0; j < l; ++j) Ln[(size_t)l * (l + 1) + j] = YTs[j];
          h->L.swap(Ln);
          h->D.push_back(sTy);
          h->l_curr = l + 1;
        } else {  // shift (:322-329, updateL :828, updateD :861)
          if(n > 0) {
            hipLaunchKernelGGL(secant_shift_append_kernel, dim3((unsigned)((n + 2 * (int64_t)kBlock - 1) / (2 * (int64_t)kBlock))),
                               dim3(kBlock), 0, ctx->stream, l, n, h->St, h->Yt, s_new, y_new);
            HIOPAMD_CHECK(hipGetLastError());
          }
          const int lm1 = l - 1;
          for(int i = 1; i < lm1; ++i)
            for(int j = 0; j < i; ++j) h->L[(size_t)i * l + j] = h->L[(size_t)(i + 1) * l + j + 1];
          for(int j = 0; j < lm1; ++j) h->L[(size_t)lm1 * l + j] = YTs[j + 1];
          h->L[(size_t)lm1 * l + lm1] = 0.0;
          for(int i = 0; i < l - 1; ++i) h->D[i] = h->D[i + 1];
          h->D[l - 1] = sTy;
        }
      }
      switch(h->strategy) {  // (:339-359)
        case 1: h->sigma = sTy / (s_nrm2 * s_nrm2); break;
        case 2: h->sigma = y_nrm2 * y_nrm2 / sTy; break;
        case 3: h->sigma = std::sqrt(s_nrm2 * s_nrm2 / y_nrm2 / y_nrm2); break;
        case 4: h->sigma = 0.5 * (sTy / (s_nrm2 * s_nrm2) + y_nrm2 * y_nrm2 / sTy); break;
        default: h->sigma = h->sigma0; break;
      }
      h->sigma = std::fmax(std::fmin(1e+8, h->sigma), 1e-8);
      h->matrix_changed = true;
      h->version++;
      if(stored_host) *stored_host = 1;
    }
  }
  RC(save_prev());
  return HIOPAMD_OK;
}

// reference :400
static int update_internal_bfgs_representation(hiopamd_hess_lowrank* h)
{
  hiopamd_ctx* ctx = h->ctx;
  const int64_t n = h->n;
  const int l = h->l_curr < 0 ? 0 : h->l_curr;
  h->matrix_changed = false;
  if(l == 0) return HIOPAMD_OK;
  double* G = h->dG;
  const double sigma = h->sigma;
  // G0 = Yt DhInv Yt^T ; G1 = St (sigma DhInv) Yt^T ; G2 = St (sigma (sigma DhInv - 1)) St^T (:440-458) and G3 = sigma St St^T, the
  // leading block of the middle matrix of the compact DIRECT form for timesVec, M = [sigma S^T S, L; L^T, -D]: one pass over S, Y and
  // DhInv, one all-reduce (round 3: four passes, two collectives)
  RC(hiopamd_gram_lowrank_blocks(ctx, l, n, h->St, h->Yt, n, h->DhInv, sigma, G));
  RC(allreduce_dev(ctx, G, (size_t)4 * l * l, HIOPAMD_SUM));                  // (:459)
  HIOPAMD_CHECK(hipMemcpyAsync(h->dL, h->L.data(), sizeof(double) * l * l, hipMemcpyHostToDevice, ctx->stream));
  HIOPAMD_CHECK(hipMemcpyAsync(h->dD, h->D.data(), sizeof(double) * l, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(assemble_V_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, l, G, h->dL, h->dD, h->dV);
  RC(small_lu_factor(ctx, 2 * l, h->dV, h->dVlu, h->dVpiv, h->dinfo));
  {
    double* M = h->dSS;
    const double* SS = G + 3 * l * l;
    const double* Lm = h->dL;
    const double* Dv = h->dD;
    const int nv = 2 * l;
    RC(launch_ew(ctx, (int64_t)nv * nv, [=] __device__(int64_t e) {
      const int i = (int)(e / nv), j = (int)(e % nv);
      double v;
      if(i < l && j < l) v = SS[i * l + j];
      else if(i < l) v = Lm[i * l + (j - l)];
      else if(j < l) v = Lm[j * l + (i - l)];
      else v = (i == j) ? -Dv[i - l] : 0.0;
      M[e] = v;
    }));
  }
  RC(small_lu_factor(ctx, 2 * l, h->dSS, h->dSSlu, h->dSSpiv, h->dinfo + 1));
  HIOPAMD_CHECK(hipGetLastError());
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));  // h->L / h->D host buffers must outlive the async copies
  return HIOPAMD_OK;
}

// reference :495   x = (B + Dx)^-1 rhs
int hiopamd_hess_lowrank_solve(hiopamd_hess_lowrank* h, const double* rhs, double* x)
{
  if(!h) return HIOPAMD_ERR_ARG;
  if(h->matrix_changed) RC(update_internal_bfgs_representation(h));
  hiopamd_ctx* ctx = h->ctx;
  const int64_t n = h->n;
  const int l = h->l_curr < 0 ? 0 : h->l_curr;
  const double* DhInv = h->DhInv;
  const double sigma = h->sigma;
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) { x[i] = rhs[i] * DhInv[i]; }));
  if(l == 0) return HIOPAMD_OK;
  double* sy = h->dsmall;  // [stx (l) ; ytx (l)]
  RC(hiopamd_mat_times_vec(ctx, l, n, h->St, n, 0.0, sy, sigma, x));       // S^T B0 DhInv r
  RC(hiopamd_mat_times_vec(ctx, l, n, h->Yt, n, 0.0, sy + l, 1.0, x));
  RC(allreduce_dev(ctx, sy, (size_t)2 * l, HIOPAMD_SUM));
  RC(small_lu_apply(ctx, 2 * l, h->dVlu, h->dVpiv, 1, sy, 2 * l, sy, 2 * l));   // solveWithV (:677)
  double* res = h->nv1;
  RC(hiopamd_mat_trans_times_vec(ctx, l, n, h->St, n, 0.0, res, sigma, sy));
  RC(hiopamd_mat_trans_times_vec(ctx, l, n, h->Yt, n, 1.0, res, 1.0, sy + l));
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) { x[i] -= res[i] * DhInv[i]; }));
  return HIOPAMD_OK;
}

// reference :549   W(k x k) = beta*W + alpha * X (B+Dx)^-1 X^T, X is k x n (local slice), W replicated
int hiopamd_hess_lowrank_sym_mat_times_inverse_times_mat_trans(hiopamd_hess_lowrank* h, double beta, double* W, int k,
                                                               double alpha, const double* X, double* work)
{
  if(!h || k < 0) return HIOPAMD_ERR_ARG;
  if(h->matrix_changed) RC(update_internal_bfgs_representation(h));
  hiopamd_ctx* ctx = h->ctx;
  const int64_t n = h->n;
  const int l = h->l_curr < 0 ? 0 : h->l_curr;
  const int kw = k + 2 * l;
  // work: G (k x kw) | S2Y2 (k x 2l)
  double* G = work;
  double* S2Y2 = G + (size_t)k * kw;
  // one pass: G = X DhInv [X; S; Y]^T
  RC(hiopamd_gram_weighted_stacked(ctx, k, n, X, n, k, X, n, l, h->St, n, l, h->Yt, n, h->DhInv, 0.0, G, kw, 1.0));
  RC(allreduce_dev(ctx, G, (size_t)k * kw, HIOPAMD_SUM));                    // (:590-591 fused)
  const double sigma = h->sigma;
  // W = beta*W + alpha*G[:, :k] ; S1Y1 = [sigma*G[:, k:k+l] , G[:, k+l:]] (in place) ; S2Y2 = copy
  RC(launch_ew(ctx, (int64_t)k * kw, [=] __device__(int64_t e) {
    const int i = (int)(e / kw), j = (int)(e % kw);
    if(j < k) {
      W[(int64_t)i * k + j] = (beta == 0.0 ? 0.0 : beta * W[(int64_t)i * k + j]) + alpha * G[e];
    } else {
      double v = G[e];
      if(j < k + l) v *= sigma;
      G[e] = v;
      S2Y2[(int64_t)i * 2 * l + (j - k)] = v;
    }
  }));
  if(l > 0) {
    RC(small_lu_apply(ctx, 2 * l, h->dVlu, h->dVpiv, k, S2Y2, 2 * l, S2Y2, 2 * l));   // (:606)
    // W -= alpha * [S1 Y1] [S2 Y2]^T                                         (:614-618)
    RC(hiopamd_mat_times_mat_trans(ctx, k, 2 * l, k, G + k, kw, 1.0, W, k, -alpha, S2Y2, 2 * l));
  }
  return HIOPAMD_OK;
}

// reference :974 (timesVecCmn with addLogTerm = true):  y = beta*y + alpha*(B + Dx) x.
// The reference rebuilds the recursive a_k, b_k vectors (O(l^2) passes over n) on every call; here the
// mathematically identical compact form  B = sigma I - [sigma S, Y] M^-1 [sigma S, Y]^T  is applied with the
// middle matrix M cached per update: 2 skinny GEMVs, one 2l x 2l solve, 2 transposed GEMVs.
int hiopamd_hess_lowrank_times_vec(hiopamd_hess_lowrank* h, double beta, double* y, double alpha, const double* x,
                                   int add_log_barrier_term)
{
  if(!h) return HIOPAMD_ERR_ARG;
  if(h->matrix_changed) RC(update_internal_bfgs_representation(h));
  hiopamd_ctx* ctx = h->ctx;
  const int64_t n = h->n;
  const int l = h->l_curr < 0 ? 0 : h->l_curr;
  const double sigma = h->sigma;
  const double* Dx = h->Dx;
  const bool add_log = add_log_barrier_term != 0;   // timesVecCmn's addLogTerm (:975, :1045-1047)
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
    y[i] = (beta == 0.0 ? 0.0 : beta * y[i]) + alpha * (sigma + (add_log ? Dx[i] : 0.0)) * x[i];
  }));
  if(l == 0) return HIOPAMD_OK;
  double* sy = h->dsmall + 4 * h->l_max;
  RC(hiopamd_mat_times_vec(ctx, l, n, h->St, n, 0.0, sy, sigma, x));
  RC(hiopamd_mat_times_vec(ctx, l, n, h->Yt, n, 0.0, sy + l, 1.0, x));
  RC(allreduce_dev(ctx, sy, (size_t)2 * l, HIOPAMD_SUM));
  RC(small_lu_apply(ctx, 2 * l, h->dSSlu, h->dSSpiv, 1, sy, 2 * l, sy, 2 * l));
  RC(hiopamd_mat_trans_times_vec(ctx, l, n, h->St, n, 1.0, y, -alpha * sigma, sy));
  RC(hiopamd_mat_trans_times_vec(ctx, l, n, h->Yt, n, 1.0, y, -alpha, sy + l));
  return HIOPAMD_OK;
}

}  // extern "C"

// ===========================================================================================
// hiopKKTLinSysLowRank
// ===========================================================================================
struct hiopamd_kkt_lowrank {
  hiopamd_ctx* ctx = nullptr;
  hiopamd_hess_lowrank* H = nullptr;
  int64_t n = 0;
  int m_eq = 0, m_ineq = 0;
  double *J = nullptr;        // k x n   ([Jc; Jd]) owned copy, used when the caller's Jc / Jd are not adjacent
  const double* Jcur = nullptr;   // the [Jc; Jd] in use: K->J or the caller's own contiguous storage (no copy)
  double *N = nullptr;        // k x k
  double *Dd_inv = nullptr;   // m_ineq
  double *Dx = nullptr;       // n
  double *rhs = nullptr;      // k
  double *tsy = nullptr;      // k + 4 l_max: [t ; sy ; z] of solveCompressed
  double *work = nullptr;     // gram / posv workspace
  size_t work_cnt = 0;
  double last_resid = 0.0;
  // N = J (H+Dx)^-1 J^T + Dd^-1 and its equilibrated factor are cached between the solveCompressed calls of one outer
  // iteration (the reference rebuilds them on every call, hiopKKTLinSys.cpp:1132-1135 — SURVEY.md §3.1): valid while
  // neither the Hessian (version), nor the Jacobians / Dd (any update call) changed.  Same kernels, same order, same
  // inputs => bit-identical to the rebuilt N (tests/test_gpu_lowrank.py::test_cached_N_is_bit_identical).
  bool cache_enabled = true;
  bool N_valid = false;
  unsigned long long N_version = 0;
  int N_info = 0;
  // Round 6, the fused solve (k <= kFusedMaxK): the equilibrated N is factored by a solver object of order k (its solve is ONE launch of
  // the dataflow solve), and what surrounds that solve runs in three one-workgroup kernels (lowrank_mid_*): 8 launches and one host
  // round trip per solveCompressed instead of ~38 launches.  nvec = [sc | b0 | xs | r | c0 | c1 | xu0 | xu1] (k each) + two 64-bit words (the residual maximum, the arrival counter); h_nrm: pinned, written by the device.
  hiopamd_linsolver* nls = nullptr;
  double* nvec = nullptr;
  double* h_nrm = nullptr;
  double* h_nrm_dev = nullptr;
};
constexpr int kFusedMaxK = 1024;   // the one-workgroup kernels keep a k-vector in LDS

// ---- the fused solveCompressed (round 6) -----------------------------------------------------------------------------------------
// The small algebra of one solveCompressed (reference hiopKKTLinSys.cpp:1110-1187; the formulas are in the comment of
// hiopamd_kkt_lowrank_solve_compressed) as three one-workgroup kernels around ONE launch of the order-k solver's solve.
namespace {
// x <- V^-1 x for the 2l x 2l middle matrix (factors of small_lu_factor_kernel) by ONE WAVE (call with every lane of wave 0): lane i
// holds x_i, a solved entry is broadcast by a shuffle and the lanes below / above subtract their multiple of it — nv dependent steps of
// (shuffle, LDS read, fused multiply-add) instead of nv^2 / 2 dependent memory reads by one thread.  LUs: the factors in LDS, row-major
// nv x nv (unit lower below the diagonal, upper on and above); piv: the row exchanges in the order they were made.  nv <= kMaxV <= 64.
// Returns x_lane (lanes >= nv: 0).  The subtractions of a row happen in the order of small_lu_apply_kernel (c ascending / descending).
__device__ inline double lu_apply_wave(int nv, const double* LUs, const int* __restrict__ piv, double xv, int lane)
{
  for(int k = 0; k < nv; ++k) {   // partial pivoting's exchanges: x_k <-> x_p
    const int p = piv[k];
    const double xk = __shfl(xv, k, 64), xp = __shfl(xv, p, 64);
    xv = (lane == k) ? xp : (lane == p) ? xk : xv;
  }
  for(int c = 0; c + 1 < nv; ++c) {   // unit lower
    const double xc = __shfl(xv, c, 64);
    if(lane > c && lane < nv) xv -= LUs[lane * nv + c] * xc;
  }
  for(int c = nv - 1; c >= 0; --c) {   // upper
    const double d = LUs[c * nv + c];
    const double xc = __shfl(xv, c, 64) / d;
    if(lane == c) xv = xc;
    if(lane < c) xv -= LUs[lane * nv + c] * xc;
  }
  return (lane < nv) ? xv : 0.0;
}

// sc_i = 1 / sqrt(N_ii);  M = upper(diag(sc) N diag(sc)) into the solver object's matrix (both in one pass: the factor of the
// EQUILIBRATED matrix is what DPOSVX-style refinement works with, hiopKKTLinSys.cpp:1192-1330)
__global__ __launch_bounds__(kBlock) void lowrank_equilibrate_kernel(int k, const double* __restrict__ Nm, int64_t ldn, double* __restrict__ sc,
                                                                     double* __restrict__ M)
{
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if(e >= (int64_t)k * k) return;
  const int64_t i = e / k, j = e - i * k;
  const double si = 1.0 / sqrt(Nm[i * ldn + i]), sj = 1.0 / sqrt(Nm[j * ldn + j]);
  if(i == j) sc[i] = si;
  M[e] = (j >= i) ? Nm[i * ldn + j] * si * sj : 0.0;
}

// z = V^-1 sy;  rhs = t - S1Y1 z;  b0 = rhs;  x = rhs .* sc     (nv = 2 l, may be 0)
__global__ __launch_bounds__(kBlock) void lowrank_mid_pre_kernel(int k, int nv, const double* __restrict__ LU, const int* __restrict__ piv,
                                                                 const double* __restrict__ S1Y1, int64_t kw, const double* __restrict__ t,
                                                                 const double* __restrict__ sy,
                                                                 const double* __restrict__ sc, double* __restrict__ b0, double* __restrict__ x)
{
  __shared__ double zs[kMaxV];
  __shared__ double LUs[kMaxV * kMaxV];
  const int tid = threadIdx.x;
  if(nv > 0) {
    for(int e = tid; e < nv * nv; e += kBlock) LUs[e] = LU[e];
    __syncthreads();
    if(tid < 64) {
      const double v = lu_apply_wave(nv, LUs, piv, (tid < nv) ? sy[tid] : 0.0, tid);
      if(tid < nv) zs[tid] = v;
    }
    __syncthreads();
  }
  for(int i = tid; i < k; i += kBlock) {
    double v = 0.0;
    for(int q = 0; q < nv; ++q) v = fma(S1Y1[(int64_t)i * kw + q], zs[q], v);
    const double r = t[i] - v;
    b0[i] = r;
    x[i] = r * sc[i];
  }
}

// xu = xs .* sc (the solver's answer in the unscaled variables; refine != 0: xu = xu_prev + xs .* sc, xs = the solver's answer for the
// last residual);  r = b0 - N xu (upper triangle of N referenced), c = r .* sc (right-hand side of a refinement step, should the host
// ask for one);  *nrm = max |r_i|.  One wave per row (like sym_upper_residual), kBlock / 64 rows per workgroup; every workgroup forms xu in
// its own LDS from read-only inputs, workgroup 0 also writes it out (to a buffer nobody reads in this launch); the maximum goes through
// an integer atomic on the bit pattern (non-negative doubles order like their bits; a NaN's pattern is above every number's, so it reaches
// the host), and the LAST workgroup to arrive hands it to the pinned word and re-arms the two words for the next launch.
__global__ __launch_bounds__(kBlock) void lowrank_mid_post_kernel(int k, const double* __restrict__ Nm, int64_t ldn, const double* __restrict__ sc,
                                                                  const double* __restrict__ b0, const double* __restrict__ xs,
                                                                  const double* __restrict__ xu_prev, double* __restrict__ xu,
                                                                  double* __restrict__ r, double* __restrict__ c,
                                                                  unsigned long long* __restrict__ acc2, double* __restrict__ nrm_out, int refine)
{
  __shared__ double xl[kFusedMaxK];
  __shared__ unsigned long long wmax[kBlock / 64];
  __shared__ int last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for(int i = tid; i < k; i += kBlock) {
    const double v = refine ? (xu_prev[i] + xs[i] * sc[i]) : (xs[i] * sc[i]);
    xl[i] = v;
    if(blockIdx.x == 0) xu[i] = v;
  }
  __syncthreads();
  unsigned long long mx = 0ull;
  const int row = blockIdx.x * (kBlock / 64) + wave;
  if(row < k) {
    double acc = 0.0;
    for(int j = lane; j < k; j += 64) {
      const double a = (j >= row) ? Nm[(int64_t)row * ldn + j] : Nm[(int64_t)j * ldn + row];
      acc = fma(a, xl[j], acc);
    }
    for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if(lane == 0) {
      const double rv = b0[row] - acc;
      r[row] = rv;
      c[row] = rv * sc[row];
      mx = (unsigned long long)__double_as_longlong(fabs(rv));
    }
  }
  if(lane == 0) wmax[wave] = mx;
  __syncthreads();
  if(tid == 0) {
    unsigned long long m = wmax[0];
    for(int w = 1; w < kBlock / 64; ++w) m = wmax[w] > m ? wmax[w] : m;
    (void)atomicMax(acc2, m);
    __threadfence();
    last = (atomicAdd(acc2 + 1, 1ull) == (unsigned long long)gridDim.x - 1ull) ? 1 : 0;
    if(last) {
      __threadfence();
      const unsigned long long tot = atomicExch(acc2, 0ull);   // (read and re-arm)
      (void)atomicExch(acc2 + 1, 0ull);
      *nrm_out = __longlong_as_double((long long)tot);
    }
  }
}

// t = dy = x; dyc / dyd;  syp = V^-1 (sy - S1Y1^T dy)
__global__ __launch_bounds__(kBlock) void lowrank_mid_tail_kernel(int k, int me, int nv, const double* __restrict__ LU, const int* __restrict__ piv,
                                                                  const double* __restrict__ S1Y1, int64_t kw, const double* __restrict__ x,
                                                                  double* __restrict__ t, double* __restrict__ dyc, double* __restrict__ dyd,
                                                                  const double* __restrict__ sy, double* __restrict__ syp)
{
  __shared__ double ss[kMaxV];
  __shared__ double LUs[kMaxV * kMaxV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for(int i = tid; i < k; i += kBlock) {
    const double v = x[i];
    t[i] = v;
    if(i < me) dyc[i] = v;
    else dyd[i - me] = v;
  }
  if(nv <= 0) return;
  for(int q = wave; q < nv; q += kBlock / 64) {
    double acc = 0.0;
    for(int i = lane; i < k; i += 64) acc = fma(S1Y1[(int64_t)i * kw + q], x[i], acc);
    for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if(lane == 0) ss[q] = sy[q] - acc;
  }
  for(int e = tid; e < nv * nv; e += kBlock) LUs[e] = LU[e];
  __syncthreads();
  if(tid < 64) {
    const double v = lu_apply_wave(nv, LUs, piv, (tid < nv) ? ss[tid] : 0.0, tid);
    if(tid < nv) syp[tid] = v;
  }
}
}  // namespace

static int lowrank_set_J(hiopamd_kkt_lowrank* K, const double* Jc, const double* Jd);
namespace hiopamd {
int posv_refine_impl(hiopamd_ctx* ctx, int k, const double* N_upper, int64_t ldn, double* rhs_inout, double* work,
                     int* info_host, double* resid_host, int reuse_factor);
}

extern "C" {

int hiopamd_kkt_lowrank_create(hiopamd_kkt_lowrank** out, hiopamd_ctx* ctx, hiopamd_hess_lowrank* H)
{
  if(!out || !ctx || !H) return HIOPAMD_ERR_ARG;
  hiopamd_kkt_lowrank* K = new hiopamd_kkt_lowrank();
  K->ctx = ctx;
  K->H = H;
  K->n = H->n;
  K->m_eq = H->m_eq;
  K->m_ineq = H->m_ineq;
  const size_t k = (size_t)(K->m_eq + K->m_ineq), n = (size_t)(K->n > 0 ? K->n : 1);
  const size_t kw = k + 2 * (size_t)H->l_max;
  K->work_cnt = k * kw + k * 2 * H->l_max + 3 * k * k + 16 * k + 64;
  auto A = [](double** p, size_t cnt) { return hipMalloc((void**)p, sizeof(double) * (cnt ? cnt : 1)) == hipSuccess; };
  (void)n;   // K->J (k x n) is allocated lazily, only if the caller's Jacobians are not one contiguous block
  if(!(A(&K->N, k * k) && A(&K->Dd_inv, K->m_ineq) && A(&K->Dx, n) && A(&K->rhs, k) && A(&K->tsy, k + 4 * (size_t)H->l_max + 8) &&
       A(&K->work, K->work_cnt))) {
    hiopamd_kkt_lowrank_destroy(K);
    return HIOPAMD_ERR_HIP;
  }
  if(k > 0 && k <= (size_t)kFusedMaxK) {
    bool ok = hiopamd_linsolver_create(&K->nls, ctx, (int)k) == HIOPAMD_OK;
    ok = ok && A(&K->nvec, 8 * k + 2);
    ok = ok && hipMemsetAsync(K->nvec, 0, sizeof(double) * (8 * k + 2), ctx->stream) == hipSuccess;   // (the two words start at zero; the kernel re-arms them)
    ok = ok && hipHostMalloc((void**)&K->h_nrm, 4 * sizeof(double), hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void**)&K->h_nrm_dev, K->h_nrm, 0) == hipSuccess;
    if(!ok) {
      hiopamd_kkt_lowrank_destroy(K);
      return HIOPAMD_ERR_HIP;
    }
    (void)hiopamd_linsolver_set_retry_copy(K->nls, 0);   // (N is rebuilt by this object if a factorisation has to be repeated)
  }
  *out = K;
  return HIOPAMD_OK;
}

int hiopamd_kkt_lowrank_destroy(hiopamd_kkt_lowrank* K)
{
  if(!K) return HIOPAMD_OK;
  (void)hipStreamSynchronize(K->ctx->stream);
  double* ps[] = {K->J, K->N, K->Dd_inv, K->Dx, K->rhs, K->tsy, K->work, K->nvec};
  for(double* p : ps) (void)hipFree(p);
  if(K->nls) (void)hiopamd_linsolver_destroy(K->nls);
  if(K->h_nrm) (void)hipHostFree(K->h_nrm);
  delete K;
  return HIOPAMD_OK;
}

// reference :1057-1096: Dx = zl/sxl (on ixl) + zu/sxu (on ixu); Dd = vl/sdl (on idl) + vu/sdu (on idu)
int hiopamd_kkt_lowrank_update(hiopamd_kkt_lowrank* K, const double* zl, const double* sxl, const double* ixl,
                               const double* zu, const double* sxu, const double* ixu, const double* vl,
                               const double* sdl, const double* idl, const double* vu, const double* sdu,
                               const double* idu, const double* Jc, const double* Jd)
{
  if(!K) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = K->ctx;
  const int64_t n = K->n;
  double* Dx = K->Dx;
  // fused: setToZero + two axdzpy_w_pattern (:1072-1074)
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
    double d = 0.0;
    if(ixl[i] == 1.0) d += zl[i] / sxl[i];
    if(ixu[i] == 1.0) d += zu[i] / sxu[i];
    Dx[i] = d;
  }));
  RC(hiopamd_hess_lowrank_update_log_barrier_diagonal(K->H, Dx));
  double* Ddi = K->Dd_inv;
  RC(launch_ew(ctx, K->m_ineq, [=] __device__(int64_t i) {
    double d = 0.0;
    if(idl[i] == 1.0) d += vl[i] / sdl[i];
    if(idu[i] == 1.0) d += vu[i] / sdu[i];
    Ddi[i] = 1.0 / d;   // (:1081-1088)
  }));
  // J = [Jc; Jd]  (copyRowsFrom, :1127-1128 — done once per update instead of once per solveCompressed)
  RC(lowrank_set_J(K, Jc, Jd));
  K->N_valid = false;
  return HIOPAMD_OK;
}

// direct variant for callers that already hold Dx and Dd (= vl/sdl + vu/sdu)
// [Jc; Jd] as one k x n block: borrowed if the caller already stores Jd right after Jc (the 2 GB copy per update is
// 1.6 ms at n_local = 1.25e6, k = 200), copied otherwise
static int lowrank_set_J(hiopamd_kkt_lowrank* K, const double* Jc, const double* Jd)
{
  const int64_t n = K->n;
  if(K->m_eq == 0 && K->m_ineq > 0) {
    K->Jcur = Jd;
  } else if(K->m_ineq == 0 || Jd == Jc + (int64_t)K->m_eq * n) {
    K->Jcur = Jc;
  } else {
    if(!K->J && hipMalloc((void**)&K->J, sizeof(double) * (size_t)(K->m_eq + K->m_ineq) * (size_t)(n > 0 ? n : 1)) != hipSuccess)
      return HIOPAMD_ERR_HIP;
    RC(hiopamd_vec_copy(K->ctx, (int64_t)K->m_eq * n, K->J, Jc));
    RC(hiopamd_vec_copy(K->ctx, (int64_t)K->m_ineq * n, K->J + (int64_t)K->m_eq * n, Jd));
    K->Jcur = K->J;
  }
  return HIOPAMD_OK;
}

int hiopamd_kkt_lowrank_set_jacobians(hiopamd_kkt_lowrank* K, const double* Jc, const double* Jd)
{
  if(!K || (K->m_eq > 0 && !Jc) || (K->m_ineq > 0 && !Jd)) return HIOPAMD_ERR_ARG;
  K->N_valid = false;
  return lowrank_set_J(K, Jc, Jd);
}

double* hiopamd_kkt_lowrank_Dd_inv(hiopamd_kkt_lowrank* K) { return K ? K->Dd_inv : nullptr; }
double* hiopamd_kkt_lowrank_J(hiopamd_kkt_lowrank* K) { return K ? const_cast<double*>(K->Jcur ? K->Jcur : K->J) : nullptr; }
hiopamd_hess_lowrank* hiopamd_kkt_lowrank_hess(hiopamd_kkt_lowrank* K) { return K ? K->H : nullptr; }
int hiopamd_kkt_lowrank_dims(const hiopamd_kkt_lowrank* K, int64_t* n_local_host, int* m_eq_host, int* m_ineq_host)
{
  if(!K) return HIOPAMD_ERR_ARG;
  if(n_local_host) *n_local_host = K->n;
  if(m_eq_host) *m_eq_host = K->m_eq;
  if(m_ineq_host) *m_ineq_host = K->m_ineq;
  return HIOPAMD_OK;
}

int hiopamd_kkt_lowrank_update_diag(hiopamd_kkt_lowrank* K, const double* Dx_in, const double* Dd, const double* Jc,
                                    const double* Jd)
{
  if(!K) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = K->ctx;
  const int64_t n = K->n;
  RC(hiopamd_vec_copy(ctx, n, K->Dx, Dx_in));
  RC(hiopamd_hess_lowrank_update_log_barrier_diagonal(K->H, K->Dx));
  double* Ddi = K->Dd_inv;
  RC(launch_ew(ctx, K->m_ineq, [=] __device__(int64_t i) { Ddi[i] = 1.0 / Dd[i]; }));
  RC(lowrank_set_J(K, Jc, Jd));
  K->N_valid = false;
  return HIOPAMD_OK;
}

// reference :1110-1187.  rx is modified (as in the reference, :1178); ryc, ryd are inputs.
int hiopamd_kkt_lowrank_solve_compressed(hiopamd_kkt_lowrank* K, double* rx, const double* ryc, const double* ryd,
                                         double* dx, double* dyc, double* dyd, int* ok_host)
{
  if(!K) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = K->ctx;
  const int64_t n = K->n;
  const int me = K->m_eq, mi = K->m_ineq, k = me + mi;
  if(ok_host) *ok_host = 1;
  const bool reuse = K->cache_enabled && K->N_valid && K->N_version == K->H->version;
  if(!reuse) {
    // N = J (H+Dx)^-1 J^T                                                     (:1132)
    RC(hiopamd_hess_lowrank_sym_mat_times_inverse_times_mat_trans(K->H, 0.0, K->N, k, 1.0, K->Jcur, K->work));
    // N[me.., me..] += Dd^-1                                                   (:1135)
    RC(hiopamd_mat_add_sub_diagonal(ctx, K->N, k, me, 1.0, K->Dd_inv, 0, mi));
  }
  // The reference's sequence is  dx = (H+Dx)^-1 rx (:1147);  rhs = J dx - [ryc; ryd] (:466, :1157);  N dy = rhs (:1169);
  // rx -= J^T dy;  dx = (H+Dx)^-1 rx (:1178-1180)  -- two applications of the low-rank inverse, each with its own all-reduce of
  // [sigma S; Y] DhInv r, and the all-reduce of rhs: three collectives behind the one of N.  Round 4 writes the same algebra with
  // (H+Dx)^-1 = DhInv - DhInv [sigma S, Y]^T V^-1 [sigma S; Y] DhInv  multiplied out against J, using the block
  // S1Y1 = J DhInv [sigma S, Y]^T (k x 2l) that the build of N left in the work area (replicated, already reduced):
  //    w   = DhInv rx                       t  = J w - [ryc; ryd]        sy = [sigma S; Y] w     -> ONE all-reduce of [t; sy]
  //    rhs = t - S1Y1 V^-1 sy               ( = J dx - ry )
  //    sy' = sy - S1Y1^T dy                 ( = [sigma S; Y] DhInv (rx - J^T dy): no second reduction )
  //    dx  = DhInv (rx - J^T dy) - DhInv [sigma S, Y]^T V^-1 sy'
  // Per call: 2 passes over J, 4 over the 2l secant rows (8 before), ONE collective (+ the one of N when it is rebuilt).
  hiopamd_hess_lowrank* H = K->H;
  if(H->matrix_changed) RC(update_internal_bfgs_representation(H));
  const int l = H->l_curr < 0 ? 0 : H->l_curr;
  const int kw = k + 2 * l;
  const double* DhInv = H->DhInv;
  const double sigma = H->sigma;
  const double* S1Y1 = K->work + k;           // columns k .. k+2l of G (leading dimension kw), scaled by sigma where it applies
  double* t = K->tsy;                         // [t (k) ; sy (2l)] contiguous: one buffer for the collective
  double* sy = t + k;
  double* z = sy + 2 * (size_t)H->l_max;
  if(K->nls && n > 0 && 2 * l <= kMaxV) {
    // ---- round 6: the same algebra in 8 launches and one host round trip
    //  (1, 2)  [t; sy] = [J; sigma S; Y] (DhInv .* rx) - [ry; 0]: ONE pass over J and the secant rows, DhInv applied on the way in
    const double* Ag[3] = {K->Jcur, H->St, H->Yt};
    const int mg[3] = {k, l, l};
    const double ag[3] = {1.0, sigma, 1.0};
    const bool sub = ctx->comm_rank == 0;   // only rank 0 subtracts ry, then all-reduce (:466, :1157)
    RC(gemv_n_groups(ctx, n, n, l > 0 ? 3 : 1, Ag, mg, rx, DhInv, t, ag, sub ? (me > 0 ? ryc : ryd) : nullptr, sub ? (me > 0 ? me : mi) : 0, ryd));
    RC(allreduce_dev(ctx, t, (size_t)k + 2 * (size_t)l, HIOPAMD_SUM));
    // scale, rhs, solver's vector, residual, correction (two copies: a launch reads the solver's answer in one and writes the next
    // right-hand side into the other), solution (two copies for the same reason)
    double *sc = K->nvec, *b0 = sc + k, *xk = b0 + k, *rk = xk + k, *c0 = rk + k, *c1 = c0 + k, *xu0 = c1 + k, *xu1 = xu0 + k;
    unsigned long long* acc2 = reinterpret_cast<unsigned long long*>(xu1 + k);
    if(!reuse) {
      //  the equilibrated N into the solver object, factored there (once per rebuilt N)
      int info = 0;
      for(int attempt = 0; attempt < 2; ++attempt) {
        hipLaunchKernelGGL(lowrank_equilibrate_kernel, dim3((unsigned)(((int64_t)k * k + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, k, K->N,
                           (int64_t)k, sc, hiopamd_linsolver_sys_matrix(K->nls));
        int nneg = 0;
        const int rcf = hiopamd_linsolver_matrix_changed(K->nls, &nneg);
        if(rcf == HIOPAMD_ERR_TIMEOUT && attempt == 0) continue;   // (dataflow factorisation of a large k gave up: once more, stepwise kernels)
        if(rcf != HIOPAMD_OK) return rcf;
        info = (nneg != 0) ? 1 : 0;                                 // not (numerically) positive definite: DPOSVX INFO > 0
        if(!hiopamd_linsolver_factored(K->nls)) info = 2;           // singular: nothing to solve with
        break;
      }
      K->N_info = info;
    }
    K->N_version = K->H->version;
    K->N_valid = true;
    if(K->N_info != 0 && ok_host) *ok_host = 0;
    double resid = 0.0;
    double* syp = z;   // sy' = V^-1 (sy - S1Y1^T dy) goes here: sy itself survives (a refinement repeats the end of the call from it)
    auto finish = [&](const double* xs_) -> int {
      //  (6) dy out, sy' = V^-1 (sy - S1Y1^T dy)
      hipLaunchKernelGGL(lowrank_mid_tail_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, k, me, 2 * l, H->dVlu, H->dVpiv, S1Y1, (int64_t)kw, xs_, t, dyc, dyd, sy, syp);
      //  (7, 8) rx -= J^T dy and, per column, dx = DhInv (rx - [sigma S, Y]^T sy')
      GemvtTail tail{H->St, H->Yt, l, n, sigma, syp, DhInv, dx};
      return gemv_t_tail(ctx, k, n, K->Jcur, n, rx, -1.0, t, tail);
    };
    if(K->N_info != 2) {
      //  (3) z = V^-1 sy, rhs = t - S1Y1 z, scaled   (4) the solve   (5) unscale, residual, its norm to the host
      hipLaunchKernelGGL(lowrank_mid_pre_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, k, 2 * l, H->dVlu, H->dVpiv, S1Y1, (int64_t)kw, t, sy, sc, b0, xk);
      RC(hiopamd_linsolver_solve(K->nls, xk, 1));
      const unsigned gpost = (unsigned)((k + kBlock / 64 - 1) / (kBlock / 64));
      hipLaunchKernelGGL(lowrank_mid_post_kernel, dim3(gpost), dim3(kBlock), 0, ctx->stream, k, K->N, (int64_t)k, sc, b0, xk, xu0, xu0, rk, c0, acc2, K->h_nrm_dev, 0);
      // The rest of the call is queued at once, on the expectation that this residual passes (it does unless N is badly conditioned):
      // the J^T pass runs while the host waits for the norm, instead of after it.
      RC(finish(xu0));
      // ONE host round trip: the solver's status read-back (the error word of its dataflow solve) synchronises the stream, the norm has
      // then arrived as well
      auto wait_norm = [&](double* out) -> int {
        HIOPAMD_CHECK(hipGetLastError());
        int oks = 1;
        RC(hiopamd_linsolver_solve_status(K->nls, &oks));
        HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));   // (no-op behind the status read-back; the wait itself when the solve was stepwise)
        if(!oks) return HIOPAMD_ERR_SOLVE;
        *out = K->h_nrm[0];
        return HIOPAMD_OK;
      };
      RC(wait_norm(&resid));
      const int MAX_ITER_REFIN = 3;
      if(!(resid < 1e-8)) {
        // refinement (hiopKKTLinSys.cpp:1192-1330, up to three steps), then the end of the call once more: what the expectation above
        // already applied to rx is taken back first (t still holds that dy)
        RC(hiopamd_mat_trans_times_vec(ctx, k, n, K->Jcur, n, 1.0, rx, 1.0, t));
        double* xsol = xu0;
        for(int it = 1; it <= MAX_ITER_REFIN; ++it) {
          double* cin = (it & 1) ? c0 : c1;    // the right-hand side r .* sc of this step, answered in place
          double* cout = (it & 1) ? c1 : c0;
          double* xnew = (it & 1) ? xu1 : xu0;
          RC(hiopamd_linsolver_solve(K->nls, cin, 1));
          hipLaunchKernelGGL(lowrank_mid_post_kernel, dim3(gpost), dim3(kBlock), 0, ctx->stream, k, K->N, (int64_t)k, sc, b0, cin, xsol, xnew, rk, cout, acc2, K->h_nrm_dev, 1);
          xsol = xnew;
          RC(wait_norm(&resid));
          if(resid < 1e-8) break;
        }
        RC(finish(xsol));
      }
    } else {
      RC(hiopamd_vec_copy(ctx, k, xu0, t));   // (the old path left the right-hand side in place for a singular N)
      RC(finish(xu0));
    }
    K->last_resid = resid;
    HIOPAMD_CHECK(hipGetLastError());
    return HIOPAMD_OK;
  }
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) { dx[i] = rx[i] * DhInv[i]; }));   // w lives in dx until the end
  if(ctx->comm_rank == 0) {   // only rank 0 subtracts ry, then all-reduce (:466, :1157)
    RC(launch_ew(ctx, k, [=] __device__(int64_t i) { t[i] = (i < me) ? ryc[i] : ryd[i - me]; }));
    RC(hiopamd_mat_times_vec(ctx, k, n, K->Jcur, n, -1.0, t, 1.0, dx));
  } else {
    RC(hiopamd_mat_times_vec(ctx, k, n, K->Jcur, n, 0.0, t, 1.0, dx));
  }
  if(l > 0) {
    RC(hiopamd_mat_times_vec(ctx, l, n, H->St, n, 0.0, sy, sigma, dx));
    RC(hiopamd_mat_times_vec(ctx, l, n, H->Yt, n, 0.0, sy + l, 1.0, dx));
  }
  RC(allreduce_dev(ctx, t, (size_t)k + 2 * (size_t)l, HIOPAMD_SUM));
  if(l > 0) {
    RC(small_lu_apply(ctx, 2 * l, H->dVlu, H->dVpiv, 1, sy, 2 * l, z, 2 * l));   // solveWithV (:677)
    RC(hiopamd_mat_times_vec(ctx, k, 2 * l, S1Y1, kw, 1.0, t, -1.0, z));
  }
  // solve N [dyc; dyd] = rhs with equilibration + refinement                  (:1169, solveWithRefin :1192)
  int info = reuse ? K->N_info : 0;
  double resid = 0.0;
  double* pw = K->work + (size_t)k * (k + 2 * K->H->l_max) + (size_t)k * 2 * K->H->l_max;
  RC(posv_refine_impl(ctx, k, K->N, k, t, pw, &info, &resid, reuse ? 1 : 0));
  K->N_info = info;
  K->N_version = K->H->version;   // (the Hessian's lazily refreshed internal representation does not bump the version)
  K->N_valid = true;
  K->last_resid = resid;
  if(info != 0 && ok_host) *ok_host = 0;
  RC(launch_ew(ctx, k, [=] __device__(int64_t i) {
    if(i < me) dyc[i] = t[i];
    else dyd[i - me] = t[i];
  }));
  // rx = rx - J^T [dyc; dyd]  (rx is modified, as in the reference :1178)
  RC(hiopamd_mat_trans_times_vec(ctx, k, n, K->Jcur, n, 1.0, rx, -1.0, t));
  if(l > 0) {
    RC(hiopamd_mat_trans_times_vec(ctx, k, 2 * l, S1Y1, kw, 1.0, sy, -1.0, t));
    RC(small_lu_apply(ctx, 2 * l, H->dVlu, H->dVpiv, 1, sy, 2 * l, sy, 2 * l));
    double* res = H->nv1;
    RC(hiopamd_mat_trans_times_vec(ctx, l, n, H->St, n, 0.0, res, sigma, sy));
    RC(hiopamd_mat_trans_times_vec(ctx, l, n, H->Yt, n, 1.0, res, 1.0, sy + l));
    RC(launch_ew(ctx, n, [=] __device__(int64_t i) { dx[i] = rx[i] * DhInv[i] - res[i] * DhInv[i]; }));
  } else {
    RC(launch_ew(ctx, n, [=] __device__(int64_t i) { dx[i] = rx[i] * DhInv[i]; }));
  }
  return HIOPAMD_OK;
}

double* hiopamd_kkt_lowrank_N(hiopamd_kkt_lowrank* K) { return K ? K->N : nullptr; }
int hiopamd_kkt_lowrank_set_cache(hiopamd_kkt_lowrank* K, int enable)
{
  if(!K) return HIOPAMD_ERR_ARG;
  K->cache_enabled = enable != 0;
  K->N_valid = false;
  return HIOPAMD_OK;
}
double hiopamd_kkt_lowrank_last_residual(const hiopamd_kkt_lowrank* K) { return K ? K->last_resid : -1.0; }

}  // extern "C"
