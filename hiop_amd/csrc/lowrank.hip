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

// Solve V X = B for `nrhs` right-hand sides by LU with partial pivoting; V (nv x nv, full symmetric,
// row-major, ld = nv) is NOT modified; B is nrhs x nv row-major (each ROW is one right-hand side — the
// "RHS_fortran" view of the reference, hiopHessianLowRank.cpp:601-606) and is overwritten by the solutions.
// Every workgroup factors V redundantly in LDS (nv <= 64) and solves 256 right-hand sides.
__global__ __launch_bounds__(kBlock) void small_lu_solve_kernel(int nv, const double* __restrict__ V, int nrhs,
                                                                double* __restrict__ B, int64_t ldb, int* __restrict__ info)
{
  __shared__ double LU[kMaxV][kMaxV + 1];
  __shared__ int piv[kMaxV];
  __shared__ int pr;
  const int tid = threadIdx.x;
  for(int e = tid; e < nv * nv; e += kBlock) LU[e / nv][e % nv] = V[e];
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
      piv[k] = p;
      if((best == 0.0 || !isfinite(best)) && blockIdx.x == 0) atomicCAS(info, 0, k + 1);
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
  // one right-hand side per thread, kept in LDS-free local storage of fixed size
  const int j = blockIdx.x * kBlock + tid;
  if(j >= nrhs) return;
  double x[kMaxV];
  double* b = B + (int64_t)j * ldb;
#pragma unroll 1
  for(int i = 0; i < nv; ++i) x[i] = b[i];
  for(int k = 0; k < nv; ++k) {
    const int p = piv[k];
    if(p != k) {
      const double t = x[k];
      x[k] = x[p];
      x[p] = t;
    }
  }
  for(int i = 1; i < nv; ++i) {
    double acc = x[i];
    for(int c = 0; c < i; ++c) acc -= LU[i][c] * x[c];
    x[i] = acc;
  }
  for(int i = nv - 1; i >= 0; --i) {
    double acc = x[i];
    for(int c = i + 1; c < nv; ++c) acc -= LU[i][c] * x[c];
    x[i] = acc / LU[i][i];
  }
  for(int i = 0; i < nv; ++i) b[i] = x[i];
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

// three dots in one pass: out = [x.y, x.x, y.y]
// Secant update, Jacobian part in ONE pass (reference: four GEMVs over Jc, Jc_prev, Jd, Jd_prev followed by two full copies
// Jac_prev <- Jac, hiopHessianLowRank.cpp:293-299,366-371 — 8 x the Jacobian's bytes; here 3 x):
//   y_new[j] += sum_r (J[r][j] - Jprev[r][j]) * mult[r]      and      Jprev[r][j] = J[r][j]
// one thread per two adjacent columns, the multipliers staged in LDS per chunk of rows.
constexpr int SJ_ROWCHUNK = 256;
__global__ __launch_bounds__(kBlock) void secant_jac_kernel(int m, int64_t n, const double* __restrict__ J, double* __restrict__ Jp,
                                                            const double* __restrict__ mult, double* __restrict__ y_new)
{
  __shared__ double ms[SJ_ROWCHUNK];
  const int64_t j0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2;
  const bool two = (j0 + 1 < n) && ((n & 1) == 0) && ((((uintptr_t)J) & 15) == 0) && ((((uintptr_t)Jp) & 15) == 0);
  double a0 = 0.0, a1 = 0.0;
  for(int rb = 0; rb < m; rb += SJ_ROWCHUNK) {
    const int rc = (m - rb < SJ_ROWCHUNK) ? (m - rb) : SJ_ROWCHUNK;
    __syncthreads();
    if((int)threadIdx.x < rc) ms[threadIdx.x] = mult[rb + threadIdx.x];
    __syncthreads();
    if(j0 < n) {
      const double* Jr = J + (int64_t)rb * n + j0;
      double* Pr = Jp + (int64_t)rb * n + j0;
      if(two) {
#pragma unroll 4
        for(int r = 0; r < rc; ++r) {
          const double2 v = *reinterpret_cast<const double2*>(Jr + (int64_t)r * n);
          const double2 q = *reinterpret_cast<const double2*>(Pr + (int64_t)r * n);
          a0 = fma(v.x - q.x, ms[r], a0);
          a1 = fma(v.y - q.y, ms[r], a1);
          *reinterpret_cast<double2*>(Pr + (int64_t)r * n) = v;
        }
      } else {
        for(int r = 0; r < rc; ++r) {
          const double v0 = Jr[(int64_t)r * n], q0 = Pr[(int64_t)r * n];
          a0 = fma(v0 - q0, ms[r], a0);
          Pr[(int64_t)r * n] = v0;
          if(j0 + 1 < n) {
            const double v1 = Jr[(int64_t)r * n + 1], q1 = Pr[(int64_t)r * n + 1];
            a1 = fma(v1 - q1, ms[r], a1);
            Pr[(int64_t)r * n + 1] = v1;
          }
        }
      }
    }
  }
  if(j0 < n) {
    y_new[j0] += a0;
    if(j0 + 1 < n) y_new[j0 + 1] += a1;
  }
}

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
  // host mirrors of the tiny BFGS bookkeeping (reference keeps L_, D_ on the host as well)
  std::vector<double> L, D;
  bool have_prev = false;
};

static int allreduce_dev(hiopamd_ctx* ctx, double* buf, size_t count, int op)
{
  // a hook installed on a 1-rank partition is still called (lets the RCCL path run on a single GPU)
  if(!ctx->allreduce) return HIOPAMD_OK;
  return ctx->allreduce(ctx->allreduce_user, buf, count, op, (void*)ctx->stream) == 0 ? HIOPAMD_OK : HIOPAMD_ERR_HIP;
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

static int small_lu_solve(hiopamd_ctx* ctx, int nv, const double* V, int nrhs, double* B, int64_t ldb, int* dinfo)
{
  if(nv == 0 || nrhs == 0) return HIOPAMD_OK;
  if(nv > kMaxV) return HIOPAMD_ERR_ARG;
  HIOPAMD_CHECK(hipMemsetAsync(dinfo, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(small_lu_solve_kernel, dim3((nrhs + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, nv, V, nrhs,
                     B, ldb, dinfo);
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
            A(&h->nv2, n) && A(&h->dL, lm * lm) && A(&h->dD, lm) && A(&h->dG, 3 * lm * lm) && A(&h->dV, 4 * lm * lm) &&
            A(&h->dSS, 4 * lm * lm) && A(&h->dsmall, 16 * 2 * lm + 64);
  ok = ok && hipMalloc((void**)&h->dinfo, 64) == hipSuccess;
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
                  h->dL, h->dD, h->dG,    h->dV, h->dSS,    h->dsmall};
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
  double s_inf = 0.0;
  {
    ReduceNow now(ctx);
    RC(hiopamd_vec_infnorm(ctx, n, s_new, &s_inf));
  }
  if(ctx->allreduce) {
    RC(to_dev(ctx, h->dsmall, &s_inf, sizeof(double)));
    RC(allreduce_dev(ctx, h->dsmall, 1, HIOPAMD_MAX));
    RC(to_host(ctx, &s_inf, h->dsmall, sizeof(double)));
  }
  if(s_inf >= 100 * std::numeric_limits<double>::epsilon()) {
    // y_new += (Jc - Jc_prev)^T yc + (Jd - Jd_prev)^T yd                      (:293-299)
    // (one pass per Jacobian that also refreshes Jc_prev / Jd_prev: secant_jac_kernel)
    {
      const unsigned gx = (unsigned)((n + 2 * (int64_t)kBlock - 1) / (2 * (int64_t)kBlock));
      if(me > 0 && n > 0) hipLaunchKernelGGL(secant_jac_kernel, dim3(gx), dim3(kBlock), 0, ctx->stream, me, n, Jc, h->Jc_prev, yc, y_new);
      if(mi > 0 && n > 0) hipLaunchKernelGGL(secant_jac_kernel, dim3(gx), dim3(kBlock), 0, ctx->stream, mi, n, Jd, h->Jd_prev, yd, y_new);
      HIOPAMD_CHECK(hipGetLastError());
      jac_saved = true;
    }
    // [s^T y, s^T s, y^T y] in one pass + one all-reduce                      (:301)
    dot3_t d3{0, 0, 0};
    RC(launch_reduce<dot3_t>(ctx, n, OpDot3{s_new, y_new}, &d3));
    if(ctx->allreduce) {
      RC(to_dev(ctx, h->dsmall, &d3, sizeof(d3)));
      RC(allreduce_dev(ctx, h->dsmall, 3, HIOPAMD_SUM));
      RC(to_host(ctx, &d3, h->dsmall, sizeof(d3)));
    }
    const double sTy = d3.a, s_nrm2 = std::sqrt(d3.b), y_nrm2 = std::sqrt(d3.c);
    if(sTy > s_nrm2 * y_nrm2 * std::sqrt(std::numeric_limits<double>::epsilon())) {
      if(h->l_max > 0) {
        const int l = h->l_curr;
        std::vector<double> YTs(l > 0 ? l : 1, 0.0);
        if(l > 0) {  // YTs = Yt * s_new (:311)
          RC(hiopamd_mat_times_vec(ctx, l, n, h->Yt, n, 0.0, h->dsmall, 1.0, s_new));
          RC(allreduce_dev(ctx, h->dsmall, l, HIOPAMD_SUM));
          RC(to_host(ctx, YTs.data(), h->dsmall, sizeof(double) * l));
        }
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
          RC(hiopamd_mat_shift_rows(ctx, l, n, h->St, n, -1));
          RC(hiopamd_mat_shift_rows(ctx, l, n, h->Yt, n, -1));
          RC(hiopamd_vec_copy(ctx, n, h->St + (int64_t)(l - 1) * n, s_new));
          RC(hiopamd_vec_copy(ctx, n, h->Yt + (int64_t)(l - 1) * n, y_new));
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
  double* w = h->nv1;
  const double* DhInv = h->DhInv;
  const double sigma = h->sigma;
  // G0 = Yt DhInv Yt^T ; G1 = St (sigma DhInv) Yt^T ; G2 = St (sigma (sigma DhInv - 1)) St^T
  RC(hiopamd_gram_weighted(ctx, l, l, n, h->Yt, n, h->Yt, n, DhInv, 0.0, G, l, 1.0, 1));
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) { w[i] = DhInv[i] * sigma; }));
  RC(hiopamd_gram_weighted(ctx, l, l, n, h->St, n, h->Yt, n, w, 0.0, G + l * l, l, 1.0, 0));
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) { w[i] = (DhInv[i] * sigma - 1.0) * sigma; }));
  RC(hiopamd_gram_weighted(ctx, l, l, n, h->St, n, h->St, n, w, 0.0, G + 2 * l * l, l, 1.0, 1));
  RC(allreduce_dev(ctx, G, (size_t)3 * l * l, HIOPAMD_SUM));                  // (:459)
  HIOPAMD_CHECK(hipMemcpyAsync(h->dL, h->L.data(), sizeof(double) * l * l, hipMemcpyHostToDevice, ctx->stream));
  HIOPAMD_CHECK(hipMemcpyAsync(h->dD, h->D.data(), sizeof(double) * l, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(assemble_V_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, l, G, h->dL, h->dD, h->dV);
  // middle matrix of the compact DIRECT form for timesVec: M = [sigma S^T S, L; L^T, -D]
  RC(hiopamd_gram_weighted(ctx, l, l, n, h->St, n, h->St, n, nullptr, 0.0, G, l, sigma, 1));
  RC(allreduce_dev(ctx, G, (size_t)l * l, HIOPAMD_SUM));
  {
    double* M = h->dSS;
    const double* SS = G;
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
  RC(small_lu_solve(ctx, 2 * l, h->dV, 1, sy, 2 * l, h->dinfo));            // solveWithV (:677)
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
    RC(small_lu_solve(ctx, 2 * l, h->dV, k, S2Y2, 2 * l, h->dinfo));        // (:606)
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
  RC(small_lu_solve(ctx, 2 * l, h->dSS, 1, sy, 2 * l, h->dinfo));
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
};

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
  if(!(A(&K->N, k * k) && A(&K->Dd_inv, K->m_ineq) && A(&K->Dx, n) && A(&K->rhs, k) &&
       A(&K->work, K->work_cnt))) {
    hiopamd_kkt_lowrank_destroy(K);
    return HIOPAMD_ERR_HIP;
  }
  *out = K;
  return HIOPAMD_OK;
}

int hiopamd_kkt_lowrank_destroy(hiopamd_kkt_lowrank* K)
{
  if(!K) return HIOPAMD_OK;
  (void)hipStreamSynchronize(K->ctx->stream);
  double* ps[] = {K->J, K->N, K->Dd_inv, K->Dx, K->rhs, K->work};
  for(double* p : ps) (void)hipFree(p);
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
  // dx = (H+Dx)^-1 rx                                                          (:1147)
  RC(hiopamd_hess_lowrank_solve(K->H, rx, dx));
  // rhs = J dx - [ryc; ryd]   (only rank 0 subtracts, then all-reduce: :466, :1157)
  double* rhs = K->rhs;
  if(ctx->comm_rank == 0) {
    RC(hiopamd_vec_copy(ctx, me, rhs, ryc));
    RC(hiopamd_vec_copy(ctx, mi, rhs + me, ryd));
    RC(hiopamd_mat_times_vec(ctx, k, n, K->Jcur, n, -1.0, rhs, 1.0, dx));
  } else {
    RC(hiopamd_mat_times_vec(ctx, k, n, K->Jcur, n, 0.0, rhs, 1.0, dx));
  }
  RC(allreduce_dev(ctx, rhs, (size_t)k, HIOPAMD_SUM));
  // solve N [dyc; dyd] = rhs with equilibration + refinement                  (:1169, solveWithRefin :1192)
  int info = reuse ? K->N_info : 0;
  double resid = 0.0;
  double* pw = K->work + (size_t)k * (k + 2 * K->H->l_max) + (size_t)k * 2 * K->H->l_max;
  RC(posv_refine_impl(ctx, k, K->N, k, rhs, pw, &info, &resid, reuse ? 1 : 0));
  K->N_info = info;
  K->N_version = K->H->version;   // (the Hessian's lazily refreshed internal representation does not bump the version)
  K->N_valid = true;
  K->last_resid = resid;
  if(info != 0 && ok_host) *ok_host = 0;
  RC(hiopamd_vec_copy(ctx, me, dyc, rhs));
  RC(hiopamd_vec_copy(ctx, mi, dyd, rhs + me));
  // rx = rx - J^T [dyc; dyd] ; dx = (H+Dx)^-1 rx                               (:1178-1180)
  RC(hiopamd_mat_trans_times_vec(ctx, k, n, K->Jcur, n, 1.0, rx, -1.0, rhs));
  RC(hiopamd_hess_lowrank_solve(K->H, rx, dx));
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
