// k x k SPD solve with equilibration and iterative refinement for the reduced system of the
// quasi-Newton low-rank KKT  (N = J H^-1 J^T + Dd^-1, k = #constraints <= a few hundred).
//
// reference: hiopKKTLinSysLowRank::solveWithRefin (src/Optimization/hiopKKTLinSys.cpp:1192-1330):
// DPOSVX(FACT='E') then a residual loop (inf-norm tolerance 1e-8, at most 3 refinements with a fresh
// Cholesky solve of the residual).  Here: diagonal equilibration s_i = 1/sqrt(N_ii) (what DPOEQU
// computes), the equilibrated matrix is factored U^T D U by the same blocked no-pivot kernels as the
// big KKT matrix (for an SPD matrix this is Cholesky with the square roots left out; D>0 is checked
// through the inertia), and the same residual loop runs with device-resident vectors.
#include "device_utils.hpp"

namespace hiopamd {
// y = b - Nsym*x with only the upper triangle of N referenced; one wave per row
__global__ __launch_bounds__(kBlock) void sym_upper_residual(int k, const double* __restrict__ Nm, int64_t ldn,
                                                             const double* __restrict__ x,
                                                             const double* __restrict__ b, double* __restrict__ r)
{
  const int row = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if(row >= k) return;
  double acc = 0.0;
  for(int j = lane; j < k; j += 64) {
    const double a = (j >= row) ? Nm[(int64_t)row * ldn + j] : Nm[(int64_t)j * ldn + row];
    acc = fma(a, x[j], acc);
  }
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if(lane == 0) r[row] = b[row] - acc;
}
}  // namespace hiopamd

using namespace hiopamd;

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

// reuse_factor != 0: `work` still holds the equilibration + factor of the SAME matrix from an earlier call (the low-rank
// KKT caches N between the solves of one outer iteration): only the solve + residual loop run, *info_host keeps the
// caller's cached value.  Every operation that does run is the one the full call would run, so results are bit-identical.
namespace hiopamd {
int posv_refine_impl(hiopamd_ctx* ctx, int k, const double* N_upper, int64_t ldn, double* rhs_inout, double* work,
                     int* info_host, double* resid_host, int reuse_factor);
}
extern "C" int hiopamd_posv_refine(hiopamd_ctx* ctx, int k, const double* N_upper, int64_t ldn, double* rhs_inout,
                                   double* work, int* info_host, double* resid_host)
{
  return hiopamd::posv_refine_impl(ctx, k, N_upper, ldn, rhs_inout, work, info_host, resid_host, 0);
}

int hiopamd::posv_refine_impl(hiopamd_ctx* ctx, int k, const double* N_upper, int64_t ldn, double* rhs_inout, double* work,
                              int* info_host, double* resid_host, int reuse_factor)
{
  if(k < 0 || !info_host) return HIOPAMD_ERR_ARG;
  if(!reuse_factor) *info_host = 0;
  if(resid_host) *resid_host = 0.0;
  if(k == 0) return HIOPAMD_OK;
  // work layout: M (k*k) | sc (k) | dinv (k) | b0 (k) | x (k) | r (k) | t (k)
  double* M = work;
  double* sc = M + (size_t)k * k;
  double* dinv = sc + k;
  double* b0 = dinv + k;
  double* x = b0 + k;
  double* r = x + k;
  double* t = r + k;
  const double* Nm = N_upper;
  if(!reuse_factor) {
    RC(launch_ew(ctx, k, [=] __device__(int64_t i) { sc[i] = 1.0 / sqrt(Nm[i * ldn + i]); }));
    RC(launch_ew(ctx, (int64_t)k * k, [=] __device__(int64_t e) {
      const int64_t i = e / k, j = e - i * k;
      M[e] = (j >= i) ? Nm[i * ldn + j] * sc[i] * sc[j] : 0.0;
    }));
    int inertia[3] = {0, 0, 0};
    int rc = hiopamd_ldlt_factor(ctx, k, M, k, dinv, inertia);
    if(rc == HIOPAMD_ERR_SINGULAR || inertia[1] > 0 || inertia[2] > 0) {
      *info_host = 1;  // not (numerically) positive definite -- DPOSVX INFO>0
      if(rc == HIOPAMD_ERR_SINGULAR) {
        *info_host = 2;   // singular: nothing to solve with
        return HIOPAMD_OK;
      }
    } else if(rc != HIOPAMD_OK) {
      return rc;
    }
  } else if(*info_host == 2) {
    return HIOPAMD_OK;
  }
  // b0 = b ; x = S * (M^-1 (S b))
  RC(launch_ew(ctx, k, [=] __device__(int64_t i) {
    const double b = rhs_inout[i];
    b0[i] = b;
    x[i] = b * sc[i];
  }));
  RC(hiopamd_ldlt_solve(ctx, k, M, k, dinv, x, 1));
  RC(hiopamd_vec_component_mult(ctx, k, x, sc));
  const int MAX_ITER_REFIN = 3;
  double nrm = 0.0;
  for(int it = 0;; ++it) {
    const int64_t threads = (int64_t)k * 64;
    hipLaunchKernelGGL(sym_upper_residual, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       ctx->stream, k, Nm, ldn, x, b0, r);
    {
      ReduceNow now(ctx);
      RC(hiopamd_vec_infnorm(ctx, k, r, &nrm));
    }
    if(nrm < 1e-8 || it >= MAX_ITER_REFIN) break;
    RC(launch_ew(ctx, k, [=] __device__(int64_t i) { t[i] = r[i] * sc[i]; }));
    RC(hiopamd_ldlt_solve(ctx, k, M, k, dinv, t, 1));
    RC(launch_ew(ctx, k, [=] __device__(int64_t i) { x[i] += t[i] * sc[i]; }));
  }
  if(resid_host) *resid_host = nrm;
  RC(hiopamd_vec_copy(ctx, k, rhs_inout, x));
  return HIOPAMD_OK;
}
