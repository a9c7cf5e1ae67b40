// Device-resident user callbacks of the reference's MDS example problem.
//
// reference: src/Drivers/MDS/NlpMdsEx1.hpp (class MdsEx1, the problem hiopInterfaceMDS examples and tests run) and its RAJA
// twin src/Drivers/MDS/NlpMdsRajaEx1.cpp — the callbacks of hiopInterfaceMDS (src/Interface/hiopInterface.hpp:582-780) with
// every array argument a DEVICE pointer, which is what `mem_space = device` hands to the user (and what the C FFI of
// src/Interface/hiopInterface.h:63-98 would carry).  With them an iteration of the MDS path moves no Jacobian / Hessian value
// through the host: eval_Jac_* / eval_Hess_Lagr write straight into the arrays hiopamd_kkt_mds_set_values() consumes.
//
//   min  0.5 sum_i x_i (x_i - 1) + 0.5 y' Qd y + 0.5 s' s          x, s in R^ns, y in R^nd
//   s.t. x + s + Md y = 0           (ns equalities, Md = -1)
//        -2 <= x_1 + e's + e'y <= 2,   x_2 + e'y <= 2,   -2 <= x_3 + e'y
//        x <= 3, s >= 0, -4 <= y_1 <= 4
#include "common.hpp"

namespace hiopamd {

// Qd (:79-88): 1e-8 everywhere, + 2 on the diagonal, the entries (i, i+1) and (i+1, i) SET to 1 for i = 1 .. nd-2
__global__ __launch_bounds__(kBlock) void mdsex1_q_kernel(int nd, double* __restrict__ Q)
{
  const int64_t total = (int64_t)nd * nd;
  for(int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / nd), c = (int)(e % nd);
    double v = 1e-8;
    if(r == c) v = 1e-8 + 2.0;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    if(hi == lo + 1 && lo >= 1 && lo <= nd - 2) v = 1.0;
    Q[e] = v;
  }
}

// get_vars_info (:115-141)
__global__ __launch_bounds__(kBlock) void mdsex1_vars_kernel(int ns, int nd, double* __restrict__ xlow, double* __restrict__ xupp)
{
  const int64_t n = 2 * (int64_t)ns + nd;
  for(int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double lo, up;
    if(i < ns) {
      lo = -1e+20;
      up = 3.0;
    } else if(i < 2 * (int64_t)ns) {
      lo = 0.0;
      up = 1e+20;
    } else if(i == 2 * (int64_t)ns) {
      lo = -4.0;
      up = 4.0;
    } else {
      lo = -1e+20;
      up = 1e+20;
    }
    if(xlow) xlow[i] = lo;
    if(xupp) xupp[i] = up;
  }
}

// get_cons_info (:143-163)
__global__ __launch_bounds__(kBlock) void mdsex1_consinfo_kernel(int ns, double* __restrict__ clow, double* __restrict__ cupp)
{
  const int m = ns + 3;
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    double lo = 0.0, up = 0.0;
    if(i == ns) {
      lo = -2.0;
      up = 2.0;
    } else if(i == ns + 1) {
      lo = -1e+20;
      up = 2.0;
    } else if(i == ns + 2) {
      lo = -2.0;
      up = 1e+20;
    }
    if(clow) clow[i] = lo;
    if(cupp) cupp[i] = up;
  }
}

// one workgroup, fixed summation order: sums[0] = e's, sums[1] = e'y, sums[2] = sum_i x_i (x_i - 1) + s_i^2
__global__ __launch_bounds__(1024) void mdsex1_sums_kernel(int ns, int nd, const double* __restrict__ x, double* __restrict__ sums)
{
  __shared__ double red[3][1024];
  const int tid = threadIdx.x;
  const double* s = x + ns;
  const double* y = x + 2 * (int64_t)ns;
  double a = 0.0, b = 0.0, c = 0.0;
  for(int i = tid; i < ns; i += 1024) {
    a += s[i];
    c += x[i] * (x[i] - 1.0) + s[i] * s[i];
  }
  for(int i = tid; i < nd; i += 1024) b += y[i];
  red[0][tid] = a;
  red[1][tid] = b;
  red[2][tid] = c;
  __syncthreads();
  for(int w = 512; w >= 1; w >>= 1) {
    if(tid < w) {
      red[0][tid] += red[0][tid + w];
      red[1][tid] += red[1][tid + w];
      red[2][tid] += red[2][tid + w];
    }
    __syncthreads();
  }
  if(tid < 3) sums[tid] = red[tid][0];
}

// eval_cons (:211-266): all ns + 3 constraints, equalities first
__global__ __launch_bounds__(kBlock) void mdsex1_cons_kernel(int ns, int empty_sp_row, const double* __restrict__ x,
                                                            const double* __restrict__ sums, double* __restrict__ cons)
{
  const double es = sums[0], ey = sums[1];
  const int m = ns + 3;
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    double v;
    if(i < ns) v = x[i] + x[ns + i] - ey;                       // x + s + Md y, Md = -1
    else if(i == ns) v = (ns > 0 ? x[0] : 0.0) + es + ey;
    else if(i == ns + 1) v = ((empty_sp_row || ns < 2) ? 0.0 : x[1]) + ey;
    else v = (ns > 2 ? x[2] : 0.0) + ey;
    cons[i] = v;
  }
}

// eval_grad_f (:269-289), the x and s parts (the y part is Qd y)
__global__ __launch_bounds__(kBlock) void mdsex1_grad_kernel(int ns, const double* __restrict__ x, double* __restrict__ g)
{
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    g[i] = x[i] - 0.5;
    g[ns + i] = x[ns + i];
  }
}

// eval_Jac_cons, equalities (:294-303, :330-341, :385-387): rows (i, i) and (i, ns + i), values 1; dense part Md = -1
__global__ __launch_bounds__(kBlock) void mdsex1_jac_eq_kernel(int ns, int nd, int* __restrict__ iJ, int* __restrict__ jJ,
                                                              double* __restrict__ M, double* __restrict__ JacD)
{
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for(int64_t i = gid; i < ns; i += stride) {
    if(iJ) {
      iJ[2 * i] = (int)i;
      iJ[2 * i + 1] = (int)i;
    }
    if(jJ) {
      jJ[2 * i] = (int)i;
      jJ[2 * i + 1] = (int)(ns + i);
    }
    if(M) {
      M[2 * i] = 1.0;
      M[2 * i + 1] = 1.0;
    }
  }
  if(JacD)
    for(int64_t e = gid; e < (int64_t)ns * nd; e += stride) JacD[e] = -1.0;
}

// eval_Jac_cons, inequalities (:305-327, :343-361, :375-383): row 0 = x_1 and every s, row 1 = x_2 (unless the example is
// run with an empty sparse row), row 2 = x_3; dense part = ones
__global__ __launch_bounds__(kBlock) void mdsex1_jac_ineq_kernel(int ns, int nd, int empty_sp_row, int row_offset,
                                                                int* __restrict__ iJ, int* __restrict__ jJ,
                                                                double* __restrict__ M, double* __restrict__ JacD)
{
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  if(ns > 0) {
    const int nnz = ns + (empty_sp_row ? 2 : 3);
    for(int64_t e = gid; e < nnz; e += stride) {
      int r, c;
      if(e == 0) {
        r = 0;
        c = 0;
      } else if(e <= ns) {
        r = 0;
        c = ns + (int)(e - 1);
      } else if(!empty_sp_row && e == ns + 1) {
        r = 1;
        c = 1;
      } else {
        r = 2;
        c = 2;
      }
      if(iJ) iJ[e] = r + row_offset;
      if(jJ) jJ[e] = c;
      if(M) M[e] = 1.0;
    }
  }
  if(JacD)
    for(int64_t e = gid; e < 3 * (int64_t)nd; e += stride) JacD[e] = 1.0;
}

// eval_Hess_Lagr (:403-440): the constraints are linear; HSS = obj_factor I (2 ns entries), HDD = obj_factor Qd
__global__ __launch_bounds__(kBlock) void mdsex1_hess_kernel(int ns, int nd, double obj_factor, const double* __restrict__ Q,
                                                            int* __restrict__ iH, int* __restrict__ jH, double* __restrict__ M,
                                                            double* __restrict__ HDD)
{
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for(int64_t i = gid; i < 2 * (int64_t)ns; i += stride) {
    if(iH) iH[i] = (int)i;
    if(jH) jH[i] = (int)i;
    if(M) M[i] = obj_factor;
  }
  if(HDD)
    for(int64_t e = gid; e < (int64_t)nd * nd; e += stride) HDD[e] = obj_factor * Q[e];
}

}  // namespace hiopamd

using namespace hiopamd;

struct hiopamd_mdsex1 {
  hiopamd_ctx* ctx = nullptr;
  int ns = 0, nd = 0, empty_sp_row = 0;
  double* Q = nullptr;      // nd x nd
  double* buf = nullptr;    // nd: Qd y
  double* sums = nullptr;   // 3 device scalars (mdsex1_sums_kernel)
};

extern "C" {

int hiopamd_mdsex1_create(hiopamd_mdsex1** out, hiopamd_ctx* ctx, int ns, int nd, int empty_sp_row)
{
  if(!out || !ctx) return HIOPAMD_ERR_ARG;
  auto* p = new hiopamd_mdsex1;
  p->ctx = ctx;
  // (:66-77) a negative size means none; the number of sparse variables is rounded up to a multiple of four
  if(ns < 0) ns = 0;
  else if(4 * (ns / 4) != ns) ns = 4 * ((4 + ns) / 4);
  p->ns = ns;
  p->nd = nd < 0 ? 0 : nd;
  p->empty_sp_row = empty_sp_row ? 1 : 0;
  const size_t qn = (size_t)p->nd * p->nd;
  if(hipMalloc((void**)&p->Q, sizeof(double) * (qn ? qn : 1)) != hipSuccess ||
     hipMalloc((void**)&p->buf, sizeof(double) * (size_t)(p->nd ? p->nd : 1)) != hipSuccess ||
     hipMalloc((void**)&p->sums, sizeof(double) * 4) != hipSuccess) {
    (void)hipFree(p->Q);
    (void)hipFree(p->buf);
    (void)hipFree(p->sums);
    delete p;
    return HIOPAMD_ERR_HIP;
  }
  if(qn) hipLaunchKernelGGL(mdsex1_q_kernel, dim3(grid_for((int64_t)qn)), dim3(kBlock), 0, ctx->stream, p->nd, p->Q);
  HIOPAMD_CHECK(hipGetLastError());
  *out = p;
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_destroy(hiopamd_mdsex1* p)
{
  if(!p) return HIOPAMD_OK;
  (void)hipStreamSynchronize(p->ctx->stream);
  (void)hipFree(p->Q);
  (void)hipFree(p->buf);
  (void)hipFree(p->sums);
  delete p;
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_get_prob_sizes(const hiopamd_mdsex1* p, int64_t* n, int64_t* m)
{
  if(!p || !n || !m) return HIOPAMD_ERR_ARG;
  *n = 2 * (int64_t)p->ns + p->nd;
  *m = (int64_t)p->ns + 3;
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_get_sparse_dense_blocks_info(const hiopamd_mdsex1* p, int* nx_sparse, int* nx_dense, int* nnz_sparse_Jaceq,
                                                int* nnz_sparse_Jacineq, int* nnz_sparse_Hess_Lagr_SS,
                                                int* nnz_sparse_Hess_Lagr_SD)
{
  if(!p || !nx_sparse || !nx_dense || !nnz_sparse_Jaceq || !nnz_sparse_Jacineq || !nnz_sparse_Hess_Lagr_SS ||
     !nnz_sparse_Hess_Lagr_SD)
    return HIOPAMD_ERR_ARG;
  *nx_sparse = 2 * p->ns;
  *nx_dense = p->nd;
  *nnz_sparse_Jaceq = 2 * p->ns;
  *nnz_sparse_Jacineq = (p->ns == 0) ? 0 : (p->empty_sp_row ? 2 : 3) + p->ns;
  *nnz_sparse_Hess_Lagr_SS = 2 * p->ns;
  *nnz_sparse_Hess_Lagr_SD = 0;
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_get_vars_info(hiopamd_mdsex1* p, double* xlow, double* xupp)
{
  if(!p) return HIOPAMD_ERR_ARG;
  const int64_t n = 2 * (int64_t)p->ns + p->nd;
  if(n == 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(mdsex1_vars_kernel, dim3(grid_for(n)), dim3(kBlock), 0, p->ctx->stream, p->ns, p->nd, xlow, xupp);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_get_cons_info(hiopamd_mdsex1* p, double* clow, double* cupp)
{
  if(!p) return HIOPAMD_ERR_ARG;
  hipLaunchKernelGGL(mdsex1_consinfo_kernel, dim3(grid_for(p->ns + 3)), dim3(kBlock), 0, p->ctx->stream, p->ns, clow, cupp);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_get_starting_point(hiopamd_mdsex1* p, double* x0)   /* :442-447 */
{
  if(!p || !x0) return HIOPAMD_ERR_ARG;
  return hiopamd_vec_set_to_constant(p->ctx, 2 * (int64_t)p->ns + p->nd, x0, 1.0);
}

int hiopamd_mdsex1_eval_f(hiopamd_mdsex1* p, const double* x, double* obj_host)
{
  if(!p || !x || !obj_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = p->ctx;
  hipLaunchKernelGGL(mdsex1_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, p->ns, p->nd, x, p->sums);
  HIOPAMD_CHECK(hipGetLastError());
  double yQy = 0.0;
  if(p->nd > 0) {
    const double* y = x + 2 * (int64_t)p->ns;
    int rc = hiopamd_mat_times_vec(ctx, p->nd, p->nd, p->Q, p->nd, 0.0, p->buf, 1.0, y);
    if(rc != HIOPAMD_OK) return rc;
    {
      hiopamd::ReduceNow now(ctx);
      rc = hiopamd_vec_dot(ctx, p->nd, p->buf, y, &yQy);   // synchronises
    }
    if(rc != HIOPAMD_OK) return rc;
  }
  double h[3];
  int rc = hiopamd_copy_d2h(ctx, h, p->sums, sizeof(h));
  if(rc != HIOPAMD_OK) return rc;
  rc = hiopamd_ctx_sync(ctx);
  if(rc != HIOPAMD_OK) return rc;
  *obj_host = 0.5 * h[2] + 0.5 * yQy;
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_eval_grad_f(hiopamd_mdsex1* p, const double* x, double* gradf)
{
  if(!p || !x || !gradf) return HIOPAMD_ERR_ARG;
  if(p->ns > 0) {
    hipLaunchKernelGGL(mdsex1_grad_kernel, dim3(grid_for(p->ns)), dim3(kBlock), 0, p->ctx->stream, p->ns, x, gradf);
    HIOPAMD_CHECK(hipGetLastError());
  }
  if(p->nd > 0)
    return hiopamd_mat_times_vec(p->ctx, p->nd, p->nd, p->Q, p->nd, 0.0, gradf + 2 * (int64_t)p->ns, 1.0, x + 2 * (int64_t)p->ns);
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_eval_cons(hiopamd_mdsex1* p, const double* x, double* cons)
{
  if(!p || !x || !cons) return HIOPAMD_ERR_ARG;
  hipLaunchKernelGGL(mdsex1_sums_kernel, dim3(1), dim3(1024), 0, p->ctx->stream, p->ns, p->nd, x, p->sums);
  hipLaunchKernelGGL(mdsex1_cons_kernel, dim3(grid_for(p->ns + 3)), dim3(kBlock), 0, p->ctx->stream, p->ns, p->empty_sp_row, x,
                     p->sums, cons);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_eval_Jac_cons_eq(hiopamd_mdsex1* p, const double* x, int* iJacS, int* jJacS, double* MJacS, double* JacD)
{
  (void)x;   // linear constraints
  if(!p) return HIOPAMD_ERR_ARG;
  const int64_t work = (int64_t)p->ns * (JacD ? (p->nd > 1 ? p->nd : 1) : 1);
  if(work == 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(mdsex1_jac_eq_kernel, dim3(grid_for(work)), dim3(kBlock), 0, p->ctx->stream, p->ns, p->nd, iJacS, jJacS,
                     MJacS, JacD);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_eval_Jac_cons_ineq(hiopamd_mdsex1* p, const double* x, int row_offset, int* iJacS, int* jJacS, double* MJacS,
                                      double* JacD)
{
  (void)x;
  if(!p) return HIOPAMD_ERR_ARG;
  const int64_t work = (int64_t)p->ns + 3 + 3 * (int64_t)p->nd;
  hipLaunchKernelGGL(mdsex1_jac_ineq_kernel, dim3(grid_for(work)), dim3(kBlock), 0, p->ctx->stream, p->ns, p->nd, p->empty_sp_row,
                     row_offset, iJacS, jJacS, MJacS, JacD);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_mdsex1_eval_Hess_Lagr(hiopamd_mdsex1* p, const double* x, double obj_factor, const double* lambda, int* iHSS,
                                  int* jHSS, double* MHSS, double* HDD)
{
  (void)x;
  (void)lambda;   // (:421-422) the constraints are linear and do not contribute
  if(!p) return HIOPAMD_ERR_ARG;
  const int64_t work = 2 * (int64_t)p->ns + (HDD ? (int64_t)p->nd * p->nd : 0);
  if(work == 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(mdsex1_hess_kernel, dim3(grid_for(work)), dim3(kBlock), 0, p->ctx->stream, p->ns, p->nd, obj_factor, p->Q,
                     iHSS, jHSS, MHSS, HDD);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

}  // extern "C"
