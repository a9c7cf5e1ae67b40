// Device-resident user callbacks of the reference's dense-constraints example DenseConsEx2 —
// the problem of the memory-distributed quasi-Newton path (A): variables split by columns across ranks, four constraints.
//
// reference: src/Drivers/Dense/NlpDenseConsEx2.{hpp,cpp} — the callbacks of hiopInterfaceDenseConstraints
// (src/Interface/hiopInterface.hpp:420-560) on the LOCAL part of x, every array a DEVICE pointer; the two global reductions
// of the example (MPI_Allreduce in eval_f and eval_cons, .cpp:112-116, :211-218) go through the context's all-reduce hook
// (RCCL when hiopamd_ctx_init_rccl was called).
//
//   min sum_i 1/4 (x_i - 1)^4   s.t.  sum x_i = n + 1;   5 <= 2 x_1 + sum_{i>=2} x_i;
//       1 <= 2 x_1 + 0.5 x_2 + sum_{i>=3} x_i <= 2n;   4 x_1 + 2 x_2 + 2 x_3 + sum_{i>=4} x_i <= 4n;
//       x_1 free, x_2 >= 0, 1.5 <= x_3 <= 10, x_i >= 0.5 (i >= 4);   x0 = 0
#include "common.hpp"

namespace hiopamd {

// get_vars_info (.cpp:52-80): global index decides the bounds
__global__ __launch_bounds__(kBlock) void denseex2_vars_kernel(int64_t nloc, int64_t c0, double* __restrict__ xlow,
                                                              double* __restrict__ xupp)
{
  for(int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nloc; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t gi = c0 + i;
    double lo = 0.5, up = 1e20;
    if(gi == 0) lo = -1e20;
    else if(gi == 1) lo = 0.0;
    else if(gi == 2) {
      lo = 1.5;
      up = 10.0;
    }
    if(xlow) xlow[i] = lo;
    if(xupp) xupp[i] = up;
  }
}

// one workgroup, fixed summation order.  out[0] = sum 1/4 (x_i - 1)^4 (eval_f, .cpp:104-111); out[1..4] = the local parts of
// the four constraint bodies (eval_cons, .cpp:146-208): the plain sum plus the extra weights of the global entries 0, 1, 2
__global__ __launch_bounds__(1024) void denseex2_sums_kernel(int64_t nloc, int64_t c0, const double* __restrict__ x,
                                                            double* __restrict__ out)
{
  __shared__ double red[2][1024];
  const int tid = threadIdx.x;
  double f = 0.0, s = 0.0;
  for(int64_t i = tid; i < nloc; i += 1024) {
    const double d = x[i] - 1.0, d2 = d * d;
    f += 0.25 * (d2 * d2);
    s += x[i];
  }
  red[0][tid] = f;
  red[1][tid] = s;
  __syncthreads();
  for(int w = 512; w >= 1; w >>= 1) {
    if(tid < w) {
      red[0][tid] += red[0][tid + w];
      red[1][tid] += red[1][tid + w];
    }
    __syncthreads();
  }
  if(tid == 0) {
    const double S = red[1][0];
    // entries with global index 0, 1, 2 when this rank owns them
    const double g0 = (c0 <= 0 && 0 < c0 + nloc) ? x[0 - c0] : 0.0;
    const double g1 = (c0 <= 1 && 1 < c0 + nloc) ? x[1 - c0] : 0.0;
    const double g2 = (c0 <= 2 && 2 < c0 + nloc) ? x[2 - c0] : 0.0;
    out[0] = red[0][0];
    out[1] = S;
    out[2] = S + g0;                    // 2 x_1 + sum_{i>=2}
    out[3] = S + g0 - 0.5 * g1;         // 2 x_1 + 0.5 x_2 + sum_{i>=3}
    out[4] = S + 3.0 * g0 + g1 + g2;    // 4 x_1 + 2 x_2 + 2 x_3 + sum_{i>=4}
  }
}

// eval_grad_f (.cpp:119-126)
__global__ __launch_bounds__(kBlock) void denseex2_grad_kernel(int64_t nloc, const double* __restrict__ x, double* __restrict__ g)
{
  for(int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nloc; i += (int64_t)gridDim.x * blockDim.x) {
    const double d = x[i] - 1.0;
    g[i] = d * d * d;
  }
}

// eval_Jac_cons (.cpp:224-290): 4 x n_local, row-major; ones except the columns of the global entries 0, 1, 2
__global__ __launch_bounds__(kBlock) void denseex2_jac_kernel(int64_t nloc, int64_t c0, double* __restrict__ Jac)
{
  for(int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < 4 * nloc; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / nloc);
    const int64_t gi = c0 + (e - (int64_t)r * nloc);
    double v = 1.0;
    if(r == 1 && gi == 0) v = 2.0;
    if(r == 2) v = (gi == 0) ? 2.0 : ((gi == 1) ? 0.5 : 1.0);
    if(r == 3) v = (gi == 0) ? 4.0 : ((gi == 1 || gi == 2) ? 2.0 : 1.0);
    Jac[e] = v;
  }
}

}  // namespace hiopamd

using namespace hiopamd;

struct hiopamd_denseex2 {
  hiopamd_ctx* ctx = nullptr;
  int64_t n = 0, c0 = 0, c1 = 0;   // global size, local columns [c0, c1)
  int m = 4;
  double* red = nullptr;           // 5 device scalars: objective | four constraint bodies
};

// the example's partition (.cpp:24-38): the first `remainder` ranks get one column more
static int64_t denseex2_col(int64_t n, int size, int r)
{
  const int64_t q = n / size, rem = n - (int64_t)size * q;
  return (r <= rem) ? (int64_t)r * (q + 1) : rem * (q + 1) + ((int64_t)r - rem) * q;
}

extern "C" {

int hiopamd_denseex2_create(hiopamd_denseex2** out, hiopamd_ctx* ctx, int64_t n_global, int unconstrained)
{
  if(!out || !ctx || n_global < 0) return HIOPAMD_ERR_ARG;
  auto* p = new hiopamd_denseex2;
  p->ctx = ctx;
  p->n = n_global;
  p->m = unconstrained ? 0 : 4;
  const int size = ctx->comm_size > 0 ? ctx->comm_size : 1, rank = ctx->comm_rank;
  p->c0 = denseex2_col(n_global, size, rank);
  p->c1 = denseex2_col(n_global, size, rank + 1);
  if(hipMalloc((void**)&p->red, sizeof(double) * 8) != hipSuccess) {
    delete p;
    return HIOPAMD_ERR_HIP;
  }
  *out = p;
  return HIOPAMD_OK;
}

int hiopamd_denseex2_destroy(hiopamd_denseex2* p)
{
  if(!p) return HIOPAMD_OK;
  (void)hipStreamSynchronize(p->ctx->stream);
  (void)hipFree(p->red);
  delete p;
  return HIOPAMD_OK;
}

int hiopamd_denseex2_get_prob_sizes(const hiopamd_denseex2* p, int64_t* n, int64_t* m)   /* .cpp:45-50 */
{
  if(!p || !n || !m) return HIOPAMD_ERR_ARG;
  *n = p->n;
  *m = p->m;
  return HIOPAMD_OK;
}

int hiopamd_denseex2_get_vecdistrib_info(const hiopamd_denseex2* p, int64_t* cols_host)   /* .cpp:293-304: comm_size + 1 entries */
{
  if(!p || !cols_host) return HIOPAMD_ERR_ARG;
  const int size = p->ctx->comm_size > 0 ? p->ctx->comm_size : 1;
  for(int r = 0; r <= size; ++r) cols_host[r] = denseex2_col(p->n, size, r);
  return HIOPAMD_OK;
}

int hiopamd_denseex2_get_vars_info(hiopamd_denseex2* p, double* xlow, double* xupp)
{
  if(!p) return HIOPAMD_ERR_ARG;
  const int64_t nloc = p->c1 - p->c0;
  if(nloc == 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(denseex2_vars_kernel, dim3(grid_for(nloc)), dim3(kBlock), 0, p->ctx->stream, nloc, p->c0, xlow, xupp);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_denseex2_get_cons_info(const hiopamd_denseex2* p, double* clow_host, double* cupp_host)   /* .cpp:82-102 (replicated) */
{
  if(!p || !clow_host || !cupp_host) return HIOPAMD_ERR_ARG;
  if(p->m == 0) return HIOPAMD_OK;
  const double n = (double)p->n;
  clow_host[0] = n + 1.0; cupp_host[0] = n + 1.0;
  clow_host[1] = 5.0;     cupp_host[1] = 1e20;
  clow_host[2] = 1.0;     cupp_host[2] = 2.0 * n;
  clow_host[3] = -1e20;   cupp_host[3] = 4.0 * n;
  return HIOPAMD_OK;
}

int hiopamd_denseex2_get_starting_point(hiopamd_denseex2* p, double* x0)   /* .cpp:307-315 */
{
  if(!p || !x0) return HIOPAMD_ERR_ARG;
  return hiopamd_vec_set_to_constant(p->ctx, p->c1 - p->c0, x0, 0.0);
}

static int denseex2_reduce(hiopamd_denseex2* p, const double* x)
{
  hiopamd_ctx* ctx = p->ctx;
  hipLaunchKernelGGL(denseex2_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, p->c1 - p->c0, p->c0, x, p->red);
  HIOPAMD_CHECK(hipGetLastError());
  if(ctx->allreduce && ctx_allreduce(ctx, p->red, (size_t)5, HIOPAMD_SUM) != 0)
    return HIOPAMD_ERR_HIP;
  return HIOPAMD_OK;
}

int hiopamd_denseex2_eval_f(hiopamd_denseex2* p, const double* x, double* obj_host)
{
  if(!p || !x || !obj_host) return HIOPAMD_ERR_ARG;
  int rc = denseex2_reduce(p, x);
  if(rc != HIOPAMD_OK) return rc;
  rc = hiopamd_copy_d2h(p->ctx, obj_host, p->red, sizeof(double));
  if(rc != HIOPAMD_OK) return rc;
  return hiopamd_ctx_sync(p->ctx);
}

int hiopamd_denseex2_eval_grad_f(hiopamd_denseex2* p, const double* x, double* gradf)
{
  if(!p || !x || !gradf) return HIOPAMD_ERR_ARG;
  const int64_t nloc = p->c1 - p->c0;
  if(nloc == 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(denseex2_grad_kernel, dim3(grid_for(nloc)), dim3(kBlock), 0, p->ctx->stream, nloc, x, gradf);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_denseex2_eval_cons(hiopamd_denseex2* p, const double* x, double* cons_dev)
{
  if(!p || !x) return HIOPAMD_ERR_ARG;
  if(p->m == 0) return HIOPAMD_OK;
  if(!cons_dev) return HIOPAMD_ERR_ARG;
  const int rc = denseex2_reduce(p, x);
  if(rc != HIOPAMD_OK) return rc;
  HIOPAMD_CHECK(hipMemcpyAsync(cons_dev, p->red + 1, 4 * sizeof(double), hipMemcpyDeviceToDevice, p->ctx->stream));
  return HIOPAMD_OK;
}

int hiopamd_denseex2_eval_Jac_cons(hiopamd_denseex2* p, const double* x, double* Jac_dev)
{
  (void)x;   // linear constraints
  if(!p) return HIOPAMD_ERR_ARG;
  const int64_t nloc = p->c1 - p->c0;
  if(p->m == 0 || nloc == 0) return HIOPAMD_OK;
  if(!Jac_dev) return HIOPAMD_ERR_ARG;
  hipLaunchKernelGGL(denseex2_jac_kernel, dim3(grid_for(4 * nloc)), dim3(kBlock), 0, p->ctx->stream, nloc, p->c0, Jac_dev);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

}  // extern "C"
