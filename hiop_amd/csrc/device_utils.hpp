// Generic element-wise and reduction launch machinery shared by the LinAlg kernels (gfx950).
#pragma once
#include "common.hpp"

#include <cfloat>
#include <cmath>

namespace hiopamd {

// ------------------------------------------------------------------------------------------
// element-wise: y[i] = f(i)   — functor-driven, 2 elements per thread per trip
// ------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(kBlock) void ew_kernel(int64_t n, F f)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  // unroll by 4 with independent loads in flight
  for(; i + 3 * stride < n; i += 4 * stride) {
    f(i);
    f(i + stride);
    f(i + 2 * stride);
    f(i + 3 * stride);
  }
  for(; i < n; i += stride) f(i);
}

template <class F>
static inline int launch_ew(hiopamd_ctx* ctx, int64_t n, F f)
{
  if(n < 0) return HIOPAMD_ERR_ARG;
  if(n == 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(ew_kernel<F>, dim3(grid_for(n, 4)), dim3(kBlock), 0, ctx->stream, n, f);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

// ------------------------------------------------------------------------------------------
// reductions.  Op concept: T identity(); T map(i); T combine(T,T)
// two launches: (1) per-block partials (LDS-staged tree, fixed order); (2) one block folds the
// partials in index order and writes the pinned host slot.
// ------------------------------------------------------------------------------------------
struct kahan_t;
__device__ inline double shfl_down_t(double v, int off) { return __shfl_down(v, off, 64); }
template <class T>
__device__ inline T shfl_down_t(T v, int off)
{
  // generic: shuffle as 64-bit words
  static_assert(sizeof(T) % sizeof(double) == 0, "");
  union {
    T t;
    double d[sizeof(T) / sizeof(double)];
  } u;
  u.t = v;
#pragma unroll
  for(unsigned q = 0; q < sizeof(T) / sizeof(double); ++q) u.d[q] = __shfl_down(u.d[q], off, 64);
  return u.t;
}

template <class T, class Op>
__device__ inline T block_reduce(T v, Op op)
{
  // wave64 shuffle tree, then LDS across the 4 waves
  for(int off = 32; off > 0; off >>= 1) {
    T o = shfl_down_t(v, off);
    v = op.combine(v, o);
  }
  __shared__ T smem[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if(lane == 0) smem[wave] = v;
  __syncthreads();
  if(threadIdx.x == 0) {
    T r = smem[0];
    for(int w = 1; w < kBlock / 64; ++w) r = op.combine(r, smem[w]);
    v = r;
  }
  return v;  // valid in thread 0
}

template <class T, class Op>
__global__ __launch_bounds__(kBlock) void reduce_stage1(int64_t n, Op op, T* partials)
{
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  T acc = op.identity();
  for(int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    acc = op.combine(acc, op.map(i));
  }
  acc = block_reduce<T, Op>(acc, op);
  if(threadIdx.x == 0) partials[blockIdx.x] = acc;
}

template <class T, class Op>
__global__ __launch_bounds__(kBlock) void reduce_stage2(int nparts, Op op, const T* partials, T* out_dev, T* out_host)
{
  T acc = op.identity();
  // fixed association: thread t folds partials t, t+256, ...
  for(int i = threadIdx.x; i < nparts; i += kBlock) acc = op.combine(acc, partials[i]);
  acc = block_reduce<T, Op>(acc, op);
  if(threadIdx.x == 0) {
    if(out_dev) *out_dev = acc;
    if(out_host) *out_host = acc;
  }
}

// Kahan pair for the log-barrier sum (reference keeps a compensated sum: hiopVectorPar.cpp:863-881)
struct kahan_t {
  double s, c;
};

// fin(const T&) runs on the host when the value has arrived: right away (one stream synchronisation per reduction: the behaviour of
// the reference's bool / double returning methods), or — inside hiopamd_ctx_reduce_begin / _end — when the bracket is closed.
// may_defer = false: the caller needs the value now (host logic follows): the result takes the next free slot, the stream is
// synchronised, pending results of the bracket stay pending.
template <class T, class Op, class Fin>
static inline int launch_reduce_fin(hiopamd_ctx* ctx, int64_t n, Op op, Fin fin, bool may_defer = true)
{
  static_assert(sizeof(T) <= 4 * sizeof(double), "partials slot too small");
  constexpr int kSlots = kHostSlots / 4;
  if(ctx->n_pending >= kSlots) {
    const int rf = reduce_flush(ctx);
    if(rf != HIOPAMD_OK) return rf;
  }
  const int slot = ctx->n_pending;
  T* partials = reinterpret_cast<T*>(ctx->d_partials);
  T* out_host_dev = reinterpret_cast<T*>(ctx->h_result_dev + 4 * slot);
  T* out_host = reinterpret_cast<T*>(ctx->h_result + 4 * slot);
  int g = n > 0 ? grid_for(n, 8) : 1;
  hipLaunchKernelGGL((reduce_stage1<T, Op>), dim3(g), dim3(kBlock), 0, ctx->stream, n, op, partials);
  hipLaunchKernelGGL((reduce_stage2<T, Op>), dim3(1), dim3(kBlock), 0, ctx->stream, g, op, partials,
                     reinterpret_cast<T*>(ctx->d_result), out_host_dev);
  HIOPAMD_CHECK(hipGetLastError());
  if(may_defer && ctx->defer_depth > 0) {
    ctx->pending.emplace_back([out_host, fin]() { fin(*out_host); });
    ctx->n_pending += 1;
    HIOPAMD_CHECK(hipEventRecord(ctx_named_event(ctx->ev_pending), ctx->stream));
    return HIOPAMD_OK;
  }
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  fin(*out_host);
  return HIOPAMD_OK;
}
template <class T, class Op>
static inline int launch_reduce(hiopamd_ctx* ctx, int64_t n, Op op, T* result_host)
{
  return launch_reduce_fin<T>(ctx, n, op, [result_host](const T& v) { *result_host = v; }, false);
}
// ... for the public scalar-returning entry points: the caller's result variable is written when the bracket is closed
template <class T, class Op>
static inline int launch_reduce_deferrable(hiopamd_ctx* ctx, int64_t n, Op op, T* result_host)
{
  return launch_reduce_fin<T>(ctx, n, op, [result_host](const T& v) { *result_host = v; }, true);
}

}  // namespace hiopamd
