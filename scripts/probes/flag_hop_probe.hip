// Latency of one producer->consumer hop between workgroups of ONE kernel (the dataflow solve's critical path):
// a chain of workgroups, each waits for its predecessor, reads its 256 doubles, adds 1, publishes its own.
//   v0: thread 0 polls a flag (relaxed, agent) | barrier | acquire fence | plain loads ... plain stores | release fence | flag add
//   v1: same flag, data moved with agent-scope relaxed atomics (sc1: coherent without cache maintenance), s_waitcnt instead of fences
//   v2: no flag at all: every thread polls its own data word until it is no longer the poison value
//   v3: v1 with the flag increment issued by every writer wave (no barrier on the producer side): flag target = 4 per hop
// build: hipcc --offload-arch=gfx950 -O3 -o flag_hop_probe flag_hop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while(0)

typedef unsigned long long u64;
__device__ __forceinline__ double ld_sc1(const double* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int V>
__global__ __launch_bounds__(256) void chain(u64* ticket, u64* flag, double* data, int n, int sleep_n)
{
  __shared__ int s_t;
  const int tid = threadIdx.x;
  if(tid == 0) s_t = (int)__hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int t = s_t;
  double v = 0.0;
  if(t > 0) {
    const double* src = data + (size_t)(t - 1) * 256;
    if(V == 2) {
      for(;;) {
        v = ld_sc1(src + tid);
        if(v >= 0.0) break;   // poison = -1
        if(sleep_n) __builtin_amdgcn_s_sleep(1);
      }
    } else {
      if(tid == 0) {
        const u64 target = (V == 3) ? 4ull : 1ull;
        while(__hip_atomic_load(flag + (t - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
          if(sleep_n) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      if(V == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        v = src[tid];
      } else {
        v = ld_sc1(src + tid);
      }
    }
  }
  // a token amount of work through LDS, like the real task
  __shared__ double sh[256];
  sh[tid] = v + 1.0;
  __syncthreads();
  const double o = sh[tid ^ 1] ;
  double* dst = data + (size_t)t * 256;
  if(V == 0) {
    dst[tid] = o;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if(tid == 0) (void)__hip_atomic_fetch_add(flag + t, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if(V == 1) {
    st_sc1(dst + tid, o);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(tid == 0) (void)__hip_atomic_fetch_add(flag + t, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if(V == 2) {
    st_sc1(dst + tid, o);
  } else {
    st_sc1(dst + tid, o);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if((tid & 63) == 0) (void)__hip_atomic_fetch_add(flag + t, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int V>
static void run(const char* name, int n, int sleep_n, u64* ticket, u64* flag, double* data)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  double last = 0.0;
  for(int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(ticket, 0, 8));
    CK(hipMemset(flag, 0, 8 * (size_t)n));
    // poison: -1.0
    double* h = (double*)std::malloc(sizeof(double) * 256 * (size_t)n);
    for(size_t i = 0; i < (size_t)256 * n; ++i) h[i] = -1.0;
    CK(hipMemcpy(data, h, sizeof(double) * 256 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chain<V>, dim3(n), dim3(256), 0, 0, ticket, flag, data, n, sleep_n);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if(ms < best) best = ms;
    CK(hipMemcpy(h, data + (size_t)(n - 1) * 256, 8, hipMemcpyDeviceToHost));
    last = h[0];
    std::free(h);
  }
  std::printf("%-34s sleep=%d  chain %d: %.3f ms  = %.2f us per hop   (last value %.0f, expect %d)\n", name, sleep_n, n, best,
              best * 1e3 / n, last, n);
}

int main()
{
  const int n = 2000;
  u64 *ticket, *flag;
  double* data;
  CK(hipMalloc((void**)&ticket, 8));
  CK(hipMalloc((void**)&flag, 8 * (size_t)n));
  CK(hipMalloc((void**)&data, sizeof(double) * 256 * (size_t)n));
  for(int sl = 1; sl >= 0; --sl) {
    run<0>("v0 flag + fences + plain data", n, sl, ticket, flag, data);
    run<1>("v1 flag + sc1 data + waitcnt", n, sl, ticket, flag, data);
    run<2>("v2 data is the flag (poison)", n, sl, ticket, flag, data);
    run<3>("v3 v1, per-wave flag increments", n, sl, ticket, flag, data);
  }
  return 0;
}
