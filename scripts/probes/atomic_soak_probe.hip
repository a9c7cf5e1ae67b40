// Probe: do contended read-modify-writes on flag words behave when they are mixed with sc1 polls / stores from 480 + 16 persistent
// workgroups on two CU-masked streams (the access pattern of the dataflow LDL^T's flags)?  Every round: memset, two kernels side by side,
// every workgroup's lane 0 takes T tickets from ONE counter (returning add), counts each ticket in an uncontended word, adds to one of 8
// shared counters (non-returning add), stores a "version" word and polls neighbours.  Host: tickets unique? sums exact?
// Build: hipcc --offload-arch=gfx950 -O2 atomic_soak_probe.hip -o atomic_soak_probe ; run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("ERR %s at %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while(0)

__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flags layout (words): [0] ticket, [32..39] shared counters (one line), [64..127] version words, [128..) seen[total]
__global__ void worker(unsigned* flags, int T, int total, int heavy)
{
  __shared__ double pad[8192];   // 64 KB: two per CU at most, like the wide kernel
  if(threadIdx.x == 999) pad[0] = 1.0;
  if(threadIdx.x != 0) return;
  unsigned acc = 0;
  for(int it = 0; it < T; ++it) {
    const unsigned i = __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(i < (unsigned)total) (void)__hip_atomic_fetch_add(flags + 128 + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_add(flags + 32 + (i & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st(flags + 64 + (i & 63u), i);
    for(int q = 0; q < heavy; ++q) {
      acc += ld(flags + 32 + ((i + q) & 7u)) + ld(flags + 64 + ((i + 3 * q) & 63u)) + ld(flags);
      __builtin_amdgcn_s_sleep(4);
    }
  }
  if(acc == 0xdeadbeefu) flags[127] = acc;
}

int main(int argc, char** argv)
{
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000, T = argc > 2 ? atoi(argv[2]) : 64, heavy = argc > 3 ? atoi(argv[3]) : 4;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  std::vector<uint32_t> m16(words, 0), m240(words, 0);
  for(int b = 0; b < ncu; ++b) {   // bit i = XCD i % 8, CU i / 8: CUs 0-1 of each XCD for the small stream
    if(b / 8 < 2) m16[b / 32] |= 1u << (b % 32);
    else m240[b / 32] |= 1u << (b % 32);
  }
  hipStream_t s16, s240, st0;
  CK(hipExtStreamCreateWithCUMask(&s16, words, m16.data()));
  CK(hipExtStreamCreateWithCUMask(&s240, words, m240.data()));
  CK(hipStreamCreate(&st0));
  const int nwg = 480 + 16, total = nwg * T;
  unsigned* flags;
  CK(hipMalloc(&flags, sizeof(unsigned) * (128 + (size_t)total)));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  std::vector<unsigned> h(128 + (size_t)total);
  long bad_rounds = 0;
  for(int r = 0; r < rounds; ++r) {
    CK(hipMemsetAsync(flags, 0, sizeof(unsigned) * (128 + (size_t)total), st0));
    CK(hipEventRecord(e0, st0));
    CK(hipStreamWaitEvent(s16, e0, 0)); CK(hipStreamWaitEvent(s240, e0, 0));
    hipLaunchKernelGGL(worker, dim3(16), dim3(256), 0, s16, flags, T, total, heavy);
    hipLaunchKernelGGL(worker, dim3(480), dim3(256), 0, s240, flags, T, total, heavy);
    CK(hipEventRecord(e1, s16)); CK(hipEventRecord(e2, s240));
    CK(hipStreamWaitEvent(st0, e1, 0)); CK(hipStreamWaitEvent(st0, e2, 0));
    CK(hipMemcpyAsync(h.data(), flags, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost, st0));
    CK(hipStreamSynchronize(st0));
    unsigned long sum8 = 0;
    for(int q = 0; q < 8; ++q) sum8 += h[32 + q];
    int dup = 0, miss = 0;
    for(int i = 0; i < total; ++i) { if(h[128 + i] == 0) ++miss; else if(h[128 + i] > 1) ++dup; }
    if(h[0] != (unsigned)total || sum8 != (unsigned long)total || dup || miss) {
      if(bad_rounds++ < 10) printf("round %d: ticket counter %u (expected %d), shared counters sum %lu, tickets duplicated %d, never issued %d\n", r, h[0], total, sum8, dup, miss);
    }
  }
  printf("%d rounds x %d read-modify-writes per counter: %ld rounds with an anomaly\n", rounds, total, bad_rounds);
  return 0;
}
