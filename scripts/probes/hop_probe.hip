// Probe: latency of a cross-stream dependency (hipEventRecord on A, hipStreamWaitEvent on B, tiny kernel on B) for
// plain, high-priority and CU-masked streams; and of same-stream back-to-back tiny kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("ERR %s at %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while(0)
__global__ void tiny(int* p) { if(threadIdx.x == 0) atomicAdd(p, 1); }

static double pingpong(hipStream_t a, hipStream_t b, int hops, int* d, std::vector<hipEvent_t>& ev)
{
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  hipStream_t s[2] = {a, b};
  for(int h = 0; h < hops; ++h) {
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[h & 1], d);
    hipEventRecord(ev[h % ev.size()], s[h & 1]);
    hipStreamWaitEvent(s[(h + 1) & 1], ev[h % ev.size()], 0);
  }
  hipDeviceSynchronize();
  auto t1 = std::chrono::high_resolution_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / hops;
}

int main()
{
  int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
  std::vector<hipEvent_t> ev(64);
  for(auto& evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
  std::vector<hipEvent_t> evt(64);
  for(auto& evx : evt) CK(hipEventCreate(&evx));
  hipStream_t p1, p2, h1, m1, m2;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&p1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&p2, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&h1, hipStreamNonBlocking, hi));
  uint32_t m[8] = {0xffu, 0, 0, 0, 0, 0, 0, 0};
  uint32_t mc[8] = {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  CK(hipExtStreamCreateWithCUMask(&m1, 8, m));
  CK(hipExtStreamCreateWithCUMask(&m2, 8, mc));
  for(int rep = 0; rep < 2; ++rep) {
    printf("same stream back-to-back          : %.2f us/kernel\n", pingpong(p1, p1, 2000, d, ev));
    printf("plain <-> plain                   : %.2f us/hop\n", pingpong(p1, p2, 2000, d, ev));
    printf("plain <-> plain (timing events)   : %.2f us/hop\n", pingpong(p1, p2, 2000, d, evt));
    printf("plain <-> high priority           : %.2f us/hop\n", pingpong(p1, h1, 2000, d, ev));
    printf("masked(8) <-> masked(248)         : %.2f us/hop\n", pingpong(m1, m2, 2000, d, ev));
    printf("plain <-> masked(8)               : %.2f us/hop\n", pingpong(p1, m1, 2000, d, ev));
    printf("masked(248) <-> high priority     : %.2f us/hop\n", pingpong(m2, h1, 2000, d, ev));
  }
  return 0;
}
