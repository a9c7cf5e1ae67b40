// Can a HIP stream wait for a DEVICE flag that a running kernel raises (hipStreamWaitValue32 on hipMalloc'd memory written with
// agent-scope atomics), and how late does the dependent kernel start?  (Round 5: the LDL^T's epilogue — diagonal-block inverses for the
// solve — per group of super-panels behind the chain kernel's `cdone` words, without a resident polling kernel.)
//   producer: one workgroup, raises flag to 1..NSTEP, one step every `gap_us`, stamping the wall clock of every raise
//   consumer streams: for v in {2, 5, 9}: hipStreamWaitValue32(flag >= v) then a one-thread kernel that stamps the wall clock
//   report: consumer stamp - producer stamp of step v (us); "NOT BEFORE END" when the consumer only ran after the producer had finished
// build: hipcc --offload-arch=gfx950 -O3 -o stream_wait_value_probe stream_wait_value_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::printf("HIP error %s (%d) line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); std::exit(1); } } while(0)
typedef unsigned long long u64;

__global__ void producer(unsigned* flag, u64* stamps, int nstep, long long gap_ticks)
{
  if(threadIdx.x != 0) return;
  u64 t = wall_clock64();
  stamps[0] = t;
  for(int s = 1; s <= nstep; ++s) {
    while((long long)(wall_clock64() - t) < gap_ticks) __builtin_amdgcn_s_sleep(32);
    t = wall_clock64();
    stamps[s] = t;
    __hip_atomic_store(flag, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  stamps[nstep + 1] = wall_clock64();
}
__global__ void consumer(u64* out, const unsigned* flag)
{
  out[0] = wall_clock64();
  out[1] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv)
{
  const int nstep = 10;
  const double gap_us = argc > 1 ? std::atof(argv[1]) : 200.0;
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if(!can) return 0;
  for(int mode = 0; mode < 2; ++mode) {   // 0: hipMalloc, 1: hipExtMallocWithFlags(hipMallocSignalMemory)
    unsigned* flag = nullptr;
    if(mode == 0) CK(hipMalloc(&flag, 64));
    else if(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) != hipSuccess) {
      std::printf("signal memory: allocation failed\n");
      (void)hipGetLastError();
      continue;
    }
    CK(hipMemset(flag, 0, 8));
    u64 *stamps, *outs;
    CK(hipMalloc(&stamps, sizeof(u64) * 16));
    CK(hipMalloc(&outs, sizeof(u64) * 8));
    CK(hipMemset(outs, 0, sizeof(u64) * 8));
    hipStream_t sp, sc[3];
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    for(int q = 0; q < 3; ++q) CK(hipStreamCreateWithFlags(&sc[q], hipStreamNonBlocking));
    const unsigned want[3] = {2, 5, 9};
    for(int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(flag, 0, 8));
      CK(hipDeviceSynchronize());
      for(int q = 0; q < 3; ++q) {
        hipError_t e = hipStreamWaitValue32(sc[q], flag, want[q], hipStreamWaitValueGte, 0xffffffffu);
        if(e != hipSuccess) {
          std::printf("mode %d: hipStreamWaitValue32 -> %s\n", mode, hipGetErrorString(e));
          (void)hipGetLastError();
          goto next_mode;
        }
        hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, sc[q], outs + 2 * q, flag);
      }
      hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, sp, flag, stamps, nstep, (long long)(gap_us * 100.0));
      CK(hipDeviceSynchronize());
      u64 hs[16], ho[8];
      CK(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
      CK(hipMemcpy(ho, outs, sizeof(ho), hipMemcpyDeviceToHost));
      std::printf("mode %d (%s) rep %d, gap %.0f us:", mode, mode ? "signal memory" : "hipMalloc", rep, gap_us);
      for(int q = 0; q < 3; ++q) {
        const double late = ((double)ho[2 * q] - (double)hs[want[q]]) * 0.01;
        const bool after_end = ho[2 * q] > hs[nstep + 1];
        std::printf("  wait>=%u: +%.1f us (flag seen %llu)%s", want[q], late, ho[2 * q + 1], after_end ? " NOT BEFORE END" : "");
      }
      std::printf("\n");
    }
  next_mode:
    (void)hipFree(flag);
  }
  return 0;
}
