// Probe: where do workgroups of a CU-masked stream land (XCC id, SE, CU), and can a full-LDS workgroup on a masked
// stream run concurrently with a device-filling kernel on the complementary mask?  Run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("ERR %s at %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while(0)

__global__ void where(unsigned* out)
{
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if(threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
}

__global__ void spin(long long ticks, int* sink)
{
  const long long t0 = wall_clock64();
  while(wall_clock64() - t0 < ticks) { }
  if(sink && threadIdx.x == 999) *sink = 1;
}

__global__ void biglds(long long* stamp)
{
  extern __shared__ double sm[];
  sm[threadIdx.x] = 1.0;
  __syncthreads();
  if(threadIdx.x == 0) *stamp = wall_clock64();
}

int main(int argc, char** argv)
{
  int nbits = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d\n", p.name, p.multiProcessorCount);
  const int words = (p.multiProcessorCount + 31) / 32;
  std::vector<uint32_t> m(words, 0), mc(words, 0xffffffffu);
  for(int b = 0; b < nbits; ++b) { m[b / 32] |= 1u << (b % 32); mc[b / 32] &= ~(1u << (b % 32)); }
  hipStream_t sm_, sc_;
  hipError_t e = hipExtStreamCreateWithCUMask(&sm_, words, m.data());
  printf("create masked: %s\n", hipGetErrorString(e));
  if(e != hipSuccess) return 2;
  CK(hipExtStreamCreateWithCUMask(&sc_, words, mc.data()));
  unsigned* d; CK(hipMalloc(&d, 8 * 4096));
  CK(hipMemset(d, 0xff, 8 * 4096));
  for(int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(where, dim3(rep == 0 ? 1 : 64), dim3(64), 0, sm_, d);
    CK(hipStreamSynchronize(sm_));
    std::vector<unsigned> h(128);
    CK(hipMemcpy(h.data(), d, 128 * 4, hipMemcpyDeviceToHost));
    printf("masked stream rep %d:", rep);
    for(int i = 0; i < (rep == 0 ? 1 : 64); ++i) {
      unsigned hw = h[2 * i + 1];
      printf(" [x%u se%u cu%u]", h[2 * i] & 0xf, (hw >> 13) & 7, (hw >> 8) & 0xf);
    }
    printf("\n");
  }
  // complementary stream: which CUs?
  hipLaunchKernelGGL(where, dim3(2048), dim3(64), 0, sc_, d);
  CK(hipStreamSynchronize(sc_));
  {
    std::vector<unsigned> h(4096);
    CK(hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost));
    int cnt[8] = {0};
    for(int i = 0; i < 2048; ++i) cnt[h[2 * i] & 7]++;
    printf("complement stream WG per xcc:");
    for(int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    printf("\n");
  }
  // concurrency: fill the complement with spinning WGs (2 per CU like the update kernel), then launch a full-LDS WG
  // on the masked stream; its time stamp tells when it actually started
  long long* st; CK(hipMalloc(&st, 16)); CK(hipMemset(st, 0, 16));
  CK(hipFuncSetAttribute((const void*)biglds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, sc_));
  hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, sc_, 100000LL /*1 ms at 100 MHz*/, (int*)nullptr);
  CK(hipEventRecord(e1, sc_));
  hipLaunchKernelGGL(biglds, dim3(1), dim3(256), 160 * 1024, sm_, st);
  CK(hipEventRecord(e2, sm_));
  CK(hipDeviceSynchronize());
  float t_spin, t_big;
  CK(hipEventElapsedTime(&t_spin, e0, e1));
  CK(hipEventElapsedTime(&t_big, e0, e2));
  printf("spin kernel %.3f ms; masked full-LDS WG finished %.3f ms after spin start (small => concurrent)\n", t_spin, t_big);
  // same without masks: plain high-priority stream vs plain stream
  hipStream_t s1, s2; int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, s1));
  hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s1, 100000LL, (int*)nullptr);
  CK(hipEventRecord(e1, s1));
  hipLaunchKernelGGL(biglds, dim3(1), dim3(256), 160 * 1024, s2, st);
  CK(hipEventRecord(e2, s2));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&t_spin, e0, e1));
  CK(hipEventElapsedTime(&t_big, e0, e2));
  printf("unmasked: spin %.3f ms; high-priority full-LDS WG finished %.3f ms after spin start\n", t_spin, t_big);
  return 0;
}
