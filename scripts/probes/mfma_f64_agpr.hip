// Probe (round 2): does v_mfma_f64_16x16x4_f64 run faster with its accumulators in AGPRs (what the vendor's Tensile DGEMM
// does: 76.5 TFLOP/s) than in arch VGPRs (what hipcc picks for the builtin: round 1 measured 36-46 TFLOP/s)?
// Register-only loops, whole device.  Variants: builtin (compiler's choice), inline asm with "+v" accumulators, inline asm with
// "+a" accumulators; 16 / 8 / 4 independent accumulators per wave; 1 or 2 workgroups of 256 threads per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>   // MODE 0 builtin, 1 asm "+v", 2 asm "+a"
__global__ __launch_bounds__(256, 2) void loop16(int iters, double* out)
{
  double4_t acc[NACC];
#pragma unroll
  for(int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
#pragma unroll
  for(int i = 0; i < 4; ++i) { a[i] = 1.0 + (threadIdx.x + i) * 1e-9; b[i] = 1.0 - (threadIdx.x + 3 * i) * 1e-9; }
  for(int it = 0; it < iters; ++it) {
#pragma unroll
    for(int i = 0; i < NACC; ++i) {
      const double av = a[(i >> 2) & 3], bv = b[i & 3];
      if(MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
      else if(MODE == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
      else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(av), "v"(bv));
    }
  }
  double s = 0.0;
#pragma unroll
  for(int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE>
static int run(int wgs, int iters, double* d, const char* tag)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((loop16<NACC, MODE>), dim3(wgs), dim3(256), 0, 0, 10, d);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((loop16<NACC, MODE>), dim3(wgs), dim3(256), 0, 0, iters, d);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2048.0 * NACC * (double)iters * 4.0 * wgs;
  printf("%-14s acc=%2d wgs=%4d: %8.3f ms  %6.2f TFLOP/s\n", tag, NACC, wgs, ms, flop / (ms * 1e-3) / 1e12);
  return 0;
}

int main()
{
  double* d;
  CK(hipMalloc(&d, 8 * 256 * 4096));
  for(int rep = 0; rep < 2; ++rep) {
    run<16, 0>(256, 20000, d, "builtin");
    run<16, 0>(512, 20000, d, "builtin");
    run<16, 1>(512, 20000, d, "asm +v");
    run<16, 2>(256, 20000, d, "asm +a (AGPR)");
    run<16, 2>(512, 20000, d, "asm +a (AGPR)");
    run<8, 2>(512, 40000, d, "asm +a (AGPR)");
    run<4, 2>(512, 80000, d, "asm +a (AGPR)");
    run<4, 0>(512, 80000, d, "builtin");
    run<4, 1>(512, 80000, d, "asm +v");
  }
  // sustained: ~0.5 s and ~2 s of back-to-back MFMAs on every CU (does the part hold its clock under fp64 matrix load?)
  run<16, 0>(512, 600000, d, "builtin 0.5s");
  run<16, 0>(512, 2400000, d, "builtin 2s");
  run<16, 0>(480, 600000, d, "builtin 480wg");
  return 0;
}
