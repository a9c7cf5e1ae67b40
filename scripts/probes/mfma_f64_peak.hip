// Probe: sustained rate of v_mfma_f64_16x16x4_f64 on the whole device (register-resident, no memory traffic):
// the practical ceiling of the LDL^T trailing update, to compare with the 78.6 TFLOP/s datasheet figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(int iters, double* out, long long* cyc)
{
  double4_t acc[NACC];
#pragma unroll
  for(int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const long long t0 = clock64();
  for(int it = 0; it < iters; ++it) {
#pragma unroll
    for(int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for(int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if(threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma4_loop(int iters, double* out, long long* cyc)
{
  double acc[NACC];
#pragma unroll
  for(int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const long long t0 = clock64();
  for(int it = 0; it < iters; ++it) {
#pragma unroll
    for(int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for(int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if(threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the update kernel's inner step: 4 A x 4 B operands, B rotated by DPP, 64 accumulators
__device__ __forceinline__ int ror4i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); }
__device__ __forceinline__ double ror4(double v)
{
  union { double d; int i[2]; } u; u.d = v; u.i[0] = ror4i(u.i[0]); u.i[1] = ror4i(u.i[1]); return u.d;
}
template <int DPP>
__global__ __launch_bounds__(256, 2) void mfma4_tile_loop(int iters, double* out)
{
  double acc[4][4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) acc[i][j][s] = 0.0;
  double a[4], b[4];
#pragma unroll
  for(int i = 0; i < 4; ++i) { a[i] = 1.0 + (threadIdx.x + i) * 1e-9; b[i] = 1.0 - (threadIdx.x + i) * 1e-9; }
  for(int it = 0; it < iters; ++it) {
    double br[4][4];
#pragma unroll
    for(int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(b[j]));
      br[j][0] = b[j];
      if(DPP) { br[j][1] = ror4(br[j][0]); br[j][2] = ror4(br[j][1]); br[j][3] = ror4(br[j][2]); }
      else { br[j][1] = br[j][0]; br[j][2] = br[j][0]; br[j][3] = br[j][0]; }
    }
#pragma unroll
    for(int i = 0; i < 4; ++i)
#pragma unroll
      for(int j = 0; j < 4; ++j)
#pragma unroll
        for(int s = 0; s < 4; ++s) acc[i][j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], br[j][s], acc[i][j][s], 0, 0, 0);
  }
  double t = 0.0;
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) t += acc[i][j][s];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int DPP>
__global__ __launch_bounds__(256, 1) void mfma4_tile_loop_agpr(int iters, double* out)
{
  double acc[4][4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) acc[i][j][s] = 0.0;
  double a[4], b[4];
#pragma unroll
  for(int i = 0; i < 4; ++i) { a[i] = 1.0 + (threadIdx.x + i) * 1e-9; b[i] = 1.0 - (threadIdx.x + i) * 1e-9; }
  for(int it = 0; it < iters; ++it) {
    double br[4][4];
#pragma unroll
    for(int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(b[j]));
      br[j][0] = b[j];
      if(DPP) { br[j][1] = ror4(br[j][0]); br[j][2] = ror4(br[j][1]); br[j][3] = ror4(br[j][2]); }
      else { br[j][1] = br[j][0]; br[j][2] = br[j][0]; br[j][3] = br[j][0]; }
    }
#pragma unroll
    for(int i = 0; i < 4; ++i)
#pragma unroll
      for(int j = 0; j < 4; ++j)
#pragma unroll
        for(int s = 0; s < 4; ++s) acc[i][j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], br[j][s], acc[i][j][s], 0, 0, 0);
  }
  double t = 0.0;
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) t += acc[i][j][s];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int DPP, int AGPR>
static int run_tile(int wgs, int iters, double* d, const char* tag)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if(AGPR) hipLaunchKernelGGL(mfma4_tile_loop_agpr<DPP>, dim3(wgs), dim3(256), 0, 0, 10, d);
  else hipLaunchKernelGGL(mfma4_tile_loop<DPP>, dim3(wgs), dim3(256), 0, 0, 10, d);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  if(AGPR) hipLaunchKernelGGL(mfma4_tile_loop_agpr<DPP>, dim3(wgs), dim3(256), 0, 0, iters, d);
  else hipLaunchKernelGGL(mfma4_tile_loop<DPP>, dim3(wgs), dim3(256), 0, 0, iters, d);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 512.0 * 64 * (double)iters * 4.0 * wgs;
  printf("tile step %-22s wgs=%5d: %.3f ms  %.2f TFLOP/s\n", tag, wgs, ms, flop / (ms * 1e-3) / 1e12);
  return 0;
}

template <int NACC>
static int run4(int wgs, int iters, double* d, long long* dc, const char* tag)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma4_loop<NACC>, dim3(wgs), dim3(256), 0, 0, 10, d, dc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(mfma4_loop<NACC>, dim3(wgs), dim3(256), 0, 0, iters, d, dc);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 512.0 * NACC * (double)iters * 4.0 * wgs;   // 4 blocks of 4x4x4, 2 flop per MAC
  printf("4x4x4_4b %-20s wgs=%5d acc=%2d: %.3f ms  %.2f TFLOP/s\n", tag, wgs, NACC, ms, flop / (ms * 1e-3) / 1e12);
  return 0;
}

template <int NACC>
static int run(int wgs, int iters, double* d, long long* dc, const char* tag)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(wgs), dim3(256), 0, 0, 10, d, dc);   // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(wgs), dim3(256), 0, 0, iters, d, dc);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
  const double flop = 2048.0 * NACC * (double)iters * 4.0 * wgs;   // per MFMA 16*16*4*2, 4 waves per WG
  printf("%-28s wgs=%5d acc=%2d: %.3f ms  %.2f TFLOP/s   shader clocks per MFMA per wave (wg0): %.1f  => clock %.2f GHz\n", tag, wgs,
         NACC, ms, flop / (ms * 1e-3) / 1e12, (double)c / ((double)NACC * iters), (double)c / (ms * 1e-3) / 1e9);
  return 0;
}

int main()
{
  double* d; long long* dc;
  CK(hipMalloc(&d, 8 * 256 * 8192)); CK(hipMalloc(&dc, 8));
  for(int rep = 0; rep < 2; ++rep) {
    run<16>(256, 20000, d, dc, "1 WG/CU (1 wave/SIMD)");
    run<16>(512, 20000, d, dc, "2 WG/CU (2 waves/SIMD)");
    run<4>(512, 80000, d, dc, "2 WG/CU, 4 accumulators");
    run<16>(1024, 10000, d, dc, "4 WG/CU");
    run<16>(512, 200000, d, dc, "2 WG/CU, long (~0.5 s)");
    run<8>(512, 40000, d, dc, "2 WG/CU, 8 accumulators");
    run<2>(1024, 80000, d, dc, "4 WG/CU, 2 accumulators");
    run4<16>(256, 80000, d, dc, "1 WG/CU");
    run4<16>(512, 80000, d, dc, "2 WG/CU");
    run4<4>(512, 320000, d, dc, "2 WG/CU");
    run_tile<0, 0>(512, 20000, d, "no DPP, 2 WG/CU");
    run_tile<1, 0>(512, 20000, d, "DPP rotations, 2 WG/CU");
    run_tile<1, 0>(256, 20000, d, "DPP rotations, 1 WG/CU");
    run_tile<0, 1>(256, 20000, d, "AGPR acc, no DPP, 1 WG/CU");
    run_tile<0, 1>(512, 20000, d, "AGPR acc, no DPP, 512 WGs");
    run_tile<1, 1>(256, 20000, d, "AGPR acc, DPP, 1 WG/CU");
  }
  return 0;
}
