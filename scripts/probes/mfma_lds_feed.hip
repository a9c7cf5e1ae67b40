// Probe: how fast can one workgroup per CU issue the 4x4x4 fp64 MFMA tile step when its operands come from LDS
// (the Gram / trailing-update inner loop without any global traffic)?  Variants add, one at a time: LDS operand reads,
// software-pipelined reads, per-stage LDS stores + barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
constexpr int LDSW = 34;

template <int MODE, int WGPCU>
__global__ __launch_bounds__(256, WGPCU) void feed(int stages, double* out)
{
  __shared__ __attribute__((aligned(16))) double As[2][128][LDSW];
  __shared__ __attribute__((aligned(16))) double Bs[2][128][LDSW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lk = lane >> 4, li = lane & 15;
  for(int e = tid; e < 2 * 128 * LDSW; e += 256) { (&As[0][0][0])[e] = 1.0 + e * 1e-9; (&Bs[0][0][0])[e] = 1.0 - e * 1e-9; }
  __syncthreads();
  double acc[4][4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) acc[i][j][s] = 0.0;
  double st[32];
#pragma unroll
  for(int q = 0; q < 32; ++q) st[q] = tid * 1e-3 + q;
  int buf = 0;
  for(int sg = 0; sg < stages; ++sg) {
    double a[2][4], b[2][4][4];
    auto lread = [&](int kk, double (&aa)[4], double (&bb)[4][4]) {
#pragma unroll
      for(int i = 0; i < 4; ++i) aa[i] = As[buf][wr * 64 + i * 16 + li][kk * 4 + lk];
#pragma unroll
      for(int j = 0; j < 4; ++j)
#pragma unroll
        for(int s = 0; s < 4; ++s) bb[j][s] = Bs[buf][wc * 64 + j * 16 + ((li + 4 * s) & 15)][kk * 4 + lk];
    };
    if(MODE >= 1) lread(0, a[0], b[0]);
    else {
#pragma unroll
      for(int i = 0; i < 4; ++i) a[0][i] = st[i];
#pragma unroll
      for(int j = 0; j < 4; ++j)
#pragma unroll
        for(int s = 0; s < 4; ++s) b[0][j][s] = st[4 + 4 * j + s];
    }
#pragma unroll
    for(int kk = 0; kk < 8; ++kk) {
      const int cur = (MODE >= 2) ? (kk & 1) : 0;
      if(MODE >= 2 && kk + 1 < 8) lread(kk + 1, a[(kk + 1) & 1], b[(kk + 1) & 1]);
      if(MODE == 1 && kk > 0) lread(kk, a[0], b[0]);
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int j = 0; j < 4; ++j)
#pragma unroll
          for(int s = 0; s < 4; ++s) acc[i][j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[cur][i], b[cur][j][s], acc[i][j][s], 0, 0, 0);
    }
    if(MODE >= 3 && MODE < 7) {   // per-stage LDS stores into the other buffer + barrier
      const int kk2 = (tid & 15) * 2;
#pragma unroll
      for(int p = 0; p < 8; ++p) {
        const int r = p * 16 + (tid >> 4);
        if(MODE == 3) {
          As[buf ^ 1][r][kk2] = st[p];
          As[buf ^ 1][r][kk2 + 1] = st[8 + p];
          Bs[buf ^ 1][r][kk2] = st[16 + p];
          Bs[buf ^ 1][r][kk2 + 1] = st[24 + p];
        } else {   // one 16-byte store per row piece
          *reinterpret_cast<double2*>(&As[buf ^ 1][r][kk2]) = double2{st[p], st[8 + p]};
          *reinterpret_cast<double2*>(&Bs[buf ^ 1][r][kk2]) = double2{st[16 + p], st[24 + p]};
        }
      }
      if(MODE != 5) __syncthreads();
      if(MODE != 6) buf ^= 1;
    }
    if(MODE == 7) __syncthreads();
    if(MODE == 8) __builtin_amdgcn_s_barrier();
    if(MODE == 9 && (sg & 3) == 3) __syncthreads();
  }
  double t = 0.0;
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) t += acc[i][j][s];
  out[blockIdx.x * 256 + tid] = t;
}

// wave-private LDS: every wave stages its OWN 64 A-rows and 64 B-rows (single buffer) -> no workgroup barrier at all
__global__ __launch_bounds__(256, 1) void feed_private(int stages, double* out)
{
  __shared__ __attribute__((aligned(16))) double Ws[4][128][LDSW];   // per wave: rows 0..63 = A, 64..127 = B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, li = lane & 15;
  double(*W)[LDSW] = Ws[wave];
  for(int e = lane; e < 128 * LDSW; e += 64) (&W[0][0])[e] = 1.0 + e * 1e-9;
  double acc[4][4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) acc[i][j][s] = 0.0;
  double st[64];
#pragma unroll
  for(int q = 0; q < 64; ++q) st[q] = tid * 1e-3 + q;
  for(int sg = 0; sg < stages; ++sg) {
    double a[2][4], b[2][4][4];
    auto lread = [&](int kk, double (&aa)[4], double (&bb)[4][4]) {
#pragma unroll
      for(int i = 0; i < 4; ++i) aa[i] = W[i * 16 + li][kk * 4 + lk];
#pragma unroll
      for(int j = 0; j < 4; ++j)
#pragma unroll
        for(int s = 0; s < 4; ++s) bb[j][s] = W[64 + j * 16 + ((li + 4 * s) & 15)][kk * 4 + lk];
    };
    lread(0, a[0], b[0]);
#pragma unroll
    for(int kk = 0; kk < 8; ++kk) {
      if(kk + 1 < 8) lread(kk + 1, a[(kk + 1) & 1], b[(kk + 1) & 1]);
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int j = 0; j < 4; ++j)
#pragma unroll
          for(int s = 0; s < 4; ++s)
            acc[i][j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[kk & 1][i], b[kk & 1][j][s], acc[i][j][s], 0, 0, 0);
    }
    // the wave refills its own region: 128 rows x 32 k = 4096 doubles = 64 per lane (2 k's per lane, 32 rows)
    const int kk2 = (lane & 15) * 2;
#pragma unroll
    for(int p = 0; p < 32; ++p) {
      const int r = p * 4 + (lane >> 4);
      *reinterpret_cast<double2*>(&W[r][kk2]) = double2{st[2 * p], st[2 * p + 1]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  double t = 0.0;
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
#pragma unroll
      for(int s = 0; s < 4; ++s) t += acc[i][j][s];
  out[blockIdx.x * 256 + tid] = t;
}

template <int MODE, int WGPCU>
static int run(int wgs, int stages, double* d, const char* tag)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((feed<MODE, WGPCU>), dim3(wgs), dim3(256), 0, 0, 10, d);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((feed<MODE, WGPCU>), dim3(wgs), dim3(256), 0, 0, stages, d);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 512.0 * 64 * 8 * (double)stages * 4.0 * wgs;
  printf("%-52s wgs=%4d: %8.3f ms  %6.2f TFLOP/s  (%.2f us per 32-deep stage)\n", tag, wgs, ms, flop / (ms * 1e-3) / 1e12,
         ms * 1e3 / stages);
  return 0;
}

int main()
{
  double* d; CK(hipMalloc(&d, 8 * 256 * 1024));
  run<0, 1>(256, 4000, d, "registers only, 1 WG/CU");
  run<1, 1>(256, 4000, d, "LDS reads before each k-step, 1 WG/CU");
  run<2, 1>(256, 4000, d, "LDS reads pipelined one k-step ahead, 1 WG/CU");
  run<3, 1>(256, 4000, d, "+ per-stage LDS stores and barrier, 1 WG/CU");
  run<4, 1>(256, 4000, d, "+ stores as ds_write_b128 and barrier, 1 WG/CU");
  run<5, 1>(256, 4000, d, "stores, NO barrier (racy), 1 WG/CU");
  run<6, 1>(256, 4000, d, "stores + barrier, same buffer re-read, 1 WG/CU");
  run<7, 1>(256, 4000, d, "barrier only, no stores, 1 WG/CU");
  run<8, 1>(256, 4000, d, "raw s_barrier only (no fences), 1 WG/CU");
  run<9, 1>(256, 4000, d, "__syncthreads every 4th stage, 1 WG/CU");
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(feed_private, dim3(256), dim3(256), 0, 0, 10, d);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(feed_private, dim3(256), dim3(256), 0, 0, 4000, d);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s wgs= 256: %8.3f ms  %6.2f TFLOP/s  (%.2f us per 32-deep stage)\n", "wave-private LDS refill, no barrier, 1 WG/CU", ms,
           512.0 * 64 * 8 * 4000.0 * 4.0 * 256 / (ms * 1e-3) / 1e12, ms * 1e3 / 4000);
  }
  run<1, 2>(512, 4000, d, "LDS reads before each k-step, 2 WG/CU");
  run<3, 2>(512, 4000, d, "pipelined + stores + barrier, 2 WG/CU");
  return 0;
}
