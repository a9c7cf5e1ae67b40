// Probe (round 2): what limits the 128 x 128 x 256 trailing-update tile of the dataflow LDL^T at ~45 % of the fp64 MFMA peak
// although its stage loop is 64 back-to-back MFMAs + 16 ds_read_b128 + 8 buffer loads + 8 ds_write_b128 + a barrier?
// The loop is rebuilt here one ingredient at a time, whole device, 2 workgroups of 256 threads per CU like the wide kernel:
//   L0 registers only                 L1 + operands from LDS (4 ds_read_b128 per k-step, prefetched one k-step ahead)
//   L2 + one barrier per stage        L3 + 8 ds_write_b128 per stage (register data)
//   L4 + 8 x 16-byte sc1 buffer loads per stage from a 64 MB array (streaming; the values are stored by L3's writes)
// Prints TFLOP/s and the shader clock during the kernel (s_memtime ticks per 100 MHz s_memrealtime tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int KT = 16, LD = 144;

template <int LEVEL>
__global__ __launch_bounds__(256, 2) void tile_loop(int tiles, const double* __restrict__ src, size_t src_doubles, double* out,
                                                   unsigned long long* clk)
{
  __shared__ __attribute__((aligned(16))) double smem[4 * KT * LD];
  double(*Vs)[KT][LD] = reinterpret_cast<double(*)[KT][LD]>(smem);
  double(*Us)[KT][LD] = reinterpret_cast<double(*)[KT][LD]>(smem + 2 * KT * LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, lk = lane >> 4, li = lane & 15;
  for(int e = tid; e < 4 * KT * LD; e += 256) smem[e] = 1.0 + e * 1e-9;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  double4_t acc[4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int q = 0; q < 4; ++q) acc[i][q] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int arow = wr * 64 + 2 * li, bcol = wc * 64 + 2 * li, col2 = 2 * lane;
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0xffffffff, 0x00020000);
  double2_t vreg[4], ureg[4];
#pragma unroll
  for(int p = 0; p < 4; ++p) { vreg[p] = double2_t{1.0 + tid * 1e-9, 1.0}; ureg[p] = double2_t{1.0, 1.0 - tid * 1e-9}; }
  unsigned stream_off = (unsigned)(((size_t)blockIdx.x * 1048576u) % (src_doubles * 8));
  double2_t av0[2] = {double2_t{1.0 + lane * 1e-9, 1.0}, double2_t{1.0, 1.0 - lane * 1e-9}};
  double2_t bv0[2] = {double2_t{1.0, 1.0 + lane * 2e-9}, double2_t{1.0 - lane * 2e-9, 1.0}};
  for(int t = 0; t < tiles; ++t) {
    for(int st = 0; st < 16; ++st) {
      const int cur = st & 1;
      double2_t av[2][2], bv[2][2];
#pragma unroll
      for(int h = 0; h < 2; ++h) {
        if(LEVEL >= 1) {
          av[0][h] = *reinterpret_cast<const double2_t*>(&Vs[cur][lk][arow + 32 * h]);
          bv[0][h] = *reinterpret_cast<const double2_t*>(&Us[cur][lk][bcol + 32 * h]);
        } else { av[0][h] = av0[h]; bv[0][h] = bv0[h]; }
      }
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) {
        const int pb = kk & 1;
        if(kk + 1 < 4) {
#pragma unroll
          for(int h = 0; h < 2; ++h) {
            if(LEVEL >= 1) {
              av[pb ^ 1][h] = *reinterpret_cast<const double2_t*>(&Vs[cur][4 * (kk + 1) + lk][arow + 32 * h]);
              bv[pb ^ 1][h] = *reinterpret_cast<const double2_t*>(&Us[cur][4 * (kk + 1) + lk][bcol + 32 * h]);
            } else { av[pb ^ 1][h] = av0[h]; bv[pb ^ 1][h] = bv0[h]; }
          }
        }
        if(kk == 1) {
          if(LEVEL >= 3) {
#pragma unroll
            for(int p = 0; p < 4; ++p) {
              *reinterpret_cast<double2_t*>(&Vs[cur ^ 1][4 * p + wave][col2]) = vreg[p];
              *reinterpret_cast<double2_t*>(&Us[cur ^ 1][4 * p + wave][col2]) = -ureg[p];
            }
          }
          if(LEVEL >= 4) {
#pragma unroll
            for(int p = 0; p < 4; ++p) {
              vreg[p] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid, (int)(stream_off + 65536u * p), 16));
              ureg[p] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid, (int)(stream_off + 65536u * p + 4096u), 16));
            }
            stream_off += 8192u;
            if(stream_off + 4 * 65536u + 8192u >= (unsigned)(src_doubles * 8)) stream_off = 0;
          }
        }
#pragma unroll
        for(int i = 0; i < 4; ++i)
#pragma unroll
          for(int q = 0; q < 4; ++q)
            acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb][i >> 1][i & 1], bv[pb][q >> 1][q & 1], acc[i][q], 0, 0, 0);
      }
      if(LEVEL >= 2) __syncthreads();
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  double s = 0.0;
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int q = 0; q < 4; ++q) s += acc[i][q][0] + acc[i][q][1] + acc[i][q][2] + acc[i][q][3];
  out[blockIdx.x * 256 + tid] = s + vreg[0].x + ureg[3].y;
  if(tid == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int LEVEL>
static int run(int wgs, int tiles, const double* src, size_t nsrc, double* d, unsigned long long* clk, const char* tag)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((tile_loop<LEVEL>), dim3(wgs), dim3(256), 0, 0, 2, src, nsrc, d, clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((tile_loop<LEVEL>), dim3(wgs), dim3(256), 0, 0, tiles, src, nsrc, d, clk);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  const double flop = 2.0 * 128 * 128 * 256 * (double)tiles * wgs;
  printf("L%d %-34s wgs=%4d: %8.3f ms  %6.2f TFLOP/s  %5.1f us/tile-slot  shader clock %.0f MHz\n", LEVEL, tag, wgs, ms,
         flop / (ms * 1e-3) / 1e12, ms * 1e3 / tiles, (double)h[0] / (double)h[1] * 100.0);
  return 0;
}

int main()
{
  double *d, *src; unsigned long long* clk;
  const size_t nsrc = (size_t)64 << 17;   // 64 MB
  CK(hipMalloc(&d, 8 * 256 * 1024)); CK(hipMalloc(&src, nsrc * 8)); CK(hipMalloc(&clk, 16));
  CK(hipMemset(src, 0, nsrc * 8));
  for(int wgs : {512, 480, 256}) {
    run<0>(wgs, 200, src, nsrc, d, clk, "registers only");
    run<1>(wgs, 200, src, nsrc, d, clk, "+ LDS operand reads");
    run<2>(wgs, 200, src, nsrc, d, clk, "+ barrier per stage");
    run<3>(wgs, 200, src, nsrc, d, clk, "+ LDS staging writes");
    run<4>(wgs, 200, src, nsrc, d, clk, "+ sc1 buffer loads (32 KB/stage)");
  }
  return 0;
}
