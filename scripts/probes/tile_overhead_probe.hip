// Probe (round 2): would MORE, SMALLER workgroups per CU hide the per-task latency phases of the dataflow update better?
// The stage loop of tile_loop_probe.hip (LDS feed + barrier + staging + sc1 loads) for a TM x 128 x 256 tile (TM = 128: 16
// accumulators per wave, 2 workgroups per CU; TM = 64: 8 accumulators, 3-4 per CU), followed after every tile by an emulated
// task overhead: a chain of DEP dependent sc1 loads (~2 us each under load: ticket, flags, prologue, drain, publish).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int KT = 16, LD = 144;

template <int TM, int WGPC>
__global__ __launch_bounds__(256, WGPC) void tile_loop(int tiles, int dep, const double* __restrict__ src, size_t src_doubles,
                                                      unsigned* __restrict__ chase, double* out)
{
  constexpr int RT = TM / 32;   // MFMA row tiles per wave (waves 2 x 2, each (TM/2) x 64)
  __shared__ __attribute__((aligned(16))) double smem[2 * KT * LD + 2 * KT * (TM + 16)];
  double(*Us)[KT][LD] = reinterpret_cast<double(*)[KT][LD]>(smem);
  double(*Vs)[KT][TM + 16] = reinterpret_cast<double(*)[KT][TM + 16]>(smem + 2 * KT * LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, lk = lane >> 4, li = lane & 15;
  for(int e = tid; e < 2 * KT * LD + 2 * KT * (TM + 16); e += 256) smem[e] = 1.0 + e * 1e-9;
  __syncthreads();
  double4_t acc[RT][4];
#pragma unroll
  for(int i = 0; i < RT; ++i)
#pragma unroll
    for(int q = 0; q < 4; ++q) acc[i][q] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int arow = wr * (TM / 2) + 2 * li, bcol = wc * 64 + 2 * li, col2 = 2 * lane;
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0xffffffff, 0x00020000);
  double2_t vreg[RT], ureg[4];
#pragma unroll
  for(int p = 0; p < RT; ++p) vreg[p] = double2_t{1.0 + tid * 1e-9, 1.0};
#pragma unroll
  for(int p = 0; p < 4; ++p) ureg[p] = double2_t{1.0, 1.0 - tid * 1e-9};
  unsigned stream_off = (unsigned)(((size_t)blockIdx.x * 1048576u) % (src_doubles * 8));
  unsigned pos = blockIdx.x * 64u;
  for(int t = 0; t < tiles; ++t) {
    for(int st = 0; st < 16; ++st) {
      const int cur = st & 1;
      double2_t av[2][RT / 2 > 0 ? RT / 2 : 1], bv[2][2];
#pragma unroll
      for(int h = 0; h < RT / 2; ++h) av[0][h] = *reinterpret_cast<const double2_t*>(&Vs[cur][lk][arow + 32 * h]);
#pragma unroll
      for(int h = 0; h < 2; ++h) bv[0][h] = *reinterpret_cast<const double2_t*>(&Us[cur][lk][bcol + 32 * h]);
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) {
        const int pb = kk & 1;
        if(kk + 1 < 4) {
#pragma unroll
          for(int h = 0; h < RT / 2; ++h) av[pb ^ 1][h] = *reinterpret_cast<const double2_t*>(&Vs[cur][4 * (kk + 1) + lk][arow + 32 * h]);
#pragma unroll
          for(int h = 0; h < 2; ++h) bv[pb ^ 1][h] = *reinterpret_cast<const double2_t*>(&Us[cur][4 * (kk + 1) + lk][bcol + 32 * h]);
        }
        if(kk == 1) {
#pragma unroll
          for(int p = 0; p < 4; ++p) *reinterpret_cast<double2_t*>(&Us[cur ^ 1][4 * p + wave][col2]) = -ureg[p];
          if(col2 < TM) {
#pragma unroll
            for(int p = 0; p < 4; ++p) *reinterpret_cast<double2_t*>(&Vs[cur ^ 1][4 * p + wave][col2]) = vreg[p % RT];
          }
#pragma unroll
          for(int p = 0; p < 4; ++p)
            ureg[p] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid, (int)(stream_off + 65536u * p + 4096u), 16));
#pragma unroll
          for(int p = 0; p < RT; ++p)
            vreg[p] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * tid, (int)(stream_off + 65536u * p), 16));
          stream_off += 8192u;
          if(stream_off + 4 * 65536u + 8192u >= (unsigned)(src_doubles * 8)) stream_off = 0;
        }
#pragma unroll
        for(int i = 0; i < RT; ++i)
#pragma unroll
          for(int q = 0; q < 4; ++q)
            acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb][i >> 1][i & 1], bv[pb][q >> 1][q & 1], acc[i][q], 0, 0, 0);
      }
      __syncthreads();
    }
    // emulated per-task overhead: lane 0 chases `dep` dependent sc1 loads, everybody waits at the barrier
    if(tid == 0) {
      for(int d = 0; d < dep; ++d) pos = __hip_atomic_load(chase + (pos & 0xfffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (unsigned)d;
      smem[0] = (double)pos * 1e-30 + 1.0;
    }
    __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for(int i = 0; i < RT; ++i)
#pragma unroll
    for(int q = 0; q < 4; ++q) s += acc[i][q][0] + acc[i][q][1] + acc[i][q][2] + acc[i][q][3];
  out[blockIdx.x * 256 + tid] = s + vreg[0].x + ureg[3].y + smem[0];
}

template <int TM, int WGPC>
static int run(int cus, int tiles128, int dep, const double* src, size_t nsrc, unsigned* chase, double* d)
{
  const int wgs = cus * WGPC;
  const int tiles = tiles128 * (128 / TM) * 2 / WGPC;   // same total flops for every configuration
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((tile_loop<TM, WGPC>), dim3(wgs), dim3(256), 0, 0, 2, dep, src, nsrc, chase, d);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((tile_loop<TM, WGPC>), dim3(wgs), dim3(256), 0, 0, tiles, dep, src, nsrc, chase, d);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * TM * 128 * 256 * (double)tiles * wgs;
  printf("tile %3d x 128, %d WG/CU, overhead chain %2d loads: %8.3f ms  %6.2f TFLOP/s\n", TM, WGPC, dep, ms, flop / (ms * 1e-3) / 1e12);
  return 0;
}

int main()
{
  double *d, *src; unsigned* chase;
  const size_t nsrc = (size_t)64 << 17;
  CK(hipMalloc(&d, 8 * 256 * 2048)); CK(hipMalloc(&src, nsrc * 8)); CK(hipMalloc(&chase, 4u << 20));
  CK(hipMemset(src, 0, nsrc * 8)); CK(hipMemset(chase, 0, 4u << 20));
  for(int dep : {0, 6, 12}) {
    run<128, 2>(240, 120, dep, src, nsrc, chase, d);
    run<64, 2>(240, 120, dep, src, nsrc, chase, d);
    run<64, 3>(240, 120, dep, src, nsrc, chase, d);
    run<64, 4>(240, 120, dep, src, nsrc, chase, d);
  }
  return 0;
}
