// Probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 with and without A-block broadcast (cbsz/abid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

template <int CBSZ, int ABID>
__global__ void probe(unsigned long long* out)
{
  const int lane = threadIdx.x;
  for(int la = 0; la < 64; ++la)
    for(int lb = 0; lb < 64; ++lb) {
      const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if(lane == 0) out[la * 64 + lb] = m;
    }
}

template <int CBSZ, int ABID>
static int run(unsigned long long* d)
{
  hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(4096);
  CK(hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost));
  printf("---- cbsz=%d abid=%d: for each A lane la: the B lanes lb it pairs with -> output lane(s)\n", CBSZ, ABID);
  for(int la = 0; la < 64; ++la) {
    printf("la=%2d:", la);
    int cnt = 0;
    for(int lb = 0; lb < 64; ++lb) {
      unsigned long long m = h[la * 64 + lb];
      if(!m) continue;
      printf(" lb%d->", lb);
      for(int l = 0; l < 64; ++l) if((m >> l) & 1ull) printf("%d,", l);
      if(++cnt >= 20) { printf("..."); break; }
    }
    printf("\n");
  }
  return 0;
}

int main()
{
  unsigned long long* d; CK(hipMalloc(&d, 4096 * 8));
  run<0, 0>(d);
  run<2, 0>(d);
  run<2, 1>(d);
  run<2, 3>(d);
  return 0;
}
