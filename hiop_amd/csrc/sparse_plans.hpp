// Internal structures of the sparse plans shared between translation units (csrc/sparse_kernels.hip builds them, csrc/kkt_mds.hip
// walks them inside its fused solve kernels).
#pragma once
#include "common.hpp"

// transposed product through a column-side plan (the pattern is analysed once on the host): per column the positions of its entries
// in the row-sorted triplet list, in list order
struct hiopamd_sp_tplan {
  int nrows = 0, ncols = 0, nnz = 0, max_len = 0;
  int64_t* cptr = nullptr;   // ncols + 1
  int* perm = nullptr;       // nnz: position in the triplet list
  int* prow = nullptr;       // nnz: row of that entry
};

namespace hiopamd {
// first k in [lo, hi) with iRow[k] >= row (hi if none), searched by a whole wave: 64 probes per step instead of one, so
// a row boundary among nnz entries costs log64(nnz) dependent loads (3 for 2^18) instead of log2(nnz) (18)
__device__ __forceinline__ int wave_lower_bound(const int* __restrict__ iRow, int lo, int hi, int row, int lane)
{
  while(hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int pos = lo + lane * step;
    const int v = (pos < hi) ? iRow[pos] : 0x7fffffff;
    const unsigned long long m = __ballot(v >= row);
    const int f = m ? (__ffsll((long long)m) - 1) : 64;   // first probe that is >= row
    const int nlo = (f > 0) ? lo + (f - 1) * step + 1 : lo;
    const int nhi = (f < 64 && lo + f * step < hi) ? lo + f * step : hi;
    lo = nlo;
    hi = nhi;
  }
  const int pos = lo + lane;
  const int v = (pos < hi) ? iRow[pos] : 0x7fffffff;
  const unsigned long long m = __ballot(v >= row);
  return m ? lo + (__ffsll((long long)m) - 1) : hi;
}

}  // namespace hiopamd
