// hiopMatrixSparseTriplet kernels for gfx950: COO SpMV and the MDS Schur row-build
//   W[r0+i][c0+j] += alpha * sum_c M1[i,c] * M2[j,c] / D[c].
//
// reference: src/LinAlg/hiopMatrixSparseTriplet.cpp:73 (timesVec), :110 (transTimesVec),
// :390 (addMDinvMtransToDiagBlockOfSymDeMatUTri), :447 (addMDinvNtransToSymDeMatUTri).
// The reference walks ALL m1*m2 row pairs with a sorted merge (O(m^2) merges even when rows are
// disjoint) and its RAJA GPU port runs one thread per row over all j>i
// (src/LinAlg/hiopMatrixRajaSparseTripletImpl.hpp:807-847).  Here the sparsity pattern — fixed over
// the IPM iterations (reference comment :479-489) — is analysed ONCE on the host into a plan:
// the list of structurally non-zero outputs (i,j) and, for each, its (k1,k2,col) product triples
// in increasing column order.  The numeric kernel is then one thread (short lists) or one wave
// (long lists) per output: no atomics, fixed summation order = the reference's merge order.
#include "device_utils.hpp"
#include "sparse_plans.hpp"

#include <algorithm>
#include <vector>

struct hiopamd_sp_plan {
  int m1 = 0, m2 = 0;
  int64_t n_out = 0, n_prod = 0;
  int64_t n_short = 0, n_long = 0;  // outputs are stored short-first
  // device arrays
  int* out_i = nullptr;
  int* out_j = nullptr;
  int64_t* out_ptr = nullptr;  // n_out+1
  int* k1 = nullptr;
  int* k2 = nullptr;
  int* col = nullptr;
};

namespace hiopamd {

constexpr int kLongList = 32;

__global__ __launch_bounds__(kBlock) void coo_spmv_rows(int nrows, int nnz, const int* __restrict__ iRow,
                                                        const int* __restrict__ jCol, const double* __restrict__ val,
                                                        double beta, double* __restrict__ y, double alpha,
                                                        const double* __restrict__ x, double* __restrict__ y2)
{
  // one wave per row; the row's [lo,hi) range found by binary search in the row-sorted COO
  const int row = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if(row >= nrows) return;   // wave-uniform
  const int start = wave_lower_bound(iRow, 0, nnz, row, lane);
  const int end = wave_lower_bound(iRow, start, nnz, row + 1, lane);
  double acc = 0.0;
  for(int k = start + lane; k < end; k += 64) acc += x[jCol[k]] * val[k];
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if(lane == 0) {
    const double r = (beta == 0.0 ? 0.0 : beta * y[row]) + alpha * acc;
    y[row] = r;
    if(y2) y2[row] = r;
  }
}

// few-row matrices (e.g. the 3 inequality rows of MdsEx1, one of them with n_s/2 entries): every row is cut
// into `split` slices (16, or more when the rows are very long: one slice per ~16K entries of the average row -- the border rows of
// the bordered-diagonal solver hold 1e6 entries), one workgroup per (row, slice); the slice sums are folded in slice order.
constexpr int SPMV_SPLIT = 16, SPMV_SPLIT_MAX = 1024;
__global__ __launch_bounds__(kBlock) void coo_spmv_rows_split(int nrows, int nnz, const int* __restrict__ iRow,
                                                              const int* __restrict__ jCol,
                                                              const double* __restrict__ val,
                                                              const double* __restrict__ x, double* __restrict__ part, int split)
{
  const int row = blockIdx.x / split, sl = blockIdx.x % split;
  // (a single row owns every entry: no searches -- two dependent-load chains of 20 steps were most of the kernel's 38 us at nnz = 1e6)
  const int start = nrows == 1 ? 0 : wave_lower_bound(iRow, 0, nnz, row, threadIdx.x & 63);       // every wave of the workgroup: same result
  const int end = nrows == 1 ? nnz : wave_lower_bound(iRow, start, nnz, row + 1, threadIdx.x & 63);
  const int len = end - start;
  const int chunk = (len + split - 1) / split;
  const int b0 = start + sl * chunk;
  int b1 = b0 + chunk;
  if(b1 > end) b1 = end;
  double acc = 0.0;
  for(int k = b0 + threadIdx.x; k < b1; k += kBlock) acc += x[jCol[k]] * val[k];
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double sm[kBlock / 64];
  if((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if(threadIdx.x == 0) part[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}
__global__ void coo_spmv_fold(int nrows, const double* __restrict__ part, double beta, double* __restrict__ y,
                              double alpha, double* __restrict__ y2, int split)
{
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if(row >= nrows) return;
  double s = 0.0;
  for(int q = 0; q < split; ++q) s += part[(int64_t)row * split + q];
  const double r = (beta == 0.0 ? 0.0 : beta * y[row]) + alpha * s;
  y[row] = r;
  if(y2) y2[row] = r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Transposed and symmetric products: many entries add to one output element, in an order the hardware chooses.  Floating-point
// atomics made the result depend on that order (rounds 1-3: unsafeAtomicAdd on doubles; the IPM branches on reductions of these
// vectors).  Here every contribution is added EXACTLY, as a 96-bit fixed-point number (three 32-bit limbs, each accumulated in a
// 64-bit word with integer atomics: exact, commutative, associative — any order gives the same words), and rounded once at the end:
//   pass 1  max |contribution|  (integer atomicMax on the bit patterns of non-negative doubles)  -> quantum 2^(E - 93)
//   pass 2  limbs of round-toward-zero(contribution / quantum) added to acc[4 * output + 0..2] (word 3: non-finite contributions seen)
//   pass 3  y = beta * y + (acc as a double) — bitwise the same run to run, and more accurate than a chain of rounded additions
//           (a term smaller than 2^-93 of the largest one is dropped).
// Up to 2^32 contributions per output element; a +-Inf / NaN contribution makes ITS output +-Inf / NaN, as a chain of additions would.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int exact_ilogb(unsigned long long bits)
{
  const int e = (int)((bits >> 52) & 0x7ffull);
  return e ? e - 1023 : -1022;
}
// acc3 += trunc(p * 2^-eq) as a signed 96-bit integer
__device__ __forceinline__ void exact_add(unsigned long long* acc3, double p, int eq)
{
  const unsigned long long bits = (unsigned long long)__double_as_longlong(p);
  const int eb = (int)((bits >> 52) & 0x7ffull);
  unsigned long long m = bits & 0x000fffffffffffffull;
  if(eb == 0x7ff) {   // word 3: bit 0 = +Inf seen, bit 1 = -Inf seen, bit 2 = NaN seen
    atomicOr(acc3 + 3, m ? 4ull : ((bits >> 63) ? 2ull : 1ull));
    return;
  }
  if(eb) m |= 0x0010000000000000ull;
  if(m == 0ull) return;
  const int sh = (eb ? eb - 1023 : -1022) - 52 - eq;   // p = m * 2^(e - 52) = (m << sh) quanta
  unsigned long long lo, hi;                          // |value| = hi * 2^64 + lo, < 2^95
  if(sh >= 0) {
    if(sh >= 64) {   // (cannot happen with eq = E - 93, kept for safety)
      lo = 0ull;
      hi = m << (sh - 64);
    } else {
      lo = m << sh;
      hi = sh ? (m >> (64 - sh)) : 0ull;
    }
  } else {
    if(sh <= -53) return;
    lo = m >> (-sh);
    hi = 0ull;
  }
  if(bits >> 63) {   // two's complement of the 128-bit magnitude
    lo = ~lo + 1ull;
    hi = ~hi + (lo == 0ull ? 1ull : 0ull);
  }
  const unsigned long long l0 = lo & 0xffffffffull, l1 = lo >> 32;
  const unsigned long long l2 = (unsigned long long)(long long)(int)(unsigned)hi;   // sign-extended low 32 bits of hi
  if(l0) atomicAdd(acc3, l0);
  if(l1) atomicAdd(acc3 + 1, l1);
  if(l2) atomicAdd(acc3 + 2, l2);
}
__device__ __forceinline__ double exact_value(const unsigned long long* acc3, int eq)
{
  // total = acc0 + acc1 * 2^32 + (signed) acc2 * 2^64
  unsigned __int128 t = (unsigned __int128)acc3[0] + ((unsigned __int128)acc3[1] << 32) + ((unsigned __int128)acc3[2] << 64);
  __int128 st = (__int128)t;   // (mod 2^128 arithmetic: the true total fits)
  const bool negative = st < 0;
  unsigned __int128 mag = negative ? (unsigned __int128)(-st) : (unsigned __int128)st;
  if(mag == 0) return 0.0;
  const unsigned long long mh = (unsigned long long)(mag >> 64), ml = (unsigned long long)mag;
  int shift;
  unsigned long long top;   // the leading 64 bits
  if(mh) {
    const int lz = __clzll((long long)mh);
    top = lz ? ((mh << lz) | (ml >> (64 - lz))) : mh;
    shift = 64 - lz;
  } else {
    const int lz = __clzll((long long)ml);
    top = ml << lz;
    shift = -lz;
  }
  const double d = (double)top;   // correctly rounded; the bits below `top` are dropped (deterministically)
  const double v = ldexp(d, shift + eq);
  return negative ? -v : v;
}
__device__ __forceinline__ void exact_max(unsigned long long* word, double p)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(p));   // NaN compares above Inf: propagates
  if(b) atomicMax(word, b);
}

// mode 0: y^T += alpha * M^T x (COO);  mode 1: symmetric product from upper-triangle triplets
template <int MODE, int PASS>
__global__ __launch_bounds__(kBlock) void sp_exact_pass(int nnz, const int* __restrict__ iRow, const int* __restrict__ jCol,
                                                         const double* __restrict__ val, double alpha, const double* __restrict__ x,
                                                         unsigned long long* __restrict__ maxw, unsigned long long* __restrict__ acc)
{
  int eq = 0;
  if(PASS == 2) {
    const unsigned long long mb = *maxw;
    eq = exact_ilogb(mb) - 93;   // (mb = largest FINITE magnitude; 0: only zeros and non-finite terms, whose flags still have to be set)
  }
  unsigned long long lmax = 0ull;
  for(int k = blockIdx.x * kBlock + threadIdx.x; k < nnz; k += gridDim.x * kBlock) {
    const int i = iRow[k], j = jCol[k];
    const double v = val[k];
    const double p1 = MODE == 0 ? alpha * x[i] * v : alpha * x[j] * v;   // MODE 0 -> y[j];  MODE 1 -> y[i]
    const double p2 = (MODE == 1 && i != j) ? alpha * x[i] * v : 0.0;      //                    MODE 1 -> y[j]
    if(PASS == 1) {
      const unsigned long long b1 = (unsigned long long)__double_as_longlong(fabs(p1)), b2 = (unsigned long long)__double_as_longlong(fabs(p2));
      if(((b1 >> 52) & 0x7ffull) != 0x7ffull) lmax = lmax > b1 ? lmax : b1;
      if(((b2 >> 52) & 0x7ffull) != 0x7ffull) lmax = lmax > b2 ? lmax : b2;
    } else {
      exact_add(acc + 4 * (int64_t)(MODE == 0 ? j : i), p1, eq);
      if(MODE == 1 && i != j) exact_add(acc + 4 * (int64_t)j, p2, eq);
    }
  }
  if(PASS == 1) {
    for(int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_down(lmax, off, 64);
      lmax = lmax > o ? lmax : o;
    }
    if((threadIdx.x & 63) == 0 && lmax) atomicMax(maxw, lmax);
  }
}
__global__ __launch_bounds__(kBlock) void sp_exact_finish(int n, const unsigned long long* __restrict__ maxw,
                                                           const unsigned long long* __restrict__ acc, double beta, double* __restrict__ y)
{
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if(c >= n) return;
  const unsigned long long mb = *maxw;
  const unsigned long long fl = acc[4 * (int64_t)c + 3];
  double add = mb == 0ull ? 0.0 : exact_value(acc + 4 * (int64_t)c, exact_ilogb(mb) - 93);
  if(fl) {
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    add = (fl & 4ull) || (fl & 3ull) == 3ull ? inf - inf : ((fl & 1ull) ? inf : -inf);
  }
  y[c] = (beta == 0.0 ? 0.0 : beta * y[c]) + add;
}

__global__ __launch_bounds__(kBlock) void mdinv_short(int64_t n_short, const int* __restrict__ out_i,
                                                      const int* __restrict__ out_j,
                                                      const int64_t* __restrict__ out_ptr, const int* __restrict__ k1,
                                                      const int* __restrict__ k2, const int* __restrict__ col,
                                                      const double* __restrict__ v1, const double* __restrict__ v2,
                                                      const double* __restrict__ D, double alpha,
                                                      double* __restrict__ W, int64_t ldw, int r0, int c0,
                                                      const int64_t* __restrict__ pos /* != null: W[pos[p]] (sparse sink) */)
{
  for(int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n_short; p += (int64_t)gridDim.x * kBlock) {
    double acc = 0.0;
    const int64_t e = out_ptr[p + 1];
    for(int64_t q = out_ptr[p]; q < e; ++q) acc += v1[k1[q]] / D[col[q]] * v2[k2[q]];
    if(pos) W[pos[p]] += alpha * acc;
    else W[(int64_t)(r0 + out_i[p]) * ldw + (c0 + out_j[p])] += alpha * acc;
  }
}

// long outputs (a pair of rows with thousands of common columns — the inequality rows of MdsEx1): the product list of one
// output is cut into MDINV_SPLIT slices, one workgroup per (output, slice) — a single workgroup per output walked its
// 1e5 gather-products in 400 dependent rounds (170 us for 6 outputs on an otherwise idle device); the slice sums are
// folded in slice order, so the result does not depend on the schedule.
constexpr int MDINV_SPLIT = 64;
__global__ __launch_bounds__(kBlock) void mdinv_long(int64_t first, int64_t n_long, const int64_t* __restrict__ out_ptr,
                                                     const int* __restrict__ k1, const int* __restrict__ k2,
                                                     const int* __restrict__ col, const double* __restrict__ v1,
                                                     const double* __restrict__ v2, const double* __restrict__ D,
                                                     double* __restrict__ part)
{
  const int64_t o = blockIdx.x / MDINV_SPLIT;
  const int sl = blockIdx.x % MDINV_SPLIT;
  if(o >= n_long) return;
  const int64_t p = first + o;
  const int64_t b = out_ptr[p], e = out_ptr[p + 1];
  const int64_t chunk = (e - b + MDINV_SPLIT - 1) / MDINV_SPLIT;
  const int64_t q0 = b + sl * chunk;
  int64_t q1 = q0 + chunk;
  if(q1 > e) q1 = e;
  double acc = 0.0;
  for(int64_t q = q0 + threadIdx.x; q < q1; q += kBlock) acc += v1[k1[q]] / D[col[q]] * v2[k2[q]];
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double sm[kBlock / 64];
  if((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if(threadIdx.x == 0) part[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}
__global__ __launch_bounds__(64) void mdinv_long_fold(int64_t first, int64_t n_long, const int* __restrict__ out_i,
                                                      const int* __restrict__ out_j, const double* __restrict__ part,
                                                      double alpha, double* __restrict__ W, int64_t ldw, int r0, int c0,
                                                      const int64_t* __restrict__ pos)
{
  const int64_t o = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if(o >= n_long) return;
  double s = 0.0;
  for(int q = 0; q < MDINV_SPLIT; ++q) s += part[o * MDINV_SPLIT + q];
  const int64_t p = first + o;
  if(pos) W[pos[p]] += alpha * s;
  else W[(int64_t)(r0 + out_i[p]) * ldw + (c0 + out_j[p])] += alpha * s;
}

// y[vec_start+row] += alpha * Msym[row,row] for row in [diag_src_start, diag_src_start+num_elems)
// (the destination index is vec_start+row exactly as in the reference loop,
//  hiopMatrixSparseTriplet.cpp:1034-1042; every caller passes diag_src_start = 0)
__global__ __launch_bounds__(kBlock) void spsym_diag_to_vec(int nnz, const int* __restrict__ iRow,
                                                            const int* __restrict__ jCol,
                                                            const double* __restrict__ val, double alpha,
                                                            double* __restrict__ y, int vec_start, int diag_src_start,
                                                            int num_elems)
{
  // The triplets are ordered by (row, column) (hiopMatrixSparseTriplet::checkIndexesAreOrdered), so repeated entries of one (i, j)
  // are neighbours: the FIRST entry of such a run sums the run in storage order in a register and adds it to the destination with ONE
  // atomic add — with ordered triplets that is the only add the destination receives (one fixed order of additions: bitwise
  // reproducible; a sym-sparse matrix normally holds one entry per (i, j): the run has length 1); with unordered input that repeats an
  // (i, j) in non-adjacent places every run still arrives (only the order of those few adds is then not fixed).
  for(int k = blockIdx.x * kBlock + threadIdx.x; k < nnz; k += gridDim.x * kBlock) {
    const int r = iRow[k];
    if(r == jCol[k] && r >= diag_src_start && r < diag_src_start + num_elems) {
      if(k > 0 && iRow[k - 1] == r && jCol[k - 1] == r) continue;   // not the first of its run
      double acc = alpha * val[k];
      for(int q = k + 1; q < nnz && iRow[q] == r && jCol[q] == r; ++q) acc += alpha * val[q];
      atomicAdd(&y[vec_start + r], acc);
    }
  }
}

__global__ __launch_bounds__(kBlock) void spsym_add_upper(int nnz, const int* __restrict__ iRow,
                                                          const int* __restrict__ jCol, const double* __restrict__ val,
                                                          int diag_start, double alpha, double* __restrict__ W,
                                                          int64_t ldw)
{
  for(int k = blockIdx.x * kBlock + threadIdx.x; k < nnz; k += gridDim.x * kBlock) {   // (run ownership as in spsym_diag_to_vec)
    const int r = iRow[k], c = jCol[k];
    if(r <= c) {
      if(k > 0 && iRow[k - 1] == r && jCol[k - 1] == c) continue;
      double acc = alpha * val[k];
      for(int q = k + 1; q < nnz && iRow[q] == r && jCol[q] == c; ++q) acc += alpha * val[q];
      atomicAdd(&W[(int64_t)(diag_start + r) * ldw + (diag_start + c)], acc);
    }
  }
}

// y[c] = beta y[c] + alpha sum_{entries of column c, in list order} x[row] * val: one thread per column (columns of the KKT Jacobians
// hold a handful of entries), or one wave per column with a fixed-order tree when some column is long — no atomics either way
__global__ __launch_bounds__(kBlock) void sp_tplan_cols(int ncols, const int64_t* __restrict__ cptr, const int* __restrict__ perm,
                                                         const int* __restrict__ prow, const double* __restrict__ val, double beta,
                                                         double* __restrict__ y, double alpha, const double* __restrict__ x)
{
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if(c >= ncols) return;
  double acc = 0.0;
  for(int64_t p = cptr[c]; p < cptr[c + 1]; ++p) acc += x[prow[p]] * val[perm[p]];
  y[c] = (beta == 0.0 ? 0.0 : beta * y[c]) + alpha * acc;
}
__global__ __launch_bounds__(kBlock) void sp_tplan_cols_wave(int ncols, const int64_t* __restrict__ cptr, const int* __restrict__ perm,
                                                              const int* __restrict__ prow, const double* __restrict__ val, double beta,
                                                              double* __restrict__ y, double alpha, const double* __restrict__ x)
{
  const int c = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if(c >= ncols) return;   // wave-uniform
  double acc = 0.0;
  for(int64_t p = cptr[c] + lane; p < cptr[c + 1]; p += 64) acc += x[prow[p]] * val[perm[p]];
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if(lane == 0) y[c] = (beta == 0.0 ? 0.0 : beta * y[c]) + alpha * acc;
}

template <class T>
static int upload(T** d, const std::vector<T>& h)
{
  *d = nullptr;
  size_t bytes = sizeof(T) * (h.size() ? h.size() : 1);
  HIOPAMD_CHECK(hipMalloc((void**)d, bytes));
  if(h.size()) HIOPAMD_CHECK(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return HIOPAMD_OK;
}

}  // namespace hiopamd

using namespace hiopamd;

namespace {
struct PredUnordered {   // hiopMatrixSparseTriplet::checkIndexesAreOrdered :377-388
  const int *i, *j;
  __device__ bool operator()(int64_t k) const
  {
    return k > 0 && (i[k] < i[k - 1] || (i[k] == i[k - 1] && j[k] < j[k - 1]));
  }
};
struct PredOffDiag {     // is_diagonal :1338-1353 / hiopMatrixSymSparseTriplet::numberOfOffDiagNonzeros :1171-1183
  const int *i, *j;
  __device__ bool operator()(int64_t k) const { return i[k] != j[k]; }
};
template <class Pred>
struct OpCountSp {
  Pred p;
  __device__ double identity() const { return 0.0; }
  __device__ double map(int64_t i) const { return p(i) ? 1.0 : 0.0; }
  __device__ double combine(double a, double b) const { return a + b; }
};
}  // namespace

namespace {
template <int MODE>
int sp_exact_product(hiopamd_ctx* ctx, int nout, int nnz, const int* iRow, const int* jCol, const double* val, double beta, double* y,
                     double alpha, const double* x)
{
  // workspace: [max word | pad] [4 words per output: three limbs + the non-finite flags]
  const size_t words = 2 + 4 * (size_t)nout;
  unsigned long long* w = (unsigned long long*)ctx_workspace(ctx, sizeof(unsigned long long) * words);
  HIOPAMD_CHECK(hipMemsetAsync(w, 0, sizeof(unsigned long long) * words, ctx->stream));
  if(nnz > 0) {
    hipLaunchKernelGGL((sp_exact_pass<MODE, 1>), dim3(grid_for(nnz)), dim3(kBlock), 0, ctx->stream, nnz, iRow, jCol, val, alpha, x, w, w + 2);
    hipLaunchKernelGGL((sp_exact_pass<MODE, 2>), dim3(grid_for(nnz)), dim3(kBlock), 0, ctx->stream, nnz, iRow, jCol, val, alpha, x, w, w + 2);
  }
  hipLaunchKernelGGL(sp_exact_finish, dim3((unsigned)((nout + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, nout, w, w + 2, beta, y);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
}  // namespace

extern "C" {

// y = beta y + alpha M x; y2 != nullptr: the result is stored there as well (the MDS solve packs it into its right-hand side)
int hiopamd_sp_times_vec_copy(hiopamd_ctx* ctx, int nrows, int ncols, int nnz, const int* iRow, const int* jCol, const double* val,
                              double beta, double* y, double alpha, const double* x, double* y2)
{
  (void)ncols;
  if(nrows < 0 || nnz < 0) return HIOPAMD_ERR_ARG;
  if(nrows == 0) return HIOPAMD_OK;
  if(nrows <= 256 && nnz > 64 * nrows) {
    int split = SPMV_SPLIT;
    while(split < SPMV_SPLIT_MAX && (int64_t)nnz / nrows > (int64_t)split * 16384) split *= 2;
    double* part = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)nrows * split);
    hipLaunchKernelGGL(coo_spmv_rows_split, dim3(nrows * split), dim3(kBlock), 0, ctx->stream, nrows, nnz, iRow,
                       jCol, val, x, part, split);
    hipLaunchKernelGGL(coo_spmv_fold, dim3((nrows + 63) / 64), dim3(64), 0, ctx->stream, nrows, part, beta, y, alpha, y2, split);
    HIOPAMD_CHECK(hipGetLastError());
    return HIOPAMD_OK;
  }
  const int64_t threads = (int64_t)nrows * 64;
  hipLaunchKernelGGL(coo_spmv_rows, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                     nrows, nnz, iRow, jCol, val, beta, y, alpha, x, y2);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
int hiopamd_sp_times_vec(hiopamd_ctx* ctx, int nrows, int ncols, int nnz, const int* iRow, const int* jCol,
                         const double* val, double beta, double* y, double alpha, const double* x)
{
  return hiopamd_sp_times_vec_copy(ctx, nrows, ncols, nnz, iRow, jCol, val, beta, y, alpha, x, nullptr);
}

int hiopamd_sp_trans_times_vec(hiopamd_ctx* ctx, int nrows, int ncols, int nnz, const int* iRow, const int* jCol,
                               const double* val, double beta, double* y, double alpha, const double* x)
{
  (void)nrows;
  if(ncols < 0 || nnz < 0) return HIOPAMD_ERR_ARG;
  if(ncols == 0) return HIOPAMD_OK;
  return sp_exact_product<0>(ctx, ncols, nnz, iRow, jCol, val, beta, y, alpha, x);
}

int hiopamd_spsym_times_vec(hiopamd_ctx* ctx, int n, int nnz, const int* iRow, const int* jCol, const double* val,
                            double beta, double* y, double alpha, const double* x)
{
  if(n < 0 || nnz < 0) return HIOPAMD_ERR_ARG;
  if(n == 0) return HIOPAMD_OK;
  return sp_exact_product<1>(ctx, n, nnz, iRow, jCol, val, beta, y, alpha, x);
}

int hiopamd_sp_tplan_create(hiopamd_sp_tplan** out, int nrows, int ncols, int nnz, const int* iRow_host, const int* jCol_host)
{
  if(!out || nrows < 0 || ncols < 0 || nnz < 0 || (nnz > 0 && (!iRow_host || !jCol_host))) return HIOPAMD_ERR_ARG;
  for(int k = 0; k < nnz; ++k)
    if(iRow_host[k] < 0 || iRow_host[k] >= nrows || jCol_host[k] < 0 || jCol_host[k] >= ncols) return HIOPAMD_ERR_ARG;
  std::vector<int64_t> cptr((size_t)ncols + 1, 0);
  for(int k = 0; k < nnz; ++k) cptr[(size_t)jCol_host[k] + 1]++;
  int max_len = 0;
  for(int c = 0; c < ncols; ++c) {
    max_len = std::max<int64_t>(max_len, cptr[(size_t)c + 1]);
    cptr[(size_t)c + 1] += cptr[c];
  }
  std::vector<int> perm((size_t)nnz), prow((size_t)nnz);
  {
    std::vector<int64_t> pos(cptr.begin(), cptr.end() - 1);
    for(int k = 0; k < nnz; ++k) {   // stable: list order inside a column
      const int64_t p = pos[jCol_host[k]]++;
      perm[(size_t)p] = k;
      prow[(size_t)p] = iRow_host[k];
    }
  }
  hiopamd_sp_tplan* pl = new hiopamd_sp_tplan();
  pl->nrows = nrows;
  pl->ncols = ncols;
  pl->nnz = nnz;
  pl->max_len = max_len;
  int st = upload(&pl->cptr, cptr);
  if(st == HIOPAMD_OK) st = upload(&pl->perm, perm);
  if(st == HIOPAMD_OK) st = upload(&pl->prow, prow);
  if(st != HIOPAMD_OK) {
    hiopamd_sp_tplan_destroy(pl);
    return st;
  }
  *out = pl;
  return HIOPAMD_OK;
}
int hiopamd_sp_tplan_destroy(hiopamd_sp_tplan* pl)
{
  if(!pl) return HIOPAMD_OK;
  (void)hipFree(pl->cptr);
  (void)hipFree(pl->perm);
  (void)hipFree(pl->prow);
  delete pl;
  return HIOPAMD_OK;
}
int hiopamd_sp_tplan_trans_times_vec(hiopamd_ctx* ctx, const hiopamd_sp_tplan* pl, const double* val, double beta, double* y,
                                     double alpha, const double* x)
{
  if(!ctx || !pl) return HIOPAMD_ERR_ARG;
  if(pl->ncols == 0) return HIOPAMD_OK;
  if(pl->max_len <= 64)
    hipLaunchKernelGGL(sp_tplan_cols, dim3((unsigned)((pl->ncols + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, pl->ncols,
                       pl->cptr, pl->perm, pl->prow, val, beta, y, alpha, x);
  else
    hipLaunchKernelGGL(sp_tplan_cols_wave, dim3((unsigned)(((int64_t)pl->ncols * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       pl->ncols, pl->cptr, pl->perm, pl->prow, val, beta, y, alpha, x);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_sp_plan_create(hiopamd_sp_plan** out, int m1, int m2, int ncols, int nnz1, const int* iRow1, const int* jCol1,
                           int nnz2, const int* iRow2, const int* jCol2, int same_matrix_upper_only)
{
  if(!out || m1 < 0 || m2 < 0 || ncols < 0 || nnz1 < 0 || nnz2 < 0) return HIOPAMD_ERR_ARG;
  for(int k = 0; k < nnz1; ++k)
    if(iRow1[k] < 0 || iRow1[k] >= m1 || jCol1[k] < 0 || jCol1[k] >= ncols || (k && iRow1[k] < iRow1[k - 1]))
      return HIOPAMD_ERR_ARG;
  for(int k = 0; k < nnz2; ++k)
    if(iRow2[k] < 0 || iRow2[k] >= m2 || jCol2[k] < 0 || jCol2[k] >= ncols || (k && iRow2[k] < iRow2[k - 1]))
      return HIOPAMD_ERR_ARG;

  // CSC of M2: for each column, its (row j, k2) entries in increasing j (input is row-sorted)
  std::vector<int64_t> cptr((size_t)ncols + 1, 0);
  for(int k = 0; k < nnz2; ++k) cptr[jCol2[k] + 1]++;
  for(int c = 0; c < ncols; ++c) cptr[c + 1] += cptr[c];
  std::vector<int> crow(nnz2), ck2(nnz2);
  {
    std::vector<int64_t> pos(cptr.begin(), cptr.end() - 1);
    for(int k = 0; k < nnz2; ++k) {
      int64_t p = pos[jCol2[k]]++;
      crow[p] = iRow2[k];
      ck2[p] = k;
    }
  }
  struct Trip {
    int j, col, k1, k2;
  };
  std::vector<int> oi, oj, pk1, pk2, pcol;
  std::vector<int64_t> optr;
  std::vector<int64_t> len;
  optr.push_back(0);
  std::vector<Trip> cand;
  int k = 0;
  for(int i = 0; i < m1; ++i) {
    cand.clear();
    for(; k < nnz1 && iRow1[k] == i; ++k) {
      const int c = jCol1[k];
      for(int64_t p = cptr[c]; p < cptr[c + 1]; ++p) {
        if(same_matrix_upper_only && crow[p] < i) continue;
        cand.push_back(Trip{crow[p], c, k, ck2[p]});
      }
    }
    std::sort(cand.begin(), cand.end(), [](const Trip& a, const Trip& b) {
      if(a.j != b.j) return a.j < b.j;
      if(a.col != b.col) return a.col < b.col;
      if(a.k1 != b.k1) return a.k1 < b.k1;
      return a.k2 < b.k2;
    });
    for(size_t t = 0; t < cand.size();) {
      size_t e = t;
      while(e < cand.size() && cand[e].j == cand[t].j) ++e;
      oi.push_back(i);
      oj.push_back(cand[t].j);
      for(size_t q = t; q < e; ++q) {
        pk1.push_back(cand[q].k1);
        pk2.push_back(cand[q].k2);
        pcol.push_back(cand[q].col);
      }
      optr.push_back((int64_t)pk1.size());
      t = e;
    }
  }
  // reorder outputs: short lists first, long lists last (stable)
  const int64_t n_out = (int64_t)oi.size();
  std::vector<int64_t> order;
  order.reserve(n_out);
  int64_t n_short = 0;
  for(int64_t p = 0; p < n_out; ++p)
    if(optr[p + 1] - optr[p] <= kLongList) {
      order.push_back(p);
      ++n_short;
    }
  for(int64_t p = 0; p < n_out; ++p)
    if(optr[p + 1] - optr[p] > kLongList) order.push_back(p);
  std::vector<int> oi2(n_out), oj2(n_out), a1(pk1.size()), a2(pk1.size()), ac(pk1.size());
  std::vector<int64_t> optr2(n_out + 1, 0);
  int64_t w = 0;
  for(int64_t t = 0; t < n_out; ++t) {
    const int64_t p = order[t];
    oi2[t] = oi[p];
    oj2[t] = oj[p];
    for(int64_t q = optr[p]; q < optr[p + 1]; ++q, ++w) {
      a1[w] = pk1[q];
      a2[w] = pk2[q];
      ac[w] = pcol[q];
    }
    optr2[t + 1] = w;
  }
  hiopamd_sp_plan* pl = new hiopamd_sp_plan();
  pl->m1 = m1;
  pl->m2 = m2;
  pl->n_out = n_out;
  pl->n_prod = (int64_t)a1.size();
  pl->n_short = n_short;
  pl->n_long = n_out - n_short;
  int st = upload(&pl->out_i, oi2);
  if(st == HIOPAMD_OK) st = upload(&pl->out_j, oj2);
  if(st == HIOPAMD_OK) st = upload(&pl->out_ptr, optr2);
  if(st == HIOPAMD_OK) st = upload(&pl->k1, a1);
  if(st == HIOPAMD_OK) st = upload(&pl->k2, a2);
  if(st == HIOPAMD_OK) st = upload(&pl->col, ac);
  if(st != HIOPAMD_OK) {
    hiopamd_sp_plan_destroy(pl);
    return st;
  }
  *out = pl;
  return HIOPAMD_OK;
}

int hiopamd_sp_plan_destroy(hiopamd_sp_plan* pl)
{
  if(!pl) return HIOPAMD_OK;
  hipFree(pl->out_i);
  hipFree(pl->out_j);
  hipFree(pl->out_ptr);
  hipFree(pl->k1);
  hipFree(pl->k2);
  hipFree(pl->col);
  delete pl;
  return HIOPAMD_OK;
}
int64_t hiopamd_sp_plan_num_outputs(const hiopamd_sp_plan* pl) { return pl ? pl->n_out : 0; }
int64_t hiopamd_sp_plan_num_products(const hiopamd_sp_plan* pl) { return pl ? pl->n_prod : 0; }

static int sp_add_MDinvNt_impl(hiopamd_ctx* ctx, const hiopamd_sp_plan* pl, const double* val1, const double* val2,
                               const double* D, double alpha, double* W, int64_t ldw, int r0, int c0, const int64_t* pos)
{
  if(!pl) return HIOPAMD_ERR_ARG;
  if(pl->n_short > 0) {
    hipLaunchKernelGGL(mdinv_short, dim3(grid_for(pl->n_short)), dim3(kBlock), 0, ctx->stream, pl->n_short, pl->out_i,
                       pl->out_j, pl->out_ptr, pl->k1, pl->k2, pl->col, val1, val2, D, alpha, W, ldw, r0, c0, pos);
  }
  if(pl->n_long > 0) {
    double* part = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)pl->n_long * MDINV_SPLIT);
    hipLaunchKernelGGL(mdinv_long, dim3((unsigned)(pl->n_long * MDINV_SPLIT)), dim3(kBlock), 0, ctx->stream, pl->n_short,
                       pl->n_long, pl->out_ptr, pl->k1, pl->k2, pl->col, val1, val2, D, part);
    hipLaunchKernelGGL(mdinv_long_fold, dim3((unsigned)((pl->n_long + 63) / 64)), dim3(64), 0, ctx->stream, pl->n_short,
                       pl->n_long, pl->out_i, pl->out_j, part, alpha, W, ldw, r0, c0, pos);
  }
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
int hiopamd_sp_add_MDinvNt(hiopamd_ctx* ctx, const hiopamd_sp_plan* pl, const double* val1, const double* val2,
                           const double* D, double alpha, double* W, int64_t ldw, int r0, int c0)
{
  return sp_add_MDinvNt_impl(ctx, pl, val1, val2, D, alpha, W, ldw, r0, c0, nullptr);
}
/* the same products scattered into a SPARSE destination: out_vals[pos[p]] += alpha * (output p), pos a device array of
 * hiopamd_sp_plan_num_outputs entries (used by the condensed sparse KKT assembly, csrc/csr_condensed.hip) */
int hiopamd_sp_MDinvNt_scatter(hiopamd_ctx* ctx, const hiopamd_sp_plan* pl, const double* val1, const double* val2,
                               const double* D, double alpha, double* out_vals, const int64_t* pos_dev)
{
  if(!pos_dev) return HIOPAMD_ERR_ARG;
  return sp_add_MDinvNt_impl(ctx, pl, val1, val2, D, alpha, out_vals, 0, 0, 0, pos_dev);
}
/* (row, column) of every output of the plan, in the plan's own order (host arrays of hiopamd_sp_plan_num_outputs ints) */
int hiopamd_sp_plan_outputs(const hiopamd_sp_plan* pl, int* out_i_host, int* out_j_host)
{
  if(!pl || !out_i_host || !out_j_host) return HIOPAMD_ERR_ARG;
  if(pl->n_out == 0) return HIOPAMD_OK;
  HIOPAMD_CHECK(hipMemcpy(out_i_host, pl->out_i, sizeof(int) * (size_t)pl->n_out, hipMemcpyDeviceToHost));
  HIOPAMD_CHECK(hipMemcpy(out_j_host, pl->out_j, sizeof(int) * (size_t)pl->n_out, hipMemcpyDeviceToHost));
  return HIOPAMD_OK;
}

int hiopamd_spsym_add_diag_to_vec(hiopamd_ctx* ctx, int nnz, const int* iRow, const int* jCol, const double* val,
                                  double alpha, double* y, int vec_start, int n_vec, int diag_src_start, int num_elems)
{
  if(nnz < 0 || vec_start < 0 || diag_src_start < 0) return HIOPAMD_ERR_ARG;
  if(num_elems < 0) num_elems = n_vec;
  if(diag_src_start + num_elems + vec_start > n_vec) num_elems = n_vec - vec_start - diag_src_start;
  if(nnz == 0 || num_elems <= 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(spsym_diag_to_vec, dim3(grid_for(nnz)), dim3(kBlock), 0, ctx->stream, nnz, iRow, jCol, val, alpha, y,
                     vec_start, diag_src_start, num_elems);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_spsym_add_upper_to_sym_upper(hiopamd_ctx* ctx, int nnz, const int* iRow, const int* jCol, const double* val,
                                         int diag_start, double alpha, double* W, int64_t ldw)
{
  if(nnz < 0 || diag_start < 0) return HIOPAMD_ERR_ARG;
  if(nnz == 0) return HIOPAMD_OK;
  hipLaunchKernelGGL(spsym_add_upper, dim3(grid_for(nnz)), dim3(kBlock), 0, ctx->stream, nnz, iRow, jCol, val,
                     diag_start, alpha, W, ldw);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

/* ---- the rest of the hiopMatrixSparseTriplet surface used outside the Schur build (one thread per triplet) ---- */
// block of W += alpha*this^T, destination inside the upper triangle (hiopMatrixSparseTriplet.cpp:255-275)
int hiopamd_sp_trans_add_to_sym_upper(hiopamd_ctx* ctx, int nnz, const int* iRow, const int* jCol, const double* val,
                                      int row_start, int col_start, double alpha, double* W, int64_t ldw)
{
  if(nnz < 0 || row_start < 0 || col_start < 0) return HIOPAMD_ERR_ARG;
  // entries of one triplet matrix are distinct (i,j) pairs: no two threads touch the same W element
  return hiopamd::launch_ew(ctx, nnz, [=] __device__(int64_t k) {
    W[(int64_t)(jCol[k] + row_start) * ldw + (iRow[k] + col_start)] += alpha * val[k];
  });
}
// ret[i] = max_k |val[k]| over the triplets of row i, 0 for an empty row (:285-300).  |v| >= 0, so the IEEE bit pattern
// orders like an unsigned integer and a 64-bit atomic max is exact and order-independent.
int hiopamd_sp_row_max_abs(hiopamd_ctx* ctx, int nrows, int nnz, const int* iRow, const double* val, double* ret)
{
  if(nrows < 0 || nnz < 0) return HIOPAMD_ERR_ARG;
  int st = hiopamd::launch_ew(ctx, nrows, [=] __device__(int64_t i) { ret[i] = 0.0; });
  if(st != HIOPAMD_OK) return st;
  unsigned long long* r = reinterpret_cast<unsigned long long*>(ret);
  return hiopamd::launch_ew(ctx, nnz, [=] __device__(int64_t k) {
    const double a = fabs(val[k]);
    if(a == a) atomicMax(&r[iRow[k]], (unsigned long long)__double_as_longlong(a));
  });
}
// val[k] *= scal[iRow[k]]  or  *= 1/scal[iRow[k]]  (:303-319)
int hiopamd_sp_scale_rows(hiopamd_ctx* ctx, int nnz, const int* iRow, double* val, const double* scal, int inv)
{
  if(nnz < 0) return HIOPAMD_ERR_ARG;
  if(inv) return hiopamd::launch_ew(ctx, nnz, [=] __device__(int64_t k) { val[k] *= 1.0 / scal[iRow[k]]; });
  return hiopamd::launch_ew(ctx, nnz, [=] __device__(int64_t k) { val[k] *= scal[iRow[k]]; });
}
// W = dense(this)  (:363-374; duplicates, if any, accumulate like the reference's +=)
int hiopamd_sp_copy_to_dense(hiopamd_ctx* ctx, int nrows, int ncols, int nnz, const int* iRow, const int* jCol,
                             const double* val, double* W, int64_t ldw)
{
  if(nrows < 0 || ncols < 0 || nnz < 0) return HIOPAMD_ERR_ARG;
  int st = hiopamd_mat_set_to_constant(ctx, nrows, ncols, W, ldw, 0.0);
  if(st != HIOPAMD_OK) return st;
  // ordered triplets (the class invariant): repeated entries of one (i, j) are neighbours; the first of a run adds the run in storage
  // order and issues ONE atomic add onto the zeroed destination — a fixed result.  (Unordered input with scattered duplicates still
  // gets every contribution, through several atomics.)
  return hiopamd::launch_ew(ctx, nnz, [=] __device__(int64_t k) {
    const int r = iRow[k], c = jCol[k];
    if(k > 0 && iRow[k - 1] == r && jCol[k - 1] == c) return;
    double acc = 0.0;
    for(int64_t q = k; q < nnz && iRow[q] == r && jCol[q] == c; ++q) acc += val[q];
    atomicAdd(&W[(int64_t)r * ldw + c], acc);
  });
}
int hiopamd_sp_indexes_ordered(hiopamd_ctx* ctx, int nnz, const int* iRow, const int* jCol, int* out_host)
{
  if(nnz < 0 || !out_host) return HIOPAMD_ERR_ARG;
  double c = 0.0;
  int st = hiopamd::launch_reduce<double>(ctx, nnz, OpCountSp<PredUnordered>{{iRow, jCol}}, &c);
  *out_host = (c == 0.0);
  return st;
}
int hiopamd_sp_num_offdiag(hiopamd_ctx* ctx, int nnz, const int* iRow, const int* jCol, int64_t* out_host)
{
  if(nnz < 0 || !out_host) return HIOPAMD_ERR_ARG;
  double c = 0.0;
  int st = hiopamd::launch_reduce<double>(ctx, nnz, OpCountSp<PredOffDiag>{{iRow, jCol}}, &c);
  *out_host = (int64_t)c;
  return st;
}
// diag[i] = this[i][i] (0 when absent)  (:1355-1372)
int hiopamd_sp_extract_diagonal(hiopamd_ctx* ctx, int n, int nnz, const int* iRow, const int* jCol, const double* val,
                                double* diag)
{
  if(n < 0 || nnz < 0) return HIOPAMD_ERR_ARG;
  int st = hiopamd::launch_ew(ctx, n, [=] __device__(int64_t i) { diag[i] = 0.0; });
  if(st != HIOPAMD_OK) return st;
  return hiopamd::launch_ew(ctx, nnz, [=] __device__(int64_t k) {
    if(iRow[k] == jCol[k]) diag[iRow[k]] = val[k];
  });
}

}  // extern "C"
