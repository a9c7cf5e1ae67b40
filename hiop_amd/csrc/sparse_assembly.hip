// The triplet ASSEMBLY surface of hiopMatrixSparseTriplet (what the sparse KKT classes and the feasibility-restoration problem
// use to build big triplet matrices out of small ones): copyRowsFrom, copyRowsBlockFrom, copySubmatrixFrom(+Trans),
// copySubDiagonalFrom, setSubDiagonalTo, copyDiagMatrixToSubblock(_w_pattern), setSubmatrixToConstantDiag_w_{col,row}pattern,
// set_Jac_FR, set_Hess_FR.
//
// reference: src/LinAlg/hiopMatrixSparseTriplet.cpp:216-250, :562-720, :790-922, :1042-1172, :1374-1497 — serial loops that walk
// a source in order and write destination entry k, k+1, ...  On the device every destination position is computed in closed
// form: positions that depend on "how many selected / kept entries come before me" come from an exclusive prefix sum
// (three launches: per-block counts, one block over the block counts, per-block scan), row ranges of a row-sorted source from
// binary searches on its row indices.  All index arrays are int32 on the device (hiop_types.h:12-13).
#include "device_utils.hpp"

namespace hiopamd {

constexpr int SCAN_ITEMS = 4;                       // per thread
constexpr int SCAN_BLOCK = kBlock * SCAN_ITEMS;     // per workgroup

template <class F>
__global__ __launch_bounds__(kBlock) void scan_block_sums(int64_t n, F f, int* __restrict__ bsum)
{
  __shared__ int red[kBlock];
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for(int q = 0; q < SCAN_ITEMS; ++q)
    if(base + q < n) s += f(base + q);
  red[threadIdx.x] = s;
  __syncthreads();
  for(int off = kBlock / 2; off > 0; off >>= 1) {
    if((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if(threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
// exclusive scan of the block sums by ONE workgroup (sequential over chunks of kBlock); total -> bsum[nblocks]
__global__ __launch_bounds__(kBlock) void scan_of_block_sums(int nblocks, int* __restrict__ bsum)
{
  __shared__ int sh[kBlock];
  __shared__ int carry;
  if(threadIdx.x == 0) carry = 0;
  __syncthreads();
  for(int c0 = 0; c0 < nblocks; c0 += kBlock) {
    const int i = c0 + threadIdx.x;
    const int v = (i < nblocks) ? bsum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for(int off = 1; off < kBlock; off <<= 1) {   // Hillis-Steele inclusive scan
      const int t = ((int)threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const int incl = sh[threadIdx.x];
    if(i < nblocks) bsum[i] = carry + incl - v;
    __syncthreads();
    if(threadIdx.x == kBlock - 1) carry += incl;
    __syncthreads();
  }
  if(threadIdx.x == 0) bsum[nblocks] = carry;
}
template <class F>
__global__ __launch_bounds__(kBlock) void scan_final(int64_t n, F f, const int* __restrict__ bsum, int* __restrict__ out)
{
  __shared__ int sh[kBlock];
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS], s = 0;
#pragma unroll
  for(int q = 0; q < SCAN_ITEMS; ++q) {
    v[q] = (base + q < n) ? f(base + q) : 0;
    s += v[q];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for(int off = 1; off < kBlock; off <<= 1) {
    const int t = ((int)threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int run = bsum[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for(int q = 0; q < SCAN_ITEMS; ++q) {
    if(base + q < n) out[base + q] = run;
    run += v[q];
  }
}
// out[i] = sum_{k < i} f(k), i = 0..n-1; out[n] is NOT written (the total is in the workspace: returned through *total_dev)
// (workspace: scan_alloc gives the output array and the block sums out of ONE request to the context's grow-only buffer)
static inline int scan_blocks(int64_t n) { return (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK); }
static inline int* scan_alloc(hiopamd_ctx* ctx, int64_t n, int** bsum)
{
  int* w = (int*)ctx_workspace(ctx, sizeof(int) * (size_t)(n + 1 + scan_blocks(n) + 2));
  *bsum = w + n + 1;
  return w;
}
template <class F>
static int exclusive_scan(hiopamd_ctx* ctx, int64_t n, F f, int* out, int* bsum, const int** total_dev = nullptr)
{
  const int nblocks = scan_blocks(n);
  if(total_dev) *total_dev = bsum + nblocks;
  if(n <= 0) {
    HIOPAMD_CHECK(hipMemsetAsync(bsum, 0, sizeof(int) * 2, ctx->stream));
    return HIOPAMD_OK;
  }
  hipLaunchKernelGGL(scan_block_sums<F>, dim3(nblocks), dim3(kBlock), 0, ctx->stream, n, f, bsum);
  hipLaunchKernelGGL(scan_of_block_sums, dim3(1), dim3(kBlock), 0, ctx->stream, nblocks, bsum);
  hipLaunchKernelGGL(scan_final<F>, dim3(nblocks), dim3(kBlock), 0, ctx->stream, n, f, bsum, out);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

// first k with a[k] >= key in the sorted array a[0..n)
__device__ __forceinline__ int lower_bound_dev(const int* a, int n, int key)
{
  int lo = 0, hi = n;
  while(lo < hi) {
    const int mid = (lo + hi) >> 1;
    if(a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

}  // namespace hiopamd

using namespace hiopamd;

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

extern "C" {

// this(start + r, start + r) = scal * d[r], stored at triplet position start_on_nnz_idx + r          (:216-233)
int hiopamd_sp_copy_sub_diagonal_from(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, int start_on_dest_diag, int num_elems,
                                      const double* d, int start_on_nnz_idx, double scal)
{
  if(num_elems < 0 || start_on_dest_diag < 0 || start_on_nnz_idx < 0) return HIOPAMD_ERR_ARG;
  return launch_ew(ctx, num_elems, [=] __device__(int64_t r) {
    const int64_t k = start_on_nnz_idx + r;
    iRow[k] = jCol[k] = start_on_dest_diag + (int)r;
    val[k] = scal * d[r];
  });
}
// the same with a constant                                                                           (:235-249)
int hiopamd_sp_set_sub_diagonal_to(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, int start_on_dest_diag, int num_elems,
                                   double c, int start_on_nnz_idx)
{
  if(num_elems < 0 || start_on_dest_diag < 0 || start_on_nnz_idx < 0) return HIOPAMD_ERR_ARG;
  return launch_ew(ctx, num_elems, [=] __device__(int64_t r) {
    const int64_t k = start_on_nnz_idx + r;
    iRow[k] = jCol[k] = start_on_dest_diag + (int)r;
    val[k] = c;
  });
}
// a constant diagonal block: entry e at (row_st + e, col_st + e), triplet position nnz_st + e         (:671-687)
int hiopamd_sp_copy_diag_matrix_to_subblock(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, double src_val, int dest_row_st,
                                            int dest_col_st, int dest_nnz_st, int nnz_to_copy)
{
  if(nnz_to_copy < 0 || dest_nnz_st < 0) return HIOPAMD_ERR_ARG;
  return launch_ew(ctx, nnz_to_copy, [=] __device__(int64_t e) {
    const int64_t k = dest_nnz_st + e;
    iRow[k] = dest_row_st + (int)e;
    jCol[k] = dest_col_st + (int)e;
    val[k] = src_val;
  });
}

struct PredPattern {
  const double* ix;
  __device__ int operator()(int64_t i) const { return ix[i] != 0.0 ? 1 : 0; }
};
// the selected entries of dx as a diagonal block: the q-th selected entry at (row_st + q, col_st + q)  (:689-719);
// *nnz_found_host (may be null) receives the number of selected entries (the reference asserts it equals nnz_to_copy)
int hiopamd_sp_copy_diag_matrix_to_subblock_w_pattern(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, const double* dx,
                                                      int dest_row_st, int dest_col_st, int dest_nnz_st, int n, const double* ix,
                                                      int* nnz_found_host)
{
  if(n < 0 || dest_nnz_st < 0) return HIOPAMD_ERR_ARG;
  int* bsum = nullptr;
  int* pos = scan_alloc(ctx, n, &bsum);
  const int* total = nullptr;
  RC(exclusive_scan(ctx, n, PredPattern{ix}, pos, bsum, &total));
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
    if(ix[i] != 0.0) {
      const int64_t k = dest_nnz_st + pos[i];
      iRow[k] = dest_row_st + pos[i];
      jCol[k] = dest_col_st + pos[i];
      val[k] = dx[i];
    }
  }));
  if(nnz_found_host) {
    HIOPAMD_CHECK(hipMemcpyAsync(nnz_found_host, total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  }
  return HIOPAMD_OK;
}
// rowpattern = 0: the q-th selected i gives (row_st + i, col_st + q)   (:1110-1138, _w_colpattern)
// rowpattern = 1:                          (row_st + q, col_st + i)   (:1140-1168, _w_rowpattern)
int hiopamd_sp_set_submatrix_to_constant_diag_w_pattern(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, double scalar,
                                                        int dest_row_st, int dest_col_st, int dest_nnz_st, int n, const double* ix,
                                                        int rowpattern, int* nnz_found_host)
{
  if(n < 0 || dest_nnz_st < 0) return HIOPAMD_ERR_ARG;
  int* bsum = nullptr;
  int* pos = scan_alloc(ctx, n, &bsum);
  const int* total = nullptr;
  RC(exclusive_scan(ctx, n, PredPattern{ix}, pos, bsum, &total));
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
    if(ix[i] != 0.0) {
      const int64_t k = dest_nnz_st + pos[i];
      iRow[k] = dest_row_st + (rowpattern ? pos[i] : (int)i);
      jCol[k] = dest_col_st + (rowpattern ? (int)i : pos[i]);
      val[k] = scalar;
    }
  }));
  if(nnz_found_host) {
    HIOPAMD_CHECK(hipMemcpyAsync(nnz_found_host, total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  }
  return HIOPAMD_OK;
}

struct PredKeep {
  const int *i, *j;
  int offdiag_only;
  __device__ int operator()(int64_t k) const { return (offdiag_only && i[k] == j[k]) ? 0 : 1; }
};
// this gets a copy of src (trans = 1: of its transpose) shifted to (row_st, col_st), written from triplet position nnz_st on,
// in the source's order; offdiag_only skips the source's diagonal entries            (:1042-1074, :1076-1108)
int hiopamd_sp_copy_submatrix_from(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, int nnz_src, const int* iRow_src,
                                   const int* jCol_src, const double* val_src, int dest_row_st, int dest_col_st, int dest_nnz_st,
                                   int offdiag_only, int trans)
{
  if(nnz_src < 0 || dest_nnz_st < 0) return HIOPAMD_ERR_ARG;
  const int* si = trans ? jCol_src : iRow_src;
  const int* sj = trans ? iRow_src : jCol_src;
  if(!offdiag_only) {
    return launch_ew(ctx, nnz_src, [=] __device__(int64_t k) {
      const int64_t d = dest_nnz_st + k;
      iRow[d] = dest_row_st + si[k];
      jCol[d] = dest_col_st + sj[k];
      val[d] = val_src[k];
    });
  }
  int* bsum = nullptr;
  int* pos = scan_alloc(ctx, nnz_src, &bsum);
  RC(exclusive_scan(ctx, nnz_src, PredKeep{si, sj, 1}, pos, bsum));
  return launch_ew(ctx, nnz_src, [=] __device__(int64_t k) {
    if(si[k] != sj[k]) {
      const int64_t d = dest_nnz_st + pos[k];
      iRow[d] = dest_row_st + si[k];
      jCol[d] = dest_col_st + sj[k];
      val[d] = val_src[k];
    }
  });
}

// rows rows_src_st .. rows_src_st + n_rows - 1 of the row-sorted src go to rows rows_dest_st .. of this, triplet positions
// dest_nnz_st ..                                                                                       (:619-669)
int hiopamd_sp_copy_rows_block_from(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, int nnz_src, const int* iRow_src,
                                    const int* jCol_src, const double* val_src, int rows_src_idx_st, int n_rows,
                                    int rows_dest_idx_st, int dest_nnz_st)
{
  if(nnz_src < 0 || n_rows < 0 || dest_nnz_st < 0) return HIOPAMD_ERR_ARG;
  return launch_ew(ctx, nnz_src, [=] __device__(int64_t k) {
    const int r = iRow_src[k];
    if(r >= rows_src_idx_st && r < rows_src_idx_st + n_rows) {
      const int lo = lower_bound_dev(iRow_src, nnz_src, rows_src_idx_st);
      const int64_t d = dest_nnz_st + (k - lo);
      iRow[d] = rows_dest_idx_st + (r - rows_src_idx_st);
      jCol[d] = jCol_src[k];
      val[d] = val_src[k];
    }
  });
}

struct RowLen {
  const int* iRow_src;
  int nnz_src;
  const int* rows;
  __device__ int operator()(int64_t q) const
  {
    const int r = rows[q];
    return lower_bound_dev(iRow_src, nnz_src, r + 1) - lower_bound_dev(iRow_src, nnz_src, r);
  }
};
// row q of this = row rows_idxs[q] of the row-sorted src (rows_idxs ascending, device int array)      (:562-611)
int hiopamd_sp_copy_rows_from(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, int nnz_src, const int* iRow_src,
                              const int* jCol_src, const double* val_src, const int* rows_idxs, int n_rows)
{
  if(nnz_src < 0 || n_rows < 0) return HIOPAMD_ERR_ARG;
  int* bsum = nullptr;
  int* pos = scan_alloc(ctx, n_rows, &bsum);
  RC(exclusive_scan(ctx, n_rows, RowLen{iRow_src, nnz_src, rows_idxs}, pos, bsum));
  // one thread per destination row walks its source row (rows are short: a handful of entries)
  return launch_ew(ctx, n_rows, [=] __device__(int64_t q) {
    const int r = rows_idxs[q];
    const int lo = lower_bound_dev(iRow_src, nnz_src, r), hi = lower_bound_dev(iRow_src, nnz_src, r + 1);
    int d = pos[q];
    for(int k = lo; k < hi; ++k, ++d) {
      iRow[d] = (int)q;
      jCol[d] = jCol_src[k];
      val[d] = val_src[k];
    }
  });
}

// this = [Jc -I I 0 0; Jd 0 0 -I I] (the Jacobian of the feasibility-restoration problem), rows of Jc / Jd followed by their
// two slack entries; iJacS / jJacS / MJacS (device, any may be null): the user's copies the reference fills alongside  (:790-922)
int hiopamd_sp_set_jac_fr(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, int n, int m_c, int nnz_c, const int* ic,
                          const int* jc, const double* vc, int m_d, int nnz_d, const int* id, const int* jd, const double* vd,
                          int* iJacS, int* jJacS, double* MJacS)
{
  if(n < 0 || m_c < 0 || m_d < 0 || nnz_c < 0 || nnz_d < 0) return HIOPAMD_ERR_ARG;
  const bool idx = iJacS != nullptr && jJacS != nullptr, vals = MJacS != nullptr;
  // entries of the base Jacobians: entry k of row r moves to k + 2 r (+ the whole c part for Jd)
  RC(launch_ew(ctx, (int64_t)nnz_c + nnz_d, [=] __device__(int64_t t) {
    const bool isc = t < nnz_c;
    const int k = isc ? (int)t : (int)(t - nnz_c);
    const int r = isc ? ic[k] : id[k];
    const int64_t dst = (isc ? 0 : (int64_t)nnz_c + 2 * m_c) + k + 2 * (int64_t)r;
    if(idx) {
      iRow[dst] = iJacS[dst] = r + (isc ? 0 : m_c);
      jCol[dst] = jJacS[dst] = isc ? jc[k] : jd[k];
    }
    if(vals) val[dst] = MJacS[dst] = isc ? vc[k] : vd[k];
  }));
  // the -I / +I entries behind every row
  return launch_ew(ctx, (int64_t)m_c + m_d, [=] __device__(int64_t t) {
    const bool isc = t < m_c;
    const int i = isc ? (int)t : (int)(t - m_c);
    const int end = isc ? lower_bound_dev(ic, nnz_c, i + 1) : lower_bound_dev(id, nnz_d, i + 1);   // row_starts[i + 1]
    const int64_t dst = (isc ? 0 : (int64_t)nnz_c + 2 * m_c) + end + 2 * (int64_t)i;
    if(idx) {
      iRow[dst] = iJacS[dst] = i + (isc ? 0 : m_c);
      iRow[dst + 1] = iJacS[dst + 1] = i + (isc ? 0 : m_c);
      jCol[dst] = jJacS[dst] = isc ? (n + i) : (n + 2 * m_c + i);
      jCol[dst + 1] = jJacS[dst + 1] = isc ? (n + m_c + i) : (n + 2 * m_c + m_d + i);
    }
    if(vals) {
      val[dst] = MJacS[dst] = -1.0;
      val[dst + 1] = MJacS[dst + 1] = 1.0;
    }
  });
}

struct RowHasDiag {   // does row i of the row- and column-sorted upper-triangle Hessian start with its diagonal entry?
  const int *ih, *jh;
  int nnz_h;
  __device__ int operator()(int64_t i) const
  {
    const int lo = lower_bound_dev(ih, nnz_h, (int)i);
    return (lo < nnz_h && ih[lo] == (int)i && jh[lo] == (int)i) ? 1 : 0;
  }
};
// Hessian of the feasibility-restoration problem: every row gets a diagonal entry add_diag[i] (added to the base diagonal
// entry when the row has one), followed by the row's off-diagonal entries; an empty base Hessian (m_h = 0) gives the
// diagonal alone.  iHSS / jHSS / MHSS as above.                                                       (:1374-1497)
int hiopamd_spsym_set_hess_fr(hiopamd_ctx* ctx, int* iRow, int* jCol, double* val, int m_h, int nnz_h, const int* ih,
                              const int* jh, const double* vh, int n_diag, const double* add_diag, int* iHSS, int* jHSS,
                              double* MHSS)
{
  if(m_h < 0 || nnz_h < 0 || n_diag < 0) return HIOPAMD_ERR_ARG;
  const bool idx = iHSS != nullptr && jHSS != nullptr, vals = MHSS != nullptr;
  if(m_h == 0) {
    return launch_ew(ctx, n_diag, [=] __device__(int64_t i) {
      if(idx) {
        iRow[i] = iHSS[i] = (int)i;
        jCol[i] = jHSS[i] = (int)i;
      }
      if(vals) val[i] = MHSS[i] = add_diag[i];
    });
  }
  int* bsum = nullptr;
  int* ndiag_before = scan_alloc(ctx, m_h, &bsum);
  RC(exclusive_scan(ctx, m_h, RowHasDiag{ih, jh, nnz_h}, ndiag_before, bsum));
  // one thread per row: destination of the row = row_start - (diagonals of earlier rows) + i
  return launch_ew(ctx, m_h, [=] __device__(int64_t i) {
    int k_base = lower_bound_dev(ih, nnz_h, (int)i);
    const int k_end = lower_bound_dev(ih, nnz_h, (int)i + 1);
    int64_t k = (int64_t)k_base - ndiag_before[i] + i;
    double dv = add_diag[i];
    const bool has_diag = k_base < k_end && ih[k_base] == jh[k_base];
    if(has_diag) {
      dv += vh[k_base];
      ++k_base;
    }
    if(idx) {
      iRow[k] = iHSS[k] = (int)i;
      jCol[k] = jHSS[k] = (int)i;
    }
    if(vals) val[k] = MHSS[k] = dv;
    ++k;
    for(; k_base < k_end; ++k_base, ++k) {
      if(idx) {
        iRow[k] = iHSS[k] = (int)i;
        jCol[k] = jHSS[k] = jh[k_base];
      }
      if(vals) val[k] = MHSS[k] = vh[k_base];
    }
  });
}

}  // extern "C"
