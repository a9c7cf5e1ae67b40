// Sparse condensed KKT matrix  M = Jd^T diag(Hd) Jd + H + Dx + delta_wx I  in CSR on the device, and the CSR kernels around it
// (first pinnable piece).
//
// reference: hiopKKTLinSysCondensedSparse::build_kkt_matrix (src/Optimization/hiopKKTLinSysSparseCondensed.cpp:205-335) builds
// M through a chain of general CSR operations — triplet -> CSR, transpose, scale_rows, SpGEMM (times_mat_symbolic/numeric,
// cuSPARSE SpGEMM-reuse on the GPU: src/LinAlg/hiopMatrixSparseCsrCuda.cpp:695-871), three add_matrix passes — each with a
// symbolic and a numeric phase.  The sparsity patterns are fixed over the IPM iterations, so here the WHOLE chain is analysed
// once on the host into one plan: the CSR pattern of M (full symmetric, columns sorted), for every nonzero of Jd^T D Jd the
// list of its (k1, k2, constraint) product triples (the Schur-row-build plan of csrc/sparse_kernels.hip applied to Jd^T), and
// the destination of every Hessian triplet (and of its mirror image) and of every diagonal entry.  The numeric phase is four
// launches: zero, products (one thread / one wave per output nonzero, fixed summation order), Hessian scatter, diagonal.
// The six CSR "diagonal" methods of hiopMatrixSparseCSR (src/LinAlg/hiopMatrixSparseCSR.hpp:97-255: extract_diagonal,
// set_diagonal, scale_rows, scale_cols, form_diag_from_symbolic, form_diag_from_numeric) and the CSR mat-vec are plain entry
// points on (rowptr, colidx, values).
#include "device_utils.hpp"

#include <algorithm>
#include <map>
#include <vector>

struct hiopamd_csr_condensed {
  hiopamd_ctx* ctx = nullptr;
  int n = 0, m = 0, nnzJ = 0, nnzH = 0;
  int64_t nnzM = 0, n_out = 0;
  hiopamd_sp_plan* plan = nullptr;
  // device
  int* rowptr = nullptr;
  int* colidx = nullptr;
  double* vals = nullptr;
  int* permJt = nullptr;        // Jt_val[k] = J_val[permJt[k]]  (Jd^T in row-sorted triplets)
  double* Jt_val = nullptr;
  double* Dinv = nullptr;       // m: 1 / Hd
  int64_t* pos_plan = nullptr;  // plan output -> CSR position
  int64_t* hpos_u = nullptr;    // Hessian triplet -> CSR position of (i, j)
  int64_t* hpos_l = nullptr;    // ... of (j, i), -1 for a diagonal triplet
  int64_t* dpos = nullptr;      // row i -> CSR position of (i, i)
  // rows of M with more than CSR_LONG entries (a dense column of Jd makes a dense row AND column of Jd^T D Jd: the arrowhead of the
  // SparseEx2 pattern, 1e6 entries in row 0): cut in chunks of CSR_LONG, one workgroup per chunk, partial sums folded in chunk order
  struct Long {
    int n_long = 0, n_chunks = 0;
    int* long_row = nullptr;      // n_long: row index
    int* long_first = nullptr;    // n_long + 1: first chunk of the row
    int* chunk_beg = nullptr;     // chunk c covers [chunk_beg[c], chunk_end[c]) of the CSR arrays
    int* chunk_end = nullptr;
    double* chunk_part = nullptr;
    void free_all()
    {
      (void)hipFree(long_row); (void)hipFree(long_first); (void)hipFree(chunk_beg); (void)hipFree(chunk_end); (void)hipFree(chunk_part);
    }
  };
  Long lM, lJt, lJ;
  int* j_rowptr = nullptr;      // Jd (m rows) as CSR over the caller's row-sorted triplets: row pointers (null: the triplets were not row-sorted)
  double avg_row_J = 0.0;
  // Jd^T in CSR (n rows): row pointers and column indices (= constraint rows); values are Jt_val (refreshed by the numeric phase).
  // The transposed product Jd^T y runs as a CSR product — no floating-point atomics (the COO scatter form serialises a million
  // atomic adds on y[0] for the dense column of the SparseEx2 pattern: 12 ms per call), results bit-reproducible.
  int* jt_rowptr = nullptr;
  int* jt_colidx = nullptr;
  double avg_row_M = 0.0, avg_row_Jt = 0.0;
  std::vector<int> h_rowptr, h_colidx;
};
constexpr int CSR_LONG = 4096;

namespace {
template <class T>
int up(T** d, const std::vector<T>& h)
{
  *d = nullptr;
  if(hipMalloc((void**)d, sizeof(T) * (h.size() ? h.size() : 1)) != hipSuccess) return HIOPAMD_ERR_HIP;
  if(!h.empty() && hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return HIOPAMD_ERR_HIP;
  return HIOPAMD_OK;
}
}  // namespace

extern "C" {

int hiopamd_csr_condensed_destroy(hiopamd_csr_condensed* c)
{
  if(!c) return HIOPAMD_OK;
  if(c->ctx) (void)hipStreamSynchronize(c->ctx->stream);
  if(c->plan) hiopamd_sp_plan_destroy(c->plan);
  (void)hipFree(c->rowptr); (void)hipFree(c->colidx); (void)hipFree(c->vals); (void)hipFree(c->permJt);
  (void)hipFree(c->Jt_val); (void)hipFree(c->Dinv); (void)hipFree(c->pos_plan); (void)hipFree(c->hpos_u);
  (void)hipFree(c->hpos_l); (void)hipFree(c->dpos);
  c->lM.free_all();
  c->lJt.free_all();
  c->lJ.free_all();
  (void)hipFree(c->jt_rowptr); (void)hipFree(c->jt_colidx); (void)hipFree(c->j_rowptr);
  delete c;
  return HIOPAMD_OK;
}

// symbolic phase.  Jd: m x n triplets (row-sorted, as hiopMatrixSparseTriplet keeps them); H: upper-triangle triplets of the
// n x n Hessian of the Lagrangian.  Index arrays are HOST pointers.
int hiopamd_csr_condensed_create(hiopamd_csr_condensed** out, hiopamd_ctx* ctx, int n, int m, int nnzJ, const int* iJ_host,
                                 const int* jJ_host, int nnzH, const int* iH_host, const int* jH_host)
{
  if(!out || !ctx || n < 0 || m < 0 || nnzJ < 0 || nnzH < 0) return HIOPAMD_ERR_ARG;
  *out = nullptr;
  for(int k = 0; k < nnzJ; ++k)
    if(iJ_host[k] < 0 || iJ_host[k] >= m || jJ_host[k] < 0 || jJ_host[k] >= n) return HIOPAMD_ERR_ARG;
  for(int k = 0; k < nnzH; ++k)
    if(iH_host[k] < 0 || iH_host[k] >= n || jH_host[k] < iH_host[k] || jH_host[k] >= n) return HIOPAMD_ERR_ARG;   // upper triangle
  auto* c = new hiopamd_csr_condensed();
  c->ctx = ctx; c->n = n; c->m = m; c->nnzJ = nnzJ; c->nnzH = nnzH;
  // Jd^T as row-sorted triplets (n x m): stable sort of the entries by column of Jd
  std::vector<int> perm(nnzJ);
  for(int k = 0; k < nnzJ; ++k) perm[k] = k;
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
    return jJ_host[a] != jJ_host[b] ? jJ_host[a] < jJ_host[b] : iJ_host[a] < iJ_host[b];
  });
  std::vector<int> ti(nnzJ), tj(nnzJ);
  for(int k = 0; k < nnzJ; ++k) {
    ti[k] = jJ_host[perm[k]];
    tj[k] = iJ_host[perm[k]];
  }
  int rc = hiopamd_sp_plan_create(&c->plan, n, n, m, nnzJ, ti.data(), tj.data(), nnzJ, ti.data(), tj.data(), 0);
  if(rc != HIOPAMD_OK) { hiopamd_csr_condensed_destroy(c); return rc; }
  c->n_out = hiopamd_sp_plan_num_outputs(c->plan);
  std::vector<int> oi((size_t)c->n_out + 1), oj((size_t)c->n_out + 1);
  rc = hiopamd_sp_plan_outputs(c->plan, oi.data(), oj.data());
  if(rc != HIOPAMD_OK) { hiopamd_csr_condensed_destroy(c); return rc; }
  // pattern of M = union{ outputs of J^T D J, H and its mirror image, the diagonal }
  std::vector<std::vector<int>> rows((size_t)n);
  for(int64_t p = 0; p < c->n_out; ++p) rows[oi[p]].push_back(oj[p]);
  for(int k = 0; k < nnzH; ++k) {
    rows[iH_host[k]].push_back(jH_host[k]);
    if(iH_host[k] != jH_host[k]) rows[jH_host[k]].push_back(iH_host[k]);
  }
  for(int i = 0; i < n; ++i) rows[i].push_back(i);
  c->h_rowptr.assign((size_t)n + 1, 0);
  for(int i = 0; i < n; ++i) {
    auto& r = rows[i];
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
    c->h_rowptr[i + 1] = c->h_rowptr[i] + (int)r.size();
  }
  c->nnzM = c->h_rowptr[n];
  c->h_colidx.resize((size_t)c->nnzM);
  for(int i = 0; i < n; ++i) std::copy(rows[i].begin(), rows[i].end(), c->h_colidx.begin() + c->h_rowptr[i]);
  auto find = [&](int i, int j) -> int64_t {
    const int* b = c->h_colidx.data() + c->h_rowptr[i];
    const int* e = c->h_colidx.data() + c->h_rowptr[i + 1];
    return (int64_t)(std::lower_bound(b, e, j) - c->h_colidx.data());
  };
  std::vector<int64_t> pos((size_t)c->n_out), hu((size_t)nnzH), hl((size_t)nnzH), dp((size_t)n);
  for(int64_t p = 0; p < c->n_out; ++p) pos[p] = find(oi[p], oj[p]);
  for(int k = 0; k < nnzH; ++k) {
    hu[k] = find(iH_host[k], jH_host[k]);
    hl[k] = iH_host[k] != jH_host[k] ? find(jH_host[k], iH_host[k]) : -1;
  }
  for(int i = 0; i < n; ++i) dp[i] = find(i, i);
  rc = up(&c->rowptr, c->h_rowptr);
  if(rc == HIOPAMD_OK) rc = up(&c->colidx, c->h_colidx);
  if(rc == HIOPAMD_OK) rc = up(&c->permJt, perm);
  if(rc == HIOPAMD_OK) rc = up(&c->pos_plan, pos);
  if(rc == HIOPAMD_OK) rc = up(&c->hpos_u, hu);
  if(rc == HIOPAMD_OK) rc = up(&c->hpos_l, hl);
  if(rc == HIOPAMD_OK) rc = up(&c->dpos, dp);
  {
    auto build_long = [&](const std::vector<int>& rp, int nrows, hiopamd_csr_condensed::Long& L) {
      std::vector<int> lr, lf, cb, ce;
      for(int i = 0; i < nrows; ++i) {
        const int b = rp[i], e = rp[i + 1];
        if(e - b > CSR_LONG) {
          lr.push_back(i);
          lf.push_back((int)cb.size());
          for(int k = b; k < e; k += CSR_LONG) {
            cb.push_back(k);
            ce.push_back(std::min(e, k + CSR_LONG));
          }
        }
      }
      lf.push_back((int)cb.size());
      L.n_long = (int)lr.size();
      L.n_chunks = (int)cb.size();
      int r = up(&L.long_row, lr);
      if(r == HIOPAMD_OK) r = up(&L.long_first, lf);
      if(r == HIOPAMD_OK) r = up(&L.chunk_beg, cb);
      if(r == HIOPAMD_OK) r = up(&L.chunk_end, ce);
      if(r == HIOPAMD_OK && hipMalloc((void**)&L.chunk_part, sizeof(double) * (size_t)std::max(L.n_chunks, 1)) != hipSuccess) r = HIOPAMD_ERR_HIP;
      return r;
    };
    if(rc == HIOPAMD_OK) rc = build_long(c->h_rowptr, n, c->lM);
    std::vector<int> trp((size_t)n + 1, 0);
    for(int k = 0; k < nnzJ; ++k) trp[(size_t)ti[k] + 1] += 1;
    for(int i = 0; i < n; ++i) trp[(size_t)i + 1] += trp[i];
    if(rc == HIOPAMD_OK) rc = up(&c->jt_rowptr, trp);
    if(rc == HIOPAMD_OK) rc = up(&c->jt_colidx, tj);
    if(rc == HIOPAMD_OK) rc = build_long(trp, n, c->lJt);
    // Jd itself: the caller's triplets are row-sorted by the reference's convention (hiopMatrixSparseTriplet): their row pointers make
    // them a CSR matrix, and y = Jd x gets the same short-row / long-row treatment as the other two products (a wave per row of two
    // entries was 0.67 ms for 1e6 rows: two thirds of a solveCompressed of the SparseEx2-shaped bench case)
    bool sorted = true;
    for(int k = 1; k < nnzJ && sorted; ++k) sorted = iJ_host[k - 1] <= iJ_host[k];
    if(sorted && m > 0) {
      std::vector<int> jrp((size_t)m + 1, 0);
      for(int k = 0; k < nnzJ; ++k) jrp[(size_t)iJ_host[k] + 1] += 1;
      for(int i = 0; i < m; ++i) jrp[(size_t)i + 1] += jrp[i];
      if(rc == HIOPAMD_OK) rc = up(&c->j_rowptr, jrp);
      if(rc == HIOPAMD_OK) rc = build_long(jrp, m, c->lJ);
      c->avg_row_J = (double)nnzJ / m;
    }
    c->avg_row_M = n ? (double)c->nnzM / n : 0.0;
    c->avg_row_Jt = n ? (double)nnzJ / n : 0.0;
  }
  if(rc == HIOPAMD_OK && hipMalloc((void**)&c->vals, sizeof(double) * (size_t)(c->nnzM ? c->nnzM : 1)) != hipSuccess) rc = HIOPAMD_ERR_HIP;
  if(rc == HIOPAMD_OK && hipMalloc((void**)&c->Jt_val, sizeof(double) * (size_t)(nnzJ ? nnzJ : 1)) != hipSuccess) rc = HIOPAMD_ERR_HIP;
  if(rc == HIOPAMD_OK && hipMalloc((void**)&c->Dinv, sizeof(double) * (size_t)(m ? m : 1)) != hipSuccess) rc = HIOPAMD_ERR_HIP;
  if(rc != HIOPAMD_OK) { hiopamd_csr_condensed_destroy(c); return rc; }
  *out = c;
  return HIOPAMD_OK;
}

int64_t hiopamd_csr_condensed_nnz(const hiopamd_csr_condensed* c) { return c ? c->nnzM : 0; }
int64_t hiopamd_csr_condensed_num_products(const hiopamd_csr_condensed* c) { return c ? hiopamd_sp_plan_num_products(c->plan) : 0; }
const int* hiopamd_csr_condensed_rowptr(const hiopamd_csr_condensed* c) { return c ? c->rowptr : nullptr; }
const int* hiopamd_csr_condensed_colidx(const hiopamd_csr_condensed* c) { return c ? c->colidx : nullptr; }
double* hiopamd_csr_condensed_values(hiopamd_csr_condensed* c) { return c ? c->vals : nullptr; }
int hiopamd_csr_condensed_pattern(const hiopamd_csr_condensed* c, int* rowptr_host, int* colidx_host)
{
  if(!c || !rowptr_host || !colidx_host) return HIOPAMD_ERR_ARG;
  std::copy(c->h_rowptr.begin(), c->h_rowptr.end(), rowptr_host);
  std::copy(c->h_colidx.begin(), c->h_colidx.end(), colidx_host);
  return HIOPAMD_OK;
}

// numeric phase: values of M for the current iterate.  J_val (nnzJ, in the order of the triplets given at creation), H_val (nnzH),
// Hd (m: the diagonal weights of the condensation, hiopKKTLinSysSparseCondensed.cpp:240-250), Dx (n), all device pointers.
int hiopamd_csr_condensed_numeric(hiopamd_csr_condensed* c, const double* J_val, const double* H_val, const double* Hd,
                                  const double* Dx, double delta_wx)
{
  if(!c || (c->nnzJ && !J_val) || (c->nnzH && !H_val) || (c->m && !Hd) || (c->n && !Dx)) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = c->ctx;
  double* vals = c->vals;
  int rc = hiopamd_vec_set_to_constant(ctx, c->nnzM, vals, 0.0);
  if(rc != HIOPAMD_OK) return rc;
  {
    const int* perm = c->permJt;
    double* jt = c->Jt_val;
    rc = hiopamd::launch_ew(ctx, c->nnzJ, [=] __device__(int64_t k) { jt[k] = J_val[perm[k]]; });
    if(rc != HIOPAMD_OK) return rc;
    double* di = c->Dinv;
    rc = hiopamd::launch_ew(ctx, c->m, [=] __device__(int64_t r) { di[r] = 1.0 / Hd[r]; });
    if(rc != HIOPAMD_OK) return rc;
  }
  // J^T D J: sum_r Jt[i, r] Jt[j, r] / (1 / Hd[r]), fixed order (increasing r) per output nonzero
  rc = hiopamd_sp_MDinvNt_scatter(ctx, c->plan, c->Jt_val, c->Jt_val, c->Dinv, 1.0, vals, c->pos_plan);
  if(rc != HIOPAMD_OK) return rc;
  {
    const int64_t *hu = c->hpos_u, *hl = c->hpos_l;
    const int64_t nnzH = c->nnzH;
    rc = hiopamd::launch_ew(ctx, nnzH, [=] __device__(int64_t k) {   // duplicates in the triplet list accumulate, like add_matrix
      // (ordered triplets: the duplicates of one (i, j) are neighbours and map to the same position; the first of the run adds them in
      //  storage order and touches the destination once — no result depends on the order in which threads arrive)
      if(k > 0 && hu[k - 1] == hu[k]) return;
      double acc = 0.0;
      for(int64_t q = k; q < nnzH && hu[q] == hu[k]; ++q) acc += H_val[q];
      atomicAdd(&vals[hu[k]], acc);
      if(hl[k] >= 0) atomicAdd(&vals[hl[k]], acc);
    });
    if(rc != HIOPAMD_OK) return rc;
    const int64_t* dp = c->dpos;
    rc = hiopamd::launch_ew(ctx, c->n, [=] __device__(int64_t i) { vals[dp[i]] += Dx[i] + delta_wx; });
  }
  return rc;
}

/* ---- generic CSR kernels (row pointers / column indices int32, device) ---- */
// y = beta y + alpha A x : one wave per row (rows of the condensed KKT hold 5-50 entries)
namespace {
__global__ __launch_bounds__(hiopamd::kBlock) void csr_spmv_kernel(int nrows, const int* __restrict__ rowptr,
                                                                   const int* __restrict__ colidx, const double* __restrict__ val,
                                                                   double beta, double* __restrict__ y, double alpha,
                                                                   const double* __restrict__ x)
{
  const int lane = threadIdx.x & 63;
  const int wpb = hiopamd::kBlock / 64;
  for(int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < nrows; row += gridDim.x * wpb) {
    double acc = 0.0;
    for(int k = rowptr[row] + lane; k < rowptr[row + 1]; k += 64) acc += val[k] * x[colidx[k]];
    for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if(lane == 0) y[row] = (beta == 0.0 ? 0.0 : beta * y[row]) + alpha * acc;
  }
}
// the same, rows with more than `long_thresh` entries left to csr_spmv_chunk_kernel
__global__ __launch_bounds__(hiopamd::kBlock) void csr_spmv_short_kernel(int nrows, const int* __restrict__ rowptr,
                                                                         const int* __restrict__ colidx, const double* __restrict__ val,
                                                                         double beta, double* __restrict__ y, double alpha,
                                                                         const double* __restrict__ x, int long_thresh)
{
  const int lane = threadIdx.x & 63;
  const int wpb = hiopamd::kBlock / 64;
  for(int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < nrows; row += gridDim.x * wpb) {
    const int b = rowptr[row], e = rowptr[row + 1];
    if(e - b > long_thresh) continue;
    double acc = 0.0;
    for(int k = b + lane; k < e; k += 64) acc += val[k] * x[colidx[k]];
    for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if(lane == 0) y[row] = (beta == 0.0 ? 0.0 : beta * y[row]) + alpha * acc;
  }
}
// rows of a handful of entries (the condensed matrix of a sparse NLP with 2-3 entries per constraint): one THREAD per row
__global__ __launch_bounds__(hiopamd::kBlock) void csr_spmv_thread_kernel(int nrows, const int* __restrict__ rowptr,
                                                                          const int* __restrict__ colidx, const double* __restrict__ val,
                                                                          double beta, double* __restrict__ y, double alpha,
                                                                          const double* __restrict__ x, int long_thresh)
{
  for(int64_t row = (int64_t)blockIdx.x * hiopamd::kBlock + threadIdx.x; row < nrows; row += (int64_t)gridDim.x * hiopamd::kBlock) {
    const int b = rowptr[row], e = rowptr[row + 1];
    if(e - b > long_thresh) continue;
    double acc = 0.0;
    for(int k = b; k < e; ++k) acc += val[k] * x[colidx[k]];
    y[row] = (beta == 0.0 ? 0.0 : beta * y[row]) + alpha * acc;
  }
}
// one workgroup per chunk of a long row: partial sum, fixed order inside the chunk
__global__ __launch_bounds__(hiopamd::kBlock) void csr_spmv_chunk_kernel(const int* __restrict__ cb, const int* __restrict__ ce,
                                                                         const int* __restrict__ colidx, const double* __restrict__ val,
                                                                         const double* __restrict__ x, double* __restrict__ part)
{
  const int b = cb[blockIdx.x], e = ce[blockIdx.x];
  double acc = 0.0;
  for(int k = b + (int)threadIdx.x; k < e; k += hiopamd::kBlock) acc += val[k] * x[colidx[k]];
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double sm[hiopamd::kBlock / 64];
  if((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if(threadIdx.x == 0) {
    double v = sm[0];
    for(int w = 1; w < hiopamd::kBlock / 64; ++w) v += sm[w];
    part[blockIdx.x] = v;
  }
}
}  // namespace
int hiopamd_csr_times_vec(hiopamd_ctx* ctx, int nrows, const int* rowptr, const int* colidx, const double* val, double beta,
                          double* y, double alpha, const double* x)
{
  if(nrows < 0) return HIOPAMD_ERR_ARG;
  if(nrows == 0) return HIOPAMD_OK;
  const int wpb = hiopamd::kBlock / 64;
  int grid = (nrows + wpb - 1) / wpb;
  if(grid > 65536) grid = 65536;
  hipLaunchKernelGGL(csr_spmv_kernel, dim3(grid), dim3(hiopamd::kBlock), 0, ctx->stream, nrows, rowptr, colidx, val, beta, y, alpha, x);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
// hiopMatrixSparseCSR::extract_diagonal (:97) — 0 where the diagonal entry is not stored
int hiopamd_csr_extract_diagonal(hiopamd_ctx* ctx, int n, const int* rowptr, const int* colidx, const double* val, double* diag)
{
  return hiopamd::launch_ew(ctx, n, [=] __device__(int64_t i) {
    double d = 0.0;
    for(int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if(colidx[k] == (int)i) d = val[k];
    diag[i] = d;
  });
}
// set_diagonal (:105): every STORED diagonal entry = value
int hiopamd_csr_set_diagonal(hiopamd_ctx* ctx, int n, const int* rowptr, const int* colidx, double* val, double value)
{
  return hiopamd::launch_ew(ctx, n, [=] __device__(int64_t i) {
    for(int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if(colidx[k] == (int)i) val[k] = value;
  });
}
// scale_rows (:178): A <- diag(D) A ; scale_cols (:175): A <- A diag(D)
int hiopamd_csr_scale_rows(hiopamd_ctx* ctx, int n, const int* rowptr, double* val, const double* D)
{
  return hiopamd::launch_ew(ctx, n, [=] __device__(int64_t i) {
    const double d = D[i];
    for(int k = rowptr[i]; k < rowptr[i + 1]; ++k) val[k] *= d;
  });
}
int hiopamd_csr_scale_cols(hiopamd_ctx* ctx, int64_t nnz, const int* colidx, double* val, const double* D)
{
  return hiopamd::launch_ew(ctx, nnz, [=] __device__(int64_t k) { val[k] *= D[colidx[k]]; });
}
// form_diag_from_symbolic (:244) / _numeric (:255): the n x n diagonal matrix diag(D) as CSR
int hiopamd_csr_form_diag_symbolic(hiopamd_ctx* ctx, int n, int* rowptr, int* colidx)
{
  return hiopamd::launch_ew(ctx, (int64_t)n + 1, [=] __device__(int64_t i) {
    rowptr[i] = (int)i;
    if(i < n) colidx[i] = (int)i;
  });
}
int hiopamd_csr_form_diag_numeric(hiopamd_ctx* ctx, int n, double* val, const double* D)
{
  return hiopamd_vec_copy(ctx, n, val, D);
}

/* operator callbacks for hiopamd_krylov_create (hiopamd_linop_fn): y = M x and the Jacobi preconditioner y = x ./ diag(M) */
// y = beta y + alpha A x for a CSR matrix whose long rows were listed at creation: short rows one wave (or one thread, when rows
// hold a handful of entries) each, long rows one workgroup per chunk + a fold in chunk order
// one wave per long row: its chunk sums folded lane-strided + shuffle tree (a fixed order; a single thread walking the 245 chunks of
// a 1e6-entry row took 21 us)
__global__ __launch_bounds__(64) void csr_long_fold_kernel(const int* __restrict__ lr, const int* __restrict__ lf,
                                                           const double* __restrict__ part, double beta, double* __restrict__ y, double alpha)
{
  const int q = blockIdx.x, lane = threadIdx.x;
  double acc = 0.0;
  for(int k = lf[q] + lane; k < lf[q + 1]; k += 64) acc += part[k];
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if(lane == 0) {
    const int r = lr[q];
    y[r] = (beta == 0.0 ? 0.0 : beta * y[r]) + alpha * acc;
  }
}

static int csr_apply_long(hiopamd_ctx* ctx, int n, const int* rowptr, const int* colidx, const double* vals,
                          const hiopamd_csr_condensed::Long& L, double avg_row, double beta, double* y, double alpha, const double* x)
{
  if(n == 0) return HIOPAMD_OK;
  if(avg_row <= 6.0) {
    hipLaunchKernelGGL(csr_spmv_thread_kernel, dim3(hiopamd::grid_for(n)), dim3(hiopamd::kBlock), 0, ctx->stream, n, rowptr, colidx, vals,
                       beta, y, alpha, x, CSR_LONG);
  } else {
    const int wpb = hiopamd::kBlock / 64;
    int grid = (n + wpb - 1) / wpb;
    if(grid > 65536) grid = 65536;
    hipLaunchKernelGGL(csr_spmv_short_kernel, dim3(grid), dim3(hiopamd::kBlock), 0, ctx->stream, n, rowptr, colidx, vals, beta, y, alpha,
                       x, CSR_LONG);
  }
  if(L.n_long == 0) {
    HIOPAMD_CHECK(hipGetLastError());
    return HIOPAMD_OK;
  }
  hipLaunchKernelGGL(csr_spmv_chunk_kernel, dim3(L.n_chunks), dim3(hiopamd::kBlock), 0, ctx->stream, L.chunk_beg, L.chunk_end, colidx, vals,
                     x, L.chunk_part);
  hipLaunchKernelGGL(csr_long_fold_kernel, dim3(L.n_long), dim3(64), 0, ctx->stream, L.long_row, L.long_first, L.chunk_part, beta, y, alpha);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_csr_condensed_apply(void* user, const double* x_dev, double* y_dev)
{
  auto* c = static_cast<hiopamd_csr_condensed*>(user);
  return csr_apply_long(c->ctx, c->n, c->rowptr, c->colidx, c->vals, c->lM, c->avg_row_M, 0.0, y_dev, 1.0, x_dev);
}
// Jd^T's values for new Jacobian values (device, in the order of the triplets given at creation); the numeric phase does the same
int hiopamd_csr_condensed_refresh_jt(hiopamd_csr_condensed* c, const double* J_val)
{
  if(!c || (c->nnzJ && !J_val)) return HIOPAMD_ERR_ARG;
  const int* perm = c->permJt;
  double* jt = c->Jt_val;
  return hiopamd::launch_ew(c->ctx, c->nnzJ, [=] __device__(int64_t k) { jt[k] = J_val[perm[k]]; });
}
// y (n) = beta y + alpha Jd^T x (m) with the values of the last numeric phase / refresh
int hiopamd_csr_condensed_jac_trans_times_vec(hiopamd_csr_condensed* c, double beta, double* y, double alpha, const double* x)
{
  if(!c) return HIOPAMD_ERR_ARG;
  return csr_apply_long(c->ctx, c->n, c->jt_rowptr, c->jt_colidx, c->Jt_val, c->lJt, c->avg_row_Jt, beta, y, alpha, x);
}
// y (m) = beta y + alpha Jd x (n) on the caller's own (row-sorted) triplet arrays: jJ_dev = their column indices, J_val their values;
// HIOPAMD_ERR_STATE when the triplets given at creation were not row-sorted (the caller then uses hiopamd_sp_times_vec)
int hiopamd_csr_condensed_jac_times_vec(hiopamd_csr_condensed* c, const int* jJ_dev, const double* J_val, double beta, double* y,
                                        double alpha, const double* x)
{
  if(!c || (c->nnzJ && (!jJ_dev || !J_val))) return HIOPAMD_ERR_ARG;
  if(c->m == 0) return HIOPAMD_OK;
  if(!c->j_rowptr) return HIOPAMD_ERR_STATE;
  return csr_apply_long(c->ctx, c->m, c->j_rowptr, jJ_dev, J_val, c->lJ, c->avg_row_J, beta, y, alpha, x);
}
// diag(M) through the diagonal positions found at creation (hiopamd_csr_extract_diagonal scans a row per thread: a million
// sequential loads for the dense row of an arrowhead)
int hiopamd_csr_condensed_diagonal(hiopamd_csr_condensed* c, double* diag_dev)
{
  if(!c || !diag_dev) return HIOPAMD_ERR_ARG;
  const double* vals = c->vals;
  const int64_t* dp = c->dpos;
  return hiopamd::launch_ew(c->ctx, c->n, [=] __device__(int64_t i) { diag_dev[i] = vals[dp[i]]; });
}
int hiopamd_csr_condensed_jacobi(void* user, const double* x_dev, double* y_dev)
{
  auto* c = static_cast<hiopamd_csr_condensed*>(user);
  const double* vals = c->vals;
  const int64_t* dp = c->dpos;
  return hiopamd::launch_ew(c->ctx, c->n, [=] __device__(int64_t i) { y_dev[i] = x_dev[i] / vals[dp[i]]; });
}

}  // extern "C"
