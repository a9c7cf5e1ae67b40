#include "hiopLinSolverSymSparseHipNative.hpp"

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <vector>

namespace hiop
{
hiopLinSolverSymSparseHipNative::hiopLinSolverSymSparseHipNative(size_type n, size_type nnz, hiopNlpFormulation* nlp)
    : hiopLinSolverSymSparse(nlp), ctx_(hiopamd_default_ctx()), n_((int)n)
{
  // (the barebone base constructor leaves M_ unset and not owned: this class owns the matrix it creates here)
  M_ = new hiopMatrixSymSparseTripletHipNative((int)n, (int)nnz);
  sys_mat_owned_ = true;
}

hiopLinSolverSymSparseHipNative::hiopLinSolverSymSparseHipNative(hiopMatrixSparse* M, hiopNlpFormulation* nlp)
    : hiopLinSolverSymSparse(M, nlp), ctx_(hiopamd_default_ctx()), n_(M ? (int)M->m() : 0)
{
}

hiopLinSolverSymSparseHipNative::~hiopLinSolverSymSparseHipNative()
{
  hiopamd_ctx_sync(ctx_);
  if(ldl_) hiopamd_sparse_ldl_destroy(ldl_);
  if(csr_vals_) hiopamd_free(csr_vals_);
  if(gather_) hiopamd_free(gather_);
}

int hiopLinSolverSymSparseHipNative::first_call()
{
  const int nnz = (int)M_->numberOfNonzeros();
  std::vector<int> ti((size_t)nnz), tj((size_t)nnz);
  int rc = hiopamd_copy_d2h(ctx_, ti.data(), static_cast<const hiopMatrixSparse*>(M_)->i_row(), sizeof(int) * (size_t)nnz);
  if(rc == HIOPAMD_OK) rc = hiopamd_copy_d2h(ctx_, tj.data(), static_cast<const hiopMatrixSparse*>(M_)->j_col(), sizeof(int) * (size_t)nnz);
  if(rc != HIOPAMD_OK) return rc;
  // both triangles, row by row, columns ascending; entry (i, j), i != j, of the triplet appears at (i, j) and (j, i)
  struct E {
    int r, c, t;
  };
  std::vector<E> es;
  es.reserve(2 * (size_t)nnz);
  for(int t = 0; t < nnz; ++t) {
    if(ti[(size_t)t] < 0 || ti[(size_t)t] >= n_ || tj[(size_t)t] < 0 || tj[(size_t)t] >= n_) return HIOPAMD_ERR_ARG;
    es.push_back({ti[(size_t)t], tj[(size_t)t], t});
    if(ti[(size_t)t] != tj[(size_t)t]) es.push_back({tj[(size_t)t], ti[(size_t)t], t});
  }
  std::sort(es.begin(), es.end(), [](const E& a, const E& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
  for(size_t q = 1; q < es.size(); ++q)
    if(es[q].r == es[q - 1].r && es[q].c == es[q - 1].c) return HIOPAMD_ERR_ARG;   // duplicate entries: not the reference's symmetric triplet
  std::vector<int> rowptr((size_t)n_ + 1, 0), col(es.size()), src(es.size());
  for(size_t q = 0; q < es.size(); ++q) {
    rowptr[(size_t)es[q].r + 1] += 1;
    col[q] = es[q].c;
    src[q] = es[q].t;
  }
  std::partial_sum(rowptr.begin(), rowptr.end(), rowptr.begin());
  nnz_csr_ = (long long)es.size();
  rc = hiopamd_sparse_ldl_create(&ldl_, ctx_, n_, rowptr.data(), col.data());   // HIOPAMD_ERR_STATE: the dense root would exceed the solver's limit
  if(rc != HIOPAMD_OK) return rc;
  rc = hiopamd_alloc((void**)&csr_vals_, sizeof(double) * es.size());
  if(rc == HIOPAMD_OK) rc = hiopamd_alloc((void**)&gather_, sizeof(int) * es.size());
  if(rc == HIOPAMD_OK) rc = hiopamd_copy_h2d(ctx_, gather_, src.data(), sizeof(int) * es.size());
  if(rc != HIOPAMD_OK) {   // all or nothing: a half-built object must not look initialised to the next matrixChanged()
    hiopamd_sparse_ldl_destroy(ldl_);
    ldl_ = nullptr;
    if(csr_vals_) hiopamd_free(csr_vals_);
    if(gather_) hiopamd_free(gather_);
    csr_vals_ = nullptr;
    gather_ = nullptr;
  }
  return rc;
}

int hiopLinSolverSymSparseHipNative::matrixChanged()
{
  assert(M_ && M_->m() == M_->n() && (int)M_->m() == n_);
  if(nlp_) nlp_->runStats.linsolv.tmFactTime.start();
  int rc = HIOPAMD_OK;
  if(pattern_status_ != HIOPAMD_OK) {
    rc = pattern_status_;   // the pattern was rejected once (malformed triplets, root beyond the solver's limit): it will not change — fail fast
  } else if(!ldl_) {
    rc = first_call();
    if(rc == HIOPAMD_ERR_ARG || rc == HIOPAMD_ERR_STATE) {
      pattern_status_ = rc;
      std::fprintf(stderr, "hiop_amd: hiopLinSolverSymSparseHipNative: this sparsity pattern is not accepted (%s); every later matrixChanged() fails at once\n",
                   rc == HIOPAMD_ERR_ARG ? "out-of-range or duplicate triplets" : "the dense root of the elimination tree exceeds the solver's limit");
    }
  }
  if(rc == HIOPAMD_OK) rc = hiopamd_vec_copy_from_indexes(ctx_, nnz_csr_, csr_vals_, static_cast<const hiopMatrixSparse*>(M_)->M(), gather_);
  if(rc == HIOPAMD_OK) rc = hiopamd_sparse_ldl_factorize(ldl_, csr_vals_, &n_neg_, &n_zero_);
  if(nlp_) nlp_->runStats.linsolv.tmFactTime.stop();
  if(rc != HIOPAMD_OK) {
    // not "singular": a pattern this solver does not take (HIOPAMD_ERR_STATE at creation), malformed triplets, or a device failure.
    // Same policy as the dense class: say so, answer -1 (the only other answer the contract has), refuse to solve until a factorisation succeeds.
    ++device_failures_;
    factored_ = false;
    std::fprintf(stderr, "hiop_amd: hiopLinSolverSymSparseHipNative::matrixChanged failed in the device layer (status %d, %d in a row) -- not a singular matrix\n", rc,
                 device_failures_);
    return -1;
  }
  device_failures_ = 0;
  factored_ = n_zero_ == 0;
  return n_zero_ > 0 ? -1 : n_neg_;
}

bool hiopLinSolverSymSparseHipNative::solve(hiopVector& x)
{
  assert((int)x.get_size() == n_);
  if(device_failures_ > 0 || !factored_) return false;
  if(nlp_) nlp_->runStats.linsolv.tmTriuSolves.start();
  const int rc = hiopamd_sparse_ldl_solve(ldl_, x.local_data());
  if(nlp_) nlp_->runStats.linsolv.tmTriuSolves.stop();
  return rc == HIOPAMD_OK;
}

bool hiopLinSolverSymSparseHipNative::compute_inertia(int& pos, int& neg, int& zero) const
{
  if(!ldl_ || device_failures_ > 0) return false;
  neg = n_neg_;
  zero = n_zero_;
  pos = n_ - n_neg_ - n_zero_;
  return true;
}

bool hiopLinSolverSymSparseHipNative::analysis_info(long long info8[8]) const
{
  if(!ldl_) return false;
  int64_t i8[8];
  if(hiopamd_sparse_ldl_info(ldl_, i8) != HIOPAMD_OK) return false;
  for(int q = 0; q < 8; ++q) info8[q] = (long long)i8[q];
  return true;
}
}  // namespace hiop
