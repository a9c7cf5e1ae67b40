#include "hiopMatrixSparseTripletHipNative.hpp"

#include <vector>

namespace hiop
{
namespace
{
const hiopMatrixSparseTripletHipNative& as_triplet(const hiopMatrix& M)
{
  return dynamic_cast<const hiopMatrixSparseTripletHipNative&>(M);
}
hiopMatrixDenseHipNative& as_dense(hiopMatrix& M) { return dynamic_cast<hiopMatrixDenseHipNative&>(M); }
}  // namespace

hiopMatrixSparseTripletHipNative::hiopMatrixSparseTripletHipNative(int rows, int cols, int nnz)
    : hiopMatrixSparse(rows, cols, nnz),
      ctx_(hiopamd_default_ctx()),
      iRow_(hiopamd_new_int_array(nnz)),
      jCol_(hiopamd_new_int_array(nnz)),
      values_(hiopamd_new_array(nnz)),
      ones_(nullptr)
{
}
hiopMatrixSparseTripletHipNative::~hiopMatrixSparseTripletHipNative()
{
  hiopamd_ctx_sync(ctx_);
  structure_changed();
  hiopamd_free(iRow_);
  hiopamd_free(jCol_);
  hiopamd_free(values_);
  if(ones_) hiopamd_free(ones_);
}
void hiopMatrixSparseTripletHipNative::structure_changed() const
{
  for(auto& kv : plans_) hiopamd_sp_plan_destroy(kv.second);
  plans_.clear();
}

void hiopMatrixSparseTripletHipNative::setToZero() { setToConstant(0.0); }
void hiopMatrixSparseTripletHipNative::setToConstant(double c)
{
  hiopamd_ok(hiopamd_vec_set_to_constant(ctx_, nnz_, values_, c));
}
void hiopMatrixSparseTripletHipNative::copy_to(int* irow, int* jcol, double* val)
{
  assert(irow && jcol && val);   // device destinations (same memory space as the matrix)
  hiopamd_ok(hiopamd_copy_d2d(ctx_, irow, iRow_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2d(ctx_, jcol, jCol_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2d(ctx_, val, values_, sizeof(double) * nnz_));
}
void hiopMatrixSparseTripletHipNative::copy_to(hiopMatrixDense& W)
{
  assert(W.m() == nrows_ && W.n() == ncols_);
  hiopamd_ok(hiopamd_sp_copy_to_dense(ctx_, nrows_, ncols_, nnz_, iRow_, jCol_, values_, W.local_data(), W.get_local_size_n()));
}

void hiopMatrixSparseTripletHipNative::timesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const
{
  assert(x.get_size() == ncols_ && y.get_size() == nrows_);
  timesVec(beta, y.local_data(), alpha, x.local_data_const());
}
void hiopMatrixSparseTripletHipNative::timesVec(double beta, double* y, double alpha, const double* x) const
{
  hiopamd_ok(hiopamd_sp_times_vec(ctx_, nrows_, ncols_, nnz_, iRow_, jCol_, values_, beta, y, alpha, x));
}
void hiopMatrixSparseTripletHipNative::transTimesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const
{
  assert(x.get_size() == nrows_ && y.get_size() == ncols_);
  transTimesVec(beta, y.local_data(), alpha, x.local_data_const());
}
void hiopMatrixSparseTripletHipNative::transTimesVec(double beta, double* y, double alpha, const double* x) const
{
  hiopamd_ok(hiopamd_sp_trans_times_vec(ctx_, nrows_, ncols_, nnz_, iRow_, jCol_, values_, beta, y, alpha, x));
}

hiopamd_sp_plan* hiopMatrixSparseTripletHipNative::plan_with(const hiopMatrixSparseTripletHipNative& N, bool same_upper) const
{
  const void* key = same_upper ? static_cast<const void*>(this) : static_cast<const void*>(&N);
  auto it = plans_.find(key);
  if(it != plans_.end()) return it->second;
  // the symbolic phase runs once per sparsity pattern, on the host (the pattern is fixed over the IPM iterations:
  // hiopMatrixSparseTriplet.cpp:479-489)
  std::vector<int> i1(nnz_ + 1), j1(nnz_ + 1), i2(N.nnz_ + 1), j2(N.nnz_ + 1);
  hiopamd_ok(hiopamd_copy_d2h(ctx_, i1.data(), iRow_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2h(ctx_, j1.data(), jCol_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2h(ctx_, i2.data(), N.iRow_, sizeof(int) * N.nnz_));
  hiopamd_ok(hiopamd_copy_d2h(ctx_, j2.data(), N.jCol_, sizeof(int) * N.nnz_));
  hiopamd_ok(hiopamd_ctx_sync(ctx_));
  hiopamd_sp_plan* plan = nullptr;
  hiopamd_ok(hiopamd_sp_plan_create(&plan, nrows_, N.nrows_, ncols_, nnz_, i1.data(), j1.data(), N.nnz_, i2.data(), j2.data(),
                                    same_upper ? 1 : 0));
  plans_[key] = plan;
  return plan;
}

// W = beta*W + alpha*this*X^T, X sparse, W dense (:144-201)
void hiopMatrixSparseTripletHipNative::timesMatTrans(double beta, hiopMatrix& Wmat, double alpha, const hiopMatrix& Xmat) const
{
  auto& W = as_dense(Wmat);
  const auto& X = as_triplet(Xmat);
  assert(ncols_ == X.ncols_ && nrows_ == W.m() && X.nrows_ == W.n());
  if(!ones_) {
    ones_ = hiopamd_new_array(ncols_);
    hiopamd_ok(hiopamd_vec_set_to_constant(ctx_, ncols_, ones_, 1.0));
  }
  if(beta != 1.0) hiopamd_ok(hiopamd_vec_scale(ctx_, static_cast<int64_t>(W.m()) * W.get_local_size_n(), W.local_data(), beta));
  hiopamd_ok(hiopamd_sp_add_MDinvNt(ctx_, plan_with(X, false), values_, X.values_, ones_, alpha, W.local_data(),
                                    W.get_local_size_n(), 0, 0));
}
void hiopMatrixSparseTripletHipNative::transAddToSymDenseMatrixUpperTriangle(int row_start, int col_start, double alpha,
                                                                             hiopMatrixDense& W) const
{
  assert(row_start >= 0 && row_start + ncols_ <= W.m() && col_start >= 0 && col_start + nrows_ <= W.n() && W.n() == W.m());
  hiopamd_ok(hiopamd_sp_trans_add_to_sym_upper(ctx_, nnz_, iRow_, jCol_, values_, row_start, col_start, alpha, W.local_data(),
                                               W.get_local_size_n()));
}
void hiopMatrixSparseTripletHipNative::addUpperTriangleToSymDenseMatrixUpperTriangle(int, double, hiopMatrixDense&) const
{
  assert(false && "counterpart method of hiopMatrixSymSparseTripletHipNative should be used");
}
// W(diag block) += alpha * this * D^-1 * this^T, upper triangle (:390-445)
void hiopMatrixSparseTripletHipNative::addMDinvMtransToDiagBlockOfSymDeMatUTri(int start, const double& alpha,
                                                                                const hiopVector& D, hiopMatrixDense& W) const
{
  assert(start >= 0 && start + nrows_ <= W.m() && D.get_size() == ncols_);
  hiopamd_ok(hiopamd_sp_add_MDinvNt(ctx_, plan_with(*this, true), values_, values_, D.local_data_const(), alpha, W.local_data(),
                                    W.get_local_size_n(), start, start));
}
// W(block) += alpha * this * D^-1 * N^T (:447-527)
void hiopMatrixSparseTripletHipNative::addMDinvNtransToSymDeMatUTri(int row_start, int col_start, const double& alpha,
                                                                     const hiopVector& D, const hiopMatrixSparse& N_,
                                                                     hiopMatrixDense& W) const
{
  const auto& N = as_triplet(N_);
  assert(ncols_ == N.ncols_ && D.get_size() == ncols_ && row_start >= 0 && row_start + nrows_ <= W.m() && col_start >= 0 &&
         col_start + N.nrows_ <= W.n());
  hiopamd_ok(hiopamd_sp_add_MDinvNt(ctx_, plan_with(N, false), values_, N.values_, D.local_data_const(), alpha, W.local_data(),
                                    W.get_local_size_n(), row_start, col_start));
}

double hiopMatrixSparseTripletHipNative::max_abs_value()
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_infnorm(ctx_, nnz_, values_, &v));
  return v;
}
void hiopMatrixSparseTripletHipNative::row_max_abs_value(hiopVector& ret_vec)
{
  assert(ret_vec.get_local_size() == nrows_);
  hiopamd_ok(hiopamd_sp_row_max_abs(ctx_, nrows_, nnz_, iRow_, values_, ret_vec.local_data()));
}
void hiopMatrixSparseTripletHipNative::scale_row(hiopVector& vec_scal, const bool inv_scale)
{
  assert(vec_scal.get_local_size() == nrows_);
  hiopamd_ok(hiopamd_sp_scale_rows(ctx_, nnz_, iRow_, values_, vec_scal.local_data_const(), inv_scale ? 1 : 0));
}
bool hiopMatrixSparseTripletHipNative::isfinite() const
{
  int ok = 0;
  hiopamd_ok(hiopamd_vec_isfinite(ctx_, nnz_, values_, &ok));
  return ok != 0;
}
void hiopMatrixSparseTripletHipNative::print(FILE* file, const char* msg, int maxRows, int, int) const
{
  if(!file) file = stdout;
  const int max_elems = maxRows >= 0 ? (maxRows < nnz_ ? maxRows : nnz_) : nnz_;
  std::vector<int> hi(nnz_ + 1), hj(nnz_ + 1);
  std::vector<double> hv(nnz_ + 1);
  hiopamd_ok(hiopamd_copy_d2h(ctx_, hi.data(), iRow_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2h(ctx_, hj.data(), jCol_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2h(ctx_, hv.data(), values_, sizeof(double) * nnz_));
  hiopamd_ok(hiopamd_ctx_sync(ctx_));
  if(msg)
    std::fprintf(file, "%s ", msg);
  else
    std::fprintf(file, "matrix of size %d %d and nonzeros %d, printing %d elems\n", (int)nrows_, (int)ncols_, (int)nnz_, max_elems);
  std::fprintf(file, "iRow=[");
  for(int it = 0; it < max_elems; it++) std::fprintf(file, "%d; ", hi[it]);
  std::fprintf(file, "];\njCol=[");
  for(int it = 0; it < max_elems; it++) std::fprintf(file, "%d; ", hj[it]);
  std::fprintf(file, "];\nv=[");
  for(int it = 0; it < max_elems; it++) std::fprintf(file, "%22.16e; ", hv[it]);
  std::fprintf(file, "];\n");
}
void hiopMatrixSparseTripletHipNative::startingAtAddSubDiagonalToStartingAt(int, const double&, hiopVector&, int, int) const
{
  assert(false && "counterpart method of hiopMatrixSymSparseTripletHipNative should be used");
}
hiopMatrixSparse* hiopMatrixSparseTripletHipNative::alloc_clone() const
{
  return new hiopMatrixSparseTripletHipNative(nrows_, ncols_, nnz_);
}
hiopMatrixSparse* hiopMatrixSparseTripletHipNative::new_copy() const
{
  auto* c = new hiopMatrixSparseTripletHipNative(nrows_, ncols_, nnz_);
  hiopamd_ok(hiopamd_copy_d2d(ctx_, c->iRow_, iRow_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2d(ctx_, c->jCol_, jCol_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2d(ctx_, c->values_, values_, sizeof(double) * nnz_));
  return c;
}
size_type hiopMatrixSparseTripletHipNative::numberOfOffDiagNonzeros() const
{
  assert(false && "not needed (reference: hiopMatrixSparseTriplet.hpp, general triplets)");
  return 0;
}
bool hiopMatrixSparseTripletHipNative::is_diagonal() const
{
  int64_t off = 0;
  hiopamd_ok(hiopamd_sp_num_offdiag(ctx_, nnz_, iRow_, jCol_, &off));
  return off == 0;
}
void hiopMatrixSparseTripletHipNative::extract_diagonal(hiopVector& diag_out) const
{
  assert(diag_out.get_local_size() == nrows_);
  hiopamd_ok(hiopamd_sp_extract_diagonal(ctx_, nrows_, nnz_, iRow_, jCol_, values_, diag_out.local_data()));
}
bool hiopMatrixSparseTripletHipNative::checkIndexesAreOrdered() const
{
  int ok = 0;
  hiopamd_ok(hiopamd_sp_indexes_ordered(ctx_, nnz_, iRow_, jCol_, &ok));
  return ok != 0;
}

// ---- (2) ----
void hiopMatrixSparseTripletHipNative::copyFrom(const hiopMatrixSparse&)
{
  assert(false && "this is to be implemented - method def too vague for now (reference :348)");
}
void hiopMatrixSparseTripletHipNative::timesMat(double, hiopMatrix&, double, const hiopMatrix&) const { assert(false && "not needed"); }
void hiopMatrixSparseTripletHipNative::transTimesMat(double, hiopMatrix&, double, const hiopMatrix&) const
{
  assert(false && "not needed");
}
void hiopMatrixSparseTripletHipNative::addDiagonal(const double&, const hiopVector&) { assert(false && "not needed"); }
void hiopMatrixSparseTripletHipNative::addDiagonal(const double&) { assert(false && "not needed"); }
void hiopMatrixSparseTripletHipNative::addSubDiagonal(const double&, index_type, const hiopVector&) { assert(false && "not needed"); }
void hiopMatrixSparseTripletHipNative::addSubDiagonal(int, const double&, const hiopVector&, int, int)
{
  assert(false && "not needed / implemented");
}
void hiopMatrixSparseTripletHipNative::addSubDiagonal(int, int, const double&) { assert(false && "not needed / implemented"); }
void hiopMatrixSparseTripletHipNative::addMatrix(double, const hiopMatrix&) { assert(false && "not needed"); }

// ---- (3) the assembly surface (hiopMatrixSparseTriplet.cpp:216-250, :562-720, :790-922, :1042-1172): one C-ABI call each;
// index arrays the caller hands in (rows_idxs, iJacS / jJacS / MJacS) are device arrays like every array of this mem-space ----
void hiopMatrixSparseTripletHipNative::copyRowsFrom(const hiopMatrix& src_gen, const index_type* rows_idxs, size_type n_rows)
{
  const auto& src = as_triplet(src_gen);
  structure_changed();
  hiopamd_ok(hiopamd_sp_copy_rows_from(ctx_, iRow_, jCol_, values_, src.numberOfNonzeros(), src.i_row(), src.j_col(), src.M(), rows_idxs, n_rows));
}
void hiopMatrixSparseTripletHipNative::copySubDiagonalFrom(const index_type& start_on_dest_diag, const size_type& num_elems,
                                                           const hiopVector& d_, const index_type& start_on_nnz_idx, double scal)
{
  structure_changed();
  hiopamd_ok(hiopamd_sp_copy_sub_diagonal_from(ctx_, iRow_, jCol_, values_, start_on_dest_diag, num_elems, d_.local_data_const(), start_on_nnz_idx, scal));
}
void hiopMatrixSparseTripletHipNative::setSubDiagonalTo(const index_type& start_on_dest_diag, const size_type& num_elems,
                                                        const double& c, const index_type& start_on_nnz_idx)
{
  structure_changed();
  hiopamd_ok(hiopamd_sp_set_sub_diagonal_to(ctx_, iRow_, jCol_, values_, start_on_dest_diag, num_elems, c, start_on_nnz_idx));
}
void hiopMatrixSparseTripletHipNative::copyRowsBlockFrom(const hiopMatrix& src_gen, const index_type& rows_src_idx_st,
                                                         const size_type& n_rows, const index_type& rows_dest_idx_st,
                                                         const size_type& dest_nnz_st)
{
  const auto& src = as_triplet(src_gen);
  structure_changed();
  hiopamd_ok(hiopamd_sp_copy_rows_block_from(ctx_, iRow_, jCol_, values_, src.numberOfNonzeros(), src.i_row(), src.j_col(), src.M(), rows_src_idx_st, n_rows, rows_dest_idx_st, dest_nnz_st));
}
void hiopMatrixSparseTripletHipNative::copySubmatrixFrom(const hiopMatrix& src_gen, const index_type& dest_row_st,
                                                         const index_type& dest_col_st, const size_type& dest_nnz_st,
                                                         const bool offdiag_only)
{
  const auto& src = as_triplet(src_gen);
  structure_changed();
  hiopamd_ok(hiopamd_sp_copy_submatrix_from(ctx_, iRow_, jCol_, values_, src.numberOfNonzeros(), src.i_row(), src.j_col(), src.M(), dest_row_st, dest_col_st, dest_nnz_st, offdiag_only ? 1 : 0, 0));
}
void hiopMatrixSparseTripletHipNative::copySubmatrixFromTrans(const hiopMatrix& src_gen, const index_type& dest_row_st,
                                                              const index_type& dest_col_st, const size_type& dest_nnz_st,
                                                              const bool offdiag_only)
{
  const auto& src = as_triplet(src_gen);
  structure_changed();
  hiopamd_ok(hiopamd_sp_copy_submatrix_from(ctx_, iRow_, jCol_, values_, src.numberOfNonzeros(), src.i_row(), src.j_col(), src.M(), dest_row_st, dest_col_st, dest_nnz_st, offdiag_only ? 1 : 0, 1));
}
void hiopMatrixSparseTripletHipNative::setSubmatrixToConstantDiag_w_colpattern(const double& scalar, const index_type& dest_row_st,
                                                                               const index_type& dest_col_st, const size_type& dest_nnz_st,
                                                                               const size_type& nnz_to_copy, const hiopVector& ix)
{
  (void)nnz_to_copy;
  structure_changed();
  hiopamd_ok(hiopamd_sp_set_submatrix_to_constant_diag_w_pattern(ctx_, iRow_, jCol_, values_, scalar, dest_row_st, dest_col_st, dest_nnz_st, ix.get_local_size(), ix.local_data_const(), 0, nullptr));
}
void hiopMatrixSparseTripletHipNative::setSubmatrixToConstantDiag_w_rowpattern(const double& scalar, const index_type& dest_row_st,
                                                                               const index_type& dest_col_st, const size_type& dest_nnz_st,
                                                                               const size_type& nnz_to_copy, const hiopVector& ix)
{
  (void)nnz_to_copy;
  structure_changed();
  hiopamd_ok(hiopamd_sp_set_submatrix_to_constant_diag_w_pattern(ctx_, iRow_, jCol_, values_, scalar, dest_row_st, dest_col_st, dest_nnz_st, ix.get_local_size(), ix.local_data_const(), 1, nullptr));
}
void hiopMatrixSparseTripletHipNative::copyDiagMatrixToSubblock(const double& src_val, const index_type& dest_row_st,
                                                                const index_type& dest_col_st, const size_type& dest_nnz_st,
                                                                const size_type& nnz_to_copy)
{
  structure_changed();
  hiopamd_ok(hiopamd_sp_copy_diag_matrix_to_subblock(ctx_, iRow_, jCol_, values_, src_val, dest_row_st, dest_col_st, dest_nnz_st, nnz_to_copy));
}
void hiopMatrixSparseTripletHipNative::copyDiagMatrixToSubblock_w_pattern(const hiopVector& x, const index_type& dest_row_st,
                                                                          const index_type& dest_col_st, const size_type& dest_nnz_st,
                                                                          const size_type& nnz_to_copy, const hiopVector& pattern)
{
  (void)nnz_to_copy;
  structure_changed();
  hiopamd_ok(hiopamd_sp_copy_diag_matrix_to_subblock_w_pattern(ctx_, iRow_, jCol_, values_, x.local_data_const(), dest_row_st, dest_col_st, dest_nnz_st, pattern.get_local_size(), pattern.local_data_const(), nullptr));
}
void hiopMatrixSparseTripletHipNative::set_Jac_FR(const hiopMatrixSparse& Jac_c, const hiopMatrixSparse& Jac_d, int* iJacS, int* jJacS,
                                                  double* MJacS)
{
  structure_changed();
  hiopamd_ok(hiopamd_sp_set_jac_fr(ctx_, iRow_, jCol_, values_, Jac_c.n(), Jac_c.m(), Jac_c.numberOfNonzeros(), Jac_c.i_row(), Jac_c.j_col(), Jac_c.M(), Jac_d.m(), Jac_d.numberOfNonzeros(), Jac_d.i_row(), Jac_d.j_col(), Jac_d.M(), iJacS, jJacS, MJacS));
}
void hiopMatrixSparseTripletHipNative::set_Hess_FR(const hiopMatrixSparse& Hess, int* iHSS, int* jHSS, double* MHSS,
                                                   const hiopVector& add_diag)
{
  // (the symmetric class owns the meaning of this method in the reference, hiopMatrixSparseTriplet.cpp:1374; the general
  // triplet class only declares it, hiopMatrixSparseTriplet.hpp:247-255)
  structure_changed();
  hiopamd_ok(hiopamd_spsym_set_hess_fr(ctx_, iRow_, jCol_, values_, Hess.m(), Hess.numberOfNonzeros(), Hess.i_row(), Hess.j_col(), Hess.M(), add_diag.get_local_size(), add_diag.local_data_const(), iHSS, jHSS, MHSS));
}

// ---------------------------------------------------------------------------------------------------------------
// symmetric: only the upper triangle is stored
// ---------------------------------------------------------------------------------------------------------------
void hiopMatrixSymSparseTripletHipNative::timesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const
{
  assert(ncols_ == nrows_ && x.get_size() == ncols_ && y.get_size() == nrows_);
  timesVec(beta, y.local_data(), alpha, x.local_data_const());
}
void hiopMatrixSymSparseTripletHipNative::timesVec(double beta, double* y, double alpha, const double* x) const
{
  hiopamd_ok(hiopamd_spsym_times_vec(ctx_, nrows_, nnz_, iRow_, jCol_, values_, beta, y, alpha, x));
}
void hiopMatrixSymSparseTripletHipNative::transAddToSymDenseMatrixUpperTriangle(int, int, double, hiopMatrixDense&) const
{
  assert(false && "not yet implemented (reference: hiopMatrixSparseTriplet.hpp:391-393)");
}
void hiopMatrixSymSparseTripletHipNative::addUpperTriangleToSymDenseMatrixUpperTriangle(int diag_start, double alpha,
                                                                                        hiopMatrixDense& W) const
{
  assert(diag_start >= 0 && diag_start + nrows_ <= W.m() && diag_start + ncols_ <= W.n() && W.n() == W.m());
  hiopamd_ok(hiopamd_spsym_add_upper_to_sym_upper(ctx_, nnz_, iRow_, jCol_, values_, diag_start, alpha, W.local_data(),
                                                  W.get_local_size_n()));
}
void hiopMatrixSymSparseTripletHipNative::startingAtAddSubDiagonalToStartingAt(int diag_src_start, const double& alpha,
                                                                               hiopVector& vec_dest, int vec_start,
                                                                               int num_elems) const
{
  hiopamd_ok(hiopamd_spsym_add_diag_to_vec(ctx_, nnz_, iRow_, jCol_, values_, alpha, vec_dest.local_data(), vec_start,
                                           vec_dest.get_local_size(), diag_src_start, num_elems));
}
hiopMatrixSparse* hiopMatrixSymSparseTripletHipNative::alloc_clone() const
{
  assert(nrows_ == ncols_);
  return new hiopMatrixSymSparseTripletHipNative(nrows_, nnz_);
}
hiopMatrixSparse* hiopMatrixSymSparseTripletHipNative::new_copy() const
{
  auto* c = new hiopMatrixSymSparseTripletHipNative(nrows_, nnz_);
  hiopamd_ok(hiopamd_copy_d2d(ctx_, c->i_row(), iRow_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2d(ctx_, c->j_col(), jCol_, sizeof(int) * nnz_));
  hiopamd_ok(hiopamd_copy_d2d(ctx_, c->M(), values_, sizeof(double) * nnz_));
  return c;
}
size_type hiopMatrixSymSparseTripletHipNative::numberOfOffDiagNonzeros() const
{
  int64_t off = 0;
  hiopamd_ok(hiopamd_sp_num_offdiag(ctx_, nnz_, iRow_, jCol_, &off));
  return static_cast<size_type>(off);
}
}  // namespace hiop
