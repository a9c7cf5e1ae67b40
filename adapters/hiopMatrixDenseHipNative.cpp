#include "hiopMatrixDenseHipNative.hpp"

#include <cmath>
#include <vector>

namespace hiop
{
namespace
{
const hiopMatrixDenseHipNative& as_dense(const hiopMatrix& M)
{
  return dynamic_cast<const hiopMatrixDenseHipNative&>(M);
}
hiopMatrixDenseHipNative& as_dense(hiopMatrix& M)
{
  return dynamic_cast<hiopMatrixDenseHipNative&>(M);
}
}  // namespace

hiopMatrixDenseHipNative::hiopMatrixDenseHipNative(const size_type& m, const size_type& glob_n, index_type* col_part,
                                                   MPI_Comm comm, const size_type& m_max_alloc)
    : hiopMatrixDense(m, glob_n, comm),
      ctx_(hiopamd_default_ctx()),
      data_(nullptr),
      owns_data_(true),
      myrank_(0),
      comm_size_(1)
{
  int P = 0;
  if(col_part) {
#ifdef HIOP_USE_MPI
    int ierr = MPI_Comm_rank(comm_, &P);
    assert(ierr == MPI_SUCCESS);
    ierr = MPI_Comm_size(comm_, &comm_size_);
    assert(ierr == MPI_SUCCESS);
#endif
    glob_jl_ = col_part[P];
    glob_ju_ = col_part[P + 1];
  } else {
    glob_jl_ = 0;
    glob_ju_ = n_global_;
  }
  n_local_ = glob_ju_ - glob_jl_;
  myrank_ = P;
  max_rows_ = m_max_alloc == -1 ? m_local_ : m_max_alloc;
  assert(max_rows_ >= m_local_);
  data_ = hiopamd_new_array(static_cast<size_t>(max_rows_) * n_local_);
  hiopamd_ok(hiopamd_mat_set_to_constant(ctx_, max_rows_, n_local_, data_, n_local_, 0.0));
}

hiopMatrixDenseHipNative::hiopMatrixDenseHipNative(const hiopMatrixDenseHipNative& o)
    : hiopMatrixDense(o.m_local_, o.n_global_, o.comm_),
      ctx_(o.ctx_),
      data_(nullptr),
      owns_data_(true),
      n_local_(o.n_local_),
      glob_jl_(o.glob_jl_),
      glob_ju_(o.glob_ju_),
      max_rows_(o.max_rows_),
      myrank_(o.myrank_),
      comm_size_(o.comm_size_)
{
  data_ = hiopamd_new_array(static_cast<size_t>(max_rows_) * n_local_);
}

hiopMatrixDenseHipNative::hiopMatrixDenseHipNative(const size_type& m, const size_type& n, double* external)
    : hiopMatrixDense(m, n, MPI_COMM_SELF),
      ctx_(hiopamd_default_ctx()),
      data_(external),
      owns_data_(false),
      n_local_(n),
      glob_jl_(0),
      glob_ju_(n),
      max_rows_(m),
      myrank_(0),
      comm_size_(1)
{
}

hiopMatrixDenseHipNative::~hiopMatrixDenseHipNative()
{
  if(owns_data_) {
    hiopamd_ctx_sync(ctx_);
    hiopamd_free(data_);
  }
}

void hiopMatrixDenseHipNative::allreduce_sum(double* dev_buf, size_type count) const
{
#ifdef HIOP_USE_MPI
  if(comm_size_ > 1 && count > 0) {
    std::vector<double> loc(count), glob(count);
    hiopamd_ok(hiopamd_copy_d2h(ctx_, loc.data(), dev_buf, sizeof(double) * count));
    hiopamd_ok(hiopamd_ctx_sync(ctx_));
    int ierr = MPI_Allreduce(loc.data(), glob.data(), static_cast<int>(count), MPI_DOUBLE, MPI_SUM, comm_);
    assert(ierr == MPI_SUCCESS);
    (void)ierr;
    hiopamd_ok(hiopamd_copy_h2d(ctx_, dev_buf, glob.data(), sizeof(double) * count));
    hiopamd_ok(hiopamd_ctx_sync(ctx_));
  }
#else
  (void)dev_buf;
  (void)count;
#endif
}

void hiopMatrixDenseHipNative::to_host(double* host) const
{
  hiopamd_ok(hiopamd_copy_d2h(ctx_, host, data_, sizeof(double) * static_cast<size_t>(m_local_) * n_local_));
  hiopamd_ok(hiopamd_ctx_sync(ctx_));
}

void hiopMatrixDenseHipNative::setToZero() { setToConstant(0.0); }
void hiopMatrixDenseHipNative::setToConstant(double c)
{
  hiopamd_ok(hiopamd_mat_set_to_constant(ctx_, m_local_, n_local_, data_, n_local_, c));
}

void hiopMatrixDenseHipNative::copyFrom(const hiopMatrixDense& dm)
{
  const auto& o = as_dense(dm);
  assert(n_local_ == o.n_local_ && m_local_ == o.m_local_ && n_global_ == o.n_global_);
  hiopamd_ok(hiopamd_copy_d2d(ctx_, data_, o.data_, sizeof(double) * static_cast<size_t>(m_local_) * n_local_));
}
void hiopMatrixDenseHipNative::copyFrom(const double* buffer)
{
  if(buffer) hiopamd_ok(hiopamd_copy_d2d(ctx_, data_, buffer, sizeof(double) * static_cast<size_t>(m_local_) * n_local_));
}
void hiopMatrixDenseHipNative::copy_to(double* buffer)
{
  if(buffer) hiopamd_ok(hiopamd_copy_d2d(ctx_, buffer, data_, sizeof(double) * static_cast<size_t>(m_local_) * n_local_));
}

// y = beta*y + alpha*this*x ; y replicated, x distributed: beta*y on rank 0 only, then the sum over the ranks
// (hiopMatrixDenseRowMajor.cpp:458-490)
void hiopMatrixDenseHipNative::timesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const
{
  assert(y.get_local_size() == m_local_ && x.get_local_size() == n_local_);
  timesVec(beta, y.local_data(), alpha, x.local_data_const());
}
void hiopMatrixDenseHipNative::timesVec(double beta, double* y, double alpha, const double* x) const
{
  if(m_local_ == 0) return;
  if(comm_size_ > 1 && myrank_ != 0) beta = 0.0;
  hiopamd_ok(hiopamd_mat_times_vec(ctx_, m_local_, n_local_, data_, n_local_, beta, y, alpha, x));
  allreduce_sum(y, m_local_);
}
// y = beta*y + alpha*this^T*x ; y distributed, x replicated (:494-528)
void hiopMatrixDenseHipNative::transTimesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const
{
  assert(x.get_local_size() == m_local_ && y.get_local_size() == n_local_);
  transTimesVec(beta, y.local_data(), alpha, x.local_data_const());
}
void hiopMatrixDenseHipNative::transTimesVec(double beta, double* y, double alpha, const double* x) const
{
  if(n_local_ == 0) return;
  hiopamd_ok(hiopamd_mat_trans_times_vec(ctx_, m_local_, n_local_, data_, n_local_, beta, y, alpha, x));
}

// W = beta*W + alpha*this*X, all three local (:537-612)
void hiopMatrixDenseHipNative::timesMat(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const
{
  assert(as_dense(X).n_local_ == as_dense(X).n_global_ && n_local_ == n_global_ &&
         "'timesMat' involving distributed matrices is not needed/supported");
  timesMat_local(beta, W, alpha, X);
}
void hiopMatrixDenseHipNative::timesMat_local(double beta, hiopMatrix& W_, double alpha, const hiopMatrix& X_) const
{
  auto& W = as_dense(W_);
  const auto& X = as_dense(X_);
  assert(W.m() == m() && X.m() == n() && W.n() == X.n());
  if(W.m() == 0 || X.m() == 0 || W.n() == 0) return;
  hiopamd_ok(hiopamd_mat_times_mat(ctx_, m_local_, n_local_, X.n_local_, data_, n_local_, beta, W.data_, W.n_local_, alpha,
                                   X.data_, X.n_local_));
}
// W = beta*W + alpha*this^T*X ; W and X column-distributed alike, `this` local (:616-643)
void hiopMatrixDenseHipNative::transTimesMat(double beta, hiopMatrix& W_, double alpha, const hiopMatrix& X_) const
{
  auto& W = as_dense(W_);
  const auto& X = as_dense(X_);
  assert(W.m() == n_local_ && X.m() == m_local_ && W.n_local_ == X.n_local_);
  if(W.m() == 0) return;
  hiopamd_ok(hiopamd_mat_trans_times_mat(ctx_, m_local_, n_local_, X.n_local_, data_, n_local_, beta, W.data_, W.n_local_,
                                         alpha, X.data_, X.n_local_));
}
// W = beta*W + alpha*this*X^T ; `this` and X column-distributed, W replicated: beta*W on rank 0 only + all-reduce (:646-700)
void hiopMatrixDenseHipNative::timesMatTrans_local(double beta, hiopMatrix& W_, double alpha, const hiopMatrix& X_) const
{
  auto& W = as_dense(W_);
  const auto& X = as_dense(X_);
  assert(W.n_local_ == W.n_global_ && n_local_ == X.n_local_ && m_local_ == W.m() && X.m_local_ == W.n());
  hiopamd_ok(hiopamd_mat_times_mat_trans(ctx_, m_local_, n_local_, X.m_local_, data_, n_local_, beta, W.data_, W.n_local_,
                                         alpha, X.data_, X.n_local_));
}
void hiopMatrixDenseHipNative::timesMatTrans(double beta, hiopMatrix& W_, double alpha, const hiopMatrix& X) const
{
  auto& W = as_dense(W_);
  if(W.m() == 0 || W.n() == 0) return;
  timesMatTrans_local(myrank_ == 0 ? beta : 0.0, W_, alpha, X);
  allreduce_sum(W.data_, static_cast<size_type>(W.m()) * W.n());
}

void hiopMatrixDenseHipNative::addDiagonal(const double& alpha, const hiopVector& d)
{
  assert(m_local_ == n_local_ && d.get_local_size() == n_local_);
  hiopamd_ok(hiopamd_mat_add_diagonal_vec(ctx_, n_local_, data_, n_local_, alpha, d.local_data_const()));
}
void hiopMatrixDenseHipNative::addDiagonal(const double& value)
{
  hiopamd_ok(hiopamd_mat_add_diagonal_const(ctx_, n_local_, data_, n_local_, value));
}
void hiopMatrixDenseHipNative::addSubDiagonal(const double& alpha, index_type start, const hiopVector& d)
{
  assert(start + d.get_local_size() <= n_local_);
  hiopamd_ok(hiopamd_mat_add_sub_diagonal(ctx_, data_, n_local_, start, alpha, d.local_data_const(), 0, d.get_local_size()));
}
void hiopMatrixDenseHipNative::addSubDiagonal(int start_on_dest_diag, const double& alpha, const hiopVector& d,
                                              int start_on_src_vec, int num_elems)
{
  if(num_elems < 0) num_elems = d.get_local_size() - start_on_src_vec;
  assert(num_elems <= d.get_local_size() && start_on_dest_diag + num_elems <= n_local_);
  hiopamd_ok(hiopamd_mat_add_sub_diagonal(ctx_, data_, n_local_, start_on_dest_diag, alpha, d.local_data_const(),
                                          start_on_src_vec, num_elems));
}
void hiopMatrixDenseHipNative::addSubDiagonal(int start_on_dest_diag, int num_elems, const double& c)
{
  assert(num_elems >= 0 && start_on_dest_diag + num_elems <= n_local_);
  hiopamd_ok(hiopamd_mat_add_sub_diagonal_const(ctx_, data_, n_local_, start_on_dest_diag, num_elems, c));
}
void hiopMatrixDenseHipNative::addMatrix(double alpha, const hiopMatrix& X_)
{
  const auto& X = as_dense(X_);
  assert(m_local_ == X.m_local_ && n_local_ == X.n_local_);
  hiopamd_ok(hiopamd_mat_add_matrix(ctx_, m_local_, n_local_, data_, n_local_, alpha, X.data_, X.n_local_));
}
void hiopMatrixDenseHipNative::transAddToSymDenseMatrixUpperTriangle(int row_start, int col_start, double alpha,
                                                                     hiopMatrixDense& W_) const
{
  auto& W = as_dense(W_);
  assert(row_start >= 0 && n() + row_start <= W.m() && col_start >= 0 && m() + col_start <= W.n() && W.n() == W.m());
  hiopamd_ok(hiopamd_mat_trans_add_to_sym_upper(ctx_, m_local_, n_local_, data_, n_local_, row_start, col_start, alpha,
                                                W.data_, W.n_local_));
}
void hiopMatrixDenseHipNative::addUpperTriangleToSymDenseMatrixUpperTriangle(int diag_start, double alpha,
                                                                             hiopMatrixDense& W_) const
{
  auto& W = as_dense(W_);
  assert(m_local_ == n_local_ && diag_start + n_local_ <= W.n());
  hiopamd_ok(hiopamd_mat_add_upper_to_sym_upper(ctx_, n_local_, data_, n_local_, diag_start, alpha, W.data_, W.n_local_));
}

double hiopMatrixDenseHipNative::max_abs_value()
{
  double v = 0.0;
  hiopamd_ok(hiopamd_mat_max_abs(ctx_, m_local_, n_local_, data_, n_local_, &v));
#ifdef HIOP_USE_MPI
  double g = v;
  int ierr = MPI_Allreduce(&v, &g, 1, MPI_DOUBLE, MPI_MAX, comm_);
  assert(ierr == MPI_SUCCESS);
  (void)ierr;
  v = g;
#endif
  return v;
}
void hiopMatrixDenseHipNative::row_max_abs_value(hiopVector& ret_vec)
{
  assert(ret_vec.get_local_size() == m_local_);
  hiopamd_ok(hiopamd_mat_row_max_abs(ctx_, m_local_, n_local_, data_, n_local_, ret_vec.local_data()));
#ifdef HIOP_USE_MPI
  if(comm_size_ > 1) {
    std::vector<double> loc(m_local_), glob(m_local_);
    hiopamd_ok(hiopamd_copy_d2h(ctx_, loc.data(), ret_vec.local_data(), sizeof(double) * m_local_));
    hiopamd_ok(hiopamd_ctx_sync(ctx_));
    int ierr = MPI_Allreduce(loc.data(), glob.data(), m_local_, MPI_DOUBLE, MPI_MAX, comm_);
    assert(ierr == MPI_SUCCESS);
    (void)ierr;
    hiopamd_ok(hiopamd_copy_h2d(ctx_, ret_vec.local_data(), glob.data(), sizeof(double) * m_local_));
    hiopamd_ok(hiopamd_ctx_sync(ctx_));
  }
#endif
}
void hiopMatrixDenseHipNative::scale_row(hiopVector& vec_scal, const bool inv_scale)
{
  assert(vec_scal.get_local_size() == m_local_);
  hiopamd_ok(hiopamd_mat_scale_rows(ctx_, m_local_, n_local_, data_, n_local_, vec_scal.local_data_const(), inv_scale ? 1 : 0));
}
bool hiopMatrixDenseHipNative::isfinite() const
{
  int ok = 0;
  hiopamd_ok(hiopamd_mat_is_finite(ctx_, m_local_, n_local_, data_, n_local_, &ok));
  return ok != 0;
}

void hiopMatrixDenseHipNative::print(FILE* f, const char* msg, int maxRows, int maxCols, int rank) const
{
  if(rank != -1 && rank != myrank_) return;
  if(!f) f = stdout;
  if(maxRows > m_local_ || maxRows < 0) maxRows = m_local_;
  if(maxCols > n_local_ || maxCols < 0) maxCols = n_local_;
  std::vector<double> h(static_cast<size_t>(m_local_) * n_local_ + 1);
  to_host(h.data());
  if(msg)
    std::fprintf(f, "%s (local_dims=[%d,%d])\n", msg, (int)m_local_, (int)n_local_);
  else
    std::fprintf(f, "hiopMatrixDenseHipNative::printing max=[%d,%d] (local_dims=[%d,%d], on rank=%d)\n", maxRows, maxCols,
                 (int)m_local_, (int)n_local_, myrank_);
  for(int i = 0; i < maxRows; i++) {
    std::fprintf(f, i == 0 ? "[" : " ");
    for(int j = 0; j < maxCols; j++) std::fprintf(f, "%20.12e ", h[static_cast<size_t>(i) * n_local_ + j]);
    std::fprintf(f, i < maxRows - 1 ? "; ...\n" : "];\n");
  }
}

hiopMatrixDense* hiopMatrixDenseHipNative::alloc_clone() const
{
  auto* c = new hiopMatrixDenseHipNative(*this);
  hiopamd_ok(hiopamd_mat_set_to_constant(ctx_, c->max_rows_, c->n_local_, c->data_, c->n_local_, 0.0));
  return c;
}
hiopMatrixDense* hiopMatrixDenseHipNative::new_copy() const
{
  auto* c = new hiopMatrixDenseHipNative(*this);
  hiopamd_ok(hiopamd_copy_d2d(ctx_, c->data_, data_, sizeof(double) * static_cast<size_t>(max_rows_) * n_local_));
  return c;
}

void hiopMatrixDenseHipNative::appendRow(const hiopVector& rowv)
{
  assert(rowv.get_local_size() == n_local_ && m_local_ < max_rows_ && "no more space to append rows");
  hiopamd_ok(hiopamd_copy_d2d(ctx_, row(m_local_), rowv.local_data_const(), sizeof(double) * n_local_));
  m_local_++;
}
void hiopMatrixDenseHipNative::copyRowsFrom(const hiopMatrixDense& src_, int num_rows, int row_dest)
{
  const auto& src = as_dense(src_);
  assert(row_dest >= 0 && n_global_ == src.n_global_ && n_local_ == src.n_local_ && row_dest + num_rows <= m_local_ &&
         num_rows <= src.m_local_);
  if(num_rows > 0)
    hiopamd_ok(hiopamd_mat_copy_rows_from(ctx_, num_rows, n_local_, data_, n_local_, row_dest, src.data_, src.n_local_));
}
void hiopMatrixDenseHipNative::copyRowsFrom(const hiopMatrix& src_gen, const index_type* rows_idxs, size_type n_rows)
{
  // rows_idxs: DEVICE array of n_rows row indexes into src (same memory space as the matrix, like hiopMatrixRajaDense)
  const auto& src = as_dense(src_gen);
  assert(n_global_ == src.n_global_ && n_local_ == src.n_local_ && n_rows <= src.m_local_ && n_rows == m_local_);
  hiopamd_ok(hiopamd_mat_copy_rows_from_idx(ctx_, n_rows, n_local_, data_, n_local_, src.data_, src.n_local_, rows_idxs));
}
void hiopMatrixDenseHipNative::copyBlockFromMatrix(const index_type i_start, const index_type j_start,
                                                   const hiopMatrixDense& src_)
{
  const auto& src = as_dense(src_);
  assert(n_local_ == n_global_ && "this method should be used only in 'serial' mode");
  assert(src.n_local_ == src.n_global_ && m_local_ >= i_start + src.m_local_ && n_local_ >= j_start + src.n_local_);
  hiopamd_ok(hiopamd_mat_copy_block(ctx_, src.m_local_, src.n_local_, row(i_start) + j_start, n_local_, src.data_, src.n_local_));
}
void hiopMatrixDenseHipNative::copyFromMatrixBlock(const hiopMatrixDense& src_, const int i_block, const int j_block)
{
  const auto& src = as_dense(src_);
  assert(n_local_ == n_global_ && src.n_local_ == src.n_global_ && m_local_ + i_block <= src.m_local_ &&
         n_local_ + j_block <= src.n_local_);
  hiopamd_ok(hiopamd_mat_copy_block(ctx_, m_local_, n_local_, data_, n_local_, src.row(i_block) + j_block, src.n_local_));
}
void hiopMatrixDenseHipNative::shiftRows(size_type shift)
{
  if(shift == 0) return;
  assert(std::abs(static_cast<long>(shift)) < m_local_);
  if(m_local_ <= 1) return;
  hiopamd_ok(hiopamd_mat_shift_rows(ctx_, m_local_, n_local_, data_, n_local_, static_cast<int>(shift)));
}
void hiopMatrixDenseHipNative::replaceRow(index_type r, const hiopVector& vec)
{
  assert(r >= 0 && r < m_local_ && vec.get_local_size() == n_local_);
  hiopamd_ok(hiopamd_copy_d2d(ctx_, row(r), vec.local_data_const(), sizeof(double) * n_local_));
}
void hiopMatrixDenseHipNative::getRow(index_type irow, hiopVector& row_vec)
{
  assert(irow >= 0 && irow < m_local_ && row_vec.get_local_size() == n_local_);
  hiopamd_ok(hiopamd_copy_d2d(ctx_, row_vec.local_data(), row(irow), sizeof(double) * n_local_));
}
void hiopMatrixDenseHipNative::set_Hess_FR(const hiopMatrixDense& Hess, const hiopVector& add_diag_de)
{
  copyFrom(Hess);
  addDiagonal(1.0, add_diag_de);
}
// [Jc 0 .. -I I 0 0; Jd 0 .. 0 0 -I I] on the last rank, plain row copies elsewhere (hiopMatrixDenseRowMajor.cpp:301-338)
void hiopMatrixDenseHipNative::set_Jac_FR(const hiopMatrixDense& Jac_c, const hiopMatrixDense& Jac_d)
{
  const auto& Jeq = as_dense(Jac_c);
  const auto& Jin = as_dense(Jac_d);
  assert(Jeq.n() == Jin.n() && Jeq.n_local_ == Jin.n_local_ && Jeq.m() + Jin.m() == m() && Jeq.n() <= n());
  setToZero();
  const int me = Jeq.m_local_, mi = Jin.m_local_, nb = Jeq.n_local_;
  if(me) hiopamd_ok(hiopamd_mat_copy_block(ctx_, me, nb, data_, n_local_, Jeq.data_, Jeq.n_local_));
  if(mi) hiopamd_ok(hiopamd_mat_copy_block(ctx_, mi, nb, row(me), n_local_, Jin.data_, Jin.n_local_));
  if(myrank_ == comm_size_ - 1) {
    assert(nb + 2 * me + 2 * mi == n_local_);
    // the four +-identity blocks are sub-diagonals of the matrix seen from a shifted origin
    if(me) {
      hiopamd_ok(hiopamd_mat_add_sub_diagonal_const(ctx_, data_ + nb, n_local_, 0, me, -1.0));
      hiopamd_ok(hiopamd_mat_add_sub_diagonal_const(ctx_, data_ + nb + me, n_local_, 0, me, 1.0));
    }
    if(mi) {
      hiopamd_ok(hiopamd_mat_add_sub_diagonal_const(ctx_, row(me) + nb + 2 * me, n_local_, 0, mi, -1.0));
      hiopamd_ok(hiopamd_mat_add_sub_diagonal_const(ctx_, row(me) + nb + 2 * me + mi, n_local_, 0, mi, 1.0));
    }
  }
}

#ifdef HIOP_DEEPCHECKS
// HIOP_DEEPCHECKS-only helpers of the reference (:341-353, :886-909): host round trips, not a hot path
void hiopMatrixDenseHipNative::overwriteUpperTriangleWithLower()
{
  assert(n_local_ == n_global_ && "Use only with local, non-distributed matrices");
  std::vector<double> h(static_cast<size_t>(m_local_) * n_local_ + 1);
  to_host(h.data());
  for(int i = 0; i < m_local_; i++)
    for(int j = i + 1; j < n_local_ && j < m_local_; j++) h[static_cast<size_t>(i) * n_local_ + j] = h[static_cast<size_t>(j) * n_local_ + i];
  hiopamd_ok(hiopamd_copy_h2d(ctx_, data_, h.data(), sizeof(double) * static_cast<size_t>(m_local_) * n_local_));
  hiopamd_ok(hiopamd_ctx_sync(ctx_));
}
void hiopMatrixDenseHipNative::overwriteLowerTriangleWithUpper()
{
  assert(n_local_ == n_global_ && m_local_ == n_local_ && "Use only with local, non-distributed square matrices");
  hiopamd_ok(hiopamd_mat_symmetrize(ctx_, n_local_, data_, n_local_));
}
bool hiopMatrixDenseHipNative::assertSymmetry(double tol) const
{
  if(n_local_ != n_global_ || m_local_ != n_global_) {
    assert(false && "should be used only for local square matrices");
    return false;
  }
  std::vector<double> h(static_cast<size_t>(m_local_) * n_local_ + 1);
  to_host(h.data());
  for(int i = 0; i < n_local_; i++)
    for(int j = 0; j < n_local_; j++) {
      const double ij = h[static_cast<size_t>(i) * n_local_ + j], ji = h[static_cast<size_t>(j) * n_local_ + i];
      if(std::abs(ij - ji) / (1 + std::abs(ij)) >= tol) return false;
    }
  return true;
}
#endif  // HIOP_DEEPCHECKS
bool hiopMatrixDenseHipNative::symmetrize()
{
  if(n_local_ != n_global_ || m_local_ != n_global_) {
    assert(false && "should be used only for local square matrices");
    return false;
  }
  hiopamd_ok(hiopamd_mat_symmetrize(ctx_, n_local_, data_, n_local_));
  return true;
}
}  // namespace hiop
