// hiopMatrixDense on libhiopamd.so (MI355X native), row-major with the columns distributed exactly like
// hiopMatrixDenseRowMajor (src/LinAlg/hiopMatrixDenseRowMajor.hpp).  Every virtual of src/LinAlg/hiopMatrix.hpp:67-219 and
// src/LinAlg/hiopMatrixDense.hpp:72-253 is overridden (the base class' bodies are `assert(false)`), forwarding to the
// hiopamd_mat_* entry points that cite the same reference lines in include/hiop_amd.h.
// Raw `double*` arguments are DEVICE pointers, as in the reference's device back-ends (hiopMatrixRajaDense).
#pragma once
#include "hiopMatrixDense.hpp"
#include "hiopVectorHipNative.hpp"
#include "hiopamd_runtime.hpp"

namespace hiop
{
class hiopMatrixDenseHipNative : public hiopMatrixDense
{
public:
  hiopMatrixDenseHipNative(const size_type& m, const size_type& glob_n, index_type* col_part = nullptr,
                           MPI_Comm comm = MPI_COMM_SELF, const size_type& m_max_alloc = -1);
  /// non-owning view of an m x n row-major device array (the system matrix a hiopamd_linsolver owns)
  hiopMatrixDenseHipNative(const size_type& m, const size_type& n, double* external_device_storage);
  virtual ~hiopMatrixDenseHipNative();

  void setToZero() override;
  void setToConstant(double c) override;
  void copyFrom(const hiopMatrixDense& dm) override;
  void copyFrom(const double* buffer) override;
  void copy_to(double* buffer) override;
  void timesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const override;
  void timesVec(double beta, double* y, double alpha, const double* x) const override;
  void transTimesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const override;
  void transTimesVec(double beta, double* y, double alpha, const double* x) const override;
  void timesMat(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void timesMat_local(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void transTimesMat(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void timesMatTrans(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void timesMatTrans_local(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void addDiagonal(const double& alpha, const hiopVector& d) override;
  void addDiagonal(const double& value) override;
  void addSubDiagonal(const double& alpha, index_type start_on_dest_diag, const hiopVector& d) override;
  void addSubDiagonal(int start_on_dest_diag, const double& alpha, const hiopVector& d, int start_on_src_vec,
                      int num_elems = -1) override;
  void addSubDiagonal(int start_on_dest_diag, int num_elems, const double& c) override;
  void addMatrix(double alpha, const hiopMatrix& X) override;
  void transAddToSymDenseMatrixUpperTriangle(int row_dest_start, int col_dest_start, double alpha,
                                             hiopMatrixDense& W) const override;
  void addUpperTriangleToSymDenseMatrixUpperTriangle(int diag_start, double alpha, hiopMatrixDense& W) const override;
  double max_abs_value() override;
  void row_max_abs_value(hiopVector& ret_vec) override;
  void scale_row(hiopVector& vec_scal, const bool inv_scale) override;
  bool isfinite() const override;
  void print(FILE* f = nullptr, const char* msg = nullptr, int maxRows = -1, int maxCols = -1, int rank = -1) const override;
  hiopMatrixDense* alloc_clone() const override;
  hiopMatrixDense* new_copy() const override;
  void appendRow(const hiopVector& row) override;
  void copyRowsFrom(const hiopMatrixDense& src, int num_rows, int row_dest) override;
  void copyRowsFrom(const hiopMatrix& src_gen, const index_type* rows_idxs, size_type n_rows) override;
  void copyBlockFromMatrix(const index_type i_block_start, const index_type j_block_start,
                           const hiopMatrixDense& src) override;
  void copyFromMatrixBlock(const hiopMatrixDense& src, const int i_src_block_start, const int j_src_block_start) override;
  void shiftRows(size_type shift) override;
  void replaceRow(index_type row, const hiopVector& vec) override;
  void getRow(index_type irow, hiopVector& row_vec) override;
  void set_Hess_FR(const hiopMatrixDense& Hess, const hiopVector& add_diag_de) override;
  void set_Jac_FR(const hiopMatrixDense& Jac_c, const hiopMatrixDense& Jac_d) override;
#ifdef HIOP_DEEPCHECKS
  void overwriteUpperTriangleWithLower() override;
  void overwriteLowerTriangleWithUpper() override;
  bool assertSymmetry(double tol = 1e-16) const override;
#endif
  size_type get_local_size_n() const override { return n_local_; }
  size_type get_local_size_m() const override { return m_local_; }
  double* local_data_const() const override { return data_; }
  double* local_data() override { return data_; }
  bool symmetrize() override;

  /// device address of row i (the role of hiopMatrixDenseRowMajor::M_[i])
  double* row(index_type i) const { return data_ + static_cast<size_t>(i) * n_local_; }
  size_type max_rows() const { return max_rows_; }

private:
  hiopMatrixDenseHipNative(const hiopMatrixDenseHipNative& other);   // allocation only, like the reference's private copy ctor
  void allreduce_sum(double* dev_buf, size_type count) const;        // MPI_SUM over comm_ (identity without MPI)
  void to_host(double* host) const;
  hiopamd_ctx* ctx_;
  double* data_;          // max_rows_ x n_local_, row-major, device
  bool owns_data_;
  size_type n_local_;
  size_type glob_jl_, glob_ju_;
  size_type max_rows_;
  int myrank_;
  int comm_size_;
};
}  // namespace hiop
