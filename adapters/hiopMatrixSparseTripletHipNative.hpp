// hiopMatrixSparseTriplet / hiopMatrixSymSparseTriplet on libhiopamd.so (MI355X native): int32 row-sorted triplets resident in
// HBM.  Every pure virtual of src/LinAlg/hiopMatrix.hpp and src/LinAlg/hiopMatrixSparse.hpp:72-378 is overridden.
//
// Three groups (adapters/check_adapters.sh prints the split):
//  (1) the methods the MDS KKT path and the NLP scaling / residual code call: forwarded to hiopamd_sp_* / hiopamd_spsym_*;
//  (2) the methods the reference's own triplet class answers with assert(false && "not needed")
//      (src/LinAlg/hiopMatrixSparseTriplet.cpp:128-141, :203-214, :251-253, :348-351): same here;
//  (3) the sparse-NLP KKT *assembly* helpers (copySubmatrixFrom, copyRowsBlockFrom, set_Jac_FR, ...), used only by
//      hiopKKTLinSysSparse* / hiopNlpSparse: overridden, but they stop loudly (hiopamd_not_in_path) — they belong to the
//      sparse-NLP row, not to the MDS / dense hot path this library is a drop-in for.
#pragma once
#include "hiopMatrixSparse.hpp"
#include "hiopMatrixDenseHipNative.hpp"
#include "hiopVectorHipNative.hpp"
#include "hiopamd_runtime.hpp"

#include <unordered_map>

namespace hiop
{
class hiopMatrixSparseTripletHipNative : public hiopMatrixSparse
{
public:
  hiopMatrixSparseTripletHipNative(int rows, int cols, int nnz);
  virtual ~hiopMatrixSparseTripletHipNative();

  // ---- (1) device implementations ----
  void setToZero() override;
  void setToConstant(double c) override;
  void copy_to(int* irow, int* jcol, double* val) override;
  void copy_to(hiopMatrixDense& W) override;
  void timesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const override;
  void timesVec(double beta, double* y, double alpha, const double* x) const override;
  void transTimesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const override;
  void transTimesVec(double beta, double* y, double alpha, const double* x) const override;
  void timesMatTrans(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void transAddToSymDenseMatrixUpperTriangle(int row_dest_start, int col_dest_start, double alpha,
                                             hiopMatrixDense& W) const override;
  void addUpperTriangleToSymDenseMatrixUpperTriangle(int diag_start, double alpha, hiopMatrixDense& W) const override;
  void addMDinvMtransToDiagBlockOfSymDeMatUTri(int rowCol_dest_start, const double& alpha, const hiopVector& D,
                                               hiopMatrixDense& W) const override;
  void addMDinvNtransToSymDeMatUTri(int row_dest_start, int col_dest_start, const double& alpha, const hiopVector& D,
                                    const hiopMatrixSparse& N, hiopMatrixDense& W) const override;
  double max_abs_value() override;
  void row_max_abs_value(hiopVector& ret_vec) override;
  void scale_row(hiopVector& vec_scal, const bool inv_scale) override;
  bool isfinite() const override;
  void print(FILE* f = nullptr, const char* msg = nullptr, int maxRows = -1, int maxCols = -1, int rank = -1) const override;
  void startingAtAddSubDiagonalToStartingAt(int diag_src_start, const double& alpha, hiopVector& vec_dest, int vec_start,
                                            int num_elems = -1) const override;
  hiopMatrixSparse* alloc_clone() const override;
  hiopMatrixSparse* new_copy() const override;
  index_type* i_row() override { structure_changed(); return iRow_; }
  index_type* j_col() override { structure_changed(); return jCol_; }
  double* M() override { return values_; }
  const index_type* i_row() const override { return iRow_; }
  const index_type* j_col() const override { return jCol_; }
  const double* M() const override { return values_; }
  size_type numberOfOffDiagNonzeros() const override;
  bool is_diagonal() const override;
  void extract_diagonal(hiopVector& diag_out) const override;
#ifdef HIOP_DEEPCHECKS
  bool checkIndexesAreOrdered() const override;
#else
  bool checkIndexesAreOrdered() const;
#endif
  size_type m() const override { return nrows_; }
  size_type n() const override { return ncols_; }

  // ---- (2) "not needed" in the reference's triplet class as well ----
  void copyFrom(const hiopMatrixSparse& dm) override;
  void timesMat(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void transTimesMat(double beta, hiopMatrix& W, double alpha, const hiopMatrix& X) const override;
  void addDiagonal(const double& alpha, const hiopVector& D) override;
  void addDiagonal(const double& value) override;
  void addSubDiagonal(const double& alpha, index_type start, const hiopVector& D) override;
  void addSubDiagonal(int start_on_dest_diag, const double& alpha, const hiopVector& d_, int start_on_src_vec,
                      int num_elems = -1) override;
  void addSubDiagonal(int start_on_dest_diag, int num_elems, const double& c) override;
  void addMatrix(double alpha, const hiopMatrix& X) override;

  // ---- (3) sparse-NLP KKT assembly helpers: not in the MDS / dense path ----
  void copyRowsFrom(const hiopMatrix& src, const index_type* rows_idxs, size_type n_rows) override;
  void copySubDiagonalFrom(const index_type& start_on_dest_diag, const size_type& num_elems, const hiopVector& d_,
                           const index_type& start_on_nnz_idx, double scal = 1.0) override;
  void setSubDiagonalTo(const index_type& start_on_dest_diag, const size_type& num_elems, const double& c,
                        const index_type& start_on_nnz_idx) override;
  void copyRowsBlockFrom(const hiopMatrix& src_gen, const index_type& rows_src_idx_st, const size_type& n_rows,
                         const index_type& rows_dest_idx_st, const size_type& dest_nnz_st) override;
  void copySubmatrixFrom(const hiopMatrix& src_gen, const index_type& dest_row_st, const index_type& dest_col_st,
                         const size_type& dest_nnz_st, const bool offdiag_only = false) override;
  void copySubmatrixFromTrans(const hiopMatrix& src_gen, const index_type& dest_row_st, const index_type& dest_col_st,
                              const size_type& dest_nnz_st, const bool offdiag_only = false) override;
  void setSubmatrixToConstantDiag_w_colpattern(const double& scalar, const index_type& dest_row_st,
                                               const index_type& dest_col_st, const size_type& dest_nnz_st,
                                               const size_type& nnz_to_copy, const hiopVector& ix) override;
  void setSubmatrixToConstantDiag_w_rowpattern(const double& scalar, const index_type& dest_row_st,
                                               const index_type& dest_col_st, const size_type& dest_nnz_st,
                                               const size_type& nnz_to_copy, const hiopVector& ix) override;
  void copyDiagMatrixToSubblock(const double& src_val, const index_type& dest_row_st, const index_type& dest_col_st,
                                const size_type& dest_nnz_st, const size_type& nnz_to_copy) override;
  void copyDiagMatrixToSubblock_w_pattern(const hiopVector& dx, const index_type& dest_row_st,
                                          const index_type& dest_col_st, const size_type& dest_nnz_st,
                                          const size_type& nnz_to_copy, const hiopVector& pattern) override;
  void set_Jac_FR(const hiopMatrixSparse& Jac_c, const hiopMatrixSparse& Jac_d, int* iJacS, int* jJacS,
                  double* MJacS) override;
  void set_Hess_FR(const hiopMatrixSparse& Hess, int* iHSS, int* jHSS, double* MHSS, const hiopVector& add_diag) override;

  /// the sparsity pattern changed (non-const i_row()/j_col() call it): cached Schur plans are rebuilt on next use
  void structure_changed() const;

protected:
  hiopamd_sp_plan* plan_with(const hiopMatrixSparseTripletHipNative& N, bool same_upper) const;
  hiopamd_ctx* ctx_;
  int* iRow_;        // device
  int* jCol_;        // device
  double* values_;   // device
  mutable std::unordered_map<const void*, hiopamd_sp_plan*> plans_;   // partner matrix -> symbolic Schur plan
  mutable double* ones_;                                              // ncols_ ones (timesMatTrans = M I^-1 N^T)
};

/// upper-triangle triplets of a symmetric matrix (hiopMatrixSymSparseTriplet, src/LinAlg/hiopMatrixSparseTriplet.hpp:366-430)
class hiopMatrixSymSparseTripletHipNative : public hiopMatrixSparseTripletHipNative
{
public:
  hiopMatrixSymSparseTripletHipNative(int n, int nnz) : hiopMatrixSparseTripletHipNative(n, n, nnz) {}
  virtual ~hiopMatrixSymSparseTripletHipNative() {}
  void timesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const override;
  void timesVec(double beta, double* y, double alpha, const double* x) const override;
  void transTimesVec(double beta, hiopVector& y, double alpha, const hiopVector& x) const override
  {
    timesVec(beta, y, alpha, x);
  }
  void transTimesVec(double beta, double* y, double alpha, const double* x) const override { timesVec(beta, y, alpha, x); }
  void transAddToSymDenseMatrixUpperTriangle(int row_dest_start, int col_dest_start, double alpha,
                                             hiopMatrixDense& W) const override;
  void addUpperTriangleToSymDenseMatrixUpperTriangle(int diag_start, double alpha, hiopMatrixDense& W) const override;
  void startingAtAddSubDiagonalToStartingAt(int diag_src_start, const double& alpha, hiopVector& vec_dest, int vec_start,
                                            int num_elems = -1) const override;
  hiopMatrixSparse* alloc_clone() const override;
  hiopMatrixSparse* new_copy() const override;
#ifdef HIOP_DEEPCHECKS
  bool assertSymmetry(double tol = 1e-16) const override { return true; }
#endif
  size_type numberOfOffDiagNonzeros() const override;
};
}  // namespace hiop
