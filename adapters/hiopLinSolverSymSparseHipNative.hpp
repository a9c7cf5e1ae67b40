// hiopLinSolverSymSparse on libhiopamd.so: the general sparse LDL^T without numerical pivoting (csrc/sparse_ldl.hip: nested dissection,
// multifrontal by tree levels on the device, dense root through the dense LDL^T), behind the reference's sparse linear-solver interface
// (src/LinAlg/hiopLinSolver.hpp:133-200).  Plays the role the reference gives to MA57 / the cuSOLVER sparse Cholesky inside
// hiopKKTLinSysCondensedSparse (src/Optimization/hiopKKTLinSysSparseCondensed.cpp:469-496): matrixChanged() factorises the
// symmetric system matrix and returns the number of negative pivots (-1 when a pivot is zero: "not factorisable", which the
// inertia-correction loop treats like a failed Cholesky, :386-388); solve() overwrites the right-hand side.
//
// The system matrix is the reference's symmetric sparse TRIPLET (one triangle, unique entries: what the KKT classes assemble into and
// what hiopLinSolverSymSparseMA57 reads, hiopLinSolverSymSparseMA57.cpp:140-170).  Its pattern is fixed over the IPM iterations: the first
// matrixChanged() reads the indices back once, builds the full (both triangles) CSR pattern, the symbolic analysis and a gather map
// CSR position -> triplet entry; every call then gathers the values on the device (one launch) and factorises.
#pragma once
#include "hiopLinSolver.hpp"
#include "hiopMatrixSparseTripletHipNative.hpp"
#include "hiopamd_runtime.hpp"

namespace hiop
{
class hiopLinSolverSymSparseHipNative : public hiopLinSolverSymSparse
{
public:
  hiopLinSolverSymSparseHipNative(size_type n, size_type nnz, hiopNlpFormulation* nlp);
  /// uses (does not own) the matrix the caller assembles into
  hiopLinSolverSymSparseHipNative(hiopMatrixSparse* M, hiopNlpFormulation* nlp);
  virtual ~hiopLinSolverSymSparseHipNative();

  int matrixChanged() override;
  bool solve(hiopVector& x) override;
  using hiopLinSolver::solve;   // (the base class's solve(hiopMatrix&) stays what it is: "not yet supported")

  /// (pos, neg, zero) pivots of the last factorisation
  bool compute_inertia(int& pos, int& neg, int& zero) const;
  /// supernodes, fronts, tree levels, order of the dense root, nnz(L) of the symbolic analysis (valid after the first matrixChanged())
  bool analysis_info(long long info8[8]) const;

private:
  int first_call();   // pattern -> CSR, symbolic analysis, gather map; HIOPAMD status

  hiopamd_ctx* ctx_;
  hiopamd_sparse_ldl* ldl_ = nullptr;
  int n_ = 0;
  long long nnz_csr_ = 0;
  double* csr_vals_ = nullptr;   // device: values of the full CSR pattern
  int* gather_ = nullptr;        // device: CSR position -> triplet entry
  int n_neg_ = 0, n_zero_ = 0;
  bool factored_ = false;
  int pattern_status_ = 0;       // != 0: first_call() rejected the pattern for good (HIOPAMD_ERR_ARG / _STATE); matrixChanged() fails fast
  int device_failures_ = 0;      // as in hiopLinSolverSymDenseHipNative: a failed device call is not "singular matrix"; solve() refuses while > 0
};
}  // namespace hiop
