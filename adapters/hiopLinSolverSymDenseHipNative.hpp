// hiopLinSolverSymDense on libhiopamd.so: the no-pivot blocked LDL^T on fp64 MFMA with inertia, behind the reference's
// linear-solver interface (src/LinAlg/hiopLinSolver.hpp:78-130).  Plays the role of hiopLinSolverSymDenseMagmaNopiv
// (src/LinAlg/hiopLinSolverSymDenseMagma.hpp:145-185, .cpp:324-480): matrixChanged() factorises the system matrix in place
// and returns the number of negative pivots (-1 when a pivot is zero / non-finite), solve() overwrites the right-hand side.
// The system matrix the KKT classes assemble into (sysMatrix()) IS the factorisation's storage in HBM — no host copy, no
// magma_dsetmatrix per iteration.
#pragma once
#include "hiopLinSolver.hpp"
#include "hiopMatrixDenseHipNative.hpp"
#include "hiopamd_runtime.hpp"

namespace hiop
{
class hiopLinSolverSymDenseHipNative : public hiopLinSolverSymDense
{
public:
  /// pivoted = true: the Bunch-Kaufman factorisation (hiopamd_linsolver_set_pivoting) instead of the no-pivot one
  hiopLinSolverSymDenseHipNative(int n, hiopNlpFormulation* nlp, bool pivoted = false);
  virtual ~hiopLinSolverSymDenseHipNative();

  int matrixChanged() override;
  bool solve(hiopVector& x) override;
  /// several right-hand sides: the rows of x (x is nrhs x n row-major, i.e. one right-hand side after the other)
  bool solve(hiopMatrix& x) override;

  /// inertia of the last factorisation (magmablas_ddiinertia equivalent)
  bool compute_inertia(int& pos, int& neg, int& zero) const;

private:
  hiopamd_ctx* ctx_;
  hiopamd_linsolver* ls_;
  int device_failures_ = 0;   // consecutive matrixChanged() calls that failed in the device layer (not: singular matrix); solve() refuses while > 0
  int n_;
};

// The safe solver: plays the role of hiopLinSolverSymDenseMagmaBuKa (src/LinAlg/hiopLinSolverSymDenseMagma.hpp:60-140, .cpp:120-250) --
// what hiopKKTLinSysCompressedMDSXYcYd::determineAndCreateLinsys creates when safe_mode_ is on (hiopKKTLinSysMDS.cpp:446-456).
// Same object, pivoted mode: LAPACK DSYTRF's pivots on the device (csrc/ldlt_bk.hip), exact inertia, DSYTRS solves.
class hiopLinSolverSymDenseHipNativeBuKa : public hiopLinSolverSymDenseHipNative
{
public:
  hiopLinSolverSymDenseHipNativeBuKa(int n, hiopNlpFormulation* nlp) : hiopLinSolverSymDenseHipNative(n, nlp, true) {}
};
}  // namespace hiop
