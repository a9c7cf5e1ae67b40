#include "hiopLinSolverSymDenseHipNative.hpp"

namespace hiop
{
hiopLinSolverSymDenseHipNative::hiopLinSolverSymDenseHipNative(int n, hiopNlpFormulation* nlp, bool pivoted)
    : hiopLinSolverSymDense(n, nlp), ctx_(hiopamd_default_ctx()), ls_(nullptr), n_(n)
{
  hiopamd_ok(hiopamd_linsolver_create(&ls_, ctx_, n));
  if(pivoted) hiopamd_ok(hiopamd_linsolver_set_pivoting(ls_, 1));
  // the base class allocated a system matrix through the LinAlg factory; the KKT classes must write into the solver's own
  // storage instead, so M_ becomes a non-owning view of it (the base destructor deletes the view, not the storage)
  delete M_;
  M_ = new hiopMatrixDenseHipNative(n, n, hiopamd_linsolver_sys_matrix(ls_));
}

hiopLinSolverSymDenseHipNative::~hiopLinSolverSymDenseHipNative()
{
  hiopamd_ctx_sync(ctx_);
  hiopamd_linsolver_destroy(ls_);
}

int hiopLinSolverSymDenseHipNative::matrixChanged()
{
  assert(M_->n() == M_->m() && M_->n() == n_);
  if(nlp_) nlp_->runStats.linsolv.tmFactTime.start();
  int n_neg = -1;
  // The solver object keeps a copy of the assembled upper triangle while its dataflow factorisation runs (its default,
  // hiopamd_linsolver_set_retry_copy): a bounded wait that expires is answered INSIDE this call by a stepwise factorisation of
  // the restored matrix, so HIOPAMD_ERR_TIMEOUT never arrives here.  HIOPAMD_ERR_SOLVE = "a solve since the last factorisation
  // delivered invalid results; the matrix is intact, call again" (the solve() that was affected has already returned false).
  int rc = hiopamd_linsolver_matrix_changed(ls_, &n_neg);
  if(rc == HIOPAMD_ERR_SOLVE) rc = hiopamd_linsolver_matrix_changed(ls_, &n_neg);
  if(nlp_) {
    nlp_->runStats.linsolv.tmFactTime.stop();
    double ff = 0.0, fs = 0.0;
    if(hiopamd_linsolver_flops(ls_, &ff, &fs) == HIOPAMD_OK) nlp_->runStats.linsolv.flopsFact = ff / 1e12;
  }
  if(rc != HIOPAMD_OK) {
    // A runtime failure of the device layer (HIP error, lost device).  The reference's contract has exactly two answers — the number
    // of negative eigenvalues or -1 (hiopLinSolver.hpp:117-130; the LAPACK class returns -1 for info != 0 as well,
    // hiopLinSolverSymDenseLapack.hpp:103-117) — so this is reported on stderr and answered with -1: the IPM's inertia-correction
    // loop re-assembles and calls again (hiopKKTLinSys.cpp:316-372), and gives up in its own way if the failure persists.
    // What the -1 must NOT do is pass for "singular matrix" silently: the failure is counted, every message says DEVICE FAILURE, and while
    // the count is non-zero solve() returns false (the reference's callers stop on a failed solve with "linear solver error" instead of
    // taking a direction computed from a factorisation that never happened).  A later successful factorisation clears the count.
    ++device_failures_;
    std::fprintf(stderr,
                 "hiop_amd: DEVICE FAILURE in hiopamd_linsolver_matrix_changed (status %d, %d in a row) -- this is not a singular matrix; "
                 "reporting -1 to the caller, solve() will return false until a factorisation succeeds\n",
                 rc, device_failures_);
    return -1;
  }
  // the reference's callers own sysMatrix() between calls and may write it by any means (other streams, blocking copies): nothing of
  // this call may still be reading it when it returns (hiop_amd.h, "STREAM CONTRACT" of hiopamd_linsolver_matrix_changed)
  hiopamd_ctx_sync(ctx_);
  device_failures_ = 0;
  return n_neg;   // -1: zero / non-finite pivot (or, in safe mode, a probe solve that did not converge): the reference's "singular" answer
}

bool hiopLinSolverSymDenseHipNative::solve(hiopVector& x)
{
  assert(x.get_size() == n_);
  if(device_failures_ > 0) return false;   // the last factorisation failed in the device layer (see matrixChanged)
  if(nlp_) nlp_->runStats.linsolv.tmTriuSolves.start();
  const int rc = hiopamd_linsolver_solve(ls_, x.local_data(), 1);
  int ok = 1;
  const int rc2 = hiopamd_linsolver_solve_status(ls_, &ok);
  if(nlp_) nlp_->runStats.linsolv.tmTriuSolves.stop();
  return rc == HIOPAMD_OK && rc2 == HIOPAMD_OK && ok != 0;
}

bool hiopLinSolverSymDenseHipNative::solve(hiopMatrix& x_)
{
  auto& x = dynamic_cast<hiopMatrixDenseHipNative&>(x_);
  assert(x.n() == n_);
  if(device_failures_ > 0) return false;
  const int rc = hiopamd_linsolver_solve(ls_, x.local_data(), x.m());
  int ok = 1;
  const int rc2 = hiopamd_linsolver_solve_status(ls_, &ok);
  return rc == HIOPAMD_OK && rc2 == HIOPAMD_OK && ok != 0;
}

bool hiopLinSolverSymDenseHipNative::compute_inertia(int& pos, int& neg, int& zero) const
{
  return hiopamd_linsolver_inertia(ls_, &pos, &neg, &zero) == HIOPAMD_OK;
}
}  // namespace hiop
