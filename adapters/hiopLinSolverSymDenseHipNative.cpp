#include "hiopLinSolverSymDenseHipNative.hpp"

namespace hiop
{
hiopLinSolverSymDenseHipNative::hiopLinSolverSymDenseHipNative(int n, hiopNlpFormulation* nlp)
    : hiopLinSolverSymDense(n, nlp), ctx_(hiopamd_default_ctx()), ls_(nullptr), n_(n)
{
  hiopamd_ok(hiopamd_linsolver_create(&ls_, ctx_, n));
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
  const int rc = hiopamd_linsolver_matrix_changed(ls_, &n_neg);
  if(nlp_) {
    nlp_->runStats.linsolv.tmFactTime.stop();
    double ff = 0.0, fs = 0.0;
    if(hiopamd_linsolver_flops(ls_, &ff, &fs) == HIOPAMD_OK) nlp_->runStats.linsolv.flopsFact = ff / 1e12;
  }
  if(rc == HIOPAMD_ERR_TIMEOUT) {
    // the dataflow kernels gave up and the matrix is overwritten: this is NOT "singular".  The solver object has switched to
    // its stepwise kernels; the KKT class has to assemble again, which only the caller of matrixChanged() can do — the
    // reference has no channel for that, so stop loudly rather than send the IPM into inertia correction on a lie
    std::fprintf(stderr, "hiop_amd: the LDL^T factorisation timed out (device shared with another process?); re-assemble and call matrixChanged() again\n");
    std::abort();
  }
  if(rc != HIOPAMD_OK) {   // a runtime failure (HIP error, invalid solve results seen since the last factorisation): not "singular" either
    std::fprintf(stderr, "hiop_amd: hiopamd_linsolver_matrix_changed failed with status %d\n", rc);
    std::abort();
  }
  return n_neg;   // -1: zero / non-finite pivot (or, in safe mode, a probe solve that did not converge): the reference's "singular" answer
}

bool hiopLinSolverSymDenseHipNative::solve(hiopVector& x)
{
  assert(x.get_size() == n_);
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
