#include "DenseConsEx2HipNative.hpp"

#include <cassert>
#include <vector>

namespace hiop {

DenseConsEx2HipNative::DenseConsEx2HipNative(size_type n, bool unconstrained)
{
  hiopamd_ctx* ctx = hiopamd_default_ctx();
  if(hiopamd_denseex2_create(&ex_, ctx, (int64_t)n, unconstrained ? 1 : 0) != HIOPAMD_OK) ex_ = nullptr;
  int64_t n64 = 0, m64 = 0;
  if(ex_ && hiopamd_denseex2_get_prob_sizes(ex_, &n64, &m64) == HIOPAMD_OK) {
    n_ = (size_type)n64;
    m_ = (size_type)m64;
  }
  int rank = 0, size = 1;
  if(hiopamd_ctx_comm(ctx, &rank, &size) == HIOPAMD_OK && size > 0) nranks_ = size;
  std::vector<int64_t> cols((size_t)nranks_ + 1, 0);
  if(ex_ && hiopamd_denseex2_get_vecdistrib_info(ex_, cols.data()) == HIOPAMD_OK) nlocal_ = (size_type)(cols[rank + 1] - cols[rank]);
}

DenseConsEx2HipNative::~DenseConsEx2HipNative() { hiopamd_denseex2_destroy(ex_); }

bool DenseConsEx2HipNative::get_prob_sizes(size_type& n, size_type& m)
{
  n = n_;
  m = m_;
  return ex_ != nullptr;
}

bool DenseConsEx2HipNative::get_vars_info(const size_type& n, double* xlow, double* xupp, NonlinearityType* type)
{
  assert(n == n_);
  for(size_type i = 0; i < nlocal_; ++i) type[i] = hiopNonlinear;   // host array (local part), as in the reference example
  return ex_ && hiopamd_denseex2_get_vars_info(ex_, xlow, xupp) == HIOPAMD_OK;
}

bool DenseConsEx2HipNative::get_cons_info(const size_type& m, double* clow, double* cupp, NonlinearityType* type)
{
  assert(m == m_);
  if(m_ == 0) return true;
  double lo[4], up[4];
  if(!ex_ || hiopamd_denseex2_get_cons_info(ex_, lo, up) != HIOPAMD_OK) return false;
  for(size_type i = 0; i < m_; ++i) type[i] = hiopInterfaceBase::hiopLinear;   // NlpDenseConsEx2.cpp:88-99
  // the bounds are replicated scalars; the solver's arrays live in its memory space (device)
  hiopamd_ctx* ctx = hiopamd_default_ctx();
  return hiopamd_copy_h2d(ctx, clow, lo, sizeof(double) * (size_t)m_) == HIOPAMD_OK &&
         hiopamd_copy_h2d(ctx, cupp, up, sizeof(double) * (size_t)m_) == HIOPAMD_OK && hiopamd_ctx_sync(ctx) == HIOPAMD_OK;
}

bool DenseConsEx2HipNative::eval_f(const size_type& n, const double* x, bool, double& obj_value)
{
  return ex_ && hiopamd_denseex2_eval_f(ex_, x, &obj_value) == HIOPAMD_OK;
}

bool DenseConsEx2HipNative::eval_grad_f(const size_type& n, const double* x, bool, double* gradf)
{
  return ex_ && hiopamd_denseex2_eval_grad_f(ex_, x, gradf) == HIOPAMD_OK;
}

bool DenseConsEx2HipNative::eval_cons(const size_type&, const size_type&, const size_type&, const index_type*, const double*, bool,
                                      double*)
{
  return false;   // use the one-call form
}

bool DenseConsEx2HipNative::eval_cons(const size_type& n, const size_type& m, const double* x, bool, double* cons)
{
  return ex_ && hiopamd_denseex2_eval_cons(ex_, x, cons) == HIOPAMD_OK;
}

bool DenseConsEx2HipNative::eval_Jac_cons(const size_type&, const size_type&, const size_type&, const index_type*, const double*,
                                          bool, double*)
{
  return false;   // use the one-call form
}

bool DenseConsEx2HipNative::eval_Jac_cons(const size_type& n, const size_type& m, const double* x, bool, double* Jac)
{
  return ex_ && hiopamd_denseex2_eval_Jac_cons(ex_, x, Jac) == HIOPAMD_OK;
}

bool DenseConsEx2HipNative::get_vecdistrib_info(size_type global_n, index_type* cols)
{
  if(!ex_ || global_n != n_) return false;
  std::vector<int64_t> c((size_t)nranks_ + 1, 0);
  if(hiopamd_denseex2_get_vecdistrib_info(ex_, c.data()) != HIOPAMD_OK) return false;
  for(size_type r = 0; r <= nranks_; ++r) cols[r] = (index_type)c[r];
  return true;
}

bool DenseConsEx2HipNative::get_starting_point(const size_type& n, double* x0)
{
  return ex_ && hiopamd_denseex2_get_starting_point(ex_, x0) == HIOPAMD_OK;
}

}  // namespace hiop
