#include "MdsEx1HipNative.hpp"

#include <cassert>

namespace hiop {

static_assert(sizeof(index_type) == sizeof(int), "hiopamd_mdsex1_* writes int index arrays (hiop_index_type = int)");

MdsEx1HipNative::MdsEx1HipNative(int ns, int nd, bool empty_sp_row)
{
  if(hiopamd_mdsex1_create(&ex_, hiopamd_default_ctx(), ns, nd, empty_sp_row ? 1 : 0) != HIOPAMD_OK) ex_ = nullptr;
  int64_t n = 0, m = 0;
  if(ex_ && hiopamd_mdsex1_get_prob_sizes(ex_, &n, &m) == HIOPAMD_OK) {
    ns_ = (size_type)(m - 3);
    nd_ = (size_type)(n - 2 * (m - 3));
  }
}

MdsEx1HipNative::~MdsEx1HipNative() { hiopamd_mdsex1_destroy(ex_); }

bool MdsEx1HipNative::get_prob_sizes(size_type& n, size_type& m)
{
  int64_t n64 = 0, m64 = 0;
  if(!ex_ || hiopamd_mdsex1_get_prob_sizes(ex_, &n64, &m64) != HIOPAMD_OK) return false;
  n = (size_type)n64;
  m = (size_type)m64;
  return true;
}

bool MdsEx1HipNative::get_vars_info(const size_type& n, double* xlow, double* xupp, NonlinearityType* type)
{
  assert(n == 2 * ns_ + nd_);
  // `type` is a host array in the reference's device example as well (NlpMdsRajaEx1.cpp:276-281)
  for(size_type i = 0; i < n; ++i) type[i] = hiopNonlinear;
  return ex_ && hiopamd_mdsex1_get_vars_info(ex_, xlow, xupp) == HIOPAMD_OK;
}

bool MdsEx1HipNative::get_cons_info(const size_type& m, double* clow, double* cupp, NonlinearityType* type)
{
  assert(m == ns_ + 3);
  for(size_type i = 0; i < m; ++i) type[i] = hiopNonlinear;
  return ex_ && hiopamd_mdsex1_get_cons_info(ex_, clow, cupp) == HIOPAMD_OK;
}

bool MdsEx1HipNative::get_sparse_dense_blocks_info(int& nx_sparse, int& nx_dense, int& nnz_sparse_Jaceq, int& nnz_sparse_Jacineq,
                                                   int& nnz_sparse_Hess_Lagr_SS, int& nnz_sparse_Hess_Lagr_SD)
{
  return ex_ && hiopamd_mdsex1_get_sparse_dense_blocks_info(ex_, &nx_sparse, &nx_dense, &nnz_sparse_Jaceq, &nnz_sparse_Jacineq,
                                                            &nnz_sparse_Hess_Lagr_SS, &nnz_sparse_Hess_Lagr_SD) == HIOPAMD_OK;
}

bool MdsEx1HipNative::eval_f(const size_type& n, const double* x, bool, double& obj_value)
{
  return ex_ && hiopamd_mdsex1_eval_f(ex_, x, &obj_value) == HIOPAMD_OK;
}

bool MdsEx1HipNative::eval_grad_f(const size_type& n, const double* x, bool, double* gradf)
{
  return ex_ && hiopamd_mdsex1_eval_grad_f(ex_, x, gradf) == HIOPAMD_OK;
}

bool MdsEx1HipNative::eval_cons(const size_type&, const size_type&, const size_type&, const index_type*, const double*, bool, double*)
{
  return false;   // use the one-call form
}

bool MdsEx1HipNative::eval_cons(const size_type& n, const size_type& m, const double* x, bool, double* cons)
{
  assert(m == ns_ + 3);
  return ex_ && hiopamd_mdsex1_eval_cons(ex_, x, cons) == HIOPAMD_OK;
}

bool MdsEx1HipNative::eval_Jac_cons(const size_type&, const size_type&, const size_type&, const index_type*, const double*, bool,
                                    const size_type&, const size_type&, const size_type&, index_type*, index_type*, double*,
                                    double*)
{
  return false;   // use the one-call form
}

bool MdsEx1HipNative::eval_Jac_cons(const size_type& n, const size_type& m, const double* x, bool, const size_type& nsparse,
                                    const size_type& ndense, const size_type& nnzJacS, index_type* iJacS, index_type* jJacS,
                                    double* MJacS, double* JacD)
{
  if(!ex_) return false;
  assert(nsparse == 2 * ns_ && ndense == nd_);
  // equalities in the head of the arrays, the three inequalities behind them with their rows shifted by ns
  // (the layout of MdsEx1OneCallCons::eval_Jac_cons, NlpMdsRajaEx1.cpp:894-1011)
  const size_type off = 2 * ns_;
  int* ie = reinterpret_cast<int*>(iJacS);
  int* je = reinterpret_cast<int*>(jJacS);
  if(hiopamd_mdsex1_eval_Jac_cons_eq(ex_, x, ie, je, MJacS, JacD) != HIOPAMD_OK) return false;
  return hiopamd_mdsex1_eval_Jac_cons_ineq(ex_, x, (int)ns_, ie ? ie + off : nullptr, je ? je + off : nullptr,
                                           MJacS ? MJacS + off : nullptr, JacD ? JacD + ns_ * nd_ : nullptr) == HIOPAMD_OK;
}

bool MdsEx1HipNative::eval_Hess_Lagr(const size_type& n, const size_type& m, const double* x, bool, const double& obj_factor,
                                     const double* lambda, bool, const size_type& nsparse, const size_type& ndense,
                                     const size_type& nnzHSS, index_type* iHSS, index_type* jHSS, double* MHSS, double* HDD,
                                     size_type& nnzHSD, index_type* iHSD, index_type* jHSD, double* MHSD)
{
  assert(nnzHSS == 2 * ns_);
  assert(nnzHSD == 0 && iHSD == nullptr && jHSD == nullptr && MHSD == nullptr);
  return ex_ && hiopamd_mdsex1_eval_Hess_Lagr(ex_, x, obj_factor, lambda, reinterpret_cast<int*>(iHSS),
                                              reinterpret_cast<int*>(jHSS), MHSS, HDD) == HIOPAMD_OK;
}

bool MdsEx1HipNative::get_starting_point(const size_type& n, double* x0)
{
  return ex_ && hiopamd_mdsex1_get_starting_point(ex_, x0) == HIOPAMD_OK;
}

}  // namespace hiop
