// The reference's MDS example as a user problem whose callbacks run on the device through libhiopamd.so.
//
// A `hiop::hiopInterfaceMDS` (src/Interface/hiopInterface.hpp:582-780) implementation equivalent to MdsEx1OneCallCons of
// src/Drivers/MDS/NlpMdsRajaEx1.{hpp,cpp}: with `mem_space = hip-native` the solver hands DEVICE pointers to these
// callbacks ("managed by Umpire" in the reference's notes) and every one of them forwards to `hiopamd_mdsex1_*`
// (hiop_amd/csrc/example_mds.hip) — no Jacobian / Hessian value touches the host.  The split (num_cons / idx_cons) forms of
// eval_cons / eval_Jac_cons return false, which makes HiOp use the one-call forms (hiopInterface.hpp:236-255, :655-704).
// Compile-checked against the reference headers by adapters/check_adapters.sh.
#pragma once
#include "hiopInterface.hpp"
#include "hiopamd_runtime.hpp"

namespace hiop {

class MdsEx1HipNative : public hiopInterfaceMDS
{
public:
  MdsEx1HipNative(int ns, int nd, bool empty_sp_row = false);
  ~MdsEx1HipNative() override;

  bool get_prob_sizes(size_type& n, size_type& m) override;
  bool get_vars_info(const size_type& n, double* xlow, double* xupp, NonlinearityType* type) override;
  bool get_cons_info(const size_type& m, double* clow, double* cupp, NonlinearityType* type) override;
  bool get_sparse_dense_blocks_info(int& nx_sparse, int& nx_dense, int& nnz_sparse_Jaceq, int& nnz_sparse_Jacineq,
                                    int& nnz_sparse_Hess_Lagr_SS, int& nnz_sparse_Hess_Lagr_SD) override;
  bool eval_f(const size_type& n, const double* x, bool new_x, double& obj_value) override;
  bool eval_grad_f(const size_type& n, const double* x, bool new_x, double* gradf) override;
  // split forms: not provided (false) -> HiOp calls the one-call forms below
  bool eval_cons(const size_type& n, const size_type& m, const size_type& num_cons, const index_type* idx_cons, const double* x,
                 bool new_x, double* cons) override;
  bool eval_cons(const size_type& n, const size_type& m, const double* x, bool new_x, double* cons) override;
  bool eval_Jac_cons(const size_type& n, const size_type& m, const size_type& num_cons, const index_type* idx_cons,
                     const double* x, bool new_x, const size_type& nsparse, const size_type& ndense, const size_type& nnzJacS,
                     index_type* iJacS, index_type* jJacS, double* MJacS, double* JacD) override;
  bool eval_Jac_cons(const size_type& n, const size_type& m, const double* x, bool new_x, const size_type& nsparse,
                     const size_type& ndense, const size_type& nnzJacS, index_type* iJacS, index_type* jJacS, double* MJacS,
                     double* JacD) override;
  bool eval_Hess_Lagr(const size_type& n, const size_type& m, const double* x, bool new_x, const double& obj_factor,
                      const double* lambda, bool new_lambda, const size_type& nsparse, const size_type& ndense,
                      const size_type& nnzHSS, index_type* iHSS, index_type* jHSS, double* MHSS, double* HDD, size_type& nnzHSD,
                      index_type* iHSD, index_type* jHSD, double* MHSD) override;
  using hiopInterfaceBase::get_starting_point;   // (the primal-dual overload keeps its default: not provided)
  bool get_starting_point(const size_type& n, double* x0) override;
  bool get_MPI_comm(MPI_Comm& comm_out) override
  {
    comm_out = MPI_COMM_SELF;   // the MDS interface is local (hiopInterface.hpp:582-584)
    return true;
  }

private:
  hiopamd_mdsex1* ex_ = nullptr;
  size_type ns_ = 0, nd_ = 0;
};

}  // namespace hiop
