// The reference's dense-constraints example DenseConsEx2 as a user problem whose callbacks run on the device.
//
// A `hiop::hiopInterfaceDenseConstraints` (src/Interface/hiopInterface.hpp:515-570) implementation equivalent to
// src/Drivers/Dense/NlpDenseConsEx2.{hpp,cpp}: local columns of x / gradf / the Jacobian on this rank's GPU, the example's two
// MPI_Allreduce through the all-reduce hook of the hiop_amd context (RCCL), every callback forwarded to `hiopamd_denseex2_*`
// (hiop_amd/csrc/example_dense.hip).  The split forms of eval_cons / eval_Jac_cons return false: HiOp then uses the one-call
// forms (hiopInterface.hpp:236-255, :549-561).  Compile-checked against the reference headers by adapters/check_adapters.sh.
#pragma once
#include "hiopInterface.hpp"
#include "hiopamd_runtime.hpp"

namespace hiop {

class DenseConsEx2HipNative : public hiopInterfaceDenseConstraints
{
public:
  explicit DenseConsEx2HipNative(size_type n, bool unconstrained = false);
  ~DenseConsEx2HipNative() override;

  bool get_prob_sizes(size_type& n, size_type& m) override;
  bool get_vars_info(const size_type& n, double* xlow, double* xupp, NonlinearityType* type) override;
  bool get_cons_info(const size_type& m, double* clow, double* cupp, NonlinearityType* type) override;
  bool eval_f(const size_type& n, const double* x, bool new_x, double& obj_value) override;
  bool eval_grad_f(const size_type& n, const double* x, bool new_x, double* gradf) override;
  bool eval_cons(const size_type& n, const size_type& m, const size_type& num_cons, const index_type* idx_cons, const double* x,
                 bool new_x, double* cons) override;
  bool eval_cons(const size_type& n, const size_type& m, const double* x, bool new_x, double* cons) override;
  bool eval_Jac_cons(const size_type& n, const size_type& m, const size_type& num_cons, const index_type* idx_cons,
                     const double* x, bool new_x, double* Jac) override;
  bool eval_Jac_cons(const size_type& n, const size_type& m, const double* x, bool new_x, double* Jac) override;
  bool get_vecdistrib_info(size_type global_n, index_type* cols) override;
  using hiopInterfaceBase::get_starting_point;   // (the primal-dual overload keeps its default: not provided)
  bool get_starting_point(const size_type& n, double* x0) override;

private:
  hiopamd_denseex2* ex_ = nullptr;
  size_type n_ = 0, m_ = 0, nranks_ = 1, nlocal_ = 0;
};

}  // namespace hiop
