// hiopVector on libhiopamd.so (MI355X native).  Every pure virtual of src/LinAlg/hiopVector.hpp:74-1003 is overridden and
// forwards to the C ABI entry point that carries the same reference line number in include/hiop_amd.h; semantics are those
// of hiopVectorPar (src/LinAlg/hiopVectorPar.cpp), incl. the MPI reductions of the distributed methods.
#pragma once
#include "hiopVector.hpp"
#include "hiopamd_runtime.hpp"

namespace hiop
{
class hiopVectorHipNative : public hiopVector
{
public:
  hiopVectorHipNative(const size_type& glob_n, index_type* col_part = nullptr, MPI_Comm comm = MPI_COMM_SELF);
  virtual ~hiopVectorHipNative();

  void setToZero() override;
  void setToConstant(double c) override;
  void set_to_random_uniform(double minv, double maxv) override;
  void setToConstant_w_patternSelect(double c, const hiopVector& select) override;
  void copyFrom(const hiopVector& vec) override;
  void copyFrom(const double* local_array) override;
  void copy_from_w_pattern(const hiopVector& src, const hiopVector& select) override;
  void copyFromStarting(int start_index_in_this, const double* v, int nv) override;
  void copyFromStarting(int start_index, const hiopVector& src) override;
  void copy_from_starting_at(const double* v, int start_index_in_v, int n) override;
  void copy_from_vectorpar(const hiopVectorPar& vsrc) override;
  void copy_from_indexes(const hiopVector& src, const hiopVectorInt& index_in_src) override;
  void copy_from_indexes(const double* src, const hiopVectorInt& index_in_src) override;
  void startingAtCopyFromStartingAt(int start_idx_dest, const hiopVector& v, int start_idx_src) override;
  void copyTo(double* dest) const override;
  void copy_to_vectorpar(hiopVectorPar& vdest) const override;
  void copyToStarting(int start_index, hiopVector& dst) const override;
  void copyToStarting(hiopVector& vec, int start_index_in_dest) const override;
  void copyToStartingAt_w_pattern(hiopVector& vec, index_type start_index_in_dest, const hiopVector& ix) const override;
  void copy_from_two_vec_w_pattern(const hiopVector& c, const hiopVectorInt& c_map, const hiopVector& d,
                                   const hiopVectorInt& d_map) override;
  void copy_to_two_vec_w_pattern(hiopVector& c, const hiopVectorInt& c_map, hiopVector& d,
                                 const hiopVectorInt& d_map) const override;
  void startingAtCopyToStartingAt(index_type start_idx_in_src, hiopVector& dest, index_type start_idx_dest,
                                  size_type num_elems = -1) const override;
  void startingAtCopyToStartingAt_w_pattern(index_type start_idx_in_src, hiopVector& dest, index_type start_idx_dest,
                                            const hiopVector& selec_dest, size_type num_elems = -1) const override;
  double twonorm() const override;
  double infnorm() const override;
  double infnorm_local() const override;
  double onenorm() const override;
  double onenorm_local() const override;
  void componentMult(const hiopVector& vec) override;
  void componentDiv(const hiopVector& vec) override;
  void componentDiv_w_selectPattern(const hiopVector& vec, const hiopVector& select) override;
  void component_min(const double constant) override;
  void component_min(const hiopVector& vec) override;
  void component_max(const double constant) override;
  void component_max(const hiopVector& v) override;
  void component_abs() override;
  void component_sgn() override;
  void component_sqrt() override;
  void scale(double c) override;
  void axpy(double alpha, const hiopVector& xvec) override;
  void axpy_w_pattern(double alpha, const hiopVector& xvec, const hiopVector& select) override;
  void axpy(double alpha, const hiopVector& xvec, const hiopVectorInt& i) override;
  void axzpy(double alpha, const hiopVector& xvec, const hiopVector& zvec) override;
  void axdzpy(double alpha, const hiopVector& xvec, const hiopVector& zvec) override;
  void axdzpy_w_pattern(double alpha, const hiopVector& xvec, const hiopVector& zvec, const hiopVector& select) override;
  void addConstant(double c) override;
  void addConstant_w_patternSelect(double c, const hiopVector& select) override;
  double dotProductWith(const hiopVector& vec) const override;
  void negate() override;
  void invert() override;
  double logBarrier_local(const hiopVector& select) const override;
  void addLogBarrierGrad(double alpha, const hiopVector& xvec, const hiopVector& select) override;
  double sum_local() const override;
  double linearDampingTerm_local(const hiopVector& ixleft, const hiopVector& ixright, const double& mu,
                                 const double& kappa_d) const override;
  void addLinearDampingTerm(const hiopVector& ixleft, const hiopVector& ixright, const double& alpha,
                            const double& ct) override;
  int allPositive() override;
  int allPositive_w_patternSelect(const hiopVector& select) override;
  double min() const override;
  double min_w_pattern(const hiopVector& select) const override;
  void min(double& minval, int& index) const override;
  bool projectIntoBounds_local(const hiopVector& xlo, const hiopVector& ixl, const hiopVector& xup, const hiopVector& ixu,
                               double kappa1, double kappa2) override;
  double fractionToTheBdry_local(const hiopVector& dvec, const double& tau) const override;
  double fractionToTheBdry_w_pattern_local(const hiopVector& dvec, const double& tau, const hiopVector& select) const override;
  void selectPattern(const hiopVector& select) override;
  bool matchesPattern(const hiopVector& select) override;
  void adjustDuals_plh(const hiopVector& xvec, const hiopVector& ixvec, const double& mu, const double& kappa) override;
  bool is_zero() const override;
  bool isnan_local() const override;
  bool isinf_local() const override;
  bool isfinite_local() const override;
  void print(FILE* file = nullptr, const char* message = nullptr, int max_elems = -1, int rank = -1) const override;
  hiopVector* alloc_clone() const override;
  hiopVector* new_copy() const override;
  size_type get_local_size() const override { return n_local_; }
  double* local_data() override { return data_; }
  const double* local_data_const() const override { return data_; }
  double* local_data_host() override;
  const double* local_data_host_const() const override;
  size_type numOfElemsLessThan(const double& val) const override;
  size_type numOfElemsAbsLessThan(const double& val) const override;
  void set_array_from_to(hiopInterfaceBase::NonlinearityType* arr, const int start, const int end,
                         const hiopInterfaceBase::NonlinearityType* arr_src, const int start_src) const override;
  void set_array_from_to(hiopInterfaceBase::NonlinearityType* arr, const int start, const int end,
                         const hiopInterfaceBase::NonlinearityType arr_src) const override;
  bool is_equal(const hiopVector& vec) const override;

  /// host mirror <-> device (what the RAJA / Hip back-ends expose as copyToDev / copyFromDev)
  void copyToDev() const;
  void copyFromDev() const;
  MPI_Comm get_mpi_comm() const { return comm_; }

private:
  static const double* dev(const hiopVector& v);
  static double* dev(hiopVector& v);
  double reduce_sum(double local) const;   // MPI_SUM over comm_ (identity without MPI)
  double reduce_max(double local) const;
  double reduce_min(double local) const;
  hiopamd_ctx* ctx_;
  double* data_;
  mutable double* host_mirror_;
  size_type glob_il_, glob_iu_;
  size_type n_local_;
  MPI_Comm comm_;
  int comm_size_;
};
}  // namespace hiop
