// see hiopVectorHipNative.hpp.  Reference behaviour: src/LinAlg/hiopVectorPar.cpp (line numbers in include/hiop_amd.h).
#include "hiopVectorHipNative.hpp"

#include "hiopVectorPar.hpp"

#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

namespace hiop
{
#define C_ ctx_
#define N_ (int64_t) n_local_

hiopVectorHipNative::hiopVectorHipNative(const size_type& glob_n, index_type* col_part, MPI_Comm comm)
    : hiopVector(), ctx_(hiopamd_default_ctx()), data_(nullptr), host_mirror_(nullptr), comm_(comm), comm_size_(1)
{
  n_ = glob_n;
#ifdef HIOP_USE_MPI
  if(comm_ == MPI_COMM_NULL) comm_ = MPI_COMM_SELF;
  int ierr = MPI_Comm_size(comm_, &comm_size_);
  assert(MPI_SUCCESS == ierr);
  int P = 0;
  if(col_part) {
    ierr = MPI_Comm_rank(comm_, &P);
    assert(ierr == MPI_SUCCESS);
    glob_il_ = col_part[P];
    glob_iu_ = col_part[P + 1];
  } else {
    glob_il_ = 0;
    glob_iu_ = n_;
  }
#else
  (void)col_part;
  glob_il_ = 0;
  glob_iu_ = n_;
#endif
  n_local_ = glob_iu_ - glob_il_;
  data_ = hiopamd_new_array((size_t)n_local_);
}

hiopVectorHipNative::~hiopVectorHipNative()
{
  hiopamd_ok(hiopamd_ctx_sync(ctx_));
  hiopamd_ok(hiopamd_free(data_));
  delete[] host_mirror_;
}

const double* hiopVectorHipNative::dev(const hiopVector& v) { return v.local_data_const(); }
double* hiopVectorHipNative::dev(hiopVector& v) { return v.local_data(); }

double hiopVectorHipNative::reduce_sum(double local) const
{
#ifdef HIOP_USE_MPI
  double g;
  int ierr = MPI_Allreduce(&local, &g, 1, MPI_DOUBLE, MPI_SUM, comm_);
  assert(MPI_SUCCESS == ierr);
  return g;
#else
  return local;
#endif
}
double hiopVectorHipNative::reduce_max(double local) const
{
#ifdef HIOP_USE_MPI
  double g;
  int ierr = MPI_Allreduce(&local, &g, 1, MPI_DOUBLE, MPI_MAX, comm_);
  assert(MPI_SUCCESS == ierr);
  return g;
#else
  return local;
#endif
}
double hiopVectorHipNative::reduce_min(double local) const
{
#ifdef HIOP_USE_MPI
  double g;
  int ierr = MPI_Allreduce(&local, &g, 1, MPI_DOUBLE, MPI_MIN, comm_);
  assert(MPI_SUCCESS == ierr);
  return g;
#else
  return local;
#endif
}

void hiopVectorHipNative::setToZero() { hiopamd_ok(hiopamd_vec_set_to_constant(C_, N_, data_, 0.0)); }
void hiopVectorHipNative::setToConstant(double c) { hiopamd_ok(hiopamd_vec_set_to_constant(C_, N_, data_, c)); }
void hiopVectorHipNative::set_to_random_uniform(double minv, double maxv)
{
  // test helper of the reference (hiopVectorPar.cpp:131): filled on the host, one upload
  std::vector<double> h((size_t)n_local_);
  std::mt19937_64 gen(20240916u);
  std::uniform_real_distribution<double> dist(minv, maxv);
  for(auto& v : h) v = dist(gen);
  hiopamd_ok(hiopamd_copy_h2d(C_, data_, h.data(), sizeof(double) * h.size()));
}
void hiopVectorHipNative::setToConstant_w_patternSelect(double c, const hiopVector& select)
{
  hiopamd_ok(hiopamd_vec_set_to_constant_w_pattern(C_, N_, data_, c, dev(select)));
}
void hiopVectorHipNative::copyFrom(const hiopVector& vec)
{
  assert(vec.get_local_size() == n_local_);
  hiopamd_ok(hiopamd_vec_copy(C_, N_, data_, dev(vec)));
}
void hiopVectorHipNative::copyFrom(const double* local_array)
{
  if(local_array) hiopamd_ok(hiopamd_vec_copy(C_, N_, data_, local_array));
}
void hiopVectorHipNative::copy_from_w_pattern(const hiopVector& src, const hiopVector& select)
{
  hiopamd_ok(hiopamd_vec_copy_from_w_pattern(C_, N_, data_, dev(src), dev(select)));
}
void hiopVectorHipNative::copyFromStarting(int start_index_in_this, const double* v, int nv)
{
  assert(start_index_in_this + nv <= n_local_);
  hiopamd_ok(hiopamd_vec_copy(C_, nv, data_ + start_index_in_this, v));
}
void hiopVectorHipNative::copyFromStarting(int start_index, const hiopVector& src)
{
  assert(start_index + src.get_local_size() <= n_local_);
  hiopamd_ok(hiopamd_vec_copy(C_, src.get_local_size(), data_ + start_index, dev(src)));
}
void hiopVectorHipNative::copy_from_starting_at(const double* v, int start_index_in_v, int n)
{
  hiopamd_ok(hiopamd_vec_copy(C_, n, data_, v + start_index_in_v));
}
void hiopVectorHipNative::copy_from_vectorpar(const hiopVectorPar& vsrc)
{
  assert(vsrc.get_local_size() == n_local_);
  hiopamd_ok(hiopamd_copy_h2d(C_, data_, vsrc.local_data_const(), sizeof(double) * (size_t)n_local_));
}
void hiopVectorHipNative::copy_from_indexes(const hiopVector& src, const hiopVectorInt& index_in_src)
{
  assert(index_in_src.get_local_size() == n_local_);
  hiopamd_ok(hiopamd_vec_copy_from_indexes(C_, N_, data_, dev(src), index_in_src.local_data_const()));
}
void hiopVectorHipNative::copy_from_indexes(const double* src, const hiopVectorInt& index_in_src)
{
  hiopamd_ok(hiopamd_vec_copy_from_indexes(C_, N_, data_, src, index_in_src.local_data_const()));
}
void hiopVectorHipNative::startingAtCopyFromStartingAt(int start_idx_dest, const hiopVector& v, int start_idx_src)
{
  hiopamd_ok(hiopamd_vec_starting_at_copy_from_starting_at(C_, data_, N_, start_idx_dest, dev(v), v.get_local_size(), start_idx_src));
}
void hiopVectorHipNative::copyTo(double* dest) const { hiopamd_ok(hiopamd_vec_copy(C_, N_, dest, data_)); }
void hiopVectorHipNative::copy_to_vectorpar(hiopVectorPar& vdest) const
{
  assert(vdest.get_local_size() == n_local_);
  hiopamd_ok(hiopamd_copy_d2h(C_, vdest.local_data(), data_, sizeof(double) * (size_t)n_local_));
}
void hiopVectorHipNative::copyToStarting(int start_index, hiopVector& dst) const
{
  // "this" starting at start_index into dst from 0 (hiopVectorPar.cpp:287-296)
  assert(start_index + dst.get_local_size() <= n_local_);
  hiopamd_ok(hiopamd_vec_copy(C_, dst.get_local_size(), dev(dst), data_ + start_index));
}
void hiopVectorHipNative::copyToStarting(hiopVector& vec, int start_index_in_dest) const
{
  assert(start_index_in_dest + n_local_ <= vec.get_local_size());
  hiopamd_ok(hiopamd_vec_copy(C_, N_, dev(vec) + start_index_in_dest, data_));
}
void hiopVectorHipNative::copyToStartingAt_w_pattern(hiopVector& vec, index_type start_index_in_dest, const hiopVector& ix) const
{
  int64_t nnz = 0;
  hiopamd_ok(hiopamd_vec_copy_to_starting_at_w_pattern(C_, N_, data_, dev(vec), start_index_in_dest, dev(ix), &nnz));
}
void hiopVectorHipNative::copy_from_two_vec_w_pattern(const hiopVector& c, const hiopVectorInt& c_map, const hiopVector& d,
                                                      const hiopVectorInt& d_map)
{
  hiopamd_ok(hiopamd_vec_copy_from_two_vec_w_pattern(C_, data_, dev(c), c_map.local_data_const(), c.get_local_size(), dev(d),
                                                     d_map.local_data_const(), d.get_local_size()));
}
void hiopVectorHipNative::copy_to_two_vec_w_pattern(hiopVector& c, const hiopVectorInt& c_map, hiopVector& d,
                                                    const hiopVectorInt& d_map) const
{
  hiopamd_ok(hiopamd_vec_copy_to_two_vec_w_pattern(C_, data_, dev(c), c_map.local_data_const(), c.get_local_size(), dev(d),
                                                   d_map.local_data_const(), d.get_local_size()));
}
void hiopVectorHipNative::startingAtCopyToStartingAt(index_type start_idx_in_src, hiopVector& dest, index_type start_idx_dest,
                                                     size_type num_elems) const
{
  hiopamd_ok(hiopamd_vec_starting_at_copy_to_starting_at(C_, data_, N_, start_idx_in_src, dev(dest), dest.get_local_size(), start_idx_dest, num_elems));
}
void hiopVectorHipNative::startingAtCopyToStartingAt_w_pattern(index_type start_idx_in_src, hiopVector& dest,
                                                               index_type start_idx_dest, const hiopVector& selec_dest,
                                                               size_type num_elems) const
{
  hiopamd_ok(hiopamd_vec_starting_at_copy_to_starting_at_w_pattern(C_, data_, start_idx_in_src, dev(dest), dest.get_local_size(),
                                                                   start_idx_dest, dev(selec_dest), num_elems));
}

double hiopVectorHipNative::twonorm() const
{
  double nrm = 0.0;
  if(comm_size_ == 1) {
    hiopamd_ok(hiopamd_vec_twonorm(C_, N_, data_, &nrm));
    return nrm;
  }
  hiopamd_ok(hiopamd_vec_dot(C_, N_, data_, data_, &nrm));   // local sum of squares, then the all-reduce (:463-478)
  return std::sqrt(reduce_sum(nrm));
}
double hiopVectorHipNative::infnorm() const { return reduce_max(infnorm_local()); }
double hiopVectorHipNative::infnorm_local() const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_infnorm(C_, N_, data_, &v));
  return v;
}
double hiopVectorHipNative::onenorm() const { return reduce_sum(onenorm_local()); }
double hiopVectorHipNative::onenorm_local() const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_onenorm(C_, N_, data_, &v));
  return v;
}
void hiopVectorHipNative::componentMult(const hiopVector& vec) { hiopamd_ok(hiopamd_vec_component_mult(C_, N_, data_, dev(vec))); }
void hiopVectorHipNative::componentDiv(const hiopVector& vec) { hiopamd_ok(hiopamd_vec_component_div(C_, N_, data_, dev(vec))); }
void hiopVectorHipNative::componentDiv_w_selectPattern(const hiopVector& vec, const hiopVector& select)
{
  hiopamd_ok(hiopamd_vec_component_div_w_pattern(C_, N_, data_, dev(vec), dev(select)));
}
void hiopVectorHipNative::component_min(const double constant) { hiopamd_ok(hiopamd_vec_component_min_c(C_, N_, data_, constant)); }
void hiopVectorHipNative::component_min(const hiopVector& vec) { hiopamd_ok(hiopamd_vec_component_min_v(C_, N_, data_, dev(vec))); }
void hiopVectorHipNative::component_max(const double constant) { hiopamd_ok(hiopamd_vec_component_max_c(C_, N_, data_, constant)); }
void hiopVectorHipNative::component_max(const hiopVector& v) { hiopamd_ok(hiopamd_vec_component_max_v(C_, N_, data_, dev(v))); }
void hiopVectorHipNative::component_abs() { hiopamd_ok(hiopamd_vec_component_abs(C_, N_, data_)); }
void hiopVectorHipNative::component_sgn() { hiopamd_ok(hiopamd_vec_component_sgn(C_, N_, data_)); }
void hiopVectorHipNative::component_sqrt() { hiopamd_ok(hiopamd_vec_component_sqrt(C_, N_, data_)); }
void hiopVectorHipNative::scale(double c) { hiopamd_ok(hiopamd_vec_scale(C_, N_, data_, c)); }
void hiopVectorHipNative::axpy(double alpha, const hiopVector& xvec) { hiopamd_ok(hiopamd_vec_axpy(C_, N_, data_, alpha, dev(xvec))); }
void hiopVectorHipNative::axpy_w_pattern(double alpha, const hiopVector& xvec, const hiopVector& select)
{
  hiopamd_ok(hiopamd_vec_axpy_w_pattern(C_, N_, data_, alpha, dev(xvec), dev(select)));
}
void hiopVectorHipNative::axpy(double alpha, const hiopVector& xvec, const hiopVectorInt& i)
{
  assert(xvec.get_local_size() == i.get_local_size());
  hiopamd_ok(hiopamd_vec_axpy_w_map(C_, i.get_local_size(), data_, alpha, dev(xvec), i.local_data_const()));
}
void hiopVectorHipNative::axzpy(double alpha, const hiopVector& xvec, const hiopVector& zvec)
{
  hiopamd_ok(hiopamd_vec_axzpy(C_, N_, data_, alpha, dev(xvec), dev(zvec)));
}
void hiopVectorHipNative::axdzpy(double alpha, const hiopVector& xvec, const hiopVector& zvec)
{
  hiopamd_ok(hiopamd_vec_axdzpy(C_, N_, data_, alpha, dev(xvec), dev(zvec)));
}
void hiopVectorHipNative::axdzpy_w_pattern(double alpha, const hiopVector& xvec, const hiopVector& zvec, const hiopVector& select)
{
  hiopamd_ok(hiopamd_vec_axdzpy_w_pattern(C_, N_, data_, alpha, dev(xvec), dev(zvec), dev(select)));
}
void hiopVectorHipNative::addConstant(double c) { hiopamd_ok(hiopamd_vec_add_constant(C_, N_, data_, c)); }
void hiopVectorHipNative::addConstant_w_patternSelect(double c, const hiopVector& select)
{
  hiopamd_ok(hiopamd_vec_add_constant_w_pattern(C_, N_, data_, c, dev(select)));
}
double hiopVectorHipNative::dotProductWith(const hiopVector& vec) const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_dot(C_, N_, data_, dev(vec), &v));
  return reduce_sum(v);
}
void hiopVectorHipNative::negate() { hiopamd_ok(hiopamd_vec_negate(C_, N_, data_)); }
void hiopVectorHipNative::invert() { hiopamd_ok(hiopamd_vec_invert(C_, N_, data_)); }
double hiopVectorHipNative::logBarrier_local(const hiopVector& select) const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_log_barrier(C_, N_, data_, dev(select), &v));
  return v;
}
void hiopVectorHipNative::addLogBarrierGrad(double alpha, const hiopVector& xvec, const hiopVector& select)
{
  hiopamd_ok(hiopamd_vec_add_log_barrier_grad(C_, N_, data_, alpha, dev(xvec), dev(select)));
}
double hiopVectorHipNative::sum_local() const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_sum(C_, N_, data_, &v));
  return v;
}
double hiopVectorHipNative::linearDampingTerm_local(const hiopVector& ixleft, const hiopVector& ixright, const double& mu,
                                                    const double& kappa_d) const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_linear_damping_term(C_, N_, data_, dev(ixleft), dev(ixright), mu, kappa_d, &v));
  return v;
}
void hiopVectorHipNative::addLinearDampingTerm(const hiopVector& ixleft, const hiopVector& ixright, const double& alpha,
                                               const double& ct)
{
  hiopamd_ok(hiopamd_vec_add_linear_damping_term(C_, N_, data_, dev(ixleft), dev(ixright), alpha, ct));
}
int hiopVectorHipNative::allPositive()
{
  int loc = 0;
  hiopamd_ok(hiopamd_vec_all_positive(C_, N_, data_, &loc));
#ifdef HIOP_USE_MPI
  int g;
  int ierr = MPI_Allreduce(&loc, &g, 1, MPI_INT, MPI_MIN, comm_);
  assert(MPI_SUCCESS == ierr);
  return g;
#else
  return loc;
#endif
}
int hiopVectorHipNative::allPositive_w_patternSelect(const hiopVector& select)
{
  int loc = 0;
  hiopamd_ok(hiopamd_vec_all_positive_w_pattern(C_, N_, data_, dev(select), &loc));
#ifdef HIOP_USE_MPI
  int g;
  int ierr = MPI_Allreduce(&loc, &g, 1, MPI_INT, MPI_MIN, comm_);
  assert(MPI_SUCCESS == ierr);
  return g;
#else
  return loc;
#endif
}
double hiopVectorHipNative::min() const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_min(C_, N_, data_, &v));
  return reduce_min(v);
}
double hiopVectorHipNative::min_w_pattern(const hiopVector& select) const
{
  double v = 0.0;
  hiopamd_ok(hiopamd_vec_min_w_pattern(C_, N_, data_, dev(select), &v));
  return reduce_min(v);
}
void hiopVectorHipNative::min(double& minval, int& index) const
{
  (void)minval;
  (void)index;
  assert(false && "not implemented");   // as in the reference (hiopVectorPar.cpp:841)
}
bool hiopVectorHipNative::projectIntoBounds_local(const hiopVector& xlo, const hiopVector& ixl, const hiopVector& xup,
                                                  const hiopVector& ixu, double kappa1, double kappa2)
{
  int ok = 0;
  hiopamd_ok(hiopamd_vec_project_into_bounds(C_, N_, data_, dev(xlo), dev(ixl), dev(xup), dev(ixu), kappa1, kappa2, &ok));
  return ok != 0;
}
double hiopVectorHipNative::fractionToTheBdry_local(const hiopVector& dvec, const double& tau) const
{
  double v = 1.0;
  hiopamd_ok(hiopamd_vec_fraction_to_the_bdry(C_, N_, data_, dev(dvec), tau, &v));
  return v;
}
double hiopVectorHipNative::fractionToTheBdry_w_pattern_local(const hiopVector& dvec, const double& tau,
                                                              const hiopVector& select) const
{
  double v = 1.0;
  hiopamd_ok(hiopamd_vec_fraction_to_the_bdry_w_pattern(C_, N_, data_, dev(dvec), tau, dev(select), &v));
  return v;
}
void hiopVectorHipNative::selectPattern(const hiopVector& select) { hiopamd_ok(hiopamd_vec_select_pattern(C_, N_, data_, dev(select))); }
bool hiopVectorHipNative::matchesPattern(const hiopVector& select)
{
  int loc = 0;
  hiopamd_ok(hiopamd_vec_matches_pattern(C_, N_, data_, dev(select), &loc));
#ifdef HIOP_USE_MPI
  int g;
  int ierr = MPI_Allreduce(&loc, &g, 1, MPI_INT, MPI_MIN, comm_);
  assert(MPI_SUCCESS == ierr);
  return g != 0;
#else
  return loc != 0;
#endif
}
void hiopVectorHipNative::adjustDuals_plh(const hiopVector& xvec, const hiopVector& ixvec, const double& mu, const double& kappa)
{
  hiopamd_ok(hiopamd_vec_adjust_duals_plh(C_, N_, data_, dev(xvec), dev(ixvec), mu, kappa));
}
bool hiopVectorHipNative::is_zero() const
{
  int loc = 0;
  hiopamd_ok(hiopamd_vec_is_zero(C_, N_, data_, &loc));
#ifdef HIOP_USE_MPI
  int g;
  int ierr = MPI_Allreduce(&loc, &g, 1, MPI_INT, MPI_MIN, comm_);
  assert(MPI_SUCCESS == ierr);
  return g != 0;
#else
  return loc != 0;
#endif
}
bool hiopVectorHipNative::isnan_local() const
{
  int v = 0;
  hiopamd_ok(hiopamd_vec_isnan(C_, N_, data_, &v));
  return v != 0;
}
bool hiopVectorHipNative::isinf_local() const
{
  int v = 0;
  hiopamd_ok(hiopamd_vec_isinf(C_, N_, data_, &v));
  return v != 0;
}
bool hiopVectorHipNative::isfinite_local() const
{
  int v = 0;
  hiopamd_ok(hiopamd_vec_isfinite(C_, N_, data_, &v));
  return v != 0;
}
void hiopVectorHipNative::print(FILE* file, const char* message, int max_elems, int rank) const
{
  int myrank = 0, numranks = 1;
  if(nullptr == file) file = stdout;
#ifdef HIOP_USE_MPI
  if(rank >= 0) {
    int err = MPI_Comm_rank(comm_, &myrank);
    assert(err == MPI_SUCCESS);
    err = MPI_Comm_size(comm_, &numranks);
    assert(err == MPI_SUCCESS);
  }
#endif
  if(myrank == rank || rank == -1) {
    copyFromDev();
    if(max_elems > n_local_) max_elems = n_local_;
    if(nullptr == message) {
      std::fprintf(file, "vector of size %d, printing %d elems (on rank=%d of %d)\n", (int)n_, max_elems < 0 ? (int)n_local_ : max_elems,
                   myrank, numranks);
    } else {
      std::fprintf(file, "%s ", message);
    }
    std::fprintf(file, "=[");
    max_elems = max_elems >= 0 ? max_elems : n_local_;
    for(int it = 0; it < max_elems; it++) std::fprintf(file, "%22.16e ; ", host_mirror_[it]);
    std::fprintf(file, "];\n");
  }
}
hiopVector* hiopVectorHipNative::alloc_clone() const
{
  hiopVectorHipNative* v = new hiopVectorHipNative(n_, nullptr, comm_);
  // same partition as this (the constructor derives one from col_part only): rebuild with the local extents
  if(v->n_local_ != n_local_) {
    hiopamd_ok(hiopamd_free(v->data_));
    v->glob_il_ = glob_il_;
    v->glob_iu_ = glob_iu_;
    v->n_local_ = n_local_;
    v->data_ = hiopamd_new_array((size_t)n_local_);
  }
  return v;
}
hiopVector* hiopVectorHipNative::new_copy() const
{
  hiopVector* v = alloc_clone();
  v->copyFrom(*this);
  return v;
}
double* hiopVectorHipNative::local_data_host()
{
  if(!host_mirror_) host_mirror_ = new double[n_local_ > 0 ? n_local_ : 1];
  return host_mirror_;
}
const double* hiopVectorHipNative::local_data_host_const() const
{
  if(!host_mirror_) host_mirror_ = new double[n_local_ > 0 ? n_local_ : 1];
  return host_mirror_;
}
void hiopVectorHipNative::copyToDev() const
{
  if(host_mirror_) hiopamd_ok(hiopamd_copy_h2d(ctx_, data_, host_mirror_, sizeof(double) * (size_t)n_local_));
}
void hiopVectorHipNative::copyFromDev() const
{
  (void)local_data_host_const();
  hiopamd_ok(hiopamd_copy_d2h(ctx_, host_mirror_, data_, sizeof(double) * (size_t)n_local_));
}
size_type hiopVectorHipNative::numOfElemsLessThan(const double& val) const
{
  int64_t cnt = 0;
  hiopamd_ok(hiopamd_vec_num_elems_less_than(C_, N_, data_, val, &cnt));
#ifdef HIOP_USE_MPI
  size_type loc = (size_type)cnt, g;
  int ierr = MPI_Allreduce(&loc, &g, 1, MPI_HIOP_SIZE_TYPE, MPI_SUM, comm_);
  assert(MPI_SUCCESS == ierr);
  return g;
#else
  return (size_type)cnt;
#endif
}
size_type hiopVectorHipNative::numOfElemsAbsLessThan(const double& val) const
{
  int64_t cnt = 0;
  hiopamd_ok(hiopamd_vec_num_elems_abs_less_than(C_, N_, data_, val, &cnt));
#ifdef HIOP_USE_MPI
  size_type loc = (size_type)cnt, g;
  int ierr = MPI_Allreduce(&loc, &g, 1, MPI_HIOP_SIZE_TYPE, MPI_SUM, comm_);
  assert(MPI_SUCCESS == ierr);
  return g;
#else
  return (size_type)cnt;
#endif
}
void hiopVectorHipNative::set_array_from_to(hiopInterfaceBase::NonlinearityType* arr, const int start, const int end,
                                            const hiopInterfaceBase::NonlinearityType* arr_src, const int start_src) const
{
  assert(end <= n_local_ && start <= end && start >= 0 && start_src >= 0);
  for(int i = start; i < end; i++) arr[i] = arr_src[start_src + i - start];   // host arrays (hiopVectorPar.cpp:1262-1271)
}
void hiopVectorHipNative::set_array_from_to(hiopInterfaceBase::NonlinearityType* arr, const int start, const int end,
                                            const hiopInterfaceBase::NonlinearityType arr_src) const
{
  assert(end <= n_local_ && start <= end && start >= 0);
  for(int i = start; i < end; i++) arr[i] = arr_src;
}
bool hiopVectorHipNative::is_equal(const hiopVector& vec) const
{
  int eq = 0;
  hiopamd_ok(hiopamd_vec_is_equal(C_, N_, data_, dev(vec), &eq));
  return eq != 0;
}
}  // namespace hiop
