// The C interfaces of the solvers (include/hiop_amd_interface.h): the reference's C FFI for MDS problems
// (src/Interface/hiopInterface.h:63-98, chiopInterface.cpp:64-95) and for dense-constraint problems (hiopInterface.h:150-176,
// chiopInterface.cpp:129-159: quasi-Newton, hiopAlgFilterIPMQuasiNewton::run :960-1480) in front of this library's device path.
//
// What is here is HOST control flow only — the counterpart of
//   hiopNlpMDS / hiopNlpFormulation::finalizeInitialization   src/Optimization/hiopNlpFormulation.cpp:205-700 (equality / inequality
//                                                              split, bound patterns, hiopBoundsRelaxer)
//   hiopAlgFilterIPMBase::startingProcedure                    src/Optimization/hiopAlgFilterIPM.cpp:290-425
//   hiopAlgFilterIPMNewton::run                                :2101-2770 (evalNlpAndLogErrors :636, checkTermination :814,
//                                                              update_log_barrier_params :556, accept_line_search_conditions :2852,
//                                                              apply_second_order_correction :2949, outputIteration :2783)
//   hiopDualsNewtonLinearUpdate::go                            src/Optimization/hiopDualsUpdater.hpp:412-431
// every numerical step is a call into the device layer of hiop_amd.h (hiopamd_residual_update, hiopamd_kkt_xycyd_update /
// _compute_directions_w_IR, hiopamd_iterate_*, the vector kernels).  The iterate, trial iterate, direction and residual are the
// 12-part slabs of that layer.
#include "device_utils.hpp"

#include "../../include/hiop_amd_interface.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

using namespace hiopamd;

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

namespace {

// hiopSolveStatus (src/Interface/hiopInterface.hpp:65-100), the values the reference returns
enum SolveStatus {
  Solve_Success = 0,
  Solve_Success_RelTol = 1,
  Solve_Acceptable_Level = 2,
  Max_Iter_Exceeded = 10,
  Steplength_Too_Small = -4,
  Err_Step_Computation = -5,
  Invalid_Problem_Definition = -11,
  Error_In_User_Function = -15,
  NlpSolve_SolveNotCalled = -10002,
  NlpSolve_Pending = -10003
};

struct Options {   // defaults of src/Utils/hiopOptions.cpp:560-850; mu0 as set by hiop_mds_create_problem (chiopInterface.cpp:75)
  double mu0 = 1e-1, tolerance = 1e-8, kappa_mu = 0.2, theta_mu = 1.5, kappa_eps = 10., tau_min = 0.99, kappa1 = 1e-2,
         kappa2 = 1e-2, smax = 100., kappa_d = 1e-5, eta_phi = 1e-8, gamma_theta = 1e-5, gamma_phi = 1e-8, s_theta = 1.1,
         s_phi = 2.3, delta = 1., theta_max_fact = 1e4, theta_min_fact = 1e-4, dual_tol = 1., cons_tol = 1e-4, comp_tol = 1e-4,
         rel_tolerance = 0., acceptable_tolerance = 1e-6, min_step_size = 1e-16, kappa_Sigma = 1e10, bound_relax_perturb = 1e-8,
         kappa_soc = 0.99, scaling_max_grad = 100., ir_outer_tol_factor = 1e-2, ir_outer_tol_min = 1e-6;
  int acceptable_iterations = 10, max_iter = 3000, max_soc_iter = 4, verbosity_level = 3, ir_outer_maxit = 8;
  // 0 (default): after a step accepted through the second-order correction the filter entry keeps the FIRST trial point's theta — the
  // reference passes theta_trial to apply_second_order_correction by value (hiopAlgFilterIPM.cpp:2949-2973, call :2561-2570), so its
  // filter.add (:2629-2639) never sees the corrected point's; 1: the corrected point's theta (rounds 3-4 of this library)
  int soc_theta_corrected = 0;
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count)
  {
    n = count;
    HIOPAMD_CHECK(hipMalloc((void**)&p, sizeof(T) * std::max<size_t>(count, 1)));
    return HIOPAMD_OK;
  }
  ~DevBuf() { (void)hipFree(p); }
};

struct Filter {   // src/Optimization/hiopFilter.hpp:60-75, .cpp:55-68
  std::vector<std::pair<double, double>> e;
  void initialize(double theta_max)
  {
    e.clear();
    e.emplace_back(theta_max, -1e20);
  }
  void add(double theta, double phi) { e.insert(e.begin(), std::make_pair(theta, phi)); }
  bool contains(double theta, double phi) const
  {
    for(const auto& q : e)
      if(theta >= q.first && phi >= q.second) return true;
    return false;
  }
};

struct MdsSolver {
  cHiopMDSProblem* prob = nullptr;
  cHiopDenseProblem* dprob = nullptr;   // != nullptr: the dense-constraints quasi-Newton variant
  hiopamd_hess_lowrank* hess = nullptr;
  hiopamd_kkt_lowrank* klr = nullptr;
  DevBuf<double> d_Jall, d_J;           // dense variant: the user's m x n Jacobian, and [Jc; Jd] in one block
  // Everything the caller can SET between create and solve lives in this one struct: a second solve on the same problem object rebuilds
  // the solver state from the callbacks and carries `user` over wholesale (resolve_from_scratch) — a setter added later cannot be forgotten there.
  struct UserSettings {
    Options o;
    bool dev_cb = false;
    int secant_memory_len = 6;
    double sigma0 = 1.0;
    double scaling_min_grad = 1e-8;
  } user;
  Options& o = user.o;
  bool& dev_cb = user.dev_cb;
  int& secant_memory_len = user.secant_memory_len;
  double& sigma0 = user.sigma0;
  double& scaling_min_grad = user.scaling_min_grad;
  hiopamd_ctx* ctx = nullptr;
  hiopamd_kkt_mds* kkt = nullptr;
  hiopamd_kkt_xycyd* full = nullptr;
  int n = 0, m = 0, ns = 0, nd = 0, nnzJ = 0, nnzJeq = 0, nnzJineq = 0, nnzH = 0, neq = 0, nineq = 0;
  int64_t off[13] = {0};
  int64_t dim = 0;
  // host copies of the problem description
  std::vector<double> xl, xu, cl, cu;
  std::vector<int> eq_map, ineq_map, iJ, jJ, iH, jH;
  // device: problem data
  DevBuf<double> d_xl, d_xu, d_dl, d_du, d_crhs, d_ixl, d_ixu, d_idl, d_idu;
  // gradient-based NLP scaling (hiopNLPObjGradScaling, hiopNlpTransforms.cpp:423-499): objective factor, one factor per constraint row
  // ([equalities; inequalities] in the solver's order), decided at the user's starting point (hiopNlpFormulation.cpp:671-714)
  bool scaled = false;
  double s_f = 1.0;
  bool solved_once = false;   // hiop_*_solve_problem ran: the next call starts over from the user's data (resolve_from_scratch)
  DevBuf<double> d_scal;   // neq + nineq
  DevBuf<int> d_eq_map, d_ineq_map, d_jc_src, d_jd_src, d_Jcs_i, d_Jcs_j, d_Jds_i, d_Jds_j, d_Hss_i, d_Hss_j;
  std::vector<int> h_Jcs_i, h_Jcs_j, h_Jds_i, h_Jds_j;
  // device: values of the current iterate (what hiopamd_kkt_mds_set_values borrows)
  DevBuf<double> d_MJ, d_JacD, d_Jcs_v, d_Jds_v, d_Jcd, d_Jdd, d_MH, d_HDD, d_lambda;
  // device: function values
  DevBuf<double> d_grad, d_cons, d_c, d_d, d_cons_t, d_c_t, d_d_t, d_gx, d_gd, d_csoc, d_dsoc, d_tmpc, d_tmpd;
  // device: slabs
  DevBuf<double> it, trial, dir, dir_soc, resid, resid_soc;
  // host staging for host-space callbacks
  std::vector<double> h_x, h_buf;
  // results
  int status = NlpSolve_SolveNotCalled, iters = 0, nfact = 0;
  double t_total = 0.0, t_kkt = 0.0;   // seconds inside run(); inside the KKT span (update + directions: the reference's runStats.kkt.tmTotal)

  ~MdsSolver()
  {
    if(full) hiopamd_kkt_xycyd_destroy(full);
    if(kkt) hiopamd_kkt_mds_destroy(kkt);
    if(klr) hiopamd_kkt_lowrank_destroy(klr);
    if(hess) hiopamd_hess_lowrank_destroy(hess);
    if(ctx) hiopamd_ctx_destroy(ctx);
  }

  // ---- staging ------------------------------------------------------------------------------------------------------
  int h2d(void* dst, const void* src, size_t bytes) { return hiopamd_copy_h2d(ctx, dst, src, bytes); }
  int d2h(void* dst, const void* src, size_t bytes)
  {
    RC(hiopamd_copy_d2h(ctx, dst, src, bytes));
    return hiopamd_ctx_sync(ctx);
  }
  double* part(DevBuf<double>& slab, int p) { return slab.p + off[p]; }

  // x for a callback: the device pointer itself, or a host copy
  int x_for_cb(const double* x_dev, double** out)
  {
    if(dev_cb) {
      RC(hiopamd_ctx_sync(ctx));   // the callback runs on its own stream / the host: everything queued on ours must have landed
      *out = const_cast<double*>(x_dev);
      return HIOPAMD_OK;
    }
    h_x.resize((size_t)n);
    RC(d2h(h_x.data(), x_dev, sizeof(double) * (size_t)n));
    *out = h_x.data();
    return HIOPAMD_OK;
  }

  // The return values of the user's callbacks are IGNORED, as by the reference's wrappers cppUserProblemMDS / cppUserProblemDense
  // (src/Interface/chiopInterface.hpp:125-214 call the C function and return true whatever it returned): a binding whose callbacks
  // return 1, "true" or nothing in particular behaves here as it does there.

  // eval_f (+ eval_cons): hiopAlgFilterIPMBase::evalNlp_funcOnly, hiopAlgFilterIPM.cpp:714-731
  int eval_func(const double* x_dev, double* f, double* cons_dev, double* c_dev, double* d_dev)
  {
    double* x = nullptr;
    RC(x_for_cb(x_dev, &x));
    void* ud = dprob ? dprob->user_data : prob->user_data;
    auto cb_f = dprob ? dprob->eval_f : prob->eval_f;
    auto cb_c = dprob ? dprob->eval_cons : prob->eval_cons;
    (void)cb_f(n, x, 1, f, ud);
    if(dev_cb) {
      (void)cb_c(n, m, x, 0, cons_dev, ud);
    } else {
      h_buf.resize((size_t)std::max(m, 1));
      (void)cb_c(n, m, x, 0, h_buf.data(), ud);
      RC(h2d(cons_dev, h_buf.data(), sizeof(double) * (size_t)m));
    }
    // c = cons[eq], d = cons[ineq]   (hiopNlpFormulation::eval_c_d, hiopNlpFormulation.cpp:1045-1075)
    RC(hiopamd_vec_copy_from_indexes(ctx, neq, c_dev, cons_dev, d_eq_map.p));
    RC(hiopamd_vec_copy_from_indexes(ctx, nineq, d_dev, cons_dev, d_ineq_map.p));
    if(scaled) {   // apply_to_obj / apply_to_cons_eq / apply_to_cons_ineq (hiopNlpTransforms.hpp:389, 413-433)
      *f *= s_f;
      if(neq) RC(hiopamd_vec_component_mult(ctx, neq, c_dev, d_scal.p));
      if(nineq) RC(hiopamd_vec_component_mult(ctx, nineq, d_dev, d_scal.p + neq));
    }
    return HIOPAMD_OK;
  }

  // gradient, Jacobian, Hessian of the Lagrangian at the iterate: evalNlp_derivOnly, hiopAlgFilterIPM.cpp:733-762
  int eval_deriv(const double* x_dev, const double* yc, const double* yd)
  {
    double* x = nullptr;
    RC(x_for_cb(x_dev, &x));
    if(dprob) {
      // gradient and the dense m x n Jacobian; rows split into [Jc; Jd] (hiopNlpDenseConstraints::eval_Jac_c_d,
      // hiopNlpFormulation.cpp:1480-1530).  The Hessian is the secant approximation: nothing to evaluate.
      const size_t szJ = (size_t)m * (size_t)n;
      if(dev_cb) {
        RC(hiopamd_ctx_sync(ctx));
        (void)dprob->eval_grad_f(n, x, 0, d_grad.p, dprob->user_data);
        (void)dprob->eval_Jac_cons(n, m, x, 0, d_Jall.p, dprob->user_data);
      } else {
        std::vector<double> g((size_t)n), jj(std::max<size_t>(szJ, 1));
        (void)dprob->eval_grad_f(n, x, 0, g.data(), dprob->user_data);
        (void)dprob->eval_Jac_cons(n, m, x, 0, jj.data(), dprob->user_data);
        RC(h2d(d_grad.p, g.data(), sizeof(double) * (size_t)n));
        RC(h2d(d_Jall.p, jj.data(), sizeof(double) * szJ));
        RC(hiopamd_ctx_sync(ctx));
      }
      RC(hiopamd_mat_copy_rows_from_idx(ctx, neq, n, d_J.p, n, d_Jall.p, n, d_eq_map.p));
      RC(hiopamd_mat_copy_rows_from_idx(ctx, nineq, n, d_J.p + (size_t)neq * n, n, d_Jall.p, n, d_ineq_map.p));
      if(scaled) {   // apply_to_grad_obj, apply_to_jacob_eq / _ineq (hiopNlpTransforms.hpp:399-403, 436-470)
        RC(hiopamd_vec_scale(ctx, n, d_grad.p, s_f));
        if(m) RC(hiopamd_mat_scale_rows(ctx, m, n, d_J.p, n, d_scal.p, 0));
      }
      RC(hiopamd_kkt_xycyd_set_matrices(full, nullptr, d_J.p, d_J.p + (size_t)neq * n));
      return HIOPAMD_OK;
    }
    const size_t szJD = (size_t)m * (size_t)nd, szHD = (size_t)nd * (size_t)nd;
    // lambda in the user's constraint order (hiopNlpMDS::eval_Hess_Lagr, hiopNlpFormulation.cpp:1757-1800)
    {
      double* lam = d_lambda.p;
      const int *em = d_eq_map.p, *im = d_ineq_map.p;
      const int ne = neq;
      const double* sc = scaled ? d_scal.p : nullptr;   // the user's Hessian is evaluated with the multipliers of the UNSCALED constraints
      RC(launch_ew(ctx, (int64_t)m, [=] __device__(int64_t i) {
        const double w = sc ? sc[i] : 1.0;
        if(i < ne) lam[em[i]] = w * yc[i];
        else lam[im[i - ne]] = w * yd[i - ne];
      }));
    }
    if(dev_cb) {
      RC(hiopamd_ctx_sync(ctx));
      (void)prob->eval_grad_f(n, x, 0, d_grad.p, prob->user_data);
      (void)prob->eval_Jac_cons(n, m, x, 0, ns, nd, nnzJ, nullptr, nullptr, d_MJ.p, d_JacD.p, prob->user_data);
      (void)prob->eval_Hess_Lagr(n, m, x, 0, s_f, d_lambda.p, 1, ns, nd, nnzH, nullptr, nullptr, d_MH.p, d_HDD.p, 0, nullptr, nullptr,
                              nullptr, prob->user_data);
    } else {
      std::vector<double> g((size_t)n), mj((size_t)std::max(nnzJ, 1)), jd(std::max<size_t>(szJD, 1)), lam((size_t)std::max(m, 1)),
          mh((size_t)std::max(nnzH, 1)), hd(std::max<size_t>(szHD, 1));
      (void)prob->eval_grad_f(n, x, 0, g.data(), prob->user_data);
      (void)prob->eval_Jac_cons(n, m, x, 0, ns, nd, nnzJ, nullptr, nullptr, mj.data(), jd.data(), prob->user_data);
      RC(d2h(lam.data(), d_lambda.p, sizeof(double) * (size_t)m));
      (void)prob->eval_Hess_Lagr(n, m, x, 0, s_f, lam.data(), 1, ns, nd, nnzH, nullptr, nullptr, mh.data(), hd.data(), 0, nullptr,
                              nullptr, nullptr, prob->user_data);
      RC(h2d(d_grad.p, g.data(), sizeof(double) * (size_t)n));
      RC(h2d(d_MJ.p, mj.data(), sizeof(double) * (size_t)nnzJ));
      RC(h2d(d_JacD.p, jd.data(), sizeof(double) * szJD));
      RC(h2d(d_MH.p, mh.data(), sizeof(double) * (size_t)nnzH));
      RC(h2d(d_HDD.p, hd.data(), sizeof(double) * szHD));
      RC(hiopamd_ctx_sync(ctx));   // the host vectors above go out of scope
    }
    // split the rows of the one-call Jacobian into the equality and the inequality blocks
    // (hiopNlpMDS::eval_Jac_c_d, hiopNlpFormulation.cpp:1725-1755: copyRowsFrom with the two mappings)
    RC(hiopamd_vec_copy_from_indexes(ctx, nnzJeq, d_Jcs_v.p, d_MJ.p, d_jc_src.p));
    RC(hiopamd_vec_copy_from_indexes(ctx, nnzJineq, d_Jds_v.p, d_MJ.p, d_jd_src.p));
    RC(hiopamd_mat_copy_rows_from_idx(ctx, neq, nd, d_Jcd.p, nd, d_JacD.p, nd, d_eq_map.p));
    RC(hiopamd_mat_copy_rows_from_idx(ctx, nineq, nd, d_Jdd.p, nd, d_JacD.p, nd, d_ineq_map.p));
    if(scaled) {   // gradient and the four Jacobian blocks, row by row (the Hessian came out scaled through obj_factor and lambda)
      RC(hiopamd_vec_scale(ctx, n, d_grad.p, s_f));
      if(nnzJeq) RC(hiopamd_sp_scale_rows(ctx, nnzJeq, d_Jcs_i.p, d_Jcs_v.p, d_scal.p, 0));
      if(nnzJineq) RC(hiopamd_sp_scale_rows(ctx, nnzJineq, d_Jds_i.p, d_Jds_v.p, d_scal.p + neq, 0));
      if(neq && nd) RC(hiopamd_mat_scale_rows(ctx, neq, nd, d_Jcd.p, nd, d_scal.p, 0));
      if(nineq && nd) RC(hiopamd_mat_scale_rows(ctx, nineq, nd, d_Jdd.p, nd, d_scal.p + neq, 0));
    }
    RC(hiopamd_kkt_mds_set_values(kkt, d_Jcs_v.p, d_Jds_v.p, d_MH.p, d_Jcd.p, d_Jdd.p, d_HDD.p, nullptr, nullptr));
    return HIOPAMD_OK;
  }

  // ---- pieces of the loop -----------------------------------------------------------------------------------------------
  int logbar(DevBuf<double>& slab, double f, double mu, double* out)   // hiopLogBarProblem.hpp:94-113, :128-129
  {
    double lb = 0.0, damp = 0.0;
    RC(hiopamd_iterate_eval_log_barrier(full, slab.p, &lb));
    if(o.kappa_d > 0) RC(hiopamd_iterate_linear_damping_term(full, slab.p, mu, o.kappa_d, &damp));
    *out = f - mu * lb + damp;
    return HIOPAMD_OK;
  }

  int theta_of(DevBuf<double>& slab, const double* c, const double* d, double* out)   // hiopResidual.cpp:101-115
  {
    const double *crhs = d_crhs.p, *dd = part(slab, 1);
    double *tc = d_tmpc.p, *td = d_tmpd.p;
    RC(launch_ew(ctx, neq, [=] __device__(int64_t i) { tc[i] = crhs[i] - c[i]; }));
    RC(launch_ew(ctx, nineq, [=] __device__(int64_t i) { td[i] = dd[i] - d[i]; }));
    double a = 0.0, b = 0.0;
    ReduceBatch rb(ctx);   // both norms in one host round trip
    if(neq) RC(hiopamd_vec_onenorm(ctx, neq, tc, &a));
    if(nineq) RC(hiopamd_vec_onenorm(ctx, nineq, td, &b));
    RC(rb.flush());
    *out = a + b;
    return HIOPAMD_OK;
  }

  int grad_phi_dx(double mu, const double* dirp, double* out)   // hiopLogBarProblem.hpp:91-117, :149-156
  {
    double *gx = d_gx.p, *gd = d_gd.p;
    RC(hiopamd_vec_copy(ctx, n, gx, d_grad.p));
    HIOPAMD_CHECK(hipMemsetAsync(gd, 0, sizeof(double) * (size_t)std::max(nineq, 1), ctx->stream));
    RC(hiopamd_vec_add_log_barrier_grad(ctx, n, gx, -mu, part(it, 4), d_ixl.p));
    RC(hiopamd_vec_add_log_barrier_grad(ctx, n, gx, mu, part(it, 5), d_ixu.p));
    RC(hiopamd_vec_add_log_barrier_grad(ctx, nineq, gd, -mu, part(it, 6), d_idl.p));
    RC(hiopamd_vec_add_log_barrier_grad(ctx, nineq, gd, mu, part(it, 7), d_idu.p));
    if(o.kappa_d > 0) {
      RC(hiopamd_vec_add_linear_damping_term(ctx, n, gx, d_ixl.p, d_ixu.p, 1.0, o.kappa_d * mu));
      RC(hiopamd_vec_add_linear_damping_term(ctx, nineq, gd, d_idl.p, d_idu.p, 1.0, o.kappa_d * mu));
    }
    double a = 0.0, b = 0.0;
    ReduceBatch rb(ctx);
    RC(hiopamd_vec_dot(ctx, n, dirp + off[0], gx, &a));
    if(nineq) RC(hiopamd_vec_dot(ctx, nineq, dirp + off[1], gd, &b));
    RC(rb.flush());
    *out = a + b;
    return HIOPAMD_OK;
  }

  struct Errors {
    double optim, feas, complem, cons_violation, nlp, log;
  };

  int errors(const double* norms, Errors* e)   // evalNlpAndLogErrors, hiopAlgFilterIPM.cpp:636-712
  {
    double bou = 0.0, eq = 0.0;
    {
      double v[12] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      ReduceBatch rb(ctx);   // the six one-norms of the multipliers in one host round trip (they were six)
      for(int p = 8; p < 12; ++p)
        if(off[p + 1] > off[p]) RC(hiopamd_vec_onenorm(ctx, off[p + 1] - off[p], it.p + off[p], &v[p]));
      for(int p = 2; p < 4; ++p)
        if(off[p + 1] > off[p]) RC(hiopamd_vec_onenorm(ctx, off[p + 1] - off[p], it.p + off[p], &v[p]));
      RC(rb.flush());
      for(int p = 8; p < 12; ++p) bou += v[p];
      for(int p = 2; p < 4; ++p) eq += v[p];
    }
    const double ncomp = n_complem, mm = (double)m;
    double sd = std::fmax(o.smax, (bou + eq) / (ncomp + mm)) / o.smax;
    double sc = ncomp == 0 ? 0.0 : std::fmax(o.smax, bou / ncomp) / o.smax;
    sd = std::fmin(sd, 1e8);
    sc = std::fmin(sc, 1e8);
    e->optim = norms[0];
    e->feas = norms[1];
    e->complem = norms[2];
    e->cons_violation = norms[10];
    e->nlp = std::fmax(e->optim / sd, std::fmax(e->cons_violation, e->complem / sc));
    e->log = std::fmax(norms[3] / sd, std::fmax(e->cons_violation, norms[5] / sc));
    return HIOPAMD_OK;
  }
  double n_complem = 0;

  int accept(const Filter& filt, double theta, double theta_trial, double ap, double f_logbar, double f_logbar_trial,
             double theta_min, bool& gpd_computed, double& gpd, double mu, const double* dirp, int* st)   // :2852-2944
  {
    const bool suff = theta_trial <= (1 - o.gamma_theta) * theta || f_logbar_trial <= f_logbar - o.gamma_phi * theta;
    int s;
    if(theta >= theta_min) {
      s = suff ? 1 : 0;
    } else {
      if(!gpd_computed) {
        RC(grad_phi_dx(mu, dirp, &gpd));
        gpd_computed = true;
      }
      if(gpd < 0. && ap * std::pow(-gpd, o.s_phi) > o.delta * std::pow(theta, o.s_theta))
        s = (f_logbar_trial <= f_logbar + o.eta_phi * ap * gpd) ? 3 : 0;
      else s = suff ? 2 : 0;
    }
    if(s > 0 && filt.contains(theta_trial, f_logbar_trial)) s = 0;
    *st = s;
    return HIOPAMD_OK;
  }

  int trial_primals(const double* dirp, double ap, double ad, double mu, int* nadj)   // :2527-2528
  {
    RC(hiopamd_vec_copy(ctx, dim, trial.p, it.p));
    RC(hiopamd_iterate_take_step(full, trial.p, it.p, dirp, ap, ad, 1, 0));
    RC(hiopamd_iterate_determine_slacks(full, trial.p));   // compute_safe_slacks, hiopIterate.cpp:293-304
    RC(hiopamd_iterate_adjust_small_slacks(full, trial.p, it.p, mu, nadj));
    return HIOPAMD_OK;
  }

  // hiopNlpFormulation::apply_scaling (hiopNlpFormulation.cpp:671-714) + hiopNLPObjGradScaling's constructor
  // (hiopNlpTransforms.cpp:423-499) on the derivatives of the first evaluation (unscaled, in d_grad and the Jacobian blocks)
  int decide_scaling()
  {
    const double max_grad = o.scaling_max_grad;
    double g = 0.0, mc = 0.0, md = 0.0, v = 0.0;
    RC(hiopamd_vec_infnorm(ctx, n, d_grad.p, &g));
    RC(d_scal.alloc((size_t)std::max(m, 1)));
    DevBuf<double> tmp;
    RC(tmp.alloc((size_t)std::max(m, 1)));
    // row maxima of [Jc; Jd] into d_scal (hiopMatrix::row_max_abs_value)
    if(dprob) {
      if(m) RC(hiopamd_mat_row_max_abs(ctx, m, n, d_J.p, n, d_scal.p));
    } else {
      RC(hiopamd_vec_set_to_constant(ctx, m, d_scal.p, 0.0));
      RC(hiopamd_vec_set_to_constant(ctx, m, tmp.p, 0.0));
      if(nnzJeq) RC(hiopamd_sp_row_max_abs(ctx, neq, nnzJeq, d_Jcs_i.p, d_Jcs_v.p, d_scal.p));
      if(nnzJineq) RC(hiopamd_sp_row_max_abs(ctx, nineq, nnzJineq, d_Jds_i.p, d_Jds_v.p, d_scal.p + neq));
      if(neq && nd) RC(hiopamd_mat_row_max_abs(ctx, neq, nd, d_Jcd.p, nd, tmp.p));
      if(nineq && nd) RC(hiopamd_mat_row_max_abs(ctx, nineq, nd, d_Jdd.p, nd, tmp.p + neq));
      if(m) RC(hiopamd_vec_component_max_v(ctx, m, d_scal.p, tmp.p));
    }
    if(neq) RC(hiopamd_vec_infnorm(ctx, neq, d_scal.p, &mc));
    if(nineq) RC(hiopamd_vec_infnorm(ctx, nineq, d_scal.p + neq, &md));
    if(g < max_grad && mc < max_grad && md < max_grad) {   // :691-696: nothing to scale
      scaled = false;
      s_f = 1.0;
      return HIOPAMD_OK;
    }
    s_f = g > max_grad ? max_grad / g : 1.0;
    if(scaling_min_grad > 0.0 && s_f < scaling_min_grad) s_f = scaling_min_grad;
    // per block: 1 / max(1, rowmax / max_grad) if ANY row of the block exceeds max_grad, else 1   (:473-490)
    auto block = [&](double* p, int cnt, double blockmax) -> int {
      if(cnt == 0) return HIOPAMD_OK;
      if(blockmax > max_grad) {
        RC(hiopamd_vec_scale(ctx, cnt, p, 1.0 / max_grad));
        RC(hiopamd_vec_component_max_c(ctx, cnt, p, 1.0));
        RC(hiopamd_vec_invert(ctx, cnt, p));
      } else {
        RC(hiopamd_vec_set_to_constant(ctx, cnt, p, 1.0));
      }
      if(scaling_min_grad > 0.0) RC(hiopamd_vec_component_max_c(ctx, cnt, p, scaling_min_grad));
      return HIOPAMD_OK;
    };
    RC(block(d_scal.p, neq, mc));
    RC(block(d_scal.p + neq, nineq, md));
    (void)v;
    // the constraint right-hand sides and bounds live in the scaled space from here on (:707-709)
    if(neq) RC(hiopamd_vec_component_mult(ctx, neq, d_crhs.p, d_scal.p));
    if(nineq) RC(hiopamd_vec_component_mult(ctx, nineq, d_dl.p, d_scal.p + neq));
    if(nineq) RC(hiopamd_vec_component_mult(ctx, nineq, d_du.p, d_scal.p + neq));
    scaled = true;
    if(o.verbosity_level >= 3)
      std::printf("hiop_amd: gradient-based scaling on (max |grad f| = %.3e, max |Jac_c| = %.3e, max |Jac_d| = %.3e): objective factor %.6e\n", g, mc, md, s_f);
    return HIOPAMD_OK;
  }
  int setup();
  int run();
};

int MdsSolver::setup()
{
  RC(hiopamd_ctx_create(&ctx, nullptr));
  hiop_size_type nn = 0, mm = 0;
  void* ud = dprob ? dprob->user_data : prob->user_data;
  (void)(dprob ? dprob->get_prob_sizes : prob->get_prob_sizes)(&nn, &mm, ud);
  n = nn;
  m = mm;
  if(!dprob) {
    hiop_size_type a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
    (void)prob->get_sparse_dense_blocks_info(&a, &b, &c, &d, &e, &f, prob->user_data);
    ns = a;
    nd = b;
    nnzJ = c + d;
    nnzH = e;
    if(ns + nd != n || f != 0) {   // hiopNlpFormulation.cpp:1873 asserts nnz_sparse_Hess_Lagr_SD == 0 as well
      std::fprintf(stderr, "hiop_amd: MDS problem needs nx_sparse + nx_dense == n and an empty sparse-dense Hessian block\n");
      status = Invalid_Problem_Definition;
      return HIOPAMD_ERR_ARG;
    }
  }
  xl.resize(n);
  xu.resize(n);
  cl.resize(std::max(m, 1));
  cu.resize(std::max(m, 1));
  (void)(dprob ? dprob->get_vars_info : prob->get_vars_info)(n, xl.data(), xu.data(), ud);
  (void)(dprob ? dprob->get_cons_info : prob->get_cons_info)(m, cl.data(), cu.data(), ud);
  // Fixed variables (xlow == xupp).  MDS interface: the reference's fixed_var option is at its default "none" there and the solver
  // terminates (hiopNlpFormulation.cpp:359-366) — so does this one.  Dense interface: hiop_dense_create_problem sets fixed_var = relax
  // (chiopInterface.cpp:138), and with bound_relax_perturb > 0 (default 1e-8) it is the bounds relaxer that opens them
  // (hiopNlpFormulation.cpp:342-347, 398-402) — the relaxation below, applied to every bound, does exactly that.
  for(int i = 0; i < n && !dprob; ++i) {
    if(xl[i] == xu[i]) {
      std::fprintf(stderr, "hiop_amd: fixed variable %d (xlow == xupp) and fixed_var = none (the MDS interface's setting): Invalid_Problem_Definition\n", i);
      status = Invalid_Problem_Definition;
      return HIOPAMD_ERR_ARG;
    }
  }
  // equality / inequality split (hiopNlpFormulation.cpp:560-604)
  for(int i = 0; i < m; ++i) (cl[i] == cu[i] ? eq_map : ineq_map).push_back(i);
  neq = (int)eq_map.size();
  nineq = (int)ineq_map.size();
  std::vector<double> crhs(std::max(neq, 1)), dl(std::max(nineq, 1)), du(std::max(nineq, 1));
  for(int i = 0; i < neq; ++i) crhs[i] = cl[eq_map[i]];
  for(int i = 0; i < nineq; ++i) {
    dl[i] = cl[ineq_map[i]];
    du[i] = cu[ineq_map[i]];
  }
  // patterns (:469-495, :628-645), then hiopBoundsRelaxer::relax (hiopNlpTransforms.cpp:366-389) on every entry
  std::vector<double> ixl(n), ixu(n), idl(std::max(nineq, 1)), idu(std::max(nineq, 1));
  for(int i = 0; i < n; ++i) {
    ixl[i] = xl[i] > -1e20 ? 1.0 : 0.0;
    ixu[i] = xu[i] < 1e20 ? 1.0 : 0.0;
  }
  for(int i = 0; i < nineq; ++i) {
    idl[i] = dl[i] > -1e20 ? 1.0 : 0.0;
    idu[i] = du[i] < 1e20 ? 1.0 : 0.0;
  }
  n_complem = 0;
  for(int i = 0; i < n; ++i) n_complem += ixl[i] + ixu[i];
  for(int i = 0; i < nineq; ++i) n_complem += idl[i] + idu[i];
  if(o.bound_relax_perturb > 0) {
    const double r = o.bound_relax_perturb;
    for(int i = 0; i < n; ++i) {
      xl[i] = xl[i] - r * std::fmax(std::fabs(xl[i]), 1.0);
      xu[i] = xu[i] + r * std::fmax(std::fabs(xu[i]), 1.0);
    }
    for(int i = 0; i < nineq; ++i) {
      dl[i] = dl[i] - r * std::fmax(std::fabs(dl[i]), 1.0);
      du[i] = du[i] + r * std::fmax(std::fabs(du[i]), 1.0);
    }
  }
  // sparsity patterns (host, once)
  iJ.resize(std::max(nnzJ, 1));
  jJ.resize(std::max(nnzJ, 1));
  iH.resize(std::max(nnzH, 1));
  jH.resize(std::max(nnzH, 1));
  h_x.assign((size_t)n, 0.0);
  if((dprob ? dprob->get_starting_point : prob->get_starting_point)(n, h_x.data(), ud) != 0) {
    std::fprintf(stderr, "hiop_amd: user did not provide a starting point; will be set to all zeros\n");   // :326-331
    std::fill(h_x.begin(), h_x.end(), 0.0);
  }
  std::vector<double> x0 = h_x;
  RC(it.alloc(1));   // (placeholder so that x_for_cb below has a context; real slabs follow)
  if(!dprob) {
    // the pattern calls take x as well (the reference passes the starting point); device mode needs it on the device
    DevBuf<double> x0d;
    RC(x0d.alloc((size_t)n));
    RC(h2d(x0d.p, x0.data(), sizeof(double) * (size_t)n));
    RC(hiopamd_ctx_sync(ctx));
    double* xcb = dev_cb ? x0d.p : x0.data();
    (void)prob->eval_Jac_cons(n, m, xcb, 1, ns, nd, nnzJ, iJ.data(), jJ.data(), nullptr, nullptr, prob->user_data);
    DevBuf<double> lam0;
    RC(lam0.alloc((size_t)std::max(m, 1)));
    HIOPAMD_CHECK(hipMemset(lam0.p, 0, sizeof(double) * (size_t)std::max(m, 1)));
    std::vector<double> lamh((size_t)std::max(m, 1), 0.0);
    (void)prob->eval_Hess_Lagr(n, m, xcb, 1, 1.0, dev_cb ? lam0.p : lamh.data(), 1, ns, nd, nnzH, iH.data(), jH.data(), nullptr, nullptr,
                            0, nullptr, nullptr, nullptr, prob->user_data);
  }
  // split the Jacobian triplets by row class, order preserved (copyRowsFrom keeps the (row, col) order of the source)
  std::vector<int> rank_eq(std::max(m, 1), -1), rank_in(std::max(m, 1), -1);
  for(int i = 0; i < neq; ++i) rank_eq[eq_map[i]] = i;
  for(int i = 0; i < nineq; ++i) rank_in[ineq_map[i]] = i;
  std::vector<int> jc_src, jd_src;
  for(int t = 0; t < nnzJ; ++t) {
    const int r = iJ[t];
    if(r < 0 || r >= m || jJ[t] < 0 || jJ[t] >= ns) {
      std::fprintf(stderr, "hiop_amd: sparse Jacobian entry %d out of range\n", t);
      status = Invalid_Problem_Definition;
      return HIOPAMD_ERR_ARG;
    }
    if(rank_eq[r] >= 0) {
      h_Jcs_i.push_back(rank_eq[r]);
      h_Jcs_j.push_back(jJ[t]);
      jc_src.push_back(t);
    } else {
      h_Jds_i.push_back(rank_in[r]);
      h_Jds_j.push_back(jJ[t]);
      jd_src.push_back(t);
    }
  }
  nnzJeq = (int)jc_src.size();
  nnzJineq = (int)jd_src.size();
  auto sorted = [](const std::vector<int>& i, const std::vector<int>& j) {
    for(size_t t = 1; t < i.size(); ++t)
      if(i[t] < i[t - 1] || (i[t] == i[t - 1] && j[t] <= j[t - 1])) return false;
    return true;
  };
  if(!sorted(h_Jcs_i, h_Jcs_j) || !sorted(h_Jds_i, h_Jds_j)) {
    std::fprintf(stderr, "hiop_amd: the sparse Jacobian triplets must be ordered by (row, column) (hiopInterface.hpp:617-626)\n");
    status = Invalid_Problem_Definition;
    return HIOPAMD_ERR_ARG;
  }
  auto up_i = [&](DevBuf<int>& b, const std::vector<int>& v) {
    RC(b.alloc(v.size()));
    if(!v.empty()) RC(h2d(b.p, v.data(), sizeof(int) * v.size()));
    return (int)HIOPAMD_OK;
  };
  auto up_d = [&](DevBuf<double>& b, const std::vector<double>& v, size_t cnt) {
    RC(b.alloc(cnt));
    if(cnt) RC(h2d(b.p, v.data(), sizeof(double) * cnt));
    return (int)HIOPAMD_OK;
  };
  RC(up_i(d_eq_map, eq_map));
  RC(up_i(d_ineq_map, ineq_map));
  RC(up_i(d_jc_src, jc_src));
  RC(up_i(d_jd_src, jd_src));
  RC(up_i(d_Jcs_i, h_Jcs_i));
  RC(up_i(d_Jcs_j, h_Jcs_j));
  RC(up_i(d_Jds_i, h_Jds_i));
  RC(up_i(d_Jds_j, h_Jds_j));
  iH.resize((size_t)nnzH);
  jH.resize((size_t)nnzH);
  RC(up_i(d_Hss_i, iH));
  RC(up_i(d_Hss_j, jH));
  RC(up_d(d_xl, xl, n));
  RC(up_d(d_xu, xu, n));
  RC(up_d(d_dl, dl, nineq));
  RC(up_d(d_du, du, nineq));
  RC(up_d(d_crhs, crhs, neq));
  RC(up_d(d_ixl, ixl, n));
  RC(up_d(d_ixu, ixu, n));
  RC(up_d(d_idl, idl, nineq));
  RC(up_d(d_idu, idu, nineq));
  RC(hiopamd_ctx_sync(ctx));

  if(dprob) {
    // hiopHessianLowRank + hiopKKTLinSysLowRank behind the full-space layer (options at their defaults: secant_memory_len 6,
    // sigma0 1, sigma_update_strategy sty); the perturbation object is hiopPDPerturbationNull (hiopAlgFilterIPM.cpp:1054)
    RC(hiopamd_hess_lowrank_create(&hess, ctx, n, neq, nineq, secant_memory_len, sigma0, 1));
    RC(hiopamd_kkt_lowrank_create(&klr, ctx, hess));
    RC(hiopamd_kkt_xycyd_create_lowrank(&full, ctx, klr, d_ixl.p, d_ixu.p, d_idl.p, d_idu.p));
    RC(d_Jall.alloc((size_t)m * (size_t)n));
    RC(d_J.alloc((size_t)m * (size_t)n));
  }
  hiopamd_mds_structure s;
  std::memset(&s, 0, sizeof(s));
  s.nxs = ns;
  s.nxd = nd;
  s.neq = neq;
  s.nineq = nineq;
  s.nnz_Jcs = nnzJeq;
  s.Jcs_i = d_Jcs_i.p;
  s.Jcs_j = d_Jcs_j.p;
  s.Jcs_i_host = h_Jcs_i.data();
  s.Jcs_j_host = h_Jcs_j.data();
  s.nnz_Jds = nnzJineq;
  s.Jds_i = d_Jds_i.p;
  s.Jds_j = d_Jds_j.p;
  s.Jds_i_host = h_Jds_i.data();
  s.Jds_j_host = h_Jds_j.data();
  s.nnz_Hss = nnzH;
  s.Hss_i = d_Hss_i.p;
  s.Hss_j = d_Hss_j.p;
  if(!dprob) {
    RC(hiopamd_kkt_mds_create(&kkt, ctx, &s));
    RC(hiopamd_kkt_xycyd_create_mds(&full, ctx, kkt, d_ixl.p, d_ixu.p, d_idl.p, d_idu.p));
  }
  RC(hiopamd_kkt_xycyd_set_bounds(full, d_xl.p, d_xu.p, d_dl.p, d_du.p, d_crhs.p));
  RC(hiopamd_kkt_xycyd_offsets(full, off));
  dim = hiopamd_kkt_xycyd_dim(full);

  const size_t szJD = (size_t)m * (size_t)nd, szHD = (size_t)nd * (size_t)nd;
  (void)hipFree(it.p);
  it.p = nullptr;
  for(DevBuf<double>* b : {&it, &trial, &dir, &dir_soc, &resid, &resid_soc}) RC(b->alloc((size_t)dim));
  RC(d_MJ.alloc(nnzJ));
  RC(d_JacD.alloc(szJD));
  RC(d_Jcs_v.alloc(nnzJeq));
  RC(d_Jds_v.alloc(nnzJineq));
  RC(d_Jcd.alloc((size_t)neq * nd));
  RC(d_Jdd.alloc((size_t)nineq * nd));
  RC(d_MH.alloc(nnzH));
  RC(d_HDD.alloc(szHD));
  RC(d_lambda.alloc(m));
  RC(d_grad.alloc(n));
  RC(d_gx.alloc(n));
  for(DevBuf<double>* b : {&d_cons, &d_cons_t}) RC(b->alloc(m));
  for(DevBuf<double>* b : {&d_c, &d_c_t, &d_csoc, &d_tmpc}) RC(b->alloc(neq));
  for(DevBuf<double>* b : {&d_d, &d_d_t, &d_dsoc, &d_tmpd, &d_gd}) RC(b->alloc(nineq));
  HIOPAMD_CHECK(hipMemsetAsync(it.p, 0, sizeof(double) * (size_t)dim, ctx->stream));
  RC(h2d(it.p + off[0], x0.data(), sizeof(double) * (size_t)n));
  RC(hiopamd_ctx_sync(ctx));
  return HIOPAMD_OK;
}

int MdsSolver::run()
{
  using clk = std::chrono::steady_clock;
  const auto t_run0 = clk::now();
  t_kkt = 0.0;
  struct Stop {
    MdsSolver* s;
    clk::time_point t0;
    ~Stop() { s->t_total = std::chrono::duration<double>(clk::now() - t0).count(); }
  } stop{this, t_run0};
  const double eps_tol = o.tolerance;
  double mu = o.mu0;
  double tau = std::fmax(o.tau_min, 1.0 - mu);   // :269
  double f = 0.0, f_trial = 0.0;
  // ---- startingProcedure (:290-425) with duals_init = zero, no warm start
  int okp = 1;
  RC(eval_func(it.p, &f, d_cons.p, d_c.p, d_d.p));   // :345 evalNlp_noHess at the user's starting point
  RC(eval_deriv(it.p, part(it, 2), part(it, 3)));    //      (yc = yd = 0)
  RC(decide_scaling());                              // :351 apply_scaling
  RC(hiopamd_vec_project_into_bounds(ctx, n, part(it, 0), d_xl.p, d_ixl.p, d_xu.p, d_ixu.p, o.kappa1, o.kappa2, &okp));   // :357
  if(!okp) {
    std::fprintf(stderr, "hiop_amd: inconsistent variable bounds (projectIntoBounds failed)\n");
    status = Invalid_Problem_Definition;
    return HIOPAMD_ERR_ARG;
  }
  RC(eval_func(it.p, &f, d_cons.p, d_c.p, d_d.p));                                       // :364 again, after the projection / with the scaling
  RC(eval_deriv(it.p, part(it, 2), part(it, 3)));
  RC(hiopamd_vec_copy(ctx, nineq, part(it, 1), d_d.p));                                  // :374
  RC(hiopamd_vec_project_into_bounds(ctx, nineq, part(it, 1), d_dl.p, d_idl.p, d_du.p, d_idu.p, o.kappa1, o.kappa2, &okp));   // :378
  if(!okp) {
    status = Invalid_Problem_Definition;
    return HIOPAMD_ERR_ARG;
  }
  int nadj = 0;
  RC(hiopamd_iterate_determine_slacks(full, it.p));                                      // :380 compute_safe_slacks
  RC(hiopamd_iterate_adjust_small_slacks(full, it.p, it.p, mu, &nadj));
  if(nadj > 0) {   // :383-386
    if(o.verbosity_level >= 2) std::printf("%d slacks are too small. Adjust corresponding variable slacks!\n", nadj);
    RC(hiopamd_iterate_adjust_bounds(full, it.p, d_xl.p, d_xu.p, d_dl.p, d_du.p));
  }
  RC(hiopamd_vec_copy(ctx, n, part(it, 8), d_ixl.p));                                    // :390 setBoundsDualsToConstant(1.)
  RC(hiopamd_vec_copy(ctx, n, part(it, 9), d_ixu.p));
  RC(hiopamd_vec_copy(ctx, nineq, part(it, 10), d_idl.p));
  RC(hiopamd_vec_copy(ctx, nineq, part(it, 11), d_idu.p));

  double norms[11];
  double f_logbar = 0.0, f_logbar_trial = 0.0;
  RC(logbar(it, f, mu, &f_logbar));                                                      // :2145
  RC(hiopamd_residual_update(full, it.p, d_c.p, d_d.p, d_grad.p, mu, o.kappa_d, resid.p, norms));   // :2148
  const double theta_max = o.theta_max_fact * std::fmax(1.0, norms[6]);                  // :2157-2158
  const double theta_min = o.theta_min_fact * std::fmax(1.0, norms[6]);
  Filter filt;
  int iter_num = 0, n_accep = 0, ls_status = -1, ls_num = 0, use_soc = 0;
  double ap = 0.0, ad = 0.0;
  Errors e0{-1, -1, -1, 0, 0, 0};
  status = NlpSolve_Pending;
  nfact = 0;
  char line[256];
  for(;;) {
    Errors e;
    RC(errors(norms, &e));                                                               // :2219
    if(o.verbosity_level >= 3) {                                                         // outputIteration, :2783-2812
      if(iter_num % 10 == 0) {
        hiopamd_io_iteration_header(line, sizeof(line));
        std::fputs(line, stdout);
      }
      hiopamd_io_format_iteration(line, sizeof(line), dprob ? 1 : 0, iter_num, f / s_f, e.feas, e.optim, mu, ad, ap, ls_status, ls_num, use_soc, 0);
      std::fputs(line, stdout);
    }
    if(e0.optim < 0) e0 = e;
    // ---- checkTermination, :814-845
    if(e.nlp <= eps_tol && e.optim <= o.dual_tol && e.cons_violation <= o.cons_tol && e.complem <= o.comp_tol) {
      status = Solve_Success;
      break;
    }
    if(iter_num >= o.max_iter) {
      status = Max_Iter_Exceeded;
      break;
    }
    if(o.rel_tolerance > 0 && e.optim <= o.rel_tolerance * e0.optim && e.feas <= o.rel_tolerance * e0.feas &&
       e.complem <= std::fmax(o.rel_tolerance, 1e-6) * std::fmin(1., e0.complem)) {
      status = Solve_Success_RelTol;
      break;
    }
    n_accep = e.nlp <= o.acceptable_tolerance ? n_accep + 1 : 0;
    if(n_accep >= o.acceptable_iterations) {
      status = Solve_Acceptable_Level;
      break;
    }
    // ---- barrier update, :2291-2328 with update_log_barrier_params :556-567
    while(e.log <= o.kappa_eps * mu) {
      double new_mu = std::fmax(0.0, std::fmin(o.kappa_mu * mu, std::pow(mu, o.theta_mu)));
      new_mu = std::fmax(new_mu, std::fmin(eps_tol, o.comp_tol / s_f) / (10. + 1.));   // target_comp_tol = comp_tol / obj_scale, hiopAlgFilterIPM.cpp:561
      if(std::fabs(new_mu - mu) < 1e-16) break;
      mu = new_mu;
      tau = std::fmax(o.tau_min, 1.0 - mu);
      RC(logbar(it, f, mu, &f_logbar));
      RC(hiopamd_residual_update(full, it.p, d_c.p, d_d.p, d_grad.p, mu, o.kappa_d, resid.p, norms));
      RC(errors(norms, &e));
      filt.initialize(theta_max);                                                        // :2321
    }
    // ---- search direction, :2333-2462
    const auto t_k0 = clk::now();   // nlp->runStats.kkt.start_optimiz_iteration(), hiopAlgFilterIPM.cpp:2339 / :1209
    RC(hiopamd_kkt_xycyd_set_mu(full, mu));
    int ok = 0;
    if(dprob) {   // Hess->update(*it_curr, *_grad_f, *_Jac_c, *_Jac_d), hiopAlgFilterIPM.cpp:1212
      int stored = 0;
      RC(hiopamd_hess_lowrank_update(hess, part(it, 0), d_grad.p, d_J.p, d_J.p + (size_t)neq * n, part(it, 2), part(it, 3), &stored));
    }
    RC(hiopamd_kkt_xycyd_update(full, it.p, &ok));
    nfact += 1 + hiopamd_kkt_xycyd_num_refactorizations(full);
    if(!ok) {
      std::fprintf(stderr, "hiop_amd: unrecoverable error in step computation (factorization) at iteration %d\n", iter_num);
      status = Err_Step_Computation;
      break;
    }
    int conv = 0;
    double info4[4];
    RC(hiopamd_kkt_xycyd_compute_directions_w_IR(full, resid.p, dir.p, o.ir_outer_tol_factor, o.ir_outer_tol_min, o.ir_outer_maxit,
                                                 &ok, &conv, info4));
    if(!ok) {
      status = Err_Step_Computation;
      break;
    }
    {
      // A direction with a NaN or an Inf in it is a FAILED direction (the reference's compute_search_direction would hand the line search
      // nothing else either, hiopAlgFilterIPM.cpp:3335-3390): said here, with the place, instead of fifty-three halvings of alpha and a
      // "step length too small" that blames the model for what the linear algebra delivered.
      int finite = 1;
      RC(hiopamd_vec_isfinite(ctx, dim, dir.p, &finite));
      if(!finite) {
        static const char* const names[12] = {"x", "d", "yc", "yd", "sxl", "sxu", "sdl", "sdu", "zl", "zu", "vl", "vu"};
        std::fprintf(stderr, "hiop_amd: the search direction of iteration %d is not finite (parts:", iter_num);
        for(int p = 0; p < 12; ++p) {
          int fp = 1;
          if(off[p + 1] > off[p]) RC(hiopamd_vec_isfinite(ctx, off[p + 1] - off[p], dir.p + off[p], &fp));
          if(!fp) std::fprintf(stderr, " %s", names[p]);
        }
        std::fprintf(stderr, "): Err_Step_Computation\n");
        status = Err_Step_Computation;
        break;
      }
    }
    t_kkt += std::chrono::duration<double>(clk::now() - t_k0).count();   // end_optimiz_iteration (:2461); the call above synchronised
    // ---- backtracking line search, :2477-2588
    RC(hiopamd_iterate_fraction_to_the_bdry(full, it.p, dir.p, tau, &ap, &ad));
    const double theta = norms[6];   // resid->get_theta()
    double theta_trial = 0.0;
    ls_status = 0;
    ls_num = 0;
    use_soc = 0;
    bool gpd_computed = false, ini_step = true, small_step = false;
    double gpd = 0.0;
    double* dirp = dir.p;
    for(;;) {
      if(!ini_step && ap < o.min_step_size) {
        small_step = true;
        break;
      }
      RC(trial_primals(dirp, ap, ad, mu, &nadj));
      RC(eval_func(trial.p, &f_trial, d_cons_t.p, d_c_t.p, d_d_t.p));
      RC(logbar(trial, f_trial, mu, &f_logbar_trial));
      RC(theta_of(trial, d_c_t.p, d_d_t.p, &theta_trial));
      ++ls_num;
      RC(accept(filt, theta, theta_trial, ap, f_logbar, f_logbar_trial, theta_min, gpd_computed, gpd, mu, dirp, &ls_status));
      if(ls_status > 0) break;
      if(ini_step && theta <= theta_trial && o.max_soc_iter > 0) {
        // ---- apply_second_order_correction, :2949-3038
        double theta_last = 0.0, th = theta_trial, ap_soc = ap, ad_soc = ap;
        int num_soc = 0, st = 0;
        bool gpd_soc_computed = false;
        double gpd_soc = 0.0;
        {
          const double *crhs = d_crhs.p, *c = d_c.p, *d = d_d.p, *itd = part(it, 1);
          double *cs = d_csoc.p, *ds = d_dsoc.p;
          RC(launch_ew(ctx, neq, [=] __device__(int64_t i) { cs[i] = crhs[i] - c[i]; }));
          RC(launch_ew(ctx, nineq, [=] __device__(int64_t i) { ds[i] = itd[i] - d[i]; }));
        }
        while(num_soc < o.max_soc_iter && (num_soc == 0 || th <= o.kappa_soc * theta_last)) {
          theta_last = th;
          {
            const double *crhs = d_crhs.p, *ct = d_c_t.p, *dt = d_d_t.p, *trd = part(trial, 1);
            double *cs = d_csoc.p, *ds = d_dsoc.p;
            const double a = ap_soc;
            RC(launch_ew(ctx, neq, [=] __device__(int64_t i) { cs[i] = a * cs[i] + crhs[i] - ct[i]; }));
            RC(launch_ew(ctx, nineq, [=] __device__(int64_t i) { ds[i] = a * ds[i] + trd[i] - dt[i]; }));
          }
          // hiopResidual::update_soc (hiopResidual.cpp:425-600): the residual of the iterate with ryc, ryd replaced
          RC(hiopamd_vec_copy(ctx, dim, resid_soc.p, resid.p));
          RC(hiopamd_vec_copy(ctx, neq, resid_soc.p + off[2], d_csoc.p));
          RC(hiopamd_vec_copy(ctx, nineq, resid_soc.p + off[3], d_dsoc.p));
          RC(hiopamd_kkt_xycyd_compute_directions(full, resid_soc.p, dir_soc.p, &ok));
          if(!ok) {
            status = Err_Step_Computation;
            break;
          }
          RC(hiopamd_iterate_fraction_to_the_bdry(full, it.p, dir_soc.p, tau, &ap_soc, &ad_soc));
          RC(trial_primals(dir_soc.p, ap_soc, ad_soc, mu, &nadj));
          RC(eval_func(trial.p, &f_trial, d_cons_t.p, d_c_t.p, d_d_t.p));
          RC(logbar(trial, f_trial, mu, &f_logbar_trial));
          RC(theta_of(trial, d_c_t.p, d_d_t.p, &th));
          RC(accept(filt, theta, th, ap, f_logbar, f_logbar_trial, theta_min, gpd_soc_computed, gpd_soc, mu, dir.p, &st));
          if(st > 0) break;
          ++num_soc;
        }
        if(status == Err_Step_Computation) break;
        if(st > 0) {
          ls_status = st;
          ap = ap_soc;
          dirp = dir_soc.p;
          if(o.soc_theta_corrected) theta_trial = th;   // (default: by value, like the reference — see Options::soc_theta_corrected)
          gpd_computed = gpd_soc_computed;
          gpd = gpd_soc;
          use_soc = 1;
          break;
        }
      }
      ap *= 0.5;
      ini_step = false;
    }
    if(status == Err_Step_Computation) break;
    if(small_step) {
      std::fprintf(stderr, "hiop_amd: minimum step size reached at iteration %d; feasibility restoration is not implemented\n", iter_num);
      std::fprintf(stderr, "hiop_amd:   last trial point: theta %.6e (current %.6e), barrier objective %.6e (current %.6e), %d trial points\n",
                   theta_trial, theta, f_logbar_trial, f_logbar, ls_num);
      status = Steplength_Too_Small;
      break;
    }
    if(nadj > 0) {   // :2589-2593: the bounds follow the slacks adjust_small_slacks moved in the accepted trial point
      if(o.verbosity_level >= 2) std::printf("%d slacks are too small. Adjust corresponding variable slacks!\n", nadj);
      RC(hiopamd_iterate_adjust_bounds(full, trial.p, d_xl.p, d_xu.p, d_dl.p, d_du.p));
    }
    // ---- filter augmentation, :2616-2653
    if(ls_status == 1) {
      if(!gpd_computed) {
        RC(grad_phi_dx(mu, dirp, &gpd));
        gpd_computed = true;
      }
      if(gpd < 0 && ap * std::pow(-gpd, o.s_phi) > o.delta * std::pow(theta, o.s_theta)) {
        if(!(f_logbar_trial <= f_logbar + o.eta_phi * ap * gpd)) filt.add(theta_trial, f_logbar_trial);
      } else {
        filt.add(theta_trial, f_logbar_trial);
      }
    } else if(ls_status == 2) {
      filt.add(theta_trial, f_logbar_trial);
    }
    ++iter_num;
    // ---- duals (hiopDualsNewtonLinearUpdate::go), derivatives at the accepted point, swap, :2714-2754
    RC(hiopamd_iterate_take_step(full, trial.p, it.p, dirp, ap, ad, 0, 1));
    RC(hiopamd_iterate_adjust_duals_plh(full, trial.p, mu, o.kappa_Sigma));
    std::swap(it.p, trial.p);
    std::swap(d_cons.p, d_cons_t.p);
    std::swap(d_c.p, d_c_t.p);
    std::swap(d_d.p, d_d_t.p);
    f = f_trial;
    RC(eval_deriv(it.p, part(it, 2), part(it, 3)));
    RC(logbar(it, f, mu, &f_logbar));
    RC(hiopamd_residual_update(full, it.p, d_c.p, d_d.p, d_grad.p, mu, o.kappa_d, resid.p, norms));
  }
  iters = iter_num;
  double* sol = dprob ? dprob->solution : prob->solution;
  if(dprob) {
    dprob->obj_value = f / s_f;   // apply_inv_to_obj (hiopNlpTransforms.hpp:387)
    dprob->niters = iter_num;
    dprob->status = status;
  } else {
    prob->obj_value = f / s_f;
  }
  if(sol) RC(d2h(sol, part(it, 0), sizeof(double) * (size_t)n));
  return HIOPAMD_OK;
}

MdsSolver* solver_of(const cHiopMDSProblem* p) { return p ? static_cast<MdsSolver*>(p->refcppHiop) : nullptr; }
MdsSolver* solver_of(const cHiopDenseProblem* p) { return p ? static_cast<MdsSolver*>(p->refcppHiop) : nullptr; }

// A second hiop_*_solve_problem on one problem object.  The reference builds a fresh hiopAlgFilterIPM* per call (chiopInterface.cpp:79-87,
// :141-150) whose run() re-initialises from the user's data (starting point, bounds), so a binding may solve twice and gets the same
// answer twice.  The solver state here scales its bounds and right-hand sides in place and moves them with the slacks; instead of
// keeping pristine copies of every such array the state is dropped and rebuilt from the callbacks: what survives is what the caller set
// between create and solve (options, callback memory space, secant memory), exactly what survives in the reference's nlp object.
MdsSolver* resolve_from_scratch(MdsSolver* old)
{
  MdsSolver* s = new(std::nothrow) MdsSolver();
  if(!s) return nullptr;
  s->prob = old->prob;
  s->dprob = old->dprob;
  s->user = old->user;
  delete old;   // (device memory of the first solve goes back before the second allocates)
  return s;
}

}  // namespace

extern "C" {

int hiop_mds_create_problem(cHiopMDSProblem* problem)
{
  if(!problem || !problem->get_prob_sizes || !problem->get_vars_info || !problem->get_cons_info || !problem->eval_f ||
     !problem->eval_grad_f || !problem->eval_cons || !problem->get_sparse_dense_blocks_info || !problem->eval_Jac_cons ||
     !problem->eval_Hess_Lagr || !problem->get_starting_point)
    return HIOPAMD_ERR_ARG;
  MdsSolver* s = new(std::nothrow) MdsSolver();
  if(!s) return HIOPAMD_ERR_HIP;
  s->prob = problem;
  problem->refcppHiop = s;
  problem->hiopinterface = nullptr;
  return 0;
}

int hiop_mds_solve_problem(cHiopMDSProblem* problem)
{
  MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  if(s->solved_once) {   // solve again: from the user's data, like the reference (see resolve_from_scratch)
    s = resolve_from_scratch(s);
    problem->refcppHiop = s;
    if(!s) return HIOPAMD_ERR_HIP;
  }
  s->solved_once = true;
  if(!s->full) {
    const int rc = s->setup();
    if(rc != HIOPAMD_OK) return rc;
  }
  const int rc = s->run();
  if(rc != HIOPAMD_OK) return rc;
  return s->status >= 0 ? 0 : s->status;
}

int hiop_mds_destroy_problem(cHiopMDSProblem* problem)
{
  MdsSolver* s = solver_of(problem);
  delete s;
  if(problem) problem->refcppHiop = problem->hiopinterface = nullptr;
  return 0;
}

// ---- dense-constraints problems (hiopInterface.h:150-176; chiopInterface.cpp:129-159: quasi-Newton, duals linear / zero) --------
int hiop_dense_create_problem(cHiopDenseProblem* problem)
{
  if(!problem || !problem->get_prob_sizes || !problem->get_vars_info || !problem->get_cons_info || !problem->eval_f ||
     !problem->eval_grad_f || !problem->eval_cons || !problem->eval_Jac_cons || !problem->get_starting_point)
    return HIOPAMD_ERR_ARG;
  MdsSolver* s = new(std::nothrow) MdsSolver();
  if(!s) return HIOPAMD_ERR_HIP;
  s->dprob = problem;
  s->o.mu0 = 1.0;   // hiop_dense_create_problem leaves mu0 at the option default (hiopOptions.cpp: 1.)
  problem->refcppHiop = s;
  problem->hiopinterface = nullptr;
  problem->niters = 0;
  problem->status = NlpSolve_SolveNotCalled;
  return 0;
}

int hiop_dense_solve_problem(cHiopDenseProblem* problem)
{
  MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  if(s->solved_once) {   // solve again: from the user's data, like the reference (see resolve_from_scratch)
    s = resolve_from_scratch(s);
    problem->refcppHiop = s;
    if(!s) return HIOPAMD_ERR_HIP;
  }
  s->solved_once = true;
  if(!s->full) {
    const int rc = s->setup();
    if(rc != HIOPAMD_OK) return rc;
  }
  const int rc = s->run();
  if(rc != HIOPAMD_OK) return rc;
  return s->status >= 0 ? 0 : s->status;
}

int hiop_dense_destroy_problem(cHiopDenseProblem* problem)
{
  MdsSolver* s = solver_of(problem);
  delete s;
  if(problem) problem->refcppHiop = problem->hiopinterface = nullptr;
  return 0;
}

int hiopamd_dense_set_callback_mem_space(cHiopDenseProblem* problem, int device)
{
  MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  // before the first solve, or between solves (the next solve rebuilds the state from the callbacks anyway); not while a set-up state
  // that was never run exists
  if(s->full && !s->solved_once) return HIOPAMD_ERR_STATE;
  s->dev_cb = device != 0;
  return HIOPAMD_OK;
}

int hiopamd_dense_set_numeric_option(cHiopDenseProblem* problem, const char* name, double v)
{
  MdsSolver* s = solver_of(problem);
  if(!s || !name) return HIOPAMD_ERR_ARG;
  if(std::string(name) == "secant_memory_len") {
    s->secant_memory_len = (int)v;
    return HIOPAMD_OK;
  }
  if(std::string(name) == "sigma0") {
    s->sigma0 = v;
    return HIOPAMD_OK;
  }
  cHiopMDSProblem shim;
  std::memset(&shim, 0, sizeof(shim));
  shim.refcppHiop = s;
  return hiopamd_mds_set_numeric_option(&shim, name, v);
}

int hiopamd_dense_get_solve_info(const cHiopDenseProblem* problem, int* status, int* num_iterations, int* num_factorizations)
{
  const MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  if(status) *status = s->status;
  if(num_iterations) *num_iterations = s->iters;
  if(num_factorizations) *num_factorizations = s->nfact;
  return HIOPAMD_OK;
}

int hiopamd_mds_set_callback_mem_space(cHiopMDSProblem* problem, int device)
{
  MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  // before the first solve, or between solves (the next solve rebuilds the state from the callbacks anyway); not while a set-up state
  // that was never run exists
  if(s->full && !s->solved_once) return HIOPAMD_ERR_STATE;
  s->dev_cb = device != 0;
  return HIOPAMD_OK;
}

int hiopamd_mds_set_numeric_option(cHiopMDSProblem* problem, const char* name, double v)
{
  MdsSolver* s = solver_of(problem);
  if(!s || !name) return HIOPAMD_ERR_ARG;
  Options& o = s->o;
  const std::string k(name);
#define OPT(x)   \
  if(k == #x) {  \
    o.x = v;     \
    return HIOPAMD_OK; \
  }
  OPT(mu0) OPT(tolerance) OPT(kappa_d) OPT(tau_min) OPT(kappa_mu) OPT(theta_mu) OPT(kappa_eps) OPT(kappa1) OPT(kappa2) OPT(smax)
  OPT(bound_relax_perturb) OPT(acceptable_tolerance) OPT(dual_tol) OPT(cons_tol) OPT(comp_tol) OPT(min_step_size) OPT(kappa_soc)
  OPT(rel_tolerance)
#undef OPT
#define OPTI(x)       \
  if(k == #x) {       \
    o.x = (int)v;     \
    return HIOPAMD_OK; \
  }
  OPTI(max_iter) OPTI(acceptable_iterations) OPTI(max_soc_iter) OPTI(verbosity_level) OPTI(soc_theta_corrected)
#undef OPTI
  return HIOPAMD_ERR_ARG;
}

int hiopamd_mds_get_solve_times(const cHiopMDSProblem* problem, double* total_seconds, double* kkt_seconds)
{
  const MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  if(total_seconds) *total_seconds = s->t_total;
  if(kkt_seconds) *kkt_seconds = s->t_kkt;
  return HIOPAMD_OK;
}

int hiopamd_dense_get_solve_times(const cHiopDenseProblem* problem, double* total_seconds, double* kkt_seconds)
{
  const MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  if(total_seconds) *total_seconds = s->t_total;
  if(kkt_seconds) *kkt_seconds = s->t_kkt;
  return HIOPAMD_OK;
}

int hiopamd_mds_get_solve_info(const cHiopMDSProblem* problem, int* status, int* num_iterations, int* num_factorizations)
{
  const MdsSolver* s = solver_of(problem);
  if(!s) return HIOPAMD_ERR_ARG;
  if(status) *status = s->status;
  if(num_iterations) *num_iterations = s->iters;
  if(num_factorizations) *num_factorizations = s->nfact;
  return HIOPAMD_OK;
}

}  // extern "C"
