/* hiop_amd — the C interface of the MDS solver.
 *
 * Drop-in for the reference's C FFI, src/Interface/hiopInterface.h:63-98 (implementation src/Interface/chiopInterface.cpp:64-95,
 * user of it: src/Drivers/MDS/NlpMdsEx1.c): the SAME struct of callbacks (member for member, so a program or a Julia / Fortran
 * binding built against the reference's header links against libhiopamd.so unchanged) and the SAME three entry points.  What
 * runs behind them is this library's device path: hiopAlgFilterIPMNewton's loop (src/Optimization/hiopAlgFilterIPM.cpp:2101-2770)
 * with every per-iteration operation — residuals, condensed MDS KKT assembly, LDL^T, inertia correction, BiCGStab refinement,
 * line search, steps — on the MI355X, with the options hiop_mds_create_problem sets in the reference (duals_update_type linear,
 * duals_init zero, mu0 = 0.1; everything else at the reference's defaults).
 *
 * Callback memory space.  By default the callbacks get HOST pointers, exactly like the reference's (the library stages x down and
 * the values up every iteration: the PCIe-inclusive mode).  After hiopamd_mds_set_callback_mem_space(problem, 1) every ARRAY
 * argument of eval_grad_f / eval_cons / eval_Jac_cons / eval_Hess_Lagr and the `x` of eval_f is a DEVICE pointer (what
 * src/Drivers/MDS/NlpMdsRajaEx1.cpp does with RAJA; hiopamd_mdsex1_* in hiop_amd.h are such callbacks for the example problem);
 * scalars (`obj`) and the one-time set-up calls (sizes, bounds, starting point, sparsity pattern) stay on the host.
 * The Jacobian / Hessian callbacks are called once for the pattern (index arrays non-NULL, value arrays NULL) and once per
 * iteration for the values (index arrays NULL), the convention of hiopInterface.hpp:635-643.
 * Fixed variables (xlow == xupp): Invalid_Problem_Definition, as in the reference at this interface's fixed_var = none.
 * NLP scaling: the reference's default scaling_type = gradient (hiopNlpFormulation.cpp:671-714, hiopNlpTransforms.cpp:423-499) — when
 * |grad f(x0)| or a Jacobian entry reaches scaling_max_grad = 100 the objective and the offending constraint rows are scaled (the user's
 * eval_Hess_Lagr is then called with obj_factor = the objective's factor and the multipliers of the unscaled constraints); obj_value is
 * handed back unscaled.
 * Not supported (the call fails with a message instead of computing something else): a nonzero
 * sparse-dense Hessian block, feasibility restoration.
 * The callbacks' return values are ignored, as by the reference's wrappers (chiopInterface.hpp:125-214 call the C function and
 * return true whatever it returned).
 */
#ifndef HIOP_AMD_INTERFACE_H
#define HIOP_AMD_INTERFACE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef int hiop_index_type; /* src/Interface/hiop_types.h:12-13 */
typedef int hiop_size_type;

typedef struct cHiopMDSProblem {
  void* refcppHiop;    /* owned by the library: the solver state (NULL before create / after destroy) */
  void* hiopinterface; /* owned by the library */
  void* user_data;     /* passed back to every callback */
  double* solution;    /* host array of n doubles provided by the caller; filled by hiop_mds_solve_problem */
  double obj_value;    /* filled by hiop_mds_solve_problem */
  int (*get_starting_point)(hiop_size_type n, double* x0, void* user_data);
  int (*get_prob_sizes)(hiop_size_type* n, hiop_size_type* m, void* user_data);
  int (*get_vars_info)(hiop_size_type n, double* xlow, double* xupp, void* user_data);
  int (*get_cons_info)(hiop_size_type m, double* clow, double* cupp, void* user_data);
  int (*eval_f)(hiop_size_type n, double* x, int new_x, double* obj, void* user_data);
  int (*eval_grad_f)(hiop_size_type n, double* x, int new_x, double* gradf, void* user_data);
  int (*eval_cons)(hiop_size_type n, hiop_size_type m, double* x, int new_x, double* cons, void* user_data);
  int (*get_sparse_dense_blocks_info)(hiop_size_type* nx_sparse, hiop_size_type* nx_dense, hiop_size_type* nnz_sparse_Jaceq,
                                      hiop_size_type* nnz_sparse_Jacineq, hiop_size_type* nnz_sparse_Hess_Lagr_SS,
                                      hiop_size_type* nnz_sparse_Hess_Lagr_SD, void* user_data);
  int (*eval_Jac_cons)(hiop_size_type n, hiop_size_type m, double* x, int new_x, hiop_size_type nsparse, hiop_size_type ndense,
                       hiop_size_type nnzJacS, hiop_index_type* iJacS, hiop_index_type* jJacS, double* MJacS, double* JacD,
                       void* user_data);
  int (*eval_Hess_Lagr)(hiop_size_type n, hiop_size_type m, double* x, int new_x, double obj_factor, double* lambda,
                        int new_lambda, hiop_size_type nsparse, hiop_size_type ndense, hiop_size_type nnzHSS,
                        hiop_index_type* iHSS, hiop_index_type* jHSS, double* MHSS, double* HDD, hiop_size_type nnzHSD,
                        hiop_index_type* iHSD, hiop_index_type* jHSD, double* MHSD, void* user_data);
} cHiopMDSProblem;

/* chiopInterface.cpp:64-95.  Return 0 on success; a negative hiopamd status otherwise (the reference asserts).
 * A problem object may be solved again (chiopInterface.cpp:79-87 builds a fresh solver per call): every hiop_mds_solve_problem starts from
 * the user's data (sizes, bounds, starting point are queried again), with the options set so far; options may be changed between solves. */
int hiop_mds_create_problem(cHiopMDSProblem* problem);
int hiop_mds_solve_problem(cHiopMDSProblem* problem);
int hiop_mds_destroy_problem(cHiopMDSProblem* problem);

/* ---- additions (between create and solve) ---------------------------------------------------------------------------
 * 0 = host pointers (default, the reference's contract), 1 = device pointers (see above) */
int hiopamd_mds_set_callback_mem_space(cHiopMDSProblem* problem, int device);
/* numeric options by the reference's names (src/Utils/hiopOptions.cpp): mu0, tolerance, max_iter, kappa_d, tau_min, kappa_mu,
 * theta_mu, kappa_eps, kappa1, kappa2, smax, bound_relax_perturb, acceptable_tolerance, acceptable_iterations, dual_tol,
 * cons_tol, comp_tol, min_step_size, max_soc_iter, kappa_soc, verbosity_level (>= 3 prints the reference's iteration table);
 * soc_theta_corrected (not a reference option; 0 = the reference's by-value theta_trial after a second-order correction, the default).
 * Unknown name: HIOPAMD_ERR_ARG. */
int hiopamd_mds_set_numeric_option(cHiopMDSProblem* problem, const char* name, double value);
/* after solve: status = the reference's hiopSolveStatus value (hiopInterface.hpp:78-110: 0 Solve_Success, 2 Solve_Acceptable_Level,
 * 10 Max_Iter_Exceeded, -5 Err_Step_Computation, ...), iterations, KKT factorisations (inertia corrections included) */
int hiopamd_mds_get_solve_info(const cHiopMDSProblem* problem, int* status, int* num_iterations, int* num_factorizations);
/* wall-clock seconds of the last solve: the whole optimisation loop, and the part inside the KKT span the reference's metric is defined on
 * (runStats.kkt.tmTotal: start_optimiz_iteration ... end_optimiz_iteration, hiopAlgFilterIPM.cpp:2339,2461 — update + factorisation(s) +
 * directions with refinement); iterations / kkt_seconds is BASELINE's 'KKT iterations per second' measured in a real run */
int hiopamd_mds_get_solve_times(const cHiopMDSProblem* problem, double* total_seconds, double* kkt_seconds);

/* ---- dense-constraints problems: src/Interface/hiopInterface.h:150-176, chiopInterface.cpp:129-159 ---------------------------
 * The quasi-Newton solver (hiopAlgFilterIPMQuasiNewton: secant Hessian hiopHessianLowRank, KKT class hiopKKTLinSysLowRank — the
 * memory-distributed dense path of this library on one rank) with the options hiop_dense_create_problem sets in the reference:
 * duals_update_type linear, duals_init zero; mu0 and everything else at the reference's defaults.  `MJac` is the m x n row-major
 * Jacobian.  hiopamd_dense_set_callback_mem_space(problem, 1): x, gradf, cons and MJac are DEVICE pointers.
 * Fixed variables (xlow == xupp) are relaxed like every other bound (the reference sets fixed_var = relax here and, with
 * bound_relax_perturb > 0, leaves them to the bounds relaxer: hiopNlpFormulation.cpp:342-347).
 * Gradient-based NLP scaling as in the MDS interface.  Not supported (fails loudly): feasibility restoration. */
typedef struct cHiopDenseProblem {
  void* refcppHiop;    /* owned by the library */
  void* hiopinterface; /* owned by the library */
  void* user_data;
  double* solution;    /* host array of n doubles provided by the caller */
  double obj_value;
  int niters;
  int status;          /* the reference's hiopSolveStatus value */
  int (*get_starting_point)(hiop_size_type n, double* x0, void* user_data);
  int (*get_prob_sizes)(hiop_size_type* n, hiop_size_type* m, void* user_data);
  int (*get_vars_info)(hiop_size_type n, double* xlow, double* xupp, void* user_data);
  int (*get_cons_info)(hiop_size_type m, double* clow, double* cupp, void* user_data);
  int (*eval_f)(hiop_size_type n, double* x, int new_x, double* obj, void* user_data);
  int (*eval_grad_f)(hiop_size_type n, double* x, int new_x, double* gradf, void* user_data);
  int (*eval_cons)(hiop_size_type n, hiop_size_type m, double* x, int new_x, double* cons, void* user_data);
  int (*eval_Jac_cons)(hiop_size_type n, hiop_size_type m, double* x, int new_x, double* MJac, void* user_data);
} cHiopDenseProblem;

int hiop_dense_create_problem(cHiopDenseProblem* problem);
int hiop_dense_solve_problem(cHiopDenseProblem* problem);
int hiop_dense_destroy_problem(cHiopDenseProblem* problem);
int hiopamd_dense_set_callback_mem_space(cHiopDenseProblem* problem, int device);
/* the names of hiopamd_mds_set_numeric_option, plus secant_memory_len and sigma0 */
int hiopamd_dense_set_numeric_option(cHiopDenseProblem* problem, const char* name, double value);
int hiopamd_dense_get_solve_info(const cHiopDenseProblem* problem, int* status, int* num_iterations, int* num_factorizations);
int hiopamd_dense_get_solve_times(const cHiopDenseProblem* problem, double* total_seconds, double* kkt_seconds);

#ifdef __cplusplus
}
#endif
#endif /* HIOP_AMD_INTERFACE_H */
