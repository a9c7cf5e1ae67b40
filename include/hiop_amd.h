/* hiop_amd.h — C ABI of the MI355X-native KKT hot path for HiOp.
 *
 * Every entry point takes plain device pointers (HBM, fp64 / int32) and sizes; nothing in this
 * header depends on PyTorch, RAJA, Umpire or MAGMA.  Each group cites the reference interface
 * (file:line under LLNL/hiop @ v1.1.0) that a HiOp-side subclass forwards to it; the subclass
 * stubs are shown in INTEGRATION.md.
 *
 * Conventions
 *  - all pointers are DEVICE pointers unless the name ends in `_host`;
 *  - all kernels are enqueued on the context's HIP stream; functions that return a scalar
 *    (`double* out_host` / `int* out_host`) synchronise that stream before returning;
 *  - dense matrices are ROW-MAJOR with explicit leading dimension `ld` (reference:
 *    src/LinAlg/hiopMatrixDenseRowMajor.cpp:90-93);
 *  - symmetric KKT matrices hold only their UPPER triangle (reference: src/LinAlg/readme.md:24-26);
 *  - "pattern"/"select" vectors are fp64 arrays of exact 0.0 / 1.0
 *    (reference: src/LinAlg/hiopVectorPar.cpp:144,782,1053);
 *  - return value: HIOPAMD_OK (0) or a negative error code; functions never fall back to a CPU
 *    path.
 */
#ifndef HIOP_AMD_H
#define HIOP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  HIOPAMD_OK = 0,
  HIOPAMD_ERR_HIP = -1,       /* a HIP runtime call failed */
  HIOPAMD_ERR_ARG = -2,       /* invalid argument */
  HIOPAMD_ERR_NODEVICE = -3,  /* no gfx950 device visible */
  HIOPAMD_ERR_SINGULAR = -4,  /* zero / non-finite pivot met */
  HIOPAMD_ERR_STATE = -5,     /* call sequence error (e.g. solve before factorize) */
  HIOPAMD_ERR_TIMEOUT = -6,   /* a bounded wait of a dataflow kernel expired: the factorisation did not complete and the matrix is
                               * overwritten — re-assemble and call again (that call runs the stepwise kernels).  Only returned by a
                               * solver object whose retry copy was switched off (hiopamd_linsolver_set_retry_copy) */
  HIOPAMD_ERR_SOLVE = -7      /* a solve since the last check delivered invalid results (dataflow time-out) */
} hiopamd_status;

typedef struct hiopamd_ctx hiopamd_ctx;

/* reduction ops for the all-reduce hook (reference: MPI_SUM/MPI_MIN/MPI_MAX call sites listed in
 * SURVEY.md §2.2, e.g. src/LinAlg/hiopVectorPar.cpp:474-548) */
typedef enum { HIOPAMD_SUM = 0, HIOPAMD_MIN = 1, HIOPAMD_MAX = 2 } hiopamd_redop;

/* All-reduce hook: reduce `count` fp64 values in DEVICE buffer `buf` in place across the ranks
 * of the column partition, ordered on `stream` (a hipStream_t).  Return 0 on success. */
typedef int (*hiopamd_allreduce_fn)(void* user, double* buf, size_t count, int op, void* stream);

/* ---- context (plays ExecSpace<MemBackendHip,ExecPolicyHip>; src/ExecBackends/ExecSpace.hpp:345) */
int hiopamd_ctx_create(hiopamd_ctx** out, void* hip_stream /* hipStream_t or NULL */);
int hiopamd_ctx_destroy(hiopamd_ctx* ctx);
int hiopamd_ctx_sync(hiopamd_ctx* ctx);
void* hiopamd_ctx_stream(hiopamd_ctx* ctx);
/* Batched reductions.  Every hiopVector reduction of the reference returns its scalar to the caller (hiopVectorPar.cpp:463-555,
 * 806-907, 1017-1060) — one device round trip each here (~19 us).  Between hiopamd_ctx_reduce_begin and hiopamd_ctx_reduce_end the
 * scalar-returning entry points hiopamd_vec_dot / twonorm / infnorm / onenorm / sum / min / min_w_pattern / log_barrier /
 * linear_damping_term / fraction_to_the_bdry(_w_pattern, _multi) only LAUNCH; their `double* out` arguments — which must stay
 * valid until then — are written by hiopamd_ctx_reduce_end after ONE synchronisation of the context's stream (up to 64 results per
 * round trip; more flush in between).  The norms / step lengths / barrier terms an IPM iteration needs (hiopIterate.cpp:330-365,
 * hiopResidual.cpp:154-360) cost one round trip instead of a dozen.  Brackets nest; results are bitwise those of the unbatched calls. */
int hiopamd_ctx_reduce_begin(hiopamd_ctx* ctx);
int hiopamd_ctx_reduce_end(hiopamd_ctx* ctx);
/* Run-stats spans: the reference's per-iteration KKT timers (src/Utils/hiopRunStats.hpp:82-140: tmUpdateInit,
 * tmUpdateLinsys, tmUpdateInnerFact, tmSolveRhsManip, tmSolveInner) and linear-solver timers (:244-300: tmFactTime,
 * tmInertiaComp, tmTriuSolves), placed where the reference starts/stops them (hiopKKTLinSysMDS.cpp:121-401,
 * hiopKKTLinSys.cpp:221-689, hiopLinSolverSymDenseLapack.hpp:80-195).  Each span is always a roctx range; with
 * hiopamd_ctx_spans_enable(ctx, 1) it is also timed with HIP events on the context's stream (no host sync inside the
 * span).  hiopamd_ctx_spans_read synchronises the stream and returns the accumulated milliseconds and call counts
 * (arrays of HIOPAMD_SPAN_COUNT entries) since the last enable. */
typedef enum {
  HIOPAMD_SPAN_KKT_UPDATE_INIT = 0,
  HIOPAMD_SPAN_KKT_UPDATE_LINSYS = 1,
  HIOPAMD_SPAN_KKT_UPDATE_INNER_FACT = 2,
  HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP = 3,
  HIOPAMD_SPAN_KKT_SOLVE_INNER = 4,
  HIOPAMD_SPAN_LINSOLV_FACT = 5,
  HIOPAMD_SPAN_LINSOLV_INERTIA = 6,
  HIOPAMD_SPAN_LINSOLV_TRIU_SOLVES = 7,
  HIOPAMD_SPAN_COUNT = 8
} hiopamd_span;
int hiopamd_ctx_spans_enable(hiopamd_ctx* ctx, int enable);
int hiopamd_ctx_spans_read(hiopamd_ctx* ctx, double* ms_host, int64_t* count_host);
const char* hiopamd_span_name(int span_id);

int hiopamd_ctx_set_allreduce(hiopamd_ctx* ctx, hiopamd_allreduce_fn fn, void* user, int rank, int size);
/* rank and size the context was given (0 and 1 without a hook) */
int hiopamd_ctx_comm(const hiopamd_ctx* ctx, int* rank_host, int* size_host);
/* RCCL-backed all-reduce: `unique_id_128` is the 128-byte ncclUniqueId produced by
 * hiopamd_rccl_unique_id on rank 0 and broadcast by the host side. */
int hiopamd_rccl_unique_id(unsigned char* unique_id_128_host);
int hiopamd_ctx_init_rccl(hiopamd_ctx* ctx, const unsigned char* unique_id_128_host, int rank, int size);
/* ranks of the RCCL communicator behind the context's hook (ncclCommCount); 0 when the hook is not the RCCL one */
int hiopamd_ctx_rccl_ranks(const hiopamd_ctx* ctx, int* ranks_host);
/* collective statistics: _begin zeroes the call counter of the all-reduce hook (timed != 0: every call is bracketed by HIP events on the
 * context's stream from now on); _read synchronises the stream and returns the number of hook calls since _begin and, in timed mode, the
 * summed device time between the brackets in ms (-1 otherwise), then leaves timed mode.  The replacement of the reference's MPI_Allreduce
 * count (hiopHessianLowRank.cpp:459,590-591; hiopMatrixDenseRowMajor.cpp:466-487) made visible to a benchmark. */
int hiopamd_ctx_collective_stats_begin(hiopamd_ctx* ctx, int timed);
int hiopamd_ctx_collective_stats_read(hiopamd_ctx* ctx, int64_t* count_host, double* ms_host);
const char* hiopamd_version(void);
int hiopamd_device_info(char* name_host, size_t name_len, int* cu_count_host, size_t* hbm_bytes_host);

/* ---- memory (AllocImpl / DeAllocImpl / TransferImpl; src/ExecBackends/MemBackendHipImpl.hpp:73-135) */
int hiopamd_alloc(void** dptr, size_t bytes);
int hiopamd_free(void* dptr);
int hiopamd_copy_h2d(hiopamd_ctx* ctx, void* dst, const void* src_host, size_t bytes);
int hiopamd_copy_d2h(hiopamd_ctx* ctx, void* dst_host, const void* src, size_t bytes);
int hiopamd_copy_d2d(hiopamd_ctx* ctx, void* dst, const void* src, size_t bytes);

/* =====================================================================================
 * hiopVector element-wise family  (reference: src/LinAlg/hiopVector.hpp:74-1003, CPU semantics
 * src/LinAlg/hiopVectorPar.cpp:120-1320).  `y` is always `this`.
 * ===================================================================================== */
int hiopamd_vec_set_to_constant(hiopamd_ctx*, int64_t n, double* y, double c);                          /* :127 */
int hiopamd_vec_set_to_constant_w_pattern(hiopamd_ctx*, int64_t n, double* y, double c, const double* select); /* :139 */
int hiopamd_vec_copy(hiopamd_ctx*, int64_t n, double* y, const double* x);                              /* copyFrom :153 */
int hiopamd_vec_copy_from_w_pattern(hiopamd_ctx*, int64_t n, double* y, const double* x, const double* select); /* :179 */
int hiopamd_vec_copy_from_indexes(hiopamd_ctx*, int64_t n, double* y, const double* src, const int* idx); /* :194 */
int hiopamd_vec_copy_to_starting_at_w_pattern(hiopamd_ctx*, int64_t n, const double* x, double* dest,
                                              int64_t start_in_dest, const double* select, int64_t* nnz_out_host); /* :322 */
int hiopamd_vec_starting_at_copy_to_starting_at_w_pattern(hiopamd_ctx*, const double* src, int64_t start_src,
                                                          double* dest, int64_t n_dest, int64_t start_dest,
                                                          const double* select_dest, int64_t num_elems); /* :431 */
int hiopamd_vec_copy_from_two_vec_w_pattern(hiopamd_ctx*, double* y, const double* c, const int* c_map, int64_t nc,
                                            const double* d, const int* d_map, int64_t nd);             /* :345 */
int hiopamd_vec_copy_to_two_vec_w_pattern(hiopamd_ctx*, const double* y, double* c, const int* c_map, int64_t nc,
                                          double* d, const int* d_map, int64_t nd);                     /* :372 */
int hiopamd_vec_component_mult(hiopamd_ctx*, int64_t n, double* y, const double* x);                    /* :564 */
int hiopamd_vec_component_div(hiopamd_ctx*, int64_t n, double* y, const double* x);                     /* :573 */
int hiopamd_vec_component_div_w_pattern(hiopamd_ctx*, int64_t n, double* y, const double* x, const double* select); /* :580 */
int hiopamd_vec_component_min_c(hiopamd_ctx*, int64_t n, double* y, double c);                          /* :594 */
int hiopamd_vec_component_min_v(hiopamd_ctx*, int64_t n, double* y, const double* x);                   /* :603 */
int hiopamd_vec_component_max_c(hiopamd_ctx*, int64_t n, double* y, double c);                          /* :613 */
int hiopamd_vec_component_max_v(hiopamd_ctx*, int64_t n, double* y, const double* x);                   /* :622 */
int hiopamd_vec_component_abs(hiopamd_ctx*, int64_t n, double* y);                                      /* :632 */
int hiopamd_vec_component_sgn(hiopamd_ctx*, int64_t n, double* y);                                      /* :639 */
int hiopamd_vec_component_sqrt(hiopamd_ctx*, int64_t n, double* y);                                     /* :649 */
int hiopamd_vec_scale(hiopamd_ctx*, int64_t n, double* y, double c);                                    /* :657 */
int hiopamd_vec_axpy(hiopamd_ctx*, int64_t n, double* y, double alpha, const double* x);                /* :664 */
int hiopamd_vec_axpy_w_pattern(hiopamd_ctx*, int64_t n, double* y, double alpha, const double* x, const double* select); /* :692 */
int hiopamd_vec_axpy_w_map(hiopamd_ctx*, int64_t nidx, double* y, double alpha, const double* x, const int* idx);      /* :676 */
int hiopamd_vec_axzpy(hiopamd_ctx*, int64_t n, double* y, double alpha, const double* x, const double* z);  /* :710 */
int hiopamd_vec_axdzpy(hiopamd_ctx*, int64_t n, double* y, double alpha, const double* x, const double* z); /* :736 */
int hiopamd_vec_axdzpy_w_pattern(hiopamd_ctx*, int64_t n, double* y, double alpha, const double* x, const double* z,
                                 const double* select);                                                 /* :767 */
int hiopamd_vec_add_constant(hiopamd_ctx*, int64_t n, double* y, double c);                             /* :793 */
int hiopamd_vec_add_constant_w_pattern(hiopamd_ctx*, int64_t n, double* y, double c, const double* select); /* :798 */
int hiopamd_vec_negate(hiopamd_ctx*, int64_t n, double* y);                                             /* :846 */
int hiopamd_vec_invert(hiopamd_ctx*, int64_t n, double* y);                                             /* :852 */
int hiopamd_vec_add_log_barrier_grad(hiopamd_ctx*, int64_t n, double* y, double alpha, const double* x,
                                     const double* select);                                             /* :893 */
int hiopamd_vec_add_linear_damping_term(hiopamd_ctx*, int64_t n, double* y, const double* ixleft,
                                        const double* ixright, double alpha, double ct);                /* :927 */
int hiopamd_vec_select_pattern(hiopamd_ctx*, int64_t n, double* y, const double* select);               /* :1063 */
int hiopamd_vec_adjust_duals_plh(hiopamd_ctx*, int64_t n, double* z, const double* x, const double* select,
                                 double mu, double kappa);                                              /* :1117 */
int hiopamd_vec_project_into_bounds(hiopamd_ctx*, int64_t n, double* x, const double* xl, const double* ixl,
                                    const double* xu, const double* ixu, double kappa1, double kappa2,
                                    int* ok_out_host);                                                  /* :964 */
int hiopamd_vec_set_to_linspace(hiopamd_ctx*, int64_t n, double* y, double x0, double dx);

/* ---- hiopVector reductions (local part; the *_global variants add the all-reduce hook) */
int hiopamd_vec_dot(hiopamd_ctx*, int64_t n, const double* x, const double* y, double* out_host);        /* :480 */
int hiopamd_vec_twonorm(hiopamd_ctx*, int64_t n, const double* x, double* out_host);                     /* :463 */
int hiopamd_vec_infnorm(hiopamd_ctx*, int64_t n, const double* x, double* out_host);                     /* :501 */
int hiopamd_vec_onenorm(hiopamd_ctx*, int64_t n, const double* x, double* out_host);                     /* :540 */
int hiopamd_vec_sum(hiopamd_ctx*, int64_t n, const double* x, double* out_host);                         /* :883 */
int hiopamd_vec_min(hiopamd_ctx*, int64_t n, const double* x, double* out_host);                         /* :806 */
int hiopamd_vec_min_w_pattern(hiopamd_ctx*, int64_t n, const double* x, const double* select, double* out_host); /* :821 */
int hiopamd_vec_log_barrier(hiopamd_ctx*, int64_t n, const double* x, const double* select, double* out_host);   /* :863 */
int hiopamd_vec_linear_damping_term(hiopamd_ctx*, int64_t n, const double* x, const double* ixleft,
                                    const double* ixright, double mu, double kappa_d, double* out_host); /* :907 */
int hiopamd_vec_fraction_to_the_bdry(hiopamd_ctx*, int64_t n, const double* x, const double* d, double tau,
                                     double* out_host);                                                  /* :1017 */
int hiopamd_vec_fraction_to_the_bdry_w_pattern(hiopamd_ctx*, int64_t n, const double* x, const double* d,
                                               double tau, const double* select, double* out_host);      /* :1038 */
int hiopamd_vec_all_positive(hiopamd_ctx*, int64_t n, const double* x, int* out_host);                   /* :946 */
int hiopamd_vec_all_positive_w_pattern(hiopamd_ctx*, int64_t n, const double* x, const double* select, int* out_host); /* :1095 */
int hiopamd_vec_matches_pattern(hiopamd_ctx*, int64_t n, const double* x, const double* select, int* out_host); /* :1073 */
int hiopamd_vec_is_zero(hiopamd_ctx*, int64_t n, const double* x, int* out_host);                        /* :1150 */
int hiopamd_vec_isnan(hiopamd_ctx*, int64_t n, const double* x, int* out_host);                          /* :1167 */
int hiopamd_vec_isinf(hiopamd_ctx*, int64_t n, const double* x, int* out_host);                          /* :1173 */
int hiopamd_vec_isfinite(hiopamd_ctx*, int64_t n, const double* x, int* out_host);                       /* :1179 */
int hiopamd_vec_num_elems_less_than(hiopamd_ctx*, int64_t n, const double* x, double val, int64_t* out_host);     /* :1222 */
int hiopamd_vec_num_elems_abs_less_than(hiopamd_ctx*, int64_t n, const double* x, double val, int64_t* out_host); /* :1241 */
int hiopamd_vec_is_equal(hiopamd_ctx*, int64_t n, const double* x, const double* y, int* out_host);      /* :1283 */
/* startingAtCopyFromStartingAt (:241-251) / startingAtCopyToStartingAt (:409-420) with the reference's clamping of the count
 * (num_elems < 0: everything from start_idx_in_src on) */
int hiopamd_vec_starting_at_copy_from_starting_at(hiopamd_ctx*, double* dest, int64_t n_dest, int64_t start_idx_dest,
                                                  const double* src, int64_t n_src, int64_t start_idx_src);
int hiopamd_vec_starting_at_copy_to_starting_at(hiopamd_ctx*, const double* src, int64_t n_src, int64_t start_idx_in_src,
                                                double* dest, int64_t n_dest, int64_t start_idx_dest, int64_t num_elems);
/* hiopVectorInt of the same mem-space (src/LinAlg/hiopVectorInt.hpp:64-118): int32 device arrays (hiopamd_alloc) */
int hiopamd_ivec_set_to_constant(hiopamd_ctx*, int64_t n, int* x, int c);          /* set_to_zero :103, set_to_constant :106 */
int hiopamd_ivec_linspace(hiopamd_ctx*, int64_t n, int* x, int i0, int di);        /* :117 */
int hiopamd_ivec_copy(hiopamd_ctx*, int64_t n, int* dst, const int* src);          /* copy_from :83 (device source) */
/* fused step-length kernel used by hiopIterate::fractionToTheBdry (src/Optimization/hiopIterate.cpp:330-365):
 * `k` (x,d,select) triples reduced in ONE launch; out_host[0] = min over all. */
int hiopamd_vec_fraction_to_the_bdry_multi(hiopamd_ctx*, int k, const int64_t* n_host, const double* const* x_host,
                                           const double* const* d_host, const double* const* select_host,
                                           double tau, double* out_host);

/* =====================================================================================
 * hiopMatrixDense (row-major)  (reference: src/LinAlg/hiopMatrixDenseRowMajor.cpp)
 * A is m x n with leading dimension lda (doubles).
 * ===================================================================================== */
int hiopamd_mat_set_to_constant(hiopamd_ctx*, int m, int64_t n, double* A, int64_t lda, double c);       /* :374 */
/* y = beta*y + alpha*A*x  (local part; DGEMV-T at :471) */
int hiopamd_mat_times_vec(hiopamd_ctx*, int m, int64_t n, const double* A, int64_t lda, double beta, double* y,
                          double alpha, const double* x);                                               /* :458 */
/* y = beta*y + alpha*A^T*x */
int hiopamd_mat_trans_times_vec(hiopamd_ctx*, int m, int64_t n, const double* A, int64_t lda, double beta,
                                double* y, double alpha, const double* x);                               /* :510 */
/* W(m x k) = beta*W + alpha*A(m x n)*X(n x k) */
int hiopamd_mat_times_mat(hiopamd_ctx*, int m, int n, int k, const double* A, int64_t lda, double beta, double* W,
                          int64_t ldw, double alpha, const double* X, int64_t ldx);                      /* :578 */
/* W(n x k) = beta*W + alpha*A^T(n x m)*X(m x k) */
int hiopamd_mat_trans_times_mat(hiopamd_ctx*, int m, int n, int k, const double* A, int64_t lda, double beta,
                                double* W, int64_t ldw, double alpha, const double* X, int64_t ldx);     /* :616 */
/* W(m x k) = beta*W + alpha*A(m x n)*X(k x n)^T  (local part; caller all-reduces) */
int hiopamd_mat_times_mat_trans(hiopamd_ctx*, int m, int64_t n, int k, const double* A, int64_t lda, double beta,
                                double* W, int64_t ldw, double alpha, const double* X, int64_t ldx);     /* :646 */
int hiopamd_mat_add_diagonal_vec(hiopamd_ctx*, int n, double* A, int64_t lda, double alpha, const double* d); /* :703 */
int hiopamd_mat_add_diagonal_const(hiopamd_ctx*, int n, double* A, int64_t lda, double value);           /* :715 */
/* A[start+i][start+i] += alpha*d[src_start+i], i<num  (the three addSubDiagonal overloads :719,735,755) */
int hiopamd_mat_add_sub_diagonal(hiopamd_ctx*, double* A, int64_t lda, int start_on_dest_diag, double alpha,
                                 const double* d, int start_on_src_vec, int num_elems);
int hiopamd_mat_add_sub_diagonal_const(hiopamd_ctx*, double* A, int64_t lda, int start_on_dest_diag, int num_elems,
                                       double c);
int hiopamd_mat_add_matrix(hiopamd_ctx*, int m, int64_t n, double* A, int64_t lda, double alpha, const double* X,
                           int64_t ldx);                                                                /* :766 */
/* block of W += alpha*A^T at (row_start, col_start) of W (upper triangle)  :779 */
int hiopamd_mat_trans_add_to_sym_upper(hiopamd_ctx*, int m, int n, const double* A, int64_t lda, int row_start,
                                       int col_start, double alpha, double* W, int64_t ldw);
/* diagonal block of W += alpha*upper(A), A n x n  :810 */
int hiopamd_mat_add_upper_to_sym_upper(hiopamd_ctx*, int n, const double* A, int64_t lda, int diag_start,
                                       double alpha, double* W, int64_t ldw);
int hiopamd_mat_copy_rows_from(hiopamd_ctx*, int num_rows, int64_t n, double* A, int64_t lda, int row_dest,
                               const double* src, int64_t ldsrc);                                       /* :169 */
int hiopamd_mat_copy_rows_from_idx(hiopamd_ctx*, int num_rows, int64_t n, double* A, int64_t lda,
                                   const double* src, int64_t ldsrc, const int* rows_idxs);              /* :182 */
int hiopamd_mat_copy_block(hiopamd_ctx*, int m, int n, double* dst, int64_t lddst, const double* src,
                           int64_t ldsrc);                                                              /* :200,222 */
int hiopamd_mat_shift_rows(hiopamd_ctx*, int m, int64_t n, double* A, int64_t lda, int shift);           /* :238 */
int hiopamd_mat_max_abs(hiopamd_ctx*, int m, int64_t n, const double* A, int64_t lda, double* out_host);  /* :832 */
int hiopamd_mat_row_max_abs(hiopamd_ctx*, int m, int64_t n, const double* A, int64_t lda, double* ret_vec); /* :844 */
int hiopamd_mat_scale_rows(hiopamd_ctx*, int m, int64_t n, double* A, int64_t lda, const double* scal, int inv); /* :865 */
int hiopamd_mat_symmetrize(hiopamd_ctx*, int n, double* A, int64_t lda);                                  /* :912 (upper -> lower) */
int hiopamd_mat_is_finite(hiopamd_ctx*, int m, int64_t n, const double* A, int64_t lda, int* out_host);   /* :889 */

/* =====================================================================================
 * Weighted Gram kernels of the low-rank (quasi-Newton) KKT
 * (reference: src/Optimization/hiopHessianLowRank.cpp:1079 symmMatTimesDiagTimesMatTrans_local,
 *  :1119 matTimesDiagTimesMatTrans_local).  fp64 MFMA (v_mfma_f64_16x16x4_f64).
 * W(ma x mb) = beta*W + alpha * A(ma x n) * diag(d) * B(mb x n)^T ; d may be NULL (= ones).
 * If A==B && sym!=0 only the upper triangle is computed and it is mirrored onto the lower one
 * (reference :1079-1108: W[i][j] = W[j][i] = beta*W[i][j] + alpha*acc for j >= i).
 * ===================================================================================== */
int hiopamd_gram_weighted(hiopamd_ctx*, int ma, int mb, int64_t n, const double* A, int64_t lda, const double* B,
                          int64_t ldb, const double* d, double beta, double* W, int64_t ldw, double alpha,
                          int sym);

/* one pass over the long dimension for several right factors:
 * W(ma x (m0+m1+m2)) = beta*W + alpha * A diag(d) [B0;B1;B2]^T  -- the low-rank KKT needs X D X^T, X D S^T and
 * X D Y^T of the same X (reference: hiopHessianLowRank.cpp:566-583 makes three separate passes). */
int hiopamd_gram_weighted_stacked(hiopamd_ctx*, int ma, int64_t n, const double* A, int64_t lda, int m0,
                                  const double* B0, int64_t ldb0, int m1, const double* B1, int64_t ldb1, int m2,
                                  const double* B2, int64_t ldb2, const double* d, double beta, double* W, int64_t ldw,
                                  double alpha);

/* the four l x l blocks of hiopHessianLowRank::updateInternalBFGSRepresentation (hiopHessianLowRank.cpp:400-460: the three
 * weighted products :440-458) and of the middle matrix of the compact direct form in ONE pass over S, Y (l x n, leading dimension ld,
 * l <= 8) and DhInv:  G (4 l^2 doubles, l x l row-major each) =
 *   [ Y DhInv Y^T | S (sigma DhInv) Y^T | S (sigma (sigma DhInv - 1)) S^T | sigma S S^T ];  local sums (the caller all-reduces G). */
int hiopamd_gram_lowrank_blocks(hiopamd_ctx*, int l, int64_t n, const double* St, const double* Yt, int64_t ld,
                                const double* DhInv, double sigma, double* G);

/* =====================================================================================
 * hiopMatrixSparseTriplet (row-sorted COO, int32 indices)
 * (reference: src/LinAlg/hiopMatrixSparseTriplet.cpp)
 * ===================================================================================== */
int hiopamd_sp_times_vec(hiopamd_ctx*, int nrows, int ncols, int nnz, const int* iRow, const int* jCol,
                         const double* val, double beta, double* y, double alpha, const double* x);
/* the same product with the result stored a second time at y_copy (may be NULL) */
int hiopamd_sp_times_vec_copy(hiopamd_ctx*, int nrows, int ncols, int nnz, const int* iRow, const int* jCol, const double* val,
                              double beta, double* y, double alpha, const double* x, double* y_copy);      /* :73 */
int hiopamd_sp_trans_times_vec(hiopamd_ctx*, int nrows, int ncols, int nnz, const int* iRow, const int* jCol,
                               const double* val, double beta, double* y, double alpha, const double* x); /* :110 */
/* symbolic plan for  W[r0+i][c0+j] += alpha * sum_c M1[i,c]*M2[j,c]/D[c]  (pattern fixed across IPM iterations,
 * reference comment :479-489).  Index arrays are HOST pointers here (the plan is built once on the host). */
/* transTimesVec through a column-side plan (hiopMatrixSparseTriplet.cpp:110): the pattern (host index arrays, any order) is turned
 * ONCE into a per-column list of entries; the product is then one gather per column in list order — no atomics, bitwise
 * reproducible, one launch.  The plan-less hiopamd_sp_trans_times_vec gives the same guarantee for ANY index arrays by exact
 * (fixed-point) accumulation, at three launches. */
typedef struct hiopamd_sp_tplan hiopamd_sp_tplan;
int hiopamd_sp_tplan_create(hiopamd_sp_tplan** out, int nrows, int ncols, int nnz, const int* iRow_host, const int* jCol_host);
int hiopamd_sp_tplan_destroy(hiopamd_sp_tplan* plan);
int hiopamd_sp_tplan_trans_times_vec(hiopamd_ctx* ctx, const hiopamd_sp_tplan* plan, const double* val, double beta, double* y,
                                     double alpha, const double* x);
typedef struct hiopamd_sp_plan hiopamd_sp_plan;
int hiopamd_sp_plan_create(hiopamd_sp_plan** out, int m1, int m2, int ncols, int nnz1, const int* iRow1_host,
                           const int* jCol1_host, int nnz2, const int* iRow2_host, const int* jCol2_host,
                           int same_matrix_upper_only);
int hiopamd_sp_plan_destroy(hiopamd_sp_plan* plan);
int64_t hiopamd_sp_plan_num_outputs(const hiopamd_sp_plan* plan);
int64_t hiopamd_sp_plan_num_products(const hiopamd_sp_plan* plan);
/* addMDinvMtransToDiagBlockOfSymDeMatUTri :390 (same matrix) / addMDinvNtransToSymDeMatUTri :447 */
int hiopamd_sp_add_MDinvNt(hiopamd_ctx*, const hiopamd_sp_plan* plan, const double* val1, const double* val2,
                           const double* D, double alpha, double* W, int64_t ldw, int row_dest_start,
                           int col_dest_start);
/* y[start_dest+i] += alpha * diag(Msym)[i]  (hiopMatrixSymSparseTriplet::startingAtAddSubDiagonalToStartingAt :1018) */
int hiopamd_spsym_add_diag_to_vec(hiopamd_ctx*, int nnz, const int* iRow, const int* jCol, const double* val,
                                  double alpha, double* y, int vec_start, int n_vec, int diag_src_start,
                                  int num_elems);
/* y = beta*y + alpha*Msym*x for upper-triangle triplets (hiopMatrixSymSparseTriplet::timesVec :924-958) */
int hiopamd_spsym_times_vec(hiopamd_ctx*, int n, int nnz, const int* iRow, const int* jCol, const double* val,
                            double beta, double* y, double alpha, const double* x);
/* W upper += alpha * Msym (upper triangle entries) at diag_start (:980) */
int hiopamd_spsym_add_upper_to_sym_upper(hiopamd_ctx*, int nnz, const int* iRow, const int* jCol, const double* val,
                                         int diag_start, double alpha, double* W, int64_t ldw);

/* the Schur products scattered into a SPARSE destination: out_vals[pos[p]] += alpha * (output p of the plan) */
int hiopamd_sp_MDinvNt_scatter(hiopamd_ctx*, const hiopamd_sp_plan* plan, const double* val1, const double* val2,
                               const double* D, double alpha, double* out_vals, const int64_t* pos_dev);
int hiopamd_sp_plan_outputs(const hiopamd_sp_plan* plan, int* out_i_host, int* out_j_host);
/* the rest of the hiopMatrixSparseTriplet surface (one thread per triplet; used by the HiOp-side adapter) */
int hiopamd_sp_trans_add_to_sym_upper(hiopamd_ctx*, int nnz, const int* iRow, const int* jCol, const double* val,
                                      int row_start, int col_start, double alpha, double* W, int64_t ldw);   /* :255 */
int hiopamd_sp_row_max_abs(hiopamd_ctx*, int nrows, int nnz, const int* iRow, const double* val, double* ret_vec); /* :285 */
int hiopamd_sp_scale_rows(hiopamd_ctx*, int nnz, const int* iRow, double* val, const double* scal, int inv);  /* :303 */
int hiopamd_sp_copy_to_dense(hiopamd_ctx*, int nrows, int ncols, int nnz, const int* iRow, const int* jCol,
                             const double* val, double* W, int64_t ldw);                                      /* :363 */
int hiopamd_sp_indexes_ordered(hiopamd_ctx*, int nnz, const int* iRow, const int* jCol, int* out_host);        /* :377 */
int hiopamd_sp_num_offdiag(hiopamd_ctx*, int nnz, const int* iRow, const int* jCol, int64_t* out_host);        /* :1171,:1338 */
int hiopamd_sp_extract_diagonal(hiopamd_ctx*, int n, int nnz, const int* iRow, const int* jCol, const double* val,
                                double* diag);                                                                /* :1355 */

/* ---- the triplet ASSEMBLY surface (big triplet matrices out of small ones: sparse KKT classes, feasibility restoration);
 * reference src/LinAlg/hiopMatrixSparseTriplet.cpp, line of each method on the right.  `iRow, jCol, val` = the destination's
 * triplet arrays (device); sources must be sorted by (row, column) like hiopMatrixSparseTriplet keeps them. ---- */
int hiopamd_sp_copy_sub_diagonal_from(hiopamd_ctx*, int* iRow, int* jCol, double* val, int start_on_dest_diag, int num_elems,
                                      const double* d, int start_on_nnz_idx, double scal);                         /* :216 */
int hiopamd_sp_set_sub_diagonal_to(hiopamd_ctx*, int* iRow, int* jCol, double* val, int start_on_dest_diag, int num_elems,
                                   double c, int start_on_nnz_idx);                                                /* :235 */
int hiopamd_sp_copy_rows_from(hiopamd_ctx*, int* iRow, int* jCol, double* val, int nnz_src, const int* iRow_src,
                              const int* jCol_src, const double* val_src, const int* rows_idxs_dev, int n_rows);   /* :562 */
int hiopamd_sp_copy_rows_block_from(hiopamd_ctx*, int* iRow, int* jCol, double* val, int nnz_src, const int* iRow_src,
                                    const int* jCol_src, const double* val_src, int rows_src_idx_st, int n_rows,
                                    int rows_dest_idx_st, int dest_nnz_st);                                        /* :619 */
int hiopamd_sp_copy_diag_matrix_to_subblock(hiopamd_ctx*, int* iRow, int* jCol, double* val, double src_val, int dest_row_st,
                                            int dest_col_st, int dest_nnz_st, int nnz_to_copy);                    /* :671 */
int hiopamd_sp_copy_diag_matrix_to_subblock_w_pattern(hiopamd_ctx*, int* iRow, int* jCol, double* val, const double* dx,
                                                      int dest_row_st, int dest_col_st, int dest_nnz_st, int n,
                                                      const double* ix, int* nnz_found_host /* may be null */);   /* :689 */
/* trans = 0: copySubmatrixFrom (:1042); trans = 1: copySubmatrixFromTrans (:1076) */
int hiopamd_sp_copy_submatrix_from(hiopamd_ctx*, int* iRow, int* jCol, double* val, int nnz_src, const int* iRow_src,
                                   const int* jCol_src, const double* val_src, int dest_row_st, int dest_col_st,
                                   int dest_nnz_st, int offdiag_only, int trans);
/* rowpattern = 0: setSubmatrixToConstantDiag_w_colpattern (:1110); 1: ..._w_rowpattern (:1140) */
int hiopamd_sp_set_submatrix_to_constant_diag_w_pattern(hiopamd_ctx*, int* iRow, int* jCol, double* val, double scalar,
                                                        int dest_row_st, int dest_col_st, int dest_nnz_st, int n,
                                                        const double* ix, int rowpattern, int* nnz_found_host);
/* this = [Jc -I I 0 0; Jd 0 0 -I I]; iJacS / jJacS (both or none) and MJacS (device, may be null) are filled alongside */
int hiopamd_sp_set_jac_fr(hiopamd_ctx*, int* iRow, int* jCol, double* val, int n, int m_c, int nnz_c, const int* ic,
                          const int* jc, const double* vc, int m_d, int nnz_d, const int* id, const int* jd, const double* vd,
                          int* iJacS, int* jJacS, double* MJacS);                                                  /* :790 */
/* hiopMatrixSymSparseTriplet::set_Hess_FR: diagonal add_diag merged into / inserted in front of every row's entries */
int hiopamd_spsym_set_hess_fr(hiopamd_ctx*, int* iRow, int* jCol, double* val, int m_h, int nnz_h, const int* ih,
                              const int* jh, const double* vh, int n_diag, const double* add_diag, int* iHSS, int* jHSS,
                              double* MHSS);                                                                       /* :1374 */

/* =====================================================================================
 * Sparse condensed KKT matrix in CSR (first piece):
 *   M = Jd^T diag(Hd) Jd + H + Dx + delta_wx I    (hiopKKTLinSysCondensedSparse::build_kkt_matrix,
 *   src/Optimization/hiopKKTLinSysSparseCondensed.cpp:205-335; CSR operations of src/LinAlg/hiopMatrixSparseCSR.hpp:97-290)
 * Symbolic analysis once per sparsity pattern (host index arrays), numeric phase on device values.  The direct solver of the
 * reference (sparse Cholesky) is NOT part of this library; PCG / BiCGStab (hiopamd_krylov_*) run on the operator callbacks.
 * ===================================================================================== */
typedef struct hiopamd_csr_condensed hiopamd_csr_condensed;
int hiopamd_csr_condensed_create(hiopamd_csr_condensed** out, hiopamd_ctx* ctx, int n, int m, int nnzJ, const int* iJ_host,
                                 const int* jJ_host, int nnzH_upper, const int* iH_host, const int* jH_host);
int hiopamd_csr_condensed_destroy(hiopamd_csr_condensed* c);
int64_t hiopamd_csr_condensed_nnz(const hiopamd_csr_condensed* c);
int64_t hiopamd_csr_condensed_num_products(const hiopamd_csr_condensed* c);
/* ---- bordered-diagonal sparse direct solver (csrc/arrow_ldl.hip): the inner solver of the condensed sparse KKT when a small set
 * of "border" variables covers every off-diagonal entry of M (the arrowhead of the reference's sparse examples) — the role of the
 * reference's sparse Cholesky (MA57 / cuSOLVER, hiopKKTLinSysSparseCondensed.cpp:469-496): exact factorisation
 * M = L diag(D, S) L^T, exact inertia (Haynsworth), solves without host round trips.  create: the pattern of the full symmetric
 * matrix in CSR (host arrays); HIOPAMD_ERR_STATE when the pattern needs more than 32 border variables (nothing created). */
typedef struct hiopamd_arrow_ldl hiopamd_arrow_ldl;
int hiopamd_arrow_ldl_create(hiopamd_arrow_ldl** out, hiopamd_ctx* ctx, int n, const int* rowptr_host, const int* colidx_host);
int hiopamd_arrow_ldl_destroy(hiopamd_arrow_ldl* s);
int hiopamd_arrow_ldl_border(const hiopamd_arrow_ldl* s, int* p_host, int* border_host);
int hiopamd_arrow_ldl_factorize(hiopamd_arrow_ldl* s, const double* csr_values, int* n_neg_host, int* n_zero_host);
int hiopamd_arrow_ldl_solve(hiopamd_arrow_ldl* s, double* x_inout);

/* General sparse symmetric direct solver M = P^T L D L^T P (csrc/sparse_ldl.hip; the sparse-Cholesky role of
 * hiopKKTLinSysSparseCondensed.cpp:469-496 for patterns that are neither small nor bordered diagonals): nested-dissection ordering,
 * supernodal multifrontal factorisation by tree levels on the device (fronts of <= 128 rows in LDS), the top of the tree as ONE dense
 * root factored by hiopamd_linsolver.  No numerical pivoting: n_neg / n_zero are the counts of pivots below -1e-14 / of magnitude below
 * 1e-14, M is positive definite <=> both are 0 — a verdict that does not depend on any right-hand side.
 * Pattern = full symmetric CSR on the host, columns ascending inside a row, diagonal structurally full; values = device array aligned
 * with it.  create returns HIOPAMD_ERR_STATE when the dense root would exceed 20480 (large patterns without small separators).
 * _analyse / _plan are host-only (no device): the ordering and the gather plans, for tests that replay the numeric phase. */
typedef struct hiopamd_sparse_ldl hiopamd_sparse_ldl;
int hiopamd_sparse_ldl_create(hiopamd_sparse_ldl** out, hiopamd_ctx* ctx, int n, const int* rowptr_host, const int* colidx_host);
int hiopamd_sparse_ldl_destroy(hiopamd_sparse_ldl* s);
int hiopamd_sparse_ldl_info(const hiopamd_sparse_ldl* s, int64_t* info8_host);   /* supernodes, fronts, levels, root order, nnz(L), levels factored with the front in registers */
int hiopamd_sparse_ldl_factorize(hiopamd_sparse_ldl* s, const double* csr_values, int* n_neg_host, int* n_zero_host);
int hiopamd_sparse_ldl_solve(hiopamd_sparse_ldl* s, double* x_inout);
int hiopamd_sparse_ldl_analyse(int n, const int* rowptr_host, const int* colidx_host, int64_t* info8_host, int* perm_host);
int hiopamd_sparse_ldl_plan(int n, const int* rowptr_host, const int* colidx_host, int64_t* sizes16, int* level_ptr, int* f_nc, int* f_nr,
                            int64_t* f_lofs, int64_t* f_uofs, int64_t* f_vofs, int64_t* f_iofs, int* fidx, int* mat_dest, int64_t* mat_ptr,
                            int64_t* mat_src, int64_t* mat_front, int* vec_dest, int64_t* vec_ptr, int64_t* vec_src, int64_t* vec_front,
                            int* rmat_dest, int64_t* rmat_ptr, int64_t* rmat_src, int* rvec_dest, int64_t* rvec_ptr, int64_t* rvec_src,
                            int* root_old);
int hiopamd_csr_condensed_pattern(const hiopamd_csr_condensed* c, int* rowptr_host, int* colidx_host);
const int* hiopamd_csr_condensed_rowptr(const hiopamd_csr_condensed* c);   /* device */
const int* hiopamd_csr_condensed_colidx(const hiopamd_csr_condensed* c);   /* device */
double* hiopamd_csr_condensed_values(hiopamd_csr_condensed* c);            /* device */
int hiopamd_csr_condensed_numeric(hiopamd_csr_condensed* c, const double* J_val, const double* H_val, const double* Hd,
                                  const double* Dx, double delta_wx);
/* hiopamd_linop_fn callbacks (user = the hiopamd_csr_condensed*): y = M x ; y = x ./ diag(M) (Jacobi preconditioner) */
int hiopamd_csr_condensed_apply(void* user, const double* x_dev, double* y_dev);
int hiopamd_csr_condensed_refresh_jt(hiopamd_csr_condensed* c, const double* J_val);   /* new Jacobian values for the Jd^T copy */
/* y (n) = beta y + alpha Jd^T x (m), Jd^T kept in CSR by the object (values of the last numeric phase): no floating-point atomics */
int hiopamd_csr_condensed_jac_trans_times_vec(hiopamd_csr_condensed* c, double beta, double* y, double alpha, const double* x);
/* y (m) = beta y + alpha Jd x on the caller's own row-sorted triplet arrays (device column indices and values): rows of a few entries one
 * thread each, long rows in chunks; HIOPAMD_ERR_STATE if the triplets given at creation were not row-sorted */
int hiopamd_csr_condensed_jac_times_vec(hiopamd_csr_condensed* c, const int* jJ_dev, const double* J_val, double beta, double* y,
                                        double alpha, const double* x);
int hiopamd_csr_condensed_diagonal(hiopamd_csr_condensed* c, double* diag_dev);   /* diag(M), device, n */
int hiopamd_csr_condensed_jacobi(void* user, const double* x_dev, double* y_dev);
/* generic CSR kernels (int32 row pointers / column indices on the device) */
int hiopamd_csr_times_vec(hiopamd_ctx*, int nrows, const int* rowptr, const int* colidx, const double* val, double beta,
                          double* y, double alpha, const double* x);
int hiopamd_csr_extract_diagonal(hiopamd_ctx*, int n, const int* rowptr, const int* colidx, const double* val, double* diag); /* :97 */
int hiopamd_csr_set_diagonal(hiopamd_ctx*, int n, const int* rowptr, const int* colidx, double* val, double value);          /* :105 */
int hiopamd_csr_scale_rows(hiopamd_ctx*, int n, const int* rowptr, double* val, const double* D);                            /* :178 */
int hiopamd_csr_scale_cols(hiopamd_ctx*, int64_t nnz, const int* colidx, double* val, const double* D);                      /* :175 */
int hiopamd_csr_form_diag_symbolic(hiopamd_ctx*, int n, int* rowptr, int* colidx);                                           /* :244 */
int hiopamd_csr_form_diag_numeric(hiopamd_ctx*, int n, double* val, const double* D);                                        /* :255 */

/* =====================================================================================
 * hiopKKTLinSysCondensedSparse — the condensed sparse KKT of the inequality-only sparse formulation (SURVEY section 8 row f2 /
 * BASELINE configs[4]); reference: src/Optimization/hiopKKTLinSysSparseCondensed.cpp:105-335 (build_kkt_matrix), :346-401
 * (solve_compressed_direct), :403-449 (solveCompressed):
 *   (H + Dx + delta_wx I + Jd^T (Dd + delta_wd I) Jd) dx = rx + Jd^T ((Dd + delta_wd I) ryd + rd)
 *   dd = Jd dx - ryd ;  dyd = (Dd + delta_wd I) dd - rd
 * The reference's inner solver is a sparse Cholesky (MA57 / cuSOLVER, :469-496: not in the image, not restated); here the
 * condensed matrix is assembled in CSR on the device (hiopamd_csr_condensed) and solved by PCG with the Jacobi preconditioner
 * (hiopamd_krylov_*, kind 0).  Index arrays are HOST pointers (symbolic phase once per pattern); values are device pointers.
 * ===================================================================================== */
typedef struct hiopamd_kkt_sparse_condensed hiopamd_kkt_sparse_condensed;
int hiopamd_kkt_sparse_condensed_create(hiopamd_kkt_sparse_condensed** out, hiopamd_ctx* ctx, int nx, int nineq, int nnzJ,
                                        const int* iJ_host, const int* jJ_host, int nnzH_upper, const int* iH_host,
                                        const int* jH_host);
int hiopamd_kkt_sparse_condensed_destroy(hiopamd_kkt_sparse_condensed* k);
/* Jd triplet values (row-sorted), Hessian upper-triangle triplet values, Dx (nx), Dd (nineq, WITHOUT delta_wd); borrowed */
int hiopamd_kkt_sparse_condensed_set_values(hiopamd_kkt_sparse_condensed* k, const double* Jd_val, const double* H_val,
                                            const double* Dx, const double* Dd);
int hiopamd_kkt_sparse_condensed_set_diagonals(hiopamd_kkt_sparse_condensed* k, const double* Dx, const double* Dd);
/* build_kkt_matrix (:105) with scalar / vector perturbations (vectors: delta_wx over nx, delta_wd over nineq; null = zero) */
int hiopamd_kkt_sparse_condensed_build(hiopamd_kkt_sparse_condensed* k, double delta_wx, double delta_wd);
int hiopamd_kkt_sparse_condensed_build_vec(hiopamd_kkt_sparse_condensed* k, const double* delta_wx, const double* delta_wd);
/* the answer matrixChanged() of the reference's Cholesky gives the inertia-correction loop: *n_neg_host = 0 when M can be
 * positive definite (every diagonal entry positive and finite), -1 otherwise; definiteness along the search directions is
 * tested by every solve (PCG's curvature test) */
int hiopamd_kkt_sparse_condensed_factorize(hiopamd_kkt_sparse_condensed* k, int* n_neg_host);
/* solveCompressed (:403): rx (nx), rd, ryd (nineq) in; dx (nx), dd, dyd (nineq) out; *ok_host = 0 when the inner solve failed
 * (negative curvature met or no convergence) — the reference's `return false` */
int hiopamd_kkt_sparse_condensed_solve_compressed(hiopamd_kkt_sparse_condensed* k, const double* rx, const double* rd,
                                                  const double* ryd, double* dx, double* dd, double* dyd, int* ok_host);
int hiopamd_kkt_sparse_condensed_set_inner_solver(hiopamd_kkt_sparse_condensed* k, double tol, int max_iter);   /* default 1e-12, 2000 */
int hiopamd_kkt_sparse_condensed_last_solve(const hiopamd_kkt_sparse_condensed* k, int* flag_host, double* iters_host,
                                            double* rel_resid_host);
int hiopamd_kkt_sparse_condensed_dims(const hiopamd_kkt_sparse_condensed* k, int* dims4_host /* nx, nineq, nnzJ, nnzH */);
/* the inner solver in use: 0 dense LDL^T of the expanded matrix (nx <= 4096), 1 bordered-diagonal direct solver (border <= 32),
 * 3 general sparse LDL^T (hiopamd_sparse_ldl: nested dissection + multifrontal + dense root), 2 PCG + Jacobi (only when the sparse LDL^T's
 * dense root would exceed its limit) */
int hiopamd_kkt_sparse_condensed_inner_kind(const hiopamd_kkt_sparse_condensed* k);
/* the analysis of the general sparse LDL^T when that is the inner solver (hiopamd_sparse_ldl_info: supernodes, fronts, levels, root order,
 * nnz(L), 0, 0, 0); HIOPAMD_ERR_STATE for the other inner solvers */
int hiopamd_kkt_sparse_condensed_ldl_info(const hiopamd_kkt_sparse_condensed* k, int64_t* info8_host);
hiopamd_csr_condensed* hiopamd_kkt_sparse_condensed_matrix(hiopamd_kkt_sparse_condensed* k);
double* hiopamd_kkt_sparse_condensed_Hd(hiopamd_kkt_sparse_condensed* k);
/* y = beta y + alpha Hess x ; y = beta y + alpha Jd x ; y = beta y + alpha Jd^T x  on the values of the last set_values */
int hiopamd_kkt_sparse_condensed_hess_times_vec(hiopamd_kkt_sparse_condensed* k, double beta, double* y, double alpha, const double* x);
int hiopamd_kkt_sparse_condensed_jac_times_vec(hiopamd_kkt_sparse_condensed* k, double beta, double* y, double alpha, const double* x);
int hiopamd_kkt_sparse_condensed_jac_trans_times_vec(hiopamd_kkt_sparse_condensed* k, double beta, double* y, double alpha,
                                                     const double* x);

/* =====================================================================================
 * hiopLinSolverSymDense operator — no-pivot blocked LDL^T on fp64 MFMA + inertia
 * (reference: src/LinAlg/hiopLinSolver.hpp:78-130; semantics of
 *  src/LinAlg/hiopLinSolverSymDenseMagma.cpp:324-480 (MagmaNopiv) and the inertia thresholds of
 *  src/LinAlg/hiopLinSolverSymDenseLapack.hpp:154-161).
 * ===================================================================================== */
typedef struct hiopamd_linsolver hiopamd_linsolver;
int hiopamd_linsolver_create(hiopamd_linsolver** out, hiopamd_ctx* ctx, int n);
int hiopamd_linsolver_destroy(hiopamd_linsolver* ls);
/* device pointer of the n x n row-major system matrix (ld = n); the KKT class writes its upper triangle.  After a successful
 * hiopamd_linsolver_matrix_changed it holds the factor (U above the diagonal, D on it).  An order n >= 1024 that the kernels' fast
 * forms do not take (odd; not a multiple of 256 / 512) is factored and solved internally as diag(M, I) of a larger order in a padded
 * copy (n = 8191: 5.3 instead of 8.9 ms; n = 8000: 5.8 instead of 6.2 ms per factorisation + 3 solves); the caller sees no difference —
 * same matrix view, same factor in it, same inertia — except that the matrix is not written before the factorisation has succeeded. */
double* hiopamd_linsolver_sys_matrix(hiopamd_linsolver* ls);
/* For callers that can assemble at ANY pitch (the native KKT objects of this library): where to write the upper triangle of the next
 * matrix — for an object that works at a padded order, the padded copy itself (*ld_out = that order), and the next
 * hiopamd_linsolver_matrix_changed factors it where it is (no copy in, no copy back).  Call it before every such assembly: it arms ONE
 * matrixChanged.  After a factorisation of this kind hiopamd_linsolver_sys_matrix does not show the factor until
 * hiopamd_linsolver_sys_matrix_sync has been called, the assembled matrix is overwritten, and an expired wait is reported as
 * HIOPAMD_ERR_TIMEOUT (assemble again and call again).  Every other object / safe mode / pivoted mode: the matrix of
 * hiopamd_linsolver_sys_matrix, *ld_out = n, nothing changes. */
int hiopamd_linsolver_assembly_matrix(hiopamd_linsolver* ls, double** M_out, int64_t* ld_out);
int hiopamd_linsolver_sys_matrix_sync(hiopamd_linsolver* ls);
/* host only: the order a solver object of order n works at — n, n made even, or the next multiple of 256 / 512 (a cost model of one
 * factorisation + three solves + the copies; n < 1024: n) */
int hiopamd_ldlt_padded_order(int n);
int hiopamd_linsolver_n(const hiopamd_linsolver* ls);
/* matrixChanged(): factorise in place; *n_neg_host = number of negative pivots, or -1 if a pivot is
 * (numerically) zero / non-finite -- the reference's "singular" return.
 * STREAM CONTRACT: the call returns when the pivot flags and the inertia have reached the host; kernels that only prepare the next
 * solve (inverted diagonal blocks, ~0.1 ms) may still be running on the context's stream and still READ the matrix.  Everything this
 * library does next is queued behind them on that stream.  A caller that touches sys_matrix() by other means — another stream, a
 * blocking hipMemcpy (the context's own stream is non-blocking: the null stream does not wait for it) — calls hiopamd_ctx_sync first;
 * the C++ adapter's matrixChanged() does (adapters/hiopLinSolverSymDenseHipNative.cpp). */
int hiopamd_linsolver_matrix_changed(hiopamd_linsolver* ls, int* n_neg_host);
/* 1 when the last matrixChanged() left a factor solve() can use (also with negative or tiny pivots: the answer -1 for a null pivot by
 * the reference's threshold does not mean there is no factor), 0 after an exactly zero / non-finite pivot or before the first call */
int hiopamd_linsolver_factored(const hiopamd_linsolver* ls);
/* solve(): rhs (device, length n * nrhs, column after column) overwritten by the solution */
int hiopamd_linsolver_solve(hiopamd_linsolver* ls, double* rhs_inout, int nrhs);
/* last factorisation: pos/neg/zero pivot counts (magmablas_ddiinertia equivalent) */
int hiopamd_linsolver_inertia(const hiopamd_linsolver* ls, int* pos_host, int* neg_host, int* zero_host);
/* hiopLinSolStats::flopsFact / flopsTriuSolves (src/Utils/hiopRunStats.hpp:262-270), cumulative over the object's life:
 * n^3/3 per matrixChanged, 2 n^2 per right-hand side; the times are the HIOPAMD_SPAN_LINSOLV_* spans of the context */
/* The triangular solves run as ONE dataflow launch (a task graph over the 256 x 256 blocks of U with inter-workgroup
 * waits).  Every wait is bounded (2 s of wall-clock time); a time-out raises a device-side error word that the next
 * synchronising call sees: hiopamd_linsolver_solve_status synchronises the context's stream and returns *ok_host = 0 if a
 * solve since the last check was invalid (the object then re-initialises its exchange buffers and uses the stepwise
 * 256-row solve from there on); matrixChanged performs the same check.  hiopamd_linsolver_set_solve_dataflow(ls, 0)
 * selects the stepwise solve explicitly (HIOPAMD_SOLVE_FLOW=0 in the environment sets that default). */
int hiopamd_linsolver_solve_status(hiopamd_linsolver* ls, int* ok_host);
/* the same question WITHOUT a synchronisation: *ok_host = 0 if the host already knows that a solve since the last factorisation
 * failed (a safe-mode refinement that did not converge — known when hiopamd_linsolver_solve returns —, or a dataflow time-out
 * seen by an earlier synchronising call).  The KKT objects call it after every solve and report ok = 0 upwards. */
int hiopamd_linsolver_last_solve_ok(hiopamd_linsolver* ls, int* ok_host);
int hiopamd_linsolver_set_solve_dataflow(hiopamd_linsolver* ls, int enable);
/* Safe mode — the role of the reference's switch from the no-pivot to the Bunch-Kaufman solver
 * (src/Optimization/hiopKKTLinSysMDS.cpp:408-430, hiopAlgFilterIPM.cpp:2400-2427).  enable != 0: matrixChanged keeps a copy of
 * the assembled matrix, factors K + delta*diag(+I_npos, -I_rest) with delta = sqrt(eps)*max|K_ij| (static quasi-definite
 * regularisation: element growth <= ~||K||/delta), and every solve is refined against the saved K until
 * ||b - K x||_inf <= 1e-13 (||K|| ||x|| + ||b||); a refinement that does not get there in 10 steps makes
 * hiopamd_linsolver_solve_status report *ok_host = 0.  n_pos_block = order of the leading positive-definite block (the x part of
 * the condensed KKT).  hiopamd_linsolver_growth: max |u_ij| and the extreme |d_i| of the last factorisation, the signal a
 * caller uses to decide that the fast path misbehaves. */
int hiopamd_linsolver_set_safe_mode(hiopamd_linsolver* ls, int enable, int n_pos_block);
/* Pivoted mode -- the reference's safe solver itself: Bunch-Kaufman partial pivoting (hiopLinSolverSymDenseMagmaBuKa,
 * hiopLinSolverSymDenseMagma.cpp:120-250; hiopLinSolverSymDenseLapack.hpp:75-195 = LAPACK DSYTRF / DSYTRS).  matrixChanged returns
 * -1 for a singular matrix (DSYTRF INFO > 0 or a null pivot by the reference's 1e-14 rule) and the exact number of negative
 * eigenvalues otherwise, for ANY symmetric matrix; solve is DSYTRS.  Takes precedence over set_safe_mode.  Cost: one launch per
 * 64-column panel whose column steps are bound by the pivot search's dependent round trips (93 ms per factorisation, 1.2 ms per solve
 * at n = 8192 against 5.2 / 0.18 ms without pivoting) -- the exceptional path, as in the reference. */
int hiopamd_linsolver_set_pivoting(hiopamd_linsolver* ls, int enable);

/* The pivoted factorisation on its own (csrc/ldlt_bk.hip): P A P^T = L D L^T, A n x n row-major with its UPPER triangle populated
 * (= LAPACK's UPLO = 'L' on the same memory), overwritten by L (strictly below the diagonal in the column-major reading), the diagonal
 * of D; the off-diagonals of the 2 x 2 blocks are kept in the object.  inertia3_host = (pos, neg, null) by the reference's rule
 * (hiopLinSolverSymDenseLapack.hpp:127-167); info_host = DSYTRF's INFO.  Pivots and D equal LAPACK's; the row interchanges are applied
 * to all previous columns at once (one permutation), so L differs from DSYTRF's by row order only.  _solve: the matrix must be the one
 * _factor left (from the second solve with the same address on, the sweeps are replayed as a HIP graph). */
typedef struct hiopamd_ldlt_bk hiopamd_ldlt_bk;
int hiopamd_ldlt_bk_create(hiopamd_ldlt_bk** out, hiopamd_ctx* ctx, int n);
int hiopamd_ldlt_bk_destroy(hiopamd_ldlt_bk* b);
int hiopamd_ldlt_bk_factor(hiopamd_ldlt_bk* b, double* A, int64_t lda, int* inertia3_host, int* info_host);
int hiopamd_ldlt_bk_solve(hiopamd_ldlt_bk* b, const double* A, int64_t lda, double* x_inout, int nrhs);
int hiopamd_ldlt_bk_pivots(hiopamd_ldlt_bk* b, int* ipiv_host, int* perm_host, double* e_host);
/* The panel kernel runs on up to 16 workgroups that wait for each other (one grid barrier per column): they must be resident together.
 * If a barrier expires (a shared or busy device) _factor returns HIOPAMD_ERR_TIMEOUT with the matrix partly overwritten — re-assemble
 * and call again — and the object factors with ONE workgroup from then on: nobody to wait for, same pivots, same factor, slower.
 * _set_single_workgroup selects that form by hand (tests; a device known to be shared). */
int hiopamd_ldlt_bk_set_single_workgroup(hiopamd_ldlt_bk* b, int enable);
int hiopamd_linsolver_safe_mode_info(const hiopamd_linsolver* ls, int* refinements_host, double* residual_rel_host);
int hiopamd_linsolver_growth(hiopamd_linsolver* ls, double* max_abs_u_host, double* min_abs_d_host, double* max_abs_d_host);
/* The factorisation runs as a dataflow of two persistent kernels (csrc/ldlt_dataflow.hpp) when the CU-masked streams are
 * available and n >= 768; enable = 0 selects the stepwise kernels (one launch per super-panel step) — same results to
 * rounding, for A/B timing and as a fallback.  HIOPAMD_DF=0 in the environment sets the default off. */
int hiopamd_linsolver_set_dataflow(hiopamd_linsolver* ls, int enable);
/* Every wait of the dataflow kernels is bounded; one that expires aborts the factorisation with the matrix overwritten.  So that
 * matrixChanged() keeps the reference's contract — "number of negative eigenvalues, or -1 for a singular matrix", nothing else
 * (src/LinAlg/hiopLinSolver.hpp:117-130) — the object takes a copy of the upper triangle before a dataflow factorisation
 * (~0.13 ms at n = 8192) and, after a time-out, restores it and factorises with the stepwise kernels inside the same call: the
 * caller never sees HIOPAMD_ERR_TIMEOUT.  enable = 0 drops the copy — for callers that can re-assemble (the native KKT objects
 * do): they get HIOPAMD_ERR_TIMEOUT and call again.  Default: on.  hiopamd_linsolver_timeouts: how many bounded waits have
 * expired over the object's life (0 in every soak of the default configuration, profiles/r04_soak). */
int hiopamd_linsolver_set_retry_copy(hiopamd_linsolver* ls, int enable);
int hiopamd_linsolver_timeouts(const hiopamd_linsolver* ls, int64_t* count_host);
/* static schedule of the dataflow factorisation for order n (host only): see csrc/ldlt.hip */
int hiopamd_ldlt_dataflow_plan(int n, int* dims8_host, int* chain_tasks_host, int* wide_tasks_host, int64_t wide_cap);
/* the wide kernel's task queues for order n (host only): per super-panel {first TR task, TR tasks, first NEAR update task,
 * NEAR update tasks, NEAR tasks of the first two tile rows} as indices into the wide task list of hiopamd_ldlt_dataflow_plan;
 * _far_queues: per super-panel {first FAR update task, FAR tasks, the super-panel whose FAR list feeds this NEAR list (-1: none),
 * how many of that list's tasks must be taken before a NEAR task of this super-panel may be} (csrc/ldlt_wide_body.inc) */
int hiopamd_ldlt_dataflow_queues(int n, int* queues5_host, int cap_panels);
int hiopamd_ldlt_dataflow_far_queues(int n, int* queues4_host, int cap_panels);
int hiopamd_ldlt_dataflow_nvb(int n);   /* row-panel workspaces a solver of order n allocates (one per super-panel up to 2 GB) */
int hiopamd_linsolver_flops(const hiopamd_linsolver* ls, double* flops_fact_host, double* flops_triu_solves_host);
/* per-launch HIP-event timing of the MFMA rank-K update kernel (bench / roofline only; off by default).
 * read: accumulated kernel milliseconds, algorithmic flops (2*K per updated element) and launch count
 * since the last hiopamd_linsolver_profile(ls, 1). */
int hiopamd_linsolver_profile(hiopamd_linsolver* ls, int enable);
int hiopamd_linsolver_profile_read(const hiopamd_linsolver* ls, double* update_ms_host, double* update_flops_host,
                                   int64_t* update_launches_host);
/* stand-alone entry points on caller-owned storage */
int hiopamd_ldlt_factor(hiopamd_ctx*, int n, double* A, int64_t lda, double* work_dinv /* n doubles */,
                        int* inertia3_host /* pos,neg,zero */);
int hiopamd_ldlt_solve(hiopamd_ctx*, int n, const double* A, int64_t lda, const double* work_dinv, double* rhs_inout,
                       int nrhs);
/* SPD solve with equilibration + refinement for the k x k low-rank KKT reduced system
 * (reference: hiopKKTLinSysLowRank::solveWithRefin, src/Optimization/hiopKKTLinSys.cpp:1192-1330). */
int hiopamd_posv_refine(hiopamd_ctx*, int k, const double* N_upper, int64_t ldn, double* rhs_inout,
                        double* work /* 3*k*k + 8*k doubles */, int* info_host, double* resid_rel_host);

/* =====================================================================================
 * KKT objects (stateful): condensed MDS KKT and quasi-Newton low-rank KKT
 * ===================================================================================== */
/* hiopKKTLinSysCompressedMDSXYcYd (reference: src/Optimization/hiopKKTLinSysMDS.cpp:112-403) */
typedef struct hiopamd_kkt_mds hiopamd_kkt_mds;
typedef struct {
  int nxs, nxd, neq, nineq;
  /* sparse Jacobian blocks, row-sorted COO; index arrays given BOTH on device (kernels) and host (plan) */
  int nnz_Jcs; const int* Jcs_i; const int* Jcs_j; const int* Jcs_i_host; const int* Jcs_j_host;
  int nnz_Jds; const int* Jds_i; const int* Jds_j; const int* Jds_i_host; const int* Jds_j_host;
  /* sparse Hessian block (diagonal entries used), COO */
  int nnz_Hss; const int* Hss_i; const int* Hss_j;
} hiopamd_mds_structure;
int hiopamd_kkt_mds_create(hiopamd_kkt_mds** out, hiopamd_ctx* ctx, const hiopamd_mds_structure* s);
int hiopamd_kkt_mds_destroy(hiopamd_kkt_mds* k);
/* values for the current iterate (device pointers, kept by reference until the next call):
 * Jcs_val, Jds_val, Hss_val (COO values), Jcd (neq x nxd), Jdd (nineq x nxd), Hdd (nxd x nxd, upper used),
 * Dx (nxs+nxd: log-barrier diagonal, sparse part first), Dd (nineq: (Sdl)^-1 Vl + (Sdu)^-1 Vu, WITHOUT delta_wd) */
int hiopamd_kkt_mds_set_values(hiopamd_kkt_mds* k, const double* Jcs_val, const double* Jds_val, const double* Hss_val,
                               const double* Jcd, const double* Jdd, const double* Hdd, const double* Dx,
                               const double* Dd);
/* build_kkt_matrix (:172) with scalar inertia-correction perturbations, then factorizeWithCurvCheck (:78).
 * *n_neg_host = #negative eigenvalues of the full XYcYd system (dense part + sparse (1,1) block), or -1. */
int hiopamd_kkt_mds_build(hiopamd_kkt_mds* k, double delta_wx, double delta_wd, double delta_cc, double delta_cd);
/* the same with the perturbations as device VECTORS — the reference's actual form (hiopKKTLinSysMDS.cpp:178-181, added entry by
 * entry at :213-215, :223-227, :245, :280, :289-290; hiopPDPerturbationPrimalFirstRand / DualFirstRand fill them with different
 * values per entry, hiopPDPerturbation.hpp:296,358): delta_wx over the nxs + nxd primal variables (sparse first), delta_wd and
 * delta_cd over the nineq inequalities, delta_cc over the neq equalities.  A null pointer stands for a zero vector. */
int hiopamd_kkt_mds_build_vec(hiopamd_kkt_mds* k, const double* delta_wx, const double* delta_wd, const double* delta_cc,
                              const double* delta_cd);
int hiopamd_kkt_mds_factorize(hiopamd_kkt_mds* k, int* n_neg_host);
/* solveCompressed (:307): all device vectors; rx (nxs+nxd), ryc (neq), ryd (nineq) are inputs (ryd is
 * overwritten like in the reference), dx, dyc, dyd outputs. */
int hiopamd_kkt_mds_solve_compressed(hiopamd_kkt_mds* k, const double* rx, const double* ryc, double* ryd,
                                     double* dx, double* dyc, double* dyd);
/* solveCompressed returns bool in the reference (:307); here the solves are asynchronous, so the answer is a separate question:
 * *ok_host = 0 if a solveCompressed since the last call is known to have delivered an invalid direction (safe-mode refinement
 * not converged, dataflow solve timed out).  sync = 0: what the host knows without waiting (free after a safe-mode solve,
 * which synchronises anyway); sync != 0: synchronise the stream and look at the dataflow solve's error word as well. */
int hiopamd_kkt_mds_solve_status(hiopamd_kkt_mds* k, int sync, int* ok_host);
/* only the log-barrier diagonals change (hiopKKTLinSysCompressedXYcYd::update, hiopKKTLinSys.cpp:562-572) */
int hiopamd_kkt_mds_set_diagonals(hiopamd_kkt_mds* k, const double* Dx, const double* Dd);
/* safe_mode_ of hiopKKTLinSysCompressedMDSXYcYd (:145, :408-430): enable = 1 regularise-and-refine (hiopamd_linsolver_set_safe_mode),
 * enable = 2 the pivoted Bunch-Kaufman solver (hiopamd_linsolver_set_pivoting: what the reference's safe mode runs), 0 off */
int hiopamd_kkt_mds_set_safe_mode(hiopamd_kkt_mds* k, int enable);
int hiopamd_kkt_mds_dims(const hiopamd_kkt_mds* k, int* dims4_host /* nxs, nxd, neq, nineq */);
/* the MDS matrices' products, on the values of the last set_values
 * (hiopMatrixSymBlockDiagMDS::timesVec hiopMatrixMDS.hpp:310, hiopMatrixMDS::timesVec :68 / transTimesVec :75);
 * which: 0 = Jac_c, 1 = Jac_d */
int hiopamd_kkt_mds_hess_times_vec(hiopamd_kkt_mds* k, double beta, double* y, double alpha, const double* x);
int hiopamd_kkt_mds_jac_times_vec(hiopamd_kkt_mds* k, int which, double beta, double* y, double alpha, const double* x);
int hiopamd_kkt_mds_jac_trans_times_vec(hiopamd_kkt_mds* k, int which, double beta, double* y, double alpha,
                                        const double* x);
/* W (m x m, ld ldw, upper triangle) = [Jc; Jd][Jc; Jd]^T (hiopMatrixMDS::timesMatTrans, hiopMatrixMDS.hpp:94-100) */
int hiopamd_kkt_mds_jac_jac_trans(hiopamd_kkt_mds* k, double* W, int64_t ldw);
double* hiopamd_kkt_mds_Dd_inv(hiopamd_kkt_mds* k);       /* device, nineq: 1/(Dd + delta_wd) of the last build */
double* hiopamd_kkt_mds_sys_matrix(hiopamd_kkt_mds* k);   /* device, N x N row-major, N = nxd+neq+nineq */
double* hiopamd_kkt_mds_Hxs(hiopamd_kkt_mds* k);          /* device, nxs */
hiopamd_linsolver* hiopamd_kkt_mds_linsolver(hiopamd_kkt_mds* k);

/* =====================================================================================
 * Quasi-Newton low-rank path: hiopHessianLowRank + hiopKKTLinSysLowRank
 * (reference: src/Optimization/hiopHessianLowRank.cpp, src/Optimization/hiopKKTLinSys.cpp:1030-1330).
 * All n-sized arguments are the LOCAL column slice of a rank (n_local); k x k / 2l x 2l objects are
 * replicated; the context's all-reduce hook sums the small blocks across the column partition.
 * sigma_update_strategy: 1 sty, 2 sty_inv, 3 snrm_ynrm, 4 sty_srnm_ynrm, 5 sigma0 (:69-73, :111-122).
 * ===================================================================================== */
typedef struct hiopamd_hess_lowrank hiopamd_hess_lowrank;
int hiopamd_hess_lowrank_create(hiopamd_hess_lowrank** out, hiopamd_ctx* ctx, int64_t n_local, int m_eq, int m_ineq,
                                int l_max, double sigma0, int sigma_update_strategy);
int hiopamd_hess_lowrank_destroy(hiopamd_hess_lowrank* h);
/* update (:262): secant pair from the current iterate; *stored_host = 1 if (s,y) entered the memory */
int hiopamd_hess_lowrank_update(hiopamd_hess_lowrank* h, const double* x, const double* grad_f, const double* Jc,
                                const double* Jd, const double* yc, const double* yd, int* stored_host);
int hiopamd_hess_lowrank_update_log_barrier_diagonal(hiopamd_hess_lowrank* h, const double* Dx);      /* :197 */
int hiopamd_hess_lowrank_solve(hiopamd_hess_lowrank* h, const double* rhs, double* x);                /* :495 */
/* W(k x k) = beta*W + alpha*X*(B+Dx)^-1*X^T (:549); work: k*(k+2l_max) + 2*k*l_max doubles */
int hiopamd_hess_lowrank_sym_mat_times_inverse_times_mat_trans(hiopamd_hess_lowrank* h, double beta, double* W, int k,
                                                               double alpha, const double* X, double* work);
/* timesVecCmn (:974-1059): y = beta*y + alpha*(B0 [+ Dx] + low-rank part)*x; hiopHessianLowRank::timesVec (:1061)
 * is add_log_barrier_term = 0, timesVec_noLogBarrierTerm's counterpart with the term is 1 */
int hiopamd_hess_lowrank_times_vec(hiopamd_hess_lowrank* h, double beta, double* y, double alpha, const double* x,
                                   int add_log_barrier_term);
int hiopamd_hess_lowrank_l_curr(const hiopamd_hess_lowrank* h);
double hiopamd_hess_lowrank_sigma(const hiopamd_hess_lowrank* h);
double* hiopamd_hess_lowrank_St(hiopamd_hess_lowrank* h);   /* l_max x n_local, rows 0..l_curr-1 valid */
double* hiopamd_hess_lowrank_Yt(hiopamd_hess_lowrank* h);

typedef struct hiopamd_kkt_lowrank hiopamd_kkt_lowrank;
int hiopamd_kkt_lowrank_create(hiopamd_kkt_lowrank** out, hiopamd_ctx* ctx, hiopamd_hess_lowrank* H);
int hiopamd_kkt_lowrank_destroy(hiopamd_kkt_lowrank* K);
/* update (:1057-1096) from the iterate's dual/slack vectors and bound patterns */
int hiopamd_kkt_lowrank_update(hiopamd_kkt_lowrank* K, const double* zl, const double* sxl, const double* ixl,
                               const double* zu, const double* sxu, const double* ixu, const double* vl,
                               const double* sdl, const double* idl, const double* vu, const double* sdu,
                               const double* idu, const double* Jc, const double* Jd);
/* same with pre-computed diagonals Dx (n_local) and Dd = vl/sdl + vu/sdu (m_ineq) */
int hiopamd_kkt_lowrank_update_diag(hiopamd_kkt_lowrank* K, const double* Dx, const double* Dd, const double* Jc,
                                    const double* Jd);
/* solveCompressed (:1110-1187); rx is modified like in the reference (:1178); *ok_host = 0 if N was not SPD */
int hiopamd_kkt_lowrank_solve_compressed(hiopamd_kkt_lowrank* K, double* rx, const double* ryc, const double* ryd,
                                         double* dx, double* dyc, double* dyd, int* ok_host);
/* Jacobians without touching the barrier diagonals (borrowed if Jd follows Jc in memory, copied into one block otherwise) */
int hiopamd_kkt_lowrank_set_jacobians(hiopamd_kkt_lowrank* K, const double* Jc, const double* Jd);
double* hiopamd_kkt_lowrank_Dd_inv(hiopamd_kkt_lowrank* K);   /* device, m_ineq */
double* hiopamd_kkt_lowrank_J(hiopamd_kkt_lowrank* K);        /* device, (m_eq+m_ineq) x n_local: [Jc; Jd] of the last update */
hiopamd_hess_lowrank* hiopamd_kkt_lowrank_hess(hiopamd_kkt_lowrank* K);
int hiopamd_kkt_lowrank_dims(const hiopamd_kkt_lowrank* K, int64_t* n_local_host, int* m_eq_host, int* m_ineq_host);
double* hiopamd_kkt_lowrank_N(hiopamd_kkt_lowrank* K);   /* device, k x k, the last reduced matrix */
double hiopamd_kkt_lowrank_last_residual(const hiopamd_kkt_lowrank* K);
/* N = J (H+Dx)^-1 J^T + Dd^-1 and its factor are kept between the solveCompressed calls of one outer iteration (the
 * reference rebuilds them on every call, hiopKKTLinSys.cpp:1132-1135); any update / update_diag / set_jacobians call or
 * a change of the Hessian invalidates them.  The Jacobians are borrowed: a caller that overwrites their VALUES in place
 * (same pointers) between two solves must announce it with one of those calls — update() does, which is the reference's
 * calling sequence (hiopAlgFilterIPM.cpp:1212-1213: update before every solve of a new iterate).  enable = 0 restores the
 * rebuild-on-every-call behaviour (same results, bit for bit; for A/B tests). */
int hiopamd_kkt_lowrank_set_cache(hiopamd_kkt_lowrank* K, int enable);

/* =====================================================================================
 * Full-space XYcYd layer: iterate/residual -> search direction
 * (reference: src/Optimization/hiopKKTLinSys.cpp — hiopKKTLinSysCompressedXYcYd::update :543,
 *  hiopKKTLinSysCurvCheck::factorize :316, ::computeDirections :585, compute_directions_for_full_space :218,
 *  compute_directions_w_IR :911, hiopMatVecKKTFullOpr::times_vec :1619, hiopPrecondKKTOpr::times_vec :1900;
 *  src/LinAlg/hiopKrylovSolver.cpp:390-700 hiopBiCGStabSolver::solve;
 *  src/Optimization/hiopPDPerturbation.cpp:161-395 hiopPDPerturbationPrimalFirstScalar;
 *  src/Optimization/hiopFactAcceptor.cpp:63-104 hiopFactAcceptorIC;
 *  src/Optimization/hiopKKTLinSysDense.hpp:84-212 hiopKKTLinSysDenseXYcYd).
 * An iterate, a direction and a residual are each ONE contiguous device slab with the 12 parts in the order of
 * hiopVectorCompoundPD (src/LinAlg/hiopVectorCompoundPD.cpp:99-210):
 *   iterate/direction: x d yc yd sxl sxu sdl sdu zl zu vl vu     residual: rx rd ryc ryd rxl rxu rdl rdu rszl rszu rsvl rsvu
 * with sizes nx nd nyc nyd nx nx nd nd nx nx nd nd (nd = nyd = #inequalities); hiopamd_kkt_xycyd_offsets returns
 * the 13 prefix offsets.  ixl/ixu/idl/idu are the 0/1 bound patterns (device, borrowed for the object's lifetime).
 * ===================================================================================== */
typedef struct hiopamd_kkt_xycyd hiopamd_kkt_xycyd;
/* on top of the condensed MDS system (values given to the MDS object by hiopamd_kkt_mds_set_values; its Dx/Dd
 * arguments may be NULL there: update() below supplies them) */
int hiopamd_kkt_xycyd_create_mds(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, hiopamd_kkt_mds* k, const double* ixl,
                                 const double* ixu, const double* idl, const double* idu);
/* hiopKKTLinSysDenseXYcYd: the whole (nx+neq+nineq)^2 system as one dense matrix; owns its linear solver */
int hiopamd_kkt_xycyd_create_dense(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, int nx, int neq, int nineq,
                                   const double* ixl, const double* ixu, const double* idl, const double* idu);
/* hiopKKTLinSysDenseXDYcYd (hiopKKTLinSysDense.hpp:229-380): the XDYcYd form, N = nx + neq + 2 nineq, unknowns
 * ordered [x | d | yc | yd]; computeDirections is hiopKKTLinSysCompressedXDYcYd's (hiopKKTLinSys.cpp:810-905) */
int hiopamd_kkt_xycyd_create_dense_xdycyd(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, int nx, int neq, int nineq,
                                          const double* ixl, const double* ixu, const double* idl, const double* idu);
/* on top of hiopKKTLinSysLowRank (column-sharded: x-sized parts are the rank's slice, the rest is replicated);
 * perturbations are hiopPDPerturbationNull like in hiopAlgFilterIPMQuasiNewton */
/* hiopKKTLinSysCondensedSparse behind the full-space layer (XDYcYd order of the compressed system, no equalities) */
int hiopamd_kkt_xycyd_create_sparse_condensed(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, hiopamd_kkt_sparse_condensed* k,
                                              const double* ixl, const double* ixu, const double* idl, const double* idu);
int hiopamd_kkt_xycyd_create_lowrank(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, hiopamd_kkt_lowrank* K,
                                     const double* ixl, const double* ixu, const double* idl, const double* idu);
int hiopamd_kkt_xycyd_destroy(hiopamd_kkt_xycyd* h);
int64_t hiopamd_kkt_xycyd_dim(const hiopamd_kkt_xycyd* h);                       /* 5nx + 5nd + nyc + nyd */
int hiopamd_kkt_xycyd_offsets(const hiopamd_kkt_xycyd* h, int64_t* off13_host);
/* dense backend: H (nx x nx, full symmetric), Jc (neq x nx), Jd (nineq x nx); low-rank backend: H = NULL,
 * Jc, Jd (k x n_local) — the arguments of update(iter, grad_f, Jac_c, Jac_d, Hess) that are matrices (:543) */
int hiopamd_kkt_xycyd_set_matrices(hiopamd_kkt_xycyd* h, const double* H, const double* Jc, const double* Jd);
int hiopamd_kkt_xycyd_set_mu(hiopamd_kkt_xycyd* h, double mu);                   /* hiopPDPerturbation::set_mu */
/* opts8 = delta_w_min_bar, delta_w_max_bar, delta_0_bar, kappa_w_minus, kappa_w_plus_bar, kappa_w_plus,
 * delta_c_bar, kappa_c (defaults of src/Utils/hiopOptions.cpp:1080-1123 are built in) */
int hiopamd_kkt_xycyd_set_perturbation_options(hiopamd_kkt_xycyd* h, const double* opts8_host);
/* Which hiopPDPerturbation the inertia-correction loop runs (hiopAlgFilterIPM.cpp:2165-2177: options
 * `normaleqn_regularization_priority` and `regularization_method`):  dual_first = 0 -> ...PrimalFirst*, 1 -> ...DualFirst*;
 * randomized = 0 -> ...Scalar (delta * I), 1 -> ...Rand: the regularisation VECTORS are uniform in [0.9, 1.0] x delta
 * (hiopPDPerturbation.hpp:53-54, .cpp:414-455, :670-711), drawn on the device from (seed, draw counter, index) and consumed by
 * every backend's build (hiopamd_kkt_mds_build_vec, ..._sparse_condensed_build_vec, the dense builds), by the 12-block operator
 * of the iterative refinement and by test_direction.  HIOPAMD_ERR_STATE on the quasi-Newton backend (hiopPDPerturbationNull). */
int hiopamd_kkt_xycyd_set_regularization(hiopamd_kkt_xycyd* h, int dual_first, int randomized, uint64_t seed);
/* device pointers of the current delta_wx [nx], delta_wd [nineq], delta_cc [neq], delta_cd [nineq] (randomized mode only) */
int hiopamd_kkt_xycyd_delta_vectors(hiopamd_kkt_xycyd* h, const double** delta_wx, const double** delta_wd,
                                    const double** delta_cc, const double** delta_cd);
int hiopamd_kkt_xycyd_set_required_neg_eig(hiopamd_kkt_xycyd* h, int n_required); /* default neq+nineq (hiopAlgFilterIPM.cpp:2096) */
/* the same choice as hiopamd_krylov_set_exit_mode for the BiCGStab of compute_directions_w_IR (default: the reference's behaviour) */
int hiopamd_kkt_xycyd_set_bicgstab_exit_mode(hiopamd_kkt_xycyd* h, int reference);
/* update (:543): barrier diagonals from the iterate, then factorize (:316): build + factor + inertia-correction
 * loop (<= 10 re-factorizations).  *ok_host = the reference's bool return. */
int hiopamd_kkt_xycyd_update(hiopamd_kkt_xycyd* h, const double* iter, int* ok_host);
int hiopamd_kkt_xycyd_factorize(hiopamd_kkt_xycyd* h, int* ok_host);
/* 0 = hiopFactAcceptorIC (inertia correction, default), 1 = hiopFactAcceptorInertiaFreeDWD (hiopFactAcceptor.cpp:106) */
int hiopamd_kkt_xycyd_set_fact_acceptor(hiopamd_kkt_xycyd* h, int kind);
/* hiopKKTLinSysCurvCheck::factorize_inertia_free (:376-448): force one more primal regularisation step, rebuild,
 * re-factor (and keep regularising while the factorisation reports singular) */
int hiopamd_kkt_xycyd_factorize_inertia_free(hiopamd_kkt_xycyd* h, int* ok_host);
/* hiopKKTLinSysCompressed::test_direction (:455-513): *accept_host = 0 if the direction has negative curvature,
 * i.e. dx'(H+Dx+delta_wx)dx + dd'(Dd+delta_wd)dd < neg_curv_test_fact*(|dx|^2+|dd|^2) (option default 1e-11) */
int hiopamd_kkt_xycyd_test_direction(hiopamd_kkt_xycyd* h, const double* dir, double neg_curv_test_fact,
                                     int* accept_host, double* dWd_host, double* xs_nrmsq_host);
int hiopamd_kkt_xycyd_deltas(const hiopamd_kkt_xycyd* h, double* deltas4_host);  /* delta_wx, wd, cc, cd in use */
int hiopamd_kkt_xycyd_num_refactorizations(const hiopamd_kkt_xycyd* h);
/* computeDirections (:585) + compute_directions_for_full_space (:218): resid -> dir (distinct slabs) */
int hiopamd_kkt_xycyd_compute_directions(hiopamd_kkt_xycyd* h, const double* resid, double* dir, int* ok_host);
/* y = KKT_full * x (:1619), including the current perturbations */
int hiopamd_kkt_xycyd_times_vec(hiopamd_kkt_xycyd* h, double* y, const double* x);
/* compute_directions_w_IR (:911): BiCGStab, left-preconditioned by compute_directions, x0 = 0,
 * tol = min(mu*ir_outer_tol_factor, ir_outer_tol_min); info4_host = flag, iter, abs_resid, rel_resid
 * (hiopKrylovSolver.hpp flag codes: 0 converged, 1 max iter, 3 stagnation, 4 breakdown) */
int hiopamd_kkt_xycyd_compute_directions_w_IR(hiopamd_kkt_xycyd* h, const double* resid, double* dir,
                                              double ir_outer_tol_factor, double ir_outer_tol_min, int ir_outer_maxit,
                                              int* ok_host, int* converged_host, double* info4_host);
hiopamd_linsolver* hiopamd_kkt_xycyd_linsolver(hiopamd_kkt_xycyd* h);            /* dense backend only */
double* hiopamd_kkt_xycyd_Dx(hiopamd_kkt_xycyd* h);                              /* device, nx */
double* hiopamd_kkt_xycyd_Dd(hiopamd_kkt_xycyd* h);                              /* device, nd */

/* ---- the steps either side of the KKT solve, on the same 12-part slabs ------------------------------
 * bounds / right-hand sides of the problem (device, borrowed): xl, xu (nx), dl, du (nd), crhs (nyc) */
int hiopamd_kkt_xycyd_set_bounds(hiopamd_kkt_xycyd* h, const double* xl, const double* xu, const double* dl,
                                 const double* du, const double* crhs);
/* hiopResidual::update (src/Optimization/hiopResidual.cpp:154-365): all 12 residual parts from the iterate, the
 * constraint bodies c(x), d(x), grad_f and the back-end's Jacobians; kappa_d > 0 adds the linear damping terms
 * (hiopLogBarProblem.hpp:135-145).  norms11_host = nrmInf_nlp_optim, nrmInf_nlp_feasib, nrmInf_nlp_complem,
 * nrmInf_bar_optim, nrmInf_bar_feasib, nrmInf_bar_complem, nrmOne_nlp_feasib, nrmOne_bar_feasib, nrmOne_nlp_optim,
 * nrmOne_bar_optim, nrmInf_cons_violation */
int hiopamd_residual_update(hiopamd_kkt_xycyd* h, const double* iter, const double* c, const double* d,
                            const double* grad_f, double mu, double kappa_d, double* resid, double* norms11_host);
/* hiopIterate (src/Optimization/hiopIterate.cpp): fractionToTheBdry :330, takeStep_primals/_duals :367-390,
 * determineSlacks :274, adjust_small_slacks :414-505, determineDualsBounds_d :314, adjustDuals_primalLogHessian :507,
 * evalLogBarrier :523, linearDampingTerm :552 */
int hiopamd_iterate_fraction_to_the_bdry(hiopamd_kkt_xycyd* h, const double* iter, const double* dir, double tau,
                                         double* alpha_primal_host, double* alpha_dual_host);
int hiopamd_iterate_take_step(hiopamd_kkt_xycyd* h, double* out, const double* iter, const double* dir,
                              double alpha_primal, double alpha_dual, int primals, int duals);
int hiopamd_iterate_determine_slacks(hiopamd_kkt_xycyd* h, double* iter);
int hiopamd_iterate_adjust_small_slacks(hiopamd_kkt_xycyd* h, double* iter, const double* iter_curr, double mu,
                                        int* num_adjusted_host);
/* hiopNlpFormulation::adjust_bounds (src/Optimization/hiopNlpFormulation.cpp:1403-1416), what the algorithm calls after
 * adjust_small_slacks moved a slack: the bounds follow the slacks — xl = x - sxl, xu = x + sxu, dl = d - sdl, du = d + sdu on the
 * respective patterns (other entries untouched).  The four arrays are the ones given to hiopamd_kkt_xycyd_set_bounds (device). */
int hiopamd_iterate_adjust_bounds(hiopamd_kkt_xycyd* h, const double* iter, double* xl, double* xu, double* dl, double* du);
int hiopamd_iterate_determine_duals_bounds_d(hiopamd_kkt_xycyd* h, double* iter, double mu);
int hiopamd_iterate_adjust_duals_plh(hiopamd_kkt_xycyd* h, double* iter, double mu, double kappa_Sigma);
int hiopamd_iterate_eval_log_barrier(hiopamd_kkt_xycyd* h, const double* iter, double* out_host);
int hiopamd_iterate_linear_damping_term(hiopamd_kkt_xycyd* h, const double* iter, double mu, double kappa_d,
                                        double* out_host);
/* hiopDualsLsqUpdateLinsysRedDense::do_lsq_update (src/Optimization/hiopDualsUpdater.cpp:239-330): least-squares
 * estimate of the constraint duals; overwrites the yc, yd parts of `iter`; *ok_host = 0 if the m x m system is not SPD */
int hiopamd_duals_lsq_update(hiopamd_kkt_xycyd* h, double* iter, const double* grad_f, int* ok_host);

/* =====================================================================================
 * `.iajaaa` linear-system dumps (reference: src/Utils/hiopCSR_IO.hpp:44-152, src/LinAlg/csr_iajaaa.md) — the
 * `write_kkt yes` debugging path; byte-compatible with upstream tooling (load_kkt_mat.m).  `path` is a host string;
 * the matrix is the row-major upper triangle BEFORE factorisation (write it between build and factorize).
 * ===================================================================================== */
int hiopamd_io_write_iajaaa_matrix(hiopamd_ctx* ctx, const char* path, int m, const double* M_dev, int64_t ld, int nx,
                                   int meq, int mineq);
int hiopamd_io_append_iajaaa_vector(hiopamd_ctx* ctx, const char* path, int m, const double* v_dev);

/* Iteration table of the interior-point loop, character for character what hiopAlgFilterIPMNewton::outputIteration /
 * hiopAlgFilterIPMQuasiNewton::outputIteration print (src/Optimization/hiopAlgFilterIPM.cpp:2783-2812, :1521-1549), so
 * that logs can be diffed against upstream runs (the reference's CPU-vs-GPU iteration-table comparison).
 * Host-only (no device work).  Both write a NUL-terminated line INCLUDING the trailing newline into `buf` and return its
 * length, or a negative status if `buflen` is too small.
 *   header: printed by the reference when iter % 10 == 0;
 *   ls_status: -1 -> "-(-)"; 1/2/3 -> s/h/f (upper case when use_soc != 0); anything else -> "?"; use_fr != 0 -> "R"
 *   (and, in the Newton variant only, the line-search count is printed as 0);  obj = f / obj_scale, lg(mu) = log10(mu). */
int hiopamd_io_iteration_header(char* buf, int buflen);
int hiopamd_io_format_iteration(char* buf, int buflen, int quasi_newton, int iter, double obj, double inf_pr, double inf_du,
                                double mu, double alpha_du, double alpha_pr, int ls_status, int ls_num, int use_soc,
                                int use_fr);

/* =====================================================================================
 * Krylov solvers (reference: src/LinAlg/hiopKrylovSolver.hpp:80-258; hiopPCGSolver::solve hiopKrylovSolver.cpp:152-373,
 * hiopBiCGStabSolver::solve :397-700).  The reference's hiopLinearOperator::times_vec(y, x) becomes a callback on device
 * pointers: return 0 on success; `y` never aliases `x`.  kind: 0 = PCG, 1 = BiCGStab.  Defaults tol = 1e-9,
 * maxit = 8 (:81-82).  `solve` overwrites the right-hand side with the solution (the minimal-residual iterate if the
 * method did not converge) and reports convergence in *converged_host.  The start vector is the solver's own buffer:
 * zero at creation, left at the last iterate by every solve, reset by set_x0 — as in the reference (xk_ = x0_).
 * Flags: 0 converged, 1 maximum number of iterations, 3 stagnation / tolerance too small, 4 breakdown of a scalar.
 * BiCGStab counts half iterations (get_sol_num_iter returns k - 0.5 when it stops after the first half step).
 * ===================================================================================== */
typedef struct hiopamd_krylov hiopamd_krylov;
typedef int (*hiopamd_linop_fn)(void* user, const double* x_dev, double* y_dev);
int hiopamd_krylov_create(hiopamd_krylov** out, hiopamd_ctx* ctx, int kind, int64_t n, hiopamd_linop_fn A, void* A_user,
                          hiopamd_linop_fn Mleft, void* Mleft_user, hiopamd_linop_fn Mright, void* Mright_user);
int hiopamd_krylov_destroy(hiopamd_krylov* k);
int hiopamd_krylov_set_tol(hiopamd_krylov* k, double tol);                 /* hiopKrylovSolver.hpp:105 */
int hiopamd_krylov_set_max_num_iter(hiopamd_krylov* k, int maxit);         /* :98 */
int hiopamd_krylov_set_x0(hiopamd_krylov* k, double xval);                 /* :95 */
double* hiopamd_krylov_x0(hiopamd_krylov* k);                              /* device pointer of the start vector */
int hiopamd_krylov_solve(hiopamd_krylov* k, double* b_inout, int* converged_host);
int hiopamd_krylov_get_convergence_flag(const hiopamd_krylov* k);          /* :125 */
/* BiCGStab's 'tol is too small' exit (100 extra steps without reaching the tolerance).  reference != 0 (default): the reference's
 * behaviour — it copies the current iterate over the right-hand side BEFORE the closing comparison of the minimal-residual iterate
 * (hiopKrylovSolver.cpp:561-566, :639-644, :671-688), which therefore runs against the overwritten vector and in practice returns the
 * last iterate.  0: the comparison against the original right-hand side (what the code's comment intends).  PCG is not affected. */
int hiopamd_krylov_set_exit_mode(hiopamd_krylov* k, int reference);
double hiopamd_krylov_get_sol_num_iter(const hiopamd_krylov* k);           /* :116 */
double hiopamd_krylov_get_sol_abs_resid(const hiopamd_krylov* k);          /* :110 */
double hiopamd_krylov_get_sol_rel_resid(const hiopamd_krylov* k);          /* :113 */

/* =====================================================================================
 * Device-resident user callbacks of the reference's MDS example.
 * reference: class MdsEx1, src/Drivers/MDS/NlpMdsEx1.hpp:54-560 (RAJA twin: NlpMdsRajaEx1.cpp) — the callbacks of
 * hiopInterfaceMDS (src/Interface/hiopInterface.hpp:582-780; C FFI src/Interface/hiopInterface.h:63-98) with every array
 * argument a DEVICE pointer.  Index arrays are int (hiop_index_type).  Any output pointer of the Jacobian / Hessian calls may
 * be NULL (the reference is called once for the sparsity pattern and once per iteration for the values).  The equalities
 * and the inequalities are evaluated by separate Jacobian calls because that is how the solver calls eval_Jac_cons
 * (num_cons = ns with idx_cons = 0..ns-1, then num_cons = 3): the outputs are exactly the arrays
 * hiopamd_kkt_mds_set_values() takes.  eval_f synchronises (it returns the objective to the host); nothing else does.
 * ===================================================================================== */
typedef struct hiopamd_mdsex1 hiopamd_mdsex1;
int hiopamd_mdsex1_create(hiopamd_mdsex1** out, hiopamd_ctx* ctx, int ns, int nd, int empty_sp_row);       /* :64-97 */
int hiopamd_mdsex1_destroy(hiopamd_mdsex1* p);
int hiopamd_mdsex1_get_prob_sizes(const hiopamd_mdsex1* p, int64_t* n, int64_t* m);                         /* :108-113 */
int hiopamd_mdsex1_get_vars_info(hiopamd_mdsex1* p, double* xlow_dev, double* xupp_dev);                    /* :115-141 */
int hiopamd_mdsex1_get_cons_info(hiopamd_mdsex1* p, double* clow_dev, double* cupp_dev);                    /* :143-163 */
int hiopamd_mdsex1_get_sparse_dense_blocks_info(const hiopamd_mdsex1* p, int* nx_sparse, int* nx_dense, int* nnz_sparse_Jaceq,
                                                int* nnz_sparse_Jacineq, int* nnz_sparse_Hess_Lagr_SS,
                                                int* nnz_sparse_Hess_Lagr_SD);                              /* :165-184 */
int hiopamd_mdsex1_get_starting_point(hiopamd_mdsex1* p, double* x0_dev);                                   /* :442-447 */
int hiopamd_mdsex1_eval_f(hiopamd_mdsex1* p, const double* x_dev, double* obj_host);                        /* :186-209 */
int hiopamd_mdsex1_eval_grad_f(hiopamd_mdsex1* p, const double* x_dev, double* gradf_dev);                  /* :269-289 */
int hiopamd_mdsex1_eval_cons(hiopamd_mdsex1* p, const double* x_dev, double* cons_dev);                     /* :211-266, all ns+3 */
int hiopamd_mdsex1_eval_Jac_cons_eq(hiopamd_mdsex1* p, const double* x_dev, int* iJacS_dev, int* jJacS_dev, double* MJacS_dev,
                                    double* JacD_dev);                                                      /* :291-400, rows 0..ns-1 */
/* row_offset: 0 for the solver's inequality call (rows 0..2); ns for the one-call form of the interface
 * (hiopInterface.hpp:691-704, MdsEx1OneCallCons in NlpMdsRajaEx1.cpp:894-1011), where the arrays are the tails of the
 * arrays the equality call filled */
int hiopamd_mdsex1_eval_Jac_cons_ineq(hiopamd_mdsex1* p, const double* x_dev, int row_offset, int* iJacS_dev, int* jJacS_dev,
                                      double* MJacS_dev, double* JacD_dev);                                 /* :291-400, the 3 inequalities */
int hiopamd_mdsex1_eval_Hess_Lagr(hiopamd_mdsex1* p, const double* x_dev, double obj_factor, const double* lambda_dev,
                                  int* iHSS_dev, int* jHSS_dev, double* MHSS_dev, double* HDD_dev);         /* :403-440 */

/* DenseConsEx2, the example of the memory-distributed dense-constraints path (reference: src/Drivers/Dense/NlpDenseConsEx2.cpp,
 * callbacks of hiopInterfaceDenseConstraints, src/Interface/hiopInterface.hpp:420-560).  x, gradf, bounds and the rows of
 * the Jacobian are the LOCAL columns [cols[rank], cols[rank+1]) of this rank (partition of the context's communicator, the
 * example's own quotient/remainder rule); the objective and the four constraint bodies are summed over the ranks through the
 * context's all-reduce hook, as the example does with MPI_Allreduce.  Jac is 4 x n_local row-major. */
typedef struct hiopamd_denseex2 hiopamd_denseex2;
int hiopamd_denseex2_create(hiopamd_denseex2** out, hiopamd_ctx* ctx, int64_t n_global, int unconstrained);   /* .cpp:7-39 */
int hiopamd_denseex2_destroy(hiopamd_denseex2* p);
int hiopamd_denseex2_get_prob_sizes(const hiopamd_denseex2* p, int64_t* n, int64_t* m);                        /* .cpp:45-50 */
int hiopamd_denseex2_get_vecdistrib_info(const hiopamd_denseex2* p, int64_t* cols_host);                       /* .cpp:293-304 */
int hiopamd_denseex2_get_vars_info(hiopamd_denseex2* p, double* xlow_dev, double* xupp_dev);                   /* .cpp:52-80 */
int hiopamd_denseex2_get_cons_info(const hiopamd_denseex2* p, double* clow_host, double* cupp_host);           /* .cpp:82-102 */
int hiopamd_denseex2_get_starting_point(hiopamd_denseex2* p, double* x0_dev);                                  /* .cpp:307-315 */
int hiopamd_denseex2_eval_f(hiopamd_denseex2* p, const double* x_dev, double* obj_host);                       /* .cpp:104-117 */
int hiopamd_denseex2_eval_grad_f(hiopamd_denseex2* p, const double* x_dev, double* gradf_dev);                 /* .cpp:119-126 */
int hiopamd_denseex2_eval_cons(hiopamd_denseex2* p, const double* x_dev, double* cons_dev);                    /* .cpp:129-221 */
int hiopamd_denseex2_eval_Jac_cons(hiopamd_denseex2* p, const double* x_dev, double* Jac_dev);                 /* .cpp:224-290 */

/* =====================================================================================
 * hiopPDPerturbation on its own (src/Optimization/hiopPDPerturbation.hpp:10-212): the scalar state machines of the
 * inertia-correction loop, host only (no device work, no context).  kind: 0 = hiopPDPerturbationPrimalFirstScalar
 * (.cpp:108-395), 1 = hiopPDPerturbationDualFirstScalar (:457-626), 2 = hiopPDPerturbationNull.  *ok = the bool the
 * reference method returns.  get: curr4 / last4 = delta_wx, delta_wd, delta_cc, delta_cd (current / last);
 * state4 = hess_degenerate, jac_degenerate (0 not established, 1 not degenerate, 2 degenerate), deltas_test_type (0..4 in the
 * order of hiopPDPerturbation.hpp:168-175), and the bit mask of vector groups the last calls marked for update
 * (1 primal, 2 dual — set_delta_curr_vec's task id).
 * ===================================================================================== */
typedef struct hiopamd_pd_perturbation hiopamd_pd_perturbation;
int hiopamd_pd_perturbation_create(hiopamd_pd_perturbation** out, int kind);
int hiopamd_pd_perturbation_destroy(hiopamd_pd_perturbation* p);
int hiopamd_pd_perturbation_set_options(hiopamd_pd_perturbation* p, const double* opts8_host);   /* as hiopamd_kkt_xycyd_set_perturbation_options */
int hiopamd_pd_perturbation_set_mu(hiopamd_pd_perturbation* p, double mu);
int hiopamd_pd_perturbation_compute_initial_deltas(hiopamd_pd_perturbation* p, int* ok);
int hiopamd_pd_perturbation_compute_perturb_wrong_inertia(hiopamd_pd_perturbation* p, int* ok);
int hiopamd_pd_perturbation_compute_perturb_singularity(hiopamd_pd_perturbation* p, int* ok);
int hiopamd_pd_perturbation_get(const hiopamd_pd_perturbation* p, double* curr4, double* last4, int* state4);

#ifdef __cplusplus
}
#endif
#endif /* HIOP_AMD_H */
