// hiopKKTLinSysCondensedSparse on MI355X: the condensed sparse KKT of the inequality-only sparse formulation
//   (H + Dx + delta_wx I + Jd^T (Dd + delta_wd I) Jd) dx = rx + Jd^T ((Dd + delta_wd I) ryd + rd)
//   dd  = Jd dx - ryd
//   dyd = (Dd + delta_wd I) dd - rd
// reference: src/Optimization/hiopKKTLinSysSparseCondensed.cpp:105-335 (build_kkt_matrix), :346-401 (solve_compressed_direct),
// :403-449 (solveCompressed).  The reference hands the condensed matrix M to a sparse Cholesky (MA57 on the CPU, cuSOLVER on
// the GPU, :469-496) — neither is in the image, so that solver cannot be pinned and is not restated.  Here the inner solve is
// preconditioned conjugate gradients on the device CSR matrix (hiopPCGSolver, src/LinAlg/hiopKrylovSolver.cpp:152-373,
// already part of this library) with the Jacobi preconditioner: M is symmetric positive definite whenever the reference's
// Cholesky exists, and PCG answers the same question Cholesky answers for the IPM — "is M positive definite along the
// directions that matter?" — by its curvature test (p^T M p <= 0 stops the iteration with a failure flag, which the
// inertia-correction loop treats like the reference treats a failed factorisation).  The accuracy of the inner solve is
// then lifted by the parent class's BiCGStab refinement on the full KKT (hiopKKTLinSys::compute_directions_w_IR), as in the
// reference ("Code for iterative refinement ... was removed since the parent KKT class performs this now", :432-435).
#include "device_utils.hpp"

#include <cstdlib>
#include <limits>
#include <vector>

struct hiopamd_kkt_sparse_condensed {
  hiopamd_ctx* ctx = nullptr;
  int nx = 0, nineq = 0, nnzJ = 0, nnzH = 0;
  hiopamd_csr_condensed* csr = nullptr;
  hiopamd_krylov* pcg = nullptr;
  // DIRECT inner solver for moderate orders (nx <= HIOPAMD_SPARSE_DIRECT_MAX, default 4096): the condensed matrix expanded to a dense
  // upper triangle and factored by this library's LDL^T with inertia — the role of the reference's sparse Cholesky (MA57 /
  // cuSOLVER, hiopKKTLinSysSparseCondensed.cpp:469-496) with the same answer to "is M positive definite" (exact, not through a
  // Krylov probe).  Beyond that order: PCG + Jacobi.
  hiopamd_linsolver* dls = nullptr;
  bool dls_factored = false;   // the dense copy holds factors of the CURRENT M
  // SPARSE direct inner solver for bordered-diagonal patterns of any order (csrc/arrow_ldl.hip: the arrowhead of the reference's
  // sparse examples): exact LDL^T with exact inertia.  HIOPAMD_SPARSE_ARROW=0 switches it off (the Krylov path stays testable).
  hiopamd_arrow_ldl* arrow = nullptr;
  // GENERAL sparse direct inner solver (csrc/sparse_ldl.hip: nested dissection, multifrontal by tree levels, dense root) for the patterns
  // the two above do not take: exact LDL^T, exact inertia.  Patterns whose dense root would exceed its limit keep PCG + Jacobi.
  hiopamd_sparse_ldl* sldl = nullptr;
  // sparsity (device copies of the triplet index arrays: the SpMVs of the right-hand side / recovery and of the operator)
  int *iJ = nullptr, *jJ = nullptr, *iH = nullptr, *jH = nullptr;
  // current values (borrowed)
  const double *J_val = nullptr, *H_val = nullptr, *Dx = nullptr, *Dd = nullptr;
  // owned
  double* Hd = nullptr;        // nineq: Dd + delta_wd
  double* Dxp = nullptr;       // nx: Dx + delta_wx
  double* rhs = nullptr;       // nx
  bool built = false;
  double tol = 1e-12;
  int maxit = 2000;
  int last_flag = 0;
  double last_iters = 0.0, last_rel = 0.0;
};

using namespace hiopamd;

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

namespace {
struct SpDelta {   // a perturbation: one value or a device vector (the reference's form), see csrc/kkt_mds.hip
  const double* v;
  double s;
  __device__ __forceinline__ double at(int64_t i) const { return v ? v[i] : s; }
};

int up_int(int** d, const int* h, int n)
{
  *d = nullptr;
  if(hipMalloc((void**)d, sizeof(int) * (size_t)(n > 0 ? n : 1)) != hipSuccess) return HIOPAMD_ERR_HIP;
  if(n > 0 && hipMemcpy(*d, h, sizeof(int) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return HIOPAMD_ERR_HIP;
  return HIOPAMD_OK;
}

// dense upper triangle of M (the direct inner solver's system matrix) from its CSR form
int build_impl_dense_copy(hiopamd_kkt_sparse_condensed* k)
{
  hiopamd_ctx* ctx = k->ctx;
  double* Md = hiopamd_linsolver_sys_matrix(k->dls);
  const int64_t n = k->nx;
  HIOPAMD_CHECK(hipMemsetAsync(Md, 0, sizeof(double) * (size_t)n * (size_t)n, ctx->stream));
  const int* rp = hiopamd_csr_condensed_rowptr(k->csr);
  const int* ci = hiopamd_csr_condensed_colidx(k->csr);
  const double* v = hiopamd_csr_condensed_values(k->csr);
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
    for(int q = rp[i]; q < rp[i + 1]; ++q)
      if(ci[q] >= i) Md[i * n + ci[q]] = v[q];
  }));
  k->dls_factored = false;
  return HIOPAMD_OK;
}

int build_impl(hiopamd_kkt_sparse_condensed* k, SpDelta dwx, SpDelta dwd)
{
  if(!k || !k->J_val || !k->Dx || (k->nineq > 0 && !k->Dd) || (k->nnzH > 0 && !k->H_val)) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = k->ctx;
  SpanScope span(ctx, HIOPAMD_SPAN_KKT_UPDATE_LINSYS);
  {   // Hd = Dd + delta_wd (:150-151), Dx + delta_wx (:196)
    const double *Dd = k->Dd, *Dx = k->Dx;
    double *Hd = k->Hd, *Dxp = k->Dxp;
    const int64_t nx = k->nx, nd = k->nineq, nmax = nx > nd ? nx : nd;
    RC(launch_ew(ctx, nmax, [=] __device__(int64_t i) {
      if(i < nd) Hd[i] = Dd[i] + dwd.at(i);
      if(i < nx) Dxp[i] = Dx[i] + dwx.at(i);
    }));
  }
  // M = Jd^T diag(Hd) Jd + H + diag(Dx + delta_wx)   (:205-318; one symbolic analysis at create, four launches here)
  RC(hiopamd_csr_condensed_numeric(k->csr, k->J_val, k->H_val, k->Hd, k->Dxp, 0.0));
  if(k->dls) RC(build_impl_dense_copy(k));
  k->built = true;
  return HIOPAMD_OK;
}
}  // namespace

extern "C" {

int hiopamd_kkt_sparse_condensed_destroy(hiopamd_kkt_sparse_condensed* k)
{
  if(!k) return HIOPAMD_OK;
  if(k->ctx) (void)hipStreamSynchronize(k->ctx->stream);
  if(k->pcg) hiopamd_krylov_destroy(k->pcg);
  if(k->dls) hiopamd_linsolver_destroy(k->dls);
  if(k->arrow) hiopamd_arrow_ldl_destroy(k->arrow);
  if(k->sldl) hiopamd_sparse_ldl_destroy(k->sldl);
  if(k->csr) hiopamd_csr_condensed_destroy(k->csr);
  (void)hipFree(k->iJ); (void)hipFree(k->jJ); (void)hipFree(k->iH); (void)hipFree(k->jH);
  (void)hipFree(k->Hd); (void)hipFree(k->Dxp); (void)hipFree(k->rhs);
  delete k;
  return HIOPAMD_OK;
}

// Jd: nineq x nx triplets (row-sorted), H: upper-triangle triplets of the nx x nx Hessian of the Lagrangian; HOST index arrays
// (the symbolic phase runs once per sparsity pattern, like the reference's *_symbolic calls at the first build, :213-300)
int hiopamd_kkt_sparse_condensed_create(hiopamd_kkt_sparse_condensed** out, hiopamd_ctx* ctx, int nx, int nineq, int nnzJ,
                                        const int* iJ_host, const int* jJ_host, int nnzH, const int* iH_host, const int* jH_host)
{
  if(!out || !ctx || nx < 0 || nineq < 0 || nnzJ < 0 || nnzH < 0) return HIOPAMD_ERR_ARG;
  *out = nullptr;
  auto* k = new hiopamd_kkt_sparse_condensed();
  k->ctx = ctx; k->nx = nx; k->nineq = nineq; k->nnzJ = nnzJ; k->nnzH = nnzH;
  int rc = hiopamd_csr_condensed_create(&k->csr, ctx, nx, nineq, nnzJ, iJ_host, jJ_host, nnzH, iH_host, jH_host);
  if(rc == HIOPAMD_OK) rc = up_int(&k->iJ, iJ_host, nnzJ);
  if(rc == HIOPAMD_OK) rc = up_int(&k->jJ, jJ_host, nnzJ);
  if(rc == HIOPAMD_OK) rc = up_int(&k->iH, iH_host, nnzH);
  if(rc == HIOPAMD_OK) rc = up_int(&k->jH, jH_host, nnzH);
  auto dalloc = [](double** p, size_t n) { return hipMalloc((void**)p, sizeof(double) * (n ? n : 1)) == hipSuccess ? HIOPAMD_OK : HIOPAMD_ERR_HIP; };
  if(rc == HIOPAMD_OK) rc = dalloc(&k->Hd, nineq);
  if(rc == HIOPAMD_OK) rc = dalloc(&k->Dxp, nx);
  if(rc == HIOPAMD_OK) rc = dalloc(&k->rhs, nx);
  // the inner "linear solver": PCG on the CSR operator with its Jacobi preconditioner (kind 0 = PCG)
  if(rc == HIOPAMD_OK)
    rc = hiopamd_krylov_create(&k->pcg, ctx, 0, nx, hiopamd_csr_condensed_apply, k->csr, hiopamd_csr_condensed_jacobi, k->csr, nullptr,
                               nullptr);
  {
    const char* e = std::getenv("HIOPAMD_SPARSE_DIRECT_MAX");
    const int direct_max = e ? std::atoi(e) : 4096;
    if(rc == HIOPAMD_OK && nx > 0 && nx <= direct_max) rc = hiopamd_linsolver_create(&k->dls, ctx, nx);
    if(rc == HIOPAMD_OK && k->dls) rc = hiopamd_linsolver_set_retry_copy(k->dls, 0);   // (the dense copy is rebuilt after a time-out, see factorize)
  }
  if(rc == HIOPAMD_OK && !k->dls && nx > 0 && !(std::getenv("HIOPAMD_SPARSE_ARROW") && std::atoi(std::getenv("HIOPAMD_SPARSE_ARROW")) == 0)) {
    // the pattern of M decides: bordered diagonal (<= 32 border variables) -> sparse direct solver, else the Krylov inner solver
    std::vector<int> rp((size_t)nx + 1), ci((size_t)hiopamd_csr_condensed_nnz(k->csr));
    rc = hiopamd_csr_condensed_pattern(k->csr, rp.data(), ci.data());
    if(rc == HIOPAMD_OK) {
      const int ra = hiopamd_arrow_ldl_create(&k->arrow, ctx, nx, rp.data(), ci.data());
      if(ra != HIOPAMD_OK && ra != HIOPAMD_ERR_STATE) rc = ra;
      if(rc == HIOPAMD_OK && !k->arrow) {   // not a bordered diagonal with a small border: the general sparse LDL^T, if its root fits
        const int rs = hiopamd_sparse_ldl_create(&k->sldl, ctx, nx, rp.data(), ci.data());
        if(rs != HIOPAMD_OK && rs != HIOPAMD_ERR_STATE) rc = rs;
        if(rs == HIOPAMD_ERR_STATE)   // said once per object: the caller should know that factorize()'s verdict is now a Krylov probe's
          std::fprintf(stderr, "[hiop_amd] condensed sparse KKT (n = %d): the sparse LDL^T does not take this pattern (its dense root would exceed the limit — a mesh-like "
                               "pattern — or a diagonal entry is structurally zero); inner solver = PCG + Jacobi: factorize() then answers from a Krylov probe (necessary, not sufficient, for positive definiteness), "
                               "hiopamd_kkt_sparse_condensed_inner_kind() = 2\n", nx);
      }
    }
  }
  if(rc == HIOPAMD_OK) rc = hiopamd_krylov_set_tol(k->pcg, k->tol);
  if(rc == HIOPAMD_OK) rc = hiopamd_krylov_set_max_num_iter(k->pcg, k->maxit);
  if(rc != HIOPAMD_OK) {
    hiopamd_kkt_sparse_condensed_destroy(k);
    return rc;
  }
  *out = k;
  return HIOPAMD_OK;
}

// values of the current iterate (device pointers, borrowed until the next call): Jd triplet values, Hessian triplet values,
// Dx (nx), Dd (nineq, WITHOUT delta_wd)
int hiopamd_kkt_sparse_condensed_set_values(hiopamd_kkt_sparse_condensed* k, const double* Jd_val, const double* H_val,
                                            const double* Dx, const double* Dd)
{
  if(!k) return HIOPAMD_ERR_ARG;
  k->J_val = Jd_val; k->H_val = H_val; k->Dx = Dx; k->Dd = Dd;
  k->built = false;
  // Jd^T (CSR) gets the new values now: the transposed product is asked for before the next build (hiopResidual::update)
  return Jd_val ? hiopamd_csr_condensed_refresh_jt(k->csr, Jd_val) : HIOPAMD_OK;
}

// only the log-barrier diagonals change (hiopKKTLinSysCompressedXDYcYd::update)
int hiopamd_kkt_sparse_condensed_set_diagonals(hiopamd_kkt_sparse_condensed* k, const double* Dx, const double* Dd)
{
  if(!k) return HIOPAMD_ERR_ARG;
  k->Dx = Dx; k->Dd = Dd;
  k->built = false;
  return HIOPAMD_OK;
}

int hiopamd_kkt_sparse_condensed_build(hiopamd_kkt_sparse_condensed* k, double delta_wx, double delta_wd)
{
  return build_impl(k, SpDelta{nullptr, delta_wx}, SpDelta{nullptr, delta_wd});
}
int hiopamd_kkt_sparse_condensed_build_vec(hiopamd_kkt_sparse_condensed* k, const double* delta_wx, const double* delta_wd)
{
  return build_impl(k, SpDelta{delta_wx, 0.0}, SpDelta{delta_wd, 0.0});
}

// What matrixChanged() of the reference's Cholesky solver tells the IPM: 0 negative eigenvalues when the factorisation exists,
// -1 otherwise.  An iterative inner solver has no factorisation; the cheap necessary condition is checked here (every diagonal
// entry of M positive and finite), the sufficient one by the curvature test of every solve.
int hiopamd_kkt_sparse_condensed_factorize(hiopamd_kkt_sparse_condensed* k, int* n_neg_host)
{
  if(!k || !n_neg_host) return HIOPAMD_ERR_ARG;
  if(!k->built) return HIOPAMD_ERR_STATE;
  SpanScope span(k->ctx, HIOPAMD_SPAN_KKT_UPDATE_INNER_FACT);
  if(k->dls) {   // a Cholesky exists iff no pivot of the LDL^T is non-positive
    int nneg = 0;
    if(k->dls_factored) RC(build_impl_dense_copy(k));   // factorize twice on one build: the copy holds factors, not M
    int rc = hiopamd_linsolver_matrix_changed(k->dls, &nneg);
    if(rc == HIOPAMD_ERR_TIMEOUT) {   // (the dataflow factorisation gave up, DESIGN.md 3.1: the copy again, then the stepwise kernels)
      RC(build_impl_dense_copy(k));
      rc = hiopamd_linsolver_matrix_changed(k->dls, &nneg);
    }
    k->dls_factored = true;
    if(rc != HIOPAMD_OK && rc != HIOPAMD_ERR_SINGULAR) return rc;
    *n_neg_host = (rc == HIOPAMD_ERR_SINGULAR || nneg != 0) ? -1 : 0;
    return HIOPAMD_OK;
  }
  if(k->arrow) {   // exact: M = L diag(D, S) L^T exists with positive pivots iff M is positive definite
    int nneg = 0, nzero = 0;
    RC(hiopamd_arrow_ldl_factorize(k->arrow, hiopamd_csr_condensed_values(k->csr), &nneg, &nzero));
    *n_neg_host = (nneg != 0 || nzero != 0) ? -1 : 0;
    return HIOPAMD_OK;
  }
  if(k->sldl) {   // exact as well: P M P^T = L D L^T with positive pivots iff M is positive definite
    int nneg = 0, nzero = 0;
    RC(hiopamd_sparse_ldl_factorize(k->sldl, hiopamd_csr_condensed_values(k->csr), &nneg, &nzero));
    *n_neg_host = (nneg != 0 || nzero != 0) ? -1 : 0;
    return HIOPAMD_OK;
  }
  double* diag = k->rhs;
  RC(hiopamd_csr_condensed_diagonal(k->csr, diag));
  int64_t nonpos = 0;
  int finite = 1;
  ReduceNow now(k->ctx);   // the values are used right below: not to be parked in a caller's reduction bracket
  RC(hiopamd_vec_num_elems_less_than(k->ctx, k->nx, diag, std::numeric_limits<double>::min(), &nonpos));
  RC(hiopamd_vec_isfinite(k->ctx, k->nx, diag, &finite));
  *n_neg_host = (nonpos > 0 || !finite) ? -1 : 0;
  return HIOPAMD_OK;
}

// solveCompressed (:403) = solve_compressed_direct (:346-401).  rx (nx), rd, ryd (nineq) inputs; dx (nx), dd, dyd (nineq) outputs;
// *ok_host = 0 when the inner solve failed (not positive definite along a search direction, or no convergence)
int hiopamd_kkt_sparse_condensed_solve_compressed(hiopamd_kkt_sparse_condensed* k, const double* rx, const double* rd,
                                                  const double* ryd, double* dx, double* dd, double* dyd, int* ok_host)
{
  if(!k || !ok_host) return HIOPAMD_ERR_ARG;
  if(!k->built) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = k->ctx;
  const int nx = k->nx, nd = k->nineq;
  *ok_host = 0;
  span_begin(ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);
  {   // rhs = rx ; dyd = Hd .* ryd + rd  (work buffer in the output, like the reference, :370-377)
    const double* Hd = k->Hd;
    double* rhs = k->rhs;
    const int64_t nmax = nx > nd ? nx : nd;
    int rc = launch_ew(ctx, nmax, [=] __device__(int64_t i) {
      if(i < nx) rhs[i] = rx[i];
      if(i < nd) dyd[i] = Hd[i] * ryd[i] + rd[i];
    });
    if(rc == HIOPAMD_OK) rc = hiopamd_csr_condensed_jac_trans_times_vec(k->csr, 1.0, k->rhs, 1.0, dyd);   // :379
    span_end(ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);
    if(rc != HIOPAMD_OK) return rc;
  }
  int conv = 0;
  if(k->dls) {
    SpanScope span(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);   // linSys_->solve(*rhs_)  (:384)
    RC(hiopamd_linsolver_solve(k->dls, k->rhs, 1));
    conv = 1;
    k->last_flag = 0;
    k->last_iters = 0.0;
    k->last_rel = 0.0;
  } else if(k->arrow) {
    SpanScope span(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);
    const int ra = hiopamd_arrow_ldl_solve(k->arrow, k->rhs);
    if(ra != HIOPAMD_OK && ra != HIOPAMD_ERR_STATE) return ra;
    conv = ra == HIOPAMD_OK ? 1 : 0;   // (ERR_STATE: the last factorisation found M singular / was never run: the reference returns false)
    k->last_flag = conv ? 0 : 4;
    k->last_iters = 0.0;
    k->last_rel = 0.0;
  } else if(k->sldl) {
    SpanScope span(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);
    const int rs = hiopamd_sparse_ldl_solve(k->sldl, k->rhs);
    if(rs != HIOPAMD_OK && rs != HIOPAMD_ERR_STATE) return rs;
    conv = rs == HIOPAMD_OK ? 1 : 0;   // (ERR_STATE: the last factorisation met a zero pivot / was never run: the reference returns false)
    k->last_flag = conv ? 0 : 4;
    k->last_iters = 0.0;
    k->last_rel = 0.0;
  } else {
    SpanScope span(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);
    RC(hiopamd_krylov_set_x0(k->pcg, 0.0));
    RC(hiopamd_krylov_solve(k->pcg, k->rhs, &conv));
    k->last_flag = hiopamd_krylov_get_convergence_flag(k->pcg);
    k->last_iters = hiopamd_krylov_get_sol_num_iter(k->pcg);
    k->last_rel = hiopamd_krylov_get_sol_rel_resid(k->pcg);
  }
  if(!conv) return HIOPAMD_OK;   // (*ok_host = 0: the reference returns false here, :386-388)
  SpanScope span(ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);
  {   // dx = rhs ; dd = ryd  (:389-391)
    const double* rhs = k->rhs;
    const int64_t nmax = nx > nd ? nx : nd;
    RC(launch_ew(ctx, nmax, [=] __device__(int64_t i) {
      if(i < nx) dx[i] = rhs[i];
      if(i < nd) dd[i] = ryd[i];
    }));
  }
  {   // dd = Jd dx - ryd  (:392)
    int rj = hiopamd_csr_condensed_jac_times_vec(k->csr, k->jJ, k->J_val, -1.0, dd, 1.0, dx);
    if(rj == HIOPAMD_ERR_STATE) rj = hiopamd_sp_times_vec(ctx, nd, nx, k->nnzJ, k->iJ, k->jJ, k->J_val, -1.0, dd, 1.0, dx);
    RC(rj);
  }
  {   // dyd = Hd .* dd - rd  (:394-396)
    const double* Hd = k->Hd;
    RC(launch_ew(ctx, nd, [=] __device__(int64_t i) { dyd[i] = Hd[i] * dd[i] - rd[i]; }));
  }
  *ok_host = 1;
  return HIOPAMD_OK;
}

int hiopamd_kkt_sparse_condensed_set_inner_solver(hiopamd_kkt_sparse_condensed* k, double tol, int max_iter)
{
  if(!k || !(tol > 0.0) || max_iter < 1) return HIOPAMD_ERR_ARG;
  k->tol = tol;
  k->maxit = max_iter;
  RC(hiopamd_krylov_set_tol(k->pcg, tol));
  RC(hiopamd_krylov_set_max_num_iter(k->pcg, max_iter));
  return HIOPAMD_OK;
}
// convergence flag (hiopKrylovSolver.hpp:125), iterations and relative residual of the last inner solve
int hiopamd_kkt_sparse_condensed_last_solve(const hiopamd_kkt_sparse_condensed* k, int* flag_host, double* iters_host, double* rel_resid_host)
{
  if(!k) return HIOPAMD_ERR_ARG;
  if(flag_host) *flag_host = k->last_flag;
  if(iters_host) *iters_host = k->last_iters;
  if(rel_resid_host) *rel_resid_host = k->last_rel;
  return HIOPAMD_OK;
}
int hiopamd_kkt_sparse_condensed_dims(const hiopamd_kkt_sparse_condensed* k, int* dims4_host /* nx, nineq, nnzJ, nnzH */)
{
  if(!k || !dims4_host) return HIOPAMD_ERR_ARG;
  dims4_host[0] = k->nx; dims4_host[1] = k->nineq; dims4_host[2] = k->nnzJ; dims4_host[3] = k->nnzH;
  return HIOPAMD_OK;
}
// which inner solver the object runs: 0 dense LDL^T of the expanded matrix, 1 bordered-diagonal direct solver, 2 PCG + Jacobi,
// 3 general sparse LDL^T (nested dissection + multifrontal + dense root)
int hiopamd_kkt_sparse_condensed_ldl_info(const hiopamd_kkt_sparse_condensed* k, int64_t* info8_host)
{
  if(!k || !info8_host) return HIOPAMD_ERR_ARG;
  if(!k->sldl) return HIOPAMD_ERR_STATE;
  return hiopamd_sparse_ldl_info(k->sldl, info8_host);
}

int hiopamd_kkt_sparse_condensed_inner_kind(const hiopamd_kkt_sparse_condensed* k)
{
  if(!k) return HIOPAMD_ERR_ARG;
  return k->dls ? 0 : (k->arrow ? 1 : (k->sldl ? 3 : 2));
}
hiopamd_csr_condensed* hiopamd_kkt_sparse_condensed_matrix(hiopamd_kkt_sparse_condensed* k) { return k ? k->csr : nullptr; }
double* hiopamd_kkt_sparse_condensed_Hd(hiopamd_kkt_sparse_condensed* k) { return k ? k->Hd : nullptr; }

// the sparse matrices' products on the values of the last set_values (for the full-space layer: hiopMatVecKKTFullOpr needs
// Hess x, Jd x and Jd^T y; hiopMatrixSymSparseTriplet::timesVec hiopMatrixSparseTriplet.cpp:952, ::timesVec :73, ::transTimesVec :110)
int hiopamd_kkt_sparse_condensed_hess_times_vec(hiopamd_kkt_sparse_condensed* k, double beta, double* y, double alpha, const double* x)
{
  if(!k || !k->J_val) return HIOPAMD_ERR_STATE;
  return hiopamd_spsym_times_vec(k->ctx, k->nx, k->nnzH, k->iH, k->jH, k->H_val, beta, y, alpha, x);
}
int hiopamd_kkt_sparse_condensed_jac_times_vec(hiopamd_kkt_sparse_condensed* k, double beta, double* y, double alpha, const double* x)
{
  if(!k || !k->J_val) return HIOPAMD_ERR_STATE;
  const int rj = hiopamd_csr_condensed_jac_times_vec(k->csr, k->jJ, k->J_val, beta, y, alpha, x);
  if(rj != HIOPAMD_ERR_STATE) return rj;
  return hiopamd_sp_times_vec(k->ctx, k->nineq, k->nx, k->nnzJ, k->iJ, k->jJ, k->J_val, beta, y, alpha, x);
}
int hiopamd_kkt_sparse_condensed_jac_trans_times_vec(hiopamd_kkt_sparse_condensed* k, double beta, double* y, double alpha, const double* x)
{
  if(!k || !k->J_val) return HIOPAMD_ERR_STATE;
  if(!k->J_val && k->nnzJ) return HIOPAMD_ERR_STATE;
  return hiopamd_csr_condensed_jac_trans_times_vec(k->csr, beta, y, alpha, x);   // (Jd^T in CSR, values of the last set_values)
}

}  // extern "C"
