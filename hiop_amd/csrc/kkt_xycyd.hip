// Full-space layer of the KKT hot path on MI355X: everything HiOp does per IPM iteration between "iterate +
// residual" and "search direction", on top of a condensed XYcYd solver (MDS, dense or quasi-Newton low-rank):
//
//   update            hiopKKTLinSysCompressedXYcYd::update            src/Optimization/hiopKKTLinSys.cpp:543-583
//   factorize         hiopKKTLinSysCurvCheck::factorize               :316-376   (inertia-correction loop)
//                     hiopPDPerturbationPrimalFirstScalar             hiopPDPerturbation.cpp:161-395
//                     hiopFactAcceptorIC::requireReFactorization      hiopFactAcceptor.cpp:63-104
//   compute_directions hiopKKTLinSysCompressedXYcYd::computeDirections :585-690
//                     hiopKKTLinSys::compute_directions_for_full_space :218-314
//   times_vec         hiopMatVecKKTFullOpr::times_vec                 :1619-1736
//   compute_directions_w_IR  hiopKKTLinSys::compute_directions_w_IR   :911-961
//                     hiopBiCGStabSolver::solve                       src/LinAlg/hiopKrylovSolver.cpp:390-700
//                     hiopPrecondKKTOpr::times_vec                    hiopKKTLinSys.cpp:1900-1909
//   dense backend     hiopKKTLinSysDenseXYcYd                         hiopKKTLinSysDense.hpp:84-212
//
// Data layout: an iterate / direction / residual is ONE contiguous fp64 slab in HBM holding the 12 parts in the
// order of hiopVectorCompoundPD (hiopVectorCompoundPD.cpp:99-210)
//     [ x | d | yc | yd | sxl | sxu | sdl | sdu | zl | zu | vl | vu ]        (residual: rx rd ryc ryd rxl ... rsvu)
// so that every compound-vector operation of the Krylov loop is a single launch, and the rhs reduction /
// direction recovery / 12-block operator are one fused element-wise kernel each instead of ~40 BLAS-1 calls.
#include "device_utils.hpp"
#include "pd_perturb.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

using namespace hiopamd;

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

namespace {

// hiopFactAcceptorIC::requireReFactorization (hiopFactAcceptor.cpp:63-104)
int require_refactorization(PdPerturb& pd, int n_required_neg_eig, int n_neg_eig)
{
  if(n_required_neg_eig > 0) {
    if(n_neg_eig < 0) return pd.compute_perturb_singularity() ? 1 : -1;
    if(n_neg_eig != n_required_neg_eig) return pd.compute_perturb_wrong_inertia() ? 1 : -1;
    return 0;
  }
  if(n_neg_eig != 0) return pd.compute_perturb_wrong_inertia() ? 1 : -1;
  return 0;
}

// hiopFactAcceptorInertiaFreeDWD::requireReFactorization (hiopFactAcceptor.cpp:106-155)
int require_refactorization_inertia_free(PdPerturb& pd, int n_required_neg_eig, int n_neg_eig, bool force_reg)
{
  if(n_required_neg_eig > 0) {
    if(n_neg_eig < 0) return pd.compute_perturb_singularity() ? 1 : -1;
    if(!force_reg) return 0;
    return pd.compute_perturb_wrong_inertia() ? 1 : -1;
  }
  if(n_neg_eig < 0) return pd.compute_perturb_wrong_inertia() ? 1 : -1;
  if(!force_reg) return 0;
  return pd.compute_perturb_wrong_inertia() ? 1 : -1;
}

enum Kind { KIND_MDS = 1, KIND_DENSE = 2, KIND_LOWRANK = 3, KIND_DENSE_XDYCYD = 4, KIND_SPARSE_CONDENSED = 5 };

// two sums in one pass: the column-partitioned parts of the slab (x-sized, all-reduced) and the replicated ones
struct dot2_t {
  double dist, repl;
};
struct OpSlabDot2 {
  const double *a, *b;
  int64_t o1, o4, o6, o8, o10;   // dist = [0,o1) u [o4,o6) u [o8,o10)
  __device__ dot2_t identity() const { return dot2_t{0.0, 0.0}; }
  __device__ dot2_t map(int64_t i) const
  {
    const double p = a[i] * b[i];
    const bool dist = i < o1 || (i >= o4 && i < o6) || (i >= o8 && i < o10);
    return dist ? dot2_t{p, 0.0} : dot2_t{0.0, p};
  }
  __device__ dot2_t combine(dot2_t p, dot2_t q) const { return dot2_t{p.dist + q.dist, p.repl + q.repl}; }
};

__device__ inline double sel(double pattern, double v) { return pattern == 0.0 ? 0.0 : v; }

}  // namespace

struct hiopamd_kkt_xycyd {
  hiopamd_ctx* ctx = nullptr;
  int kind = 0;
  int64_t nx = 0;
  int nd = 0, nyc = 0, nyd = 0;
  int64_t off[13] = {0};
  int64_t dim = 0;
  const double *ixl = nullptr, *ixu = nullptr, *idl = nullptr, *idu = nullptr;   // borrowed, device
  const double* iter = nullptr;                                                   // borrowed, device slab
  const double *xl = nullptr, *xu = nullptr, *dl = nullptr, *du = nullptr, *crhs = nullptr;   // borrowed (set_bounds)
  hiopamd_kkt_mds* mds = nullptr;
  hiopamd_kkt_lowrank* lr = nullptr;
  hiopamd_kkt_sparse_condensed* sc = nullptr;   // hiopKKTLinSysCondensedSparse (an XDYcYd class without equalities)
  // dense backend (hiopKKTLinSysDenseXYcYd) and the Jacobians of the low-rank backend
  hiopamd_linsolver* ls = nullptr;
  const double *H = nullptr, *Jc = nullptr, *Jd = nullptr;
  double *dense_rhs = nullptr, *dense_Dd_inv = nullptr;
  // owned
  double *Dx = nullptr, *Dd = nullptr, *rx_tilde = nullptr, *ryd_tilde = nullptr, *ryd2 = nullptr;
  double* krylov = nullptr;   // 9 slabs, allocated at the first IR call
  double* dsmall = nullptr;   // 4 doubles of device scratch for the sharded dot
  double* lsq = nullptr;      // LSQ dual update workspace: M (m^2) | rhs (m) | posv work (3 m^2 + 8 m)
  PdPerturb pd;
  // randomized regularisation (hiopPDPerturbation*Rand): the four vectors the builders / the 12-block operator consume
  double* dvec[4] = {nullptr, nullptr, nullptr, nullptr};   // delta_wx [nx], delta_wd [nd], delta_cc [nyc], delta_cd [nyd]
  uint64_t reg_seed = 0x9E3779B97F4A7C15ull, reg_draw = 0;
  int n_required_neg = 0;
  int bicg_ref_exit = 1;   // BiCGStab 'tol is too small' exit: 1 = the reference's (closing comparison against the overwritten right-hand side, hiopKrylovSolver.cpp:563,641), 0 = against the original one
  int num_refact = 0;
  int acceptor = 0;   // 0: hiopFactAcceptorIC, 1: hiopFactAcceptorInertiaFreeDWD
  bool is_xd() const { return kind == KIND_DENSE_XDYCYD || kind == KIND_SPARSE_CONDENSED; }
};

namespace {

const double* Dd_inv_of(hiopamd_kkt_xycyd* h)
{
  switch(h->kind) {
    case KIND_MDS: return hiopamd_kkt_mds_Dd_inv(h->mds);
    case KIND_LOWRANK: return hiopamd_kkt_lowrank_Dd_inv(h->lr);
    default: return h->dense_Dd_inv;
  }
}

// uniform in [lo, hi): counter-based (splitmix64 of seed, draw number, index) so that a draw is reproducible from
// (seed, draw) alone; hiopVector::set_to_random_uniform's role (hiopVectorPar.cpp:134, host std generator there)
int fill_uniform(hiopamd_ctx* ctx, int64_t n, double* out, double lo, double hi, uint64_t seed, uint64_t draw)
{
  return launch_ew(ctx, n, [=] __device__(int64_t i) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (draw * 0x100000001B3ull + (uint64_t)i + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);   // [0, 1)
    out[i] = lo + (hi - lo) * u;
  });
}

// set_delta_curr_vec of the *Rand classes (hiopPDPerturbation.cpp:433-455, :689-711): redraw the groups the scalar machine marked
int refresh_delta_vectors(hiopamd_kkt_xycyd* h)
{
  PdPerturb& pd = h->pd;
  if(!pd.randomized || pd.null_mode) return HIOPAMD_OK;
  const double lo = pd.min_uniform_ratio, hi = pd.max_uniform_ratio;
  const int64_t len[4] = {h->nx, h->nd, h->nyc, h->nyd};
  const double sc[4] = {pd.wx, pd.wd, pd.cc, pd.cd};
  for(int v = 0; v < 4; ++v) {
    const int group = v < 2 ? PdPerturb::PrimalUpdate : PdPerturb::DualUpdate;
    if(!(pd.dirty & group)) continue;
    RC(fill_uniform(h->ctx, len[v], h->dvec[v], lo * sc[v], hi * sc[v], h->reg_seed, h->reg_draw++));
  }
  pd.dirty = 0;
  return HIOPAMD_OK;
}

// ---- backend: (re)build the condensed matrix for the current deltas -----------------------------------
int backend_build(hiopamd_kkt_xycyd* h)
{
  const PdPerturb& pd = h->pd;
  hiopamd_ctx* ctx = h->ctx;
  const bool rnd = pd.randomized && !pd.null_mode;
  if(rnd) RC(refresh_delta_vectors(h));
  double* const* dv = h->dvec;
  if(h->kind == KIND_MDS)
    return rnd ? hiopamd_kkt_mds_build_vec(h->mds, dv[0], dv[1], dv[2], dv[3]) : hiopamd_kkt_mds_build(h->mds, pd.wx, pd.wd, pd.cc, pd.cd);
  if(h->kind == KIND_LOWRANK) return HIOPAMD_OK;   // N is formed inside solveCompressed (hiopKKTLinSys.cpp:1132)
  if(h->kind == KIND_SPARSE_CONDENSED)
    return rnd ? hiopamd_kkt_sparse_condensed_build_vec(h->sc, dv[0], dv[1]) : hiopamd_kkt_sparse_condensed_build(h->sc, pd.wx, pd.wd);
  SpanScope span(ctx, HIOPAMD_SPAN_KKT_UPDATE_LINSYS);
  if(h->kind == KIND_DENSE_XDYCYD) {
    // hiopKKTLinSysDenseXDYcYd::build_kkt_matrix (hiopKKTLinSysDense.hpp:249-328)
    if(!h->H || (!h->Jc && h->nyc > 0) || (!h->Jd && h->nyd > 0)) return HIOPAMD_ERR_STATE;
    const int nx = (int)h->nx, neq = h->nyc, nineq = h->nyd, n = nx + neq + 2 * nineq;
    double* M = hiopamd_linsolver_sys_matrix(h->ls);
    HIOPAMD_CHECK(hipMemsetAsync(M, 0, sizeof(double) * (size_t)n * (size_t)n, ctx->stream));       // :283
    RC(hiopamd_mat_add_upper_to_sym_upper(ctx, nx, h->H, nx, 0, 1.0, M, n));                         // :286
    RC(hiopamd_mat_trans_add_to_sym_upper(ctx, neq, nx, h->Jc, nx, 0, nx + nineq, 1.0, M, n));       // :288
    RC(hiopamd_mat_trans_add_to_sym_upper(ctx, nineq, nx, h->Jd, nx, 0, nx + nineq + neq, 1.0, M, n));   // :289
    RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, 0, 1.0, h->Dx, 0, nx));                               // :292
    if(rnd) RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, 0, 1.0, dv[0], 0, nx));                       // :293 (delta_wx_ is a vector)
    else RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, n, 0, nx, pd.wx));
    RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, nx, 1.0, h->Dd, 0, nineq));                           // :295
    if(rnd) RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, nx, 1.0, dv[1], 0, nineq));                   // :296
    else RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, n, nx, nineq, pd.wd));
    {
      const int64_t ld = n;
      const int c0 = nx + nineq + neq;
      RC(launch_ew(ctx, nineq, [=] __device__(int64_t i) { M[(nx + i) * ld + c0 + i] -= 1.0; }));      // :299-307
    }
    // :312 is literally addSubDiagonal(-1, nx+nineq, delta_cd): nineq entries starting at diagonal nx+nineq
    if(rnd) RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, nx + nineq, -1.0, dv[3], 0, nineq));
    else RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, n, nx + nineq, nineq, -pd.cd));
    return HIOPAMD_OK;
  }
  // hiopKKTLinSysDenseXYcYd::build_kkt_matrix (hiopKKTLinSysDense.hpp:84-170)
  if(!h->H || (!h->Jc && h->nyc > 0) || (!h->Jd && h->nyd > 0)) return HIOPAMD_ERR_STATE;
  const int nx = (int)h->nx, neq = h->nyc, nineq = h->nyd, n = nx + neq + nineq;
  double* M = hiopamd_linsolver_sys_matrix(h->ls);
  HIOPAMD_CHECK(hipMemsetAsync(M, 0, sizeof(double) * (size_t)n * (size_t)n, ctx->stream));     // :134
  RC(hiopamd_mat_add_upper_to_sym_upper(ctx, nx, h->H, nx, 0, 1.0, M, n));                       // :137
  RC(hiopamd_mat_trans_add_to_sym_upper(ctx, neq, nx, h->Jc, nx, 0, nx, 1.0, M, n));             // :139
  RC(hiopamd_mat_trans_add_to_sym_upper(ctx, nineq, nx, h->Jd, nx, 0, nx + neq, 1.0, M, n));     // :140
  RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, 0, 1.0, h->Dx, 0, nx));                             // :142
  if(rnd) RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, 0, 1.0, dv[0], 0, nx));                     // :143 (delta_wx_ is a vector)
  else RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, n, 0, nx, pd.wx));
  {
    const double* Dd = h->Dd;
    double* Ddi = h->dense_Dd_inv;
    const double dwd = pd.wd;
    const double* dwdv = rnd ? dv[1] : nullptr;
    RC(launch_ew(ctx, nineq, [=] __device__(int64_t i) { Ddi[i] = 1.0 / ((dwdv ? dwdv[i] : dwd) + Dd[i]); }));   // :146-152
  }
  RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, nx + neq, -1.0, h->dense_Dd_inv, 0, nineq));        // :155
  // :160 is literally addSubDiagonal(-1, nx, delta_cd): nineq entries starting at diagonal position nx
  if(rnd) RC(hiopamd_mat_add_sub_diagonal(ctx, M, n, nx, -1.0, dv[3], 0, nineq));
  else RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, n, nx, nineq, -pd.cd));
  return HIOPAMD_OK;
}

int backend_factorize(hiopamd_kkt_xycyd* h, int* n_neg)
{
  if(h->kind == KIND_MDS) return hiopamd_kkt_mds_factorize(h->mds, n_neg);
  if(h->kind == KIND_LOWRANK) {
    *n_neg = h->n_required_neg;
    return HIOPAMD_OK;
  }
  if(h->kind == KIND_SPARSE_CONDENSED) {
    // the condensed matrix has to be positive definite (Cholesky in the reference); the count the acceptor compares with is
    // the full system's number of constraints: "0 negative eigenvalues of M" <=> "n_required of the full KKT" (Sylvester)
    int n0 = 0;
    RC(hiopamd_kkt_sparse_condensed_factorize(h->sc, &n0));
    *n_neg = (n0 < 0) ? -1 : h->n_required_neg;
    return HIOPAMD_OK;
  }
  SpanScope span(h->ctx, HIOPAMD_SPAN_KKT_UPDATE_INNER_FACT);   // hiopKKTLinSys.cpp:347-352
  int rc = hiopamd_linsolver_matrix_changed(h->ls, n_neg);   // hiopKKTLinSys.cpp:310-313
  if(rc == HIOPAMD_ERR_TIMEOUT) {
    // the dataflow factorisation gave up and left the matrix overwritten: assemble it again for the same deltas (the
    // randomised delta vectors are not redrawn: nothing is marked dirty) and factor once more — that call runs the stepwise kernels
    RC(backend_build(h));
    rc = hiopamd_linsolver_matrix_changed(h->ls, n_neg);
  }
  return rc;
}

// solveCompressed: rx and ryd may be overwritten (the reference's classes do the same)
int backend_solve(hiopamd_kkt_xycyd* h, double* rx, const double* ryc, double* ryd, double* dx, double* dyc,
                  double* dyd, int* ok)
{
  *ok = 1;
  hiopamd_ctx* ctx = h->ctx;
  if(h->kind == KIND_MDS) return hiopamd_kkt_mds_solve_compressed(h->mds, rx, ryc, ryd, dx, dyc, dyd);
  if(h->kind == KIND_LOWRANK) return hiopamd_kkt_lowrank_solve_compressed(h->lr, rx, ryc, ryd, dx, dyc, dyd, ok);
  // hiopKKTLinSysDenseXYcYd::solveCompressed (hiopKKTLinSysDense.hpp:172-212)
  const int nx = (int)h->nx, nyc = h->nyc, nyd = h->nyd;
  RC(hiopamd_vec_copy(ctx, nx, h->dense_rhs, rx));
  RC(hiopamd_vec_copy(ctx, nyc, h->dense_rhs + nx, ryc));
  RC(hiopamd_vec_copy(ctx, nyd, h->dense_rhs + nx + nyc, ryd));
  {
    SpanScope span(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);
    RC(hiopamd_linsolver_solve(h->ls, h->dense_rhs, 1));
  }
  RC(hiopamd_vec_copy(ctx, nx, dx, h->dense_rhs));
  RC(hiopamd_vec_copy(ctx, nyc, dyc, h->dense_rhs + nx));
  RC(hiopamd_vec_copy(ctx, nyd, dyd, h->dense_rhs + nx + nyc));
  return HIOPAMD_OK;
}

// hiopKKTLinSysDenseXDYcYd::solveCompressed (hiopKKTLinSysDense.hpp:330-362): system order [x | d | yc | yd]
int backend_solve_xd(hiopamd_kkt_xycyd* h, const double* rx, const double* rd, const double* ryc, const double* ryd,
                     double* dx, double* dd, double* dyc, double* dyd, int* ok)
{
  *ok = 1;
  hiopamd_ctx* ctx = h->ctx;
  if(h->kind == KIND_SPARSE_CONDENSED)   // (no equalities: ryc / dyc are empty)
    return hiopamd_kkt_sparse_condensed_solve_compressed(h->sc, rx, rd, ryd, dx, dd, dyd, ok);
  const int nx = (int)h->nx, nyc = h->nyc, nyd = h->nyd;
  double* rhs = h->dense_rhs;
  RC(hiopamd_vec_copy(ctx, nx, rhs, rx));
  RC(hiopamd_vec_copy(ctx, nyd, rhs + nx, rd));
  RC(hiopamd_vec_copy(ctx, nyc, rhs + nx + nyd, ryc));
  RC(hiopamd_vec_copy(ctx, nyd, rhs + nx + nyd + nyc, ryd));
  {
    SpanScope span(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);
    RC(hiopamd_linsolver_solve(h->ls, rhs, 1));
  }
  RC(hiopamd_vec_copy(ctx, nx, dx, rhs));
  RC(hiopamd_vec_copy(ctx, nyd, dd, rhs + nx));
  RC(hiopamd_vec_copy(ctx, nyc, dyc, rhs + nx + nyd));
  RC(hiopamd_vec_copy(ctx, nyd, dyd, rhs + nx + nyd + nyc));
  return HIOPAMD_OK;
}

int backend_hess_times_vec(hiopamd_kkt_xycyd* h, double* y, const double* x)   // y = Hess*x
{
  if(h->kind == KIND_MDS) return hiopamd_kkt_mds_hess_times_vec(h->mds, 0.0, y, 1.0, x);
  if(h->kind == KIND_LOWRANK)   // hiopHessianLowRank::timesVec: no log-barrier term (hiopHessianLowRank.cpp:1061)
    return hiopamd_hess_lowrank_times_vec(hiopamd_kkt_lowrank_hess(h->lr), 0.0, y, 1.0, x, 0);
  if(h->kind == KIND_SPARSE_CONDENSED) return hiopamd_kkt_sparse_condensed_hess_times_vec(h->sc, 0.0, y, 1.0, x);
  return hiopamd_mat_times_vec(h->ctx, (int)h->nx, h->nx, h->H, h->nx, 0.0, y, 1.0, x);
}

// yc = Jc*x, yd = Jd*x (yc, yd contiguous: one all-reduce on a column partition)
int backend_jac_times_vec(hiopamd_kkt_xycyd* h, double* ycd, const double* x)
{
  hiopamd_ctx* ctx = h->ctx;
  if(h->kind == KIND_MDS) {
    RC(hiopamd_kkt_mds_jac_times_vec(h->mds, 0, 0.0, ycd, 1.0, x));
    return hiopamd_kkt_mds_jac_times_vec(h->mds, 1, 0.0, ycd + h->nyc, 1.0, x);
  }
  if(h->kind == KIND_LOWRANK) {
    const int k = h->nyc + h->nyd;
    RC(hiopamd_mat_times_vec(ctx, k, h->nx, hiopamd_kkt_lowrank_J(h->lr), h->nx, 0.0, ycd, 1.0, x));
    if(ctx->allreduce && ctx_allreduce(ctx, ycd, (size_t)k, HIOPAMD_SUM) != 0)
      return HIOPAMD_ERR_HIP;
    return HIOPAMD_OK;
  }
  if(h->kind == KIND_SPARSE_CONDENSED) return hiopamd_kkt_sparse_condensed_jac_times_vec(h->sc, 0.0, ycd + h->nyc, 1.0, x);
  RC(hiopamd_mat_times_vec(ctx, h->nyc, h->nx, h->Jc, h->nx, 0.0, ycd, 1.0, x));
  return hiopamd_mat_times_vec(ctx, h->nyd, h->nx, h->Jd, h->nx, 0.0, ycd + h->nyc, 1.0, x);
}

// y += Jc^T*yc + Jd^T*yd
int backend_jac_trans_times_vec_add(hiopamd_kkt_xycyd* h, double* y, const double* yc, const double* yd)
{
  hiopamd_ctx* ctx = h->ctx;
  if(h->kind == KIND_MDS) {
    RC(hiopamd_kkt_mds_jac_trans_times_vec(h->mds, 0, 1.0, y, 1.0, yc));
    return hiopamd_kkt_mds_jac_trans_times_vec(h->mds, 1, 1.0, y, 1.0, yd);
  }
  if(h->kind == KIND_LOWRANK) {   // [Jc; Jd]^T [yc; yd] in one pass (yc, yd contiguous in the slab)
    if(yd != yc + h->nyc) return HIOPAMD_ERR_ARG;
    return hiopamd_mat_trans_times_vec(ctx, h->nyc + h->nyd, h->nx, hiopamd_kkt_lowrank_J(h->lr), h->nx, 1.0, y, 1.0, yc);
  }
  if(h->kind == KIND_SPARSE_CONDENSED) return hiopamd_kkt_sparse_condensed_jac_trans_times_vec(h->sc, 1.0, y, 1.0, yd);
  RC(hiopamd_mat_trans_times_vec(ctx, h->nyc, h->nx, h->Jc, h->nx, 1.0, y, 1.0, yc));
  return hiopamd_mat_trans_times_vec(ctx, h->nyd, h->nx, h->Jd, h->nx, 1.0, y, 1.0, yd);
}

// ---- compound-vector reductions --------------------------------------------------------------------------
int slab_dot(hiopamd_kkt_xycyd* h, const double* a, const double* b, double* out)
{
  hiopamd_ctx* ctx = h->ctx;
  if(!(h->kind == KIND_LOWRANK && ctx->allreduce)) {
    ReduceNow now(ctx);   // (the Krylov loops go on with the value)
    return hiopamd_vec_dot(ctx, h->dim, a, b, out);
  }
  // column partition: x, sxl, sxu, zl, zu are distributed, the rest is replicated
  // (hiopVectorCompoundPD::dotProductWith sums the parts' own dot products)
  dot2_t r{0.0, 0.0};
  RC(launch_reduce<dot2_t>(ctx, h->dim, OpSlabDot2{a, b, h->off[1], h->off[4], h->off[6], h->off[8], h->off[10]}, &r));
  HIOPAMD_CHECK(hipMemcpyAsync(h->dsmall, &r.dist, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  if(ctx_allreduce(ctx, h->dsmall, 1, HIOPAMD_SUM) != 0) return HIOPAMD_ERR_HIP;
  HIOPAMD_CHECK(hipMemcpyAsync(&r.dist, h->dsmall, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  *out = r.dist + r.repl;
  return HIOPAMD_OK;
}

int slab_norm(hiopamd_kkt_xycyd* h, const double* a, double* out)
{
  double d = 0.0;
  RC(slab_dot(h, a, a, &d));
  *out = std::sqrt(d);
  return HIOPAMD_OK;
}

// ---- the fused element-wise stages ---------------------------------------------------------------------
// (1) update: Dx = zl/sxl + zu/sxu, Dd = vl/sdl + vu/sdu on the bound patterns                      (:562-572)
int stage_update_diagonals(hiopamd_kkt_xycyd* h)
{
  const double* it = h->iter;
  const int64_t* o = h->off;
  const int64_t nx = h->nx, nd = h->nd;
  const double *sxl = it + o[4], *sxu = it + o[5], *sdl = it + o[6], *sdu = it + o[7], *zl = it + o[8],
               *zu = it + o[9], *vl = it + o[10], *vu = it + o[11];
  const double *ixl = h->ixl, *ixu = h->ixu, *idl = h->idl, *idu = h->idu;
  double *Dx = h->Dx, *Dd = h->Dd;
  return launch_ew(h->ctx, std::max<int64_t>(nx, nd), [=] __device__(int64_t i) {
    if(i < nx) {
      double v = 0.0;
      if(ixl[i] == 1.0) v += zl[i] / sxl[i];
      if(ixu[i] == 1.0) v += zu[i] / sxu[i];
      Dx[i] = v;
    }
    if(i < nd) {
      double v = 0.0;
      if(idl[i] == 1.0) v += vl[i] / sdl[i];
      if(idu[i] == 1.0) v += vu[i] / sdu[i];
      Dd[i] = v;
    }
  });
}

// (2) reduction of the 12-part residual to the XYcYd right-hand side                                 (:599-640)
int stage_reduce_rhs(hiopamd_kkt_xycyd* h, const double* r)
{
  const double* it = h->iter;
  const int64_t* o = h->off;
  const int64_t nx = h->nx, nd = h->nd;
  const double *sxl = it + o[4], *sxu = it + o[5], *sdl = it + o[6], *sdu = it + o[7], *zl = it + o[8],
               *zu = it + o[9], *vl = it + o[10], *vu = it + o[11];
  const double *rx = r + o[0], *rd = r + o[1], *ryd = r + o[3], *rxl = r + o[4], *rxu = r + o[5], *rdl = r + o[6],
               *rdu = r + o[7], *rszl = r + o[8], *rszu = r + o[9], *rsvl = r + o[10], *rsvu = r + o[11];
  const double *ixl = h->ixl, *ixu = h->ixu, *idl = h->idl, *idu = h->idu;
  const double* Ddi = h->is_xd() ? nullptr : Dd_inv_of(h);
  double *rx_tilde = h->rx_tilde, *ryd_tilde = h->ryd_tilde, *ryd2 = h->ryd2;
  return launch_ew(h->ctx, std::max<int64_t>(nx, nd), [=] __device__(int64_t i) {
    if(i < nx) {
      double v = rx[i];
      if(ixl[i] == 1.0) v += (rszl[i] - zl[i] * rxl[i]) / sxl[i];
      if(ixu[i] == 1.0) v -= (rszu[i] - zu[i] * rxu[i]) / sxu[i];
      rx_tilde[i] = v;
    }
    if(i < nd) {
      double v = rd[i];
      if(idl[i] == 1.0) v += (rsvl[i] - vl[i] * rdl[i]) / sdl[i];
      if(idu[i] == 1.0) v -= (rsvu[i] - vu[i] * rdu[i]) / sdu[i];
      ryd2[i] = v;                                  // = rd_tilde of the XDYcYd form (hiopKKTLinSys.cpp:845-860)
      if(Ddi) ryd_tilde[i] = ryd[i] + v * Ddi[i];
    }
  });
}

// (3) dd and the eight bound-slack / bound-dual directions                              (:664-666, :218-284)
int stage_recover_directions(hiopamd_kkt_xycyd* h, const double* r, double* dir)
{
  const double* it = h->iter;
  const int64_t* o = h->off;
  const int64_t nx = h->nx, nd = h->nd;
  const double *sxl = it + o[4], *sxu = it + o[5], *sdl = it + o[6], *sdu = it + o[7], *zl = it + o[8],
               *zu = it + o[9], *vl = it + o[10], *vu = it + o[11];
  const double *rxl = r + o[4], *rxu = r + o[5], *rdl = r + o[6], *rdu = r + o[7], *rszl = r + o[8],
               *rszu = r + o[9], *rsvl = r + o[10], *rsvu = r + o[11];
  const double *ixl = h->ixl, *ixu = h->ixu, *idl = h->idl, *idu = h->idu;
  const double* Ddi = h->is_xd() ? nullptr : Dd_inv_of(h);   // XDYcYd: dd comes out of the linear solve
  const double* ryd2 = h->ryd2;
  const double *dx = dir + o[0], *dyd = dir + o[3];
  double *dd = dir + o[1], *dsxl = dir + o[4], *dsxu = dir + o[5], *dsdl = dir + o[6], *dsdu = dir + o[7],
         *dzl = dir + o[8], *dzu = dir + o[9], *dvl = dir + o[10], *dvu = dir + o[11];
  return launch_ew(h->ctx, std::max<int64_t>(nx, nd), [=] __device__(int64_t i) {
    if(i < nx) {
      const double x = dx[i];
      const double sl = sel(ixl[i], rxl[i] + x);
      dsxl[i] = sl;
      dzl[i] = ixl[i] == 0.0 ? 0.0 : (rszl[i] - zl[i] * sl) / sxl[i];
      const double su = sel(ixu[i], rxu[i] - x);
      dsxu[i] = su;
      dzu[i] = ixu[i] == 0.0 ? 0.0 : (rszu[i] - zu[i] * su) / sxu[i];
    }
    if(i < nd) {
      const double d = Ddi ? (ryd2[i] + dyd[i]) * Ddi[i] : dd[i];
      dd[i] = d;
      const double sl = sel(idl[i], rdl[i] + d);
      dsdl[i] = sl;
      dvl[i] = idl[i] == 0.0 ? 0.0 : (rsvl[i] - vl[i] * sl) / sdl[i];
      const double su = sel(idu[i], rdu[i] - d);
      dsdu[i] = su;
      dvu[i] = idu[i] == 0.0 ? 0.0 : (rsvu[i] - vu[i] * su) / sdu[i];
    }
  });
}

// (4) all element-wise terms of the 12-block operator; the matrix products are already in y's rx/ryc/ryd parts
int stage_times_vec_ew(hiopamd_kkt_xycyd* h, double* y, const double* x)
{
  const double* it = h->iter;
  const int64_t* o = h->off;
  const int64_t nx = h->nx, nd = h->nd, nyc = h->nyc;
  const double *sxl = it + o[4], *sxu = it + o[5], *sdl = it + o[6], *sdu = it + o[7], *zl = it + o[8],
               *zu = it + o[9], *vl = it + o[10], *vu = it + o[11];
  const double *dx = x + o[0], *dd = x + o[1], *dyc = x + o[2], *dyd = x + o[3], *dsxl = x + o[4], *dsxu = x + o[5],
               *dsdl = x + o[6], *dsdu = x + o[7], *dzl = x + o[8], *dzu = x + o[9], *dvl = x + o[10],
               *dvu = x + o[11];
  double *yrx = y + o[0], *yrd = y + o[1], *yryc = y + o[2], *yryd = y + o[3], *yrxl = y + o[4], *yrxu = y + o[5],
         *yrdl = y + o[6], *yrdu = y + o[7], *yrszl = y + o[8], *yrszu = y + o[9], *yrsvl = y + o[10],
         *yrsvu = y + o[11];
  const double *ixl = h->ixl, *ixu = h->ixu, *idl = h->idl, *idu = h->idu;
  const double dwx = h->pd.wx, dwd = h->pd.wd, dcc = h->pd.cc, dcd = h->pd.cd;
  const bool rnd = h->pd.randomized && !h->pd.null_mode;
  const double *vwx = rnd ? h->dvec[0] : nullptr, *vwd = rnd ? h->dvec[1] : nullptr, *vcc = rnd ? h->dvec[2] : nullptr,
               *vcd = rnd ? h->dvec[3] : nullptr;
  const int64_t n = std::max<int64_t>(std::max<int64_t>(nx, nd), nyc);
  return launch_ew(h->ctx, n, [=] __device__(int64_t i) {
    if(i < nx) {
      const double xv = dx[i];
      yrx[i] += (vwx ? vwx[i] : dwx) * xv - dzl[i] + dzu[i];      // :1672-1678
      yrxl[i] = sel(ixl[i], dsxl[i] - xv);                        // :1697-1699
      yrxu[i] = sel(ixu[i], dsxu[i] + xv);                        // :1702-1704
      yrszl[i] = sxl[i] * dzl[i] + zl[i] * dsxl[i];               // :1717-1719
      yrszu[i] = sxu[i] * dzu[i] + zu[i] * dsxu[i];               // :1722-1724
    }
    if(i < nd) {
      const double dv = dd[i], ydv = dyd[i];
      yrd[i] = -ydv - dvl[i] + dvu[i] + (vwd ? vwd[i] : dwd) * dv;   // :1681-1685
      yryd[i] += -dv - (vcd ? vcd[i] : dcd) * ydv;                // :1692-1694
      yrdl[i] = sel(idl[i], dsdl[i] - dv);                        // :1707-1709
      yrdu[i] = sel(idu[i], dsdu[i] + dv);                        // :1712-1714
      yrsvl[i] = sdl[i] * dvl[i] + vl[i] * dsdl[i];               // :1727-1729
      yrsvu[i] = sdu[i] * dvu[i] + vu[i] * dsdu[i];               // :1732-1734
    }
    if(i < nyc) yryc[i] -= (vcc ? vcc[i] : dcc) * dyc[i];         // :1688-1689
  });
}

int do_factorize(hiopamd_kkt_xycyd* h, int* ok)   // hiopKKTLinSysCurvCheck::factorize (:316-376)
{
  const int max_refactorization = 10;
  h->num_refact = 0;
  *ok = 0;
  if(!h->pd.compute_initial_deltas()) return HIOPAMD_OK;
  while(h->num_refact <= max_refactorization) {
    RC(backend_build(h));
    int n_neg = 0;
    RC(backend_factorize(h, &n_neg));
    const int cont = h->acceptor == 0 ? require_refactorization(h->pd, h->n_required_neg, n_neg)
                                      : require_refactorization_inertia_free(h->pd, h->n_required_neg, n_neg, false);
    if(cont == -1) return HIOPAMD_OK;
    if(cont == 0) break;
    h->num_refact++;
  }
  *ok = h->num_refact <= max_refactorization ? 1 : 0;
  return HIOPAMD_OK;
}

int do_factorize_inertia_free(hiopamd_kkt_xycyd* h, int* ok)   // hiopKKTLinSysCurvCheck::factorize_inertia_free (:376-448)
{
  *ok = 0;
  const int non_singular_mat = 1;
  if(h->acceptor == 0)   // :388 (result unused); hiopFactAcceptorIC ignores force_reg
    (void)require_refactorization(h->pd, h->n_required_neg, non_singular_mat);
  else
    (void)require_refactorization_inertia_free(h->pd, h->n_required_neg, non_singular_mat, true);
  RC(backend_build(h));
  int solver_flag = 0;
  RC(backend_factorize(h, &solver_flag));
  const int max_refactorization = 10;
  h->num_refact = 0;
  while(h->num_refact <= max_refactorization && solver_flag < 0) {
    const int cont = h->acceptor == 0 ? require_refactorization(h->pd, h->n_required_neg, solver_flag)
                                      : require_refactorization_inertia_free(h->pd, h->n_required_neg, solver_flag, false);
    if(cont == -1) return HIOPAMD_OK;
    RC(backend_build(h));
    RC(backend_factorize(h, &solver_flag));
    h->num_refact++;
  }
  *ok = 1;
  return HIOPAMD_OK;
}

int do_compute_directions(hiopamd_kkt_xycyd* h, const double* resid, double* dir, int* ok)
{
  if(!h->iter) return HIOPAMD_ERR_STATE;
  if(resid == dir) return HIOPAMD_ERR_ARG;
  const int64_t* o = h->off;
  {
    SpanScope span(h->ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);   // hiopKKTLinSys.cpp:589-650
    RC(stage_reduce_rhs(h, resid));
  }
  if(h->is_xd()) {   // hiopKKTLinSysCompressedXDYcYd::computeDirections (:810-905)
    RC(backend_solve_xd(h, h->rx_tilde, h->ryd2, resid + o[2], resid + o[3], dir + o[0], dir + o[1], dir + o[2],
                        dir + o[3], ok));
    SpanScope span(h->ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);   // compute_directions_for_full_space (:221-290)
    return stage_recover_directions(h, resid, dir);
  }
  RC(backend_solve(h, h->rx_tilde, resid + o[2], h->ryd_tilde, dir + o[0], dir + o[2], dir + o[3], ok));
  // the reference recovers dd before testing sol_ok and skips the rest on failure (:664-681)
  SpanScope span(h->ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);   // :666-671 + compute_directions_for_full_space (:221-290)
  return stage_recover_directions(h, resid, dir);
}

int do_times_vec(hiopamd_kkt_xycyd* h, double* y, const double* x)
{
  if(!h->iter) return HIOPAMD_ERR_STATE;
  if(x == y) return HIOPAMD_ERR_ARG;
  const int64_t* o = h->off;
  RC(backend_hess_times_vec(h, y + o[0], x + o[0]));
  RC(backend_jac_trans_times_vec_add(h, y + o[0], x + o[2], x + o[3]));
  RC(backend_jac_times_vec(h, y + o[2], x + o[0]));
  return stage_times_vec_ew(h, y, x);
}

// r = b - A*x
int residual_into(hiopamd_kkt_xycyd* h, double* r, const double* b, const double* x)
{
  RC(do_times_vec(h, r, x));
  return launch_ew(h->ctx, h->dim, [=] __device__(int64_t i) { r[i] = b[i] - r[i]; });
}

}  // namespace

extern "C" {

static int create_common(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, int kind, int64_t nx, int nd, int nyc, int nyd,
                         const double* ixl, const double* ixu, const double* idl, const double* idu)
{
  if(!out || !ctx || nx < 0 || nd < 0 || nyc < 0 || nyd < 0) return HIOPAMD_ERR_ARG;
  if((nx > 0 && (!ixl || !ixu)) || (nd > 0 && (!idl || !idu))) return HIOPAMD_ERR_ARG;
  hiopamd_kkt_xycyd* h = new hiopamd_kkt_xycyd();
  h->ctx = ctx;
  h->kind = kind;
  h->nx = nx;
  h->nd = nd;
  h->nyc = nyc;
  h->nyd = nyd;
  const int64_t sz[12] = {nx, nd, nyc, nyd, nx, nx, nd, nd, nx, nx, nd, nd};
  for(int p = 0; p < 12; ++p) h->off[p + 1] = h->off[p] + sz[p];
  h->dim = h->off[12];
  h->ixl = ixl;
  h->ixu = ixu;
  h->idl = idl;
  h->idu = idu;
  h->n_required_neg = nyc + nyd;   // hiopAlgFilterIPM.cpp:2096
  auto A = [](double** p, size_t n) { return hipMalloc((void**)p, sizeof(double) * (n ? n : 1)) == hipSuccess; };
  if(!(A(&h->Dx, nx) && A(&h->Dd, nd) && A(&h->rx_tilde, nx) && A(&h->ryd_tilde, nd) && A(&h->ryd2, nd) &&
       A(&h->dsmall, 4))) {
    hiopamd_kkt_xycyd_destroy(h);
    return HIOPAMD_ERR_HIP;
  }
  *out = h;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_create_mds(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, hiopamd_kkt_mds* k, const double* ixl,
                                 const double* ixu, const double* idl, const double* idu)
{
  if(!k) return HIOPAMD_ERR_ARG;
  int d[4];
  RC(hiopamd_kkt_mds_dims(k, d));
  RC(create_common(out, ctx, KIND_MDS, (int64_t)d[0] + d[1], d[3], d[2], d[3], ixl, ixu, idl, idu));
  (*out)->mds = k;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_create_dense(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, int nx, int neq, int nineq,
                                   const double* ixl, const double* ixu, const double* idl, const double* idu)
{
  RC(create_common(out, ctx, KIND_DENSE, nx, nineq, neq, nineq, ixl, ixu, idl, idu));
  hiopamd_kkt_xycyd* h = *out;
  int rc = hiopamd_linsolver_create(&h->ls, ctx, nx + neq + nineq);
  if(rc == HIOPAMD_OK) rc = hiopamd_linsolver_set_retry_copy(h->ls, 0);   // (backend_factorize re-assembles after a time-out)
  if(rc == HIOPAMD_OK &&
     (hipMalloc((void**)&h->dense_rhs, sizeof(double) * (size_t)(nx + neq + nineq + 1)) != hipSuccess ||
      hipMalloc((void**)&h->dense_Dd_inv, sizeof(double) * (size_t)(nineq + 1)) != hipSuccess))
    rc = HIOPAMD_ERR_HIP;
  if(rc != HIOPAMD_OK) {
    hiopamd_kkt_xycyd_destroy(h);
    *out = nullptr;
  }
  return rc;
}

int hiopamd_kkt_xycyd_create_dense_xdycyd(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, int nx, int neq, int nineq,
                                          const double* ixl, const double* ixu, const double* idl, const double* idu)
{
  RC(create_common(out, ctx, KIND_DENSE_XDYCYD, nx, nineq, neq, nineq, ixl, ixu, idl, idu));
  hiopamd_kkt_xycyd* h = *out;
  const int n = nx + neq + 2 * nineq;
  int rc = hiopamd_linsolver_create(&h->ls, ctx, n);
  if(rc == HIOPAMD_OK) rc = hiopamd_linsolver_set_retry_copy(h->ls, 0);   // (backend_factorize re-assembles after a time-out)
  if(rc == HIOPAMD_OK && hipMalloc((void**)&h->dense_rhs, sizeof(double) * (size_t)(n + 1)) != hipSuccess)
    rc = HIOPAMD_ERR_HIP;
  if(rc != HIOPAMD_OK) {
    hiopamd_kkt_xycyd_destroy(h);
    *out = nullptr;
  }
  return rc;
}

// hiopKKTLinSysCondensedSparse behind the full-space layer: an XDYcYd class for the inequality-only sparse formulation
// (hiopKKTLinSysSparseCondensed.hpp:78: public hiopKKTLinSysCompressedSparseXDYcYd; "this KKT does not support equality
// constraints", .cpp:360)
int hiopamd_kkt_xycyd_create_sparse_condensed(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, hiopamd_kkt_sparse_condensed* k,
                                              const double* ixl, const double* ixu, const double* idl, const double* idu)
{
  if(!k) return HIOPAMD_ERR_ARG;
  int d[4];
  RC(hiopamd_kkt_sparse_condensed_dims(k, d));
  RC(create_common(out, ctx, KIND_SPARSE_CONDENSED, d[0], d[1], 0, d[1], ixl, ixu, idl, idu));
  (*out)->sc = k;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_create_lowrank(hiopamd_kkt_xycyd** out, hiopamd_ctx* ctx, hiopamd_kkt_lowrank* K,
                                     const double* ixl, const double* ixu, const double* idl, const double* idu)
{
  if(!K) return HIOPAMD_ERR_ARG;
  int64_t n = 0;
  int me = 0, mi = 0;
  RC(hiopamd_kkt_lowrank_dims(K, &n, &me, &mi));
  RC(create_common(out, ctx, KIND_LOWRANK, n, mi, me, mi, ixl, ixu, idl, idu));
  (*out)->lr = K;
  (*out)->pd.null_mode = true;   // hiopAlgFilterIPMQuasiNewton uses hiopPDPerturbationNull (hiopAlgFilterIPM.cpp:1054)
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_destroy(hiopamd_kkt_xycyd* h)
{
  if(!h) return HIOPAMD_OK;
  if(h->ls) hiopamd_linsolver_destroy(h->ls);
  (void)hipFree(h->dense_rhs);
  (void)hipFree(h->dense_Dd_inv);
  (void)hipFree(h->Dx);
  (void)hipFree(h->Dd);
  (void)hipFree(h->rx_tilde);
  (void)hipFree(h->ryd_tilde);
  (void)hipFree(h->ryd2);
  (void)hipFree(h->krylov);
  (void)hipFree(h->dsmall);
  (void)hipFree(h->lsq);
  for(double* v : h->dvec) (void)hipFree(v);
  delete h;
  return HIOPAMD_OK;
}

int64_t hiopamd_kkt_xycyd_dim(const hiopamd_kkt_xycyd* h) { return h ? h->dim : -1; }

int hiopamd_kkt_xycyd_offsets(const hiopamd_kkt_xycyd* h, int64_t* off13_host)
{
  if(!h || !off13_host) return HIOPAMD_ERR_ARG;
  for(int p = 0; p < 13; ++p) off13_host[p] = h->off[p];
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_set_matrices(hiopamd_kkt_xycyd* h, const double* H, const double* Jc, const double* Jd)
{
  if(!h) return HIOPAMD_ERR_ARG;
  if(h->kind == KIND_MDS || h->kind == KIND_SPARSE_CONDENSED) return HIOPAMD_ERR_STATE;   // these objects hold their own values (set_values)
  if((h->kind == KIND_DENSE || h->kind == KIND_DENSE_XDYCYD) && !H) return HIOPAMD_ERR_ARG;
  h->H = H;
  h->Jc = Jc;
  h->Jd = Jd;
  // the low-rank object's [Jc; Jd] view is what the operator products use: make it valid before the first update
  if(h->kind == KIND_LOWRANK) return hiopamd_kkt_lowrank_set_jacobians(h->lr, Jc, Jd);
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_set_mu(hiopamd_kkt_xycyd* h, double mu)   // hiopPDPerturbation::set_mu
{
  if(!h) return HIOPAMD_ERR_ARG;
  h->pd.mu = mu;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_set_perturbation_options(hiopamd_kkt_xycyd* h, const double* o)
{
  if(!h || !o) return HIOPAMD_ERR_ARG;
  PdPerturb& p = h->pd;
  p.delta_w_min_bar = o[0];
  p.delta_w_max_bar = o[1];
  p.delta_w_0_bar = o[2];
  p.kappa_w_minus = o[3];
  p.kappa_w_plus_bar = o[4];
  p.kappa_w_plus = o[5];
  p.delta_c_bar = o[6];
  p.kappa_c = o[7];
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_set_regularization(hiopamd_kkt_xycyd* h, int dual_first, int randomized, uint64_t seed)
{
  if(!h) return HIOPAMD_ERR_ARG;
  if(h->pd.null_mode) return HIOPAMD_ERR_STATE;   // the quasi-Newton path runs with hiopPDPerturbationNull
  h->pd.kind = dual_first ? PdPerturb::DualFirst : PdPerturb::PrimalFirst;
  h->pd.randomized = randomized != 0;
  h->pd.dirty = PdPerturb::PDUpdate;
  h->reg_seed = seed;
  h->reg_draw = 0;
  if(randomized && !h->dvec[0]) {
    const int64_t len[4] = {h->nx, h->nd, h->nyc, h->nyd};
    for(int v = 0; v < 4; ++v) {
      HIOPAMD_CHECK(hipMalloc(&h->dvec[v], sizeof(double) * (size_t)std::max<int64_t>(len[v], 1)));
      HIOPAMD_CHECK(hipMemsetAsync(h->dvec[v], 0, sizeof(double) * (size_t)std::max<int64_t>(len[v], 1), h->ctx->stream));
    }
  }
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_delta_vectors(hiopamd_kkt_xycyd* h, const double** wx, const double** wd, const double** cc,
                                    const double** cd)
{
  if(!h) return HIOPAMD_ERR_ARG;
  if(!h->pd.randomized || !h->dvec[0]) return HIOPAMD_ERR_STATE;
  if(wx) *wx = h->dvec[0];
  if(wd) *wd = h->dvec[1];
  if(cc) *cc = h->dvec[2];
  if(cd) *cd = h->dvec[3];
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_set_required_neg_eig(hiopamd_kkt_xycyd* h, int n_required)
{
  if(!h) return HIOPAMD_ERR_ARG;
  h->n_required_neg = n_required;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_set_bicgstab_exit_mode(hiopamd_kkt_xycyd* h, int reference)
{
  if(!h) return HIOPAMD_ERR_ARG;
  h->bicg_ref_exit = reference != 0;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_update(hiopamd_kkt_xycyd* h, const double* iter, int* ok_host)
{
  if(!h || !iter || !ok_host) return HIOPAMD_ERR_ARG;
  h->iter = iter;
  {
    SpanScope span(h->ctx, HIOPAMD_SPAN_KKT_UPDATE_INIT);   // hiopKKTLinSys.cpp:550-576
    RC(stage_update_diagonals(h));
    if(h->kind == KIND_MDS) {
      RC(hiopamd_kkt_mds_set_diagonals(h->mds, h->Dx, h->Dd));
    } else if(h->kind == KIND_SPARSE_CONDENSED) {
      RC(hiopamd_kkt_sparse_condensed_set_diagonals(h->sc, h->Dx, h->Dd));
    } else if(h->kind == KIND_LOWRANK) {
      // hiopKKTLinSysLowRank::update (hiopKKTLinSys.cpp:1057-1096): refresh the Hessian's log-barrier diagonal, Dd^-1
      if((!h->Jc && h->nyc > 0) || (!h->Jd && h->nyd > 0)) return HIOPAMD_ERR_STATE;
      RC(hiopamd_kkt_lowrank_update_diag(h->lr, h->Dx, h->Dd, h->Jc, h->Jd));
    }
  }
  return do_factorize(h, ok_host);
}

int hiopamd_kkt_xycyd_factorize(hiopamd_kkt_xycyd* h, int* ok_host)
{
  if(!h || !ok_host) return HIOPAMD_ERR_ARG;
  if(!h->iter) return HIOPAMD_ERR_STATE;
  return do_factorize(h, ok_host);
}

int hiopamd_kkt_xycyd_set_fact_acceptor(hiopamd_kkt_xycyd* h, int kind)
{
  if(!h || (kind != 0 && kind != 1)) return HIOPAMD_ERR_ARG;
  h->acceptor = kind;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_factorize_inertia_free(hiopamd_kkt_xycyd* h, int* ok_host)
{
  if(!h || !ok_host) return HIOPAMD_ERR_ARG;
  if(!h->iter) return HIOPAMD_ERR_STATE;
  return do_factorize_inertia_free(h, ok_host);
}

// hiopKKTLinSysCompressed::test_direction (:455-513): accept iff  dx'(H + Dx + delta_wx)dx + dd'(Dd + delta_wd)dd
//   >= neg_curv_test_fact * (||dx||^2 + ||dd||^2)
int hiopamd_kkt_xycyd_test_direction(hiopamd_kkt_xycyd* h, const double* dir, double neg_curv_test_fact,
                                     int* accept_host, double* dWd_host, double* xs_nrmsq_host)
{
  if(!h || !dir || !accept_host) return HIOPAMD_ERR_ARG;
  if(!h->iter) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = h->ctx;
  const double *dx = dir + h->off[0], *dd = dir + h->off[1];
  double* hx = h->rx_tilde;   // work
  RC(backend_hess_times_vec(h, hx, dx));
  const double *Dx = h->Dx, *Dd = h->Dd;
  const double dwx = h->pd.wx, dwd = h->pd.wd;
  const int64_t nx = h->nx;
  // {curvature term, squared norm} over the x part (distributed on a column partition) and over the d part
  const bool rnd = h->pd.randomized && !h->pd.null_mode;
  struct OpCurv {
    const double *w, *v, *D;
    double delta;
    const double* dvec;
    __device__ dot2_t identity() const { return dot2_t{0.0, 0.0}; }
    __device__ dot2_t map(int64_t i) const
    {
      const double vi = v[i];
      return dot2_t{(w ? w[i] * vi : 0.0) + (D[i] * vi + (dvec ? dvec[i] : delta) * vi) * vi, vi * vi};
    }
    __device__ dot2_t combine(dot2_t p, dot2_t q) const { return dot2_t{p.dist + q.dist, p.repl + q.repl}; }
  };
  dot2_t sx{0, 0}, sd{0, 0};
  RC(launch_reduce<dot2_t>(ctx, nx, OpCurv{hx, dx, Dx, dwx, rnd ? h->dvec[0] : nullptr}, &sx));
  if(h->kind == KIND_LOWRANK && ctx->allreduce) {
    HIOPAMD_CHECK(hipMemcpyAsync(h->dsmall, &sx, sizeof(sx), hipMemcpyHostToDevice, ctx->stream));
    HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
    if(ctx_allreduce(ctx, h->dsmall, 2, HIOPAMD_SUM) != 0) return HIOPAMD_ERR_HIP;
    HIOPAMD_CHECK(hipMemcpyAsync(&sx, h->dsmall, sizeof(sx), hipMemcpyDeviceToHost, ctx->stream));
    HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  }
  RC(launch_reduce<dot2_t>(ctx, h->nd, OpCurv{nullptr, dd, Dd, dwd, rnd ? h->dvec[1] : nullptr}, &sd));
  const double dWd = sx.dist + sd.dist, xs_nrmsq = sx.repl + sd.repl;
  if(dWd_host) *dWd_host = dWd;
  if(xs_nrmsq_host) *xs_nrmsq_host = xs_nrmsq;
  *accept_host = (dWd < xs_nrmsq * neg_curv_test_fact) ? 0 : 1;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_deltas(const hiopamd_kkt_xycyd* h, double* d4)
{
  if(!h || !d4) return HIOPAMD_ERR_ARG;
  d4[0] = h->pd.wx;
  d4[1] = h->pd.wd;
  d4[2] = h->pd.cc;
  d4[3] = h->pd.cd;
  return HIOPAMD_OK;
}

int hiopamd_kkt_xycyd_num_refactorizations(const hiopamd_kkt_xycyd* h) { return h ? h->num_refact : -1; }

int hiopamd_kkt_xycyd_compute_directions(hiopamd_kkt_xycyd* h, const double* resid, double* dir, int* ok_host)
{
  if(!h || !resid || !dir || !ok_host) return HIOPAMD_ERR_ARG;
  return do_compute_directions(h, resid, dir, ok_host);
}

int hiopamd_kkt_xycyd_times_vec(hiopamd_kkt_xycyd* h, double* y, const double* x)
{
  if(!h || !x || !y) return HIOPAMD_ERR_ARG;
  return do_times_vec(h, y, x);
}

// compute_directions_w_IR (:911-961) = BiCGStab (hiopKrylovSolver.cpp:390-700) on the 12-block operator with the
// condensed solve as left preconditioner, x0 = 0, tol = min(mu*tol_factor, tol_min), relative to ||rhs||_2.
// info4_host = {flag, iter, abs_resid, rel_resid}; *ok_host = 1 always once the solve ran ("accept the step since
// this is IR", :949-953), *converged_host reports BiCGStab's own verdict.
int hiopamd_kkt_xycyd_compute_directions_w_IR(hiopamd_kkt_xycyd* h, const double* resid, double* dir,
                                              double ir_outer_tol_factor, double ir_outer_tol_min, int ir_outer_maxit,
                                              int* ok_host, int* converged_host, double* info4_host)
{
  if(!h || !resid || !dir || !ok_host) return HIOPAMD_ERR_ARG;
  if(!h->iter) return HIOPAMD_ERR_STATE;
  double info[4] = {0, 0, 0, 0};
  int conv = 1;
  auto finish = [&](int rc) {
    if(converged_host) *converged_host = conv;
    if(info4_host)
      for(int q = 0; q < 4; ++q) info4_host[q] = info[q];
    return rc;
  };
  if(ir_outer_maxit <= 0) return finish(do_compute_directions(h, resid, dir, ok_host));   // :916-919
  *ok_host = 1;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t n = h->dim;
  if(!h->krylov && hipMalloc((void**)&h->krylov, sizeof(double) * (size_t)(9 * n + 1)) != hipSuccess)
    return HIOPAMD_ERR_HIP;
  double *xk = h->krylov, *xmin = xk + n, *res = xmin + n, *pk = res + n, *ph = pk + n, *v = ph + n, *sk = v + n,
         *t = sk + n, *rt = t + n;
  const double tol = std::min(h->pd.mu * ir_outer_tol_factor, ir_outer_tol_min);
  const double* b = resid;
  double n2b = 0.0;
  RC(slab_norm(h, b, &n2b));
  if(n2b == 0.0) {   // rhs = 0 -> solution = 0 (:405-413)
    RC(hiopamd_vec_set_to_constant(ctx, n, dir, 0.0));
    return finish(HIOPAMD_OK);
  }
  int flag = 1;
  double iter = 0.0, imin = 0.0;
  const double tolb = tol * n2b;
  RC(hiopamd_vec_set_to_constant(ctx, n, xk, 0.0));     // set_x0(0.0)
  RC(hiopamd_vec_set_to_constant(ctx, n, xmin, 0.0));
  RC(residual_into(h, res, b, xk));                       // :446-449
  double normr = 0.0;
  RC(slab_norm(h, res, &normr));
  double abs_resid = normr;
  if(normr <= tolb) {                                     // :453-461
    RC(hiopamd_vec_copy(ctx, n, dir, xk));
    info[2] = normr;
    info[3] = normr / n2b;
    return finish(HIOPAMD_OK);
  }
  RC(hiopamd_vec_copy(ctx, n, rt, res));
  double normrmin = normr, rho = 1.0, omega = 1.0, alpha = 0.0, rho1;
  int stagsteps = 0, moresteps = 0;
  // the reference's 'tol is too small' exits overwrite the right-hand side with xk before the closing comparison of the minimal-residual
  // iterate (hiopKrylovSolver.cpp:561-566, :639-644, :671-688); the caller's right-hand side slab is left alone here, the comparison below
  // is made against xk instead when this is set (default; hiopamd_kkt_xycyd_set_bicgstab_exit_mode(h, 0): against the original b)
  bool b_is_xk = false;
  const double eps = std::numeric_limits<double>::epsilon();
  const int maxmsteps = 100, maxstagsteps = 3;
  int ok_prec = 1;
  int ii = 0;
  for(; ii < ir_outer_maxit; ++ii) {
    rho1 = rho;
    RC(slab_dot(h, rt, res, &rho));
    if(rho == 0 || std::abs(rho) > 1e40) {
      flag = 4;
      iter = ii + 1 - 0.5;
      break;
    }
    if(ii == 0) {
      RC(hiopamd_vec_copy(ctx, n, pk, res));
    } else {
      const double beta = rho / rho1 * (alpha / omega);
      if(beta == 0 || std::abs(beta) > 1e40) {
        flag = 4;
        iter = ii + 1 - 0.5;
        break;
      }
      const double om = omega;
      RC(launch_ew(ctx, n, [=] __device__(int64_t i) { pk[i] = (pk[i] - om * v[i]) * beta + res[i]; }));   // :498-500
    }
    RC(do_compute_directions(h, pk, ph, &ok_prec));   // ph = M^-1 pk (hiopPrecondKKTOpr::times_vec)
    if(!ok_prec) {   // the condensed solve behind the preconditioner failed: Krylov flag 4 (breakdown), never a silent continue
      flag = 4;
      iter = ii + 1 - 0.5;
      break;
    }
    RC(do_times_vec(h, v, ph));
    double rtv = 0.0;
    RC(slab_dot(h, rt, v, &rtv));
    if(rtv == 0.0 || std::abs(rtv) > 1e40) {
      flag = 4;
      iter = ii + 1 - 0.5;
      break;
    }
    alpha = rho / rtv;
    if(std::abs(alpha) > 1e20) {
      flag = 4;
      iter = ii + 1 - 0.5;
      break;
    }
    double nph = 0.0, nxk = 0.0;
    RC(slab_norm(h, ph, &nph));
    RC(slab_norm(h, xk, &nxk));
    stagsteps = (nph * std::abs(alpha) < eps * nxk) ? stagsteps + 1 : 0;   // :531-535
    {
      const double al = alpha;
      RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
        xk[i] += al * ph[i];
        sk[i] = res[i] - al * v[i];
      }));
    }
    RC(slab_norm(h, sk, &normr));
    abs_resid = normr;
    if(normr <= tolb || stagsteps >= maxstagsteps || moresteps) {   // :546-570
      RC(residual_into(h, sk, b, xk));
      RC(slab_norm(h, sk, &abs_resid));
      if(abs_resid <= tolb) {
        flag = 0;
        iter = ii + 1 - 0.5;
        break;
      }
      if(stagsteps >= maxstagsteps && moresteps == 0) stagsteps = 0;
      moresteps++;
      if(moresteps >= maxmsteps) {
        b_is_xk = h->bicg_ref_exit != 0;   // :563 b->copyFrom(*xk_)
        flag = 3;
        iter = ii + 1 - 0.5;
        break;
      }
    }
    if(stagsteps >= maxstagsteps) {
      iter = ii + 1 - 0.5;
      flag = 3;
      break;
    }
    if(abs_resid < normrmin) {
      normrmin = abs_resid;
      RC(hiopamd_vec_copy(ctx, n, xmin, xk));
      imin = ii + 1 - 0.5;
    }
    RC(do_compute_directions(h, sk, ph, &ok_prec));
    if(!ok_prec) {
      flag = 4;
      iter = ii + 1;
      break;
    }
    RC(do_times_vec(h, t, ph));
    double tt = 0.0, ts = 0.0;
    RC(slab_dot(h, t, t, &tt));
    if(tt == 0.0 || std::abs(tt) > 1e20) {
      iter = ii + 1;
      flag = 4;
      break;
    }
    RC(slab_dot(h, t, sk, &ts));
    omega = ts / tt;
    if(std::abs(omega) > 1e20) {
      iter = ii + 1;
      flag = 4;
      break;
    }
    RC(slab_norm(h, ph, &nph));
    RC(slab_norm(h, xk, &nxk));
    stagsteps = (nph * std::abs(omega) < eps * nxk) ? stagsteps + 1 : 0;
    {
      const double om = omega;
      RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
        xk[i] += om * ph[i];
        res[i] = sk[i] - om * t[i];
      }));
    }
    RC(slab_norm(h, res, &normr));
    abs_resid = normr;
    if(normr <= tolb || stagsteps >= maxstagsteps || moresteps) {   // :623-648
      RC(residual_into(h, res, b, xk));
      RC(slab_norm(h, res, &abs_resid));
      if(abs_resid <= tolb) {
        flag = 0;
        iter = ii + 1;
        break;
      }
      if(stagsteps >= maxstagsteps && moresteps == 0) stagsteps = 0;
      moresteps++;
      if(moresteps >= maxmsteps) {
        b_is_xk = h->bicg_ref_exit != 0;   // :641
        flag = 3;
        iter = ii + 1;
        break;
      }
    }
    if(abs_resid < normrmin) {
      normrmin = abs_resid;
      RC(hiopamd_vec_copy(ctx, n, xmin, xk));
      imin = ii + 1;
    }
    if(stagsteps >= maxstagsteps) {
      iter = ii + 1 - 0.5;
      flag = 3;
      break;
    }
  }
  double rel_resid;
  if(flag == 0) {                                          // :665-669
    rel_resid = abs_resid / n2b;
    RC(hiopamd_vec_copy(ctx, n, dir, xk));
    conv = 1;
  } else {                                                 // :671-688
    RC(residual_into(h, res, b_is_xk ? xk : b, xmin));
    double normr_comp = 0.0;
    RC(slab_norm(h, res, &normr_comp));
    if(normr_comp <= abs_resid) {
      RC(hiopamd_vec_copy(ctx, n, dir, xmin));
      iter = imin + 1;
      abs_resid = normr_comp;
      rel_resid = normr_comp / n2b;
    } else {
      RC(hiopamd_vec_copy(ctx, n, dir, xk));
      iter = ii + 1;
      rel_resid = abs_resid / n2b;
    }
    conv = 0;
  }
  info[0] = flag;
  info[1] = iter;
  info[2] = abs_resid;
  info[3] = rel_resid;
  return finish(HIOPAMD_OK);
}

hiopamd_linsolver* hiopamd_kkt_xycyd_linsolver(hiopamd_kkt_xycyd* h) { return h ? h->ls : nullptr; }
double* hiopamd_kkt_xycyd_Dx(hiopamd_kkt_xycyd* h) { return h ? h->Dx : nullptr; }
double* hiopamd_kkt_xycyd_Dd(hiopamd_kkt_xycyd* h) { return h ? h->Dd : nullptr; }

}  // extern "C"

// =========================================================================================================
// hiopIterate / hiopResidual on the 12-part slabs: the steps either side of the KKT solve
//   hiopResidual::update                       src/Optimization/hiopResidual.cpp:154-365
//   hiopIterate::fractionToTheBdry             src/Optimization/hiopIterate.cpp:330-362
//   hiopIterate::takeStep_primals / _duals     :367-390
//   hiopIterate::determineSlacks / compute_safe_slacks / adjust_small_slacks   :274-312, :414-505
//   hiopIterate::determineDualsBounds_d        :314-327
//   hiopIterate::adjustDuals_primalLogHessian  :507-521
//   hiopIterate::evalLogBarrier / linearDampingTerm   :523-566
// =========================================================================================================
namespace {

// four running values, max for components [0, nmax), sum for the rest
struct red4_t {
  double v[4];
};
template <class Map>
struct OpRed4 {
  Map m;
  int nmax;
  __device__ red4_t identity() const { return red4_t{{0.0, 0.0, 0.0, 0.0}}; }
  __device__ red4_t map(int64_t i) const { return m(i); }
  __device__ red4_t combine(red4_t p, red4_t q) const
  {
    red4_t r;
#pragma unroll
    for(int c = 0; c < 4; ++c) r.v[c] = (c < nmax) ? (p.v[c] > q.v[c] ? p.v[c] : q.v[c]) : (p.v[c] + q.v[c]);
    return r;
  }
};
template <class Map>
int reduce4(hiopamd_ctx* ctx, int64_t n, Map m, int nmax, red4_t* out)
{
  *out = red4_t{{0.0, 0.0, 0.0, 0.0}};
  if(n <= 0) return HIOPAMD_OK;
  return launch_reduce<red4_t>(ctx, n, OpRed4<Map>{m, nmax}, out);
}

// all-reduce of the x-part contributions on a column partition (low-rank back-end with a hook): max then sum parts
int xpart_allreduce(hiopamd_kkt_xycyd* h, red4_t* r, int nmax)
{
  hiopamd_ctx* ctx = h->ctx;
  if(!(h->kind == KIND_LOWRANK && ctx->allreduce)) return HIOPAMD_OK;
  HIOPAMD_CHECK(hipMemcpyAsync(h->dsmall, r->v, 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  if(nmax > 0 && ctx_allreduce(ctx, h->dsmall, (size_t)nmax, HIOPAMD_MAX) != 0)
    return HIOPAMD_ERR_HIP;
  if(nmax < 4 && ctx_allreduce(ctx, h->dsmall + nmax, (size_t)(4 - nmax), HIOPAMD_SUM) != 0)
    return HIOPAMD_ERR_HIP;
  HIOPAMD_CHECK(hipMemcpyAsync(r->v, h->dsmall, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  return HIOPAMD_OK;
}

}  // namespace

extern "C" {

int hiopamd_kkt_xycyd_set_bounds(hiopamd_kkt_xycyd* h, const double* xl, const double* xu, const double* dl,
                                 const double* du, const double* crhs)
{
  if(!h) return HIOPAMD_ERR_ARG;
  h->xl = xl;
  h->xu = xu;
  h->dl = dl;
  h->du = du;
  h->crhs = crhs;
  return HIOPAMD_OK;
}

// hiopResidual::update.  c, d: constraint bodies at x (device, nyc / nyd); grad_f (nx).  norms11_host =
//   [nrmInf_nlp_optim, nrmInf_nlp_feasib, nrmInf_nlp_complem, nrmInf_bar_optim, nrmInf_bar_feasib, nrmInf_bar_complem,
//    nrmOne_nlp_feasib, nrmOne_bar_feasib, nrmOne_nlp_optim, nrmOne_bar_optim, nrmInf_cons_violation]
int hiopamd_residual_update(hiopamd_kkt_xycyd* h, const double* it, const double* c, const double* d, const double* grad_f,
                            double mu, double kappa_d, double* resid, double* norms11_host)
{
  if(!h || !it || !grad_f || !resid || !norms11_host) return HIOPAMD_ERR_ARG;
  if((h->nx > 0 && (!h->xl || !h->xu)) || (h->nd > 0 && (!h->dl || !h->du || !d)) || (h->nyc > 0 && (!h->crhs || !c)))
    return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t* o = h->off;
  const int64_t nx = h->nx, nd = h->nd, nyc = h->nyc;
  const double *x = it + o[0], *dit = it + o[1], *yc = it + o[2], *yd = it + o[3], *sxl = it + o[4], *sxu = it + o[5],
               *sdl = it + o[6], *sdu = it + o[7], *zl = it + o[8], *zu = it + o[9], *vl = it + o[10], *vu = it + o[11];
  double *rx = resid + o[0], *rd = resid + o[1], *ryc = resid + o[2], *ryd = resid + o[3], *rxl = resid + o[4],
         *rxu = resid + o[5], *rdl = resid + o[6], *rdu = resid + o[7], *rszl = resid + o[8], *rszu = resid + o[9],
         *rsvl = resid + o[10], *rsvu = resid + o[11];
  const double *ixl = h->ixl, *ixu = h->ixu, *idl = h->idl, *idu = h->idu;
  const double *xl = h->xl, *xu = h->xu, *dl = h->dl, *du = h->du, *crhs = h->crhs;
  // rx = grad_f + Jc^T yc + Jd^T yd   (the back-end's matrices: those of the last update / set_values)   :181-184
  RC(hiopamd_vec_copy(ctx, nx, rx, grad_f));
  RC(backend_jac_trans_times_vec_add(h, rx, yc, yd));
  const double ctx_damp = (kappa_d > 0.0) ? kappa_d * mu : 0.0;   // addNonLogBarTermsToGrad_* (hiopLogBarProblem.hpp:135-145)
  // x-sized parts: one pass; returns {inf(rx_nlp), inf(rx_bar), one(rx_nlp), one(rx_bar)} and later the complementarity maxima
  red4_t ox, cx;
  RC(reduce4(ctx, nx,
             [=] __device__(int64_t i) {
               const double pre = rx[i] - zl[i] + zu[i];                          // :185-186
               const double bar = -(pre + (ixl[i] - ixu[i]) * ctx_damp);          // :195-196
               rx[i] = bar;
               rxl[i] = ixl[i] == 0.0 ? 0.0 : x[i] - sxl[i] - xl[i];              // :236-243
               rxu[i] = ixu[i] == 0.0 ? 0.0 : xu[i] - x[i] - sxu[i];              // :248-254
               const double cl = ixl[i] == 0.0 ? 0.0 : -sxl[i] * zl[i];           // :287-292
               const double cu = ixu[i] == 0.0 ? 0.0 : -sxu[i] * zu[i];           // :302-307
               rszl[i] = ixl[i] == 1.0 ? cl + mu : cl;                            // :295
               rszu[i] = ixu[i] == 1.0 ? cu + mu : cu;                            // :311
               return red4_t{{fabs(pre), fabs(bar), fabs(pre), fabs(bar)}};
             },
             2, &ox));
  RC(reduce4(ctx, nx,
             [=] __device__(int64_t i) {
               const double cl = ixl[i] == 0.0 ? 0.0 : -sxl[i] * zl[i];
               const double cu = ixu[i] == 0.0 ? 0.0 : -sxu[i] * zu[i];
               return red4_t{{fmax(fabs(cl), fabs(cu)), fmax(fabs(rszl[i]), fabs(rszu[i])), 0.0, 0.0}};
             },
             2, &cx));
  RC(xpart_allreduce(h, &ox, 2));
  RC(xpart_allreduce(h, &cx, 2));
  // d-sized parts (replicated)
  red4_t od, fd, cd;
  RC(reduce4(ctx, nd,
             [=] __device__(int64_t i) {
               const double pre = yd[i] + vl[i] - vu[i];                          // :203-205
               const double bar = pre + (idl[i] - idu[i]) * (-ctx_damp);          // :212
               rd[i] = bar;
               return red4_t{{fabs(pre), fabs(bar), fabs(pre), fabs(bar)}};
             },
             2, &od));
  RC(reduce4(ctx, nd,
             [=] __device__(int64_t i) {
               const double r = dit[i] - d[i];                                    // :229-230
               ryd[i] = r;
               rdl[i] = idl[i] == 0.0 ? 0.0 : dit[i] - sdl[i] - dl[i];            // :260-262
               rdu[i] = idu[i] == 0.0 ? 0.0 : du[i] - sdu[i] - dit[i];            // :268-272
               // constraint violation of the inequality bodies                    :219-227
               double viol = 0.0;
               if(idl[i] == 1.0) viol = fmax(viol, -(d[i] - dl[i]));
               if(idu[i] == 1.0) viol = fmax(viol, -(du[i] - d[i]));
               return red4_t{{fabs(r), viol, fabs(r), 0.0}};
             },
             2, &fd));
  RC(reduce4(ctx, nd,
             [=] __device__(int64_t i) {
               const double cl = idl[i] == 0.0 ? 0.0 : -sdl[i] * vl[i];           // :317-323
               const double cu = idu[i] == 0.0 ? 0.0 : -sdu[i] * vu[i];           // :332-338
               const double bl = idl[i] == 1.0 ? cl + mu : cl;
               const double bu = idu[i] == 1.0 ? cu + mu : cu;
               rsvl[i] = bl;
               rsvu[i] = bu;
               return red4_t{{fmax(fabs(cl), fabs(cu)), fmax(fabs(bl), fabs(bu)), 0.0, 0.0}};
             },
             2, &cd));
  // equality part
  red4_t fc;
  RC(reduce4(ctx, nyc,
             [=] __device__(int64_t i) {
               const double r = crhs[i] - c[i];                                   // :214-215
               ryc[i] = r;
               return red4_t{{fabs(r), 0.0, fabs(r), 0.0}};
             },
             2, &fc));
  double* n = norms11_host;
  n[0] = fmax(ox.v[0], od.v[0]);                 // nrmInf_nlp_optim
  n[8] = ox.v[2] + od.v[2];                      // nrmOne_nlp_optim
  n[3] = fmax(ox.v[1], od.v[1]);                 // nrmInf_bar_optim
  n[9] = ox.v[3] + od.v[3];                      // nrmOne_bar_optim
  n[1] = fmax(fc.v[0], fd.v[0]);                 // nrmInf_nlp_feasib
  n[6] = fc.v[2] + fd.v[2];                      // nrmOne_nlp_feasib
  n[4] = n[1];                                   // nrmInf_bar_feasib  (:279)
  n[7] = n[6];                                   // nrmOne_bar_feasib  (:280)
  n[10] = fmax(fc.v[0], fd.v[1]);                // nrmInf_cons_violation (:218-227)
  n[2] = fmax(cx.v[0], cd.v[0]);                 // nrmInf_nlp_complem
  n[5] = fmax(cx.v[1], cd.v[1]);                 // nrmInf_bar_complem
  return HIOPAMD_OK;
}

// hiopIterate::fractionToTheBdry: one reduction over the 8 slack/dual parts (x-sized ones all-reduced with MIN)
int hiopamd_iterate_fraction_to_the_bdry(hiopamd_kkt_xycyd* h, const double* it, const double* dir, double tau,
                                         double* alpha_primal_host, double* alpha_dual_host)
{
  if(!h || !it || !dir || !alpha_primal_host || !alpha_dual_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t* o = h->off;
  double ap = 10.0, ad = 10.0;   // :334
  auto pass = [&](const int (&parts)[4], double* out) -> int {
    int64_t n4[4];
    const double *x4[4], *d4[4], *s4[4];
    const double* pat[12] = {nullptr, nullptr, nullptr, nullptr, h->ixl, h->ixu, h->idl, h->idu, h->ixl, h->ixu, h->idl, h->idu};
    for(int q = 0; q < 4; ++q) {
      const int pidx = parts[q];
      n4[q] = o[pidx + 1] - o[pidx];
      x4[q] = it + o[pidx];
      d4[q] = dir + o[pidx];
      s4[q] = pat[pidx];
    }
    double r = 1.0;
    ReduceNow now(ctx);
    RC(hiopamd_vec_fraction_to_the_bdry_multi(ctx, 4, n4, x4, d4, s4, tau, &r));
    *out = r;
    return HIOPAMD_OK;
  };
  double a1 = 1.0, a2 = 1.0;
  const int prim[4] = {4, 5, 6, 7}, dual[4] = {8, 9, 10, 11};
  RC(pass(prim, &a1));
  RC(pass(dual, &a2));
  ap = std::fmin(ap, a1);
  ad = std::fmin(ad, a2);
  if(h->kind == KIND_LOWRANK && ctx->allreduce) {   // MPI_Allreduce(MIN) :355-359  (max of the negatives)
    double buf[2] = {-ap, -ad};
    HIOPAMD_CHECK(hipMemcpyAsync(h->dsmall, buf, sizeof(buf), hipMemcpyHostToDevice, ctx->stream));
    HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
    if(ctx_allreduce(ctx, h->dsmall, 2, HIOPAMD_MAX) != 0) return HIOPAMD_ERR_HIP;
    HIOPAMD_CHECK(hipMemcpyAsync(buf, h->dsmall, sizeof(buf), hipMemcpyDeviceToHost, ctx->stream));
    HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
    ap = -buf[0];
    ad = -buf[1];
  }
  *alpha_primal_host = ap;
  *alpha_dual_host = ad;
  return HIOPAMD_OK;
}

// takeStep_primals (x, d) and/or takeStep_duals (yc, yd with alpha_primal; zl, zu, vl, vu with alpha_dual)
int hiopamd_iterate_take_step(hiopamd_kkt_xycyd* h, double* out, const double* it, const double* dir, double alpha_primal,
                              double alpha_dual, int primals, int duals)
{
  if(!h || !out || !it || !dir) return HIOPAMD_ERR_ARG;
  const int64_t o2 = h->off[2], o4 = h->off[4], o8 = h->off[8], o12 = h->off[12];
  return launch_ew(h->ctx, o12, [=] __device__(int64_t i) {
    if(i < o2) {
      if(primals) out[i] = it[i] + alpha_primal * dir[i];
    } else if(i < o4) {
      if(duals) out[i] = it[i] + alpha_primal * dir[i];
    } else if(i >= o8) {
      if(duals) out[i] = it[i] + alpha_dual * dir[i];
    }
  });
}

// determineSlacks (:274-291)
int hiopamd_iterate_determine_slacks(hiopamd_kkt_xycyd* h, double* it)
{
  if(!h || !it) return HIOPAMD_ERR_ARG;
  if((h->nx > 0 && (!h->xl || !h->xu)) || (h->nd > 0 && (!h->dl || !h->du))) return HIOPAMD_ERR_STATE;
  const int64_t* o = h->off;
  const int64_t nx = h->nx, nd = h->nd;
  const double *x = it + o[0], *d = it + o[1];
  double *sxl = it + o[4], *sxu = it + o[5], *sdl = it + o[6], *sdu = it + o[7];
  const double *ixl = h->ixl, *ixu = h->ixu, *idl = h->idl, *idu = h->idu, *xl = h->xl, *xu = h->xu, *dl = h->dl, *du = h->du;
  return launch_ew(h->ctx, std::max<int64_t>(nx, nd), [=] __device__(int64_t i) {
    if(i < nx) {
      sxl[i] = ixl[i] == 0.0 ? 0.0 : x[i] - xl[i];
      sxu[i] = ixu[i] == 0.0 ? 0.0 : xu[i] - x[i];
    }
    if(i < nd) {
      sdl[i] = idl[i] == 0.0 ? 0.0 : d[i] - dl[i];
      sdu[i] = idu[i] == 0.0 ? 0.0 : du[i] - d[i];
    }
  });
}

// hiopNlpFormulation::adjust_bounds (hiopNlpFormulation.cpp:1403-1416)
int hiopamd_iterate_adjust_bounds(hiopamd_kkt_xycyd* h, const double* it, double* xl, double* xu, double* dl, double* du)
{
  if(!h || !it) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t* o = h->off;
  const int64_t nx = o[1] - o[0], nd = o[2] - o[1];
  double* bnd[4] = {xl, xu, dl, du};
  const double* pat[4] = {h->ixl, h->ixu, h->idl, h->idu};
  for(int q = 0; q < 4; ++q) {
    const int64_t n = q < 2 ? nx : nd;
    if(n <= 0) continue;
    if(!bnd[q] || !pat[q]) return HIOPAMD_ERR_ARG;
    const double* prim = it + o[q < 2 ? 0 : 1];
    RC(hiopamd_vec_copy_from_w_pattern(ctx, n, bnd[q], prim, pat[q]));
    RC(hiopamd_vec_axpy_w_pattern(ctx, n, bnd[q], (q & 1) ? 1.0 : -1.0, it + o[4 + q], pat[q]));
  }
  return HIOPAMD_OK;
}

// adjust_small_slacks (:414-505) for the four slack parts of `it`, duals taken from `it_curr`; returns the number adjusted
int hiopamd_iterate_adjust_small_slacks(hiopamd_kkt_xycyd* h, double* it, const double* it_curr, double mu,
                                        int* num_adjusted_host)
{
  if(!h || !it || !it_curr || !num_adjusted_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t* o = h->off;
  const double eps = std::numeric_limits<double>::epsilon();
  const double small_val = eps * std::fmin(1.0, mu);
  const double scale_fact = std::pow(eps, 0.75);
  const double* pat[4] = {h->ixl, h->ixu, h->idl, h->idu};
  const double* bnd[4] = {h->xl, h->xu, h->dl, h->du};
  int total = 0;
  for(int q = 0; q < 4; ++q) {
    const int64_t n = o[4 + q + 1] - o[4 + q];
    if(n <= 0) continue;
    double* slack = it + o[4 + q];
    const double* dual = it_curr + o[8 + q];
    const double *sel_ = pat[q], *bound = bnd[q];
    if(!bound) return HIOPAMD_ERR_STATE;
    double slack_min = 0.0;
    {
      ReduceNow now(ctx);
      RC(hiopamd_vec_min_w_pattern(ctx, n, slack, sel_, &slack_min));   // :435
    }
    if(q < 2) {
      // the x-sized slacks are column-sharded on the low-rank back-end: hiopVectorPar::min_w_pattern all-reduces (MIN,
      // hiopVectorPar.cpp:833-836), so every rank takes the same branch below
      red4_t mn{{-slack_min, 0.0, 0.0, 0.0}};
      RC(xpart_allreduce(h, &mn, 1));
      slack_min = -mn.v[0];
    }
    if(!(slack_min < small_val)) continue;
    red4_t cnt;
    RC(reduce4(ctx, n,
               [=] __device__(int64_t i) {
                 const double s0 = slack[i];
                 // arg1 = -sgn(min(slack - small_val [selected], 0))  -> 1 where the slack is too small, else 0
                 double a1 = s0 + (sel_[i] == 1.0 ? -small_val : 0.0);
                 a1 = a1 < 0.0 ? a1 : 0.0;
                 const double flag = a1 < 0.0 ? 1.0 : 0.0;
                 const double sl = s0 > 0.0 ? s0 : 0.0;                            // slack.component_max(0)
                 double a2 = sel_[i] == 1.0 ? mu : 0.0;                            // mu / dual on the pattern
                 a2 = sel_[i] == 0.0 ? 0.0 : a2 / dual[i];
                 const double a3 = sel_[i] == 1.0 ? small_val : 0.0;
                 a2 = (a2 > a3 ? a2 : a3) - sl;
                 double n1 = flag * a2 + sl;                                       // candidate 1
                 double b2 = sel_[i] == 1.0 ? 1.0 : 0.0;
                 const double ab = fabs(bound[i]);
                 b2 = (b2 > ab ? b2 : ab) * scale_fact + sl;                       // cap
                 n1 = n1 < b2 ? n1 : b2;
                 slack[i] = n1;
                 return red4_t{{0.0, 0.0, flag, 0.0}};
               },
               2, &cnt));
    if(q < 2) RC(xpart_allreduce(h, &cnt, 2));   // numOfElemsLessThan all-reduces (SUM, hiopVectorPar.cpp:1231-1236)
    total += (int)(cnt.v[2] + 0.5);
  }
  *num_adjusted_host = total;
  return HIOPAMD_OK;
}

// determineDualsBounds_d (:314-327): vl = mu / sdl, vu = mu / sdu on the patterns
int hiopamd_iterate_determine_duals_bounds_d(hiopamd_kkt_xycyd* h, double* it, double mu)
{
  if(!h || !it) return HIOPAMD_ERR_ARG;
  const int64_t* o = h->off;
  const double *sdl = it + o[6], *sdu = it + o[7], *idl = h->idl, *idu = h->idu;
  double *vl = it + o[10], *vu = it + o[11];
  return launch_ew(h->ctx, h->nd, [=] __device__(int64_t i) {
    vl[i] = idl[i] == 0.0 ? 0.0 : mu / sdl[i];
    vu[i] = idu[i] == 0.0 ? 0.0 : mu / sdu[i];
  });
}

// adjustDuals_primalLogHessian (:507-521)
int hiopamd_iterate_adjust_duals_plh(hiopamd_kkt_xycyd* h, double* it, double mu, double kappa_Sigma)
{
  if(!h || !it) return HIOPAMD_ERR_ARG;
  const int64_t* o = h->off;
  RC(hiopamd_vec_adjust_duals_plh(h->ctx, h->nx, it + o[8], it + o[4], h->ixl, mu, kappa_Sigma));
  RC(hiopamd_vec_adjust_duals_plh(h->ctx, h->nx, it + o[9], it + o[5], h->ixu, mu, kappa_Sigma));
  RC(hiopamd_vec_adjust_duals_plh(h->ctx, h->nd, it + o[10], it + o[6], h->idl, mu, kappa_Sigma));
  RC(hiopamd_vec_adjust_duals_plh(h->ctx, h->nd, it + o[11], it + o[7], h->idu, mu, kappa_Sigma));
  return HIOPAMD_OK;
}

// evalLogBarrier (:523-540) and linearDampingTerm (:552-566): x-sized parts all-reduced on a column partition
int hiopamd_iterate_eval_log_barrier(hiopamd_kkt_xycyd* h, const double* it, double* out_host)
{
  if(!h || !it || !out_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t* o = h->off;
  double a = 0, b = 0, c = 0, d = 0;
  {
    ReduceBatch rb(ctx);   // the four sums in one host round trip
    RC(hiopamd_vec_log_barrier(ctx, h->nx, it + o[4], h->ixl, &a));
    RC(hiopamd_vec_log_barrier(ctx, h->nx, it + o[5], h->ixu, &b));
    RC(hiopamd_vec_log_barrier(ctx, h->nd, it + o[6], h->idl, &c));
    RC(hiopamd_vec_log_barrier(ctx, h->nd, it + o[7], h->idu, &d));
    RC(rb.flush());
  }
  red4_t r{{0.0, 0.0, a + b, 0.0}};
  RC(xpart_allreduce(h, &r, 2));
  *out_host = r.v[2] + c + d;
  return HIOPAMD_OK;
}

int hiopamd_iterate_linear_damping_term(hiopamd_kkt_xycyd* h, const double* it, double mu, double kappa_d, double* out_host)
{
  if(!h || !it || !out_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t* o = h->off;
  double a = 0, b = 0, c = 0, d = 0;
  {
    ReduceBatch rb(ctx);   // the four sums in one host round trip
    RC(hiopamd_vec_linear_damping_term(ctx, h->nx, it + o[4], h->ixl, h->ixu, mu, kappa_d, &a));
    RC(hiopamd_vec_linear_damping_term(ctx, h->nx, it + o[5], h->ixu, h->ixl, mu, kappa_d, &b));
    RC(hiopamd_vec_linear_damping_term(ctx, h->nd, it + o[6], h->idl, h->idu, mu, kappa_d, &c));
    RC(hiopamd_vec_linear_damping_term(ctx, h->nd, it + o[7], h->idu, h->idl, mu, kappa_d, &d));
    RC(rb.flush());
  }
  red4_t r{{0.0, 0.0, a + b, 0.0}};
  RC(xpart_allreduce(h, &r, 2));
  *out_host = r.v[2] + c + d;
  return HIOPAMD_OK;
}

}  // extern "C"

extern "C" {

// hiopDualsLsqUpdateLinsysRedDense::do_lsq_update (src/Optimization/hiopDualsUpdater.cpp:239-330):
//   [ Jc Jc^T   Jc Jd^T     ] [yc]     [ Jc   0 ] [ grad_f - zl + zu ]
//   [ Jd Jc^T   Jd Jd^T + I ] [yd] = - [ Jd   I ] [     vl - vu      ]
// The three Grams are assembled on the device (MDS: sparse parts through the Schur plans with D = I + dense Grams;
// dense / low-rank: MFMA Gram kernel, all-reduced on a column partition), the SPD solve is hiopamd_posv_refine
// (the reference: DPOTRF/DPOTRS or MAGMA).  yc, yd of `iter` are overwritten.  *ok_host = 0 if M is not SPD.
int hiopamd_duals_lsq_update(hiopamd_kkt_xycyd* h, double* iter, const double* grad_f, int* ok_host)
{
  if(!h || !iter || !grad_f || !ok_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = h->ctx;
  const int64_t* o = h->off;
  const int me = h->nyc, mi = h->nyd, m = me + mi;
  const int64_t nx = h->nx;
  *ok_host = 1;
  if(m == 0) return HIOPAMD_OK;
  const size_t mm = (size_t)m * m;
  if(!h->lsq && hipMalloc((void**)&h->lsq, sizeof(double) * (4 * mm + 9 * (size_t)m + 8)) != hipSuccess) return HIOPAMD_ERR_HIP;
  double *M = h->lsq, *rhs = M + mm, *work = rhs + m;
  // ---- M
  if(h->kind == KIND_MDS) {
    RC(hiopamd_kkt_mds_jac_jac_trans(h->mds, M, m));
  } else {
    const double *Jc = h->Jc, *Jd = h->Jd;
    if(h->kind == KIND_LOWRANK && (!Jc || !Jd)) {   // the [Jc; Jd] copy of the last update
      Jc = hiopamd_kkt_lowrank_J(h->lr);
      Jd = Jc + (int64_t)me * nx;
    }
    if((me > 0 && !Jc) || (mi > 0 && !Jd)) return HIOPAMD_ERR_STATE;
    if(me > 0) RC(hiopamd_gram_weighted(ctx, me, me, nx, Jc, nx, Jc, nx, nullptr, 0.0, M, m, 1.0, 1));
    if(me > 0 && mi > 0) RC(hiopamd_gram_weighted(ctx, me, mi, nx, Jc, nx, Jd, nx, nullptr, 0.0, M + me, m, 1.0, 0));
    if(mi > 0) RC(hiopamd_gram_weighted(ctx, mi, mi, nx, Jd, nx, Jd, nx, nullptr, 0.0, M + (int64_t)me * m + me, m, 1.0, 1));
    if(h->kind == KIND_LOWRANK && ctx->allreduce &&
       ctx_allreduce(ctx, M, mm, HIOPAMD_SUM) != 0)
      return HIOPAMD_ERR_HIP;
  }
  RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, m, me, mi, 1.0));   // mixmi->addDiagonal(1.0)  (:256)
  // ---- rhs = -[Jc; Jd] vecx - [0; vecd],  vecx = grad_f - zl + zu,  vecd = vl - vu          (:285-297)
  {
    double* vecx = h->rx_tilde;
    const double *zl = iter + o[8], *zu = iter + o[9], *vl = iter + o[10], *vu = iter + o[11];
    RC(launch_ew(ctx, nx, [=] __device__(int64_t i) { vecx[i] = grad_f[i] - zl[i] + zu[i]; }));
    RC(backend_jac_times_vec(h, rhs, vecx));
    RC(launch_ew(ctx, m, [=] __device__(int64_t i) { rhs[i] = -rhs[i] - (i >= me ? (vl[i - me] - vu[i - me]) : 0.0); }));
  }
  int info = 0;
  double resid = 0.0;
  RC(hiopamd_posv_refine(ctx, m, M, m, rhs, work, &info, &resid));
  if(info != 0) {
    *ok_host = 0;
    return HIOPAMD_OK;
  }
  RC(hiopamd_vec_copy(ctx, me, iter + o[2], rhs));
  RC(hiopamd_vec_copy(ctx, mi, iter + o[3], rhs + me));
  return HIOPAMD_OK;
}

}  // extern "C"


// ------------------------------------------------------------------------------------------------------------------------
// The regularisation state machines on their own (host only, no device work): hiopPDPerturbation's public interface
// (hiopPDPerturbation.hpp:60-130) for callers that run the inertia-correction loop themselves, and for the CPU tests.
// ------------------------------------------------------------------------------------------------------------------------
struct hiopamd_pd_perturbation {
  PdPerturb pd;
};

int hiopamd_pd_perturbation_create(hiopamd_pd_perturbation** out, int kind)
{
  if(!out || kind < 0 || kind > 2) return HIOPAMD_ERR_ARG;
  *out = new(std::nothrow) hiopamd_pd_perturbation();
  if(!*out) return HIOPAMD_ERR_HIP;
  if(kind == 2) (*out)->pd.null_mode = true;
  else (*out)->pd.kind = kind;
  return HIOPAMD_OK;
}

int hiopamd_pd_perturbation_destroy(hiopamd_pd_perturbation* p)
{
  delete p;
  return HIOPAMD_OK;
}

int hiopamd_pd_perturbation_set_options(hiopamd_pd_perturbation* p, const double* o)
{
  if(!p || !o) return HIOPAMD_ERR_ARG;
  PdPerturb& q = p->pd;
  q.delta_w_min_bar = o[0];
  q.delta_w_max_bar = o[1];
  q.delta_w_0_bar = o[2];
  q.kappa_w_minus = o[3];
  q.kappa_w_plus_bar = o[4];
  q.kappa_w_plus = o[5];
  q.delta_c_bar = o[6];
  q.kappa_c = o[7];
  return HIOPAMD_OK;
}

int hiopamd_pd_perturbation_set_mu(hiopamd_pd_perturbation* p, double mu)
{
  if(!p) return HIOPAMD_ERR_ARG;
  p->pd.mu = mu;
  return HIOPAMD_OK;
}

int hiopamd_pd_perturbation_compute_initial_deltas(hiopamd_pd_perturbation* p, int* ok)
{
  if(!p || !ok) return HIOPAMD_ERR_ARG;
  *ok = p->pd.compute_initial_deltas() ? 1 : 0;
  return HIOPAMD_OK;
}

int hiopamd_pd_perturbation_compute_perturb_wrong_inertia(hiopamd_pd_perturbation* p, int* ok)
{
  if(!p || !ok) return HIOPAMD_ERR_ARG;
  *ok = p->pd.compute_perturb_wrong_inertia() ? 1 : 0;
  return HIOPAMD_OK;
}

int hiopamd_pd_perturbation_compute_perturb_singularity(hiopamd_pd_perturbation* p, int* ok)
{
  if(!p || !ok) return HIOPAMD_ERR_ARG;
  *ok = p->pd.compute_perturb_singularity() ? 1 : 0;
  return HIOPAMD_OK;
}

int hiopamd_pd_perturbation_get(const hiopamd_pd_perturbation* p, double* curr4, double* last4, int* state4)
{
  if(!p) return HIOPAMD_ERR_ARG;
  const PdPerturb& q = p->pd;
  if(curr4) { curr4[0] = q.wx; curr4[1] = q.wd; curr4[2] = q.cc; curr4[3] = q.cd; }
  if(last4) { last4[0] = q.wx_last; last4[1] = q.wd_last; last4[2] = q.cc_last; last4[3] = q.cd_last; }
  if(state4) { state4[0] = (int)q.hess_degenerate; state4[1] = (int)q.jac_degenerate; state4[2] = (int)q.test_type; state4[3] = q.dirty; }
  return HIOPAMD_OK;
}
