// hiopKKTLinSysCompressedMDSXYcYd on MI355X: condensed mixed-dense-sparse KKT
//   assemble (build_kkt_matrix) -> factor + inertia (factorizeWithCurvCheck) -> solveCompressed.
//
// reference: src/Optimization/hiopKKTLinSysMDS.cpp:78-110 (factorizeWithCurvCheck),
// :172-305 (build_kkt_matrix), :307-403 (solveCompressed); formulas in
// src/Optimization/hiopKKTLinSysMDS.hpp:59-95.  The condensed matrix (upper triangle, row-major) is
//   [ Hd+Dxd+dwx*I      Jcd^T                         Jdd^T                               ]
//   [                  -Jcs Hxs^-1 Jcs^T - dcc*I     -Jcs Hxs^-1 Jds^T                    ]
//   [                                                -Jds Hxs^-1 Jds^T - (Dd+dwd)^-1 - dcd*I ]
// with Hxs = diag(Hs) + Dxs + dwx*I.  Everything stays in HBM: the N x N matrix is owned by the
// linear-solver object (reference: hiopLinSolverSymDense::sysMatrix, hiopLinSolver.hpp:117-130) and
// is assembled in place, factored in place and never crosses PCIe (the reference's MAGMA path copies
// it H2D at every factorisation, hiopLinSolverSymDenseMagma.cpp:337-340).
#include "device_utils.hpp"

#include <vector>

struct hiopamd_kkt_mds {
  hiopamd_ctx* ctx = nullptr;
  hiopamd_mds_structure s{};
  hiopamd_linsolver* ls = nullptr;
  hiopamd_sp_plan* plan_cc = nullptr;
  hiopamd_sp_plan* plan_dd = nullptr;
  hiopamd_sp_plan* plan_cd = nullptr;
  // current values (borrowed device pointers)
  const double *Jcs_val = nullptr, *Jds_val = nullptr, *Hss_val = nullptr;
  const double *Jcd = nullptr, *Jdd = nullptr, *Hdd = nullptr, *Dx = nullptr, *Dd = nullptr;
  // owned device buffers
  double* Hxs = nullptr;      // nxs
  double* Dd_inv = nullptr;   // nineq
  double* rhs = nullptr;      // N
  double* buf_xs = nullptr;   // nxs
  double* ones_xs = nullptr;  // nxs, all ones (D = I for J J^T through the Schur plans)
  bool built = false;
};

using namespace hiopamd;

extern "C" {

int hiopamd_kkt_mds_create(hiopamd_kkt_mds** out, hiopamd_ctx* ctx, const hiopamd_mds_structure* st)
{
  if(!out || !ctx || !st) return HIOPAMD_ERR_ARG;
  if(st->nxs < 0 || st->nxd < 0 || st->neq < 0 || st->nineq < 0) return HIOPAMD_ERR_ARG;
  hiopamd_kkt_mds* k = new hiopamd_kkt_mds();
  k->ctx = ctx;
  k->s = *st;
  const int N = st->nxd + st->neq + st->nineq;
  int rc = hiopamd_linsolver_create(&k->ls, ctx, N);
  // symbolic plans of the three Schur blocks (pattern is fixed over the IPM iterations)
  if(rc == HIOPAMD_OK)
    rc = hiopamd_sp_plan_create(&k->plan_cc, st->neq, st->neq, st->nxs, st->nnz_Jcs, st->Jcs_i_host, st->Jcs_j_host,
                                st->nnz_Jcs, st->Jcs_i_host, st->Jcs_j_host, 1);
  if(rc == HIOPAMD_OK)
    rc = hiopamd_sp_plan_create(&k->plan_dd, st->nineq, st->nineq, st->nxs, st->nnz_Jds, st->Jds_i_host, st->Jds_j_host,
                                st->nnz_Jds, st->Jds_i_host, st->Jds_j_host, 1);
  if(rc == HIOPAMD_OK)
    rc = hiopamd_sp_plan_create(&k->plan_cd, st->neq, st->nineq, st->nxs, st->nnz_Jcs, st->Jcs_i_host, st->Jcs_j_host,
                                st->nnz_Jds, st->Jds_i_host, st->Jds_j_host, 0);
  auto dalloc = [](double** p, size_t n) { return hipMalloc((void**)p, sizeof(double) * (n ? n : 1)); };
  if(rc == HIOPAMD_OK) {
    if(dalloc(&k->Hxs, st->nxs) != hipSuccess || dalloc(&k->Dd_inv, st->nineq) != hipSuccess ||
       dalloc(&k->rhs, N) != hipSuccess || dalloc(&k->buf_xs, st->nxs) != hipSuccess ||
       dalloc(&k->ones_xs, st->nxs) != hipSuccess)
      rc = HIOPAMD_ERR_HIP;
    if(rc == HIOPAMD_OK) rc = hiopamd_vec_set_to_constant(ctx, st->nxs, k->ones_xs, 1.0);
  }
  if(rc != HIOPAMD_OK) {
    hiopamd_kkt_mds_destroy(k);
    return rc;
  }
  *out = k;
  return HIOPAMD_OK;
}

int hiopamd_kkt_mds_destroy(hiopamd_kkt_mds* k)
{
  if(!k) return HIOPAMD_OK;
  if(k->ls) hiopamd_linsolver_destroy(k->ls);
  hiopamd_sp_plan_destroy(k->plan_cc);
  hiopamd_sp_plan_destroy(k->plan_dd);
  hiopamd_sp_plan_destroy(k->plan_cd);
  (void)hipFree(k->Hxs);
  (void)hipFree(k->Dd_inv);
  (void)hipFree(k->rhs);
  (void)hipFree(k->buf_xs);
  (void)hipFree(k->ones_xs);
  delete k;
  return HIOPAMD_OK;
}

int hiopamd_kkt_mds_set_values(hiopamd_kkt_mds* k, const double* Jcs_val, const double* Jds_val, const double* Hss_val,
                               const double* Jcd, const double* Jdd, const double* Hdd, const double* Dx,
                               const double* Dd)
{
  if(!k) return HIOPAMD_ERR_ARG;
  k->Jcs_val = Jcs_val;
  k->Jds_val = Jds_val;
  k->Hss_val = Hss_val;
  k->Jcd = Jcd;
  k->Jdd = Jdd;
  k->Hdd = Hdd;
  k->Dx = Dx;
  k->Dd = Dd;
  k->built = false;
  return HIOPAMD_OK;
}

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

int hiopamd_kkt_mds_build(hiopamd_kkt_mds* k, double delta_wx, double delta_wd, double delta_cc, double delta_cd)
{
  if(!k || !k->Dx) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = k->ctx;
  const hiopamd_mds_structure& s = k->s;
  const int nxs = s.nxs, nxd = s.nxd, neq = s.neq, nineq = s.nineq;
  const int N = nxd + neq + nineq;
  double* M = hiopamd_linsolver_sys_matrix(k->ls);
  const int64_t ld = N;
  SpanScope span(ctx, HIOPAMD_SPAN_KKT_UPDATE_LINSYS);   // :193-294

  // Msys.setToZero()                                                        (:196)
  HIOPAMD_CHECK(hipMemsetAsync(M, 0, sizeof(double) * (size_t)N * (size_t)N, ctx->stream));
  // (1,1) += upper(Hd); (1,2) += Jcd^T; (1,3) += Jdd^T                       (:204-206)
  RC(hiopamd_mat_add_upper_to_sym_upper(ctx, nxd, k->Hdd, nxd, 0, 1.0, M, ld));
  RC(hiopamd_mat_trans_add_to_sym_upper(ctx, neq, nxd, k->Jcd, nxd, 0, nxd, 1.0, M, ld));
  RC(hiopamd_mat_trans_add_to_sym_upper(ctx, nineq, nxd, k->Jdd, nxd, 0, nxd + neq, 1.0, M, ld));
  // diag(1,1) += Dxd + delta_wx                                              (:213-215)
  RC(hiopamd_mat_add_sub_diagonal(ctx, M, ld, 0, 1.0, k->Dx, nxs, nxd));
  RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, ld, 0, nxd, delta_wx));
  // Hxs = Dxs + delta_wx + diag(Hss)                                         (:223-231)
  {
    const double* Dx = k->Dx;
    double* Hxs = k->Hxs;
    RC(launch_ew(ctx, nxs, [=] __device__(int64_t i) { Hxs[i] = Dx[i] + delta_wx; }));
    RC(hiopamd_spsym_add_diag_to_vec(ctx, s.nnz_Hss, s.Hss_i, s.Hss_j, k->Hss_val, 1.0, Hxs, 0, nxs, 0, nxs));
  }
  // (2,2) += -Jcs Hxs^-1 Jcs^T ; -delta_cc                                   (:239-245)
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_cc, k->Jcs_val, k->Jcs_val, k->Hxs, -1.0, M, ld, nxd, nxd));
  RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, ld, nxd, neq, -delta_cc));
  // (3,3) += -Jds Hxs^-1 Jds^T ; (2,3) += -Jcs Hxs^-1 Jds^T                  (:267-276)
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_dd, k->Jds_val, k->Jds_val, k->Hxs, -1.0, M, ld, nxd + neq, nxd + neq));
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_cd, k->Jcs_val, k->Jds_val, k->Hxs, -1.0, M, ld, nxd, nxd + neq));
  // Dd_inv = 1/(Dd + delta_wd); (3,3) -= Dd_inv + delta_cd                   (:280-290)
  {
    const double* Dd = k->Dd;
    double* Ddi = k->Dd_inv;
    RC(launch_ew(ctx, nineq, [=] __device__(int64_t i) { Ddi[i] = 1.0 / (delta_wd + Dd[i]); }));
  }
  RC(hiopamd_mat_add_sub_diagonal(ctx, M, ld, nxd + neq, -1.0, k->Dd_inv, 0, nineq));
  RC(hiopamd_mat_add_sub_diagonal_const(ctx, M, ld, nxd + neq, nineq, -delta_cd));
  k->built = true;
  return HIOPAMD_OK;
}

int hiopamd_kkt_mds_factorize(hiopamd_kkt_mds* k, int* n_neg_host)
{
  if(!k || !n_neg_host) return HIOPAMD_ERR_ARG;
  if(!k->built) return HIOPAMD_ERR_STATE;
  SpanScope span(k->ctx, HIOPAMD_SPAN_KKT_UPDATE_INNER_FACT);   // hiopKKTLinSys.cpp:347-352
  int n_neg = 0;
  RC(hiopamd_linsolver_matrix_changed(k->ls, &n_neg));
  if(n_neg >= 0) {
    // Haynsworth inertia additivity: add the negative entries of the sparse (1,1) block   (:83-108)
    int64_t nneg_xs = 0, nzero_xs = 0;
    RC(hiopamd_vec_num_elems_less_than(k->ctx, k->s.nxs, k->Hxs, -1e-14, &nneg_xs));
    RC(hiopamd_vec_num_elems_abs_less_than(k->ctx, k->s.nxs, k->Hxs, 1e-14, &nzero_xs));
    if(nzero_xs > 0) n_neg = -1;
    else n_neg += (int)nneg_xs;
  }
  *n_neg_host = n_neg;
  return HIOPAMD_OK;
}

int hiopamd_kkt_mds_solve_compressed(hiopamd_kkt_mds* k, const double* rx, const double* ryc, double* ryd, double* dx,
                                     double* dyc, double* dyd)
{
  if(!k) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = k->ctx;
  const hiopamd_mds_structure& s = k->s;
  const int nxs = s.nxs, nxd = s.nxd, neq = s.neq, nineq = s.nineq;
  double* rxs = k->buf_xs;
  const double* Hxs = k->Hxs;
  double* rhs = k->rhs;
  span_begin(ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);   // :318-363
  struct SpanGuard {   // closes whatever span is open on an early (error) return
    hiopamd_ctx* c;
    int id;
    ~SpanGuard() { if(id >= 0) span_end(c, id); }
  } guard{ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP};
  // rxs = Hxs^-1 rx_sparse; dyc = ryc  (one pass)                            (:337-338, :343)
  {
    const int64_t nmax = (nxs > neq) ? nxs : neq;
    RC(launch_ew(ctx, nmax, [=] __device__(int64_t i) {
      if(i < nxs) rxs[i] = rx[i] / Hxs[i];
      if(i < neq) dyc[i] = ryc[i];
    }));
  }
  // dyc -= Jcs rxs ; ryd -= Jds rxs                                          (:343-347)
  RC(hiopamd_sp_times_vec(ctx, neq, nxs, s.nnz_Jcs, s.Jcs_i, s.Jcs_j, k->Jcs_val, 1.0, dyc, -1.0, rxs));
  RC(hiopamd_sp_times_vec(ctx, nineq, nxs, s.nnz_Jds, s.Jds_i, s.Jds_j, k->Jds_val, 1.0, ryd, -1.0, rxs));
  // rhs = [rx_dense; dyc; ryd]  (one pass instead of three copies)            (:353-357)
  {
    const double* rxd = rx + nxs;
    RC(launch_ew(ctx, (int64_t)nxd + neq + nineq, [=] __device__(int64_t i) {
      rhs[i] = (i < nxd) ? rxd[i] : ((i < nxd + neq) ? dyc[i - nxd] : ryd[i - nxd - neq]);
    }));
  }
  // solve                                                                    (:364-368)
  span_end(ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);
  span_begin(ctx, guard.id = HIOPAMD_SPAN_KKT_SOLVE_INNER);
  RC(hiopamd_linsolver_solve(k->ls, rhs, 1));
  span_end(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);
  span_begin(ctx, guard.id = HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);   // :380-401
  // unpack dx_dense, dyc, dyd and start dxs = rx_sparse  (one pass)           (:383-390)
  double* dxs = k->buf_xs;
  {
    double* dxd = dx + nxs;
    const int64_t m = (int64_t)nxd + neq + nineq;
    const int64_t nmax = (m > nxs) ? m : nxs;
    RC(launch_ew(ctx, nmax, [=] __device__(int64_t i) {
      if(i < m) {
        const double v = rhs[i];
        if(i < nxd) dxd[i] = v;
        else if(i < nxd + neq) dyc[i - nxd] = v;
        else dyd[i - nxd - neq] = v;
      }
      if(i < nxs) dxs[i] = rx[i];
    }));
  }
  // dxs = Hxs^-1 (rxs - Jcs^T dyc - Jds^T dyd)                               (:390-395)
  RC(hiopamd_sp_trans_times_vec(ctx, neq, nxs, s.nnz_Jcs, s.Jcs_i, s.Jcs_j, k->Jcs_val, 1.0, dxs, -1.0, dyc));
  RC(hiopamd_sp_trans_times_vec(ctx, nineq, nxs, s.nnz_Jds, s.Jds_i, s.Jds_j, k->Jds_val, 1.0, dxs, -1.0, dyd));
  RC(launch_ew(ctx, nxs, [=] __device__(int64_t i) { dx[i] = dxs[i] / Hxs[i]; }));
  return HIOPAMD_OK;
}

// safe mode of the condensed system: the linear solver regularises statically and refines (the role of the reference's
// switch to MagmaBuKa, hiopKKTLinSysMDS.cpp:408-430); the positive block is the dense-x part
int hiopamd_kkt_mds_set_safe_mode(hiopamd_kkt_mds* k, int enable)
{
  if(!k) return HIOPAMD_ERR_ARG;
  return hiopamd_linsolver_set_safe_mode(k->ls, enable, k->s.nxd);
}

int hiopamd_kkt_mds_set_diagonals(hiopamd_kkt_mds* k, const double* Dx, const double* Dd)
{
  if(!k) return HIOPAMD_ERR_ARG;
  k->Dx = Dx;
  k->Dd = Dd;
  k->built = false;
  return HIOPAMD_OK;
}

int hiopamd_kkt_mds_dims(const hiopamd_kkt_mds* k, int* dims4_host)
{
  if(!k || !dims4_host) return HIOPAMD_ERR_ARG;
  dims4_host[0] = k->s.nxs;
  dims4_host[1] = k->s.nxd;
  dims4_host[2] = k->s.neq;
  dims4_host[3] = k->s.nineq;
  return HIOPAMD_OK;
}

// y = beta*y + alpha*Hess*x, Hess = blockdiag(Hss (sym sparse), Hdd (dense))   (hiopMatrixMDS.hpp:310-318)
int hiopamd_kkt_mds_hess_times_vec(hiopamd_kkt_mds* k, double beta, double* y, double alpha, const double* x)
{
  if(!k || !k->Hdd) return HIOPAMD_ERR_STATE;
  const hiopamd_mds_structure& s = k->s;
  RC(hiopamd_spsym_times_vec(k->ctx, s.nxs, s.nnz_Hss, s.Hss_i, s.Hss_j, k->Hss_val, beta, y, alpha, x));
  RC(hiopamd_mat_times_vec(k->ctx, s.nxd, s.nxd, k->Hdd, s.nxd, beta, y + s.nxs, alpha, x + s.nxs));
  return HIOPAMD_OK;
}

// y = beta*y + alpha*[Js Jd]*x                                                 (hiopMatrixMDS.hpp:68-74)
int hiopamd_kkt_mds_jac_times_vec(hiopamd_kkt_mds* k, int which, double beta, double* y, double alpha, const double* x)
{
  if(!k || !k->Jcd) return HIOPAMD_ERR_STATE;
  const hiopamd_mds_structure& s = k->s;
  if(which == 0) {
    RC(hiopamd_sp_times_vec(k->ctx, s.neq, s.nxs, s.nnz_Jcs, s.Jcs_i, s.Jcs_j, k->Jcs_val, beta, y, alpha, x));
    RC(hiopamd_mat_times_vec(k->ctx, s.neq, s.nxd, k->Jcd, s.nxd, 1.0, y, alpha, x + s.nxs));
  } else {
    RC(hiopamd_sp_times_vec(k->ctx, s.nineq, s.nxs, s.nnz_Jds, s.Jds_i, s.Jds_j, k->Jds_val, beta, y, alpha, x));
    RC(hiopamd_mat_times_vec(k->ctx, s.nineq, s.nxd, k->Jdd, s.nxd, 1.0, y, alpha, x + s.nxs));
  }
  return HIOPAMD_OK;
}

// y = beta*y + alpha*[Js Jd]^T*x                                               (hiopMatrixMDS.hpp:75-81)
int hiopamd_kkt_mds_jac_trans_times_vec(hiopamd_kkt_mds* k, int which, double beta, double* y, double alpha,
                                        const double* x)
{
  if(!k || !k->Jcd) return HIOPAMD_ERR_STATE;
  const hiopamd_mds_structure& s = k->s;
  if(which == 0) {
    RC(hiopamd_sp_trans_times_vec(k->ctx, s.neq, s.nxs, s.nnz_Jcs, s.Jcs_i, s.Jcs_j, k->Jcs_val, beta, y, alpha, x));
    RC(hiopamd_mat_trans_times_vec(k->ctx, s.neq, s.nxd, k->Jcd, s.nxd, beta, y + s.nxs, alpha, x));
  } else {
    RC(hiopamd_sp_trans_times_vec(k->ctx, s.nineq, s.nxs, s.nnz_Jds, s.Jds_i, s.Jds_j, k->Jds_val, beta, y, alpha, x));
    RC(hiopamd_mat_trans_times_vec(k->ctx, s.nineq, s.nxd, k->Jdd, s.nxd, beta, y + s.nxs, alpha, x));
  }
  return HIOPAMD_OK;
}

// W (m x m, m = neq + nineq, upper triangle) = [Jc; Jd] [Jc; Jd]^T for the MDS Jacobians: sparse parts through the Schur
// plans with D = I, dense parts as Grams (hiopMatrixMDS::timesMatTrans, hiopMatrixMDS.hpp:94-100)
int hiopamd_kkt_mds_jac_jac_trans(hiopamd_kkt_mds* k, double* W, int64_t ldw)
{
  if(!k || !W || !k->Jcd) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = k->ctx;
  const hiopamd_mds_structure& s = k->s;
  const int m = s.neq + s.nineq;
  if(ldw < m) return HIOPAMD_ERR_ARG;
  HIOPAMD_CHECK(hipMemset2DAsync(W, sizeof(double) * (size_t)ldw, 0, sizeof(double) * (size_t)m, (size_t)m, ctx->stream));
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_cc, k->Jcs_val, k->Jcs_val, k->ones_xs, 1.0, W, ldw, 0, 0));
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_dd, k->Jds_val, k->Jds_val, k->ones_xs, 1.0, W, ldw, s.neq, s.neq));
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_cd, k->Jcs_val, k->Jds_val, k->ones_xs, 1.0, W, ldw, 0, s.neq));
  if(s.nxd > 0) {
    RC(hiopamd_gram_weighted(ctx, s.neq, s.neq, s.nxd, k->Jcd, s.nxd, k->Jcd, s.nxd, nullptr, 1.0, W, ldw, 1.0, 1));
    RC(hiopamd_gram_weighted(ctx, s.neq, s.nineq, s.nxd, k->Jcd, s.nxd, k->Jdd, s.nxd, nullptr, 1.0, W + s.neq, ldw, 1.0, 0));
    RC(hiopamd_gram_weighted(ctx, s.nineq, s.nineq, s.nxd, k->Jdd, s.nxd, k->Jdd, s.nxd, nullptr, 1.0,
                             W + (int64_t)s.neq * ldw + s.neq, ldw, 1.0, 1));
  }
  return HIOPAMD_OK;
}

double* hiopamd_kkt_mds_Dd_inv(hiopamd_kkt_mds* k) { return k ? k->Dd_inv : nullptr; }
double* hiopamd_kkt_mds_sys_matrix(hiopamd_kkt_mds* k) { return k ? hiopamd_linsolver_sys_matrix(k->ls) : nullptr; }
double* hiopamd_kkt_mds_Hxs(hiopamd_kkt_mds* k) { return k ? k->Hxs : nullptr; }
hiopamd_linsolver* hiopamd_kkt_mds_linsolver(hiopamd_kkt_mds* k) { return k ? k->ls : nullptr; }

}  // extern "C"
