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
#include "sparse_plans.hpp"

#include <vector>

// a regularisation term: one value for every entry (v == nullptr) or a device vector (the reference's delta_wx / delta_wd /
// delta_cc / delta_cd are vectors: hiopKKTLinSysMDS.cpp:178-181, the randomised perturbations fill them entry by entry)
struct MdsDelta {
  const double* v;
  double s;
  __device__ __forceinline__ double at(int64_t i) const { return v ? v[i] : s; }
};

struct hiopamd_kkt_mds {
  hiopamd_ctx* ctx = nullptr;
  hiopamd_mds_structure s{};
  hiopamd_linsolver* ls = nullptr;
  hiopamd_sp_plan* plan_cc = nullptr;
  hiopamd_sp_plan* plan_dd = nullptr;
  hiopamd_sp_plan* plan_cd = nullptr;
  hiopamd_sp_tplan* tplan_c = nullptr;   // column-side plans of Jcs / Jds for the transposed products of every solve
  hiopamd_sp_tplan* tplan_d = nullptr;
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
  bool solve_failed = false;   // a solve since the last hiopamd_kkt_mds_solve_status is known to have failed
  MdsDelta last_delta[4] = {{nullptr, 0.0}, {nullptr, 0.0}, {nullptr, 0.0}, {nullptr, 0.0}};   // of the last build (re-assembly after a time-out)
};

namespace hiopamd {
// ---- solveCompressed around the dense solve, fused (hiopKKTLinSysMDS.cpp:318-401 as TWO launches + the products of the long rows)
// Before: rxs = rx_s / Hxs | dyc = ryc - Jcs rxs (also packed into rhs) | rhs[0 .. nxd) = rx_d            — one launch;
//         ryd -= Jds rxs, packed into rhs                                                                 — hiopamd_sp_times_vec_copy.
// After:  dx_d, dyc, dyd out of rhs | dx_s = (rx_s - Jcs^T dyc - Jds^T dyd) / Hxs through the column plans — one launch.
// Same operations on the same operands in the same order as the unfused sequence (which stays as the fall-back when a column of
// Jcs / Jds is long): results are bitwise identical.
__global__ __launch_bounds__(kBlock) void mds_pre_solve_kernel(int nxs, int nxd, int neq, int nnz_c, const int* __restrict__ ci,
                                                               const int* __restrict__ cj, const double* __restrict__ cv,
                                                               const double* __restrict__ rx, const double* __restrict__ Hxs,
                                                               const double* __restrict__ ryc, double* __restrict__ rxs,
                                                               double* __restrict__ dyc, double* __restrict__ rhs, int blocks_ew)
{
  if((int)blockIdx.x < blocks_ew) {   // element-wise part: rxs and the dense part of the right-hand side
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if(i < nxs) rxs[i] = rx[i] / Hxs[i];
    if(i < nxd) rhs[i] = rx[nxs + i];
    return;
  }
  // one wave per row of Jcs (the kernel of hiopamd_sp_times_vec with x = rx / Hxs formed on the fly)
  const int row = (int)((((int64_t)blockIdx.x - blocks_ew) * kBlock + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if(row >= neq) return;   // wave-uniform
  const int start = wave_lower_bound(ci, 0, nnz_c, row, lane);
  const int end = wave_lower_bound(ci, start, nnz_c, row + 1, lane);
  double acc = 0.0;
  for(int k = start + lane; k < end; k += 64) {
    const int c = cj[k];
    acc += (rx[c] / Hxs[c]) * cv[k];
  }
  for(int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if(lane == 0) {
    const double r = 1.0 * ryc[row] + (-1.0) * acc;
    dyc[row] = r;
    rhs[nxd + row] = r;
  }
}
__global__ __launch_bounds__(kBlock) void mds_post_solve_kernel(int nxs, int nxd, int neq, int nineq, const double* __restrict__ rhs,
                                                                const double* __restrict__ rx, const double* __restrict__ Hxs,
                                                                const int64_t* __restrict__ cptr_c, const int* __restrict__ perm_c,
                                                                const int* __restrict__ prow_c, const double* __restrict__ cv,
                                                                const int64_t* __restrict__ cptr_d, const int* __restrict__ perm_d,
                                                                const int* __restrict__ prow_d, const double* __restrict__ dv,
                                                                double* __restrict__ dx, double* __restrict__ dyc, double* __restrict__ dyd)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t m = (int64_t)nxd + neq + nineq;
  if(i < m) {
    const double v = rhs[i];
    if(i < nxd) dx[nxs + i] = v;
    else if(i < nxd + neq) dyc[i - nxd] = v;
    else dyd[i - nxd - neq] = v;
  }
  if(i < nxs) {
    const double* yc = rhs + nxd;          // (= dyc, dyd: read from the solution vector, not from what this launch writes)
    const double* yd = rhs + nxd + neq;
    double ac = 0.0, ad = 0.0;
    for(int64_t p = cptr_c[i]; p < cptr_c[i + 1]; ++p) ac += yc[prow_c[p]] * cv[perm_c[p]];
    for(int64_t p = cptr_d[i]; p < cptr_d[i + 1]; ++p) ad += yd[prow_d[p]] * dv[perm_d[p]];
    double t = 1.0 * rx[i] + (-1.0) * ac;
    t = 1.0 * t + (-1.0) * ad;
    dx[i] = t / Hxs[i];
  }
}
}  // namespace hiopamd

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
  if(rc == HIOPAMD_OK) rc = hiopamd_linsolver_set_retry_copy(k->ls, 0);   // (this object re-assembles after a time-out: hiopamd_kkt_mds_factorize)
  if(rc == HIOPAMD_OK) rc = hiopamd_sp_tplan_create(&k->tplan_c, st->neq, st->nxs, st->nnz_Jcs, st->Jcs_i_host, st->Jcs_j_host);
  if(rc == HIOPAMD_OK) rc = hiopamd_sp_tplan_create(&k->tplan_d, st->nineq, st->nxs, st->nnz_Jds, st->Jds_i_host, st->Jds_j_host);
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
  hiopamd_sp_tplan_destroy(k->tplan_c);
  hiopamd_sp_tplan_destroy(k->tplan_d);
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

// Dense part of build_kkt_matrix in ONE pass over the upper triangle (reference: setToZero + three block adds + four diagonal
// adds = eight passes over parts of an N x N matrix, hiopKKTLinSysMDS.cpp:196-215,245,289-290): every element (r, c), c >= r,
// is written exactly once —
//   (1,1)  upper(Hd)[r][c]  (+ Dxd + delta_wx on the diagonal)      (1,2)  Jcd[c - nxd][r]        (1,3)  Jdd[c - nxd - neq][r]
//   (2,2)  -delta_cc on the diagonal, 0 elsewhere        (2,3)  0        (3,3)  -(Dd_inv + delta_cd) on the diagonal, 0 elsewhere
// and the sparse Schur terms are added afterwards by their plans (a few entries).  64 x 64 tiles; a tile inside a transposed
// block goes through LDS (source read along its rows, matrix written along its rows), every other tile element by element.
// The strictly lower triangle is never touched (it is zero since the solver object was created and nothing reads it).
constexpr int MA_T = 64;
__global__ __launch_bounds__(hiopamd::kBlock) void kkt_mds_assemble_kernel(double* __restrict__ M, int64_t ld, int nxs, int nxd, int neq,
                                                                           int nineq, const double* __restrict__ Hdd,
                                                                           const double* __restrict__ Jcd,
                                                                           const double* __restrict__ Jdd,
                                                                           const double* __restrict__ Dx,
                                                                           const double* __restrict__ Dd_inv, MdsDelta dwx,
                                                                           MdsDelta dcc, MdsDelta dcd, int ntile)
{
  __shared__ double T[MA_T][MA_T + 1];
  // linear tile index -> (ti, tj), tj >= ti, row by row of the upper triangle
  int t = blockIdx.x, ti = 0;
  {
    // rows have ntile, ntile - 1, ... tiles: solve the quadratic, then fix up
    const double nt = (double)ntile;
    ti = (int)floor((2.0 * nt + 1.0 - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)t)) * 0.5);
    if(ti < 0) ti = 0;
    while(ti > 0 && (int64_t)ti * ntile - (int64_t)ti * (ti - 1) / 2 > t) --ti;
    while((int64_t)(ti + 1) * ntile - (int64_t)(ti + 1) * ti / 2 <= t) ++ti;
  }
  const int tj = ti + (t - (int)((int64_t)ti * ntile - (int64_t)ti * (ti - 1) / 2));
  const int N = nxd + neq + nineq;
  const int r0 = ti * MA_T, c0 = tj * MA_T;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int r1 = min(r0 + MA_T, N), c1 = min(c0 + MA_T, N);
  auto diag_term = [&](int r) -> double {
    if(r < nxd) return Dx[nxs + r] + dwx.at(nxs + r);
    if(r < nxd + neq) return -dcc.at(r - nxd);
    return -(Dd_inv[r - nxd - neq] + dcd.at(r - nxd - neq));
  };
  // tiles wholly inside (rows of x_dense) x (columns of one Jacobian): the transposed blocks
  const bool in_c = r1 <= nxd && c0 >= nxd && c1 <= nxd + neq;
  const bool in_d = r1 <= nxd && c0 >= nxd + neq;
  if(in_c || in_d) {
    const double* S = in_c ? Jcd + (int64_t)(c0 - nxd) * nxd : Jdd + (int64_t)(c0 - nxd - neq) * nxd;   // row cc of the source = column c0 + cc
    const int nr = r1 - r0, nc = c1 - c0;
#pragma unroll 4
    for(int cc = ty; cc < nc; cc += 4)
      if(tx < nr) T[tx][cc] = S[(int64_t)cc * nxd + r0 + tx];
    __syncthreads();
#pragma unroll 4
    for(int rr = ty; rr < nr; rr += 4)
      if(tx < nc) M[(int64_t)(r0 + rr) * ld + c0 + tx] = T[rr][tx];
    return;
  }
  for(int rr = ty; rr < r1 - r0; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    if(c >= c1 || c < r) continue;
    double v = 0.0;
    if(r < nxd) {
      if(c < nxd) v = Hdd[(int64_t)r * nxd + c];
      else if(c < nxd + neq) v = Jcd[(int64_t)(c - nxd) * nxd + r];
      else v = Jdd[(int64_t)(c - nxd - neq) * nxd + r];
    }
    if(r == c) v += diag_term(r);
    M[(int64_t)r * ld + c] = v;
  }
}

static int kkt_mds_build_impl(hiopamd_kkt_mds* k, MdsDelta dwx, MdsDelta dwd, MdsDelta dcc, MdsDelta dcd)
{
  if(!k || !k->Dx) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = k->ctx;
  const hiopamd_mds_structure& s = k->s;
  const int nxs = s.nxs, nxd = s.nxd, neq = s.neq, nineq = s.nineq;
  const int N = nxd + neq + nineq;
  // (an order the solver object works at a padded order is assembled straight into the padded copy, at its pitch)
  double* M = nullptr;
  int64_t ld = N;
  RC(hiopamd_linsolver_assembly_matrix(k->ls, &M, &ld));
  SpanScope span(ctx, HIOPAMD_SPAN_KKT_UPDATE_LINSYS);   // :193-294

  // Hxs = Dxs + delta_wx + diag(Hss)                                         (:223-231)
  {
    const double* Dx = k->Dx;
    double* Hxs = k->Hxs;
    RC(launch_ew(ctx, nxs, [=] __device__(int64_t i) { Hxs[i] = Dx[i] + dwx.at(i); }));
    RC(hiopamd_spsym_add_diag_to_vec(ctx, s.nnz_Hss, s.Hss_i, s.Hss_j, k->Hss_val, 1.0, Hxs, 0, nxs, 0, nxs));
  }
  // Dd_inv = 1/(Dd + delta_wd)                                                (:280-287)
  {
    const double* Dd = k->Dd;
    double* Ddi = k->Dd_inv;
    RC(launch_ew(ctx, nineq, [=] __device__(int64_t i) { Ddi[i] = 1.0 / (dwd.at(i) + Dd[i]); }));
  }
  // Msys = 0; (1,1) += upper(Hd) + Dxd + delta_wx; (1,2) += Jcd^T; (1,3) += Jdd^T; diag(2,2) -= delta_cc;
  // diag(3,3) -= Dd_inv + delta_cd — one pass over the upper triangle          (:196-215, :245, :289-290)
  if(N > 0) {
    const int ntile = (N + MA_T - 1) / MA_T;
    const int64_t ntri = (int64_t)ntile * (ntile + 1) / 2;
    hipLaunchKernelGGL(kkt_mds_assemble_kernel, dim3((unsigned)ntri), dim3(kBlock), 0, ctx->stream, M, ld, nxs, nxd, neq, nineq,
                       k->Hdd, k->Jcd, k->Jdd, k->Dx, k->Dd_inv, dwx, dcc, dcd, ntile);
    HIOPAMD_CHECK(hipGetLastError());
  }
  // (2,2) += -Jcs Hxs^-1 Jcs^T                                               (:239)
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_cc, k->Jcs_val, k->Jcs_val, k->Hxs, -1.0, M, ld, nxd, nxd));
  // (3,3) += -Jds Hxs^-1 Jds^T ; (2,3) += -Jcs Hxs^-1 Jds^T                  (:267-276)
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_dd, k->Jds_val, k->Jds_val, k->Hxs, -1.0, M, ld, nxd + neq, nxd + neq));
  RC(hiopamd_sp_add_MDinvNt(ctx, k->plan_cd, k->Jcs_val, k->Jds_val, k->Hxs, -1.0, M, ld, nxd, nxd + neq));
  k->last_delta[0] = dwx;
  k->last_delta[1] = dwd;
  k->last_delta[2] = dcc;
  k->last_delta[3] = dcd;
  k->built = true;
  return HIOPAMD_OK;
}

int hiopamd_kkt_mds_build(hiopamd_kkt_mds* k, double delta_wx, double delta_wd, double delta_cc, double delta_cd)
{
  return kkt_mds_build_impl(k, MdsDelta{nullptr, delta_wx}, MdsDelta{nullptr, delta_wd}, MdsDelta{nullptr, delta_cc},
                            MdsDelta{nullptr, delta_cd});
}

// the same with the regularisation as device VECTORS, the reference's actual form (delta_wx over the nxs + nxd primal variables
// in the solver's order sparse-then-dense, delta_wd and delta_cd over the inequalities, delta_cc over the equalities); a null
// pointer stands for a zero vector
int hiopamd_kkt_mds_build_vec(hiopamd_kkt_mds* k, const double* delta_wx, const double* delta_wd, const double* delta_cc,
                              const double* delta_cd)
{
  return kkt_mds_build_impl(k, MdsDelta{delta_wx, 0.0}, MdsDelta{delta_wd, 0.0}, MdsDelta{delta_cc, 0.0}, MdsDelta{delta_cd, 0.0});
}

static int kkt_mds_build_impl(hiopamd_kkt_mds* k, MdsDelta dwx, MdsDelta dwd, MdsDelta dcc, MdsDelta dcd);

int hiopamd_kkt_mds_factorize(hiopamd_kkt_mds* k, int* n_neg_host)
{
  if(!k || !n_neg_host) return HIOPAMD_ERR_ARG;
  if(!k->built) return HIOPAMD_ERR_STATE;
  SpanScope span(k->ctx, HIOPAMD_SPAN_KKT_UPDATE_INNER_FACT);   // hiopKKTLinSys.cpp:347-352
  int n_neg = 0;
  // Haynsworth inertia additivity: the negative / zero entries of the sparse (1,1) block are added to the dense block's inertia (:83-108).
  // Their two counting reductions are LAUNCHED here, in front of the factorisation, and read after its one synchronisation (a batch of
  // deferred reductions): behind it they cost two more launch + synchronise round trips, ~80 us of an otherwise idle device per step.
  int64_t nneg_xs = 0, nzero_xs = 0;
  ReduceBatch counts(k->ctx);
  RC(hiopamd_vec_num_elems_less_than(k->ctx, k->s.nxs, k->Hxs, -1e-14, &nneg_xs));
  RC(hiopamd_vec_num_elems_abs_less_than(k->ctx, k->s.nxs, k->Hxs, 1e-14, &nzero_xs));
  int rcm = hiopamd_linsolver_matrix_changed(k->ls, &n_neg);
  if(rcm == HIOPAMD_ERR_TIMEOUT) {
    // the dataflow factorisation gave up and left the matrix overwritten; the solver object has switched to the stepwise
    // kernels: assemble again and factor once more
    RC(kkt_mds_build_impl(k, k->last_delta[0], k->last_delta[1], k->last_delta[2], k->last_delta[3]));
    rcm = hiopamd_linsolver_matrix_changed(k->ls, &n_neg);
  }
  RC(rcm);
  RC(counts.flush());
  if(n_neg >= 0) {
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
  // the unfused sequence below is the fall-back when a column of Jcs / Jds holds more than 64 entries
  // ... and when Jcs is "few long rows" (hiopamd_sp_times_vec's split shape: nrows <= 256 && nnz > 64 nrows), where the fused kernel's
  // one-wave-per-row product would neither be bitwise equal to the unfused call nor have its parallelism
  const bool jcs_split_shape = neq <= 256 && (int64_t)s.nnz_Jcs > 64 * (int64_t)neq;
  const bool fused = !jcs_split_shape && k->tplan_c->max_len <= 64 && k->tplan_d->max_len <= 64;
  if(fused) {
    const int64_t new_ = (nxs > nxd) ? nxs : nxd;
    const int blocks_ew = (int)((new_ + kBlock - 1) / kBlock);
    const int blocks_rows = (int)(((int64_t)neq * 64 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(mds_pre_solve_kernel, dim3((unsigned)(blocks_ew + blocks_rows)), dim3(kBlock), 0, ctx->stream, nxs, nxd, neq,
                       s.nnz_Jcs, s.Jcs_i, s.Jcs_j, k->Jcs_val, rx, Hxs, ryc, rxs, dyc, rhs, blocks_ew);
    HIOPAMD_CHECK(hipGetLastError());
    // ryd -= Jds rxs, stored into the right-hand side as well                 (:345-347, :355-357)
    RC(hiopamd_sp_times_vec_copy(ctx, nineq, nxs, s.nnz_Jds, s.Jds_i, s.Jds_j, k->Jds_val, 1.0, ryd, -1.0, rxs, rhs + nxd + neq));
  } else {
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
  }
  // solve                                                                    (:364-368)
  span_end(ctx, HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);
  span_begin(ctx, guard.id = HIOPAMD_SPAN_KKT_SOLVE_INNER);
  RC(hiopamd_linsolver_solve(k->ls, rhs, 1));
  {
    int okc = 1;
    RC(hiopamd_linsolver_last_solve_ok(k->ls, &okc));   // no synchronisation: what the host knows already
    if(!okc) k->solve_failed = true;
  }
  span_end(ctx, HIOPAMD_SPAN_KKT_SOLVE_INNER);
  span_begin(ctx, guard.id = HIOPAMD_SPAN_KKT_SOLVE_RHS_MANIP);   // :380-401
  if(fused) {
    const int64_t m = (int64_t)nxd + neq + nineq;
    const int64_t nmax = (m > nxs) ? m : nxs;
    hipLaunchKernelGGL(mds_post_solve_kernel, dim3((unsigned)((nmax + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, nxs, nxd, neq,
                       nineq, rhs, rx, Hxs, k->tplan_c->cptr, k->tplan_c->perm, k->tplan_c->prow, k->Jcs_val, k->tplan_d->cptr,
                       k->tplan_d->perm, k->tplan_d->prow, k->Jds_val, dx, dyc, dyd);
    HIOPAMD_CHECK(hipGetLastError());
    return HIOPAMD_OK;
  }
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
  RC(hiopamd_sp_tplan_trans_times_vec(ctx, k->tplan_c, k->Jcs_val, 1.0, dxs, -1.0, dyc));
  RC(hiopamd_sp_tplan_trans_times_vec(ctx, k->tplan_d, k->Jds_val, 1.0, dxs, -1.0, dyd));
  RC(launch_ew(ctx, nxs, [=] __device__(int64_t i) { dx[i] = dxs[i] / Hxs[i]; }));
  return HIOPAMD_OK;
}

// did every solveCompressed since the last call deliver a valid direction?  *ok_host = 0 after a safe-mode refinement that
// did not converge or a timed-out dataflow solve.  sync != 0 also synchronises the stream and looks at the dataflow solve's
// error word (hiopamd_linsolver_solve_status); sync == 0 reports what the host knows without waiting.
int hiopamd_kkt_mds_solve_status(hiopamd_kkt_mds* k, int sync, int* ok_host)
{
  if(!k || !ok_host) return HIOPAMD_ERR_ARG;
  int ok = 1;
  if(sync) RC(hiopamd_linsolver_solve_status(k->ls, &ok));
  else RC(hiopamd_linsolver_last_solve_ok(k->ls, &ok));
  if(k->solve_failed) ok = 0;
  k->solve_failed = false;
  *ok_host = ok;
  return HIOPAMD_OK;
}

// safe mode of the condensed system: the linear solver regularises statically and refines (the role of the reference's
// switch to MagmaBuKa, hiopKKTLinSysMDS.cpp:408-430); the positive block is the dense-x part
int hiopamd_kkt_mds_set_safe_mode(hiopamd_kkt_mds* k, int enable)
{
  if(!k) return HIOPAMD_ERR_ARG;
  if(enable == 2) {
    const int rc = hiopamd_linsolver_set_safe_mode(k->ls, 0, k->s.nxd);
    return rc != HIOPAMD_OK ? rc : hiopamd_linsolver_set_pivoting(k->ls, 1);
  }
  const int rc = hiopamd_linsolver_set_pivoting(k->ls, 0);
  return rc != HIOPAMD_OK ? rc : hiopamd_linsolver_set_safe_mode(k->ls, enable, k->s.nxd);
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
    RC(hiopamd_sp_tplan_trans_times_vec(k->ctx, k->tplan_c, k->Jcs_val, beta, y, alpha, x));
    RC(hiopamd_mat_trans_times_vec(k->ctx, s.neq, s.nxd, k->Jcd, s.nxd, beta, y + s.nxs, alpha, x));
  } else {
    RC(hiopamd_sp_tplan_trans_times_vec(k->ctx, k->tplan_d, k->Jds_val, beta, y, alpha, x));
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
// (the N x N view, pitch N: brought up to date first when the object assembles into the solver's padded copy; asynchronous on the
//  context's stream, like the assembly itself)
double* hiopamd_kkt_mds_sys_matrix(hiopamd_kkt_mds* k)
{
  if(!k) return nullptr;
  (void)hiopamd_linsolver_sys_matrix_sync(k->ls);
  return hiopamd_linsolver_sys_matrix(k->ls);
}
double* hiopamd_kkt_mds_Hxs(hiopamd_kkt_mds* k) { return k ? k->Hxs : nullptr; }
hiopamd_linsolver* hiopamd_kkt_mds_linsolver(hiopamd_kkt_mds* k) { return k ? k->ls : nullptr; }

}  // extern "C"
