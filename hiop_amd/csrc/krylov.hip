// Krylov solvers of the reference on device vectors: hiopPCGSolver::solve (src/LinAlg/hiopKrylovSolver.cpp:152-373) and
// hiopBiCGStabSolver::solve (:397-700) behind the interface of hiopKrylovSolver (hiopKrylovSolver.hpp:80-156): linear
// operators for the matrix and the left/right preconditioners, set_tol / set_max_num_iter / set_x0, solve(b) in place,
// get_sol_* afterwards.  The operators are callbacks on device pointers (the C-ABI form of hiopLinearOperator::times_vec).
// Same control flow, same flags (0 converged, 1 maxit, 3 stagnation / tolerance too small, 4 breakdown), same
// half-iteration counting for BiCGStab, same "return the minimal-residual iterate" fallback, same quirk that the start
// vector x0 is the solver's own persistent buffer and ends up holding the last iterate.
// Vector work goes through the a1/a2 entry points (one fused kernel / one reduction each); every dot product and norm is
// a host round trip, exactly as in the reference.
#include "common.hpp"

#include <cmath>
#include <limits>

using namespace hiopamd;

struct hiopamd_krylov {
  hiopamd_ctx* ctx = nullptr;
  int kind = 0;   // 0 PCG, 1 BiCGStab
  int64_t n = 0;
  hiopamd_linop_fn A = nullptr, ML = nullptr, MR = nullptr;
  void *Au = nullptr, *MLu = nullptr, *MRu = nullptr;
  double tol = 1e-9;   // hiopKrylovSolver.cpp:81-82
  int maxit = 8;
  double iter = -1.0, abs_resid = -1.0, rel_resid = -1.0;
  int flag = -1;
  double* w[9] = {nullptr};   // x0 + 8 work vectors
  // BiCGStab, 'tol is too small' exit (moresteps >= maxmsteps): the reference executes b->copyFrom(*xk_) before it breaks
  // (hiopKrylovSolver.cpp:561-566, :639-644), so its closing comparison of the minimal-residual iterate (:671-688) is made against an
  // OVERWRITTEN right-hand side and in practice hands back the last iterate.  1 (default): exactly that; 0: the comparison against the
  // original right-hand side (hiopamd_krylov_set_exit_mode).  PCG does not overwrite (:300-306) and is not affected.
  int ref_exit = 1;
};

namespace {

struct Vec {
  hiopamd_ctx* c;
  int64_t n;
  int rc = HIOPAMD_OK;
  void chk(int r) { if(rc == HIOPAMD_OK && r != HIOPAMD_OK) rc = r; }
  double dot(const double* x, const double* y) { double v = 0.0; hiopamd::ReduceNow now(c); chk(hiopamd_vec_dot(c, n, x, y, &v)); return v; }
  double nrm(const double* x) { double v = 0.0; hiopamd::ReduceNow now(c); chk(hiopamd_vec_twonorm(c, n, x, &v)); return v; }
  void copy(double* y, const double* x) { chk(hiopamd_vec_copy(c, n, y, x)); }
  void axpy(double* y, double a, const double* x) { chk(hiopamd_vec_axpy(c, n, y, a, x)); }
  void scale(double* y, double a) { chk(hiopamd_vec_scale(c, n, y, a)); }
  void apply(hiopamd_linop_fn f, void* u, double* y, const double* x) { chk(f(u, x, y)); }
  // res = b - A x   (times_vec, axpy(-1, b), scale(-1))
  void resid(hiopamd_krylov* k, double* res, const double* x, const double* b)
  {
    apply(k->A, k->Au, res, x);
    axpy(res, -1.0, b);
    scale(res, -1.0);
  }
};

bool pcg_solve(hiopamd_krylov* k, double* b)
{
  Vec V{k->ctx, k->n};
  double *xk = k->w[0], *xmin = k->w[1], *res = k->w[2], *yk = k->w[3], *zk = k->w[4], *pk = k->w[5], *qk = k->w[6];
  const double n2b = V.nrm(b);
  if(n2b == 0.0) {   // :154-159
    k->flag = 0;
    k->iter = 0.0;
    return true;
  }
  k->flag = 1;
  int64_t imin = 0;
  const double tolb = k->tol * n2b;
  V.copy(xmin, xk);
  V.resid(k, res, xk, b);
  double normr = V.nrm(res);
  k->abs_resid = normr;
  if(normr <= tolb) {   // :195-201
    V.copy(b, xk);
    k->flag = 0;
    k->iter = 0.0;
    k->rel_resid = normr / n2b;
    return true;
  }
  double normrmin = normr, rho = 1.0;
  int stagsteps = 0, moresteps = 0;
  const double eps = std::numeric_limits<double>::epsilon();
  const int maxmsteps = 100, maxstagsteps = 3;
  double alpha, rho1, pq;
  int ii = 0;
  for(; ii < k->maxit; ++ii) {
    if(k->ML) V.apply(k->ML, k->MLu, yk, res); else V.copy(yk, res);
    if(k->MR) V.apply(k->MR, k->MRu, zk, yk); else V.copy(zk, yk);
    rho1 = rho;
    rho = V.dot(res, zk);
    if(rho == 0.0 || std::fabs(rho) > 1e20) {   // :233-237
      k->flag = 4;
      k->iter = ii + 1;
      break;
    }
    if(ii == 0) {
      V.copy(pk, zk);
    } else {
      const double beta = rho / rho1;
      if(beta == 0.0 || std::fabs(beta) > 1e20) {
        k->flag = 4;
        k->iter = ii + 1;
        break;
      }
      V.scale(pk, beta);
      V.axpy(pk, 1.0, zk);
    }
    V.apply(k->A, k->Au, qk, pk);
    pq = V.dot(pk, qk);
    if(pq <= 0.0 || std::fabs(pq) > 1e20) {   // :256-262
      k->flag = 4;
      k->iter = ii + 1;
      break;
    }
    alpha = rho / pq;
    if(std::fabs(alpha) > 1e20) {
      k->flag = 4;
      k->iter = ii + 1;
      break;
    }
    if(V.nrm(pk) * std::fabs(alpha) < eps * V.nrm(xk)) stagsteps++; else stagsteps = 0;   // :270-274
    V.axpy(xk, alpha, pk);
    V.axpy(res, -alpha, qk);
    normr = V.nrm(res);
    k->abs_resid = normr;
    if(normr <= tolb || stagsteps >= maxstagsteps || moresteps) {   // :284-309
      V.resid(k, res, xk, b);
      k->abs_resid = V.nrm(res);
      if(k->abs_resid <= tolb) {
        V.copy(b, xk);
        k->flag = 0;
        k->iter = ii + 1;
        break;
      }
      if(stagsteps >= maxstagsteps && moresteps == 0) stagsteps = 0;
      moresteps++;
      if(moresteps >= maxmsteps) {
        k->flag = 3;
        k->iter = ii + 1;
        break;
      }
    }
    if(k->abs_resid < normrmin) {   // :311-315
      normrmin = k->abs_resid;
      V.copy(xmin, xk);
      imin = ii;
    }
    if(stagsteps >= maxstagsteps) {
      k->flag = 3;
      k->iter = ii + 1;
      break;
    }
    if(V.rc != HIOPAMD_OK) break;
  }
  if(k->flag == 0) {   // :324-328
    k->rel_resid = k->abs_resid / n2b;
    V.copy(b, xk);
    return V.rc == HIOPAMD_OK;
  }
  V.resid(k, res, xmin, b);   // :330-352
  const double normr_comp = V.nrm(res);
  if(normr_comp <= k->abs_resid) {
    V.copy(b, xmin);
    k->iter = (double)(imin + 1);
    k->abs_resid = normr_comp;
    k->rel_resid = normr_comp / n2b;
  } else {
    V.copy(b, xk);
    k->iter = ii + 1;
    k->rel_resid = k->abs_resid / n2b;
  }
  return false;
}

bool bicgstab_solve(hiopamd_krylov* k, double* b)
{
  Vec V{k->ctx, k->n};
  double *xk = k->w[0], *xmin = k->w[1], *res = k->w[2], *pk = k->w[3], *ph = k->w[4], *v = k->w[5], *sk = k->w[6],
         *t = k->w[7], *rt = k->w[8];
  const double n2b = V.nrm(b);
  if(n2b == 0.0) {   // :402-411
    k->flag = 0;
    k->iter = 0.0;
    k->rel_resid = 0.0;
    k->abs_resid = 0.0;
    return true;
  }
  k->flag = 1;
  double imin = 0.0;
  const double tolb = k->tol * n2b;
  V.copy(xmin, xk);
  V.resid(k, res, xk, b);
  double normr = V.nrm(res);
  k->abs_resid = normr;
  if(normr <= tolb) {   // :449-458
    V.copy(b, xk);
    k->flag = 0;
    k->iter = 0.0;
    k->rel_resid = normr / n2b;
    return true;
  }
  V.copy(rt, res);
  double normrmin = normr, rho = 1.0, omega = 1.0, alpha = 0.0;
  int stagsteps = 0, moresteps = 0;
  const double eps = std::numeric_limits<double>::epsilon();
  const int maxmsteps = 100, maxstagsteps = 3;
  // preconditioned direction: ph = MR (ML x)   (MR is applied in place in the reference, :509-511)
  auto precond = [&](double* out, const double* in) {
    if(k->ML) V.apply(k->ML, k->MLu, out, in); else V.copy(out, in);
    if(k->MR) {
      V.apply(k->MR, k->MRu, t, out);   // t is free at both call sites
      V.copy(out, t);
    }
  };
  int ii = 0;
  for(; ii < k->maxit; ++ii) {
    const double rho1 = rho;
    rho = V.dot(rt, res);
    if(rho == 0.0 || std::fabs(rho) > 1e40) {   // :481-485
      k->flag = 4;
      k->iter = ii + 1 - 0.5;
      break;
    }
    if(ii == 0) {
      V.copy(pk, res);
    } else {
      const double beta = rho / rho1 * (alpha / omega);
      if(beta == 0.0 || std::fabs(beta) > 1e40) {
        k->flag = 4;
        k->iter = ii + 1 - 0.5;
        break;
      }
      V.axpy(pk, -omega, v);   // pk = (pk - omega v) beta + res, :496-498
      V.scale(pk, beta);
      V.axpy(pk, 1.0, res);
    }
    precond(ph, pk);
    V.apply(k->A, k->Au, v, ph);
    const double rtv = V.dot(rt, v);
    if(rtv == 0.0 || std::fabs(rtv) > 1e40) {
      k->flag = 4;
      k->iter = ii + 1 - 0.5;
      break;
    }
    alpha = rho / rtv;
    if(std::fabs(alpha) > 1e20) {
      k->flag = 4;
      k->iter = ii + 1 - 0.5;
      break;
    }
    if(V.nrm(ph) * std::fabs(alpha) < eps * V.nrm(xk)) stagsteps++; else stagsteps = 0;   // :531-535
    V.axpy(xk, alpha, ph);
    V.copy(sk, res);
    V.axpy(sk, -alpha, v);
    normr = V.nrm(sk);
    k->abs_resid = normr;
    if(normr <= tolb || stagsteps >= maxstagsteps || moresteps) {   // :546-570
      V.resid(k, sk, xk, b);
      k->abs_resid = V.nrm(sk);
      if(k->abs_resid <= tolb) {
        k->flag = 0;
        k->iter = ii + 1 - 0.5;
        break;
      }
      if(stagsteps >= maxstagsteps && moresteps == 0) stagsteps = 0;
      moresteps++;
      if(moresteps >= maxmsteps) {
        if(k->ref_exit) V.copy(b, xk);   // :563 (the reference's closing comparison then runs against this vector)
        k->flag = 3;
        k->iter = ii + 1 - 0.5;
        break;
      }
    }
    if(stagsteps >= maxstagsteps) {
      k->flag = 3;
      k->iter = ii + 1 - 0.5;
      break;
    }
    if(k->abs_resid < normrmin) {
      normrmin = k->abs_resid;
      V.copy(xmin, xk);
      imin = ii + 1 - 0.5;
    }
    precond(ph, sk);
    V.apply(k->A, k->Au, t, ph);
    const double tt = V.dot(t, t);
    if(tt == 0.0 || std::fabs(tt) > 1e20) {
      k->flag = 4;
      k->iter = ii + 1;
      break;
    }
    omega = V.dot(t, sk) / tt;
    if(std::fabs(omega) > 1e20) {
      k->flag = 4;
      k->iter = ii + 1;
      break;
    }
    if(V.nrm(ph) * std::fabs(omega) < eps * V.nrm(xk)) stagsteps++; else stagsteps = 0;
    V.axpy(xk, omega, ph);
    V.copy(res, sk);
    V.axpy(res, -omega, t);
    normr = V.nrm(res);
    k->abs_resid = normr;
    if(normr <= tolb || stagsteps >= maxstagsteps || moresteps) {   // :623-648
      V.resid(k, res, xk, b);
      k->abs_resid = V.nrm(res);
      if(k->abs_resid <= tolb) {
        k->flag = 0;
        k->iter = ii + 1;
        break;
      }
      if(stagsteps >= maxstagsteps && moresteps == 0) stagsteps = 0;
      moresteps++;
      if(moresteps >= maxmsteps) {
        if(k->ref_exit) V.copy(b, xk);   // :641
        k->flag = 3;
        k->iter = ii + 1;
        break;
      }
    }
    if(k->abs_resid < normrmin) {
      normrmin = k->abs_resid;
      V.copy(xmin, xk);
      imin = ii + 1;
    }
    if(stagsteps >= maxstagsteps) {
      k->flag = 3;
      k->iter = ii + 1 - 0.5;
      break;
    }
    if(V.rc != HIOPAMD_OK) break;
  }
  if(k->flag == 0) {   // :665-669
    k->rel_resid = k->abs_resid / n2b;
    V.copy(b, xk);
    return V.rc == HIOPAMD_OK;
  }
  V.resid(k, res, xmin, b);   // :671-688
  const double normr_comp = V.nrm(res);
  if(normr_comp <= k->abs_resid) {
    V.copy(b, xmin);
    k->iter = imin + 1;
    k->abs_resid = normr_comp;
    k->rel_resid = normr_comp / n2b;
  } else {
    V.copy(b, xk);
    k->iter = ii + 1;
    k->rel_resid = k->abs_resid / n2b;
  }
  return false;
}

}  // namespace

extern "C" {

int hiopamd_krylov_destroy(hiopamd_krylov* k);
int hiopamd_krylov_create(hiopamd_krylov** out, hiopamd_ctx* ctx, int kind, int64_t n, hiopamd_linop_fn A, void* A_user,
                          hiopamd_linop_fn ML, void* ML_user, hiopamd_linop_fn MR, void* MR_user)
{
  if(!out || !ctx || n < 0 || !A || (kind != 0 && kind != 1)) return HIOPAMD_ERR_ARG;
  hiopamd_krylov* k = new hiopamd_krylov();
  k->ctx = ctx;
  k->kind = kind;
  k->n = n;
  k->A = A; k->Au = A_user;
  k->ML = ML; k->MLu = ML_user;
  k->MR = MR; k->MRu = MR_user;
  const size_t bytes = sizeof(double) * (size_t)(n > 0 ? n : 1);
  *out = nullptr;
  for(int i = 0; i < 9; ++i) {
    // on any failure: release what exists (destroy tolerates null members), leave *out null
    if(hipMalloc((void**)&k->w[i], bytes) != hipSuccess ||
       hipMemsetAsync(k->w[i], 0, bytes, ctx->stream) != hipSuccess) {   // x0 = 0 until set_x0 says otherwise
      (void)hipGetLastError();
      hiopamd_krylov_destroy(k);
      return HIOPAMD_ERR_HIP;
    }
  }
  *out = k;
  return HIOPAMD_OK;
}

int hiopamd_krylov_destroy(hiopamd_krylov* k)
{
  if(!k) return HIOPAMD_OK;
  (void)hipStreamSynchronize(k->ctx->stream);
  for(int i = 0; i < 9; ++i) (void)hipFree(k->w[i]);
  delete k;
  return HIOPAMD_OK;
}

int hiopamd_krylov_set_tol(hiopamd_krylov* k, double tol)
{
  if(!k) return HIOPAMD_ERR_ARG;
  k->tol = tol;
  return HIOPAMD_OK;
}
int hiopamd_krylov_set_max_num_iter(hiopamd_krylov* k, int maxit)
{
  if(!k) return HIOPAMD_ERR_ARG;
  k->maxit = maxit;
  return HIOPAMD_OK;
}
int hiopamd_krylov_set_x0(hiopamd_krylov* k, double xval)
{
  if(!k) return HIOPAMD_ERR_ARG;
  return hiopamd_vec_set_to_constant(k->ctx, k->n, k->w[0], xval);
}
double* hiopamd_krylov_x0(hiopamd_krylov* k) { return k ? k->w[0] : nullptr; }

int hiopamd_krylov_solve(hiopamd_krylov* k, double* b_inout, int* converged_host)
{
  if(!k || !b_inout || !converged_host) return HIOPAMD_ERR_ARG;
  const bool ok = (k->kind == 0) ? pcg_solve(k, b_inout) : bicgstab_solve(k, b_inout);
  *converged_host = ok ? 1 : 0;
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

int hiopamd_krylov_get_convergence_flag(const hiopamd_krylov* k) { return k ? k->flag : -1; }
int hiopamd_krylov_set_exit_mode(hiopamd_krylov* k, int reference)
{
  if(!k) return HIOPAMD_ERR_ARG;
  k->ref_exit = reference != 0;
  return HIOPAMD_OK;
}
double hiopamd_krylov_get_sol_num_iter(const hiopamd_krylov* k) { return k ? k->iter : -1.0; }
double hiopamd_krylov_get_sol_abs_resid(const hiopamd_krylov* k) { return k ? k->abs_resid : -1.0; }
double hiopamd_krylov_get_sol_rel_resid(const hiopamd_krylov* k) { return k ? k->rel_resid : -1.0; }

}  // extern "C"
