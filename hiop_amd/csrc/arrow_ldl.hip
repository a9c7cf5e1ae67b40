// A sparse DIRECT solver for bordered-diagonal symmetric matrices — the inner solver of the condensed sparse KKT when its pattern
// allows it.
//
// reference: hiopKKTLinSysCondensedSparse hands M = H + Dx + Jd^T Dd Jd to a sparse Cholesky (MA57 / cuSOLVER,
// src/Optimization/hiopKKTLinSysSparseCondensed.cpp:469-496); whether that factorisation exists IS the positive-definiteness
// verdict the inertia-correction loop branches on (:386-388).  Neither library is in the image.  For the sparse example problems of
// the reference (src/Drivers/Sparse/NlpSparseEx1.cpp, NlpSparseEx2.cpp: every constraint couples x_1 with one other variable, the
// Hessian is diagonal) M is an ARROWHEAD: a diagonal plus one dense row / column.  More generally, whenever a small set B of
// "border" variables covers every off-diagonal entry,
//         M = [ D   E  ]   D diagonal (the other n - p variables),  E (n - p) x p sparse,  C p x p
//             [ E^T C  ]
// has the exact factorisation  M = L diag(D, S) L^T  with the Schur complement  S = C - E^T D^-1 E  (p x p, dense), and by
// Haynsworth's inertia additivity  inertia(M) = inertia(D) + inertia(S): the verdict "M is positive definite" is EXACT and does not
// depend on any right-hand side (the Krylov inner solver of round 3 could only report negative curvature along the directions it
// happened to visit).  Work per factorisation: one pass over the nonzeros (the Schur products through this library's row-build
// plan, fixed summation order) + a p x p factorisation on the host; per solve: two passes, no host round trip.
// The border is found once per pattern by a greedy vertex cover of the off-diagonal graph (highest remaining degree first), up to
// ARROW_PMAX border variables; a pattern that needs more is not handled here (the caller keeps its other inner solvers).
#include "device_utils.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

constexpr int ARROW_PMAX = 32;

struct hiopamd_arrow_ldl {
  hiopamd_ctx* ctx = nullptr;
  int n = 0, p = 0, nbt = 0;
  std::vector<int> border;       // host: the p border variables
  int* is_border = nullptr;      // n: border index or -1
  int* dpos = nullptr;           // n: CSR position of (i, i)
  int* arow_ptr = nullptr;       // n + 1: the border couplings of row i (non-border rows)
  int* a_q = nullptr;            // ... border index
  int* a_bt = nullptr;           // ... index into bt_val
  int* bt_i = nullptr;           // E^T as row-sorted triplets (row = border index, column = variable)
  int* bt_j = nullptr;
  int* bt_pos = nullptr;         // CSR position of the entry
  int* border_dev = nullptr;     // p
  int* cpos = nullptr;           // p * p: CSR position of (b_a, b_b), -1 if structurally zero
  double* bt_val = nullptr;
  double* dvec = nullptr;        // n: D (1 at the border variables)
  double* tmp = nullptr;         // n
  double* S = nullptr;           // p * p (device): C - E^T D^-1 E, upper triangle
  double* Sinv = nullptr;        // p * p (device): S^-1, full
  double* t = nullptr;           // p
  hiopamd_sp_plan* plan = nullptr;
  bool factored = false;
  int n_neg = 0, n_zero = 0;
};

using namespace hiopamd;

#define RC(x)                         \
  do {                                \
    int rc_ = (x);                    \
    if(rc_ != HIOPAMD_OK) return rc_; \
  } while(0)

namespace {
template <class T>
int upv(T** d, const std::vector<T>& h)
{
  *d = nullptr;
  if(hipMalloc((void**)d, sizeof(T) * (h.size() ? h.size() : 1)) != hipSuccess) return HIOPAMD_ERR_HIP;
  if(!h.empty() && hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return HIOPAMD_ERR_HIP;
  return HIOPAMD_OK;
}
struct cnt2_t {
  double a, b;
};
struct OpCountSignD {   // (negative, zero-or-non-finite) entries of D at the non-border variables: thresholds of the dense solver's inertia
  const double* d;
  const int* isb;
  __device__ cnt2_t identity() const { return cnt2_t{0.0, 0.0}; }
  __device__ cnt2_t map(int64_t i) const
  {
    if(isb[i] >= 0) return cnt2_t{0.0, 0.0};
    const double v = d[i];
    const bool bad = !(fabs(v) >= 1e-14) || !isfinite(v);
    return cnt2_t{(!bad && v < 0.0) ? 1.0 : 0.0, bad ? 1.0 : 0.0};
  }
  __device__ cnt2_t combine(cnt2_t a, cnt2_t b) const { return cnt2_t{a.a + b.a, a.b + b.b}; }
};
}  // namespace

extern "C" {

int hiopamd_arrow_ldl_destroy(hiopamd_arrow_ldl* s)
{
  if(!s) return HIOPAMD_OK;
  if(s->ctx) (void)hipStreamSynchronize(s->ctx->stream);
  if(s->plan) hiopamd_sp_plan_destroy(s->plan);
  void* ps[] = {s->is_border, s->dpos, s->arow_ptr, s->a_q, s->a_bt, s->bt_i, s->bt_j, s->bt_pos, s->border_dev, s->cpos,
                s->bt_val, s->dvec, s->tmp, s->S, s->Sinv, s->t};
  for(void* p : ps) (void)hipFree(p);
  delete s;
  return HIOPAMD_OK;
}

// pattern of the full symmetric matrix in CSR (host arrays, columns sorted inside a row).  HIOPAMD_ERR_STATE: the pattern is not a
// bordered diagonal with at most ARROW_PMAX border variables (nothing is created).
int hiopamd_arrow_ldl_create(hiopamd_arrow_ldl** out, hiopamd_ctx* ctx, int n, const int* rowptr_host, const int* colidx_host)
{
  if(!out || !ctx || n < 0 || !rowptr_host || (rowptr_host[n] > 0 && !colidx_host)) return HIOPAMD_ERR_ARG;
  *out = nullptr;
  // ---- greedy vertex cover of the off-diagonal graph
  std::vector<int64_t> deg((size_t)n, 0);
  std::vector<int> dpos((size_t)n, -1);
  for(int i = 0; i < n; ++i)
    for(int q = rowptr_host[i]; q < rowptr_host[i + 1]; ++q) {
      const int c = colidx_host[q];
      if(c < 0 || c >= n) return HIOPAMD_ERR_ARG;
      if(c == i) dpos[(size_t)i] = q;
      else deg[(size_t)i] += 1;
    }
  for(int i = 0; i < n; ++i)
    if(dpos[(size_t)i] < 0) return HIOPAMD_ERR_STATE;   // a structurally zero diagonal entry: not this solver's pattern
  std::vector<int> isb((size_t)n, -1), border;
  for(;;) {
    int best = -1;
    int64_t bd = 0;
    for(int i = 0; i < n; ++i)
      if(isb[(size_t)i] < 0 && deg[(size_t)i] > bd) {
        bd = deg[(size_t)i];
        best = i;
      }
    if(best < 0) break;   // no off-diagonal entry left uncovered
    if((int)border.size() >= ARROW_PMAX) return HIOPAMD_ERR_STATE;
    isb[(size_t)best] = (int)border.size();
    border.push_back(best);
    for(int q = rowptr_host[best]; q < rowptr_host[best + 1]; ++q) {   // its edges are covered (the pattern is symmetric)
      const int c = colidx_host[q];
      if(c != best && isb[(size_t)c] < 0) deg[(size_t)c] -= 1;
    }
    deg[(size_t)best] = 0;
  }
  const int p = (int)border.size();
  // ---- the couplings: E^T as triplets sorted by (border index, variable), per-row lists for the back substitution
  std::vector<int> bt_i, bt_j, bt_pos, cpos((size_t)p * (size_t)std::max(p, 1), -1);
  for(int a = 0; a < p; ++a) {
    const int b = border[(size_t)a];
    for(int q = rowptr_host[b]; q < rowptr_host[b + 1]; ++q) {
      const int c = colidx_host[q];
      if(isb[(size_t)c] >= 0) cpos[(size_t)a * p + isb[(size_t)c]] = q;
      else {
        bt_i.push_back(a);
        bt_j.push_back(c);
        bt_pos.push_back(q);
      }
    }
  }
  const int nbt = (int)bt_i.size();
  std::vector<int> arow_ptr((size_t)n + 1, 0), a_q((size_t)nbt), a_bt((size_t)nbt);
  for(int t = 0; t < nbt; ++t) arow_ptr[(size_t)bt_j[(size_t)t] + 1] += 1;
  for(int i = 0; i < n; ++i) arow_ptr[(size_t)i + 1] += arow_ptr[(size_t)i];
  {
    std::vector<int> cur(arow_ptr.begin(), arow_ptr.end() - 1);
    for(int t = 0; t < nbt; ++t) {   // (t ascends with the border index: the row lists are sorted by border index)
      const int e = cur[(size_t)bt_j[(size_t)t]]++;
      a_q[(size_t)e] = bt_i[(size_t)t];
      a_bt[(size_t)e] = t;
    }
  }
  // a non-border row may only hold its diagonal entry and border couplings (cover property + symmetry): verify
  for(int i = 0; i < n; ++i)
    if(isb[(size_t)i] < 0)
      for(int q = rowptr_host[i]; q < rowptr_host[i + 1]; ++q)
        if(colidx_host[q] != i && isb[(size_t)colidx_host[q]] < 0) return HIOPAMD_ERR_STATE;   // (an unsymmetric pattern)
  auto* s = new hiopamd_arrow_ldl();
  s->ctx = ctx;
  s->n = n;
  s->p = p;
  s->nbt = nbt;
  s->border = border;
  int rc = upv(&s->is_border, isb);
  if(rc == HIOPAMD_OK) rc = upv(&s->dpos, dpos);
  if(rc == HIOPAMD_OK) rc = upv(&s->arow_ptr, arow_ptr);
  if(rc == HIOPAMD_OK) rc = upv(&s->a_q, a_q);
  if(rc == HIOPAMD_OK) rc = upv(&s->a_bt, a_bt);
  if(rc == HIOPAMD_OK) rc = upv(&s->bt_i, bt_i);
  if(rc == HIOPAMD_OK) rc = upv(&s->bt_j, bt_j);
  if(rc == HIOPAMD_OK) rc = upv(&s->bt_pos, bt_pos);
  if(rc == HIOPAMD_OK) rc = upv(&s->border_dev, border);
  if(rc == HIOPAMD_OK) rc = upv(&s->cpos, cpos);
  auto dal = [](double** d, size_t k) { return hipMalloc((void**)d, sizeof(double) * (k ? k : 1)) == hipSuccess ? HIOPAMD_OK : HIOPAMD_ERR_HIP; };
  if(rc == HIOPAMD_OK) rc = dal(&s->bt_val, (size_t)nbt);
  if(rc == HIOPAMD_OK) rc = dal(&s->dvec, (size_t)n);
  if(rc == HIOPAMD_OK) rc = dal(&s->tmp, (size_t)n);
  if(rc == HIOPAMD_OK) rc = dal(&s->S, (size_t)p * p);
  if(rc == HIOPAMD_OK) rc = dal(&s->Sinv, (size_t)p * p);
  if(rc == HIOPAMD_OK) rc = dal(&s->t, (size_t)p);
  // S -= E^T D^-1 E: the Schur row-build plan of csrc/sparse_kernels.hip on E^T (p rows), upper triangle of the p x p result
  if(rc == HIOPAMD_OK && p > 0)
    rc = hiopamd_sp_plan_create(&s->plan, p, p, n, nbt, bt_i.data(), bt_j.data(), nbt, bt_i.data(), bt_j.data(), 1);
  if(rc != HIOPAMD_OK) {
    hiopamd_arrow_ldl_destroy(s);
    return rc;
  }
  *out = s;
  return HIOPAMD_OK;
}

int hiopamd_arrow_ldl_border(const hiopamd_arrow_ldl* s, int* p_host, int* border_host /* may be NULL; ARROW_PMAX ints */)
{
  if(!s || !p_host) return HIOPAMD_ERR_ARG;
  *p_host = s->p;
  if(border_host)
    for(int a = 0; a < s->p; ++a) border_host[a] = s->border[(size_t)a];
  return HIOPAMD_OK;
}

// numeric factorisation on the CSR values of the pattern given at create.  *n_neg_host / *n_zero_host: negative / (numerically) zero
// pivots of M = L diag(D, S) L^T — the inertia of M (thresholds of the dense solver: |d| < 1e-14 is zero).
int hiopamd_arrow_ldl_factorize(hiopamd_arrow_ldl* s, const double* vals, int* n_neg_host, int* n_zero_host)
{
  if(!s || !vals || !n_neg_host || !n_zero_host) return HIOPAMD_ERR_ARG;
  hiopamd_ctx* ctx = s->ctx;
  const int n = s->n, p = s->p, nbt = s->nbt;
  s->factored = false;
  {
    const int *isb = s->is_border, *dpos = s->dpos, *bt_pos = s->bt_pos;
    double *dvec = s->dvec, *bt_val = s->bt_val;
    const int64_t nmax = n > nbt ? n : nbt;
    RC(launch_ew(ctx, nmax, [=] __device__(int64_t i) {
      if(i < n) dvec[i] = isb[i] >= 0 ? 1.0 : vals[dpos[i]];
      if(i < nbt) bt_val[i] = vals[bt_pos[i]];
    }));
  }
  cnt2_t cnt{0.0, 0.0};
  std::vector<double> Sh((size_t)p * (size_t)std::max(p, 1), 0.0);
  if(p > 0) {
    const int* cpos = s->cpos;
    double* S = s->S;
    RC(launch_ew(ctx, (int64_t)p * p, [=] __device__(int64_t e) { S[e] = cpos[e] >= 0 ? vals[cpos[e]] : 0.0; }));
    RC(hiopamd_sp_add_MDinvNt(ctx, s->plan, s->bt_val, s->bt_val, s->dvec, -1.0, s->S, p, 0, 0));
    HIOPAMD_CHECK(hipMemcpyAsync(Sh.data(), s->S, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost, ctx->stream));
  }
  RC(launch_reduce<cnt2_t>(ctx, n, OpCountSignD{s->dvec, s->is_border}, &cnt));   // (synchronises: S has arrived too)
  int nneg = (int)cnt.a, nzero = (int)cnt.b;
  // ---- S = U^T D_S U on the host (p <= 32), its inertia and its inverse
  std::vector<double> Si((size_t)p * (size_t)std::max(p, 1), 0.0);
  bool s_singular = false;
  if(p > 0) {
    std::vector<double> A(Sh);   // upper triangle significant
    for(int i = 0; i < p; ++i)
      for(int j = 0; j < i; ++j) A[(size_t)i * p + j] = A[(size_t)j * p + i];
    std::vector<double> L((size_t)p * p, 0.0), d((size_t)p, 0.0);
    for(int k = 0; k < p; ++k) {   // A = L d L^T, L unit lower
      double dk = A[(size_t)k * p + k];
      for(int q = 0; q < k; ++q) dk -= L[(size_t)k * p + q] * L[(size_t)k * p + q] * d[(size_t)q];
      d[(size_t)k] = dk;
      L[(size_t)k * p + k] = 1.0;
      const bool bad = !(std::fabs(dk) >= 1e-14) || !std::isfinite(dk);
      if(bad) {
        nzero += 1;
        s_singular = true;
        break;
      }
      if(dk < 0.0) nneg += 1;
      for(int i = k + 1; i < p; ++i) {
        double v = A[(size_t)i * p + k];
        for(int q = 0; q < k; ++q) v -= L[(size_t)i * p + q] * L[(size_t)k * p + q] * d[(size_t)q];
        L[(size_t)i * p + k] = v / dk;
      }
    }
    if(!s_singular) {
      for(int c = 0; c < p; ++c) {   // column c of S^-1: L y = e_c, z = y / d, L^T x = z
        std::vector<double> y((size_t)p, 0.0);
        for(int i = 0; i < p; ++i) {
          double v = (i == c) ? 1.0 : 0.0;
          for(int q = 0; q < i; ++q) v -= L[(size_t)i * p + q] * y[(size_t)q];
          y[(size_t)i] = v;
        }
        for(int i = 0; i < p; ++i) y[(size_t)i] /= d[(size_t)i];
        for(int i = p - 1; i >= 0; --i) {
          double v = y[(size_t)i];
          for(int q = i + 1; q < p; ++q) v -= L[(size_t)q * p + i] * y[(size_t)q];
          y[(size_t)i] = v;
        }
        for(int i = 0; i < p; ++i) Si[(size_t)i * p + c] = y[(size_t)i];
      }
      HIOPAMD_CHECK(hipMemcpy(s->Sinv, Si.data(), sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice));
    }
  }
  s->n_neg = nneg;
  s->n_zero = nzero;
  *n_neg_host = nneg;
  *n_zero_host = nzero;
  s->factored = nzero == 0;
  return HIOPAMD_OK;
}

// x <- M^-1 x (device vector of length n), with the factors of the last hiopamd_arrow_ldl_factorize.  No host round trip.
int hiopamd_arrow_ldl_solve(hiopamd_arrow_ldl* s, double* x)
{
  if(!s || !x) return HIOPAMD_ERR_ARG;
  if(!s->factored) return HIOPAMD_ERR_STATE;
  hiopamd_ctx* ctx = s->ctx;
  const int n = s->n, p = s->p;
  const int* isb = s->is_border;
  const double* dvec = s->dvec;
  if(p == 0) return launch_ew(ctx, n, [=] __device__(int64_t i) { x[i] /= dvec[i]; });
  double *tmp = s->tmp, *t = s->t;
  // t = E^T D^-1 r_D
  RC(launch_ew(ctx, n, [=] __device__(int64_t i) { tmp[i] = isb[i] >= 0 ? 0.0 : x[i] / dvec[i]; }));
  RC(hiopamd_sp_times_vec(ctx, p, n, s->nbt, s->bt_i, s->bt_j, s->bt_val, 0.0, t, 1.0, tmp));
  // x_B = S^-1 (r_B - t)   (one small workgroup; the result replaces the border entries of x and stays in t for the next pass)
  {
    const int* border = s->border_dev;
    const double* Sinv = s->Sinv;
    RC(launch_ew(ctx, 1, [=] __device__(int64_t) {
      double rb[ARROW_PMAX], xb[ARROW_PMAX];
      for(int a = 0; a < p; ++a) rb[a] = x[border[a]] - t[a];
      for(int a = 0; a < p; ++a) {
        double v = 0.0;
        for(int b = 0; b < p; ++b) v += Sinv[a * p + b] * rb[b];
        xb[a] = v;
      }
      for(int a = 0; a < p; ++a) {
        t[a] = xb[a];
        x[border[a]] = xb[a];
      }
    }));
  }
  // x_D = D^-1 (r_D - E x_B)
  {
    const int *ap = s->arow_ptr, *aq = s->a_q, *abt = s->a_bt;
    const double* bt_val = s->bt_val;
    RC(launch_ew(ctx, n, [=] __device__(int64_t i) {
      if(isb[i] >= 0) return;
      double v = x[i];
      for(int e = ap[i]; e < ap[i + 1]; ++e) v -= bt_val[abt[e]] * t[aq[e]];
      x[i] = v / dvec[i];
    }));
  }
  return HIOPAMD_OK;
}

}  // extern "C"
