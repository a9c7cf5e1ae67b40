// Library-internal forms of the tall-skinny GEMV kernels (csrc/dense_kernels.hip) that the low-rank KKT uses to do in ONE pass what the
// reference does in several calls.  Not part of the C ABI.
#pragma once
#include "common.hpp"

namespace hiopamd {

// Up to three row groups A_g (m_g x n, the same leading dimension) walked by ONE stage-1 launch against the same x:
//   v_g = A_g (x .* xscale)                      (xscale may be null)
//   y[row0_g + r] = alpha_g * v_g[r]             groups 1, 2
//   y[r]          = alpha_0 * v_0[r] - sub(r)    group 0;  sub(r) = r < nsub0 ? sub0[r] : sub1[r - nsub0]  (sub0 == null: nothing is subtracted)
// The partial sums of a row are formed in the order of hiopamd_mat_times_vec (same column chunks, same fold).
int gemv_n_groups(hiopamd_ctx* ctx, int64_t n, int64_t lda, int ngroups, const double* const* A, const int* m, const double* x,
                  const double* xscale, double* y, const double* alpha, const double* sub0, int nsub0, const double* sub1);

// y <- y + alpha A^T x  (A m x n), and in the same pass, per column j with the finished y_j:
//   res = sigma * sum_q S[q][j] sy[q] + sum_q Y[q][j] sy[l + q]          (S, Y: l x n, leading dimension ld; sy: 2 l values)
//   dx[j] = y_j * DhInv[j] - res * DhInv[j]
struct GemvtTail {
  const double* S;
  const double* Y;
  int l;
  int64_t ld;
  double sigma;
  const double* sy;
  const double* DhInv;
  double* dx;
};
int gemv_t_tail(hiopamd_ctx* ctx, int m, int64_t n, const double* A, int64_t lda, double* y, double alpha, const double* x,
                const GemvtTail& tail);

}  // namespace hiopamd
