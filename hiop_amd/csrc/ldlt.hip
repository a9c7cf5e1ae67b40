// Dense symmetric-indefinite (quasi-definite KKT) factorisation A = U^T D U without pivoting,
// inertia, and triangular solves, for gfx950 (MI355X) — the hiopLinSolverSymDense operator.
//
// reference operator: src/LinAlg/hiopLinSolver.hpp:78-130 (sysMatrix / matrixChanged / solve);
// reference GPU implementation it replaces: MAGMA magma_dsytrf_nopiv_gpu / magma_dsytrs_nopiv_gpu /
// magmablas_ddiinertia (src/LinAlg/hiopLinSolverSymDenseMagma.cpp:324-480); CPU implementation used
// for parity checks: LAPACK DSYTRF/DSYTRS + LINPACK-dsidi style inertia
// (src/LinAlg/hiopLinSolverSymDenseLapack.hpp:75-195).
//
// Layout: the KKT matrix is row-major with only its UPPER triangle populated
// (src/LinAlg/readme.md:24-26) — i.e. column-major lower in LAPACK's eyes.  Row k of the upper
// factor U is therefore contiguous, which is what every kernel below streams.
//
// Algorithm (two-level right-looking, all on one stream, no host sync inside):
//   for each super-panel of NB=256 rows:
//     for each panel of nb=64 rows inside it:
//       ldlt_panel   : every workgroup factors the 64x64 diagonal block in LDS (redundantly, 8 us of
//                      latency, removes a launch + a grid sync), then forward-substitutes its 256
//                      columns of the row panel: V = U11^-T A12 (kept un-scaled in a workspace) and
//                      U12 = D^-1 V (in place).
//       ldlt_update  : rows of the super-panel below the panel:  A[r][c] -= sum_k V[k][r] U[k][c]  (K=64)
//     ldlt_update    : trailing matrix, K=256 rank update on fp64 MFMA (v_mfma_f64_16x16x4_f64)
// K=256 for the trailing update is what makes it MFMA-bound instead of HBM-bound: the C tile is
// read+written once per 2*256 flops/element (32 flop/B vs the ~12.5 flop/B ridge of
// 78.6 TFLOP/s / 6.3 TB/s).
#include "device_utils.hpp"

#include <vector>

namespace hiopamd {

constexpr int LD_NB = 256;   // super-panel rows (K of the trailing update)
constexpr int LD_nb = 64;    // panel rows
constexpr int LD_TM = 128;   // update tile
constexpr int LD_TN = 128;
constexpr int LD_KT = 16;    // k-depth staged in LDS per step
constexpr int LD_LDP = LD_TM + 16;  // padded LDS row stride (doubles): rows k,k+1 land on disjoint bank halves

typedef double double4_t __attribute__((ext_vector_type(4)));

// wave-uniform broadcast of lane `src` (a compile-time constant after unrolling): two v_readlane_b32
// into SGPRs instead of a ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ double bcast_lane(double v, int src)
{
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}
// reciprocal: v_rcp_f64 seed + one Newton step (error <= 1 ulp; the factor is checked by residual tests)
__device__ __forceinline__ double fast_rcp(double d)
{
  double x = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, x, 1.0);
  x = fma(x, e, x);
  return x;
}

// ------------------------------------------------------------------------------------------
// diag kernel: LDL^T of the kb x kb (<=64) diagonal block by ONE workgroup, blocked by 16:
//   (i)   the 16x16 diagonal sub-block is factored by wave 0 entirely in registers (4 entries per
//         lane, pivot row / multiplier broadcast with wave shuffles — no LDS round trips, no barriers);
//   (ii)  the 16 x (rest) row panel is forward-substituted, one column per thread;
//   (iii) the trailing part of the 64x64 block gets the rank-16 update, 9 entries per thread.
// During the factorisation S holds UN-scaled rows (v_kc = d_k * u_kc); rows are scaled at the end.
// Outputs: U11 (strictly upper, unit diagonal implied) and D (diagonal) written back into A; a compact
// zero-padded 64x64 copy Dk for the substitution kernel; dinv; and Li = the four 16x16 inverses of the
// unit-lower diagonal sub-blocks of L11 = U11^T (row-major [sb][i][j]), which turn the panel
// substitution into MFMA GEMMs.  info[0] = 1-based index of the first zero / non-finite pivot.
// ------------------------------------------------------------------------------------------
constexpr int LD_SB = 16;

__global__ __launch_bounds__(kBlock) void ldlt_diag_kernel(double* __restrict__ A, int64_t lda, int k0, int kb,
                                                           double* __restrict__ dinv, double* __restrict__ Dk,
                                                           double* __restrict__ Li, int* __restrict__ info)
{
  __shared__ double S[LD_nb][LD_nb + 1];
  __shared__ double sdinv[LD_nb];
  const int tid = threadIdx.x;
  {
    double sv[LD_nb * LD_nb / kBlock];
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      const int r = e >> 6, c = e & 63;
      const int rr = (r < kb) ? r : (kb - 1), cc = (c < kb) ? c : (kb - 1);
      const double t = A[(int64_t)(k0 + rr) * lda + (k0 + cc)];   // unconditional clamped load + select
      sv[q] = (r < kb && c < kb && c >= r) ? t : 0.0;
    }
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      S[e >> 6][e & 63] = sv[q];
    }
  }
  if(tid < LD_nb) sdinv[tid] = 1.0;
  __syncthreads();

  for(int sb = 0; sb < LD_nb / LD_SB; ++sb) {
    const int o = sb * LD_SB;
    if(o >= kb) break;  // uniform
    // ---- (i) 16x16 diagonal sub-block in registers of wave 0: lane c (< 16) holds column c
    if(tid < 64) {
      const int c = tid & 15;
      double a[LD_SB];
#pragma unroll
      for(int r = 0; r < LD_SB; ++r) a[r] = S[o + r][o + c];   // zeros below the diagonal
#pragma unroll
      for(int k = 0; k < LD_SB; ++k) {
        if(o + k < kb) {  // uniform
          const double d = bcast_lane(a[k], k);          // pivot S[k][k]
          const double di = fast_rcp(d);
          const double ukc = a[k] * di;                  // scaled pivot-row entry of my column
#pragma unroll
          for(int r = k + 1; r < LD_SB; ++r) {
            const double vkr = bcast_lane(a[k], r);      // S[k][r] = pivot-row entry of column r
            a[r] = fma(-vkr, ukc, a[r]);
          }
          if(tid == 0) {
            sdinv[o + k] = di;
            if(d == 0.0 || !isfinite(d)) atomicCAS(info, 0, k0 + o + k + 1);
          }
        }
      }
      if(tid < LD_SB) {
#pragma unroll
        for(int r = 0; r < LD_SB; ++r)
          if(c >= r) S[o + r][o + c] = a[r];
      }
    }
    __syncthreads();
    // ---- (ii) row panel: columns c >= o+16, one per thread; x_r -= S[o+s][o+r] * (x_s/d_s)
    {
      const int c = o + LD_SB + tid;
      if(c < LD_nb && c < kb) {
        double x[LD_SB];
#pragma unroll
        for(int r = 0; r < LD_SB; ++r) x[r] = S[o + r][c];
#pragma unroll
        for(int q = 0; q < LD_SB - 1; ++q) {
          const double us = x[q] * sdinv[o + q];
#pragma unroll
          for(int r = q + 1; r < LD_SB; ++r) x[r] = fma(-S[o + q][o + r], us, x[r]);
        }
#pragma unroll
        for(int r = 0; r < LD_SB; ++r) S[o + r][c] = x[r];
      }
    }
    __syncthreads();
    // ---- (iii) trailing rank-16 update inside the block: S[r][c] -= sum_k v_kr * v_kc / d_k, c >= r >= o+16
    {
      const int tr = tid >> 4, tc = tid & 15;
      double acc[3][3];
#pragma unroll
      for(int i = 0; i < 3; ++i)
#pragma unroll
        for(int j = 0; j < 3; ++j) acc[i][j] = 0.0;
      const int rb = o + LD_SB + tr, cb = o + LD_SB + tc;
#pragma unroll 4
      for(int k = 0; k < LD_SB; ++k) {
        const double di = sdinv[o + k];
        double vr[3], uc[3];
#pragma unroll
        for(int i = 0; i < 3; ++i) vr[i] = (rb + 16 * i < LD_nb) ? S[o + k][rb + 16 * i] : 0.0;
#pragma unroll
        for(int j = 0; j < 3; ++j) uc[j] = (cb + 16 * j < LD_nb) ? S[o + k][cb + 16 * j] * di : 0.0;
#pragma unroll
        for(int i = 0; i < 3; ++i)
#pragma unroll
          for(int j = 0; j < 3; ++j) acc[i][j] = fma(vr[i], uc[j], acc[i][j]);
      }
      __syncthreads();  // all reads of the k-panel done before anybody writes (rows >= o+16 only, but keep it simple)
#pragma unroll
      for(int i = 0; i < 3; ++i)
#pragma unroll
        for(int j = 0; j < 3; ++j) {
          const int r = rb + 16 * i, c = cb + 16 * j;
          if(r < LD_nb && c < LD_nb && c >= r) S[r][c] -= acc[i][j];
        }
    }
    __syncthreads();
  }
  // scale the rows, write the factor back and the compact copy
  for(int e = tid; e < LD_nb * LD_nb; e += kBlock) {
    const int r = e >> 6, c = e & 63;
    double v = S[r][c];
    if(c > r) v *= sdinv[r];
    const bool in = (r < kb && c < kb && c >= r);
    Dk[e] = in ? v : 0.0;
    if(in) A[(int64_t)(k0 + r) * lda + (k0 + c)] = v;
  }
  if(tid < kb) dinv[k0 + tid] = sdinv[tid];
  // inverses of the four unit-lower 16x16 diagonal sub-blocks of L11 = U11^T:
  // thread (sb, j) solves L x = e_j;  L[i][q] = S[o+q][o+i]*sdinv[o+q] for i > q
  if(tid < LD_nb) {
    const int sb = tid >> 4, j = tid & 15, o = sb * LD_SB;
    double x[LD_SB];
#pragma unroll
    for(int i = 0; i < LD_SB; ++i) x[i] = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for(int q = 0; q < LD_SB - 1; ++q) {
      const double xq = x[q] * sdinv[o + q];   // zero for q < j
#pragma unroll
      for(int i = q + 1; i < LD_SB; ++i) x[i] = fma(-S[o + q][o + i], xq, x[i]);
    }
#pragma unroll
    for(int i = 0; i < LD_SB; ++i) Li[sb * (LD_SB * LD_SB) + i * LD_SB + j] = x[i];
  }
}

// ------------------------------------------------------------------------------------------
// substitution kernel for the row panel right of the diagonal block, on fp64 MFMA:
//   V = L11^-1 A12  (un-scaled, to the workspace)   and   U12 = D^-1 V  (in place),
// as a block forward substitution over the four 16-row blocks I of the panel:
//   T_I = A_I - sum_{J<I} L_IJ V_J ,   V_I = Linv_II T_I .
// One wave64 per workgroup = 64 columns = 4 independent groups of 16 columns (independent MFMA
// chains hide the 64-cycle MFMA latency).  The MFMA D layout (row = (l>>4)+4*reg, col = l&15) of a
// 16x16 block is exactly the B-operand layout of its four k-steps (k-step kk <-> reg kk), so V_J feeds
// the next product straight from the accumulator registers.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ldlt_trsm_kernel(double* __restrict__ A, int64_t lda, int N, int k0,
                                                       double* __restrict__ V, int64_t ldv, int vrow0,
                                                       const double* __restrict__ dinv,
                                                       const double* __restrict__ Dk, const double* __restrict__ Li)
{
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15;
  const int64_t colbase = (int64_t)k0 + LD_nb + (int64_t)blockIdx.x * 64;
  // A operands (shared by the 4 column groups)
  double negL[4][4][4];  // [I][J][kk], J < I : -L11[16I+li][16J+4kk+g] = -U11[16J+4kk+g][16I+li]
  double inv[4][4];      // [I][kk]    : Linv_II[li][4kk+g]
#pragma unroll
  for(int I = 0; I < 4; ++I) {
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) {
      inv[I][kk] = Li[I * 256 + li * 16 + 4 * kk + g];
#pragma unroll
      for(int J = 0; J < 4; ++J)
        negL[I][J][kk] = (J < I) ? -Dk[(16 * J + 4 * kk + g) * LD_nb + 16 * I + li] : 0.0;
    }
  }
  double dsc[4][4];  // dinv of row 16I + g + 4r
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int r = 0; r < 4; ++r) dsc[I][r] = dinv[k0 + 16 * I + g + 4 * r];

#pragma unroll
  for(int grp = 0; grp < 4; ++grp) {
    const int64_t col = colbase + grp * 16 + li;
    const bool ok = col < N;
    double4_t a[4];
#pragma unroll
    for(int I = 0; I < 4; ++I)
#pragma unroll
      for(int r = 0; r < 4; ++r) a[I][r] = ok ? A[(int64_t)(k0 + 16 * I + g + 4 * r) * lda + col] : 0.0;
    double4_t Vv[4];
#pragma unroll
    for(int I = 0; I < 4; ++I) {
      double4_t t = a[I];
#pragma unroll
      for(int J = 0; J < 4; ++J) {
        if(J < I) {
#pragma unroll
          for(int kk = 0; kk < 4; ++kk) t = __builtin_amdgcn_mfma_f64_16x16x4f64(negL[I][J][kk], Vv[J][kk], t, 0, 0, 0);
        }
      }
      double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(inv[I][kk], t[kk], v, 0, 0, 0);
      Vv[I] = v;
    }
    if(ok) {
#pragma unroll
      for(int I = 0; I < 4; ++I)
#pragma unroll
        for(int r = 0; r < 4; ++r) {
          const int row = 16 * I + g + 4 * r;
          V[(int64_t)(vrow0 + row) * ldv + col] = Vv[I][r];
          A[(int64_t)(k0 + row) * lda + col] = Vv[I][r] * dsc[I][r];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// rank-K update on fp64 MFMA:  A[r][c] -= sum_{k<K} V[vrow0+k][r] * A[urow0+k][c]
//   for r in [s + ti*128 ...) ∩ [s, row_end),  c in [r, N)   (upper triangle only)
// One workgroup = 4 wave64 = one 128x128 tile; each wave owns a 64x64 quadrant = 4x4 MFMA tiles
// of 16x16 (16 x f64x4 accumulators = 128 VGPRs).  Both operands are K-major row panels, so one
// staging pattern serves A and B: 16 k-rows x 128 columns, coalesced 512 B per wave per row.
// v_mfma_f64_16x16x4_f64 lane map: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)+4*reg][col=l&15].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock, 2) void ldlt_update_kernel(double* __restrict__ A, int64_t lda, int N,
                                                                const double* __restrict__ V, int64_t ldv, int vrow0,
                                                                int urow0, int K, int s, int row_end, int xcd_map)
{
  int ti, tj;
  if(xcd_map) {
    // 1-D grid over the upper-triangular tile domain, XCD-aware: workgroup b runs on XCD b%8 (observed
    // dispatch order; a different placement only costs speed).  The tile domain is cut into 8x8-tile
    // super-tiles; super-tile k goes to XCD k%8 and the 64 workgroups an XCD holds at a time (32 CUs x 2)
    // walk ONE super-tile, so its 8+8 panel blocks are fetched once into that XCD's L2 and shared.
    const int T = xcd_map;                 // tiles per side
    const int Sside = (T + 7) >> 3;        // super-tiles per side
    const int nS = Sside * (Sside + 1) / 2;
    const int b = blockIdx.x;
    const int xcd = b & 7, q = b >> 3;
    const int k = (q >> 6) * 8 + xcd;
    if(k >= nS) return;
    // k -> (Si, Sj), Sj >= Si, rows of the triangle have lengths Sside, Sside-1, ...
    int Si = (int)((2.0 * Sside + 1.0 - sqrt((2.0 * Sside + 1.0) * (2.0 * Sside + 1.0) - 8.0 * k)) * 0.5);
    while(Si > 0 && Si * Sside - Si * (Si - 1) / 2 > k) --Si;
    while((Si + 1) * Sside - (Si + 1) * Si / 2 <= k) ++Si;
    const int Sj = Si + (k - (Si * Sside - Si * (Si - 1) / 2));
    const int t = q & 63;
    ti = Si * 8 + (t >> 3);
    tj = Sj * 8 + (t & 7);
    if(ti >= T || tj >= T) return;
  } else {
    ti = blockIdx.y;
    tj = blockIdx.x;
  }
  if(tj < ti) return;
  const int r0 = s + ti * LD_TM, c0 = s + tj * LD_TN;
  if(r0 >= row_end || c0 >= N) return;
  __shared__ double Vs[LD_KT][LD_LDP];
  __shared__ double Us[LD_KT][LD_LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lk = lane >> 4, li = lane & 15;

  double4_t acc[4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};

  const int lcol = tid & 127, lrow = tid >> 7;
  const bool vr_ok = (r0 + lcol) < N;
  const bool uc_ok = (c0 + lcol) < N;
  const double* Vp = V + (int64_t)(vrow0 + lrow) * ldv + (r0 + lcol);
  const double* Up = A + (int64_t)(urow0 + lrow) * lda + (c0 + lcol);

  // register-prefetch pipeline over one LDS buffer: the global loads of stage kt+1 are in flight while the
  // 64 MFMAs of stage kt issue
  double vreg[8], ureg[8];
#pragma unroll
  for(int q = 0; q < 8; ++q) {
    vreg[q] = vr_ok ? Vp[(int64_t)(2 * q) * ldv] : 0.0;
    ureg[q] = uc_ok ? Up[(int64_t)(2 * q) * lda] : 0.0;
  }
  for(int kt = 0; kt < K; kt += LD_KT) {
    __syncthreads();  // previous stage fully consumed
#pragma unroll
    for(int q = 0; q < 8; ++q) {
      Vs[2 * q + lrow][lcol] = vreg[q];
      Us[2 * q + lrow][lcol] = ureg[q];
    }
    __syncthreads();
    if(kt + LD_KT < K) {
#pragma unroll
      for(int q = 0; q < 8; ++q) {
        vreg[q] = vr_ok ? Vp[(int64_t)(kt + LD_KT + 2 * q) * ldv] : 0.0;
        ureg[q] = uc_ok ? Up[(int64_t)(kt + LD_KT + 2 * q) * lda] : 0.0;
      }
    }
#pragma unroll
    for(int kk = 0; kk < LD_KT / 4; ++kk) {
      double a[4], b[4];
#pragma unroll
      for(int i = 0; i < 4; ++i) a[i] = Vs[kk * 4 + lk][wr * 64 + i * 16 + li];
#pragma unroll
      for(int j = 0; j < 4; ++j) b[j] = Us[kk * 4 + lk][wc * 64 + j * 16 + li];
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: C -= acc on the upper triangle.  All 16 loads of a row-group are issued before the first
  // store (the compiler cannot reorder a load above a possibly-aliasing store, which would otherwise turn
  // the 64 read-modify-writes into 64 serialized memory round trips).
#pragma unroll
  for(int i = 0; i < 4; ++i) {
    double cv[4][4];
    bool ok[4][4];
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = r0 + wr * 64 + i * 16 + lk + 4 * reg;
      const double* Crow = A + (int64_t)row * lda;
#pragma unroll
      for(int j = 0; j < 4; ++j) {
        const int col = c0 + wc * 64 + j * 16 + li;
        ok[reg][j] = (row < row_end) && (col < N) && (col >= row);
        cv[reg][j] = ok[reg][j] ? Crow[col] : 0.0;
      }
    }
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = r0 + wr * 64 + i * 16 + lk + 4 * reg;
      double* Crow = A + (int64_t)row * lda;
#pragma unroll
      for(int j = 0; j < 4; ++j) {
        const int col = c0 + wc * 64 + j * 16 + li;
        if(ok[reg][j]) Crow[col] = cv[reg][j] - acc[i][j][reg];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// inertia from D (thresholds of the reference's LAPACK path, hiopLinSolverSymDenseLapack.hpp:154-161:
// d < -1e-14 negative, |d| < 1e-14 null, else positive)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ldlt_inertia_kernel(int N, const double* __restrict__ A, int64_t lda,
                                                              int* __restrict__ out3)
{
  int pos = 0, neg = 0, nul = 0;
  for(int i = threadIdx.x; i < N; i += kBlock) {
    const double d = A[(int64_t)i * lda + i];
    if(d < -1e-14) ++neg;
    else if(d < 1e-14) ++nul;   // includes NaN? no: NaN compares false twice -> counted below
    else if(d >= 1e-14) ++pos;
    else ++nul;                 // NaN pivot: treat as null (singular)
  }
  __shared__ int sm[3][kBlock / 64];
  for(int off = 32; off > 0; off >>= 1) {
    pos += __shfl_down(pos, off, 64);
    neg += __shfl_down(neg, off, 64);
    nul += __shfl_down(nul, off, 64);
  }
  if((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = pos;
    sm[1][threadIdx.x >> 6] = neg;
    sm[2][threadIdx.x >> 6] = nul;
  }
  __syncthreads();
  if(threadIdx.x < 3) {
    int v = 0;
    for(int w = 0; w < kBlock / 64; ++w) v += sm[threadIdx.x][w];
    out3[threadIdx.x] = v;
  }
}

// ------------------------------------------------------------------------------------------
// triangular solves  U^T y = b,  z = D^-1 y,  U x = z   with 64-row blocks, one launch per block step.
// The sequential part (the 64x64 unit-triangular solve) is taken off every workgroup's critical path by
// LOOK-AHEAD: the launch that applies block I's solution to the rest of the right-hand side also solves
// the NEXT diagonal block inside the one workgroup that owns its entries (its 64 factor loads are issued
// at kernel entry, independent of everything else), so each launch starts with its block solution
// already in memory.  Per-step latency = one round of loads + a 64-step shuffle/FMA chain.
// ------------------------------------------------------------------------------------------
// lane r of one wave holds rhs entry r; returns y_r of the unit-lower solve with L = U_bb^T, where
// u[s] = U[j0+s][j0+r] for s < r (zero otherwise)
__device__ __forceinline__ double wave_fwd_chain(const double (&u)[LD_nb], double v)
{
#pragma unroll
  for(int s2 = 0; s2 < LD_nb; ++s2) {
    const double ys = bcast_lane(v, s2);
    v = fma(-u[s2], ys, v);
  }
  return v;
}

__global__ __launch_bounds__(64) void ldlt_fwd_first(const double* __restrict__ A, int64_t lda, int ib,
                                                     const double* __restrict__ b, double* __restrict__ y)
{
  const int lane = threadIdx.x;
  double u[LD_nb];
#pragma unroll
  for(int s2 = 0; s2 < LD_nb; ++s2) {
    const int rr = (s2 < ib) ? s2 : (ib - 1);
    const int cc = (lane < ib) ? lane : (ib - 1);
    u[s2] = A[(int64_t)rr * lda + cc];
  }
#pragma unroll
  for(int s2 = 0; s2 < LD_nb; ++s2) {
    asm volatile("" : "+v"(u[s2]));
    u[s2] = (s2 < lane && lane < ib) ? u[s2] : 0.0;
  }
  double v = (lane < ib) ? b[lane] : 0.0;
  v = wave_fwd_chain(u, v);
  if(lane < ib) y[lane] = v;
}

// y_I (block [i0,i0+64)) is ready in y; apply it to b[col], col >= i0+64; workgroup 0 owns the next
// diagonal block [j0, j0+jb) and solves it.
__global__ __launch_bounds__(64) void ldlt_fwd_step(const double* __restrict__ A, int64_t lda, int N, int i0,
                                                    double* __restrict__ b, double* __restrict__ y)
{
  const int lane = threadIdx.x;
  const int j0 = i0 + LD_nb;
  const int jb = (N - j0 < LD_nb) ? (N - j0) : LD_nb;
  const bool spine = (blockIdx.x == 0);
  const double yI = y[i0 + lane];   // issued first: vmcnt retires in order, so this wait does not cover the loads below
  double u[LD_nb];
  if(spine) {
#pragma unroll
    for(int s2 = 0; s2 < LD_nb; ++s2) {
      // unconditional (clamped) loads, all 64 in flight; the select happens after the opaque barrier below
      // (a predicated load per element compiles to 64 branches, each followed by a full vmcnt wait)
      const int rr = (j0 + s2 < N) ? (j0 + s2) : (N - 1);
      const int cc = (j0 + lane < N) ? (j0 + lane) : (N - 1);
      u[s2] = A[(int64_t)rr * lda + cc];
    }

  }
  __shared__ double ysh[LD_nb];
  ysh[lane] = yI;
  __syncthreads();
  const int64_t col = (int64_t)j0 + (int64_t)blockIdx.x * 64 + lane;
  double acc = 0.0;
  if(col < N) {
    acc = b[col];
    const double* Ac = A + (int64_t)i0 * lda + col;
    // 64 independent loads in flight (4 batches of 16 staged in registers) before the FMA chain
#pragma unroll
    for(int sb = 0; sb < LD_nb; sb += 16) {
      double av[16];
#pragma unroll
      for(int q = 0; q < 16; ++q) av[q] = Ac[(int64_t)(sb + q) * lda];
#pragma unroll
      for(int q = 0; q < 16; ++q) acc = fma(-av[q], ysh[sb + q], acc);
    }
    b[col] = acc;
  }
  if(spine) {
    // the 64 factor loads issued at kernel entry are consumed only here
#pragma unroll
    for(int s2 = 0; s2 < LD_nb; ++s2) {
      asm volatile("" : "+v"(u[s2]));
      u[s2] = (s2 < lane && lane < jb) ? u[s2] : 0.0;
    }
    const double v = wave_fwd_chain(u, acc);   // acc == 0 for lanes >= jb
    if(lane < jb) y[j0 + lane] = v;
  }
}

// backward: wave 0 solves the unit-upper 64x64 (or ib x ib) block held in LDS (strictly upper, zero padded)
__device__ __forceinline__ double wave_bwd_chain(const double (*S)[LD_nb + 1], int lane, double v)
{
  double srow[LD_nb];
#pragma unroll
  for(int c = 0; c < LD_nb; ++c) srow[c] = S[lane][c];
#pragma unroll
  for(int c = LD_nb - 1; c >= 0; --c) {
    const double xc = bcast_lane(v, c);
    v = fma(-srow[c], xc, v);   // srow[c] == 0 unless c > lane
  }
  return v;
}

__global__ __launch_bounds__(kBlock) void ldlt_bwd_first(const double* __restrict__ A, int64_t lda, int i0, int ib,
                                                         const double* __restrict__ z, double* __restrict__ x)
{
  __shared__ double S[LD_nb][LD_nb + 1];
  const int tid = threadIdx.x;
  {
    double sv[LD_nb * LD_nb / kBlock];
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      const int r = e >> 6, c = e & 63;
      const int rr = (r < ib) ? r : (ib - 1), cc = (c < ib) ? c : (ib - 1);
      const double t = A[(int64_t)(i0 + rr) * lda + (i0 + cc)];
      sv[q] = (r < ib && c < ib && c > r) ? t : 0.0;
    }
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      S[e >> 6][e & 63] = sv[q];
    }
  }
  __syncthreads();
  if(tid < 64) {
    double v = (tid < ib) ? z[i0 + tid] : 0.0;
    v = wave_bwd_chain(S, tid, v);
    if(tid < ib) x[i0 + tid] = v;
  }
}

// x_I (block [i0, i0+ib)) is ready in x; rows h < i0:  z[h] -= U[h][I] . x_I  (4 lanes per row, 16 columns
// = 128 bytes each).  Workgroup 0 owns the 64 rows of block I-1 and solves that diagonal block.
__global__ __launch_bounds__(kBlock) void ldlt_bwd_step(const double* __restrict__ A, int64_t lda, int i0, int ib,
                                                        double* __restrict__ z, double* __restrict__ x)
{
  __shared__ double S[LD_nb][LD_nb + 1];
  __shared__ double xs[LD_nb];
  __shared__ double zs[LD_nb];
  const int tid = threadIdx.x;
  const bool spine = (blockIdx.x == 0);
  const int p0 = i0 - LD_nb;  // previous diagonal block (always a full one)
  if(spine) {
    double sv[LD_nb * LD_nb / kBlock];
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      const int r = e >> 6, c = e & 63;
      const double t = A[(int64_t)(p0 + r) * lda + (p0 + c)];
      sv[q] = (c > r) ? t : 0.0;
    }
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      S[e >> 6][e & 63] = sv[q];
    }
  }
  if(tid < LD_nb) xs[tid] = (tid < ib) ? x[i0 + tid] : 0.0;
  __syncthreads();
  const int rloc = tid >> 2, sub = tid & 3;
  const int64_t h = (int64_t)i0 - (int64_t)LD_nb * (blockIdx.x + 1) + rloc;
  double acc = 0.0;
  if(h >= 0) {
    const double* Ah = A + h * lda + i0 + sub * 16;
    double av[16];
#pragma unroll
    for(int q = 0; q < 16; ++q) av[q] = (sub * 16 + q < ib) ? Ah[q] : 0.0;
#pragma unroll
    for(int q = 0; q < 16; ++q) acc = fma(av[q], xs[sub * 16 + q], acc);
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  double znew = 0.0;
  if(h >= 0 && sub == 0) {
    znew = z[h] - acc;
    z[h] = znew;
  }
  if(spine) {
    if(sub == 0) zs[rloc] = znew;
    __syncthreads();
    if(tid < 64) {
      const double v = wave_bwd_chain(S, tid, zs[tid]);
      x[p0 + tid] = v;
    }
  }
}

}  // namespace hiopamd

using namespace hiopamd;

// optional per-launch timing of the MFMA update kernel (HIP events on the launch stream)
struct LdltProfile {
  bool enabled = false;
  std::vector<hipEvent_t> pool;   // start/stop pairs
  size_t used = 0;
  double flops = 0.0;             // algorithmic flops of the timed update launches (accumulated)
  double ms = 0.0;                // accumulated update-kernel time
  long launches = 0;
  hipEvent_t get()
  {
    if(used == pool.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
  void collect()
  {
    for(size_t i = 0; i + 1 < used; i += 2) {
      float t = 0.f;
      if(hipEventElapsedTime(&t, pool[i], pool[i + 1]) == hipSuccess) ms += t;
    }
    used = 0;
  }
  ~LdltProfile()
  {
    for(auto e : pool) (void)hipEventDestroy(e);
  }
};

// algorithmic flops of one update launch: 2*K per updated element (r in [s,row_end), c in [r,N))
static double update_flops(int N, int K, int s, int row_end)
{
  const double rows = (double)(row_end - s);
  const double first = (double)(N - s), last = (double)(N - row_end + 1);
  return 2.0 * K * rows * (first + last) * 0.5;
}

struct hiopamd_linsolver {
  hiopamd_ctx* ctx = nullptr;
  int n = 0;
  double* M = nullptr;      // n x n row-major
  double* dinv = nullptr;   // n
  double* V = nullptr;      // LD_NB x n workspace
  double* ybuf = nullptr;   // n
  double* Dblk = nullptr;   // ceil(n/64) staged 64x64 diagonal blocks
  int* d_info = nullptr;    // [0]=zero-pivot flag, [1..3]=pos,neg,zero
  bool factored = false;
  int inertia[3] = {0, 0, 0};
  LdltProfile prof;
};

static int ldlt_factor_impl(hiopamd_ctx* ctx, int N, double* A, int64_t lda, double* dinv, double* V, double* Dblk,
                            int* d_info, int* inertia3_host, LdltProfile* prof = nullptr)
{
  // Dblk: per 64-row panel a compact 64x64 copy of the factored diagonal block, followed (after all
  // the blocks) by the per-panel 4 x 16x16 inverses
  double* Li = Dblk + (int64_t)((N + LD_nb - 1) / LD_nb) * (LD_nb * LD_nb);
  const bool timed = prof && prof->enabled;
  auto launch_update = [&](dim3 grid, int vrow0, int urow0, int K, int s, int row_end) {
    if(timed) (void)hipEventRecord(prof->get(), ctx->stream);
    int xcd_map = 0;
    if(row_end == N && grid.x == grid.y && grid.x >= 16) {
      // square trailing update: XCD-aware 1-D launch over 8x8-tile super-tiles
      xcd_map = (int)grid.x;
      const int Sside = (xcd_map + 7) / 8;
      const int nS = Sside * (Sside + 1) / 2;
      const int per_xcd = (nS + 7) / 8;           // super-tiles per XCD
      grid = dim3((unsigned)(per_xcd * 64 * 8), 1, 1);
    }
    hipLaunchKernelGGL(ldlt_update_kernel, grid, dim3(kBlock), 0, ctx->stream, A, lda, N, V, (int64_t)N, vrow0, urow0, K, s,
                       row_end, xcd_map);
    if(timed) {
      (void)hipEventRecord(prof->get(), ctx->stream);
      prof->flops += update_flops(N, K, s, row_end);
      prof->launches += 1;
    }
  };
  if(N < 0 || lda < N) return HIOPAMD_ERR_ARG;
  if(N == 0) {
    if(inertia3_host) inertia3_host[0] = inertia3_host[1] = inertia3_host[2] = 0;
    return HIOPAMD_OK;
  }
  hipStream_t st = ctx->stream;
  HIOPAMD_CHECK(hipMemsetAsync(d_info, 0, 4 * sizeof(int), st));
  const int64_t ldv = N;
  for(int K0 = 0; K0 < N; K0 += LD_NB) {
    const int Kend = (K0 + LD_NB < N) ? K0 + LD_NB : N;
    for(int k0 = K0; k0 < Kend; k0 += LD_nb) {
      const int kb = (k0 + LD_nb <= N) ? LD_nb : (N - k0);
      const int ncols = N - k0 - kb;
      double* Dk = Dblk + (int64_t)(k0 / LD_nb) * (LD_nb * LD_nb);
      double* Lik = Li + (int64_t)(k0 / LD_nb) * (4 * LD_SB * LD_SB);
      hipLaunchKernelGGL(ldlt_diag_kernel, dim3(1), dim3(kBlock), 0, st, A, lda, k0, kb, dinv, Dk, Lik, d_info);
      if(ncols > 0) {
        // a partial panel (kb < 64) can only be the last one, which has no columns to its right
        const int g = (ncols + 63) / 64;
        hipLaunchKernelGGL(ldlt_trsm_kernel, dim3(g), dim3(64), 0, st, A, lda, N, k0, V, ldv, k0 - K0, dinv, Dk, Lik);
      }
      if(k0 + kb < Kend) {
        // rows of the super-panel below this panel
        const int s = k0 + kb;
        const int tr = (Kend - s + LD_TM - 1) / LD_TM, tc = (N - s + LD_TN - 1) / LD_TN;
        launch_update(dim3(tc, tr), k0 - K0, k0, kb, s, Kend);
      }
    }
    if(Kend < N) {
      const int s = Kend;
      const int t = (N - s + LD_TM - 1) / LD_TM;
      launch_update(dim3(t, t), 0, K0, Kend - K0, s, N);
    }
  }
  hipLaunchKernelGGL(ldlt_inertia_kernel, dim3(1), dim3(kBlock), 0, st, N, A, lda, d_info + 1);
  HIOPAMD_CHECK(hipGetLastError());
  int h[4];
  HIOPAMD_CHECK(hipMemcpyAsync(h, d_info, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIOPAMD_CHECK(hipStreamSynchronize(st));
  if(timed) prof->collect();
  if(inertia3_host) {
    inertia3_host[0] = h[1];
    inertia3_host[1] = h[2];
    inertia3_host[2] = h[3];
  }
  if(h[0] != 0) return HIOPAMD_ERR_SINGULAR;
  return HIOPAMD_OK;
}

static int ldlt_solve_impl(hiopamd_ctx* ctx, int N, const double* A, int64_t lda, const double* dinv, double* ybuf,
                           double* rhs, int nrhs)
{
  if(N < 0 || nrhs < 0) return HIOPAMD_ERR_ARG;
  if(N == 0) return HIOPAMD_OK;
  hipStream_t st = ctx->stream;
  const int nblk = (N + LD_nb - 1) / LD_nb;
  for(int j = 0; j < nrhs; ++j) {
    double* b = rhs + (int64_t)j * N;
    // forward: U^T y = b
    {
      const int ib0 = (N < LD_nb) ? N : LD_nb;
      hipLaunchKernelGGL(ldlt_fwd_first, dim3(1), dim3(64), 0, st, A, lda, ib0, b, ybuf);
      for(int I = 0; I + 1 < nblk; ++I) {
        const int i0 = I * LD_nb;
        const int g = (N - i0 - LD_nb + 63) / 64;
        hipLaunchKernelGGL(ldlt_fwd_step, dim3(g), dim3(64), 0, st, A, lda, N, i0, b, ybuf);
      }
    }
    // z = D^-1 y
    int rc = hiopamd_vec_component_mult(ctx, N, ybuf, dinv);
    if(rc != HIOPAMD_OK) return rc;
    // backward: U x = z   (x written into b)
    {
      const int iL = (nblk - 1) * LD_nb;
      hipLaunchKernelGGL(ldlt_bwd_first, dim3(1), dim3(kBlock), 0, st, A, lda, iL, N - iL, ybuf, b);
      for(int I = nblk - 1; I >= 1; --I) {
        const int i0 = I * LD_nb;
        const int ib = (i0 + LD_nb <= N) ? LD_nb : (N - i0);
        hipLaunchKernelGGL(ldlt_bwd_step, dim3(I), dim3(kBlock), 0, st, A, lda, i0, ib, ybuf, b);
      }
    }
  }
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

extern "C" {

int hiopamd_ldlt_factor(hiopamd_ctx* ctx, int n, double* A, int64_t lda, double* work_dinv, int* inertia3_host)
{
  // workspace: V panel (LD_NB x n) + info flags from the context's grow-only buffer
  const size_t nn = (size_t)(n > 0 ? n : 1);
  const size_t vbytes = sizeof(double) * (size_t)LD_NB * nn;
  const size_t dbytes = sizeof(double) * (size_t)(LD_nb * LD_nb + 4 * LD_SB * LD_SB) * ((nn + LD_nb - 1) / LD_nb);
  char* w = (char*)ctx_workspace(ctx, vbytes + dbytes + 64);
  return ldlt_factor_impl(ctx, n, A, lda, work_dinv, (double*)w, (double*)(w + vbytes), (int*)(w + vbytes + dbytes),
                          inertia3_host);
}

int hiopamd_ldlt_solve(hiopamd_ctx* ctx, int n, const double* A, int64_t lda, const double* work_dinv,
                       double* rhs_inout, int nrhs)
{
  double* y = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)(n > 0 ? n : 1));
  return ldlt_solve_impl(ctx, n, A, lda, work_dinv, y, rhs_inout, nrhs);
}

int hiopamd_linsolver_create(hiopamd_linsolver** out, hiopamd_ctx* ctx, int n)
{
  if(!out || !ctx || n < 0) return HIOPAMD_ERR_ARG;
  hiopamd_linsolver* ls = new hiopamd_linsolver();
  ls->ctx = ctx;
  ls->n = n;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  HIOPAMD_CHECK(hipMalloc((void**)&ls->M, sizeof(double) * nn * nn));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->dinv, sizeof(double) * nn));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->V, sizeof(double) * nn * LD_NB));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->ybuf, sizeof(double) * nn));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->Dblk, sizeof(double) * (LD_nb * LD_nb + 4 * LD_SB * LD_SB) * ((nn + LD_nb - 1) / LD_nb)));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->d_info, 64));
  HIOPAMD_CHECK(hipMemsetAsync(ls->M, 0, sizeof(double) * nn * nn, ctx->stream));
  *out = ls;
  return HIOPAMD_OK;
}

int hiopamd_linsolver_destroy(hiopamd_linsolver* ls)
{
  if(!ls) return HIOPAMD_OK;
  (void)hipStreamSynchronize(ls->ctx->stream);
  (void)hipFree(ls->M);
  (void)hipFree(ls->dinv);
  (void)hipFree(ls->V);
  (void)hipFree(ls->ybuf);
  (void)hipFree(ls->Dblk);
  (void)hipFree(ls->d_info);
  delete ls;
  return HIOPAMD_OK;
}

double* hiopamd_linsolver_sys_matrix(hiopamd_linsolver* ls) { return ls ? ls->M : nullptr; }
int hiopamd_linsolver_n(const hiopamd_linsolver* ls) { return ls ? ls->n : -1; }

int hiopamd_linsolver_matrix_changed(hiopamd_linsolver* ls, int* n_neg_host)
{
  if(!ls || !n_neg_host) return HIOPAMD_ERR_ARG;
  ls->factored = false;
  int rc = ldlt_factor_impl(ls->ctx, ls->n, ls->M, ls->n, ls->dinv, ls->V, ls->Dblk, ls->d_info, ls->inertia, &ls->prof);
  if(rc == HIOPAMD_ERR_SINGULAR) {
    // reference: "entry in the factorization's diagonal is exactly zero" -> matrixChanged() returns -1
    *n_neg_host = -1;
    return HIOPAMD_OK;
  }
  if(rc != HIOPAMD_OK) return rc;
  ls->factored = true;
  *n_neg_host = (ls->inertia[2] > 0) ? -1 : ls->inertia[1];
  return HIOPAMD_OK;
}

int hiopamd_linsolver_solve(hiopamd_linsolver* ls, double* rhs_inout, int nrhs)
{
  if(!ls || !rhs_inout) return HIOPAMD_ERR_ARG;
  if(!ls->factored) return HIOPAMD_ERR_STATE;
  return ldlt_solve_impl(ls->ctx, ls->n, ls->M, ls->n, ls->dinv, ls->ybuf, rhs_inout, nrhs);
}

int hiopamd_linsolver_profile(hiopamd_linsolver* ls, int enable)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  ls->prof.enabled = enable != 0;
  ls->prof.flops = 0.0;
  ls->prof.ms = 0.0;
  ls->prof.launches = 0;
  return HIOPAMD_OK;
}
int hiopamd_linsolver_profile_read(const hiopamd_linsolver* ls, double* update_ms_host, double* update_flops_host,
                                   int64_t* update_launches_host)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(update_ms_host) *update_ms_host = ls->prof.ms;
  if(update_flops_host) *update_flops_host = ls->prof.flops;
  if(update_launches_host) *update_launches_host = ls->prof.launches;
  return HIOPAMD_OK;
}

int hiopamd_linsolver_inertia(const hiopamd_linsolver* ls, int* pos, int* neg, int* zero)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(pos) *pos = ls->inertia[0];
  if(neg) *neg = ls->inertia[1];
  if(zero) *zero = ls->inertia[2];
  return HIOPAMD_OK;
}

}  // extern "C"
