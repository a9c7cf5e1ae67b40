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

#include <algorithm>
#include <chrono>
#include <fcntl.h>
#include <mutex>
#include <sys/file.h>
#include <unistd.h>
#include <map>
#include <string>
#include <vector>
#include <type_traits>

namespace hiopamd {

constexpr int LD_NB = 256;   // super-panel rows (K of the trailing update)
constexpr int LD_nb = 64;    // panel rows
constexpr int LD_PAD_MIN = 1024;   // orders from here on may be factored at a padded order (hiopamd_linsolver::npad, ldlt_padded_order)
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

// LDL^T of a kb x kb (<= 64) block resident in LDS (upper part, zero padded; sdinv preset to 1), blocked by 16:
//   (i)   wave 0 factors the 16x16 diagonal sub-block entirely in registers (lane c = column c; pivot row and
//         multipliers broadcast with v_readlane; v_rcp + one Newton step) and, Gauss-Jordan style, applies the same
//         row operations to an identity: the unit-lower inverse Linv16 comes out of the same 16 pivots for free;
//   (ii)  the 16 x (rest) row panel is V = Linv16 * A on fp64 MFMA (one 16-column group per wave);
//   (iii) the trailing part gets the rank-16 update C -= V^T D^-1 V on fp64 MFMA (6 upper 16x16 tiles over 4 waves).
// During the factorisation S holds UN-scaled rows (v_kc = d_k u_kc); they are scaled by diag_emit.
// Li (global, [sb][i][j] row-major) receives the four 16x16 inverses; info[0] the first zero / non-finite pivot.
// inter-workgroup data of the dataflow factorisation moves with agent-scope relaxed atomics (`sc1`: write-through stores,
// L1-bypassing loads) — the form MI355X_MICROARCH.md validates for hand-offs between CUs / XCDs inside a launch
__device__ __forceinline__ double ldg_sc1(const double* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stg_sc1(double* p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct DiagNoIdle {
  __device__ __forceinline__ void operator()(int) const {}
};
// MF = true: the 16 x 16 sub-block factor as 15 rank-1 MFMA updates (factor16m below); false: the v_readlane / v_fma form.
// `idle(w)` is called by the waves 1-3 while wave 0 factors the LAST 16 x 16 sub-block alone (they have nothing to do then):
// the spine of the dataflow factorisation prefetches its next tiles there.
template <bool SC1 = false, bool MF = true, class Idle = DiagNoIdle>
__device__ __forceinline__ void diag_factor_lds(double (*S)[LD_nb + 1], double* sdinv, int kb, int k0, int* info,
                                                double* __restrict__ Li, int tid, double* Li_lds = nullptr, Idle idle = Idle(),
                                                unsigned* prof = nullptr)
{
  __shared__ double Lv[LD_SB][LD_SB + 1];
  const int lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
  // (i') the same factor on the matrix pipe.  The block and the inverse being built live in the accumulator layout of
  // v_mfma_f64_16x16x4_f64 (lane (li, g), register reg <-> row g + 4 reg, column li): pivot row k = 4 ks + kg is register ks of
  // lane group kg, which is exactly where the B operand of k-slot kg is read from, and lane (r, kg) holds a[k][r], the A operand
  // of row r.  One MFMA with the other three k-slots zero is the rank-1 update  x[r][:] -= a[k][r] * (x[k][:] / d_k)  of all
  // rows r > k: 2 v_readlane + the reciprocal + 5 VALU operations + 2 MFMAs per pivot instead of ~60 VALU instructions (the
  // v_readlane form below spends 3 instructions per row and pivot: ~1000 per sub-block, 2-3 us of the spine each).
  auto factor16m = [&](int sb) {
    const int o = sb * LD_SB;
    if(o >= kb) {
#pragma unroll
      for(int q = 0; q < 4; ++q) {
        const int e = tid + 64 * q;
        const double idv = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        if constexpr(SC1) stg_sc1(Li + sb * 256 + e, idv);
        else Li[sb * 256 + e] = idv;
        if(Li_lds) Li_lds[sb * 256 + e] = idv;
      }
      return;
    }
    double4_t xa, xm;
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      xa[reg] = S[o + g + 4 * reg][o + li];   // zeros below the diagonal
      xm[reg] = (g + 4 * reg == li) ? 1.0 : 0.0;
    }
    // Timing of one pivot on the lone wave (measured, profiles/r02_probes): a v_mfma_f64_16x16x4_f64 holds the matrix pipe for
    // 64 cycles and delivers after ~76; a dependent VALU instruction issues every ~8 cycles.  Written pivot by pivot (pivot,
    // reciprocal, operands, the two MFMAs) the wave stalls 64 cycles at the second MFMA and then runs the whole reciprocal
    // chain of the next pivot behind it: ~225 cycles per pivot.  Here the loop is software-pipelined around the pipe:
    //     a(k)      update of the block by pivot k                         (issues as soon as its two operands exist)
    //     d(k+1)    the NEXT pivot, by the same fused multiply-add a(k) applies to that entry, from the values before the
    //               update; its reciprocal (v_rcp + Newton step)           (VALU work under a(k)'s 64 cycles in the pipe)
    //     m(k)      the same row operations on the inverse being built     (the pipe is free again by now)
    // so that after a(k) delivers only the two selects and one multiply of a(k+1)'s operands remain: the pivot loop runs at the
    // pipe's pace, two MFMAs = 128 cycles per pivot.  Zero / non-finite pivots are found afterwards from the reciprocals
    // (1/0 = inf, 1/inf -> NaN through the Newton step, NaN propagates), not tested pivot by pivot.
    int key[4];   // key[kg] = li in lane group kg, -1 elsewhere: lane holds a[k][r], r > k  <=>  key[k & 3] > k
#pragma unroll
    for(int q = 0; q < 4; ++q) key[q] = (g == q) ? li : -1;
    double dis[LD_SB];
#pragma unroll
    for(int k = 0; k < LD_SB; ++k) dis[k] = 1.0;
    double di = fast_rcp(bcast_lane(xa[0], 0));   // (kb > o: pivot 0 exists)
#pragma unroll
    for(int k = 0; k < LD_SB; ++k) {
      if(o + k < kb) {  // uniform
        const int ks = k >> 2, kg = k & 3;
        dis[k] = di;
        if(k + 1 < LD_SB) {
          const int k1 = k + 1;
          const double ndi = -di;
          const double aop = (key[kg] > k) ? xa[ks] : 0.0;   // zero outside lane group kg and for rows <= k: such a slot
          const double ba = xa[ks] * ndi;                     // cancels whatever B holds there, so B is not masked
          const double akk1 = bcast_lane(xa[ks], 16 * kg + k1);                 // a[k][k+1]
          const double ak1k1 = bcast_lane(xa[k1 >> 2], 16 * (k1 & 3) + k1);     // a[k+1][k+1] before the update
          xa = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, ba, xa, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          di = fast_rcp(fma(akk1, akk1 * ndi, ak1k1));
          const double bm = xm[ks] * ndi;
          __builtin_amdgcn_sched_barrier(0);
          xm = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bm, xm, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    int bad = 0;
#pragma unroll
    for(int k = LD_SB - 1; k >= 0; --k)
      if(o + k < kb && !isfinite(dis[k])) bad = k + 1;   // the FIRST one (everything after it is NaN as well)
    if(tid == 0) {
#pragma unroll
      for(int k = 0; k < LD_SB; ++k)
        if(o + k < kb) sdinv[o + k] = dis[k];
      if(bad) atomicCAS(info, 0, k0 + o + bad);
    }
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int r = g + 4 * reg;
      if(li >= r) S[o + r][o + li] = xa[reg];
      Lv[r][li] = xm[reg];
      if constexpr(SC1) stg_sc1(Li + sb * 256 + r * 16 + li, xm[reg]);
      else Li[sb * 256 + r * 16 + li] = xm[reg];
      if(Li_lds) Li_lds[sb * 256 + r * 16 + li] = xm[reg];
    }
  };
  // (i'') FOUR pivots per pass (round 5; full sub-blocks only).  The rank-1 form above is one dependent chain per pivot — MFMA (76
  // cycles), two v_readlane pairs, the next pivot and its reciprocal, the operand: ~300 cycles, 2.4 us per 16 x 16 sub-block, 40 % of every
  // spine step.  Here the 4 x 4 diagonal block of pivots k0..k0+3 (10 entries of register k0/4, one v_readlane pair each) is factored
  // in UNIFORM registers (every lane the same 4 x 4 LDL^T: four reciprocals, ~20 fused multiply-adds, no cross-lane traffic), the four
  // pivot rows R (register k0/4 of the four lane groups) become U = Linv4 R by ONE MFMA whose A operand holds Linv4 in lanes (i, k),
  // i < 4 — the result rows 0..3 are register 0 of lane group = row, i.e. every lane receives its own entry back —, and the rows below
  // get the rank-4 update x[r][:] -= sum_q (U_q[r] / d_q) U_q[:] by ONE MFMA (A operand: own entry times -1/d of the lane's group, masked
  // to r > k0 + 3; B operand: own entry).  The same two MFMAs applied to the inverse being built.  Per 4 pivots: 20 v_readlane, the
  // uniform 4 x 4 factor, 2 + 2 MFMAs — ~110 cycles per pivot instead of ~300.
  auto factor16r = [&](int sb) {
    const int o = sb * LD_SB;
    double4_t xa, xm;
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      xa[reg] = S[o + g + 4 * reg][o + li];   // zeros below the diagonal
      xm[reg] = (g + 4 * reg == li) ? 1.0 : 0.0;
    }
    double dis[LD_SB];
    const double4_t zero4 = double4_t{0.0, 0.0, 0.0, 0.0};
    // 0/1 weights of lane (li, g) for the entries of Linv4 it holds as A operand (wdg: the unit diagonal), and its lane group
    const double wdg = (li < 4 && li == g) ? 1.0 : 0.0, w10 = (li == 1 && g == 0) ? 1.0 : 0.0, w20 = (li == 2 && g == 0) ? 1.0 : 0.0,
                 w30 = (li == 3 && g == 0) ? 1.0 : 0.0, w21 = (li == 2 && g == 1) ? 1.0 : 0.0, w31 = (li == 3 && g == 1) ? 1.0 : 0.0,
                 w32 = (li == 3 && g == 2) ? 1.0 : 0.0;
    const bool is_g1 = g == 1, is_g2 = g == 2, is_g3 = g == 3;
#pragma unroll
    for(int s4 = 0; s4 < 4; ++s4) {
      const int k0 = 4 * s4;
      // the 4 x 4 diagonal block: entry (q, p), p >= q, lives in lane 16 q + k0 + p, register s4
      const double p00 = bcast_lane(xa[s4], k0), p01 = bcast_lane(xa[s4], k0 + 1), p02 = bcast_lane(xa[s4], k0 + 2),
                   p03 = bcast_lane(xa[s4], k0 + 3);
      const double p11 = bcast_lane(xa[s4], 16 + k0 + 1), p12 = bcast_lane(xa[s4], 16 + k0 + 2), p13 = bcast_lane(xa[s4], 16 + k0 + 3);
      const double p22 = bcast_lane(xa[s4], 32 + k0 + 2), p23 = bcast_lane(xa[s4], 32 + k0 + 3);
      const double p33 = bcast_lane(xa[s4], 48 + k0 + 3);
      // its LDL^T, uniform (q.. = the block after the earlier pivots of this pass)
      const double r0 = fast_rcp(p00);
      const double l10 = p01 * r0, l20 = p02 * r0, l30 = p03 * r0;
      const double q11 = fma(-l10, p01, p11), q12 = fma(-l10, p02, p12), q13 = fma(-l10, p03, p13);
      const double r1 = fast_rcp(q11);
      const double l21 = q12 * r1, l31 = q13 * r1;
      const double q22 = fma(-l21, q12, fma(-l20, p02, p22)), q23 = fma(-l21, q13, fma(-l20, p03, p23));
      const double r2 = fast_rcp(q22);
      const double l32 = q23 * r2;
      const double q33 = fma(-l32, q23, fma(-l31, q13, fma(-l30, p03, p33)));
      const double r3 = fast_rcp(q33);
      dis[k0] = r0;
      dis[k0 + 1] = r1;
      dis[k0 + 2] = r2;
      dis[k0 + 3] = r3;
      // inverse of the unit lower factor (column by column: L c = e)
      const double c10 = -l10, c21 = -l21, c32 = -l32;
      const double c20 = fma(-l21, c10, -l20);
      const double c30 = fma(-l32, c20, fma(-l31, c10, -l30));
      const double c31 = fma(-l32, c21, -l31);
      // A operand of the in-block transform: lane (li, g) <- Linv4[li][g].  At most one of the 0/1 lane weights is set per lane, so the
      // sum is a selection; written as a tree of fused multiply-adds (depth 3) instead of a chain of twelve dependent v_cndmask
      const double al = (fma(w10, c10, wdg) + fma(w20, c20, w30 * c30)) + (fma(w21, c21, w31 * c31) + w32 * c32);
      double rg = r0;
      rg = is_g1 ? r1 : rg;
      rg = is_g2 ? r2 : rg;
      rg = is_g3 ? r3 : rg;
      const double4_t u4 = __builtin_amdgcn_mfma_f64_16x16x4f64(al, xa[s4], zero4, 0, 0, 0);
      const double4_t m4 = __builtin_amdgcn_mfma_f64_16x16x4f64(al, xm[s4], zero4, 0, 0, 0);
      const double u = u4[0], mv = m4[0];   // this lane's entry of the finished pivot row of its group / of the same row of the inverse
      if(s4 < 3) {
        const double aop = (li > k0 + 3) ? u * (-rg) : 0.0;
        xa = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, u, xa, 0, 0, 0);
        xm = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, mv, xm, 0, 0, 0);
      }
      xa[s4] = u;
      xm[s4] = mv;
    }
    int bad = 0;
#pragma unroll
    for(int k = LD_SB - 1; k >= 0; --k)
      if(!isfinite(dis[k])) bad = k + 1;   // the FIRST one (everything after it is NaN as well)
    if(tid == 0) {
#pragma unroll
      for(int k = 0; k < LD_SB; ++k) sdinv[o + k] = dis[k];
      if(bad) atomicCAS(info, 0, k0 + o + bad);
    }
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int r = g + 4 * reg;
      if(li >= r) S[o + r][o + li] = xa[reg];
      Lv[r][li] = xm[reg];
      if constexpr(SC1) stg_sc1(Li + sb * 256 + r * 16 + li, xm[reg]);
      else Li[sb * 256 + r * 16 + li] = xm[reg];
      if(Li_lds) Li_lds[sb * 256 + r * 16 + li] = xm[reg];
    }
  };
  // (i) for sub-block sb, by wave 0 (call with tid < 64): in-register 16x16 Gauss-Jordan; a padded sub-block (o >= kb)
  // just gets the identity as its inverse
  auto factor16v = [&](int sb) {
    const int o = sb * LD_SB;
    if(o >= kb) {
#pragma unroll
      for(int q = 0; q < 4; ++q) {
        const int e = tid + 64 * q;
        const double idv = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        if constexpr(SC1) stg_sc1(Li + sb * 256 + e, idv);
        else Li[sb * 256 + e] = idv;
        if(Li_lds) Li_lds[sb * 256 + e] = idv;
      }
      return;
    }
    // lanes 0-15 carry column c of the block (`a`), lanes 16-31 column c of the inverse being built (`m`): the row operation
    // x[r] -= (S[k][r] / d) * x[k] is the same for both, so one fma per row serves the two halves at once
    const int c = li;
    const bool is_a = (g == 0);
    double x[LD_SB];
#pragma unroll
    for(int r = 0; r < LD_SB; ++r) x[r] = is_a ? S[o + r][o + c] /* zeros below the diagonal */ : ((r == c) ? 1.0 : 0.0);
    // reciprocal pivots and the first bad pivot are collected in registers and written once after the loop: a `tid == 0`
    // block per pivot cut the unrolled loop into 16 basic blocks (no scheduling across pivots)
    double dis[LD_SB];
    int bad = 0;
#pragma unroll
    for(int k = 0; k < LD_SB; ++k) {
      dis[k] = 1.0;
      if(o + k < kb) {  // uniform
        const double d = bcast_lane(x[k], k);          // pivot (lane k of the `a` half)
        const double di = fast_rcp(d);
        const double xkc = x[k] * di;                  // scaled pivot-row entry of my column
#pragma unroll
        for(int r = k + 1; r < LD_SB; ++r) {
          const double vkr = bcast_lane(x[k], r);      // S[k][r] (lane r of the `a` half): multiplier of row r is vkr/d
          x[r] = fma(-vkr, xkc, x[r]);
        }
        dis[k] = di;
        bad = (bad == 0 && (d == 0.0 || !isfinite(d))) ? (k + 1) : bad;
        // force the row operations of THIS pivot to be carried out here: left to itself the compiler sinks every fma to just
        // before its result is used (pivot r), keeping the broadcast multipliers (SGPR pairs from v_readlane) of all earlier
        // pivots alive — ~480 SGPRs, spilled lane by lane through v_writelane / v_readlane: the 16 x 16 factor took 3.5 us
#pragma unroll
        for(int r = k + 1; r < LD_SB; ++r) asm volatile("" : "+v"(x[r]));
      }
    }
    if(tid == 0) {
#pragma unroll
      for(int k = 0; k < LD_SB; ++k)
        if(o + k < kb) sdinv[o + k] = dis[k];
      if(bad) atomicCAS(info, 0, k0 + o + bad);
    }
    if(tid < 2 * LD_SB) {
#pragma unroll
      for(int r = 0; r < LD_SB; ++r) {
        if(is_a) {
          if(c >= r) S[o + r][o + c] = x[r];
        } else {
          Lv[r][c] = x[r];
          if constexpr(SC1) stg_sc1(Li + sb * 256 + r * 16 + c, x[r]);
          else Li[sb * 256 + r * 16 + c] = x[r];
          if(Li_lds) Li_lds[sb * 256 + r * 16 + c] = x[r];
        }
      }
    }
  };
  auto factor16 = [&](int sb) {
    const unsigned t0 = prof ? (unsigned)wall_clock64() : 0u;   // (profiling runs only: time spent in the sub-block factors)
    if constexpr(MF) {
      if(sb * LD_SB + LD_SB <= kb) factor16r(sb);   // (uniform) a full sub-block: four pivots per pass
      else factor16m(sb);
    } else factor16v(sb);
    if(prof && tid == 0) atomicAdd(prof, (unsigned)wall_clock64() - t0);
  };
  // one 16x16 tile of (iii): C -= V^T D^-1 V, (ti, tj) = 0:(0,0) 1:(0,1) 2:(0,2) 3:(1,1) 4:(1,2) 5:(2,2)
  auto update_tile = [&](int o, int t6) {
    const int ti = (t6 < 3) ? 0 : ((t6 < 5) ? 1 : 2);
    const int tj = (t6 < 3) ? t6 : ((t6 < 5) ? (t6 - 2) : 2);
    const int ro = o + LD_SB + 16 * ti, co = o + LD_SB + 16 * tj;
    if(co < LD_nb && ro < kb) {   // wave-uniform
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
      double aop[4], bop[4];
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) {
        const double dk = sdinv[o + 4 * kk + g];
        aop[kk] = S[o + 4 * kk + g][ro + li];
        bop[kk] = S[o + 4 * kk + g][co + li] * dk;
      }
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[kk], bop[kk], acc, 0, 0, 0);
#pragma unroll
      for(int reg = 0; reg < 4; ++reg) {
        const int r = ro + g + 4 * reg, c = co + li;
        if(c >= r) S[r][c] -= acc[reg];
      }
    }
  };
  if(tid < 64) factor16(0);
  __syncthreads();
  for(int sb = 0; sb + 1 < LD_nb / LD_SB; ++sb) {   // (the last sub-block has no row panel and no trailing part)
    const int o = sb * LD_SB;
    if(o >= kb) {   // uniform: nothing left (its inverse was set by the look-ahead below / above)
      if(tid < 64) factor16(sb + 1);
      else if(sb + 2 == LD_nb / LD_SB) idle(w);
      continue;
    }
    // ---- (ii) V = Linv16 * A_panel, one 16-column group per wave
    {
      const int cbase = o + LD_SB + 16 * w;
      if(cbase < LD_nb && cbase < kb) {   // wave-uniform
        double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
        double aop[4], bop[4];
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) {
          aop[kk] = Lv[li][4 * kk + g];
          bop[kk] = S[o + 4 * kk + g][cbase + li];
        }
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[kk], bop[kk], acc, 0, 0, 0);
#pragma unroll
        for(int reg = 0; reg < 4; ++reg) S[o + g + 4 * reg][cbase + li] = acc[reg];
      }
    }
    __syncthreads();
    // ---- (iii) rank-16 update of the trailing tiles, with LOOK-AHEAD: wave 0 updates the next diagonal tile and goes
    // straight on to factor it (i) while waves 1-3 update the other five tiles
    if(w == 0) {
      update_tile(o, 0);
      asm volatile("" ::: "memory");
      if(sb + 1 < LD_nb / LD_SB) factor16(sb + 1);
    } else if(w == 1) {
      update_tile(o, 1);
      update_tile(o, 4);
    } else if(w == 2) {
      update_tile(o, 2);
      update_tile(o, 5);
    } else {
      update_tile(o, 3);
    }
    if(sb + 2 == LD_nb / LD_SB && w != 0) idle(w);
    __syncthreads();
  }
}

// scale rows, write U11/D into A, the compact copy Dk and dinv
__device__ __forceinline__ void diag_emit(double (*S)[LD_nb + 1], const double* sdinv, int kb, int k0, double* A, int64_t lda,
                                          double* dinv, double* Dk, int tid)
{
  for(int e = tid; e < LD_nb * LD_nb; e += kBlock) {
    const int r = e >> 6, c = e & 63;
    double v = S[r][c];
    if(c > r) v *= sdinv[r];
    const bool in = (r < kb && c < kb && c >= r);
    Dk[e] = in ? v : 0.0;
    if(in) A[(int64_t)(k0 + r) * lda + (k0 + c)] = v;
  }
  if(tid < kb) dinv[k0 + tid] = sdinv[tid];
}

// ------------------------------------------------------------------------------------------
// Super-panel kernels (NB = 256 rows = 4 panels of 64): the whole panel chain of a super-panel in TWO launches.
//
// block_row_solve<P>: for ONE group of 16 columns held by one wave, the block-row P of the row panel,
//     T_P = A_P - sum_{q<P} L_Pq V_q            (64x64 by 64x16 products on fp64 MFMA; L_Pq = U_qP^T read from A)
//     V_P = L_PP^-1 T_P                         (16-row block substitution with the 16x16 inverses, as above)
// V_q (q < P) never leave the registers: the MFMA D layout of a 16x16 block of V_q is the B-operand layout of the
// next product (k-step kk <-> accumulator register kk), and the A-operand layout of the symmetric update below.
// BYPASS = true makes the loads of data produced earlier IN THE SAME LAUNCH by other waves go to L2
// (relaxed agent-scope atomic load = `sc1`): a CU's vector L1 is not refreshed by stores.
// ------------------------------------------------------------------------------------------
// Loads of block_row_solve: wave-scope relaxed atomics = ordinary loads (no cache-policy bits) that the compiler keeps
// where they are written.  As plain loads they were sunk next to their uses: one load, one s_waitcnt, one MFMA, 600 times
// over — every one a full memory round trip for the single wave of this workgroup.  Written as batches that are issued
// together, a block row costs a handful of round trips.
__device__ __forceinline__ double ld_batch(const double* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <int P, bool IDENT = false, bool SC1 = false>
__device__ __forceinline__ void block_row_solve(const double* A, int64_t lda, int K0, int kbs, int64_t col, bool col_ok,
                                                const double* Cd /* compact diagonal block, ld = LD_NB */,
                                                const double* Dk_sp, const double* Li_sp, double4_t (&Vv)[4][4], int g,
                                                int li)
{
  // ---- batch 0: the 64 x 16 block of the matrix, and the operands of the in-block substitution (independent of V)
  double4_t t[4];
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int r = 0; r < 4; ++r) {
      const int row = 64 * P + 16 * I + g + 4 * r;   // rows of block-row P that exist (the last super-panel may be ragged)
      if constexpr(IDENT) {   // right-hand side = identity (col = column inside the block): the result is L^-1
        t[I][r] = (row == (int)col && row < kbs) ? 1.0 : 0.0;
      } else {
        const int rowc = (row < kbs) ? row : (kbs - 1);
        // SC1: the rows were written by other workgroups of the same launch (dataflow factorisation) -> bypass L1
        const double v = SC1 ? ldg_sc1(A + (int64_t)(K0 + rowc) * lda + col) : ld_batch(A + (int64_t)(K0 + rowc) * lda + col);
        t[I][r] = (col_ok & (row < kbs)) ? v : 0.0;
      }
    }
  const double* Dk = Dk_sp + P * (LD_nb * LD_nb);
  const double* Li = Li_sp + P * (4 * LD_SB * LD_SB);
  double nl[6][4];   // pairs (I, J), J < I: (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
  double iv[4][4];
#pragma unroll
  for(int I = 1; I < 4; ++I)
#pragma unroll
    for(int J = 0; J < I; ++J)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) nl[I * (I - 1) / 2 + J][kk] = -ld_batch(Dk + (16 * J + 4 * kk + g) * LD_nb + 16 * I + li);
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) iv[I][kk] = ld_batch(Li + I * 256 + li * 16 + 4 * kk + g);
  // ---- T_P -= L_Pq V_q: one batch of 64 operand loads per q, then its 64 MFMAs
  // (accumulating the products with a plus sign and subtracting once was 17 us SLOWER per launch: the subtraction
  //  pulls the accumulators out of the AGPRs and back)
#pragma unroll
  for(int q = 0; q < P; ++q) {
    double Lop[4][4][4];  // [I][Jq][kk]: -L_Pq[16I+li][16Jq+4kk+g] = -U[K0+64q+16Jq+4kk+g][K0+64P+16I+li]
#pragma unroll
    for(int I = 0; I < 4; ++I)
#pragma unroll
      for(int Jq = 0; Jq < 4; ++Jq)
#pragma unroll
        for(int kk = 0; kk < 4; ++kk)
        {
          // IDENT (the inverse of a diagonal block, possibly the ragged last one): columns of the compact block at and behind kbs are
          // padding nobody writes for the blocks the trailing update fills (it stores the entries that exist) — they are rows of the
          // result that do not exist, but a NaN there reaches the rows that do through 0 x NaN in the 16-row substitution below
          const int cpad = 64 * P + 16 * I + li;
          const double lv = ld_batch(Cd + (64 * q + 16 * Jq + 4 * kk + g) * LD_NB + (IDENT ? (cpad < kbs ? cpad : 0) : cpad));
          Lop[I][Jq][kk] = (IDENT && cpad >= kbs) ? 0.0 : -lv;
        }
#pragma unroll
    for(int Jq = 0; Jq < 4; ++Jq)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk)
#pragma unroll
        for(int I = 0; I < 4; ++I)   // four independent accumulators back to back
          t[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lop[I][Jq][kk], Vv[q][Jq][kk], t[I], 0, 0, 0);
  }
  // ---- V_P = L_PP^-1 T_P by 16-row blocks
#pragma unroll
  for(int I = 0; I < 4; ++I) {
    double4_t u = t[I];
#pragma unroll
    for(int J = 0; J < 4; ++J) {
      if(J < I) {
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) u = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[I * (I - 1) / 2 + J][kk], Vv[P][J][kk], u, 0, 0, 0);
      }
    }
    double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[I][kk], u[kk], v, 0, 0, 0);
    Vv[P][I] = v;
  }
}

// store block-row P of V (un-scaled, workspace rows 64P..) and U = D^-1 V (in place)
template <int P, bool SC1 = false>
__device__ __forceinline__ void block_row_store(double* A, int64_t lda, double* V, int64_t ldv, int K0, int kbs, int64_t col,
                                                bool col_ok, const double* dinv_sp /*256 entries of this super-panel*/,
                                                const double4_t (&Vv)[4][4], int g, int li)
{
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int r = 0; r < 4; ++r) {
      const int row = 64 * P + 16 * I + g + 4 * r;
      if(row < kbs) {
        const double v = Vv[P][I][r];
        const double u = v * dinv_sp[row];
        if(col_ok) {
          if constexpr(SC1) {
            stg_sc1(V + (int64_t)row * ldv + col, v);
            stg_sc1(A + (int64_t)(K0 + row) * lda + col, u);
          } else {
            V[(int64_t)row * ldv + col] = v;
            A[(int64_t)(K0 + row) * lda + col] = u;
          }
        }
      }
    }
}

// cooperative (256 threads) load of a 64x64 block into S, bypassing L1 (data written earlier in this launch)
__device__ __forceinline__ void stage_block_bypass(double (*S)[LD_nb + 1], const double* src, int64_t ld, int tid)
{
  double sv[LD_nb * LD_nb / kBlock];
#pragma unroll
  for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
    const int e = tid + q * kBlock;
    sv[q] = __hip_atomic_load(src + (int64_t)(e >> 6) * ld + (e & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
    const int e = tid + q * kBlock;
    S[e >> 6][e & 63] = sv[q];
  }
}

// Phase (a) of the super-diagonal kernel for panel j, all block rows P < j, software-pipelined: the operand block of the
// NEXT stage (U_{q+1,P}, or Dk_P + the 16x16 inverses, or the first block of block row P+1) is in flight — 16 + 16 loads
// per thread, in registers — while the MFMAs of the current stage issue.  Un-pipelined, each of the 1 / 3 / 6 stages of
// panel 1 / 2 / 3 paid 2-3 us of exposed load latency (in-kernel stamps: 10 / 20 / 33 us for 64 / 192 / 384 MFMAs).
// Stage (P, q): q < P -> U_qP (from the compact block, rows 64q.., columns 64P..);  q == P -> Dk_P and Li_P.
__device__ __forceinline__ void panel_block_rows_pipelined(const int j, double* A, int64_t lda, int K0, int64_t col,
                                                           bool col_ok, int cl, const double* Dk_sp, const double* Li_sp,
                                                           const double* dall, int g, int li, double (*S)[LD_nb + 1],
                                                           double (*Vs)[80], int tid)
{
  if(j == 0) return;
  double sv[LD_nb * LD_nb / kBlock];   // the prefetched 64 x 64 block
  double liv[4][4];                    // the prefetched 16x16 inverses of a diagonal stage: [I][kk]
  auto issue = [&](int P, int q) {
    const double* src = (q < P) ? A + (int64_t)(K0 + 64 * q) * lda + (K0 + 64 * P) : Dk_sp + P * (LD_nb * LD_nb);
    const int64_t ld = (q < P) ? lda : (int64_t)LD_nb;
#pragma unroll
    for(int e4 = 0; e4 < LD_nb * LD_nb / kBlock; ++e4) {
      const int e = tid + e4 * kBlock;
      sv[e4] = __hip_atomic_load(src + (int64_t)(e >> 6) * ld + (e & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if(q == P) {
      const double* Li = Li_sp + P * (4 * LD_SB * LD_SB);
#pragma unroll
      for(int I = 0; I < 4; ++I)
#pragma unroll
        for(int kk = 0; kk < 4; ++kk)
          liv[I][kk] = __hip_atomic_load(Li + I * 256 + li * 16 + 4 * kk + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto drain_if_newest = [&](int P) {
    // only the newest block row needs what panel j-1 emitted (A rows, Dk, Li): drain the stores before its first load
    if(P == j - 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  drain_if_newest(0);
  issue(0, 0);
#pragma unroll 1
  for(int P = 0; P < j; ++P) {
    double4_t t[4];
#pragma unroll
    for(int I = 0; I < 4; ++I)
#pragma unroll
      for(int r = 0; r < 4; ++r) {
        const int row = 64 * P + 16 * I + g + 4 * r;   // block rows below the current panel are always full
        const double v = A[(int64_t)(K0 + row) * lda + col];
        t[I][r] = col_ok ? v : 0.0;
      }
#pragma unroll 1
    for(int q = 0; q <= P; ++q) {
      __syncthreads();   // everybody is done with the previous stage's S
#pragma unroll
      for(int e4 = 0; e4 < LD_nb * LD_nb / kBlock; ++e4) {
        const int e = tid + e4 * kBlock;
        S[e >> 6][e & 63] = sv[e4];
      }
      double iv[4][4];
      if(q == P) {
#pragma unroll
        for(int I = 0; I < 4; ++I)
#pragma unroll
          for(int kk = 0; kk < 4; ++kk) iv[I][kk] = liv[I][kk];
      }
      __syncthreads();
      // next stage in flight
      {
        const int nP = (q < P) ? P : P + 1, nq = (q < P) ? q + 1 : 0;
        if(nP < j) {
          if(nq == 0) drain_if_newest(nP);
          issue(nP, nq);
        }
      }
      if(q < P) {
#pragma unroll
        for(int Jq = 0; Jq < 4; ++Jq) {
#pragma unroll
          for(int kk = 0; kk < 4; ++kk) {
            const double vb = Vs[64 * q + 16 * Jq + 4 * kk + g][cl];
#pragma unroll
            for(int I = 0; I < 4; ++I)
              t[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[16 * Jq + 4 * kk + g][16 * I + li], vb, t[I], 0, 0, 0);
          }
          asm volatile("" ::: "memory");   // bound the number of LDS operand reads in flight (register pressure)
        }
      } else {
        double4_t vp[4];
#pragma unroll
        for(int I = 0; I < 4; ++I) {
          double4_t u = t[I];
#pragma unroll
          for(int J = 0; J < 4; ++J) {
            if(J < I) {
#pragma unroll
              for(int kk = 0; kk < 4; ++kk)
                u = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[16 * J + 4 * kk + g][16 * I + li], vp[J][kk], u, 0, 0, 0);
            }
          }
          double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[I][kk], u[kk], v, 0, 0, 0);
          vp[I] = v;
        }
        // U = D^-1 V (in place) and the LDS copy of V for the later block rows / phase (b)
#pragma unroll
        for(int I = 0; I < 4; ++I)
#pragma unroll
          for(int r = 0; r < 4; ++r) {
            const int row = 64 * P + 16 * I + g + 4 * r;
            const double v = col_ok ? vp[I][r] : 0.0;
            Vs[row][cl] = v;
            if(col_ok) A[(int64_t)(K0 + row) * lda + col] = v * dall[row];
          }
      }
    }
  }
}

// Compact (contiguous, ld = 256) copies of the 256x256 diagonal blocks.  The 1-workgroup super-diagonal kernel makes
// ~30 dependent global round trips; on the row-major N x N matrix consecutive rows are 64 KB apart and every round trip
// pays TLB misses (~14 us measured), on a 512 KB contiguous block it is an L2 hit.  The trailing update writes the
// next block's compact copy from its epilogue; only the very first block is gathered by ldlt_pack_diag_kernel.
__global__ __launch_bounds__(kBlock) void ldlt_pack_diag_kernel(const double* __restrict__ A, int64_t lda, int K0, int kbs,
                                                                double* __restrict__ C)
{
  const int r = blockIdx.x, c = threadIdx.x;
  C[r * LD_NB + c] = (r < kbs && c < kbs && c >= r) ? A[(int64_t)(K0 + r) * lda + (K0 + c)] : 0.0;
}

// factored block (U strictly upper, D on the diagonal) back into the matrix
// (and the transpose of its strictly upper part into CT, zero elsewhere: the backward solve reads columns of U)
// grid = (64, number of blocks): all diagonal blocks in one launch after the factorisation (nothing reads the matrix's
// own copy of a factored diagonal block before that — the panel kernels work on the compact copies).
__global__ __launch_bounds__(kBlock) void ldlt_unpack_diag_kernel(const double* __restrict__ Call, double* __restrict__ A,
                                                                  int64_t lda, int N, double* __restrict__ CTall)
{
  __shared__ double tile[32][33];
  const int K0 = blockIdx.y * LD_NB;
  const int kbs = (N - K0 < LD_NB) ? (N - K0) : LD_NB;
  const double* C = Call + (int64_t)blockIdx.y * (LD_NB * LD_NB);
  double* CT = CTall + (int64_t)blockIdx.y * (LD_NB * LD_NB);
  // 8 x 8 tiles of 32 x 32: blockIdx.x = tile id, 256 threads = 32 x 8
  const int tr = blockIdx.x >> 3, tc = blockIdx.x & 7;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for(int k = 0; k < 4; ++k) {
    const int r = 32 * tr + ty + 8 * k, c = 32 * tc + tx;
    const bool in = (r < kbs && c < kbs && c >= r);
    const double v = in ? C[r * LD_NB + c] : 0.0;
    if(in) A[(int64_t)(K0 + r) * lda + (K0 + c)] = v;
    tile[ty + 8 * k][tx] = (c > r) ? v : 0.0;
  }
  __syncthreads();
#pragma unroll
  for(int k = 0; k < 4; ++k) {
    const int c = 32 * tc + ty + 8 * k, r = 32 * tr + tx;   // CT[c][r] = U[r][c]
    CT[c * LD_NB + r] = tile[tx][ty + 8 * k];
  }
}

// one workgroup: LDL^T of the kbs x kbs (<= 256) diagonal block of a super-panel, panel after panel (left-looking):
//   (a) the 64 columns of panel j go through block rows p < j (4 waves x 16 columns, MFMA)
//   (b) A_jj -= sum_{p<j} V_pj^T D_p^-1 V_pj   (both MFMA operands from the LDS copy of V)
//   (c) A_jj = U^T D U in LDS (diag_factor_lds), factor / compact copy / 16x16 inverses written out
template <bool MF>
__global__ __launch_bounds__(kBlock) void ldlt_superdiag_kernel(double* __restrict__ A, int64_t lda, int K0, int kbs,
                                                                double* __restrict__ V, int64_t ldv,
                                                                double* __restrict__ dinv, double* __restrict__ Dk_sp,
                                                                double* __restrict__ Li_sp, int* __restrict__ info,
                                                                long long* __restrict__ tstamp)
{
  __shared__ double S[LD_nb][LD_nb + 1];
  __shared__ double sdinv[LD_nb];
  __shared__ double dall[LD_NB];
  __shared__ double Vs[192][80];
#define SD_STAMP(slot)                                                         \
  do {                                                                         \
    if(tstamp && threadIdx.x == 0) tstamp[slot] = (long long)wall_clock64(); \
  } while(0)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
  const int np = (kbs + LD_nb - 1) / LD_nb;
  for(int j = 0; j < np; ++j) {
    const int k0 = K0 + 64 * j;
    const int kbj = (kbs - 64 * j < LD_nb) ? (kbs - 64 * j) : LD_nb;
    const int cl = 16 * w + li;              // column inside panel j
    const int64_t col = (int64_t)k0 + cl;
    const bool col_ok = cl < kbj;
    const int64_t colc = col_ok ? col : (int64_t)k0;   // clamped (loads stay in bounds)
    SD_STAMP(j * 4 + 0);
    // A_jj is not touched by phase (a): issue its loads now so their HBM latency hides behind (a)
    double sv[LD_nb * LD_nb / kBlock];
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      const int r = e >> 6, c = e & 63;
      const int rr = (r < kbj) ? r : (kbj - 1), cc = (c < kbj) ? c : (kbj - 1);
      sv[q] = A[(int64_t)(k0 + rr) * lda + (k0 + cc)];
    }
    // ---- (a)
    panel_block_rows_pipelined(j, A, lda, K0, colc, col_ok, cl, Dk_sp, Li_sp, dall, g, li, S, Vs, tid);
    // ---- (b) S = A_jj (upper, zero padded)
    __syncthreads();   // every wave is done with the operand blocks staged in S during (a); Vs complete
    SD_STAMP(j * 4 + 1);
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      const int r = e >> 6, c = e & 63;
      S[r][c] = (r < kbj && c < kbj && c >= r) ? sv[q] : 0.0;
    }
    if(tid < LD_nb) sdinv[tid] = 1.0;
    __syncthreads();
    if(j > 0) {
      double4_t acc[4];
#pragma unroll
      for(int wc = 0; wc < 4; ++wc) acc[wc] = double4_t{0.0, 0.0, 0.0, 0.0};
      const int nk = 64 * j;   // rows of V above the diagonal block
      for(int kb16 = 0; kb16 < nk; kb16 += 16) {
        double aop[4], bop[4][4];
#pragma unroll
        for(int u = 0; u < 4; ++u) {
          const int row = kb16 + 4 * u + g;
          aop[u] = Vs[row][cl];                    // A operand [i = li][k = g]: V[row][my column]
          const double di = dall[row];
#pragma unroll
          for(int wc = 0; wc < 4; ++wc) bop[u][wc] = Vs[row][16 * wc + li] * di;
        }
#pragma unroll
        for(int u = 0; u < 4; ++u)
#pragma unroll
          for(int wc = 0; wc < 4; ++wc) acc[wc] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[u], bop[u][wc], acc[wc], 0, 0, 0);
      }
#pragma unroll
      for(int wc = 0; wc < 4; ++wc)
#pragma unroll
        for(int reg = 0; reg < 4; ++reg) {
          const int r = 16 * w + g + 4 * reg, c = 16 * wc + li;
          if(c >= r) S[r][c] -= acc[wc][reg];
        }
      __syncthreads();
    }
    // ---- (c)
    SD_STAMP(j * 4 + 2);
    diag_factor_lds<false, MF>(S, sdinv, kbj, k0, info, Li_sp + j * (4 * LD_SB * LD_SB), tid);
    SD_STAMP(j * 4 + 3);
    diag_emit(S, sdinv, kbj, k0, A, lda, dinv, Dk_sp + j * (LD_nb * LD_nb), tid);
    if(tid < LD_nb) dall[64 * j + tid] = sdinv[tid];
    __syncthreads();   // dall / S hand-over; the global stores are drained lazily (see phase (a))
  }
  SD_STAMP(16);
#undef SD_STAMP
}

// the row panel right of the super-panel's diagonal block: one wave per 16 columns, all (up to) 4 block rows
__global__ __launch_bounds__(64) void ldlt_supertrsm_kernel(double* __restrict__ A, int64_t lda, int N, int K0, int kbs,
                                                            double* __restrict__ V, int64_t ldv,
                                                            const double* __restrict__ dinv,
                                                            const double* __restrict__ Cd,
                                                            const double* __restrict__ Dk_sp,
                                                            const double* __restrict__ Li_sp, int col_ofs)
{
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15;
  const int64_t col = (int64_t)K0 + kbs + col_ofs + (int64_t)blockIdx.x * 16 + li;
  const bool col_ok = col < N;
  const int64_t colc = col_ok ? col : (int64_t)(N - 1);
  const int np = (kbs + LD_nb - 1) / LD_nb;
  const double* dsp = dinv + K0;
  double4_t Vv[4][4];
#pragma unroll
  for(int a = 0; a < 4; ++a)
#pragma unroll
    for(int b = 0; b < 4; ++b) Vv[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  block_row_solve<0>(A, lda, K0, kbs, colc, col_ok, Cd, Dk_sp, Li_sp, Vv, g, li);
  block_row_store<0>(A, lda, V, ldv, K0, kbs, col, col_ok, dsp, Vv, g, li);
  if(np > 1) {
    block_row_solve<1>(A, lda, K0, kbs, colc, col_ok, Cd, Dk_sp, Li_sp, Vv, g, li);
    block_row_store<1>(A, lda, V, ldv, K0, kbs, col, col_ok, dsp, Vv, g, li);
  }
  if(np > 2) {
    block_row_solve<2>(A, lda, K0, kbs, colc, col_ok, Cd, Dk_sp, Li_sp, Vv, g, li);
    block_row_store<2>(A, lda, V, ldv, K0, kbs, col, col_ok, dsp, Vv, g, li);
  }
  if(np > 3) {
    block_row_solve<3>(A, lda, K0, kbs, colc, col_ok, Cd, Dk_sp, Li_sp, Vv, g, li);
    block_row_store<3>(A, lda, V, ldv, K0, kbs, col, col_ok, dsp, Vv, g, li);
  }
}

// The HEAD of the row panel (the next super-panel's 256 columns — on the serial chain) with FOUR waves per 16 columns:
// wave I owns the 16-row sub-block I of every block row.  The single-wave kernel above spends 45 us on 640 MFMAs of
// ~130 cycles each, 384 of them in the products T_P -= L_Pq V_q; split by rows these are 96 per wave, and what stays serial
// is the 16-row substitution inside a block row: sub-step J = wave J finishes V_P[J] = Linv16_J u, publishes it in LDS,
// every wave I > J subtracts L_PP[I][J] V_P[J].  V is exchanged through LDS (Vs), one barrier per sub-step.  Only full
// panels (kbs = 256) are launched with it.  grid = 16-column groups, 256 threads.
__global__ __launch_bounds__(kBlock) void ldlt_headtrsm_kernel(double* __restrict__ A, int64_t lda, int N, int K0,
                                                               double* __restrict__ V, int64_t ldv,
                                                               const double* __restrict__ dinv,
                                                               const double* __restrict__ Cd,
                                                               const double* __restrict__ Dk_sp,
                                                               const double* __restrict__ Li_sp, int col_ofs)
{
  __shared__ double Vs[LD_NB][LD_SB + 1];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int I = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's 16-row sub-block
  const int64_t col = (int64_t)K0 + LD_NB + col_ofs + (int64_t)blockIdx.x * 16 + li;
  const bool col_ok = col < N;
  const int64_t colc = col_ok ? col : (int64_t)(N - 1);
  // the wave's rows of the matrix, all four block rows (independent of everything computed here)
  double4_t t[4];
#pragma unroll
  for(int P = 0; P < 4; ++P)
#pragma unroll
    for(int r = 0; r < 4; ++r) {
      const double v = ld_batch(A + (int64_t)(K0 + 64 * P + 16 * I + g + 4 * r) * lda + colc);
      t[P][r] = col_ok ? v : 0.0;
    }
  auto load_L = [&](double (&Lop)[4][4], int P, int q) {   // L_Pq[16 I + li][16 Jq + 4 kk + g], Jq, kk = 0..3
#pragma unroll
    for(int Jq = 0; Jq < 4; ++Jq)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) Lop[Jq][kk] = -ld_batch(Cd + (64 * q + 16 * Jq + 4 * kk + g) * LD_NB + (64 * P + 16 * I + li));
  };
#pragma unroll
  for(int P = 0; P < 4; ++P) {
    const double* Dk = Dk_sp + P * (LD_nb * LD_nb);
    const double* Li = Li_sp + P * (4 * LD_SB * LD_SB);
    // operands of the in-block substitution: L_PP[I][J], J < I, and the 16x16 inverse of sub-block I
    double nl[3][4], iv[4];
#pragma unroll
    for(int J = 0; J < 3; ++J)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) nl[J][kk] = (J < I) ? -ld_batch(Dk + (16 * J + 4 * kk + g) * LD_nb + 16 * I + li) : 0.0;
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) iv[kk] = ld_batch(Li + I * 256 + li * 16 + 4 * kk + g);
    // ---- T_P -= L_Pq V_q, q < P (V_q complete in LDS: the barriers of the previous block row)
    double4_t u = t[P];
#pragma unroll
    for(int q = 0; q < P; ++q) {
      double Lop[4][4];
      load_L(Lop, P, q);
#pragma unroll
      for(int Jq = 0; Jq < 4; ++Jq)
#pragma unroll
        for(int kk = 0; kk < 4; ++kk)
          u = __builtin_amdgcn_mfma_f64_16x16x4f64(Lop[Jq][kk], Vs[64 * q + 16 * Jq + 4 * kk + g][li], u, 0, 0, 0);
    }
    // ---- the 16-row substitution across the four waves
#pragma unroll
    for(int J = 0; J < 4; ++J) {
      if(I == J) {   // wave-uniform
        double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[kk], u[kk], v, 0, 0, 0);
#pragma unroll
        for(int r = 0; r < 4; ++r) {
          const int row = 64 * P + 16 * I + g + 4 * r;
          Vs[row][li] = v[r];
          if(col_ok) {
            V[(int64_t)row * ldv + col] = v[r];
            A[(int64_t)(K0 + row) * lda + col] = v[r] * dinv[K0 + row];
          }
        }
      }
      __syncthreads();
      if(I > J && J < 3) {
#pragma unroll
        for(int kk = 0; kk < 4; ++kk)
          u = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[J][kk], Vs[64 * P + 16 * J + 4 * kk + g][li], u, 0, 0, 0);
      }
    }
  }
}

// W_J = U_JJ^-1 = (L_JJ^-1)^T of every 256 x 256 diagonal block, for the dataflow solve: the row-panel substitution
// above applied to an identity.  grid = (16, number of blocks), one wave per 16 columns of the identity.
// wB = 512: W holds inverted 512 x 512 blocks; block Jb goes to the diagonal quadrant Jb % 2 of block Jb / 2 (row stride 512),
// the upper-right quadrants are filled by ldlt_w512_mm_kernel, the lower-left ones stay zero.
__global__ __launch_bounds__(64) void ldlt_inv_diag_kernel(int N, const double* __restrict__ Cd_all,
                                                           const double* __restrict__ Dblk, const double* __restrict__ Li_all,
                                                           double* __restrict__ W, int wB)
{
  const int Jb = blockIdx.y, K0 = Jb * LD_NB;
  const int kbs = (N - K0 < LD_NB) ? (N - K0) : LD_NB;
  const int np = (kbs + LD_nb - 1) / LD_nb;
  const double* Cd = Cd_all + (int64_t)Jb * (LD_NB * LD_NB);
  const double* Dk_sp = Dblk + (int64_t)(K0 / LD_nb) * (LD_nb * LD_nb);
  const double* Li_sp = Li_all + (int64_t)(K0 / LD_nb) * (4 * LD_SB * LD_SB);
  const int ldw = (wB == 512) ? 512 : LD_NB;
  double* Wj = (wB == 512) ? W + (int64_t)(Jb >> 1) * (512 * 512) + (int64_t)(Jb & 1) * (LD_NB * 512 + LD_NB)
                           : W + (int64_t)Jb * (LD_NB * LD_NB);
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15;
  const int c = blockIdx.x * 16 + li;
  double4_t Vv[4][4];
#pragma unroll
  for(int a = 0; a < 4; ++a)
#pragma unroll
    for(int b = 0; b < 4; ++b) Vv[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  block_row_solve<0, true>(nullptr, 0, K0, kbs, c, true, Cd, Dk_sp, Li_sp, Vv, g, li);
  if(np > 1) block_row_solve<1, true>(nullptr, 0, K0, kbs, c, true, Cd, Dk_sp, Li_sp, Vv, g, li);
  if(np > 2) block_row_solve<2, true>(nullptr, 0, K0, kbs, c, true, Cd, Dk_sp, Li_sp, Vv, g, li);
  if(np > 3) block_row_solve<3, true>(nullptr, 0, K0, kbs, c, true, Cd, Dk_sp, Li_sp, Vv, g, li);
  // W[c][row] = Linv[row][c]; outside the kbs x kbs leading part of a ragged last block: identity
#pragma unroll
  for(int P = 0; P < 4; ++P)
#pragma unroll
    for(int I = 0; I < 4; ++I)
#pragma unroll
      for(int r = 0; r < 4; ++r) {
        const int row = 64 * P + 16 * I + g + 4 * r;
        double v = (row < kbs && c < kbs) ? Vv[P][I][r] : ((row == c) ? 1.0 : 0.0);
        if(row < c) v = 0.0;
        Wj[c * ldw + row] = v;
      }
}

// C = alpha * X Y for 256 x 256 row-major blocks, one (X, Y, C) triple per blockIdx.y, one 32 x 32 tile of C per blockIdx.x:
// the two products that complete an inverted 512 x 512 diagonal block, inv([Ua Uab; 0 Ub]) = [Wa, -Wa Uab Wb; 0, Wb].
// Latency-bound work (16 pairs x 2 x 256^3 flops): 4 waves of one v_mfma_f64_16x16x4_f64 tile each, the operands of 32 k-steps
// in flight at once (two memory round trips per workgroup), 64 tiles x pairs workgroups so that every CU holds several.
__global__ __launch_bounds__(kBlock) void ldlt_w512_mm_kernel(const double* __restrict__ X, int64_t xpair, int64_t ldx,
                                                              const double* __restrict__ Y, int64_t ypair, int64_t ldy,
                                                              double* __restrict__ C, int64_t cpair, int64_t ldc, double alpha)
{
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lk = lane >> 4, li = lane & 15;
  const int wr = w >> 1, wc = w & 1;
  const int ti = blockIdx.x >> 3, tj = blockIdx.x & 7;
  const double* Xp = X + (int64_t)blockIdx.y * xpair + (int64_t)(32 * ti + 16 * wr + li) * ldx + lk;
  const double* Yp = Y + (int64_t)blockIdx.y * ypair + (int64_t)lk * ldy + 32 * tj + 16 * wc + li;
  double* Cp = C + (int64_t)blockIdx.y * cpair + (int64_t)(32 * ti + 16 * wr) * ldc + 32 * tj + 16 * wc;
  double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for(int half = 0; half < 2; ++half) {
    double av[32], bv[32];
#pragma unroll
    for(int kk = 0; kk < 32; ++kk) {
      const int k = 128 * half + 4 * kk;   // (+ lk, folded into the base pointers)
      av[kk] = Xp[k];
      bv[kk] = Yp[(int64_t)k * ldy];
    }
#pragma unroll
    for(int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
  }
#pragma unroll
  for(int reg = 0; reg < 4; ++reg) Cp[(int64_t)(lk + 4 * reg) * ldc + li] = alpha * acc[reg];
}

// ------------------------------------------------------------------------------------------
// Geometry of the 128 x 128 trailing-update tile of the dataflow factorisation (df_task_tile, ldlt_dataflow.hpp):
// 4 wave64 of 64 x 64 (4 x 4 tiles of v_mfma_f64_16x16x4_f64), K in double-buffered LDS stages of 16 k-rows, LDS row
// stride 128 + 16 doubles (the four k-rows of one ds_read_b64 fall on disjoint bank groups).
// ------------------------------------------------------------------------------------------
constexpr int UD_T = 128;             // tile edge
constexpr int UD_KT = 16;             // k-rows per stage
constexpr int UD_LD = UD_T + 16;      // LDS row stride (doubles)

// ------------------------------------------------------------------------------------------
// rank-K update on fp64 MFMA with a templated tile shape (the STEPWISE path's trailing update, 64 x 64 workgroup tiles,
// and the chain's diagonal-block update of that path, 32 x 64):
//     A[r][c] -= sum_{k<K} V[vrow0+k][r] * A[urow0+k][c]      r in [s, row_end), c in [max(r, s), col_end), upper triangle
// Small tiles = 68 VGPRs and 10 KB of LDS per workgroup: 5-7 waves per SIMD hide the un-pipelined LDS refill of its 8-deep
// stages (38.9 TFLOP/s over the 31 launches of an N = 8192 factorisation).  v_mfma_f64_16x16x4_f64 lane map:
// A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[row = (l>>4) + 4 reg][col = l&15].
// ------------------------------------------------------------------------------------------
// Wave tile = (16 WI) x (16 WJ), workgroup tile = (32 WI) x (32 WJ) (2 x 2 waves).
// KMASK (the pivoted factorisation's panels): only the first `kreal` of the K rows of U exist — the rows behind them are rows of A that
// the same launch updates and are read as zero instead (K is padded to the stage depth; a padded row must not depend on what the V
// buffer holds there: 0 * Inf from a not-yet-factored column would poison the whole trailing matrix).
template <int WI, int WJ, int KTx = LD_KT, bool KMASK = false>
__global__ __launch_bounds__(kBlock, 2) void ldlt_update_kernel_t(double* __restrict__ A, int64_t lda, int N,
                                                                  const double* __restrict__ V, int64_t ldv, int vrow0,
                                                                  int urow0, int K, int s, int row_end, int col_end,
                                                                  int skip_diag, double* __restrict__ Cnext, int kreal = 0)
{
  constexpr int TMx = 32 * WI, TNx = 32 * WJ;
  const int ti = blockIdx.y, tj = blockIdx.x;   // ti in units of TMx rows, tj in units of TNx columns
  const int r0 = s + ti * TMx, c0 = s + tj * TNx;
  if(c0 + TNx - 1 < r0) return;                 // entirely below the diagonal
  if(r0 >= row_end || c0 >= col_end) return;
  if(skip_diag && r0 < s + LD_NB && c0 < s + LD_NB) return;
  __shared__ double Vs[KTx][TMx + 16];
  __shared__ double Us[KTx][TNx + 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lk = lane >> 4, li = lane & 15;
  double4_t acc[WI][WJ];
#pragma unroll
  for(int i = 0; i < WI; ++i)
#pragma unroll
    for(int j = 0; j < WJ; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  // staging: V 16 x TMx, U 16 x TNx; a k-row is read by TMx (TNx) consecutive threads
  constexpr int VPR = kBlock / TMx, VL = KTx / VPR;   // k-rows per pass, loads per thread
  constexpr int UPR = kBlock / TNx, UL = KTx / UPR;
  const int vcol = tid % TMx, vrow = tid / TMx;
  const int ucl = tid % TNx, urw = tid / TNx;
  const bool vr_ok = (r0 + vcol) < N;
  const bool uc_ok = (c0 + ucl) < N;
  const double* Vp = V + (int64_t)(vrow0 + vrow) * ldv + (r0 + vcol);
  const double* Up = A + (int64_t)(urow0 + urw) * lda + (c0 + ucl);
  double vreg[VL], ureg[UL];
#pragma unroll
  for(int q = 0; q < VL; ++q) vreg[q] = vr_ok ? Vp[(int64_t)(VPR * q) * ldv] : 0.0;
#pragma unroll
  for(int q = 0; q < UL; ++q) ureg[q] = (uc_ok && (!KMASK || UPR * q + urw < kreal)) ? Up[(int64_t)(UPR * q) * lda] : 0.0;
  for(int kt = 0; kt < K; kt += KTx) {
    __syncthreads();
#pragma unroll
    for(int q = 0; q < VL; ++q) Vs[VPR * q + vrow][vcol] = vreg[q];
#pragma unroll
    for(int q = 0; q < UL; ++q) Us[UPR * q + urw][ucl] = ureg[q];
    __syncthreads();
    if(kt + KTx < K) {
#pragma unroll
      for(int q = 0; q < VL; ++q) vreg[q] = vr_ok ? Vp[(int64_t)(kt + KTx + VPR * q) * ldv] : 0.0;
#pragma unroll
      for(int q = 0; q < UL; ++q)
        ureg[q] = (uc_ok && (!KMASK || kt + KTx + UPR * q + urw < kreal)) ? Up[(int64_t)(kt + KTx + UPR * q) * lda] : 0.0;
    }
#pragma unroll
    for(int kk = 0; kk < KTx / 4; ++kk) {
      double a[WI], b[WJ];
#pragma unroll
      for(int i = 0; i < WI; ++i) a[i] = Vs[kk * 4 + lk][wr * 16 * WI + i * 16 + li];
#pragma unroll
      for(int j = 0; j < WJ; ++j) b[j] = Us[kk * 4 + lk][wc * 16 * WJ + j * 16 + li];
#pragma unroll
      for(int i = 0; i < WI; ++i)
#pragma unroll
        for(int j = 0; j < WJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for(int i = 0; i < WI; ++i) {
    double cv[4][WJ];
    bool ok[4][WJ];
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = r0 + wr * 16 * WI + i * 16 + lk + 4 * reg;
      const double* Crow = A + (int64_t)row * lda;
#pragma unroll
      for(int j = 0; j < WJ; ++j) {
        const int col = c0 + wc * 16 * WJ + j * 16 + li;
        ok[reg][j] = (row < row_end) && (col < col_end) && (col >= row);
        cv[reg][j] = ok[reg][j] ? Crow[col] : 0.0;
      }
    }
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = r0 + wr * 16 * WI + i * 16 + lk + 4 * reg;
      double* Crow = A + (int64_t)row * lda;
#pragma unroll
      for(int j = 0; j < WJ; ++j) {
        const int col = c0 + wc * 16 * WJ + j * 16 + li;
        if(ok[reg][j]) {
          const double nv = cv[reg][j] - acc[i][j][reg];
          Crow[col] = nv;
          // tiles of the next super-panel's diagonal block also feed its compact copy (origin s, ld = 256)
          if(Cnext && row < s + LD_NB && col < s + LD_NB) Cnext[(row - s) * LD_NB + (col - s)] = nv;
        }
      }
    }
  }
}

// The same kernel for the pivoted factorisation (ldlt_bk.hip): A[r][c] -= sum_{k<K} V[k][r] * A[urow0+k][c] for c >= r >= s, K a
// multiple of 8 (V = the panel W = L D, the A rows = the panel's columns of L in the column-major-lower reading of the storage)
// kreal <= K: the rows of U that exist (the panel's true width); rows kreal..K-1 are read as zero (see KMASK)
int ldlt_rankk_update(hiopamd_ctx* ctx, double* A, int64_t lda, int N, const double* V, int64_t ldv, int urow0, int K, int s, int kreal)
{
  if(K <= 0 || s >= N) return HIOPAMD_OK;
  if(K % 8 != 0 || kreal < 0 || kreal > K) return HIOPAMD_ERR_ARG;
  const int t = (N - s + 127) / 128;
  hipLaunchKernelGGL((ldlt_update_kernel_t<2, 2, 8, true>), dim3(2 * t, 2 * t), dim3(kBlock), 0, ctx->stream, A, lda, N, V, ldv, 0, urow0, K, s,
                     N, N, 0, (double*)nullptr, kreal);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

// ------------------------------------------------------------------------------------------
// inertia from D (thresholds of the reference's LAPACK path, hiopLinSolverSymDenseLapack.hpp:154-161:
// d < -1e-14 negative, |d| < 1e-14 null, else positive)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ldlt_inertia_kernel(int N, const double* __restrict__ A, int64_t lda,
                                                              int* __restrict__ out3)
{
  // one diagonal entry per thread, N / 256 workgroups, integer atomics onto the three (zeroed) counters: the entries are 8 (lda + 1)
  // bytes apart — one memory round trip each —, and a single workgroup walking all of them took 18 us at N = 8192
  int pos = 0, neg = 0, nul = 0;
  for(int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
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
    if(v) atomicAdd(out3 + threadIdx.x, v);
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


// ------------------------------------------------------------------------------------------
// 256-row solve steps.  A dependent kernel launch costs ~6.7 us on this machine whatever the kernel does
// (scripts/probes/hop_probe.hip), so the 64-row steps above (254 launches per solve) are launch-bound; these do the
// same work in 64 launches.  One launch = apply the solved block I to the rest of the right-hand side (every
// workgroup: 256 columns/rows) + LOOK-AHEAD solve of the next 256 x 256 diagonal block in workgroup 0, which reads the
// block from its compact copy (forward: Cd, row-major U; backward: CdT, its transpose), both contiguous 512 KB.
// Inside the block: four 64 x 64 unit-triangular chains (LDS-resident, waves 0/1 alternate) and the six 64 x 64
// off-diagonal products between them (register-resident, one per wave 2..7, loaded at kernel entry).
// ------------------------------------------------------------------------------------------
constexpr int SV_B = 256;
constexpr int SV_T = 512;   // threads per workgroup

// wave -> off-diagonal pair (p, q), p < q, of the 4 x 4 block grid: waves 2..7
__device__ __forceinline__ void sv_pair_of_wave(int w, int& p, int& q)
{
  p = (w <= 4) ? 0 : ((w <= 6) ? 1 : 2);
  q = (w == 2) ? 1 : (w == 3) ? 2 : (w == 4) ? 3 : (w == 5) ? 2 : 3;
}

// forward: workgroup g owns columns [j0 + 256 g, +256), j0 = i0 + 256 (block I = [i0, i0 + 256) is solved, always
// full); i0 < 0: first launch, only the diagonal solve of block 0.
__global__ __launch_bounds__(SV_T) void ldlt_fwd_step256(const double* __restrict__ A, int64_t lda, int N, int i0,
                                                         const double* __restrict__ Cd, double* __restrict__ b,
                                                         double* __restrict__ y)
{
  __shared__ double Sd[4][LD_nb][LD_nb + 1];
  __shared__ double ysh[SV_B];
  __shared__ double bs[SV_B];
  __shared__ double part[2][SV_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: row pointers below stay in SGPRs
  const int cgp = w & 3, rh = w >> 2;
  const bool spine = (blockIdx.x == 0);
  const int j0 = (i0 < 0) ? 0 : i0 + SV_B;
  const int jb = (N - j0 < SV_B) ? (N - j0) : SV_B;
  const int cl = 64 * cgp + lane;   // column inside the workgroup's 256
  const int64_t col = (int64_t)j0 + (int64_t)SV_B * blockIdx.x + cl;
  const bool col_ok = col < N;
  const int64_t colc = col_ok ? col : (int64_t)(N - 1);
  const double bold = (rh == 0 && col_ok) ? b[col] : 0.0;
  // Load schedule (everything below is latency-bound: keep as many independent loads in flight as registers allow):
  //   Cd diagonal blocks (spine, 32/thread) + rows 0..63 of this wave's half (64/lane)  ->  consume, then rows 64..127
  //   ->  stage the diagonal blocks in LDS, then this wave's off-diagonal block (64/lane, needed after the first chain)
  const double* Cj = Cd + (int64_t)(j0 / SV_B) * (SV_B * SV_B);
  double m[LD_nb];
  int pp = 0, pq = 0;
  double sv[32];
  if(spine) {
#pragma unroll
    for(int k = 0; k < 32; ++k) {
      const int e = tid + k * SV_T;
      const int q = e >> 12, s2 = (e >> 6) & 63, r = e & 63;
      sv[k] = Cj[(64 * q + s2) * SV_B + 64 * q + r];
    }
  }
  double acc = 0.0;
  if(i0 >= 0) {
    const double* Ar = A + (int64_t)(i0 + 128 * rh) * lda;   // uniform
    const int co = (int)colc;
    if(tid < SV_B) ysh[tid] = y[i0 + tid];
#pragma unroll 1
    for(int sb = 0; sb < 128; sb += 64) {   // 64 independent loads per lane in flight, twice
      double av[64];
#pragma unroll
      for(int q = 0; q < 64; ++q) av[q] = (Ar + (int64_t)(sb + q) * lda)[co];
      if(sb == 0) __syncthreads();   // ysh
#pragma unroll
      for(int q = 0; q < 64; ++q) acc = fma(av[q], ysh[128 * rh + sb + q], acc);
    }
    if(spine) {
#pragma unroll
      for(int k = 0; k < 32; ++k) {
        const int e = tid + k * SV_T;
        const int q = e >> 12, s2 = (e >> 6) & 63, r = e & 63;
        asm volatile("" : "+v"(sv[k]));
        Sd[q][s2][r] = (s2 < r && 64 * q + r < jb) ? sv[k] : 0.0;
      }
    }
  } else if(spine) {
#pragma unroll
    for(int k = 0; k < 32; ++k) {
      const int e = tid + k * SV_T;
      const int q = e >> 12, s2 = (e >> 6) & 63, r = e & 63;
      asm volatile("" : "+v"(sv[k]));
      Sd[q][s2][r] = (s2 < r && 64 * q + r < jb) ? sv[k] : 0.0;
    }
  }
  if(spine && w >= 2) {
    // this wave's off-diagonal block: its 64 loads must not be hoisted above the batches (they would not fit in the
    // register file next to them) -> make their address depend on the accumulated value
    int dep = 0;
    asm volatile("" : "+v"(dep) : "v"(acc));
    sv_pair_of_wave(w, pp, pq);
    const double* Cm = Cj + (64 * pp) * SV_B + 64 * pq + lane + dep;
#pragma unroll
    for(int s2 = 0; s2 < LD_nb; ++s2) m[s2] = Cm[s2 * SV_B];
  }
  part[rh][cl] = acc;
  __syncthreads();
  if(rh == 0) {
    const double nb = bold - part[0][cl] - part[1][cl];
    if(spine)
      bs[cl] = (cl < jb) ? nb : 0.0;
    else if(col_ok)
      b[col] = nb;
  }
  if(!spine) return;
  __syncthreads();
  // ---- diagonal block J: y_q = L_qq^-1 (b_q - sum_{p<q} L_qp y_p),  L = U^T
#pragma unroll
  for(int q = 0; q < 4; ++q) {
    if(w == (q & 1)) {
      double v = bs[64 * q + lane];
#pragma unroll
      for(int sb = 0; sb < LD_nb; sb += 16) {   // 16 LDS reads in flight at a time (the whole 64 would not fit next to m[])
        double u[16];
#pragma unroll
        for(int s2 = 0; s2 < 16; ++s2) u[s2] = Sd[q][sb + s2][lane];
#pragma unroll
        for(int s2 = 0; s2 < 16; ++s2) {
          const double ys = bcast_lane(v, sb + s2);
          v = fma(-u[s2], ys, v);
        }
      }
      ysh[64 * q + lane] = v;
    }
    __syncthreads();
    if(q < 3 && w >= 2 && pp == q) {
      double c = 0.0;
#pragma unroll
      for(int s2 = 0; s2 < LD_nb; ++s2) {   // the last block may be ragged: nothing outside jb x jb is defined
        asm volatile("" : "+v"(m[s2]));
        const double mv = (64 * pp + s2 < jb) ? m[s2] : 0.0;
        c = fma(mv, ysh[64 * q + s2], c);
      }
      if(64 * pq + lane < jb) bs[64 * pq + lane] -= c;
    }
    __syncthreads();
  }
  if(tid < jb) y[j0 + tid] = ysh[tid];
}

// backward: block I = [i0, i0 + ib) is solved (x); rows above it: z[h] -= U[h][I] . x_I.  Workgroup g owns the 256
// rows [p0 - 256 g, +256), p0 = i0 - 256; workgroup 0 also solves block P = [p0, i0) from CdT.  first != 0: only the
// diagonal solve of the LAST (possibly ragged) block [i0, i0 + ib).
__global__ __launch_bounds__(SV_T) void ldlt_bwd_step256(const double* __restrict__ A, int64_t lda, int i0, int ib,
                                                         const double* __restrict__ CdT, double* __restrict__ z,
                                                         double* __restrict__ x, int first)
{
  __shared__ double Sd[4][LD_nb][LD_nb + 1];
  __shared__ double xsh[SV_B];
  __shared__ double zs[SV_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
  const bool spine = (blockIdx.x == 0);
  const int p0 = first ? i0 : i0 - SV_B;          // block solved by the spine
  const int pb = first ? ib : SV_B;
  const double* Ct = CdT + (int64_t)(p0 / SV_B) * (SV_B * SV_B);
  double m[LD_nb];
  int pp = 0, pq = 0;   // pair (pp, pq), pp < pq: contribution of x_pq to z_pp
  double sv[32];
  if(spine) {
#pragma unroll
    for(int k = 0; k < 32; ++k) {
      const int e = tid + k * SV_T;
      const int q = e >> 12, c = (e >> 6) & 63, r = e & 63;
      sv[k] = Ct[(64 * q + c) * SV_B + 64 * q + r];   // U[64q + r][64q + c]
    }
  }
  if(!first) {
    if(tid < SV_B) xsh[tid] = (tid < ib) ? x[i0 + tid] : 0.0;
    __syncthreads();
    // 16 lanes per row, 4 rows per wave pass, 8 passes: wave w owns rows 32 w .. 32 w + 31 of the workgroup's 256
    const int rr = lane >> 4, pt = lane & 15;
    const int64_t hbase = (int64_t)p0 - (int64_t)SV_B * blockIdx.x + 32 * w;
#pragma unroll 1
    for(int ps = 0; ps < 8; ++ps) {
      // (more passes in flight did not help: the round trips are TLB-miss bound — 4 rows per pass, 64 KB apart —
      //  and the extra registers spilled)
      const int64_t h = hbase + 4 * ps + rr;
      const int64_t hc = (h >= 0) ? h : 0;
      const double* Ah = A + hc * lda + i0 + pt;
      double av[16];
#pragma unroll
      for(int k = 0; k < 16; ++k) av[k] = Ah[(16 * k + pt < ib) ? 16 * k : 0];
      double acc = 0.0;
#pragma unroll
      for(int k = 0; k < 16; ++k) acc = fma(av[k], xsh[16 * k + pt], acc);   // xsh is 0 beyond ib
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      acc += __shfl_xor(acc, 8, 64);
      if(pt == 0 && h >= 0) {
        const double zn = z[h] - acc;
        if(spine)
          zs[32 * w + 4 * ps + rr] = zn;
        else
          z[h] = zn;
      }
    }
  } else {
    if(tid < SV_B) zs[tid] = (tid < ib) ? z[i0 + tid] : 0.0;
  }
  if(!spine) return;
  {
#pragma unroll
    for(int k = 0; k < 32; ++k) {
      const int e = tid + k * SV_T;
      const int q = e >> 12, c = (e >> 6) & 63, r = e & 63;
      asm volatile("" : "+v"(sv[k]));
      Sd[q][c][r] = (c > r && 64 * q + c < pb) ? sv[k] : 0.0;
    }
    if(w >= 2) {
      int dep = 0;
      asm volatile("" : "+v"(dep) : "v"(sv[31]));   // after the staging above (register pressure)
      sv_pair_of_wave(w, pp, pq);
      const double* Cm = Ct + (64 * pq) * SV_B + 64 * pp + lane + dep;
#pragma unroll
      for(int c = 0; c < LD_nb; ++c) m[c] = Cm[c * SV_B];   // U[64pp + lane][64pq + c]
    }
  }
  __syncthreads();
  // ---- diagonal block P: x_q = U_qq^-1 (z_q - sum_{p>q} U_qp x_p)
#pragma unroll
  for(int qi = 0; qi < 4; ++qi) {
    const int q = 3 - qi;
    if(w == (q & 1)) {
      double v = zs[64 * q + lane];
#pragma unroll
      for(int c = LD_nb - 1; c >= 0; --c) {
        const double xc = bcast_lane(v, c);
        v = fma(-Sd[q][c][lane], xc, v);   // 0 unless c > lane
      }
      xsh[64 * q + lane] = v;
    }
    __syncthreads();
    if(q > 0 && w >= 2 && pq == q) {
      double c2 = 0.0;
#pragma unroll
      for(int c = 0; c < LD_nb; ++c) c2 = fma(m[c], xsh[64 * q + c], c2);
      zs[64 * pp + lane] -= c2;
    }
    __syncthreads();
  }
  if(tid < pb) x[p0 + tid] = xsh[tid];
}


// ------------------------------------------------------------------------------------------
// Dataflow solve.  The stepwise solves above are bound by the serial chain "solve a 256 x 256 triangle, launch, apply":
// 64 dependent launches of ~23 us per right-hand side against ~0.1 ms of memory traffic.  Here ONE launch runs the
// whole solve as a task graph over the 256 x 256 blocks of U:
//   * the diagonal blocks are inverted once per factorisation (W_J = U_JJ^-1, ldlt_inv_diag_kernel), so a diagonal
//     solve is a block mat-vec like every other task;
//   * one workgroup = one task = a 64-wide chunk of one block product.  It takes a ticket (tasks are issued in a
//     topological order, so a task only ever waits for tasks that are already running), loads its 256 x 64 piece of
//     the block into registers, and only THEN waits for its input vector: when the flag arrives the operand is
//     resident and the critical path per block step is two flag hops + 64 fused multiply-adds;
//   * off-diagonal tasks write their product to a private slot, the diagonal task adds the slots in a fixed order:
//     no floating-point atomics, results are bitwise reproducible from run to run.
// Flags are monotone 64-bit counters compared against epoch * (expected count), so nothing is reset between solves.
// ------------------------------------------------------------------------------------------
// Block size B of the task graph = threads per workgroup: 256, or 512 when the order is a multiple of 512 — the length of a
// solve is (number of block steps) x (two hand-overs + two small tasks), so half the steps is close to half the time; a 512
// block's inverse is assembled from the two 256 inverses and one extra product (ldlt_w512_mm_kernel).  B / 64 chunks per block.
constexpr int FL_LEAD = 3;   // block steps of lead of the operand loads over the chain (default; see the pacing note in the kernel)
enum { FL_FWD_OFF = 0, FL_FWD_DIAG = 1, FL_BWD_OFF = 2, FL_BWD_DIAG = 3 };

// Everything one task hands to another (y, x, the product slots) moves with agent-scope relaxed atomic loads and stores:
// they are coherent across the XCDs' L2s by themselves, so no cache write-back / invalidate sits on the critical path
// (acquire/release fences cost 1.9 us per hop against 1.1 us this way, scripts/probes/flag_hop_probe.hip — and a
// release fence per task throttles the whole kernel: every one walks the L2).
__device__ __forceinline__ double flow_ld(const double* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flow_st(double* p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The one input a task on the critical path waits for (y_I for the sub-diagonal product, that product for the next
// diagonal task) is not announced by a flag at all: the consumer polls the data words themselves, which hold a poison
// pattern until the producer's store lands — one memory round trip per hop instead of three (store-ack, flag add, flag
// poll, data load).  The exchange buffers exist twice; a launch uses the copy of its epoch's parity and every producer
// re-poisons the words it will write in the NEXT launch.
constexpr unsigned long long FL_POISON = 0x7FF8A5A5DEADBEEFull;   // a quiet NaN no arithmetic produces
// Every spin of the dataflow solve is bounded by wall-clock time (s_memrealtime, 100 MHz): FL_TIMEOUT_TICKS after the
// workgroup started, a wait gives up and raises the error word (the last word of the sync array).  The results of that
// launch are then meaningless; the host sees the word at its next synchronising call, re-initialises the exchange state
// and falls back to the stepwise solve (hiopamd_linsolver_solve_status / matrixChanged).
constexpr long long FL_TIMEOUT_TICKS = 200000000ll;   // 2 s
struct FlowGuard {
  unsigned long long* err;
  long long deadline;
  __device__ __forceinline__ bool expired(unsigned& n) const
  {
    if((++n & 1023u) != 0) return false;
    if((long long)wall_clock64() < deadline && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) return false;
    __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
};
__device__ __forceinline__ double flow_poll(const double* p, const FlowGuard& g)
{
  unsigned long long u;
  unsigned n = 0;
  while((u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == FL_POISON) {
    __builtin_amdgcn_s_sleep(1);
    if(g.expired(n)) return 0.0;
  }
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void flow_poison(double* p)
{
  __hip_atomic_store((unsigned long long*)p, FL_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// v - sum of `cnt` product slots p[0], p[stride], ... (fixed order; eight loads in flight at a time)
__device__ __forceinline__ double flow_sub_slots(double v, const double* p, int64_t stride, int cnt)
{
  int i = 0;
  for(; i + 8 <= cnt; i += 8) {
    double t[8];
#pragma unroll
    for(int k = 0; k < 8; ++k) t[k] = flow_ld(p + (int64_t)(i + k) * stride);
#pragma unroll
    for(int k = 0; k < 8; ++k) v -= t[k];
  }
  for(; i < cnt; ++i) v -= flow_ld(p + (int64_t)i * stride);
  return v;
}

__device__ __forceinline__ void flow_wait(const unsigned long long* p, unsigned long long target, const FlowGuard& g)
{
  if(threadIdx.x == 0) {
    unsigned n = 0;
    while(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if(g.expired(n)) break;
    }
  }
  __syncthreads();
}

// Wait for flag idx of a chain of flags that complete in order (step = -1: idx-1 completes before idx; +1: idx+1 does).
// Hundreds of resident tasks wait at any time; if they all polled their own flag at full rate the few flags about to
// flip would sit behind a queue of atomic loads.  So a task first watches the flags two and one steps ahead of its own
// with long sleeps and only then polls its own tightly: the number of tight pollers stays at the handful of tasks next
// in line.
__device__ __forceinline__ void flow_wait_chain(const unsigned long long* f, int idx, int step, int nb,
                                                unsigned long long target, int dist, const FlowGuard& g)
{
  if(threadIdx.x == 0) {
    const int i2 = idx + 2 * step, i1 = idx + step;
    unsigned n = 0;
    if(i2 >= 0 && i2 < nb)
      while(__hip_atomic_load(f + i2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(96);
        if(g.expired(n)) break;
      }
    if(i1 >= 0 && i1 < nb)
      while(__hip_atomic_load(f + i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(24);
        if(g.expired(n)) break;
      }
    if(dist == 0) {
      // next in line: the caller polls the data itself (flow_poll)
    } else {
      // not next in line (the product is consumed `dist` block steps later): poll slowly, and stagger the read of the
      // 2 KB input vector — every waiting task of this column reads the same lines, i.e. the same memory channel, and
      // the one task the chain is waiting for must not queue behind a hundred others
      while(__hip_atomic_load(f + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(16);
        if(g.expired(n)) break;
      }
      for(int d = 0; d < dist && d < 24; ++d) __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
}

// called by wave 0 after it stored the task's outputs (flow_st): the stores are acknowledged before the flag moves
__device__ __forceinline__ void flow_signal(unsigned long long* p)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if(threadIdx.x == 0) (void)__hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sync layout (unsigned long long): [0] ticket | fy | fa | fb | bx | ba | bb | fc | bc   (nb entries each)
// fa[J]: products (I, J) with I <= J - 3 delivered, fc[J]: I = J - 2, fb[J]: I = J - 1 (the diagonal task adds them in
// that order: the bulk is summed two block steps before the last one arrives); ba / bc / bb likewise for the rows.
// P: nb x nb slots of 256 doubles; slot (a, b) = (a * nb + b) * 256: forward uses (J, I), backward (I, J), I < J.
template <int B>
__global__ __launch_bounds__(B) void ldlt_solve_flow_kernel(const double* __restrict__ A, int64_t lda, int N, int nb,
                                                                 const double* __restrict__ W,
                                                                 const double* __restrict__ dinv,
                                                                 const int4* __restrict__ tasks, int ntasks,
                                                                 unsigned long long* sync, unsigned long long epoch,
                                                                 double* P, double* y, double* xc, double* Po, double* yo, double* xo,
                                                                 double* b, long long* ts, int lead)
{
  constexpr int FL_R = B / 64;    // 64-wide chunks per block
  constexpr int RP = B / 16;      // backward: rows per pass (16 lanes per row)
  constexpr int NPASS = 64 / RP;  // passes over the 64 rows of a chunk
  constexpr int KC = B / 16;      // 16-column strips per lane
  __shared__ int s_ticket;
  __shared__ double vsh[B];
  __shared__ double red[FL_R][64];
  const int tid = threadIdx.x;
  if(tid == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(sync, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ticket = (int)(t - (epoch - 1ull) * (unsigned long long)ntasks);
  }
  __syncthreads();
  const int4 tk = tasks[s_ticket];
  if(ts && tid == 0) ts[4 * s_ticket + 0] = wall_clock64();
  const int kind = tk.x, I = tk.y, J = tk.z, ch = tk.w;
  unsigned long long* fy = sync + 1;
  unsigned long long* fa = fy + nb;
  unsigned long long* fb = fa + nb;
  unsigned long long* bx = fb + nb;
  unsigned long long* ba = bx + nb;
  unsigned long long* bb = ba + nb;
  unsigned long long* fc = bb + nb;
  unsigned long long* bc = fc + nb;
  const unsigned long long eR = epoch * (unsigned long long)FL_R;
  const FlowGuard guard{bc + nb, (long long)wall_clock64() + FL_TIMEOUT_TICKS};   // error word = the word after the last flag
  // Pacing: tasks are issued in the order their results are needed, and the chain of diagonal solves sets the length of
  // the solve, so the operand blocks only have to arrive at a uniform rate.  Left alone, every resident task loads at
  // once and the chain's own small round trips queue behind 100 MB of streaming (block steps of 15 us instead of 4).
  // A task therefore starts loading when the chain has reached the same fraction of its way as the task's ticket,
  // minus a lead of FL_LEAD block steps.
  {
    const int half = ntasks >> 1;
    const bool fwd = s_ticket < half;
    const int rel = fwd ? s_ticket : s_ticket - half;
    const int pace = (int)(((long long)rel * nb) / half) - lead;   // block step of the chain to wait for
    if(pace >= 0) {
      const unsigned long long* f = fwd ? (fy + pace) : (bx + (nb - 1 - pace));
      if(tid == 0) {
        unsigned n = 0;
        while(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < eR) {
          __builtin_amdgcn_s_sleep(64);
          if(guard.expired(n)) break;
        }
      }
      __syncthreads();
    }
  }
  double m[64];
  if(kind <= FL_FWD_DIAG) {
    // ---- forward, out[c] = sum_r M[r][c] v[r]: lane <-> column, wave rg <-> rows 64 rg .. 64 rg + 63
    const int cl = tid & 63;
    const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t gcol = (int64_t)J * B + ch * 64 + cl;
    const bool colok = gcol < N;
    bool live = true;
    if(kind == FL_FWD_OFF) {
      const int64_t cc = colok ? gcol : (int64_t)(N - 1);
      const double* src = A + ((int64_t)I * B + rg * 64) * lda + cc;
#pragma unroll
      for(int q = 0; q < 64; ++q) m[q] = src[(int64_t)q * lda];
    } else {
      live = rg <= ch;   // W is upper triangular
      const double* src = W + (int64_t)J * (B * B) + (rg * 64) * B + ch * 64 + cl;
      if(live) {
#pragma unroll
        for(int q = 0; q < 64; ++q) m[q] = src[q * B];
      } else {
#pragma unroll
        for(int q = 0; q < 64; ++q) m[q] = 0.0;
      }
    }
    if(kind == FL_FWD_OFF) {
      flow_wait_chain(fy, I, -1, nb, eR, J - I - 1, guard);
      vsh[tid] = (J - I == 1) ? flow_poll(y + (int64_t)I * B + tid, guard) : flow_ld(y + (int64_t)I * B + tid);
    } else {
      const int64_t gi = (int64_t)J * B + tid;
      double v = (gi < N) ? b[gi] : 0.0;
      if(J >= 3) {
        flow_wait(fa + J, eR * (unsigned long long)(J - 2), guard);
        v = flow_sub_slots(v, P + (int64_t)J * nb * B + tid, B, J - 2);
      }
      if(J >= 2) {
        flow_wait(fc + J, eR, guard);
        v -= flow_ld(P + ((int64_t)J * nb + (J - 2)) * B + tid);
      }
      if(J >= 1) {
        const double* pl = P + ((int64_t)J * nb + (J - 1)) * B + tid;
        v -= flow_poll(pl, guard);
      }
      vsh[tid] = v;
    }
    if(ts && tid == 0) ts[4 * s_ticket + 1] = wall_clock64();
    __syncthreads();
    double acc = 0.0;
    if(live) {
#pragma unroll
      for(int q = 0; q < 64; ++q) acc = fma(m[q], vsh[rg * 64 + q], acc);
    }
    red[rg][cl] = acc;
    __syncthreads();
    if(tid < 64) {
      double out = red[0][tid];
#pragma unroll
      for(int q = 1; q < FL_R; ++q) out += red[q][tid];
      if(kind == FL_FWD_OFF) {
        flow_st(P + ((int64_t)J * nb + I) * B + ch * 64 + tid, colok ? out : 0.0);
        if(I == J - 1) flow_poison(Po + ((int64_t)J * nb + I) * B + ch * 64 + tid);
        flow_signal((I == J - 1) ? (fb + J) : (I == J - 2) ? (fc + J) : (fa + J));
        if(ts && tid == 0) ts[4 * s_ticket + 2] = wall_clock64();
      } else {
        if(colok) {
          flow_st(y + gcol, out);
          flow_poison(yo + gcol);
        }
        flow_signal(fy + J);
        if(ts && tid == 0) ts[4 * s_ticket + 2] = wall_clock64();
      }
    }
    return;
  }
  // ---- backward, out[r] = sum_c M[r][c] v[c]: 16 lanes per row, RP = B / 16 rows per pass, 64 / RP passes
  const int rr = tid >> 4, pt = tid & 15;
  if(kind == FL_BWD_OFF) {
#pragma unroll
    for(int ps = 0; ps < NPASS; ++ps) {
      const double* src = A + ((int64_t)I * B + ch * 64 + ps * RP + rr) * lda;
#pragma unroll
      for(int k = 0; k < KC; ++k) {
        const int64_t gc = (int64_t)J * B + 16 * k + pt;
        m[ps * KC + k] = src[(gc < N) ? gc : (int64_t)(N - 1)];
      }
    }
    flow_wait_chain(bx, J, +1, nb, eR, J - I - 1, guard);
    const int64_t gj = (int64_t)J * B + tid;
    vsh[tid] = (gj < N) ? ((J - I == 1) ? flow_poll(xc + gj, guard) : flow_ld(xc + gj)) : 0.0;
  } else {
    const double* Wi = W + (int64_t)I * (B * B);
#pragma unroll
    for(int ps = 0; ps < NPASS; ++ps) {
      const double* src = Wi + (ch * 64 + ps * RP + rr) * B + pt;
#pragma unroll
      for(int k = 0; k < KC; ++k) {
        // columns 16k..16k+15 lie left of every row of this pass: structurally zero
        m[ps * KC + k] = (16 * k + 15 < ch * 64 + ps * RP) ? 0.0 : src[16 * k];
      }
    }
    const int64_t gi = (int64_t)I * B + tid;
    flow_wait(fy + I, eR, guard);
    double v = (gi < N) ? flow_ld(y + gi) * dinv[gi] : 0.0;
    const int above = nb - 1 - I;
    if(above >= 3) {   // J = nb-1 .. I+3, descending
      flow_wait(ba + I, eR * (unsigned long long)(above - 2), guard);
      v = flow_sub_slots(v, P + ((int64_t)I * nb + (nb - 1)) * B + tid, -(int64_t)B, above - 2);
    }
    if(above >= 2) {
      flow_wait(bc + I, eR, guard);
      v -= flow_ld(P + ((int64_t)I * nb + (I + 2)) * B + tid);
    }
    if(above >= 1) {
      const double* pl = P + ((int64_t)I * nb + (I + 1)) * B + tid;
      v -= flow_poll(pl, guard);
    }
    vsh[tid] = v;
  }
  if(ts && tid == 0) ts[4 * s_ticket + 1] = wall_clock64();
  __syncthreads();
  double xv[KC];
#pragma unroll
  for(int k = 0; k < KC; ++k) xv[k] = vsh[16 * k + pt];
#pragma unroll
  for(int ps = 0; ps < NPASS; ++ps) {
    double acc = 0.0;
#pragma unroll
    for(int k = 0; k < KC; ++k) acc = fma(m[ps * KC + k], xv[k], acc);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    if(pt == 0) red[0][ps * RP + rr] = acc;
  }
  __syncthreads();
  if(tid < 64) {
    const double out = red[0][tid];
    if(kind == FL_BWD_OFF) {
      flow_st(P + ((int64_t)I * nb + J) * B + ch * 64 + tid, out);
      if(J == I + 1) flow_poison(Po + ((int64_t)I * nb + J) * B + ch * 64 + tid);
      flow_signal((J == I + 1) ? (bb + I) : (J == I + 2) ? (bc + I) : (ba + I));
      if(ts && tid == 0) ts[4 * s_ticket + 2] = wall_clock64();
    } else {
      const int64_t go = (int64_t)I * B + ch * 64 + tid;
      if(go < N) {
        flow_st(xc + go, out);   // for the tasks of this launch
        flow_poison(xo + go);
        b[go] = out;             // the caller's copy
      }
      flow_signal(bx + I);
      if(ts && tid == 0) ts[4 * s_ticket + 2] = wall_clock64();
    }
  }
}

#include "ldlt_dataflow.hpp"

// copy of the upper triangle by 128 x 128 tiles (J >= I), 16 bytes per lane where the order allows: the retry copy of the solver object
// (half the bytes of a full copy: ~0.13 ms at N = 8192)
__global__ __launch_bounds__(kBlock) void ldlt_triu_copy_kernel(int N, const double* __restrict__ src, double* __restrict__ dst, int64_t ld)
{
  const int I = blockIdx.y, J = blockIdx.x;
  if(J < I) return;
  const int r0 = 128 * I, c0 = 128 * J;
  const int tid = threadIdx.x;
  if((N & 1) == 0 && (ld & 1) == 0) {
    const int cp = 2 * (tid & 63), rq = tid >> 6;   // 64 lanes x 2 columns, 4 rows per pass
    if(c0 + cp < N)
#pragma unroll 4
      for(int r = rq; r < 128; r += 4) {
        if(r0 + r < N) {
          const int64_t e = (int64_t)(r0 + r) * ld + c0 + cp;
          *reinterpret_cast<double2*>(dst + e) = *reinterpret_cast<const double2*>(src + e);
        }
      }
  } else {
    const int c = tid & 127, rq = tid >> 7;
    if(c0 + c < N)
      for(int r = rq; r < 128; r += 2)
        if(r0 + r < N) dst[(int64_t)(r0 + r) * ld + c0 + c] = src[(int64_t)(r0 + r) * ld + c0 + c];
  }
}

// the same tiles between two pitches (8 bytes per lane: either pitch may be odd) — the solver object factors an order that the kernels'
// fast forms do not take as a larger one, diag(M, I), in a padded copy (hiopamd_linsolver::npad)
__global__ __launch_bounds__(kBlock) void ldlt_triu_repitch_kernel(int N, const double* __restrict__ src, int64_t lds, double* __restrict__ dst,
                                                                   int64_t ldd)
{
  const int I = blockIdx.y, J = blockIdx.x;
  if(J < I) return;
  const int r0 = 128 * I, c0 = 128 * J;
  const int c = threadIdx.x & 127, rq = threadIdx.x >> 7;
  if(c0 + c < N)
    for(int r = rq; r < 128; r += 2)
      if(r0 + r < N) dst[(int64_t)(r0 + r) * ldd + c0 + c] = src[(int64_t)(r0 + r) * lds + c0 + c];
}
// ... and the decoupled columns N .. NP - 1 of that copy: zeros, ones on the diagonal (one workgroup per row, upper triangle)
__global__ __launch_bounds__(kBlock) void ldlt_pad_tail_kernel(int N, int NP, double* __restrict__ dst, int64_t ldd)
{
  const int r = blockIdx.x;
  for(int c = N + threadIdx.x; c < NP; c += kBlock)
    if(c >= r) dst[(int64_t)r * ldd + c] = (c == r) ? 1.0 : 0.0;
}

// dst[0, cnt) = src[0, n) followed by zeros (a right-hand side into / out of the padded order)
__global__ __launch_bounds__(kBlock) void ldlt_vec_pad_kernel(int n, const double* __restrict__ src, double* __restrict__ dst, int cnt)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if(i < cnt) dst[i] = (i < n) ? src[i] : 0.0;
}

}  // namespace hiopamd

using namespace hiopamd;

// completion event of the last dataflow factorisation launched by this process on the current device (created on first use)
static hipEvent_t df_done_event()
{
  static std::mutex mu;
  static hipEvent_t ev[64] = {};
  int dev = 0;
  if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if(!ev[dev] && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) ev[dev] = nullptr;
  return ev[dev];
}
// One dataflow factorisation per DEVICE at a time, also across processes: the chain kernel's 16 roles wait for each other and
// need their 16 reserved CUs to themselves; two processes sharing a device could each get some of them and starve (round 2
// turned that into a 3 s time-out).  An advisory file lock per device (flock on /tmp/hiopamd_df_<pci bus id>.lock, taken
// without blocking for the duration of one factorisation — the call synchronises before it returns) decides who may use the
// dataflow kernels; whoever does not get it runs the stepwise kernels for this call.
struct DfDeviceLock {
  int fd = -1;
  bool held = false;
  bool try_acquire()
  {
    static std::mutex mu;
    static int fds[64];
    static bool init = false;
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;   // (cannot tell: behave as before)
    std::lock_guard<std::mutex> lk(mu);
    if(!init) {
      for(int& f : fds) f = -1;
      init = true;
    }
    if(fds[dev] < 0) {
      char bus[64] = "unknown";
      (void)hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev);
      for(char* c = bus; *c; ++c)
        if(*c == ':' || *c == '.' || *c == '/') *c = '_';
      char path[160];
      std::snprintf(path, sizeof(path), "/tmp/hiopamd_df_%s.lock", bus);
      fds[dev] = ::open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
      if(fds[dev] < 0) return true;   // no lock file possible: behave as before
    }
    fd = fds[dev];
    held = ::flock(fd, LOCK_EX | LOCK_NB) == 0;
    return held;
  }
  ~DfDeviceLock()
  {
    if(held && fd >= 0) (void)::flock(fd, LOCK_UN);
  }
};

// latency cuts of the spine step (df_spine_step): bit 0 = tile-solve publish deferred to the next spine step, bit 1 = next tiles
// prefetched into LDS while the last sub-block is factored, bit 2 = F published under the tile solve.  All on (each was measured on its
// own in round 2, profiles/r02_probes; the run-time switch is gone).
constexpr int DF_SPINE_OPT = 7;
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

// ---- host side of the dataflow factorisation: static task tables ------------------------------------------------
// Chain tasks of ONE super-panel, by owner (role): F(p) / T(p, c) / U(p; a, b) on the 64 x 64 tiles of the window
// (rows 0..3 = R_j, 4..7 = R_j+1; columns 0..3 = C_j, 4..7 = H_j / next diagonal block).  has_next = false: C_j only.
static void df_chain_tasks(bool has_next, std::vector<int4>& out)
{
  std::vector<std::vector<int4>> by_role(DF_ROLES);
  const int cmax = has_next ? 7 : 3;
  constexpr bool rsplit = true;   // the companion step split over roles 1 and 2 (the unsplit form was an A/B aid of round 2)
  auto nn_index = [](int a, int b) {   // upper-triangular tile (a, b) of the 4 x 4 next diagonal block -> 0..9
    int t = 0;
    for(int r = 0; r < a; ++r) t += 4 - r;
    return t + (b - a);
  };
  for(int p = 0; p < 4; ++p) {
    // the spine: S(p) = F(p) -> T(p, p+1) -> U(p; p+1, p+1) fused in one task, data carried in LDS (z = 1: with the T / U part);
    // its companion: R(p) = T(p, p+2) -> U(p; p+1, p+2) -> U(p; p+2, p+2)
    // (U(p; p+2, p+2) is a task of role 2 instead — R(p) z = 1 — so that the two updates
    //  the next spine step waits for run side by side; as one task they were 20 us in a row against the spine's 20 us period)
    by_role[0].push_back(make_int4(DF_S, p, (p + 1 <= cmax) ? 1 : 0, 0));
    if(p + 2 <= cmax) {
      by_role[1].push_back(make_int4(DF_R, p, rsplit ? 1 : 0, 0));
      if(rsplit) by_role[2].push_back(make_int4(DF_U, p, p + 2, p + 2));
    }
    for(int c = p + 1; c <= cmax; ++c) {   // tile solves of pivot p
      int role;
      if(c == p + 1 || c == p + 2) continue;   // inside S(p) / R(p)
      else if(c <= 3) role = 2;
      else role = 3 + (c - 4);
      // C(p, c) = T(p, c) + the updates U(p; a, c), a = p+1 .. min(3, c), of the same window column, one task.
      // The FIRST H column (c = 4) is on the critical path (T(2,4) of the companion and T(3,4) of the spine need it pivot
      // after pivot): its role keeps only the tile solve and the update of the next tile (p+1, 4); the other updates of
      // that column go to role 2, which has little else to do.
      if(c == 4) {
        by_role[role].push_back(make_int4(DF_C, p, c, p + 1));
        // with the companion split role 2 carries the spine-critical diagonal updates: the three left-over updates of the
        // first H column go to roles 13-15 (one non-critical next-diagonal tile each), ahead of their own tasks of this pivot
        for(int a2 = p + 2; a2 <= 3; ++a2) {
          const int lr = rsplit ? 13 + ((p == 0) ? (a2 - 2) : 2) : 2;   // U(0;2,4) -> 13, U(0;3,4) -> 14, U(1;3,4) -> 15
          by_role[lr].push_back(make_int4(DF_U, p, a2, c));
        }
      } else {
        by_role[role].push_back(make_int4(DF_C, p, c, 0));
      }
    }
    for(int a = p + 1; a <= cmax; ++a)
      for(int b = a; b <= cmax; ++b) {   // updates by pivot p
        int role;
        if(a == p + 1 && b == p + 1) continue;                                               // inside S(p)
        else if((a == p + 1 && b == p + 2) || (a == p + 2 && b == p + 2)) continue;          // inside R(p)
        else if(a <= 3 && b != p + 1 && b != p + 2) continue;                                // inside C(p, b)
        else if(b <= 3) role = 2;
        else if(a <= 3) role = 3 + (b - 4);
        else role = 7 + nn_index(a - 4, b - 4) % 9;
        by_role[role].push_back(make_int4(DF_U, p, a, b));
      }
  }
  out.assign((size_t)DF_ROLES * DF_MAXT, make_int4(DF_END, 0, 0, 0));
  for(int r = 0; r < DF_ROLES; ++r) {
    if((int)by_role[r].size() >= DF_MAXT) std::abort();
    for(size_t k = 0; k < by_role[r].size(); ++k) out[(size_t)r * DF_MAXT + k] = by_role[r][k];
  }
}

struct DfPlan {
  int N = 0, nsp = 0, nt = 0, nchain = 0, last_has_next = 0, nwide = 0;
  bool has_far = false;    // a FAR update list exists (never in the shipped plan: one update list per super-panel)
  int64_t off_chain = 0, off_tr = 0, off_ver = 0, off_trb = 0, off_cu = 0, off_wg = 0, off_snap = 0, off_where = 0, off_shadow = 0, off_run = 0, nflags = 0;
  std::vector<int4> ctasks, wtasks;
  std::vector<unsigned> upcnt, wfirst;
  std::vector<int4> wq, wf;
  double up_flops = 0.0;   // algorithmic flops of the UP tasks (2 K per updated element)
};
static DfPlan df_build_plan(int N)
{
  DfPlan P;
  P.N = N;
  P.nsp = (N + LD_NB - 1) / LD_NB;
  P.nt = (N + UD_T - 1) / UD_T;
  const int nfull = N / LD_NB;
  if(N % LD_NB == 0) {
    P.nchain = nfull;
    P.last_has_next = 0;
  } else {
    P.nchain = nfull - 1;   // the last full super-panel has a ragged successor: left to the stepwise kernels
    P.last_has_next = 1;
  }
  if(P.nchain < 0) P.nchain = 0;
  P.nwide = P.last_has_next ? P.nchain : P.nchain - 1;   // super-panels with a row-panel tail + trailing update
  if(P.nwide < 0) P.nwide = 0;
  P.off_chain = DF_HDR;
  P.off_tr = P.off_chain + (int64_t)(P.nsp + 1) * DF_CH;
  P.off_ver = P.off_tr + (int64_t)P.nsp * P.nt;
  // (+ the profiling stamps and phase sums, + per super-panel 2 x 4 block-row counters of the substitution tasks that feed the
  //  first four update tiles, see DF_UPH)
  P.off_trb = P.off_ver + (int64_t)P.nt * P.nt + 8 * (int64_t)(P.nsp + 1) + 48;
  P.off_cu = P.off_trb + 8 * (int64_t)(P.nsp + 1);
  P.off_wg = P.off_cu + 512;
  P.off_snap = P.off_wg + 2 * 512;     // (at most 480 + 16 workgroups)
  P.off_where = P.off_snap + 1024;
  P.off_shadow = P.off_where + 2 * 512;
  P.off_run = P.off_shadow + (int64_t)P.nsp * P.nt + 2 * (int64_t)P.nt * P.nt + (int64_t)P.nsp * 256 + (int64_t)P.nsp * P.nt * P.nt;   // tr, ver, (pad), executions of TR / update tasks
  P.nflags = P.off_run + 0;         // (+ wtasks.size(), added when the lists exist)
  std::vector<int4> t0, t1;
  df_chain_tasks(true, t0);
  df_chain_tasks(false, t1);
  P.ctasks = t0;
  P.ctasks.insert(P.ctasks.end(), t1.begin(), t1.end());
  P.upcnt.assign((size_t)P.nsp + 1, 0u);
  // wide-kernel queues (csrc/ldlt_wide_body.inc): TR tasks grouped by super-panel; NEAR update tasks grouped by super-panel
  // (head tiles, the two tile rows of the next row panel — counted in wfirst —, for an unpaired or paired-first super-panel also
  // the two tile rows of the row panel after it); FAR update tasks grouped by super-panel (everything else, row-major).
  std::vector<int4> trq, nearq, farq;
  std::vector<int4> wq((size_t)P.nwide + 1, make_int4(0, 0, 0, 0)), wf((size_t)P.nwide + 1, make_int4(0, 0, -1, 0));
  P.wfirst.assign((size_t)P.nwide + 1, 0u);
  constexpr bool uph = true;     // head tiles gated per block row (DF_UPH)
  // One update list per super-panel, everything NEAR (round 2's order).  Separate NEAR / FAR lists with NEAR served first were built and
  // measured in round 3 (5.27-5.47 ms against 5.17-5.30: the near tiles do jump the queue, but the far tiles they depend on through `ver`
  // then run later and the chain waits for those instead); the kernels still understand FAR lists, the plan never makes one.
  constexpr bool split = false;
  auto emit_tile = [&](std::vector<int4>& q, int kind, int j, int I, int J) {
    q.push_back(make_int4(kind, j, I, J));
    P.upcnt[j] += 1u;
    const int np = kind == DF_UP2 ? 2 : 1;
    if(kind == DF_UP2) P.upcnt[j - 1] += 1u;   // it reads the row panel of super-panel j - 1 as well
    // rows r in tile I, columns max(r, 128 J) .. of tile J, inside the matrix
    const int r0 = UD_T * I, r1 = std::min(N, r0 + UD_T), c0 = UD_T * J, c1 = std::min(N, c0 + UD_T);
    for(int r = r0; r < r1; ++r) P.up_flops += np * 2.0 * LD_NB * (double)std::max(0, c1 - std::max(c0, r));
  };
  auto is_head = [&](int j, int I, int J) { return uph && I < 2 * j + 4 && J < 2 * j + 6; };
  auto emit_up = [&](std::vector<int4>& q, int j, int I) {
    for(int J = (I < 2 * j + 4) ? 2 * j + 4 : I; J < P.nt; ++J)
      if(!is_head(j, I, J)) emit_tile(q, DF_UP, j, I, J);
  };
  // K = 512 (DF_UP2): super-panels e (even) and e + 1 are applied together to the tile rows behind super-panel e + 2
  // (I >= 2 e + 6) by FAR tasks of queue e + 1; queue e keeps the rows of super-panels e + 1 and e + 2 (what the next two chains
  // and substitution rounds wait for).  k512 = number of leading super-panels that may be paired: the update-bound first half
  // (measured at N = 8192: 5.32-5.36 ms unpaired, 5.19-5.27 with 16 of 31, 5.38 with 24 — in the chain-bound second half a fused
  // task only adds latency).
  const int k512 = (P.nwide + 1) / 2;
  auto paired_first = [&](int j) { return (j % 2 == 0) && j + 1 < P.nwide && j + 1 < k512; };
  std::vector<int> near_first((size_t)P.nwide + 1, 0), near_cnt((size_t)P.nwide + 1, 0), far_first((size_t)P.nwide + 1, 0),
      far_cnt((size_t)P.nwide + 1, 0), near_maxrow((size_t)P.nwide + 1, -1);
  for(int j = 0; j < P.nwide; ++j) {
    wq[j].x = (int)trq.size();
    for(int c = LD_NB * (j + 2); c < N; c += DF_TRW) trq.push_back(make_int4(DF_TR, j, c, DF_TRW));
    wq[j].y = (int)trq.size() - wq[j].x;
    near_first[j] = (int)nearq.size();
    far_first[j] = (int)farq.size();
    // the four tiles of H_j+1 = A[R_j+1, first 256 columns behind it] first: the NEXT super-panel's chain waits for them
    // (its first tile solves of the H column), so they follow this super-panel block row by block row (DF_UPH)
    for(int I = 2 * j + 2; I < 2 * j + 4 && I < P.nt; ++I)
      for(int J = 2 * j + 4; J < 2 * j + 6 && J < P.nt; ++J)
        if(is_head(j, I, J)) emit_tile(nearq, DF_UPH, j, I, J);
    emit_up(nearq, j, 2 * j + 2);
    if(2 * j + 3 < P.nt) emit_up(nearq, j, 2 * j + 3);
    P.wfirst[j] = (unsigned)((int)nearq.size() - near_first[j]);
    const bool first = paired_first(j), second = j >= 1 && paired_first(j - 1);
    std::vector<int4>& rest = split ? farq : nearq;
    if(second) {
      for(int I = 2 * j + 4; I < P.nt; ++I)
        for(int J = I; J < P.nt; ++J) emit_tile(rest, DF_UP2, j, I, J);
    } else {
      for(int I = 2 * j + 4; I < 2 * j + 6 && I < P.nt; ++I) emit_up(nearq, j, I);
      if(!first)
        for(int I = 2 * j + 6; I < P.nt; ++I) emit_up(rest, j, I);   // (first: the far rows follow in queue j + 1, fused)
    }
    near_cnt[j] = (int)nearq.size() - near_first[j];
    far_cnt[j] = (int)farq.size() - far_first[j];
    for(int t = near_first[j]; t < (int)nearq.size(); ++t) near_maxrow[j] = std::max(near_maxrow[j], nearq[t].z);
  }
  for(int j = 0; j < P.nwide; ++j) {
    wq[j].z = (int)trq.size() + near_first[j];
    wq[j].w = near_cnt[j];
    wf[j].x = (int)trq.size() + (int)nearq.size() + far_first[j];
    wf[j].y = far_cnt[j];
    // the FAR list that feeds this NEAR list's tiles: the latest non-empty one of an earlier super-panel (FAR lists are walked in
    // order, so its tasks being taken implies every earlier one is); needed: its tasks on tile rows up to this list's last row
    int dq = -1;
    for(int q = j - 1; q >= 0; --q)
      if(far_cnt[q] > 0) {
        dq = q;
        break;
      }
    wf[j].z = dq;
    wf[j].w = 0;
    if(dq >= 0)
      for(int t = far_first[dq]; t < far_first[dq] + far_cnt[dq]; ++t)
        if(farq[t].z <= near_maxrow[j]) wf[j].w = t - far_first[dq] + 1;
  }
  P.wtasks = trq;
  P.wtasks.insert(P.wtasks.end(), nearq.begin(), nearq.end());
  P.wtasks.insert(P.wtasks.end(), farq.begin(), farq.end());
  P.wf = wf;
  for(const int4& f : wf) P.has_far = P.has_far || f.y > 0;
  P.wq = wq;
  P.nflags = P.off_run + 2 * (int64_t)P.wtasks.size();   // handed out, completed
  return P;
}

// row-panel workspaces for order n: at least DF_NVB_MIN = 4 (three + one for the fused update tasks, which hold a buffer one super-panel
// longer)
static int df_nvb_for(int n)
{
  const int nsp = (n + LD_NB - 1) / LD_NB;
  // One workspace per super-panel while that costs at most 2 GB (N <= 16384): no buffer is ever reused, so nothing — neither the chain
  // kernel nor the idle workgroups of the wide kernel — polls the "updates of super-panel j - nvb complete" counters, which ~1500 tasks
  // increment (polling a word that is being incremented from everywhere is what delayed flag updates).
  if((double)nsp * LD_NB * (double)n * 8.0 <= 2.0e9) return std::max(nsp, DF_NVB_MIN);
  return DF_NVB_MIN;
}

struct DfDevice {
  DfPlan plan;
  unsigned* flags = nullptr;
  int4* ctasks = nullptr;
  int4* wtasks = nullptr;
  unsigned* upcnt = nullptr;
  int4* wq = nullptr;
  int4* wf = nullptr;
  unsigned* wfirst = nullptr;
  int nvb = DF_NVB_MIN;
  bool enabled = true;
  // a bounded wait expired (hiopamd_linsolver_matrix_changed): the NEXT factorisation of this object — the caller's retry on the
  // re-assembled matrix — runs the stepwise kernels, the one after that the dataflow pair again; three time-outs in a row
  // (no successful dataflow factorisation in between) switch the object to the stepwise kernels for good
  bool skip_once = false;
  int strikes = 0;
};

struct hiopamd_linsolver {
  hiopamd_ctx* ctx = nullptr;
  int n = 0;
  double* M = nullptr;      // n x n row-major
  double* dinv = nullptr;   // n
  double* V = nullptr;      // nvb x LD_NB x n workspace
  int nvb = DF_NVB_MIN;
  double* ybuf = nullptr;   // n
  double* Dblk = nullptr;   // ceil(n/64) staged 64x64 diagonal blocks
  double* Cd = nullptr;     // ceil(n/256) compact 256x256 diagonal blocks (ld = 256)
  int* d_info = nullptr;    // [0]=zero-pivot flag, [1..3]=pos,neg,zero
  // dataflow solve (ldlt_solve_flow_kernel)
  double* W = nullptr;                  // ceil(n/256) inverted diagonal blocks (fl_B = 256), or n/512 inverted 512 x 512 blocks (fl_B = 512)
  double* Wt = nullptr;                 // fl_B = 512: scratch for U_ab W_b, one 256 x 256 block per 512-block
  int fl_B = 256;                       // block size of the dataflow solve's task graph
  int fl_nb = 0;                        // its number of blocks
  int fl_lead = FL_LEAD;                // lead of its operand loads (block steps)
  double* P = nullptr;                  // nb x nb product slots of 256
  // safe mode (static quasi-definite regularisation + iterative refinement against a saved copy of the matrix)
  bool safe_mode = false;
  int safe_npos = 0;
  double safe_delta_rel = 1.4901161193847656e-08;   // sqrt(eps)
  double* Msave = nullptr;   // n x n: the matrix as assembled (symmetrised), safe mode only
  double* rbuf = nullptr;    // 3 n: right-hand side copy, residual, probe vector of matrixChanged
  bool safe_solve_failed = false;
  int safe_last_refinements = 0;
  double safe_last_residual = 0.0;
  double safe_anorm = 0.0;
  unsigned long long* fl_sync = nullptr;
  int4* fl_tasks = nullptr;
  int fl_ntasks = 0;
  unsigned long long fl_epoch = 0;
  bool flow_enabled = true;    // dataflow solve in use (false: stepwise 256-row solves)
  bool flow_dirty = false;     // dataflow solves were launched since the error word was last looked at
  bool flow_failed = false;    // sticky: a dataflow solve timed out since the last factorisation (its results were invalid)
  // Retry copy: the upper triangle as the caller assembled it, taken before a DATAFLOW factorisation overwrites the matrix, so that a
  // bounded wait that expires costs one stepwise factorisation instead of an error (hiopamd_linsolver_set_retry_copy; on by default:
  // the callers of the reference's matrixChanged() cannot re-assemble; the native KKT objects re-assemble themselves and switch it off)
  bool retry_copy = true;
  double* Mretry = nullptr;   // n x n, upper 128 x 128 tiles used; allocated on first use
  // Padded orders (round 6).  The kernels have fast forms for SOME orders: the 16-byte tile form of the dataflow kernels needs an even
  // order and pitch (N = 8191 took 8.9 ms against 5.2 ms for 8192 in the 8-byte form); an order that is not a multiple of 256 leaves
  // its last two super-panels to the stepwise kernels (8193 -> 8194: 5.9 ms); the solve runs on 512-row blocks only for multiples of
  // 512 (0.18 ms against 0.36 on 256-row blocks at N ~ 8192).  An order n >= LD_PAD_MIN is therefore factored and solved as
  // diag(M, I) of order npad >= n in Mpad, npad chosen by ldlt_padded_order(): the upper triangle is copied in and the factor copied
  // back into M (2 x 0.13 ms at 8192; what the caller reads through sys_matrix is the factor, as ever), a right-hand side goes through
  // xpad.  M itself stays as assembled until the factorisation has succeeded, so it is also the retry copy.  Every workspace and task
  // table of the object is built for npad; the pivoted mode has its own storage and works on M.
  int npad = 0;               // order the factorisation and the solves run at (>= n)
  double* Mpad = nullptr;     // npad x npad, allocated on first use
  double* xpad = nullptr;     // npad
  // hiopamd_linsolver_assembly_matrix: a caller that assembles at any pitch (the native KKT objects) writes the padded copy itself — the
  // next matrixChanged factors it where it is (no copy in, no copy back: 2 x 0.13 ms at 8192)
  bool direct_armed = false;   // the next matrixChanged finds its matrix in Mpad
  bool factor_in_pad = false;  // the last factorisation was of that kind: M does not show the factor (hiopamd_linsolver_sys_matrix_sync)
  long df_timeouts = 0;       // bounded waits that expired over the object's life (hiopamd_linsolver_timeouts)
  // pivoted mode (Bunch-Kaufman, ldlt_bk.hip): hiopamd_linsolver_set_pivoting
  bool pivoted = false;
  hiopamd_ldlt_bk* bk = nullptr;
  bool factored = false;
  int inertia[3] = {0, 0, 0};
  double flops_fact = 0.0, flops_triu = 0.0;   // hiopLinSolStats::flopsFact / flopsTriuSolves (cumulative)
  LdltProfile prof;
  DfDevice df;   // dataflow factorisation (task tables + flags)
};

static int ldlt_factor_impl(hiopamd_ctx* ctx, int N, double* A, int64_t lda, double* dinv, double* V, double* Dblk,
                            double* Cd, int* d_info, int* inertia3_host, LdltProfile* prof = nullptr, double* Winv = nullptr,
                            DfDevice* df = nullptr, int wB = 256, double* Wt = nullptr)
{
  // Dblk: per 64-row panel a compact 64x64 copy of the factored diagonal block, followed (after all
  // the blocks) by the per-panel 4 x 16x16 inverses
  double* Li = Dblk + (int64_t)((N + LD_nb - 1) / LD_nb) * (LD_nb * LD_nb);
  const bool timed = prof && prof->enabled;
  hipStream_t upd_stream = ctx->stream;
  // trailing update launch of the STEPWISE path (fallback / leftover super-panels): the 64 x 64-tile kernel, t = number of
  // 128-wide tile columns covering [s, N)
  auto launch_update = [&](int t, const double* Vb, int urow0, int K, int s, int row_end, int col_end, int skip_diag) {
    if(timed) (void)hipEventRecord(prof->get(), upd_stream);
    hipLaunchKernelGGL((ldlt_update_kernel_t<2, 2, 8>), dim3(2 * t, 2 * t), dim3(kBlock), 0, upd_stream, A, lda, N, Vb, (int64_t)N, 0,
                       urow0, K, s, row_end, col_end, skip_diag, nullptr);
    if(timed) {
      (void)hipEventRecord(prof->get(), upd_stream);
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
  const int64_t ldv = N;
  // LOOK-AHEAD on two CU-masked streams (ctx_cu_split): `sd` owns the reserved CUs (two per XCD by default) and runs the serial chain
  // of the factorisation without ever leaving its stream; `su` owns the other CUs (240 with the default of two reserved CUs per XCD) and runs the wide work.
  //   sd : superdiag(j) | trsm_head(j) | upd_diag(j) | superdiag(j+1) | trsm_head(j+1) | ...
  //   su :     wait diag(j) | trsm_tail(j) | wait head(j) | upd_rest(j) | wait diag(j+1) | trsm_tail(j+1) | ...
  // superdiag : 1 workgroup, the 256x256 diagonal block (needs a whole CU's LDS: it would starve behind the update grid
  //             on a shared CU — the dispatcher back-fills partially free CUs with update tiles);
  // trsm_head : the first 256 columns of the row panel (16 waves) — all upd_diag needs;
  // upd_diag  : the 3 tiles of the NEXT diagonal block, written to the matrix and to the block's compact copy;
  // trsm_tail / upd_rest : the rest of the row panel and of the trailing update (one launch, diagonal tiles skipped).
  // The chain crosses streams only to wait for upd_rest(j-1) (rows of panel j complete) before trsm_head(j); a
  // cross-stream dependency costs ~15 us on this machine, so none sits between the chain's own kernels.  V is
  // double-buffered (row panel j+1 is written while upd_rest(j) reads row panel j).
  const int nsp = (N + LD_NB - 1) / LD_NB;
  const int64_t cdt_ofs = (int64_t)nsp * (LD_NB * LD_NB);   // the transposed compact blocks follow the compact blocks
  const bool lookahead = nsp > 2 && ctx_cu_split(ctx);
  hipStream_t su = st, sd = st;
  if(lookahead) {
    su = ctx->upd_stream;
    sd = ctx->diag_stream;
  }
  int evn = 0;
  auto next_event = [&]() { return ctx_event(ctx, (evn++) % 160); };
  auto dep = [&](hipStream_t from, hipStream_t to) -> int {   // `to` continues after everything queued on `from`
    if(from == to) return HIOPAMD_OK;
    hipEvent_t e = next_event();
    HIOPAMD_CHECK(hipEventRecord(e, from));
    HIOPAMD_CHECK(hipStreamWaitEvent(to, e, 0));
    return HIOPAMD_OK;
  };
  struct Panel {
    int K0, Kend, kbs;
    double *Vb, *Dk_sp, *Li_sp, *Cj;
  };
  auto panel = [&](int jp) {
    Panel p;
    p.K0 = jp * LD_NB;
    p.Kend = (p.K0 + LD_NB < N) ? p.K0 + LD_NB : N;
    p.kbs = p.Kend - p.K0;
    p.Vb = V + (int64_t)(jp & 1) * LD_NB * ldv;
    p.Dk_sp = Dblk + (int64_t)(p.K0 / LD_nb) * (LD_nb * LD_nb);
    p.Li_sp = Li + (int64_t)(p.K0 / LD_nb) * (4 * LD_SB * LD_SB);
    p.Cj = Cd + (int64_t)jp * (LD_NB * LD_NB);
    return p;
  };
  auto superdiag = [&](int jp, hipStream_t stream) {
    const Panel p = panel(jp);
    // the kernel works on the compact copy of its block: matrix pointer = Cj, ld = 256, origin 0
    hipLaunchKernelGGL(ldlt_superdiag_kernel<true>, dim3(1), dim3(kBlock), 0, stream, p.Cj, (int64_t)LD_NB, 0, p.kbs, p.Vb, ldv,
                       dinv + p.K0, p.Dk_sp, p.Li_sp, d_info, (long long*)nullptr);
  };
  auto trsm = [&](int jp, hipStream_t stream, int col_ofs, int ncols) {
    if(ncols <= 0) return;
    const Panel p = panel(jp);
    if(p.kbs == LD_NB && col_ofs == 0 && ncols <= LD_NB) {   // the head, on the chain stream: four waves per 16 columns
      hipLaunchKernelGGL(ldlt_headtrsm_kernel, dim3((ncols + 15) / 16), dim3(kBlock), 0, stream, A, lda, N, p.K0, p.Vb, ldv, dinv,
                         p.Cj, p.Dk_sp, p.Li_sp, col_ofs);
      return;
    }
    hipLaunchKernelGGL(ldlt_supertrsm_kernel, dim3((ncols + 15) / 16), dim3(64), 0, stream, A, lda, N, p.K0, p.kbs, p.Vb, ldv,
                       dinv, p.Cj, p.Dk_sp, p.Li_sp, col_ofs);
  };
  // panel 0's diagonal block on the caller's stream, then fork
  int jp0 = 0;
  bool all_done = false;
  // (the chain kernel's DF_ROLES workgroups wait for each other: each needs a reserved CU of its own)
  bool use_df = df && df->enabled && df->flags && lookahead && lda == N && df->plan.N == N && df->plan.nchain >= 3 && ctx->chain_cus >= DF_ROLES &&
                ctx->wide_cus >= 8;
  DfDeviceLock df_lock;   // released when this call returns: the host has then waited for the info words' read-back, which is queued behind both dataflow kernels
  if(use_df && !df_lock.try_acquire()) {   // another process factorises on this device right now: stepwise kernels
    use_df = false;
    static bool told = false;
    if(!told) {
      told = true;
      std::fprintf(stderr, "[hiop_amd] another process holds this device's dataflow-LDL^T lock: this factorisation (and any other that finds the lock taken) runs the stepwise kernels\n");
    }
  }
  if(!use_df) {   // (the dataflow path does both in its one preparation launch, ldlt_df_prep_kernel)
    HIOPAMD_CHECK(hipMemsetAsync(d_info, 0, 4 * sizeof(int), st));
    const Panel p0 = panel(0);
    hipLaunchKernelGGL(ldlt_pack_diag_kernel, dim3(LD_NB), dim3(kBlock), 0, st, A, lda, p0.K0, p0.kbs, p0.Cj);
  }
  if(use_df) {
    // ---- dataflow factorisation of the chained super-panels: two persistent kernels, device flags only
    // One at a time per device: the chain kernel's 16 roles must all be resident together on the reserved CUs, so two
    // factorisations in flight (two solver objects of one process on different streams) could each get part of those CUs
    // and wait for each other.  Every dataflow factorisation of this process therefore starts behind the previous one's
    // completion event (a no-op on the same stream).  Other PROCESSES on the same device are not covered.
    hipEvent_t df_done = df_done_event();
    if(!df_done) return HIOPAMD_ERR_HIP;
    HIOPAMD_CHECK(hipStreamWaitEvent(st, df_done, 0));
    const DfPlan& P = df->plan;
    DfArgs a;
    a.A = A; a.lda = lda; a.N = N; a.V = V; a.nvb = df->nvb; a.ldv = ldv; a.dinv = dinv; a.Dblk = Dblk; a.Li = Li; a.Cd = Cd; a.info = d_info;
    a.flags = df->flags; a.nsp = P.nsp; a.nt = P.nt; a.nchain = P.nchain; a.last_has_next = P.last_has_next;
    a.off_chain = P.off_chain; a.off_tr = P.off_tr; a.off_ver = P.off_ver;
    a.ctasks = df->ctasks; a.wtasks = df->wtasks; a.nwtasks = (int)P.wtasks.size(); a.upcnt = df->upcnt;
    a.wq = df->wq; a.wf = df->wf; a.wfirst = df->wfirst; a.nwide = P.nwide;
    a.off_wg = P.off_wg;       // (the init kernel publishes these two offsets in the flags' header: set before its launch)
    a.off_snap = P.off_snap;
    a.off_where = P.off_where;
    {
      const int nzb = (int)(((int64_t)P.nflags + (int64_t)kBlock * 16 - 1) / ((int64_t)kBlock * 16));
      hipLaunchKernelGGL(ldlt_df_prep_kernel, dim3(LD_NB + nzb), dim3(kBlock), 0, st, a, (int64_t)P.nflags, panel(0).kbs);
    }
    int rc = dep(st, su);
    if(rc == HIOPAMD_OK) rc = dep(st, sd);
    if(rc != HIOPAMD_OK) return rc;
    a.spine_opt = DF_SPINE_OPT;
    a.has_far = P.has_far ? 1 : 0;
    a.dbg = std::getenv("HIOPAMD_DF_STAMPS") ? std::max(1, std::atoi(std::getenv("HIOPAMD_DF_STAMPS"))) : 0;   // profiling aid (1: all panels; 2 + j: phase sums of super-panel j only): per-super-panel time stamps, printed after the call
    a.off_ts = P.off_ver + (int64_t)P.nt * P.nt;
    a.off_ph = a.off_ts + 8 * (int64_t)(P.nsp + 1);
    a.off_trb = P.off_trb;
    a.off_cu = P.off_cu;
    a.off_wg = P.off_wg;
    a.off_snap = P.off_snap;
    {   // limit of every bounded wait: 20x a pessimistic estimate of the whole factorisation (20 TFLOP/s + 1 ms), at least 250 ms, at most
        // 3 s; HIOPAMD_DF_TIMEOUT_MS overrides.  (A wait that is served never lasts longer than the factorisation itself.)
      static const double env_ms = std::getenv("HIOPAMD_DF_TIMEOUT_MS") ? std::atof(std::getenv("HIOPAMD_DF_TIMEOUT_MS")) : 0.0;
      const double est_s = (double)N * N * N / 3.0 / 20e12 + 1e-3;
      const double lim_s = env_ms > 0.0 ? env_ms * 1e-3 : std::min(3.0, std::max(0.25, 20.0 * est_s));
      a.timeout_ticks = (long long)(lim_s * 1e8);
    }
    static const bool df_check = std::getenv("HIOPAMD_DF_CHECK") && std::atoi(std::getenv("HIOPAMD_DF_CHECK")) != 0;
    a.off_run = df_check ? P.off_run : 0;
    static const bool df_debug_sh = std::getenv("HIOPAMD_DF_DEBUG") && std::atoi(std::getenv("HIOPAMD_DF_DEBUG")) != 0;
    a.off_shadow = df_debug_sh ? P.off_shadow : 0;
    // ONE workgroup of the wide kernel per CU — enforced (round 5).  A grid of exactly wide_cus workgroups is NOT dealt one per CU: the
    // rank-on-CU counters of 117 factorisations showed anything from 240 CUs with one workgroup each to 224 CUs used and 16 of them
    // carrying two for the whole factorisation (profiles/r05_probes/README.md) — idle CUs in the update-bound half, and the two-per-CU
    // residency the one-per-CU shape was chosen to avoid.  So TWICE as many workgroups are launched (two fit a CU) and every workgroup
    // that is not the first on its CU leaves at once (jretire = 0: the kernel's rank test, csrc/ldlt_wide_body.inc): 240 of 240 CUs carry
    // exactly one working workgroup in every sample, linsolv.tmFactTime 5.27-5.32 ms against 5.35-5.51.
    a.jretire = 0;
    // HIOPAMD_DF_ONE=1 (measurement aid): chain + wide as ONE dispatch on the wide stream, one workgroup per CU — the form the
    // rocprofv3 counter passes can profile (see ldlt_df_one_kernel); needs the 16-byte tile form
    static const bool df_one = std::getenv("HIOPAMD_DF_ONE") && std::atoi(std::getenv("HIOPAMD_DF_ONE")) != 0;
    const bool one = df_one && (N % 2 == 0) && N >= 2 * UD_T && a.nwtasks > 0;
    if(one) {
      a.dbg = 0;
      // on the context's own stream (no CU mask): DF_ROLES + 240 workgroups of one per CU are all resident on 256 CUs
      if(timed) (void)hipEventRecord(prof->get(), st);
      hipLaunchKernelGGL(ldlt_df_one_kernel, dim3(DF_ROLES + ctx->wide_cus), dim3(kBlock), 0, st, a);
      if(timed) {
        (void)hipEventRecord(prof->get(), st);
        prof->flops += P.up_flops;
        prof->launches += 1;
      }
    } else {
    hipLaunchKernelGGL(ldlt_chain_kernel, dim3(DF_ROLES), dim3(kBlock), 0, sd, a);
    if(a.nwtasks > 0) {
      if(timed) (void)hipEventRecord(prof->get(), su);
      // ONE working workgroup of the wide kernel per CU of the wide stream (ctx->wide_cus, 240 on MI355X with two reserved CUs per XCD):
      // 73.7 KB of LDS and one wave per SIMD, i.e. every CU could take a second one — the only shape that never froze in a soak
      // (DF_WIDE_WG_PER_CU = 2: the second workgroup of every CU leaves at once, see jretire above; = 1: the kernel is compiled for one
      //  wave per SIMD and uses more than 256 registers per lane, so a CU CANNOT hold a second workgroup: a grid of wide_cus is one per CU)
      const int wmax = DF_WIDE_WG_PER_CU * ctx->wide_cus;
      const int grid = a.nwtasks < wmax ? a.nwtasks : wmax;
      // 16-byte accesses need even N, lda, ldv (ldv = N); otherwise the 8-byte tile form
      const bool form2 = (N % 2 == 0) && (lda % 2 == 0) && (ldv % 2 == 0) && N >= 2 * UD_T;
      if(form2 && a.dbg) hipLaunchKernelGGL((ldlt_wide_kernel<2, true>), dim3(grid), dim3(kBlock), 0, su, a);
      else if(form2) hipLaunchKernelGGL((ldlt_wide_kernel<2, false>), dim3(grid), dim3(kBlock), 0, su, a);
      else hipLaunchKernelGGL((ldlt_wide_kernel<1, false>), dim3(grid), dim3(kBlock), 0, su, a);
      if(timed) {
        (void)hipEventRecord(prof->get(), su);
        prof->flops += P.up_flops;
        prof->launches += 1;
      }
    }
    }
    rc = dep(su, st);
    if(rc == HIOPAMD_OK) rc = dep(sd, st);
    if(rc != HIOPAMD_OK) return rc;
    HIOPAMD_CHECK(hipEventRecord(df_done, st));
    jp0 = P.nchain;
    if(jp0 >= nsp) all_done = true;
    else superdiag(jp0, st);   // the first super-panel left to the stepwise kernels (its block holds every update)
  } else {
    superdiag(0, st);
  }
  if(lookahead && !all_done) {
    int rc = dep(st, su);
    if(rc == HIOPAMD_OK) rc = dep(st, sd);
    if(rc != HIOPAMD_OK) return rc;
  }
  hipEvent_t ev_rest_prev = nullptr;   // upd_rest(j-1) done (recorded on su)
  for(int jp = jp0; jp < nsp && !all_done; ++jp) {
    const Panel p = panel(jp);
    if(p.Kend >= N) break;
    const int s = p.Kend;
    const int head = (N - s < LD_NB) ? (N - s) : LD_NB;   // columns of the next super-panel
    const int sa_end = s + head;
    // ---- chain stream
    if(lookahead && ev_rest_prev) HIOPAMD_CHECK(hipStreamWaitEvent(sd, ev_rest_prev, 0));   // rows of panel jp complete
    trsm(jp, sd, 0, head);
    hipEvent_t ev_head = nullptr;
    if(lookahead) {
      ev_head = next_event();
      HIOPAMD_CHECK(hipEventRecord(ev_head, sd));
    }
    {
      // the next diagonal block (the only update on the chain): the tile kernel with 32 x 64 tiles = 20 live workgroups for
      // the 16 reserved CUs, written to the matrix and to the block's compact copy
      double* Cn = panel(jp + 1).Cj;
      hipLaunchKernelGGL((ldlt_update_kernel_t<1, 2, 8>), dim3(4, 8), dim3(kBlock), 0, sd, A, lda, N, p.Vb, ldv, 0, p.K0, p.kbs, s,
                         sa_end, sa_end, 0, Cn);
    }
    superdiag(jp + 1, sd);
    hipEvent_t ev_diag = nullptr;
    if(lookahead) {
      ev_diag = next_event();
      HIOPAMD_CHECK(hipEventRecord(ev_diag, sd));
    }
    // ---- wide stream: (superdiag(jp) is already waited for: fork for jp = 0, ev_diag of the previous iteration below)
    trsm(jp, su, head, N - sa_end);
    if(lookahead) HIOPAMD_CHECK(hipStreamWaitEvent(su, ev_head, 0));
    {
      const int t = (N - s + UD_T - 1) / UD_T;
      upd_stream = su;
      launch_update(t, p.Vb, p.K0, p.kbs, s, N, N, 1);
    }
    if(lookahead) {
      ev_rest_prev = next_event();
      HIOPAMD_CHECK(hipEventRecord(ev_rest_prev, su));
      HIOPAMD_CHECK(hipStreamWaitEvent(su, ev_diag, 0));   // trsm_tail(jp+1) needs superdiag(jp+1)
    }
  }
  if(lookahead && !all_done) {   // join
    int rc = dep(su, st);
    if(rc == HIOPAMD_OK) rc = dep(sd, st);
    if(rc != HIOPAMD_OK) return rc;
  }
  hipLaunchKernelGGL(ldlt_unpack_diag_kernel, dim3(64, nsp), dim3(kBlock), 0, st, Cd, A, lda, N, Cd + cdt_ofs);
  span_begin(ctx, HIOPAMD_SPAN_LINSOLV_INERTIA);   // :127-167 (tmInertiaComp)
  hipLaunchKernelGGL(ldlt_inertia_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, N, A, lda, d_info + 1);   // (d_info was zeroed at the start)
  span_end(ctx, HIOPAMD_SPAN_LINSOLV_INERTIA);
  // The results come back through a PINNED host buffer (one per thread, allocated once): these copies are enqueued while the dataflow
  // kernels are still running, and a copy into pageable memory can make the runtime map / pin pages at that moment — a change of the
  // process's GPU mappings, for which the kernel driver may preempt and restore the process's queues.  Persistent kernels whose
  // workgroups wait for each other do not survive a partial restore ("workgroups that freeze in mid-task"; tried: no effect on the rate, kept as the cheaper form).
  static thread_local unsigned* pinned = nullptr;
  if(!pinned) HIOPAMD_CHECK(hipHostMalloc((void**)&pinned, 32 * sizeof(unsigned), hipHostMallocDefault));
  int* h = reinterpret_cast<int*>(pinned);
  unsigned* dfw = pinned + 8;
  for(int q = 0; q < 16; ++q) dfw[q] = 0u;
  HIOPAMD_CHECK(hipMemcpyAsync(h, d_info, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  if(use_df) HIOPAMD_CHECK(hipMemcpyAsync(dfw, df->flags, 16 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  HIOPAMD_CHECK(hipEventRecord(ctx_named_event(ctx->ev_info), st));
  if(Winv) hipLaunchKernelGGL(ldlt_inv_diag_kernel, dim3(16, nsp), dim3(64), 0, st, N, Cd, Dblk, Li, Winv, wB);
  if(Winv && wB == 512 && Wt) {
    // T = U_ab W_b, then the upper-right quadrant = -W_a T   (N is a multiple of 512 here; U_ab = A[rows of a, columns of b])
    const int np2 = N / 512;
    const int64_t w2 = 512 * 512;
    hipLaunchKernelGGL(ldlt_w512_mm_kernel, dim3(64, np2), dim3(kBlock), 0, st, A + LD_NB, (int64_t)512 * lda + 512, lda,
                       Winv + (int64_t)LD_NB * 512 + LD_NB, w2, (int64_t)512, Wt, (int64_t)LD_NB * LD_NB, (int64_t)LD_NB, 1.0);
    hipLaunchKernelGGL(ldlt_w512_mm_kernel, dim3(64, np2), dim3(kBlock), 0, st, Winv, w2, (int64_t)512, Wt, (int64_t)LD_NB * LD_NB,
                       (int64_t)LD_NB, Winv + LD_NB, w2, (int64_t)512, -1.0);
  }
  HIOPAMD_CHECK(hipGetLastError());
  // The host waits for the info words' read-back, NOT for the stream: the kernels above (inverted diagonal blocks, 512-block products: ~0.14 ms)
  // only feed the next solve, which is queued behind them on the same stream — meanwhile the caller has its inertia and goes on enqueueing.
  HIOPAMD_CHECK(hipEventSynchronize(ctx->ev_info));
  // (profiling time lines, the HIOPAMD_DF_CHECK verification and the diagnostics + HIOPAMD_ERR_TIMEOUT return of an expired wait)
#include "ldlt_df_report.inc"
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
                           double* rhs, int nrhs, const double* Cd = nullptr, hiopamd_linsolver* flow = nullptr)
{
  if(N < 0 || nrhs < 0) return HIOPAMD_ERR_ARG;
  if(N == 0) return HIOPAMD_OK;
  hipStream_t st = ctx->stream;
  const int nblk = (N + LD_nb - 1) / LD_nb;
  if(flow && flow->W && flow->flow_enabled) {
    const int nb = flow->fl_nb, FB = flow->fl_B;
    for(int j = 0; j < nrhs; ++j) {
      flow->fl_epoch += 1;
      // exchange buffers of this epoch's parity (c) and of the next launch (n): y | x | product slots, twice
      const int64_t npad = (int64_t)nb * FB, psz = (int64_t)FB * nb * nb, half = 2 * npad + psz;
      double* cur = flow->P + (int64_t)(flow->fl_epoch & 1ull) * half;
      double* nxt = flow->P + (int64_t)((flow->fl_epoch + 1ull) & 1ull) * half;
      double *yc = cur, *xcur = cur + npad, *Pc = cur + 2 * npad;
      double *yn = nxt, *xn = nxt + npad, *Pn = nxt + 2 * npad;
      // HIOPAMD_SOLVE_STAMPS=1 (profiling aid): per-task time stamps of every solve, summarised on stderr after a
      // synchronisation (start | input vector there | result published, 100 MHz ticks)
      static const bool stamps = std::getenv("HIOPAMD_SOLVE_STAMPS") && std::atoi(std::getenv("HIOPAMD_SOLVE_STAMPS")) != 0;
      long long* ts = nullptr;
      if(stamps) {
        if(hipMalloc((void**)&ts, sizeof(long long) * 4 * (size_t)flow->fl_ntasks) != hipSuccess) ts = nullptr;
        else (void)hipMemsetAsync(ts, 0, sizeof(long long) * 4 * (size_t)flow->fl_ntasks, st);
      }
      if(FB == 512)
        hipLaunchKernelGGL(ldlt_solve_flow_kernel<512>, dim3(flow->fl_ntasks), dim3(512), 0, st, A, lda, N, nb, flow->W, dinv,
                           flow->fl_tasks, flow->fl_ntasks, flow->fl_sync, flow->fl_epoch, Pc, yc, xcur, Pn, yn, xn,
                           rhs + (int64_t)j * N, ts, flow->fl_lead);
      else
        hipLaunchKernelGGL(ldlt_solve_flow_kernel<256>, dim3(flow->fl_ntasks), dim3(256), 0, st, A, lda, N, nb, flow->W, dinv,
                           flow->fl_tasks, flow->fl_ntasks, flow->fl_sync, flow->fl_epoch, Pc, yc, xcur, Pn, yn, xn,
                           rhs + (int64_t)j * N, ts, flow->fl_lead);
      if(ts) {
        std::vector<long long> h(4 * (size_t)flow->fl_ntasks);
        std::vector<int4> tk((size_t)flow->fl_ntasks);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), ts, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpy(tk.data(), flow->fl_tasks, sizeof(int4) * tk.size(), hipMemcpyDeviceToHost);
        (void)hipFree(ts);
        long long t0 = h[0];
        for(size_t i = 0; i < tk.size(); ++i) t0 = std::min(t0, h[4 * i]);
        double sum[4][3] = {{0}};   // per kind: resident before the input arrives | input -> published | count
        std::vector<double> ydone(nb, 0.0), xdone(nb, 0.0);
        for(size_t i = 0; i < tk.size(); ++i) {
          const int k = tk[i].x;
          sum[k][0] += (h[4 * i + 1] - h[4 * i]) * 0.01;
          sum[k][1] += (h[4 * i + 2] - h[4 * i + 1]) * 0.01;
          sum[k][2] += 1.0;
          const double done = (h[4 * i + 2] - t0) * 0.01;
          if(k == FL_FWD_DIAG) ydone[tk[i].z] = std::max(ydone[tk[i].z], done);
          if(k == FL_BWD_DIAG) xdone[tk[i].y] = std::max(xdone[tk[i].y], done);
        }
        const char* nm[4] = {"fwd product", "fwd diagonal", "bwd product", "bwd diagonal"};
        for(int k = 0; k < 4; ++k)
          if(sum[k][2] > 0)
            std::fprintf(stderr, "[hiop_amd] solve %-12s %5.0f tasks: resident before the input arrived %7.2f us, input -> published %5.2f us\n",
                         nm[k], sum[k][2], sum[k][0] / sum[k][2], sum[k][1] / sum[k][2]);
        std::fprintf(stderr, "[hiop_amd] solve chain: y_J published at (us):");
        for(int J = 0; J < nb; J += 4) std::fprintf(stderr, " %d:%.0f", J, ydone[J]);
        std::fprintf(stderr, " | last %.0f\n[hiop_amd] solve chain: x_I published at (us):", ydone[nb - 1]);
        for(int I = nb - 1; I >= 0; I -= 4) std::fprintf(stderr, " %d:%.0f", I, xdone[I]);
        std::fprintf(stderr, " | last %.0f\n", xdone[0]);
      }
      if(hipGetLastError() != hipSuccess) {
        // the launch did not happen: the device-side flags / poison parity did not advance, neither may the host's epoch
        flow->fl_epoch -= 1;
        std::fprintf(stderr, "[hiop_amd] dataflow solve: launch failed, falling back to the stepwise solve\n");
        flow->flow_enabled = false;
        return ldlt_solve_impl(ctx, N, A, lda, dinv, ybuf, rhs + (int64_t)j * N, nrhs - j, Cd, flow);
      }
    }
    flow->flow_dirty = true;   // the error word is looked at by the next synchronising call (flow_check)
    return HIOPAMD_OK;
  }
  if(Cd) {   // 256-row steps on the compact diagonal blocks (the factorisation object keeps them)
    const int nsp = (N + SV_B - 1) / SV_B;
    const double* CdT = Cd + (int64_t)nsp * (SV_B * SV_B);
    for(int j = 0; j < nrhs; ++j) {
      double* b = rhs + (int64_t)j * N;
      hipLaunchKernelGGL(ldlt_fwd_step256, dim3(1), dim3(SV_T), 0, st, A, lda, N, -1, Cd, b, ybuf);
      for(int I = 0; I + 1 < nsp; ++I) {
        const int i0 = I * SV_B;
        const int g = (N - i0 - SV_B + SV_B - 1) / SV_B;
        hipLaunchKernelGGL(ldlt_fwd_step256, dim3(g), dim3(SV_T), 0, st, A, lda, N, i0, Cd, b, ybuf);
      }
      int rc = hiopamd_vec_component_mult(ctx, N, ybuf, dinv);
      if(rc != HIOPAMD_OK) return rc;
      const int iL = (nsp - 1) * SV_B;
      hipLaunchKernelGGL(ldlt_bwd_step256, dim3(1), dim3(SV_T), 0, st, A, lda, iL, N - iL, CdT, ybuf, b, 1);
      for(int I = nsp - 1; I >= 1; --I) {
        const int i0 = I * SV_B;
        const int ib = (i0 + SV_B <= N) ? SV_B : (N - i0);
        hipLaunchKernelGGL(ldlt_bwd_step256, dim3(I), dim3(SV_T), 0, st, A, lda, i0, ib, CdT, ybuf, b, 0);
      }
    }
    HIOPAMD_CHECK(hipGetLastError());
    return HIOPAMD_OK;
  }
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
  const size_t vbytes = sizeof(double) * (size_t)LD_NB * nn * 2;
  const size_t dbytes = sizeof(double) * (size_t)(LD_nb * LD_nb + 4 * LD_SB * LD_SB) * ((nn + LD_nb - 1) / LD_nb);
  const size_t cbytes = sizeof(double) * 2 * (size_t)LD_NB * LD_NB * ((nn + LD_NB - 1) / LD_NB);
  char* w = (char*)ctx_workspace(ctx, vbytes + dbytes + cbytes + 64);
  return ldlt_factor_impl(ctx, n, A, lda, work_dinv, (double*)w, (double*)(w + vbytes), (double*)(w + vbytes + dbytes),
                          (int*)(w + vbytes + dbytes + cbytes), inertia3_host);
}

int hiopamd_ldlt_solve(hiopamd_ctx* ctx, int n, const double* A, int64_t lda, const double* work_dinv,
                       double* rhs_inout, int nrhs)
{
  double* y = (double*)ctx_workspace(ctx, sizeof(double) * (size_t)(n > 0 ? n : 1));
  return ldlt_solve_impl(ctx, n, A, lda, work_dinv, y, rhs_inout, nrhs);
}

static int linsolver_create_impl(hiopamd_linsolver* ls, hiopamd_ctx* ctx, int n);
static int linsolver_ensure_pad(hiopamd_linsolver* ls);

int hiopamd_linsolver_create(hiopamd_linsolver** out, hiopamd_ctx* ctx, int n)
{
  if(!out || !ctx || n < 0) return HIOPAMD_ERR_ARG;
  *out = nullptr;
  hiopamd_linsolver* ls = new hiopamd_linsolver();
  const int rc = linsolver_create_impl(ls, ctx, n);
  if(rc != HIOPAMD_OK) {   // nothing leaks: destroy tolerates the members that were never allocated
    (void)hiopamd_linsolver_destroy(ls);
    return rc;
  }
  *out = ls;
  return HIOPAMD_OK;
}

// The order a solver object of order n works at (hiopamd_linsolver::npad): n itself, n made even, the next multiple of 256 or the next
// multiple of 512 — the larger ones only when a cost model of one factorisation + three solves + the copies calls them at least 5 %
// cheaper than the even order.  The model's constants are this machine's measurements (ms; scripts/factor_time.py,
// profiles/r06_probes/call20-24_*), x = order / 8192: the dataflow factorisation max(5.2 x^3, 0.1 per super-panel) — throughput above
// ~6000, the chain's latency below —, x 1.7 in the 8-byte tile form of an odd order, + 0.15 + 0.5 x^2 for the stepwise tail of an order
// that is not a multiple of 256; a solve = 11 us per block step (512-row blocks for multiples of 512 from 2048 on, else 256) + 0.05 x^2;
// copying the triangle in and out 0.27 x^2 (against 0.13 x^2 for the retry copy of an unpadded order).
// HIOPAMD_LDLT_PAD=0: never pad, =1: parity only (timing comparisons).
static int ldlt_padded_order(int n)
{
  if(n < LD_PAD_MIN) return n;
  int mode = 2;
  if(const char* e = std::getenv("HIOPAMD_LDLT_PAD")) mode = std::atoi(e);
  if(mode <= 0) return n;
  auto cost = [&](int c) {
    const double x = c / 8192.0;
    double t = std::max(5.2 * x * x * x, 0.1 * c / LD_NB) * ((c & 1) ? 1.7 : 1.0) + ((c % LD_NB) ? 0.15 + 0.5 * x * x : 0.0);
    const int FB = (c >= 2048 && c % 512 == 0) ? 512 : 256;
    t += 3.0 * (0.011 * ((c + FB - 1) / FB) + 0.05 * x * x);
    t += (c != n ? 0.27 : 0.13) * x * x;
    return t;
  };
  const int even = n + (n & 1);
  int best = even;
  if(mode >= 2)
    for(int c : {(n + 255) / 256 * 256, (n + 511) / 512 * 512})
      if(cost(c) < 0.95 * cost(even) && cost(c) < cost(best)) best = c;
  return best;
}

static int linsolver_create_impl(hiopamd_linsolver* ls, hiopamd_ctx* ctx, int n)
{
  ls->ctx = ctx;
  ls->n = n;
  ls->npad = ldlt_padded_order(n);
  const size_t nn = (size_t)(n > 0 ? n : 1);
  const size_t nf = (size_t)(ls->npad > 0 ? ls->npad : 1);   // the factorisation's workspaces: the padded order
  HIOPAMD_CHECK(hipMalloc((void**)&ls->M, sizeof(double) * nn * nn));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->dinv, sizeof(double) * nf));
  ls->nvb = df_nvb_for(ls->npad);
  ls->df.nvb = ls->nvb;
  HIOPAMD_CHECK(hipMalloc((void**)&ls->V, sizeof(double) * nf * LD_NB * (size_t)ls->nvb));   // row panels of nvb consecutive super-panels
  HIOPAMD_CHECK(hipMalloc((void**)&ls->ybuf, sizeof(double) * nf));
  if(ls->npad > n) HIOPAMD_CHECK(hipMalloc((void**)&ls->xpad, sizeof(double) * nf));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->Dblk, sizeof(double) * (LD_nb * LD_nb + 4 * LD_SB * LD_SB) * ((nf + LD_nb - 1) / LD_nb)));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->Cd, sizeof(double) * 2 * (size_t)LD_NB * LD_NB * ((nf + LD_NB - 1) / LD_NB)));   // blocks + transposes
  // The compact blocks are filled by the kernels that produce them with the entries that EXIST (upper triangle, rows and columns below
  // the order): the padding of a ragged last block is defined here, once — zero, and no kernel writes it afterwards (round 6: it was
  // whatever the allocation held, and the block inversion read it; see DESIGN.md 3.1, "the round-5 gate failure")
  HIOPAMD_CHECK(hipMemsetAsync(ls->Cd, 0, sizeof(double) * 2 * (size_t)LD_NB * LD_NB * ((nf + LD_NB - 1) / LD_NB), ctx->stream));
  HIOPAMD_CHECK(hipMalloc((void**)&ls->d_info, 64));
  {
    // dataflow solve: inverted diagonal blocks, product slots, flags, and the task list in issue order
    // block size of the task graph: 512 for orders that are multiples of 512 (half the serial block steps), 256 otherwise
    const int FB = (ls->npad >= 2048 && ls->npad % 512 == 0) ? 512 : SV_B;
    const int FR = FB / 64;
    const int nb = (int)((nf + FB - 1) / FB);
    ls->fl_B = FB;
    ls->fl_nb = nb;
    ls->fl_lead = (FB == 512 ? 2 : FL_LEAD);
    HIOPAMD_CHECK(hipMalloc((void**)&ls->W, sizeof(double) * (size_t)FB * FB * nb));
    if(FB == 512) {
      HIOPAMD_CHECK(hipMemsetAsync(ls->W, 0, sizeof(double) * (size_t)FB * FB * nb, ctx->stream));   // the lower-left quadrants stay zero
      HIOPAMD_CHECK(hipMalloc((void**)&ls->Wt, sizeof(double) * (size_t)SV_B * SV_B * nb));
    }
    {
      // two copies of (y | x | product slots), poisoned (flow_poll)
      const size_t words = 2 * ((size_t)2 * nb * FB + (size_t)FB * nb * nb);
      HIOPAMD_CHECK(hipMalloc((void**)&ls->P, sizeof(double) * words));
      std::vector<unsigned long long> poison(words, FL_POISON);
      HIOPAMD_CHECK(hipMemcpy(ls->P, poison.data(), sizeof(double) * words, hipMemcpyHostToDevice));
    }
    HIOPAMD_CHECK(hipMalloc((void**)&ls->fl_sync, sizeof(unsigned long long) * (size_t)(2 + 8 * nb)));   // + the error word
    HIOPAMD_CHECK(hipMemsetAsync(ls->fl_sync, 0, sizeof(unsigned long long) * (size_t)(2 + 8 * nb), ctx->stream));
    std::vector<int4> tk;
    tk.reserve((size_t)FR * nb * (nb + 1));
    for(int J = 0; J < nb; ++J) {   // forward: column J needs y_I, I < J; its diagonal task closes it
      for(int I = 0; I < J; ++I)
        for(int c = 0; c < FR; ++c) tk.push_back(make_int4(FL_FWD_OFF, I, J, c));
      for(int c = 0; c < FR; ++c) tk.push_back(make_int4(FL_FWD_DIAG, J, J, c));
    }
    for(int I = nb - 1; I >= 0; --I) {   // backward: row I needs x_J, J > I
      for(int J = nb - 1; J > I; --J)
        for(int c = 0; c < FR; ++c) tk.push_back(make_int4(FL_BWD_OFF, I, J, c));
      for(int c = 0; c < FR; ++c) tk.push_back(make_int4(FL_BWD_DIAG, I, I, c));
    }
    ls->fl_ntasks = (int)tk.size();
    HIOPAMD_CHECK(hipMalloc((void**)&ls->fl_tasks, sizeof(int4) * tk.size()));
    HIOPAMD_CHECK(hipMemcpy(ls->fl_tasks, tk.data(), sizeof(int4) * tk.size(), hipMemcpyHostToDevice));
  }
  HIOPAMD_CHECK(hipMemsetAsync(ls->M, 0, sizeof(double) * nn * nn, ctx->stream));
  {
    // dataflow factorisation: task tables and flags (HIOPAMD_DF=0 in the environment selects the stepwise kernels)
    DfDevice& df = ls->df;
    df.plan = df_build_plan(ls->npad);
    df.enabled = !(std::getenv("HIOPAMD_DF") && std::atoi(std::getenv("HIOPAMD_DF")) == 0);
    const DfPlan& P = df.plan;
    if(P.nchain >= 3) {
      HIOPAMD_CHECK(hipMalloc((void**)&df.flags, sizeof(unsigned) * (size_t)P.nflags));
      HIOPAMD_CHECK(hipMalloc((void**)&df.ctasks, sizeof(int4) * P.ctasks.size()));
      HIOPAMD_CHECK(hipMalloc((void**)&df.wtasks, sizeof(int4) * (P.wtasks.size() + 1)));
      HIOPAMD_CHECK(hipMalloc((void**)&df.upcnt, sizeof(unsigned) * P.upcnt.size()));
      HIOPAMD_CHECK(hipMemcpy(df.ctasks, P.ctasks.data(), sizeof(int4) * P.ctasks.size(), hipMemcpyHostToDevice));
      if(!P.wtasks.empty())
        HIOPAMD_CHECK(hipMemcpy(df.wtasks, P.wtasks.data(), sizeof(int4) * P.wtasks.size(), hipMemcpyHostToDevice));
      HIOPAMD_CHECK(hipMemcpy(df.upcnt, P.upcnt.data(), sizeof(unsigned) * P.upcnt.size(), hipMemcpyHostToDevice));
      HIOPAMD_CHECK(hipMalloc((void**)&df.wq, sizeof(int4) * P.wq.size()));
      HIOPAMD_CHECK(hipMalloc((void**)&df.wfirst, sizeof(unsigned) * P.wfirst.size()));
      HIOPAMD_CHECK(hipMemcpy(df.wq, P.wq.data(), sizeof(int4) * P.wq.size(), hipMemcpyHostToDevice));
      HIOPAMD_CHECK(hipMalloc((void**)&df.wf, sizeof(int4) * P.wf.size()));
      HIOPAMD_CHECK(hipMemcpy(df.wf, P.wf.data(), sizeof(int4) * P.wf.size(), hipMemcpyHostToDevice));
      HIOPAMD_CHECK(hipMemcpy(df.wfirst, P.wfirst.data(), sizeof(unsigned) * P.wfirst.size(), hipMemcpyHostToDevice));
    }
  }
  ls->flow_enabled = !(std::getenv("HIOPAMD_SOLVE_FLOW") && std::atoi(std::getenv("HIOPAMD_SOLVE_FLOW")) == 0);
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
  (void)hipFree(ls->Msave);
  (void)hipFree(ls->Mretry);
  (void)hipFree(ls->Mpad);
  (void)hipFree(ls->xpad);
  (void)hipFree(ls->rbuf);
  (void)hipFree(ls->Dblk);
  (void)hipFree(ls->Cd);
  (void)hipFree(ls->d_info);
  (void)hipFree(ls->W);
  (void)hipFree(ls->Wt);
  (void)hipFree(ls->P);
  (void)hipFree(ls->fl_sync);
  (void)hipFree(ls->fl_tasks);
  (void)hipFree(ls->df.flags);
  (void)hipFree(ls->df.ctasks);
  (void)hipFree(ls->df.wtasks);
  (void)hipFree(ls->df.upcnt);
  (void)hipFree(ls->df.wq);
  (void)hipFree(ls->df.wfirst);
  (void)hipFree(ls->df.wf);
  (void)hiopamd_ldlt_bk_destroy(ls->bk);
  delete ls;
  return HIOPAMD_OK;
}

double* hiopamd_linsolver_sys_matrix(hiopamd_linsolver* ls) { return ls ? ls->M : nullptr; }

// Where a caller that can assemble at ANY pitch should write the upper triangle of the next matrix: for an object that works at a
// padded order, the padded copy itself (*ld_out = that order) — the next hiopamd_linsolver_matrix_changed then factors it where it is.
// For every other object (and in safe / pivoted mode) the n x n matrix of hiopamd_linsolver_sys_matrix, *ld_out = n.  To be called
// before every such assembly (it arms ONE matrixChanged).  After a factorisation of this kind hiopamd_linsolver_sys_matrix does not show
// the factor (hiopamd_linsolver_sys_matrix_sync copies it there), the assembled matrix is overwritten and there is no retry copy: an
// expired wait is reported as HIOPAMD_ERR_TIMEOUT and the caller assembles again — what the native KKT objects do anyway.
int hiopamd_linsolver_assembly_matrix(hiopamd_linsolver* ls, double** M_out, int64_t* ld_out)
{
  if(!ls || !M_out || !ld_out) return HIOPAMD_ERR_ARG;
  if(ls->npad > ls->n && !ls->safe_mode && !ls->pivoted) {
    const int rc = linsolver_ensure_pad(ls);
    if(rc != HIOPAMD_OK) return rc;
    ls->direct_armed = true;
    *M_out = ls->Mpad;
    *ld_out = ls->npad;
    return HIOPAMD_OK;
  }
  ls->direct_armed = false;
  *M_out = ls->M;
  *ld_out = ls->n;
  return HIOPAMD_OK;
}

// the n x n view of hiopamd_linsolver_sys_matrix brought up to date with the padded copy (upper triangle): the matrix as assembled
// through hiopamd_linsolver_assembly_matrix, or — after matrixChanged — its factor.  A no-op for every other object.  Asynchronous on
// the context's stream.
int hiopamd_linsolver_sys_matrix_sync(hiopamd_linsolver* ls)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(!(ls->Mpad && (ls->direct_armed || ls->factor_in_pad))) return HIOPAMD_OK;
  const int n = ls->n;
  const dim3 tgrid((unsigned)((n + 127) / 128), (unsigned)((n + 127) / 128));
  hipLaunchKernelGGL(ldlt_triu_repitch_kernel, tgrid, dim3(kBlock), 0, ls->ctx->stream, n, ls->Mpad, (int64_t)ls->npad, ls->M, (int64_t)n);
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}
int hiopamd_linsolver_n(const hiopamd_linsolver* ls) { return ls ? ls->n : -1; }

// Synchronises the context's stream and looks at the dataflow solve's error word (a bounded wait timed out).  On error the
// exchange state is re-initialised (flags zeroed, buffers re-poisoned, epoch restarted) and the stepwise solve takes over.
static int flow_check(hiopamd_linsolver* ls, int* ok_host)
{
  if(ok_host) *ok_host = 1;
  if(!ls->flow_dirty || !ls->fl_sync) return HIOPAMD_OK;
  const int nb = ls->fl_nb, FB = ls->fl_B;
  unsigned long long err = 0;
  HIOPAMD_CHECK(hipMemcpyAsync(&err, ls->fl_sync + 1 + 8 * nb, sizeof(err), hipMemcpyDeviceToHost, ls->ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ls->ctx->stream));
  ls->flow_dirty = false;
  if(err == 0) return HIOPAMD_OK;
  if(ok_host) *ok_host = 0;
  ls->flow_failed = true;
  std::fprintf(stderr, "[hiop_amd] dataflow solve: a bounded wait timed out; the results of the solves since the last check are "
                       "invalid.  Exchange state re-initialised, stepwise solve from now on.\n");
  HIOPAMD_CHECK(hipMemsetAsync(ls->fl_sync, 0, sizeof(unsigned long long) * (size_t)(2 + 8 * nb), ls->ctx->stream));
  const size_t words = 2 * ((size_t)2 * nb * FB + (size_t)FB * nb * nb);
  std::vector<unsigned long long> poison(words, FL_POISON);
  HIOPAMD_CHECK(hipMemcpy(ls->P, poison.data(), sizeof(double) * words, hipMemcpyHostToDevice));
  ls->fl_epoch = 0;
  ls->flow_enabled = false;
  return HIOPAMD_OK;
}

int hiopamd_linsolver_solve_status(hiopamd_linsolver* ls, int* ok_host)
{
  if(!ls || !ok_host) return HIOPAMD_ERR_ARG;
  const int rc = flow_check(ls, ok_host);
  if(rc == HIOPAMD_OK && ls->flow_failed) {   // (a time-out seen by an earlier synchronising call)
    *ok_host = 0;
    ls->flow_failed = false;
  }
  if(rc == HIOPAMD_OK && ls->safe_mode && ls->safe_solve_failed) {   // a safe-mode refinement did not converge since the last check
    *ok_host = 0;
    ls->safe_solve_failed = false;
  }
  return rc;
}

int hiopamd_linsolver_last_solve_ok(hiopamd_linsolver* ls, int* ok_host)
{
  if(!ls || !ok_host) return HIOPAMD_ERR_ARG;
  *ok_host = (ls->flow_failed || (ls->safe_mode && ls->safe_solve_failed)) ? 0 : 1;
  return HIOPAMD_OK;
}

int hiopamd_linsolver_set_solve_dataflow(hiopamd_linsolver* ls, int enable)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  int ok = 1;
  int rc = flow_check(ls, &ok);
  if(rc != HIOPAMD_OK) return rc;
  ls->flow_enabled = enable != 0;
  return HIOPAMD_OK;
}

// ---- safe mode -----------------------------------------------------------------------------------------------
// The reference answers a misbehaving no-pivot factorisation by switching the KKT object to a Bunch-Kaufman solver
// ("safe mode": src/Optimization/hiopKKTLinSysMDS.cpp:408-430, hiopAlgFilterIPM.cpp:2400-2427, the BuKa class
// src/LinAlg/hiopLinSolverSymDenseMagma.cpp:142-250).  Pivoting inside a 256-row dataflow would serialise the factorisation;
// the condensed KKT matrices are quasi-definite ([H + Dx, J^T; J, -Dd^-1]), for which the classical alternative is a static
// regularisation K_delta = K + delta * diag(+I_npos, -I_rest) — LDL^T of a quasi-definite matrix exists for EVERY symmetric
// permutation and its element growth is bounded by ~ ||K|| / delta (Vanderbei 1995; Gill, Saunders, Shinnerl 1996) — followed
// by iterative refinement against the ORIGINAL matrix.  With delta = sqrt(eps) ||K||_max the factor is backward stable to
// ~ n eps (||K|| / delta) ||K|| ~ 1e-8 ||K||, so every refinement step gains ~ 8 digits minus log10 cond(K); the loop stops at
// ||r||_inf <= 1e-13 (||K|| ||x|| + ||b||) or gives up after 10 steps (solve_status then reports failure: the reference's
// "-1 / solve failed" answer, never a silently wrong direction).
// ------------------------------------------------------------------------------------------------------------------
// (a matrix assembled into the padded copy and not factored yet — hiopamd_linsolver_assembly_matrix — is brought into M before the
//  object switches to a mode that works on M)
static int linsolver_disarm_direct(hiopamd_linsolver* ls)
{
  if(!ls->direct_armed) return HIOPAMD_OK;
  const int rc = hiopamd_linsolver_sys_matrix_sync(ls);
  ls->direct_armed = false;
  return rc;
}

int hiopamd_linsolver_set_safe_mode(hiopamd_linsolver* ls, int enable, int n_pos_block)
{
  if(!ls || n_pos_block < 0 || n_pos_block > ls->n) return HIOPAMD_ERR_ARG;
  if(enable) {
    const int rd = linsolver_disarm_direct(ls);
    if(rd != HIOPAMD_OK) return rd;
  }
  ls->safe_mode = enable != 0;
  ls->safe_npos = n_pos_block;
  ls->factored = false;
  if(ls->safe_mode && !ls->Msave && ls->n > 0) {
    HIOPAMD_CHECK(hipMalloc((void**)&ls->Msave, sizeof(double) * (size_t)ls->n * ls->n));
    HIOPAMD_CHECK(hipMalloc((void**)&ls->rbuf, sizeof(double) * (size_t)3 * ls->n));   // rhs copy, residual, probe vector
  }
  return HIOPAMD_OK;
}
// Pivoted mode: matrixChanged / solve run the Bunch-Kaufman factorisation of ldlt_bk.hip (the reference's safe solver) instead of the
// no-pivot dataflow LDL^T; takes precedence over the regularise-and-refine safe mode.  Exact inertia for any symmetric matrix, at
// the price of ~ 3 n latency-bound launches per factorisation.
int hiopamd_linsolver_set_pivoting(hiopamd_linsolver* ls, int enable)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(enable) {
    const int rd = linsolver_disarm_direct(ls);
    if(rd != HIOPAMD_OK) return rd;
  }
  ls->factored = false;
  if(enable && !ls->bk) {
    const int rc = hiopamd_ldlt_bk_create(&ls->bk, ls->ctx, ls->n);
    if(rc != HIOPAMD_OK) return rc;
  }
  ls->pivoted = enable != 0;
  return HIOPAMD_OK;
}
int hiopamd_linsolver_safe_mode_info(const hiopamd_linsolver* ls, int* refinements_host, double* residual_rel_host)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(refinements_host) *refinements_host = ls->safe_last_refinements;
  if(residual_rel_host) *residual_rel_host = ls->safe_last_residual;
  return HIOPAMD_OK;
}
// element growth of the last factorisation: max |u_ij| over the strict upper triangle of U (A = U^T D U) and the extreme
// pivots — the quantity whose blow-up tells the caller that the no-pivot factor "misbehaves" (a well-behaved quasi-definite
// KKT factor has |u_ij| = O(||A|| / min |d|)); computed on demand, one pass over the factor
int hiopamd_linsolver_growth(hiopamd_linsolver* ls, double* max_abs_u_host, double* min_abs_d_host, double* max_abs_d_host)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(!ls->factored) return HIOPAMD_ERR_STATE;
  const int n = ls->n;
  const double* M = ls->factor_in_pad ? ls->Mpad : ls->M;
  const int64_t ld = ls->factor_in_pad ? ls->npad : n;
  struct OpU {
    const double* M;
    int n;
    int64_t ld;
    __device__ double identity() const { return 0.0; }
    __device__ double map(int64_t e) const
    {
      const int64_t r = e / n, c = e - r * n;
      return c > r ? fabs(M[r * ld + c]) : 0.0;
    }
    __device__ double combine(double a, double b) const { return fmax(a, b); }
  };
  struct OpDmin {
    const double* M;
    int64_t n;   // (the pitch)
    __device__ double identity() const { return DBL_MAX; }
    __device__ double map(int64_t i) const { return fabs(M[i * n + i]); }
    __device__ double combine(double a, double b) const { return fmin(a, b); }
  };
  struct OpDmax {
    const double* M;
    int64_t n;   // (the pitch)
    __device__ double identity() const { return 0.0; }
    __device__ double map(int64_t i) const { return fabs(M[i * n + i]); }
    __device__ double combine(double a, double b) const { return fmax(a, b); }
  };
  double u = 0.0, dmin = 0.0, dmax = 0.0;
  int rc = hiopamd::launch_reduce<double>(ls->ctx, (int64_t)n * n, OpU{M, n, ld}, &u);
  if(rc == HIOPAMD_OK) rc = hiopamd::launch_reduce<double>(ls->ctx, n, OpDmin{M, ld}, &dmin);
  if(rc == HIOPAMD_OK) rc = hiopamd::launch_reduce<double>(ls->ctx, n, OpDmax{M, ld}, &dmax);
  if(max_abs_u_host) *max_abs_u_host = u;
  if(min_abs_d_host) *min_abs_d_host = dmin;
  if(max_abs_d_host) *max_abs_d_host = dmax;
  return rc;
}

static int safe_refined_solve(hiopamd_linsolver* ls, double* x, bool* ok_out);

int hiopamd_linsolver_matrix_changed(hiopamd_linsolver* ls, int* n_neg_host)
{
  if(!ls || !n_neg_host) return HIOPAMD_ERR_ARG;
  ls->factored = false;
  {
    // a dataflow solve since the last factorisation that timed out delivered garbage: say so BEFORE this matrix is touched
    // (the caller may call again: the matrix is intact)
    int ok = 1;
    int rcf = flow_check(ls, &ok);   // (no extra synchronisation when no dataflow solve ran since the last factorisation)
    if(rcf != HIOPAMD_OK) return rcf;
    if(!ok || ls->flow_failed) {
      ls->flow_failed = false;
      return HIOPAMD_ERR_SOLVE;
    }
  }
  const int n = ls->n;
  if(ls->pivoted) {
    // the reference's safe solver: Bunch-Kaufman (hiopLinSolverSymDenseMagma.cpp:120-250 / hiopLinSolverSymDenseLapack.hpp:75-170):
    // -1 for INFO > 0 or a null pivot, the number of negative eigenvalues otherwise
    SpanScope span(ls->ctx, HIOPAMD_SPAN_LINSOLV_FACT);
    ls->flops_fact += (double)n * n * n / 3.0;
    int info = 0;
    const int rc = hiopamd_ldlt_bk_factor(ls->bk, ls->M, n, ls->inertia, &info);
    if(rc != HIOPAMD_OK) return rc;
    if(info > 0 || ls->inertia[2] > 0) {
      *n_neg_host = -1;
      return HIOPAMD_OK;
    }
    ls->factored = true;
    *n_neg_host = ls->inertia[1];
    return HIOPAMD_OK;
  }
  auto regularise = [&]() -> int {   // safe mode: M <- K + delta diag(+I, -I) (upper triangle), K kept in Msave
    const double delta = ls->safe_delta_rel * ls->safe_anorm;
    int rs = HIOPAMD_OK;
    if(ls->safe_npos > 0) rs = hiopamd_mat_add_sub_diagonal_const(ls->ctx, ls->M, n, 0, ls->safe_npos, delta);
    if(rs == HIOPAMD_OK && ls->safe_npos < n)
      rs = hiopamd_mat_add_sub_diagonal_const(ls->ctx, ls->M, n, ls->safe_npos, n - ls->safe_npos, -delta);
    return rs;
  };
  if(ls->safe_mode && n > 0) {
    // keep the matrix as assembled (both triangles, for the refinement's products) and regularise the copy to be factored
    HIOPAMD_CHECK(hipMemcpyAsync(ls->Msave, ls->M, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ls->ctx->stream));
    int rs = hiopamd_mat_symmetrize(ls->ctx, n, ls->Msave, n);
    if(rs != HIOPAMD_OK) return rs;
    rs = hiopamd_mat_max_abs(ls->ctx, n, n, ls->Msave, n, &ls->safe_anorm);
    if(rs != HIOPAMD_OK) return rs;
    rs = regularise();
    if(rs != HIOPAMD_OK) return rs;
    ls->safe_solve_failed = false;
  }
  SpanScope span(ls->ctx, HIOPAMD_SPAN_LINSOLV_FACT);   // hiopLinSolverSymDenseLapack.hpp:80-125 (tmFactTime; flopsFact = n^3/3)
  ls->flops_fact += (double)n * n * n / 3.0;
  const dim3 tgrid((unsigned)((n + 127) / 128), (unsigned)((n + 127) / 128));
  const bool padded = ls->npad > n;
  // (the caller assembled the padded copy itself: hiopamd_linsolver_assembly_matrix.  Not in safe mode, which works on M and its copy.)
  const bool direct = padded && ls->direct_armed && !ls->safe_mode;
  ls->direct_armed = false;
  ls->factor_in_pad = false;
  auto factor_once = [&]() -> int {
    // (see DfDevice::skip_once)
    const bool df_was = ls->df.enabled;
    const bool df_this = df_was && !ls->df.skip_once;
    ls->df.enabled = df_this;
    ls->df.skip_once = false;
    int r;
    if(padded) {
      // the padded order: diag(M, I), upper triangle (dataflow or stepwise kernels alike: the solves' tables are built for that
      // order); the factor comes back into M, which is untouched if the kernels give up
      const int np = ls->npad;
      const int re = linsolver_ensure_pad(ls);
      if(re != HIOPAMD_OK) return re;
      if(!direct) hipLaunchKernelGGL(ldlt_triu_repitch_kernel, tgrid, dim3(kBlock), 0, ls->ctx->stream, n, ls->M, (int64_t)n, ls->Mpad, (int64_t)np);
      hipLaunchKernelGGL(ldlt_pad_tail_kernel, dim3((unsigned)np), dim3(kBlock), 0, ls->ctx->stream, n, np, ls->Mpad, (int64_t)np);
      r = ldlt_factor_impl(ls->ctx, np, ls->Mpad, np, ls->dinv, ls->V, ls->Dblk, ls->Cd, ls->d_info, ls->inertia, &ls->prof, ls->W, &ls->df, ls->fl_B, ls->Wt);
      if(r == HIOPAMD_OK || r == HIOPAMD_ERR_SINGULAR) {
        if(!direct) hipLaunchKernelGGL(ldlt_triu_repitch_kernel, tgrid, dim3(kBlock), 0, ls->ctx->stream, n, ls->Mpad, (int64_t)np, ls->M, (int64_t)n);
        ls->factor_in_pad = direct;
        ls->inertia[0] -= np - n;   // the unit pivots
      }
    } else {
      r = ldlt_factor_impl(ls->ctx, n, ls->M, n, ls->dinv, ls->V, ls->Dblk, ls->Cd, ls->d_info, ls->inertia, &ls->prof, ls->W, &ls->df, ls->fl_B, ls->Wt);
    }
    ls->df.enabled = df_was;
    if(df_this && r == HIOPAMD_ERR_TIMEOUT) {
      ls->df.skip_once = true;
      if(++ls->df.strikes >= 3) {
        ls->df.enabled = false;
        std::fprintf(stderr, "[hiop_amd] dataflow LDL^T timed out three times in a row: this solver object uses the stepwise kernels from now on\n");
      }
    } else if(df_this && r != HIOPAMD_ERR_HIP) {
      ls->df.strikes = 0;
    }
    return r;
  };
  // the retry copy: only when this call will run the dataflow kernels (the stepwise kernels have no bounded waits) and no other copy
  // of the matrix exists (safe mode keeps Msave)
  const bool df_next = ls->df.enabled && !ls->df.skip_once && ls->df.flags && ls->df.plan.nchain >= 3;
  const bool use_retry = ls->retry_copy && !ls->safe_mode && df_next && n > 0 && !padded;   // (padded: M is not written before the factorisation has succeeded)
  if(use_retry) {
    if(!ls->Mretry) HIOPAMD_CHECK(hipMalloc((void**)&ls->Mretry, sizeof(double) * (size_t)n * n));
    hipLaunchKernelGGL(ldlt_triu_copy_kernel, tgrid, dim3(kBlock), 0, ls->ctx->stream, n, ls->M, ls->Mretry, (int64_t)n);
  }
  int rc = factor_once();
  if(rc == HIOPAMD_ERR_TIMEOUT) {
    // The dataflow kernels gave up — a flag update another workgroup was waiting for did not arrive within the limit (DESIGN.md 3.1;
    // two processes on one device can also starve the chain kernel's roles).  The matrix is overwritten.  With a copy (the retry copy,
    // or safe mode's) the factorisation is redone right away with the stepwise kernels and the caller never sees the incident;
    // without one (hiopamd_linsolver_set_retry_copy(ls, 0): the native KKT objects) the caller re-assembles and calls again, and
    // that call runs the stepwise kernels.
    ls->df_timeouts += 1;
    if(ls->safe_mode && n > 0) {
      HIOPAMD_CHECK(hipMemcpyAsync(ls->M, ls->Msave, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ls->ctx->stream));
      int rs = regularise();
      if(rs != HIOPAMD_OK) return rs;
      rc = factor_once();
    } else if(use_retry) {
      hipLaunchKernelGGL(ldlt_triu_copy_kernel, tgrid, dim3(kBlock), 0, ls->ctx->stream, n, ls->Mretry, ls->M, (int64_t)n);
      rc = factor_once();   // (skip_once is set: the stepwise kernels)
    } else if(padded && ls->retry_copy && !direct) {
      rc = factor_once();   // (M is intact; skip_once is set: the stepwise kernels)
    }   // (direct: the caller's assembly is overwritten, like an unpadded matrix without a retry copy — it re-assembles and calls again)
  }
  if(rc == HIOPAMD_ERR_SINGULAR) {
    // reference: "entry in the factorization's diagonal is exactly zero" -> matrixChanged() returns -1
    *n_neg_host = -1;
    return HIOPAMD_OK;
  }
  if(rc != HIOPAMD_OK) return rc;
  ls->factored = true;
  *n_neg_host = (ls->inertia[2] > 0) ? -1 : ls->inertia[1];
  if(ls->safe_mode && n > 0 && *n_neg_host >= 0) {
    // What was factored is K_delta, and the inertia above is K_delta's.  It is K's as well unless K has an eigenvalue within
    // ~delta of zero (Weyl) — exactly the case in which the refinement against K cannot converge (its contraction factor is
    // ~delta / |lambda_min|).  So the question "is K singular to working precision / is the factor unusable?" — what the
    // reference's Bunch-Kaufman answers with -1 — is put to a PROBE solve: K x = K e refined like every safe-mode solve; no
    // convergence => -1, and the inertia-correction loop of the IPM reacts instead of trusting the regularised matrix's count.
    double* x = ls->rbuf + 2 * (size_t)n;
    int rs = hiopamd_vec_set_to_constant(ls->ctx, n, ls->rbuf, 1.0);
    if(rs == HIOPAMD_OK) rs = hiopamd_mat_times_vec(ls->ctx, n, n, ls->Msave, n, 0.0, x, 1.0, ls->rbuf);
    bool ok = false;
    if(rs == HIOPAMD_OK) rs = safe_refined_solve(ls, x, &ok);
    if(rs != HIOPAMD_OK) return rs;
    if(!ok) {
      ls->factored = false;
      *n_neg_host = -1;
    }
  }
  return HIOPAMD_OK;
}

// the padded copy: allocated on first use; the lower triangle is never assembled, but the diagonal tiles are read and written as whole
// tiles: defined once, like a caller's matrix
static int linsolver_ensure_pad(hiopamd_linsolver* ls)
{
  if(ls->Mpad || ls->npad <= ls->n) return HIOPAMD_OK;
  const int np = ls->npad;
  HIOPAMD_CHECK(hipMalloc((void**)&ls->Mpad, sizeof(double) * (size_t)np * np));
  HIOPAMD_CHECK(hipMemsetAsync(ls->Mpad, 0, sizeof(double) * (size_t)np * np, ls->ctx->stream));
  return HIOPAMD_OK;
}

// x <- (U^T D U)^-1 x with the object's factor, at the order the object works at
static int linsolver_solve_factor(hiopamd_linsolver* ls, double* x, int nrhs)
{
  const int n = ls->n, np = ls->npad;
  if(np == n) return ldlt_solve_impl(ls->ctx, n, ls->M, n, ls->dinv, ls->ybuf, x, nrhs, ls->Cd, ls);
  for(int q = 0; q < nrhs; ++q) {
    double* xq = x + (int64_t)q * n;
    hipLaunchKernelGGL(ldlt_vec_pad_kernel, dim3(grid_for(np)), dim3(kBlock), 0, ls->ctx->stream, n, xq, ls->xpad, np);
    const int rc = ldlt_solve_impl(ls->ctx, np, ls->Mpad, np, ls->dinv, ls->ybuf, ls->xpad, 1, ls->Cd, ls);
    if(rc != HIOPAMD_OK) return rc;
    hipLaunchKernelGGL(ldlt_vec_pad_kernel, dim3(grid_for(n)), dim3(kBlock), 0, ls->ctx->stream, n, ls->xpad, xq, n);
  }
  HIOPAMD_CHECK(hipGetLastError());
  return HIOPAMD_OK;
}

// safe mode: x <- K_delta^-1 x, then refined against the saved K (see hiopamd_linsolver_set_safe_mode); *ok = 0 when the
// refinement did not reach 1e-13 (||K|| ||x|| + ||b||) in 10 steps
static int safe_refined_solve(hiopamd_linsolver* ls, double* x, bool* ok_out)
{
  const int n = ls->n;
  double* b = ls->rbuf;
  double* r = ls->rbuf + n;
  hiopamd::ReduceNow now(ls->ctx);   // (the refinement loop branches on the norms)
  int rc = hiopamd_vec_copy(ls->ctx, n, b, x);
  if(rc != HIOPAMD_OK) return rc;
  double bn = 0.0;
  rc = hiopamd_vec_infnorm(ls->ctx, n, b, &bn);
  if(rc != HIOPAMD_OK) return rc;
  rc = linsolver_solve_factor(ls, x, 1);
  if(rc != HIOPAMD_OK) return rc;
  int it = 0;
  double rel = 0.0;
  bool ok = false;
  for(; it <= 10; ++it) {
    rc = hiopamd_vec_copy(ls->ctx, n, r, b);                                              // r = b - K x
    if(rc == HIOPAMD_OK) rc = hiopamd_mat_times_vec(ls->ctx, n, n, ls->Msave, n, 1.0, r, -1.0, x);
    double rn = 0.0, xn = 0.0;
    if(rc == HIOPAMD_OK) rc = hiopamd_vec_infnorm(ls->ctx, n, r, &rn);
    if(rc == HIOPAMD_OK) rc = hiopamd_vec_infnorm(ls->ctx, n, x, &xn);
    if(rc != HIOPAMD_OK) return rc;
    const double scale = ls->safe_anorm * xn + bn;
    rel = scale > 0.0 ? rn / scale : 0.0;
    if(!(rn == rn) || !(xn == xn)) break;   // NaN: the factor is unusable
    if(rel <= 1e-13) {
      ok = true;
      break;
    }
    if(it == 10) break;
    rc = linsolver_solve_factor(ls, r, 1);                                                // dx = K_delta^-1 r
    if(rc == HIOPAMD_OK) rc = hiopamd_vec_axpy(ls->ctx, n, x, 1.0, r);
    if(rc != HIOPAMD_OK) return rc;
  }
  ls->safe_last_refinements = it;
  ls->safe_last_residual = rel;
  *ok_out = ok;
  return HIOPAMD_OK;
}

int hiopamd_linsolver_solve(hiopamd_linsolver* ls, double* rhs_inout, int nrhs)
{
  if(!ls || !rhs_inout) return HIOPAMD_ERR_ARG;
  if(!ls->factored) return HIOPAMD_ERR_STATE;
  SpanScope span(ls->ctx, HIOPAMD_SPAN_LINSOLV_TRIU_SOLVES);   // :173-195 (tmTriuSolves; flopsTriuSolves = 2 n^2 per rhs)
  ls->flops_triu += 2.0 * (double)ls->n * ls->n * nrhs;
  if(ls->pivoted) return hiopamd_ldlt_bk_solve(ls->bk, ls->M, ls->n, rhs_inout, nrhs);
  if(!ls->safe_mode) return linsolver_solve_factor(ls, rhs_inout, nrhs);
  for(int q = 0; q < nrhs; ++q) {
    bool ok = false;
    const int rc = safe_refined_solve(ls, rhs_inout + (int64_t)q * ls->n, &ok);
    if(rc != HIOPAMD_OK) return rc;
    if(!ok) ls->safe_solve_failed = true;
  }
  return HIOPAMD_OK;
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

int hiopamd_linsolver_set_retry_copy(hiopamd_linsolver* ls, int enable)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  ls->retry_copy = enable != 0;
  if(!ls->retry_copy && ls->Mretry) {
    (void)hipStreamSynchronize(ls->ctx->stream);
    (void)hipFree(ls->Mretry);
    ls->Mretry = nullptr;
  }
  return HIOPAMD_OK;
}
int hiopamd_linsolver_timeouts(const hiopamd_linsolver* ls, int64_t* count_host)
{
  if(!ls || !count_host) return HIOPAMD_ERR_ARG;
  *count_host = (int64_t)ls->df_timeouts;
  return HIOPAMD_OK;
}

int hiopamd_linsolver_set_dataflow(hiopamd_linsolver* ls, int enable)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  ls->df.enabled = enable != 0;
  return HIOPAMD_OK;
}

// the static schedule of the dataflow factorisation for a matrix of order n (host only, no device needed): used by the
// CPU-side schedule check (tests/test_ldlt_dataflow_plan.py).  dims8 = {nsp, nt, nchain, last_has_next, nwide, roles,
// max tasks per role, number of wide tasks}; chain_tasks = 2 x roles x max x 4 ints (type, p, a, b), wide_tasks = 4 ints each.
int hiopamd_ldlt_dataflow_plan(int n, int* dims8_host, int* chain_tasks_host, int* wide_tasks_host, int64_t wide_cap)
{
  if(n < 0 || !dims8_host) return HIOPAMD_ERR_ARG;
  const DfPlan P = df_build_plan(n);
  const int d[8] = {P.nsp, P.nt, P.nchain, P.last_has_next, P.nwide, DF_ROLES, DF_MAXT, (int)P.wtasks.size()};
  for(int i = 0; i < 8; ++i) dims8_host[i] = d[i];
  if(chain_tasks_host)
    for(size_t i = 0; i < P.ctasks.size(); ++i) {
      chain_tasks_host[4 * i] = P.ctasks[i].x; chain_tasks_host[4 * i + 1] = P.ctasks[i].y;
      chain_tasks_host[4 * i + 2] = P.ctasks[i].z; chain_tasks_host[4 * i + 3] = P.ctasks[i].w;
    }
  if(wide_tasks_host) {
    if((int64_t)P.wtasks.size() > wide_cap) return HIOPAMD_ERR_ARG;
    for(size_t i = 0; i < P.wtasks.size(); ++i) {
      wide_tasks_host[4 * i] = P.wtasks[i].x; wide_tasks_host[4 * i + 1] = P.wtasks[i].y;
      wide_tasks_host[4 * i + 2] = P.wtasks[i].z; wide_tasks_host[4 * i + 3] = P.wtasks[i].w;
    }
  }
  return HIOPAMD_OK;
}

int hiopamd_ldlt_dataflow_queues(int n, int* queues_host, int cap_panels)
{
  if(n < 0 || !queues_host) return HIOPAMD_ERR_ARG;
  const DfPlan P = df_build_plan(n);
  if(P.nwide > cap_panels) return HIOPAMD_ERR_ARG;
  for(int j = 0; j < P.nwide; ++j) {
    queues_host[5 * j] = P.wq[j].x; queues_host[5 * j + 1] = P.wq[j].y; queues_host[5 * j + 2] = P.wq[j].z;
    queues_host[5 * j + 3] = P.wq[j].w; queues_host[5 * j + 4] = (int)P.wfirst[j];
  }
  return HIOPAMD_OK;
}

int hiopamd_ldlt_dataflow_nvb(int n) { return df_nvb_for(n); }

// host only: the order a solver object of order n works at (ldlt_padded_order; the environment's HIOPAMD_LDLT_PAD applies)
int hiopamd_ldlt_padded_order(int n) { return n < 0 ? -1 : ldlt_padded_order(n); }

int hiopamd_ldlt_dataflow_far_queues(int n, int* far_host, int cap_panels)
{
  if(n < 0 || !far_host) return HIOPAMD_ERR_ARG;
  const DfPlan P = df_build_plan(n);
  if(P.nwide > cap_panels) return HIOPAMD_ERR_ARG;
  for(int j = 0; j < P.nwide; ++j) {
    far_host[4 * j] = P.wf[j].x; far_host[4 * j + 1] = P.wf[j].y; far_host[4 * j + 2] = P.wf[j].z; far_host[4 * j + 3] = P.wf[j].w;
  }
  return HIOPAMD_OK;
}

int hiopamd_linsolver_flops(const hiopamd_linsolver* ls, double* flops_fact_host, double* flops_triu_solves_host)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(flops_fact_host) *flops_fact_host = ls->flops_fact;
  if(flops_triu_solves_host) *flops_triu_solves_host = ls->flops_triu;
  return HIOPAMD_OK;
}

int hiopamd_linsolver_factored(const hiopamd_linsolver* ls) { return (ls && ls->factored) ? 1 : 0; }

int hiopamd_linsolver_inertia(const hiopamd_linsolver* ls, int* pos, int* neg, int* zero)
{
  if(!ls) return HIOPAMD_ERR_ARG;
  if(pos) *pos = ls->inertia[0];
  if(neg) *neg = ls->inertia[1];
  if(zero) *zero = ls->inertia[2];
  return HIOPAMD_OK;
}

}  // extern "C"
