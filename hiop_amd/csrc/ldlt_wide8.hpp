// Wide kernel of the dataflow LDL^T, second form ("W8"): ONE workgroup of EIGHT waves per CU.
// (included by ldlt_dataflow.hpp inside namespace hiopamd, after the four-wave wide kernel whose task protocol it shares)
//
// Why a second form.  The four-wave wide kernel needs two workgroups per CU to keep the fp64 matrix pipe busy (one wave per SIMD
// stalls the pipe at every barrier and LDS round trip: 72 us per K = 512 tile alone on a CU against 55.4 us of pure MFMA time),
// and with two 73.7 KB workgroups sharing a CU a workgroup froze in mid-task once per ~1.6e4 factorisations;
// with one four-wave workgroup per CU nothing froze in 41 600 factorisations, at 4 % more time.  Here the two waves that share a
// SIMD belong to the SAME workgroup:
//   * UP tile   128 x 128 as 2 x 4 waves of 64 x 32 (4 x 2 tiles of v_mfma_f64_16x16x4_f64, 64 accumulator registers per lane);
//               the partner wave of a SIMD fills the pipe while a wave sits at the stage barrier or waits for LDS;
//   * TR task   32 columns of the row panel on eight waves — one 16-column group per wave instead of two: the dependent MFMA chain
//               of a block row is half as long (a 64-column variant, two groups per wave, measured 53 us per task against 31 of the
//               four-wave form: two waves per SIMD make the task MFMA-bound, and the updates of a tile row wait for it);
//   * nobody computes on a CU while somebody else polls there: a workgroup is either in a task or looking for one.
// No correctness or liveness property depends on how many workgroups are resident or where: tasks are taken by ticket in a
// topological order, a task only waits for tasks already taken or for the chain kernel (tests/test_ldlt_dataflow_plan.py).
//
// Task pipelining (DfArgs::pipe).  With one workgroup per CU nothing hides a task's fixed costs (selection: two to three
// dependent flag / ticket round trips; the wait for its inputs; C tile and first operand stage in flight; drain of the stores
// before the publication) — ~13 us against 55 us of MFMA time per K = 512 tile.  So, in the update-bound part of the
// factorisation (queues j < DfArgs::jpipe):
//   bit 0  lane 0 selects the NEXT task during the last five stages of the tile loop, one step per stage, every step consuming
//          the loads the previous one issued (flags of the queue heads -> ticket -> task descriptor -> flags of the task's
//          inputs -> verdict): the loop never waits for a round trip;
//   bit 1  when that task is an ordinary update tile whose inputs are there, its C tile and first operand stage are loaded
//          right behind the epilogue stores of the current tile, BEFORE the drain: the two latencies overlap.
// A task taken ahead is safe: the workgroup that holds it is past every wait of its current task (gated head tiles, which wait
// inside their loop, never select ahead) and starts it next, so "a task only waits for running tasks" still holds.

constexpr int W8_THREADS = 512;
constexpr int W8_TRW = DF_TRW;                // columns per substitution task (as in the four-wave form: the task lists are the same)
constexpr int W8_SMEM_DOUBLES = 4 * UD_KT * UD_LD;   // 73,728 B: the update's two double-buffered operand stages / the substitution's V (256 x 33)
static_assert(W8_SMEM_DOUBLES >= LD_NB * (DF_TRW + 1), "the substitution's V fits into the stage buffers of the update tile");

// TR(j, c32): DF_TRW = 32 columns starting at c32 of the tail of row panel j, EIGHT waves: wave (I, gq) owns the 16-row sub-block I of
// every 64-row block row for the 16-column group gq — half the dependent MFMA chain per wave of df_task_trsm (which gives both
// groups to one wave): the task is a chain of latencies and the updates of a tile row wait for it.
__device__ __forceinline__ bool df_task_trsm8(const DfArgs& a, int j, int c32, double* smem, int tid, bool early, int* sh_ok,
                                              long long t_start, unsigned* rowflags, unsigned row_inc)
{
  double(*Vs)[DF_TRW + 1] = reinterpret_cast<double(*)[DF_TRW + 1]>(smem);   // 256 x 33
  const int lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int I = wv & 3, gq = wv >> 2;
  const int K0 = LD_NB * j;
  const int vc = 16 * gq + li;
  const int64_t col = (int64_t)c32 + vc;
  const bool col_ok = col < a.N;
  const int64_t colc = col_ok ? col : (int64_t)(a.N - 1);
  double* Vb = a.V + (int64_t)(j % a.nvb) * LD_NB * a.ldv;
  const double* Cd = a.Cd + (int64_t)j * (LD_NB * LD_NB);
  const double* Dk_sp = a.Dblk + (int64_t)(K0 / LD_nb) * (LD_nb * LD_nb);
  const double* Li_sp = a.Li + (int64_t)(K0 / LD_nb) * (4 * LD_SB * LD_SB);
  double4_t t[4];
#pragma unroll
  for(int P = 0; P < 4; ++P)
#pragma unroll
    for(int r = 0; r < 4; ++r) {
      const double v = ldg_sc1(a.A + (int64_t)(K0 + 64 * P + 16 * I + g + 4 * r) * a.lda + colc);
      t[P][r] = col_ok ? v : 0.0;
    }
  __syncthreads();   // the LDS buffer may still be read by the previous task's waves
#pragma unroll
  for(int P = 0; P < 4; ++P) {
    if(early) {
      unsigned* cf = a.flags + a.off_chain + (int64_t)j * DF_CH;
      DfWait w(a.flags + DF_ABORT);
      w.set<0>(cf + DF_CV + P * 4 + P, 4u + P + 1u);                        // F(P)
      if(P >= 1) w.set<1>(cf + DF_CV + 0 * 4 + P, 4u + 0 + 1u);             // T(0, P)
      if(P >= 2) w.set<2>(cf + DF_CV + 1 * 4 + P, 4u + 1 + 1u);             // T(1, P)
      if(P >= 3) w.set<3>(cf + DF_CV + 2 * 4 + P, 4u + 2 + 1u);             // T(2, P)
      if(!df_wait(a.flags, w, sh_ok, t_start, 4, j, c32, P, 0)) return false;
    }
    const double* Dk = Dk_sp + P * (LD_nb * LD_nb);
    const double* Li = Li_sp + P * (4 * LD_SB * LD_SB);
    double nl[3][4], iv[4], dsc[4];
#pragma unroll
    for(int r = 0; r < 4; ++r) dsc[r] = ld_batch(a.dinv + K0 + 64 * P + 16 * I + g + 4 * r);
#pragma unroll
    for(int J = 0; J < 3; ++J)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) nl[J][kk] = -ld_batch(Dk + (16 * J + 4 * kk + g) * LD_nb + 16 * I + li);
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) iv[kk] = ld_batch(Li + I * 256 + li * 16 + 4 * kk + g);
    // two accumulators per wave (even / odd k-steps of the products with the earlier block rows): the chain of dependent MFMAs
    // of a block row is half as long
    double4_t u = t[P], u2 = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for(int q = 0; q < P; ++q) {
      double Lop[4][4];
#pragma unroll
      for(int Jq = 0; Jq < 4; ++Jq)
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) Lop[Jq][kk] = -ld_batch(Cd + (64 * q + 16 * Jq + 4 * kk + g) * LD_NB + (64 * P + 16 * I + li));
#pragma unroll
      for(int Jq = 0; Jq < 4; ++Jq)
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) {
          if(kk & 1) u2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lop[Jq][kk], Vs[64 * q + 16 * Jq + 4 * kk + g][vc], u2, 0, 0, 0);
          else u = __builtin_amdgcn_mfma_f64_16x16x4f64(Lop[Jq][kk], Vs[64 * q + 16 * Jq + 4 * kk + g][vc], u, 0, 0, 0);
        }
    }
    if(P > 0) {
#pragma unroll
      for(int r = 0; r < 4; ++r) u[r] += u2[r];
    }
#pragma unroll
    for(int J = 0; J < 4; ++J) {
      if(I == J) {   // wave-uniform
        double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[kk], u[kk], v, 0, 0, 0);
#pragma unroll
        for(int r = 0; r < 4; ++r) {
          const int row = 64 * P + 16 * I + g + 4 * r;
          Vs[row][vc] = v[r];
          if(col_ok) {
            stg_sc1(Vb + (int64_t)row * a.ldv + col, v[r]);
            stg_sc1(a.A + (int64_t)(K0 + row) * a.lda + col, v[r] * dsc[r]);
          }
        }
      }
      __syncthreads();
      if(I > J && J < 3) {
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) u = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[J][kk], Vs[64 * P + 16 * J + 4 * kk + g][vc], u, 0, 0, 0);
      }
    }
    if(rowflags) {   // uniform
      df_drain();
      if(tid == 0) df_add(rowflags + P, row_inc);
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// UP tile, eight waves.  Registers of a tile that live across the task boundary when the next tile's prologue is issued
// behind the current tile's epilogue (DfArgs::pipe bit 1): the C tile in the accumulators and the first operand stage.
// element (i, q, reg) of the accumulators  <->  row  wr*64 + 32*(i>>1) + 2*(lk + 4*reg) + (i&1),  col  wc*32 + 2*li + q.
// ---------------------------------------------------------------------------------------------------------------------
// Operand stages travel memory -> registers -> LDS; TWO register sets, so that the loads of two stages are in flight at any time
// (~64 KB per CU): with one set (one stage = 1.7 us of lead) the eight-wave tile ran at the pace of the four-wave one, 59 us per
// tile instead of the pipe's 55 — the loop was waiting for its operands, not for the matrix pipe (profiles/r04_probes).
struct W8Regs {
  double4_t acc[4][2];
  df_double2 vreg[2][2], ureg[2][2];   // [stage & 1][pass]
};

template <bool FULL>
struct W8Addr {
  __amdgpu_buffer_rsrc_t rsV, rsU, rsV2, rsU2, rsC;
  unsigned lda8, ldv8, vvoff, uvoff, cvoff_full;
  int rlim, clim, wave, wr, wc, lk, li, panels;
  // panels = 2: stages 16 .. 31 read the row panel and the factor rows of super-panel j + 1 (DF_UP2)
  __device__ __forceinline__ W8Addr(const DfArgs& a, int j, int I, int J, int tid, int panels_)
  {
    panels = panels_;
    const int r0 = UD_T * I, c0 = UD_T * J;
    const int lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wr = wave >> 2;
    wc = wave & 3;
    lk = lane >> 4;
    li = lane & 15;
    lda8 = (unsigned)a.lda * 8u;
    ldv8 = (unsigned)a.ldv * 8u;
    rsV = df_rsrc(a.V + (int64_t)(j % a.nvb) * LD_NB * a.ldv + r0);
    rsU = df_rsrc(a.A + (int64_t)(LD_NB * j) * a.lda + c0);
    rsV2 = df_rsrc(a.V + (int64_t)((j + 1) % a.nvb) * LD_NB * a.ldv + r0);
    rsU2 = df_rsrc(a.A + (int64_t)(LD_NB * (j + 1)) * a.lda + c0);
    rsC = df_rsrc(a.A + (int64_t)r0 * a.lda + c0);
    rlim = a.N - r0;
    clim = a.N - c0;
    const int col2 = 2 * lane;
    vvoff = 8u * (unsigned)(FULL ? col2 : (col2 < rlim - 2 ? col2 : rlim - 2));
    uvoff = 8u * (unsigned)(FULL ? col2 : (col2 < clim - 2 ? col2 : clim - 2));
    cvoff_full = 8u * (unsigned)(2 * li) + (unsigned)(2 * lk) * lda8;
  }
  __device__ __forceinline__ int crow(int i, int reg) const { return wr * 64 + 32 * (i >> 1) + 2 * (lk + 4 * reg) + (i & 1); }
  __device__ __forceinline__ int ccol() const { return wc * 32 + 2 * li; }
  __device__ __forceinline__ void c_off(int i, int reg, unsigned& voff, unsigned& soff) const
  {
    if constexpr(FULL) {
      voff = cvoff_full;
      soff = (unsigned)(wr * 64 + 32 * (i >> 1) + 8 * reg + (i & 1)) * lda8 + 8u * (unsigned)(wc * 32);
    } else {
      const int R = crow(i, reg), Cc = ccol();
      voff = (unsigned)(R < rlim ? R : rlim - 1) * lda8 + 8u * (unsigned)(Cc < clim - 2 ? Cc : clim - 2);
      soff = 0u;
    }
  }
  // stage st: pass p moves k-row 8 p + wave of the stage, two adjacent columns per lane; into register set SET (= st & 1)
  template <int SET>
  __device__ __forceinline__ void gload(W8Regs& T, int st) const
  {
    const bool second = st >= LD_NB / UD_KT;   // (only a two-panel task has such stages)
    const int sl = second ? st - LD_NB / UD_KT : st;
#pragma unroll
    for(int p = 0; p < 2; ++p) {
      const unsigned k = (unsigned)(sl * UD_KT + 8 * p + wave);
      if(second) {
        T.vreg[SET][p] = df_bload2<HIOPAMD_DF_OPAUX>(rsV2, vvoff, k * ldv8);
        T.ureg[SET][p] = df_bload2<HIOPAMD_DF_OPAUX>(rsU2, uvoff, k * lda8);
      } else {
        T.vreg[SET][p] = df_bload2<HIOPAMD_DF_OPAUX>(rsV, vvoff, k * ldv8);
        T.ureg[SET][p] = df_bload2<HIOPAMD_DF_OPAUX>(rsU, uvoff, k * lda8);
      }
    }
  }
  __device__ __forceinline__ void cload(W8Regs& T) const
  {
#pragma unroll
    for(int i = 0; i < 4; ++i)
#pragma unroll
      for(int reg = 0; reg < 4; ++reg) {
        unsigned vo, so;
        c_off(i, reg, vo, so);
        const df_double2 c = df_bload2(rsC, vo, so);
        T.acc[i][0][reg] = c.x;
        T.acc[i][1][reg] = c.y;
      }
  }
};

// first operand stage and the C tile of task (j, I, J) in flight (no wait, no barrier)
template <bool FULL>
__device__ __forceinline__ void w8_tile_prologue(const DfArgs& a, int j, int I, int J, int tid, int panels, W8Regs& T)
{
  const W8Addr<FULL> ad(a, j, I, J, tid, panels);
  ad.template gload<0>(T, 0);
  ad.cload(T);
}

struct DfNoHook {
  static constexpr bool active = false;
  __device__ __forceinline__ void operator()(int, int) const {}
};
template <class H>
struct DfStageHook {
  static constexpr bool active = true;
  H& h;
  __device__ __forceinline__ void operator()(int st, int nst) const { h(st, nst); }
};

// the tile proper; expects w8_tile_prologue(same task) to have been issued into T.  Returns false when a gate saw the abort.
template <bool FULL, bool PROF, class Gate = DfNoGate, class Hook = DfNoHook>
__device__ __forceinline__ bool w8_tile_run(const DfArgs& a, int j, int I, int J, double* smem, int tid, int panels, unsigned (&ph)[12],
                                            W8Regs& T, Gate gate = Gate(), Hook hook = Hook())
{
  bool gate_ok = true;
  const int dbg = PROF ? a.dbg : 0;
  const unsigned tp0 = dbg ? (unsigned)wall_clock64() : 0u;
  double(*Vs)[UD_KT][UD_LD] = reinterpret_cast<double(*)[UD_KT][UD_LD]>(smem);
  double(*Us)[UD_KT][UD_LD] = reinterpret_cast<double(*)[UD_KT][UD_LD]>(smem + 2 * UD_KT * UD_LD);
  const W8Addr<FULL> ad(a, j, I, J, tid, panels);
  const int lane = tid & 63;
  const int col2 = 2 * lane;
  auto lstore = [&](int buf, auto set) {
    constexpr int SET = decltype(set)::value;
#pragma unroll
    for(int p = 0; p < 2; ++p) {
      *reinterpret_cast<df_double2*>(&Vs[buf][8 * p + ad.wave][col2]) = T.vreg[SET][p];
      *reinterpret_cast<df_double2*>(&Us[buf][8 * p + ad.wave][col2]) = -T.ureg[SET][p];
    }
  };
  auto mark = [&](unsigned stg) {
    if(tid == 0) df_st(a.flags + a.off_wg + 2 * (int64_t)blockIdx.x + 1, 0x80000000u | (stg << 16) | ((unsigned)J & 0xffffu));
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  mark(100u);
  const int nst = panels * (LD_NB / UD_KT);   // 16 or 32: even
  __syncthreads();   // the LDS buffers may still be read by the previous task's waves
  ad.template gload<1>(T, 1);
  lstore(0, Set0());                          // (waits for the prologue's stage 0)
  ad.template gload<0>(T, 2);
  __syncthreads();
  const int arow = ad.wr * 64 + 2 * ad.li, bcol = ad.wc * 32 + 2 * ad.li;
  const int lk = ad.lk;
  const unsigned tp1 = dbg ? (unsigned)wall_clock64() : 0u;
  // one stage: MFMAs on LDS buffer `cur`; at its second k-step the next stage moves from its register set into the other LDS buffer and
  // the set is refilled with the stage three ahead (a gated tile looks at its block-row flags before the first loads of a block row)
  auto stage = [&](int st, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    using Nxt = std::integral_constant<int, cur ^ 1>;
    mark((unsigned)st);
    df_double2 av[2][2], bv[2];
#pragma unroll
    for(int h = 0; h < 2; ++h) av[0][h] = *reinterpret_cast<const df_double2*>(&Vs[cur][lk][arow + 32 * h]);
    bv[0] = *reinterpret_cast<const df_double2*>(&Us[cur][lk][bcol]);
#pragma unroll
    for(int kk = 0; kk < UD_KT / 4; ++kk) {
      const int pb = kk & 1;
      if(kk + 1 < UD_KT / 4) {
#pragma unroll
        for(int h = 0; h < 2; ++h) av[pb ^ 1][h] = *reinterpret_cast<const df_double2*>(&Vs[cur][4 * (kk + 1) + lk][arow + 32 * h]);
        bv[pb ^ 1] = *reinterpret_cast<const df_double2*>(&Us[cur][4 * (kk + 1) + lk][bcol]);
      }
      if(kk == 1 && st + 1 < nst) {
        lstore(cur ^ 1, Nxt());               // stage st + 1 (register set (st + 1) & 1 = cur ^ 1)
        if(st + 3 < nst) {
          if constexpr(Gate::active) {
            if(((st + 3) & 3) == 0) gate_ok = gate((st + 3) >> 2) && gate_ok;   // (aborted: the result is discarded anyway)
          }
          ad.template gload<cur ^ 1>(T, st + 3);
        }
      }
      if constexpr(Hook::active) {
        if(kk == 2) hook(st, nst);
      }
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int q = 0; q < 2; ++q) T.acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb][i >> 1][i & 1], bv[pb][q], T.acc[i][q], 0, 0, 0);
    }
    __syncthreads();
  };
  for(int st = 0; st < nst; st += 2) {
    stage(st, Set0());
    stage(st + 1, Set1());
  }
  // ---- epilogue: stores only
  mark(101u);
  if(dbg && (dbg == 1 || j + (panels - 1) == dbg - 2)) {   // (a fused task is accounted under the queue it was taken from)
    const unsigned tp2 = (unsigned)wall_clock64();
    ph[9] += tp1 - tp0;    // prologue (LDS hand-over of the first stage, two barriers)
    ph[10] += tp2 - tp1;   // the stages
  }
  const bool diag = (I == J);
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      unsigned vo, so;
      ad.c_off(i, reg, vo, so);
      const df_double2 v = df_double2{T.acc[i][0][reg], T.acc[i][1][reg]};
      const int R = ad.crow(i, reg), Cc = ad.ccol();
      const bool inside = FULL || (R < ad.rlim && Cc < ad.clim);
      if(!diag) {
        if(inside) df_bstore2(ad.rsC, vo, so, v);
      } else if(inside) {
        if(Cc >= R) df_bstore2(ad.rsC, vo, so, v);
        else if(Cc + 1 == R)   // the pair straddles the diagonal: only its second element is in the upper triangle
        {
          const double second = T.acc[i][1][reg];
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(df_u32x2, second), ad.rsC, (int)(vo + 8u), (int)so, DF_SC1);
        }
      }
    }
  return gate_ok;
}

struct DfWide8Shared {
  double smem[W8_SMEM_DOUBLES] __attribute__((aligned(16)));
  int4 task, ntask;
  int kind, ok, nkind, nready, nidx;
};

template <bool PROF>
__global__ __launch_bounds__(W8_THREADS, 2) void ldlt_wide8_kernel(const DfArgs a)
{
  __shared__ __attribute__((aligned(16))) double smem[W8_SMEM_DOUBLES];
  __shared__ int sh_kind, sh_ok, sh_nkind, sh_nready, sh_nidx;
  __shared__ int4 sh_task, sh_ntask;
#include "ldlt_wide8_body.inc"
}
