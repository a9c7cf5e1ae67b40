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
constexpr int W8_SMEM_DOUBLES = 4 * 2 * UD_KT * UD_LD;   // 147,456 B: the update's ring of four operand stages / the substitution's V (256 x 33)
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
// UP tile, eight waves, operands by LDS-DMA.
//
// What bounds the stage loop (profiles/r02_probes, the loop rebuilt ingredient by ingredient on 256 workgroups of one per CU):
// registers only 77.6 TFLOP/s | + operands from LDS 73.5 | + one barrier per stage 68.7 | + 8 ds_write_b128 per stage 62.1 | + the
// global loads 62.1.  Round 4 measured the same ceiling inside the factorisation for the four-wave AND for the eight-wave workgroup
// (59-61 us per tile against 45 us of MFMA time), with one or with two register sets of operand stages in flight: the loop is not
// waiting for memory, and a partner wave on the SIMD only covers the barrier — what costs is the staging itself (global -> VGPR ->
// ds_write_b128: a 16-byte LDS store holds the wave's issue port for ~13 cycles and only half the SIMDs may store at a time).
// So the operands no longer pass through registers: `buffer_load_dwordx4 ... lds` writes a k-row of 128 doubles (one 1 KB
// wave-instruction: lane l -> LDS base + 16 l) straight into a ring of FOUR stage buffers, three stages ahead of their use.
//   * acc = C - V^T U without touching the operands: the NEG bit of v_mfma_f64 (blgp = 1 negates A);
//   * the DMA is issued from inline assembly: through the compiler's builtin every LDS read behind it gets an s_waitcnt vmcnt(0)
//     (the compiler cannot see that the reads go to another buffer of the ring), which serialises loads and MFMAs.  The wave waits
//     for ITS OWN loads of a stage with an explicit `s_waitcnt vmcnt(4)` (the four loads of the next stage may stay in flight; any
//     other memory operation issued since only makes that wait stricter), then the stage barrier publishes the buffer;
//   * the ring position carries over from task to task ((s + phase) mod 4 for stage s): the three buffers the next task's first
//     stages go to are never the one the last stage of the current task may still be read from, so the next task's first loads
//     can be issued behind the current tile's epilogue stores without another barrier (DfArgs::pipe bit 1).
// element (i, q, reg) of the accumulators  <->  row  wr*64 + 32*(i>>1) + 2*(lk + 4*reg) + (i&1),  col  wc*32 + 2*li + q.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int W8_RING = 4;
constexpr int W8_STAGE_DOUBLES = 2 * UD_KT * UD_LD;   // V rows, then U rows: 36,864 B

struct W8Regs {
  double4_t acc[4][2];
  int phase = 0;   // ring buffer of stage 0 of the task whose prologue is in flight / which runs
};

__device__ __forceinline__ df_u32x4 w8_rsrc(const double* base)
{
  const unsigned long long b = (unsigned long long)base;
  df_u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu;
  r.z = 0xffffffffu;
  r.w = 0x00020000u;
  return r;
}
// one wave-instruction: 64 lanes x 16 bytes from (rs.base + voff[lane] + soff) to LDS byte address lds_addr + 16 lane
__device__ __forceinline__ void w8_dma16(const df_u32x4& rs, unsigned lds_addr, unsigned voff, unsigned soff)
{
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

template <bool FULL>
struct W8Addr {
  df_u32x4 rsV, rsU, rsV2, rsU2;
  __amdgpu_buffer_rsrc_t rsC;
  unsigned lda8, ldv8, vvoff, uvoff, cvoff_full, lds0;
  int rlim, clim, wave, wr, wc, lk, li, panels;
  // panels = 2: stages 16 .. 31 read the row panel and the factor rows of super-panel j + 1 (DF_UP2)
  __device__ __forceinline__ W8Addr(const DfArgs& a, int j, int I, int J, const double* smem, int tid, int panels_)
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
    rsV = w8_rsrc(a.V + (int64_t)(j % a.nvb) * LD_NB * a.ldv + r0);
    rsU = w8_rsrc(a.A + (int64_t)(LD_NB * j) * a.lda + c0);
    rsV2 = w8_rsrc(a.V + (int64_t)((j + 1) % a.nvb) * LD_NB * a.ldv + r0);
    rsU2 = w8_rsrc(a.A + (int64_t)(LD_NB * (j + 1)) * a.lda + c0);
    rsC = df_rsrc(a.A + (int64_t)r0 * a.lda + c0);
    rlim = a.N - r0;
    clim = a.N - c0;
    const int col2 = 2 * lane;
    vvoff = 8u * (unsigned)(FULL ? col2 : (col2 < rlim - 2 ? col2 : rlim - 2));
    uvoff = 8u * (unsigned)(FULL ? col2 : (col2 < clim - 2 ? col2 : clim - 2));
    cvoff_full = 8u * (unsigned)(2 * li) + (unsigned)(2 * lk) * lda8;
    lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)smem);   // (the low half of a flat LDS address is the LDS offset)
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
  // stage st into ring buffer `buf`: this wave moves k-rows wave and 8 + wave of V and of U (four wave-instructions)
  __device__ __forceinline__ void dma_stage(int st, int buf_) const
  {
    const int buf = __builtin_amdgcn_readfirstlane(buf_);   // (wave-uniform by construction; M0 needs a scalar)
    const bool second = st >= LD_NB / UD_KT;   // (only a two-panel task has such stages)
    const int sl = second ? st - LD_NB / UD_KT : st;
    const unsigned vb = lds0 + 8u * (unsigned)(buf * W8_STAGE_DOUBLES), ub = vb + 8u * (unsigned)(UD_KT * UD_LD);
#pragma unroll
    for(int p = 0; p < 2; ++p) {
      const unsigned kr = (unsigned)(8 * p + wave), k = (unsigned)(sl * UD_KT) + kr;
      if(second) {
        w8_dma16(rsV2, vb + kr * (unsigned)(8 * UD_LD), vvoff, k * ldv8);
        w8_dma16(rsU2, ub + kr * (unsigned)(8 * UD_LD), uvoff, k * lda8);
      } else {
        w8_dma16(rsV, vb + kr * (unsigned)(8 * UD_LD), vvoff, k * ldv8);
        w8_dma16(rsU, ub + kr * (unsigned)(8 * UD_LD), uvoff, k * lda8);
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
__device__ __forceinline__ int w8_ring(int b) { return b >= W8_RING ? b - W8_RING : b; }

// the first two operand stages (by LDS-DMA, into ring buffers phase and phase + 1) and the C tile of task (j, I, J) in flight; no wait,
// no barrier.  The caller guarantees that nobody reads those two buffers any more (see the note on the ring above).
template <bool FULL>
__device__ __forceinline__ void w8_tile_prologue(const DfArgs& a, int j, int I, int J, const double* smem, int tid, int panels, W8Regs& T)
{
  const W8Addr<FULL> ad(a, j, I, J, smem, tid, panels);
  const int p0 = __builtin_amdgcn_readfirstlane(T.phase);
  ad.dma_stage(0, p0);
  ad.dma_stage(1, w8_ring(p0 + 1));
  ad.dma_stage(2, w8_ring(p0 + 2));
  ad.cload(T);
}

// the tile proper; expects w8_tile_prologue(same task) to have been issued.  Leaves T.phase at the ring position of the NEXT task.
// Returns false when a gate saw the abort.
template <bool FULL, bool PROF, class Gate = DfNoGate, class Hook = DfNoHook>
__device__ __forceinline__ bool w8_tile_run(const DfArgs& a, int j, int I, int J, double* smem, int tid, int panels, unsigned (&ph)[12],
                                            W8Regs& T, Gate gate = Gate(), Hook hook = Hook())
{
  bool gate_ok = true;
  const int dbg = PROF ? a.dbg : 0;
  const unsigned tp0 = dbg ? (unsigned)wall_clock64() : 0u;
  const W8Addr<FULL> ad(a, j, I, J, smem, tid, panels);
  auto mark = [&](unsigned stg) {
    if(tid == 0) df_st(a.flags + a.off_wg + 2 * (int64_t)blockIdx.x + 1, 0x80000000u | (stg << 16) | ((unsigned)J & 0xffffu));
  };
  mark(100u);
  const int nst = panels * (LD_NB / UD_KT);
  const int arow = ad.wr * 64 + 2 * ad.li, bcol = ad.wc * 32 + 2 * ad.li;
  const int lk = ad.lk;
  int cur = __builtin_amdgcn_readfirstlane(T.phase);   // ring buffer of stage st
  // HIOPAMD_DF_EXP (timing experiments of the PROF instantiation only; the factor is garbage): 1 = no operand loads in the loop,
  // 2 = no stage barrier, 4 = no LDS operand reads in the loop, 8 = all eight waves issue their loads at the same point of the stage
  const int ex = PROF ? a.exp : 0;
  const unsigned tp1 = dbg ? (unsigned)wall_clock64() : 0u;
  const unsigned long long tc1 = dbg ? (unsigned long long)clock64() : 0ull;
  // Before stage 0: this wave's loads of stage 0 have landed (the prologue's later ones — stages 1, 2 and most of the C tile — may
  // still be in flight: `vmcnt(8)` leaves at most eight operations outstanding, and those are younger), then everybody's have.
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  df_double2 av[2][2], bv[2];
  {
    const double* vb = smem + cur * W8_STAGE_DOUBLES;
    const double* ub = vb + UD_KT * UD_LD;
#pragma unroll
    for(int h = 0; h < 2; ++h) av[0][h] = *reinterpret_cast<const df_double2*>(vb + lk * UD_LD + arow + 32 * h);
    bv[0] = *reinterpret_cast<const df_double2*>(ub + lk * UD_LD + bcol);
  }
  // One barrier per stage, in its MIDDLE: the operands of the next k-step are already on their way out of LDS when a wave arrives at
  // it, and those of the first k-step of the NEXT stage are requested in this stage's last k-step — the matrix pipe does not run dry at
  // the stage boundary (with the barrier at the boundary every wave of the CU stopped there together, then waited for the LDS round
  // trip of its first operands: ~20 % of the loop, the same for four or eight waves, one or two stages of lead, register or DMA staging).
  // The barrier in stage st says: everybody's loads of stage st + 1 have landed, and everybody has finished stage st - 1 — whose
  // buffer the loads of stage st + 3 may now overwrite.
  for(int st = 0; st < nst; ++st) {
    const double* vb = smem + cur * W8_STAGE_DOUBLES;
    const double* ub = vb + UD_KT * UD_LD;
    const int nb = w8_ring(cur + 1);
    const double* vbn = smem + nb * W8_STAGE_DOUBLES;
    const double* ubn = vbn + UD_KT * UD_LD;
    mark((unsigned)st);
#pragma unroll
    for(int kk = 0; kk < UD_KT / 4; ++kk) {
      const int pb = kk & 1;
      if(ex & 4) {
        // (timing experiment: the operands stay what they are)
      } else if(kk + 1 < UD_KT / 4) {
#pragma unroll
        for(int h = 0; h < 2; ++h) av[pb ^ 1][h] = *reinterpret_cast<const df_double2*>(vb + (4 * (kk + 1) + lk) * UD_LD + arow + 32 * h);
        bv[pb ^ 1] = *reinterpret_cast<const df_double2*>(ub + (4 * (kk + 1) + lk) * UD_LD + bcol);
      } else if(st + 1 < nst) {   // first k-step of the next stage (its buffer is complete since this stage's barrier)
#pragma unroll
        for(int h = 0; h < 2; ++h) av[pb ^ 1][h] = *reinterpret_cast<const df_double2*>(vbn + lk * UD_LD + arow + 32 * h);
        bv[pb ^ 1] = *reinterpret_cast<const df_double2*>(ubn + lk * UD_LD + bcol);
      }
      if(kk == 1) {
        // (the hook BEFORE the barrier: what lane 0 publishes in the last stage — the task selected ahead — is read by every wave
        //  after the loop, and this is the loop's last barrier)
        if constexpr(Hook::active) hook(st, nst);
        if(!(ex & 2)) {
          if(st + 2 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // my loads of stage st + 1 (those of st + 2 may fly)
          else if(st + 1 < nst) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
        if(st + 3 < nst) {
          if constexpr(Gate::active) {
            if(((st + 3) & 3) == 0) gate_ok = gate((st + 3) >> 2) && gate_ok;   // (aborted: the result is discarded anyway)
          }
          // Issuing the four LDS-DMA instructions holds a wave's issue port for several hundred cycles (measured: the loop is 8 %
          // faster without them).  The two waves of a SIMD therefore issue theirs at different times — waves 0-3 here, waves 4-7
          // two k-steps later — so that one of them always has MFMAs to issue.
          if(!(ex & 1) && (ad.wave < 4 || (ex & 8))) ad.dma_stage(st + 3, w8_ring(cur + 3));
        }
      }
      if(kk == 3 && st + 3 < nst && !(ex & 1) && ad.wave >= 4 && !(ex & 8)) ad.dma_stage(st + 3, w8_ring(cur + 3));
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int q = 0; q < 2; ++q)
          T.acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb][i >> 1][i & 1], bv[pb][q], T.acc[i][q], 0, 0, 1);   // blgp = 1: -A
    }
    cur = nb;
  }
  T.phase = cur;   // the buffer after the last stage's: the next task's stages 0, 1, 2 go to this one and the two behind it — never the
                   // last stage's own buffer, which slower waves may still be reading
  // ---- epilogue: stores only
  mark(101u);
  if(dbg && (dbg == 1 || j + (panels - 1) == dbg - 2)) {   // (a fused task is accounted under the queue it was taken from)
    const unsigned tp2 = (unsigned)wall_clock64();
    ph[9] += tp1 - tp0;    // set-up
    ph[10] += tp2 - tp1;   // the stages
    ph[11] += (unsigned)(((unsigned long long)clock64() - tc1) >> 8);   // ... in shader clocks / 256: the clock the loop ran at
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
  constexpr int WB_WAVES = 8;
#include "ldlt_wide8_body.inc"
}
