// Dataflow LDL^T: the whole factorisation as TWO persistent kernels that talk through device flags — no kernel
// boundaries, no stream events and no host round trips between the 32 super-panels of an N = 8192 KKT matrix.
// (included by ldlt.hip inside namespace hiopamd, after the tile kernels it re-uses)
//
//   chain kernel  (16 workgroups on the reserved CUs, CU-masked stream `diag_stream`):
//       the serial spine of the factorisation on 64 x 64 tiles of a sliding window: for super-panel j the window is
//       C_j (the 256 x 256 diagonal block, compact copy), H_j = A[R_j, R_j+1] (the head of the row panel) and
//       Nn = C_j+1 (the next diagonal block).  Tasks: F(p) factor tile (p,p); T(p,c) tile solve V = L_pp^-1 X, U = D^-1 V;
//       U(p;a,b) tile(a,b) -= V(p,a)^T U(p,b).  Workgroup 0 (the spine) runs F(p) -> T(p,p+1) -> U(p;p+1,p+1) -> F(p+1) ...,
//       workgroup 1 the three tasks the spine needs next, the others the rest of the window (static ownership,
//       chain_task_table()).  Round 1's chain (one workgroup factoring the whole 256 block, then a head substitution launch,
//       then a diagonal-update launch: 203 us per super-panel) was the critical path of 20 of the 32 super-panels.
//   wide kernel   (persistent workgroups on the other 240 CUs, CU-masked stream `upd_stream`), a ticket-ordered task list:
//       TR(j, g)    16 columns of the tail of row panel j: V = L_jj^-1 A[R_j, cols], U = D^-1 V   (4 waves, the head-
//                   substitution algorithm of ldlt_headtrsm_kernel)
//       UP(j, I, J) 128 x 128 tile (I, J) of the trailing matrix -= V_j[:, I]^T U_j[:, J]           (the double-buffered
//                   MFMA tile of ldlt_update_db_kernel)
//       Tickets are handed out in a topological order (every dependency of a task has a smaller ticket or belongs to the
//       chain kernel), so a workgroup that waits only ever waits for work that is already running: no deadlock whatever
//       the residency.  TR(j+1, .) is ticketed right after the first two tile rows of UP(j, ., .): the next row panel is
//       substituted while the bulk of update j still runs, and the last tiles of update j overlap the first tiles of
//       update j+1 — the per-launch tails and the 73 us bubble per super-panel of round 1's stream schedule are gone.
//
// Flags (unsigned, zeroed by a memset node before every factorisation; all accesses agent-scope relaxed atomics):
//   ticket, abort | per super-panel j: cv[4][4] / hv[4][4] tile versions of C_j / H_j, cdone, hdone, updone |
//   tr[j][J] 16-column groups of 128-column block J substituted | ver[I][J] panels applied to trailing tile (I, J).
// Data hand-over between workgroups: producer = `sc1` (write-through) stores, s_waitcnt vmcnt(0) by every wave, barrier,
// one lane updates the flag; consumer = one lane polls, barrier, `sc1` (L1-bypassing) loads.  Every spin is bounded: on
// time-out the abort flag is raised, every wait returns, the host reports HIOPAMD_ERR_HIP.

constexpr int DF_TICKET = 0, DF_ABORT = 1, DF_HDR = 16;
constexpr int DF_CH = 64;   // flags per super-panel in the chain section
constexpr int DF_CV = 0, DF_HV = 16, DF_CDONE = 33, DF_HDONE = 34, DF_UPDONE = 35;
constexpr int DF_MAXT = 40;           // chain tasks per role per super-panel (<= 28 used)
// Row-panel workspaces V (256 x N each), used round robin by the super-panels.  Super-panel j may write its V only when the
// update of super-panel j - DF_NVB has read the buffer completely.  With two buffers that wait closed a loop
// chain(j) -> TR(j) -> UP(j) complete -> chain(j+2) of ~620 us per two super-panels in the update-bound first third of the
// factorisation (profiles/r02_probes: the chain of super-panel 4 idled until the LAST tile of update 2 was done).
// Round 3: FOUR buffers, because a fused update task (DF_UP2, below) of the pair (j, j + 1) reads V_j while queue j + 1 runs: V_j is
// released one super-panel later than before, the fourth buffer gives the look-ahead back.
// The number of buffers is a run-time value, DfArgs::nvb (csrc/ldlt.hip::df_nvb_for: 4 unless HIOPAMD_DF_NVB asks for more).
constexpr int DF_NVB_MIN = 4;
constexpr int DF_ROLES = 16;

enum { DF_F = 1, DF_T = 2, DF_U = 3, DF_S = 4, DF_R = 5, DF_C = 6, DF_END = 0 };
// DF_UP2(j, I, J): K = 512 — the updates of super-panels j AND j + 1 applied to tile (I, J) in one pass over the C tile (one
// prologue / epilogue / task selection / flag wait for two tiles' worth of MFMAs, half the C traffic); a task of queue j + 1,
// for the tile rows behind super-panel j + 2 (I >= 2 j + 6).  The sums are accumulated in the same order as by the two
// separate tasks: results are bit-identical.
enum { DF_TR = 1, DF_UP = 2, DF_UPH = 3, DF_UP2 = 4 };

struct DfArgs {
  double* A;
  int64_t lda;
  int N;
  double* V;          // nvb x 256 x N (row panels of nvb consecutive super-panels, un-scaled)
  int nvb;            // row-panel workspaces (>= DF_NVB_MIN; >= nsp: no buffer is ever reused)
  int64_t ldv;
  double* dinv;
  double* Dblk;       // per 64-panel compact factored diagonal tile (64 x 64)
  double* Li;         // per 64-panel four 16 x 16 inverses
  double* Cd;         // per super-panel compact 256 x 256 diagonal block
  int* info;
  unsigned* flags;
  int nsp;            // super-panels of the matrix
  int nt;             // 128-tiles per side
  int nchain;         // super-panels handled by the chain kernel
  int last_has_next;  // does the last chained super-panel have a (full) next one
  int64_t off_chain, off_tr, off_ver;
  int dbg;            // != 0: record the per-super-panel time stamps (HIOPAMD_DF_STAMPS)
  int64_t off_ts;     // per super-panel 8 time stamps (100 MHz ticks, low 32 bits): see df_stamp
  int64_t off_ph;     // 16 words: phase accounting of the wide kernel (summed 100 MHz ticks / task counts), a.dbg != 0 only
  const int4* ctasks;   // [2 variants][DF_ROLES][DF_MAXT]
  const int4* wtasks;      // TR tasks grouped by super-panel, then UP tasks grouped by super-panel (first two tile rows first)
  int nwtasks;
  const unsigned* upcnt;   // UP tasks per super-panel
  const int4* wq;          // per super-panel {first TR task, TR tasks, first UP task, UP tasks} in wtasks
  const unsigned* wfirst;  // per super-panel: UP tasks of its first two tile rows (the rows of the next row panel)
  const int4* wf;          // per super-panel {first FAR task in wtasks, FAR tasks, queue whose FAR list feeds this NEAR list (-1: none), how many of its tasks must be taken first}
  int nwide;               // super-panels with wide-kernel work
  int64_t off_trb;         // per super-panel [2][4]: substitution tasks of 128-column block 2j+4+b that have stored block row P
  int spine_opt;           // HIOPAMD_DF_SPINE bits (see df_spine_step)
  int64_t off_cu;          // 512 words: workgroups of the wide kernel that have reported from CU (xcc, se, cu) — a workgroup's rank on its CU
  int jretire;             // the second (third, ...) workgroup of a CU leaves the wide kernel when its queue pointers reach this super-panel
  int64_t off_shadow;      // != 0 (HIOPAMD_DF_DEBUG): second copies of the substitution counters and version words, written right after the real ones
  long long timeout_ticks; // limit of every bounded wait, 100 MHz ticks (a multiple of the expected duration of the whole factorisation)
  int64_t off_run;         // != 0 (HIOPAMD_DF_CHECK=1, soak tests): one counter per task of the wide kernel — how often its ticket was handed out
  int64_t off_where;       // 2 words per workgroup of the wide kernel: (XCC, SE, CU) it ran on when it took its current task / when it published it
  int64_t off_snap;        // 1024 words: copy of the state words taken by the waiter whose wait expired, at that moment
  int64_t off_wg;          // 2 words per workgroup of the wide kernel: what it holds right now (see df_wg_state) — read by the host after a time-out
  int has_far;             // some super-panel has a FAR update list (HIOPAMD_DF_SPLIT=1)
};

#ifndef HIOPAMD_DF_FLAG_SCOPE
#define HIOPAMD_DF_FLAG_SCOPE __HIP_MEMORY_SCOPE_AGENT   /* (system scope: measured, no effect on the rare lost flag update) */
#endif
__device__ __forceinline__ unsigned df_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, HIOPAMD_DF_FLAG_SCOPE); }
__device__ __forceinline__ void df_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, HIOPAMD_DF_FLAG_SCOPE); }
// Read-modify-write operations on the flag words (counters, tickets).  At agent scope the compiler emits them without sc1; with
// -DHIOPAMD_DF_RMW_SCOPE=__HIP_MEMORY_SCOPE_SYSTEM they carry sc1 (tried against the rare lost flag update of DESIGN.md 3.1: no effect).
#ifndef HIOPAMD_DF_RMW_SCOPE
#define HIOPAMD_DF_RMW_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#ifdef HIOPAMD_DF_ADD_RETURNS   /* experiment: the returning form of the instruction (the wave waits for the old value) */
__device__ __forceinline__ void df_add(unsigned* p, unsigned v)
{
  const unsigned r = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, HIOPAMD_DF_RMW_SCOPE);
  asm volatile("" ::"v"(r));
}
#else
__device__ __forceinline__ void df_add(unsigned* p, unsigned v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, HIOPAMD_DF_RMW_SCOPE); }
#endif
// Polling read of a flag word; -DHIOPAMD_DF_POLL_RMW=1 makes it a read-modify-write that adds 0 (queues with the writes to that line at the
// memory side instead of overtaking them).  Tried against the late flag updates of DESIGN.md 3.1: same rate, so the plain load stays.
#ifndef HIOPAMD_DF_POLL_RMW
#define HIOPAMD_DF_POLL_RMW 0
#endif
__device__ __forceinline__ unsigned df_poll(const unsigned* p)
{
#if HIOPAMD_DF_POLL_RMW
  return __hip_atomic_fetch_add(const_cast<unsigned*>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ unsigned df_ticket(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, HIOPAMD_DF_RMW_SCOPE); }
// where am I: XCC id (3 bits) | SE (2) | CU (4) of the executing wave — a workgroup that the driver context-saved and restored elsewhere shows
// a different value at the end of a task than at its start (debug dump after a time-out)
__device__ __forceinline__ unsigned df_where()
{
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  return 0x80000000u | ((xcc & 7u) << 6) | (((hwid >> 13) & 3u) << 4) | ((hwid >> 8) & 15u);
}
// state word of a wide-kernel workgroup: kind (4 bits) | phase (4: 1 taken, 2 inputs there, 3 body done) | super-panel (8) | ticket-local index (16);
// second word: the task's third / fourth field.  0 = between tasks.  Two fire-and-forget stores per phase by lane 0.
__device__ __forceinline__ void df_wg_state(const DfArgs& a, int phase, const int4& tk)
{
  unsigned* p = a.flags + a.off_wg + 2 * (int64_t)blockIdx.x;
  df_st(p, phase == 0 ? 0xF3000000u : (((unsigned)tk.x & 15u) << 28) | (((unsigned)phase & 15u) << 24) | (((unsigned)tk.y & 255u) << 16) | ((unsigned)tk.z & 0xffffu));
  if(phase == 1) df_st(p + 1, (unsigned)tk.w);
  if(phase == 1) df_st(a.flags + a.off_where + 2 * (int64_t)blockIdx.x, df_where());
  if(phase == 4) df_st(a.flags + a.off_where + 2 * (int64_t)blockIdx.x + 1, df_where());
}

// up to four (flag >= value) conditions, in fixed slots (compile-time indices keep them in registers); an unused slot
// points at a word that always satisfies ">= 0"
// profiling stamps (a.dbg != 0 only): slot k of super-panel j; even k keep the EARLIEST time (stored inverted), odd k the latest
__device__ __forceinline__ void df_stamp(const DfArgs& a, int j, int k)
{
  if(!a.dbg) return;
  const unsigned now = (unsigned)wall_clock64();
  unsigned* p = a.flags + a.off_ts + (int64_t)j * 8 + k;
  if(k & 1) atomicMax(p, now);
  else atomicMax(p, 0xffffffffu - now);
}

// profiling aid (a.dbg != 0): mean time, since F(0) of the super-panel started, at which the k-th event of row r of the first
// H column happened (second half of the super-panels only); slot = 24 + 4 r + k
__device__ __forceinline__ void df_col4_stamp(const DfArgs& a, int j, int r, int k)
{
  if(!a.dbg || j < a.nchain / 2) return;
  const unsigned start = 0xffffffffu - df_ld(a.flags + a.off_ts + (int64_t)j * 8 + 0);
  atomicAdd(a.flags + a.off_ph + 24 + 4 * r + k, (unsigned)wall_clock64() - start);
}

struct DfWait {
  const unsigned* f[4];
  unsigned v[4];
  __device__ explicit DfWait(const unsigned* always)
  {
#pragma unroll
    for(int q = 0; q < 4; ++q) {
      f[q] = always;
      v[q] = 0u;
    }
  }
  template <int Q>
  __device__ void set(const unsigned* p, unsigned val)
  {
    f[Q] = p;
    v[Q] = val;
  }
};
// workgroup-wide: lane 0 polls every (flag >= value) pair; returns false when the factorisation was aborted.  Bounded by
// WALL-CLOCK time (s_memrealtime, 100 MHz) PER WAIT: a wait that has been spinning for DF_TIMEOUT_TICKS gives up, raises the
// abort word (the first one also leaves a diagnostic record in words 2..11: who waited for what) and every other wait
// returns within a few polls.  (Round 2 measured the limit from the START of the kernel: a factorisation that legitimately
// runs longer than the limit — an order beyond ~70 000, a throttled or shared device — aborted at its first slow wait although
// it was making progress.)  The sleep between polls keeps ~500 pollers from saturating the flags' memory channel.
// Back-off of the polling loops of the WIDE kernel: dozens of workgroups can wait for the same flag word (every update task of a tile row
// waits for the row's substitution counter), each re-reading it every ~0.4 us.  Once in a few thousand factorisations a publication became
// visible only after the waiters had given up; with the waiters backing off that happens several times less often.
// A waiter polls 8 times at 0.4 us, 8 times at 1.7 us, 3.4, 6.8, then every 13.6 us (level 4).
__device__ __forceinline__ void df_nap(unsigned level)
{
  if(level == 0u) __builtin_amdgcn_s_sleep(16);
  else if(level == 1u) __builtin_amdgcn_s_sleep(64);
  else {
    __builtin_amdgcn_s_sleep(127);
    if(level >= 3u) __builtin_amdgcn_s_sleep(127);
    if(level >= 4u) {
      __builtin_amdgcn_s_sleep(127);
      __builtin_amdgcn_s_sleep(127);
    }
  }
}
constexpr long long DF_TIMEOUT_TICKS = 300000000ll;   // 3 s: the upper end; the limit in force is DfArgs::timeout_ticks (set per order by the host)
__device__ __forceinline__ bool df_wait(unsigned* flags, const DfWait& w, int* sh_ok, long long t_start, int who, int a0, int a1,
                                        int a2, int a3)
{
  const long long t_limit = t_start;   // (the callers pass DfArgs::timeout_ticks here)
  if(threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    long long t_wait = 0;   // start of this wait's slow phase (first look at the clock)
    // all four polls in flight at once: the common case (everything already satisfied) costs one round trip, not four
    const unsigned g0 = df_ld(w.f[0]), g1 = df_ld(w.f[1]), g2 = df_ld(w.f[2]), g3 = df_ld(w.f[3]);
    const bool all_there = g0 >= w.v[0] && g1 >= w.v[1] && g2 >= w.v[2] && g3 >= w.v[3];
#pragma unroll
    for(int q = 0; q < 4; ++q) {   // unrolled: the pairs stay in registers (a runtime index would put them in scratch)
      if(ok && !all_there) {
        while(df_poll(w.f[q]) < w.v[q]) {
          if(who >= 100) __builtin_amdgcn_s_sleep(16);   // the chain kernel's 16 roles: few pollers, latency matters
          else df_nap(spins >> 3 < 4u ? spins >> 3 : 4u);   // 0.4 us x 8, 1.7 x 8, 3.4 x 8, 6.8 x 8, then 13.6 us
          if((++spins & 31u) == 0) {
            const long long now = (long long)wall_clock64();
            if(t_wait == 0) t_wait = now;
            const bool late = now - t_wait > t_limit;
            if(late || df_ld(flags + DF_ABORT) != 0) {
              if(late && atomicCAS(flags + DF_ABORT, 0u, 1u) == 0u) {
                df_st(flags + 2, (unsigned)who);
                df_st(flags + 3, (unsigned)a0);
                df_st(flags + 4, (unsigned)a1);
                df_st(flags + 5, (unsigned)a2);
                df_st(flags + 6, (unsigned)a3);
                df_st(flags + 7, (unsigned)q);
                df_st(flags + 8, w.v[q]);
                df_st(flags + 9, df_ld(w.f[q]));
                df_st(flags + 10, (unsigned)(w.f[q] - flags));
                df_st(flags + 11, (unsigned)((now - t_wait) >> 4));    // how long this wait lasted, 160 ns units
                // what every workgroup / role holds NOW (before anybody reacts to the abort word): flags[14] = offset of the state words,
                // flags[15] = offset of the snapshot area (both written by ldlt_df_prep_kernel; 0: no snapshot)
                if(const unsigned ow = df_ld(flags + 14), os = df_ld(flags + 15); ow != 0u && os != 0u)
                  for(unsigned q2 = 0; q2 < 1024u; ++q2) df_st(flags + os + q2, df_ld(flags + ow + q2));
              }
              ok = 0;
              break;
            }
          }
        }
      }
    }
    *sh_ok = ok;
  }
  __syncthreads();
  const bool r = __builtin_amdgcn_readfirstlane(*sh_ok) != 0;   // wave-uniform by construction: keep the branch scalar
  __syncthreads();
  return r;
}
// one non-blocking look at the four conditions (one round trip): true when all of them hold already
__device__ __forceinline__ bool df_peek(const DfWait& w, int* sh_ok)
{
  if(threadIdx.x == 0) {
    const unsigned g0 = df_ld(w.f[0]), g1 = df_ld(w.f[1]), g2 = df_ld(w.f[2]), g3 = df_ld(w.f[3]);
    *sh_ok = (g0 >= w.v[0] && g1 >= w.v[1] && g2 >= w.v[2] && g3 >= w.v[3]) ? 1 : 0;
  }
  __syncthreads();
  const bool r = __builtin_amdgcn_readfirstlane(*sh_ok) != 0;
  __syncthreads();
  return r;
}
// every wave drains its stores, then the workgroup meets: after this lane 0 may publish
__device__ __forceinline__ void df_drain()
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

struct DfTile {
  double* p;
  int64_t ld;
};
// 64 x 64 tile (r, c) of the window of super-panel j: r in 0..7 (4.. = rows of the next diagonal block), c in 0..7
__device__ __forceinline__ DfTile df_tile(const DfArgs& a, int j, int r, int c)
{
  if(r < 4 && c < 4) return DfTile{a.Cd + (int64_t)j * (LD_NB * LD_NB) + (64 * r) * LD_NB + 64 * c, LD_NB};
  if(r < 4) return DfTile{a.A + ((int64_t)LD_NB * j + 64 * r) * a.lda + (int64_t)LD_NB * (j + 1) + 64 * (c - 4), a.lda};
  return DfTile{a.Cd + (int64_t)(j + 1) * (LD_NB * LD_NB) + (64 * (r - 4)) * LD_NB + 64 * (c - 4), LD_NB};
}
// the same next-diagonal-block tile inside the matrix itself (its first update reads it from there: the wide kernel's
// updates of earlier super-panels went to the matrix, the compact copy does not exist yet)
__device__ __forceinline__ DfTile df_tile_in_matrix(const DfArgs& a, int j, int r, int c)
{
  return DfTile{a.A + ((int64_t)LD_NB * (j + 1) + 64 * (r - 4)) * a.lda + (int64_t)LD_NB * (j + 1) + 64 * (c - 4), a.lda};
}
__device__ __forceinline__ DfTile df_vtile(const DfArgs& a, int j, int p, int c)
{
  return DfTile{a.V + (int64_t)(j % a.nvb) * LD_NB * a.ldv + (int64_t)(64 * p) * a.ldv + (int64_t)LD_NB * j + 64 * c, a.ldv};
}
// version counter of window tile (r, c) of super-panel j and the value it has before any task of this super-panel touched it
__device__ __forceinline__ unsigned* df_ver(const DfArgs& a, int j, int r, int c, unsigned* base)
{
  unsigned* cf = a.flags + a.off_chain + (int64_t)j * DF_CH;
  if(r < 4 && c < 4) {
    *base = 4u;   // the four updates of the previous super-panel's pivots (pre-credited for j = 0)
    return cf + DF_CV + r * 4 + c;
  }
  *base = 0u;
  if(r < 4) return cf + DF_HV + r * 4 + (c - 4);
  return cf + DF_CH + DF_CV + (r - 4) * 4 + (c - 4);
}
// trailing-matrix 128-tile that contains window tile (r, c) (for tiles that the wide kernel updates: H and Nn tiles)
__device__ __forceinline__ const unsigned* df_wide_ver(const DfArgs& a, int j, int r, int c)
{
  const int I = 2 * j + (r >> 1), J = 2 * j + (c >> 1);
  return a.flags + a.off_ver + (int64_t)I * a.nt + J;
}

// ---------------------------------------------------------------------------------------------------------------
// chain tasks (256 threads)
// ---------------------------------------------------------------------------------------------------------------
// F(p): tile (p, p) of C_j = U^T D U; emits U (scaled rows) + D into the tile, the compact copy Dk, dinv, the 16x16 inverses
__device__ __forceinline__ void df_task_factor(const DfArgs& a, int j, int p, double (*S)[LD_nb + 1], double* sdinv, int tid)
{
  const DfTile t = df_tile(a, j, p, p);
  double sv[LD_nb * LD_nb / kBlock];
#pragma unroll
  for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
    const int e = tid + q * kBlock;
    sv[q] = ldg_sc1(t.p + (int64_t)(e >> 6) * t.ld + (e & 63));
  }
#pragma unroll
  for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
    const int e = tid + q * kBlock;
    const int r = e >> 6, c = e & 63;
    S[r][c] = (c >= r) ? sv[q] : 0.0;
  }
  if(tid < LD_nb) sdinv[tid] = 1.0;
  __syncthreads();
  const int k0 = LD_NB * j + 64 * p;
  double* Li = a.Li + (int64_t)(k0 / LD_nb) * (4 * LD_SB * LD_SB);
  double* Dk = a.Dblk + (int64_t)(k0 / LD_nb) * (LD_nb * LD_nb);
  diag_factor_lds<true>(S, sdinv, LD_nb, k0, a.info, Li, tid);
  for(int e = tid; e < LD_nb * LD_nb; e += kBlock) {
    const int r = e >> 6, c = e & 63;
    double v = S[r][c];
    if(c > r) v *= sdinv[r];
    const bool in = c >= r;
    stg_sc1(Dk + e, in ? v : 0.0);
    if(in) stg_sc1(t.p + (int64_t)r * t.ld + c, v);
  }
  if(tid < LD_nb) stg_sc1(a.dinv + k0 + tid, sdinv[tid]);
}

// T(p, c): X = tile (p, c): V = L_pp^-1 X (to the V workspace), U = D_p^-1 V (in place).  One wave per 16 columns; the
// 16-row block substitution with the 16 x 16 inverses on fp64 MFMA (the in-block part of block_row_solve)
__device__ __forceinline__ void df_task_solve(const DfArgs& a, int j, int p, int c, int tid)
{
  const int lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
  const DfTile x = df_tile(a, j, p, c), vt = df_vtile(a, j, p, c);
  const int k0 = LD_NB * j + 64 * p;
  const double* Li = a.Li + (int64_t)(k0 / LD_nb) * (4 * LD_SB * LD_SB);
  const double* Dk = a.Dblk + (int64_t)(k0 / LD_nb) * (LD_nb * LD_nb);
  const int cl = 16 * w + li;
  double4_t t[4];
  double nl[6][4], iv[4][4], dsc[4][4];
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int r = 0; r < 4; ++r) {
      t[I][r] = ldg_sc1(x.p + (int64_t)(16 * I + g + 4 * r) * x.ld + cl);
      dsc[I][r] = ldg_sc1(a.dinv + k0 + 16 * I + g + 4 * r);
    }
#pragma unroll
  for(int I = 1; I < 4; ++I)
#pragma unroll
    for(int J = 0; J < I; ++J)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) nl[I * (I - 1) / 2 + J][kk] = -ldg_sc1(Dk + (16 * J + 4 * kk + g) * LD_nb + 16 * I + li);
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) iv[I][kk] = ldg_sc1(Li + I * 256 + li * 16 + 4 * kk + g);
  double4_t vp[4];
#pragma unroll
  for(int I = 0; I < 4; ++I) {
    double4_t u = t[I];
#pragma unroll
    for(int J = 0; J < 4; ++J) {
      if(J < I) {
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) u = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[I * (I - 1) / 2 + J][kk], vp[J][kk], u, 0, 0, 0);
      }
    }
    double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[I][kk], u[kk], v, 0, 0, 0);
    vp[I] = v;
  }
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int r = 0; r < 4; ++r) {
      const int row = 16 * I + g + 4 * r;
      stg_sc1(vt.p + (int64_t)row * vt.ld + cl, vp[I][r]);
      stg_sc1(x.p + (int64_t)row * x.ld + cl, vp[I][r] * dsc[I][r]);
    }
}

// U(p; ta, tb): tile (ta, tb) -= V(p, ta)^T U(p, tb)  (64 x 64 x 64): 4 waves as 2 x 2, each a 32 x 32 quadrant = 2 x 2
// MFMA tiles; operands straight from L2 into the MFMA register layout.
// src != dst for the first update of a next-diagonal-block tile (read from the matrix, written to the compact copy).
__device__ __forceinline__ void df_task_update(const DfArgs& a, int j, int p, int ta, int tb, const DfTile src, const DfTile dst,
                                               int tid)
{
  const int lane = tid & 63, w = tid >> 6, lk = lane >> 4, li = lane & 15;
  const int wr = w >> 1, wc = w & 1;
  const DfTile va = df_vtile(a, j, p, ta), ub = df_tile(a, j, p, tb);
  double4_t acc[2][2];
#pragma unroll
  for(int i = 0; i < 2; ++i)
#pragma unroll
    for(int q = 0; q < 2; ++q) acc[i][q] = double4_t{0.0, 0.0, 0.0, 0.0};
  double cv[2][2][4];
#pragma unroll
  for(int i = 0; i < 2; ++i)
#pragma unroll
    for(int q = 0; q < 2; ++q)
#pragma unroll
      for(int reg = 0; reg < 4; ++reg)
        cv[i][q][reg] = ldg_sc1(src.p + (int64_t)(32 * wr + 16 * i + lk + 4 * reg) * src.ld + 32 * wc + 16 * q + li);
  // every operand of the task in flight at once (80 loads per lane: the chain kernel has the registers): ONE memory round
  // trip per update instead of three — these 64 x 64 x 64 updates are links of the serial chains between the spine steps
  {
    double av[16][2], bv[16][2];
#pragma unroll
    for(int kk = 0; kk < 16; ++kk) {
      const int k = 4 * kk + lk;
#pragma unroll
      for(int i = 0; i < 2; ++i) av[kk][i] = ldg_sc1(va.p + (int64_t)k * va.ld + 32 * wr + 16 * i + li);
#pragma unroll
      for(int q = 0; q < 2; ++q) bv[kk][q] = ldg_sc1(ub.p + (int64_t)k * ub.ld + 32 * wc + 16 * q + li);
    }
#pragma unroll
    for(int kk = 0; kk < 16; ++kk)
#pragma unroll
      for(int i = 0; i < 2; ++i)
#pragma unroll
        for(int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk][i], bv[kk][q], acc[i][q], 0, 0, 0);
  }
#pragma unroll
  for(int i = 0; i < 2; ++i)
#pragma unroll
    for(int q = 0; q < 2; ++q)
#pragma unroll
      for(int reg = 0; reg < 4; ++reg)
        stg_sc1(dst.p + (int64_t)(32 * wr + 16 * i + lk + 4 * reg) * dst.ld + 32 * wc + 16 * q + li, cv[i][q][reg] - acc[i][q][reg]);
}

// LDS of a chain workgroup (one workgroup per CU: 160 KB available)
struct DfChainLds {
  double S[LD_nb][LD_nb + 1];    // the diagonal tile being factored / carried from the previous spine step
  double sdinv[LD_nb];
  double Li[4 * LD_SB * LD_SB];  // the four 16 x 16 inverses of the last factored tile
  double Vl[LD_nb][LD_nb + 1];   // V(p, p+1) (un-scaled) and U(p, p+1) of the spine step, for its own update
  double Ul[LD_nb][LD_nb + 1];   // (Vl also receives the prefetched tile (p, p+1) while F(p) runs)
  double Cl[LD_nb][LD_nb + 1];   // prefetched tile (p+1, p+1), read when its update is written out
  int pref[4];                   // per wave 1..3: bit 0 = its rows of tile (p, p+1) are in Vl, bit 1 = of (p+1, p+1) in Cl
};
// a tile solve of the spine whose results are stored but not yet published (HIOPAMD_DF_SPINE bit 0)
struct DfPendT {
  unsigned* v = nullptr;   // version counter of the solved tile
  unsigned* f = nullptr;   // CDONE / HDONE word of its super-panel
  int j = 0;
  bool stamp = false;
};
__device__ __forceinline__ void df_drain();
__device__ __forceinline__ void df_flush_pend(const DfArgs& a, DfPendT& pt, int tid)
{
  if(!pt.v) return;   // uniform
  df_drain();
  if(tid == 0) {
    if(pt.stamp) df_stamp(a, pt.j, 3);
    df_add(pt.v, 1u);
    df_add(pt.f, 1u);
  }
  pt.v = nullptr;
  __syncthreads();
}

// S(p): the spine step F(p) -> T(p, p+1) -> U(p; p+1, p+1) in ONE task.  As three tasks every hand-over costs a drain, a
// flag round trip, a poll and a reload from L2 (12 serial round trips of ~1.5 us per pivot, 131 us per super-panel, the
// critical path of the whole factorisation); here the factored tile, its 16 x 16 inverses, V / U of the solved tile and the
// updated next diagonal tile stay in LDS: F's results are published as soon as they are drained (the other roles' tile
// solves need them), T's and U's together at the end.  `carried`: the tile to factor is already in L.S (left there by the
// previous step's update).  Returns false when the factorisation was aborted.
// Latency cuts of the step (a.spine_opt, HIOPAMD_DF_SPINE; every one removes a memory round trip from the spine):
//   bit 0  the publication of T(p, p+1) (a drain of its stores) waits for the START of the next spine step, where those stores
//          have long landed; the updated tile (p+1, p+1) is not written to memory at all (it travels in LDS; only the hand-over
//          to the stepwise kernels of a ragged order reads it);
//   bit 1  while wave 0 factors the last 16 x 16 sub-block of F(p), the idle waves 1-3 look at the version counters of tiles
//          (p, p+1) and (p+1, p+1) and, when they are complete, fetch them into LDS: the tile solve starts from LDS;
//   bit 2  with the tile in LDS the tile solve's MFMA chain runs BEFORE F is published: the drain of F's stores and the flag
//          round trip of the peek are covered by it.
__device__ __forceinline__ bool df_spine_step(const DfArgs& a, int j, int p, bool with_tu, bool carried, DfChainLds& L, int* sh_ok,
                                              long long t_start, int tid, unsigned*& pending, DfPendT& pt)
{
  const int lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
  const int k0 = LD_NB * j + 64 * p;
  const int opt = a.spine_opt;
  unsigned* cf = a.flags + a.off_chain + (int64_t)j * DF_CH;
  double* Li = a.Li + (int64_t)(k0 / LD_nb) * (4 * LD_SB * LD_SB);
  double* Dk = a.Dblk + (int64_t)(k0 / LD_nb) * (LD_nb * LD_nb);
  const DfTile tpp = df_tile(a, j, p, p);
  unsigned bpp;
  unsigned* vpp = df_ver(a, j, p, p, &bpp);
  df_flush_pend(a, pt, tid);   // the previous step's tile solve: before anything here can wait
  const unsigned ts0 = a.dbg ? (unsigned)wall_clock64() : 0u;
  // the tiles of the T / U part (c = p + 1), known before F so that the idle waves can prefetch them
  const int c = p + 1;
  unsigned bpc = 0u, bcc = 0u;
  unsigned* vpc = with_tu ? df_ver(a, j, p, c, &bpc) : nullptr;
  unsigned* vcc = with_tu ? df_ver(a, j, c, c, &bcc) : nullptr;
  const DfTile x = with_tu ? df_tile(a, j, p, c) : tpp, tcc = with_tu ? df_tile(a, j, c, c) : tpp;
  auto prefetch = [&](int wv) {
    if(!(with_tu && (opt & 2))) return;   // uniform
    const unsigned gx = (unsigned)__builtin_amdgcn_readfirstlane((int)df_ld(vpc));
    const unsigned gc = (unsigned)__builtin_amdgcn_readfirstlane((int)df_ld(vcc));
    const bool okx = gx >= bpc + (unsigned)p, okc = gc >= bcc + (unsigned)p;
    constexpr int NR = 22;   // rows wv-1, wv+2, ... of each tile
    double vx[NR], vc[NR];
    if(okx) {
#pragma unroll
      for(int i = 0; i < NR; ++i) {
        const int r = wv - 1 + 3 * i;
        if(r < LD_nb) vx[i] = ldg_sc1(x.p + (int64_t)r * x.ld + lane);
      }
    }
    if(okc) {
#pragma unroll
      for(int i = 0; i < NR; ++i) {
        const int r = wv - 1 + 3 * i;
        if(r < LD_nb) vc[i] = ldg_sc1(tcc.p + (int64_t)r * tcc.ld + lane);
      }
    }
    if(okx) {
#pragma unroll
      for(int i = 0; i < NR; ++i) {
        const int r = wv - 1 + 3 * i;
        if(r < LD_nb) L.Vl[r][lane] = vx[i];
      }
    }
    if(okc) {
#pragma unroll
      for(int i = 0; i < NR; ++i) {
        const int r = wv - 1 + 3 * i;
        if(r < LD_nb) L.Cl[r][lane] = vc[i];
      }
    }
    if(lane == 0) L.pref[wv] = (okx ? 1 : 0) | (okc ? 2 : 0);
  };
  // ---- F(p)
  if(!carried) {
    DfWait w0(a.flags + DF_ABORT);
    w0.set<0>(vpp, bpp + p);
    if(!df_wait(a.flags, w0, sh_ok, t_start, 100, j, DF_S, p, 0)) return false;
    double sv[LD_nb * LD_nb / kBlock];
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      sv[q] = ldg_sc1(tpp.p + (int64_t)(e >> 6) * tpp.ld + (e & 63));
    }
#pragma unroll
    for(int q = 0; q < LD_nb * LD_nb / kBlock; ++q) {
      const int e = tid + q * kBlock;
      const int r = e >> 6, c = e & 63;
      L.S[r][c] = (c >= r) ? sv[q] : 0.0;
    }
  }
  if(tid < LD_nb) L.sdinv[tid] = 1.0;
  if(tid < 4) L.pref[tid] = 0;
  __syncthreads();
  if(tid == 0 && p == 0) df_stamp(a, j, 0);
  const unsigned tsa = a.dbg ? (unsigned)wall_clock64() : 0u;
  diag_factor_lds<true, true>(L.S, L.sdinv, LD_nb, k0, a.info, Li, tid, L.Li, prefetch, a.dbg ? a.flags + a.off_ph + 44 : nullptr);
  const int pf = __builtin_amdgcn_readfirstlane(L.pref[1] & L.pref[2] & L.pref[3]);
  const bool prefx = (pf & 1) != 0, prefc = (pf & 2) != 0;
  if(a.dbg && tid == 0) {
    atomicAdd(a.flags + a.off_ph + 16, tsa - ts0);                        // load / LDS set-up
    atomicAdd(a.flags + a.off_ph + 17, (unsigned)wall_clock64() - tsa);   // diag_factor_lds
  }
  for(int e = tid; e < LD_nb * LD_nb; e += kBlock) {
    const int r = e >> 6, c = e & 63;
    double v = L.S[r][c];
    if(c > r) v *= L.sdinv[r];
    const bool in = c >= r;
    stg_sc1(Dk + e, in ? v : 0.0);
    if(in) stg_sc1(tpp.p + (int64_t)r * tpp.ld + c, v);
  }
  if(tid < LD_nb) stg_sc1(a.dinv + k0 + tid, L.sdinv[tid]);
  auto publish_f = [&]() {
    df_drain();
    if(tid == 0 && p == 3) df_stamp(a, j, 1);
    if(tid == 0) {
      if(pending) df_add(pending, 1u);   // the previous step's update of this tile: its stores were drained just now as well
      df_add(vpp, 1u);
      df_add(cf + DF_CDONE, 1u);
      if(a.dbg && j >= a.nchain / 2) {
        const unsigned start = 0xffffffffu - df_ld(a.flags + a.off_ts + (int64_t)j * 8 + 0);
        atomicAdd(a.flags + a.off_ph + 40 + p, (unsigned)wall_clock64() - start);   // F(p) published
      }
    }
  };
  if(!with_tu) {
    publish_f();
    pending = nullptr;
    return true;
  }
  const unsigned ts1 = a.dbg ? (unsigned)wall_clock64() : 0u;
  // ---- T(p, c), c = p + 1: the tile has the updates of the pivots < p (role 1), the V workspace of this parity is free
  const DfTile vt = df_vtile(a, j, p, c);
  const int cl = 16 * w + li;
  const int wr = w >> 1, wc = w & 1, lk = g;
  double4_t vp[4];
  double dsc[4][4];
  // the tile solve up to its results in registers: X from LDS (prefetched) or from memory, operands of F from LDS
  auto solve_tile = [&](bool from_lds) {
    double4_t t[4];
    if(from_lds) {   // uniform
#pragma unroll
      for(int I = 0; I < 4; ++I)
#pragma unroll
        for(int r = 0; r < 4; ++r) t[I][r] = L.Vl[16 * I + g + 4 * r][cl];
    } else {
#pragma unroll
      for(int I = 0; I < 4; ++I)
#pragma unroll
        for(int r = 0; r < 4; ++r) t[I][r] = ldg_sc1(x.p + (int64_t)(16 * I + g + 4 * r) * x.ld + cl);
    }
    double nl[6][4], iv[4][4];
#pragma unroll
    for(int I = 0; I < 4; ++I)
#pragma unroll
      for(int r = 0; r < 4; ++r) dsc[I][r] = L.sdinv[16 * I + g + 4 * r];
#pragma unroll
    for(int I = 1; I < 4; ++I)
#pragma unroll
      for(int J = 0; J < I; ++J)
#pragma unroll
        for(int kk = 0; kk < 4; ++kk) {
          const int rr = 16 * J + 4 * kk + g;
          nl[I * (I - 1) / 2 + J][kk] = -(L.S[rr][16 * I + li] * L.sdinv[rr]);   // scaled row of U, as F emitted it
        }
#pragma unroll
    for(int I = 0; I < 4; ++I)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) iv[I][kk] = L.Li[I * 256 + li * 16 + 4 * kk + g];
#pragma unroll
    for(int I = 0; I < 4; ++I) {
      double4_t u = t[I];
#pragma unroll
      for(int J = 0; J < 4; ++J) {
        if(J < I) {
#pragma unroll
          for(int kk = 0; kk < 4; ++kk) u = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[I * (I - 1) / 2 + J][kk], vp[J][kk], u, 0, 0, 0);
        }
      }
      double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[I][kk], u[kk], v, 0, 0, 0);
      vp[I] = v;
    }
  };
  unsigned ts2 = 0u;
  {
    DfWait w1(a.flags + DF_ABORT);
    w1.set<0>(vpc, bpc + p);
    w1.set<1>(vcc, bcc + p);
    if(j >= a.nvb) w1.set<2>(a.flags + a.off_chain + (int64_t)(j - a.nvb) * DF_CH + DF_UPDONE, a.upcnt[j - a.nvb]);
    if((opt & 4) && prefx) {
      // the flag round trip and the drain of F's stores run under the solve; nothing of the solve is stored before the
      // conditions are known to hold (the V workspace of this parity may still be read by update j-2)
      unsigned g0 = 0u, g1 = 0u, g2 = 0u;
      if(tid == 0) {
        g0 = df_ld(w1.f[0]);
        g1 = df_ld(w1.f[1]);
        g2 = df_ld(w1.f[2]);
      }
      solve_tile(true);
      if(tid == 0) *sh_ok = (g0 >= w1.v[0] && g1 >= w1.v[1] && g2 >= w1.v[2]) ? 1 : 0;
      publish_f();   // (its drain is also the barrier that makes sh_ok visible)
      pending = nullptr;
      const bool ready = __builtin_amdgcn_readfirstlane(*sh_ok) != 0;
      __syncthreads();
      if(!ready && !df_wait(a.flags, w1, sh_ok, t_start, 100, j, DF_S, p, 1)) return false;
      ts2 = a.dbg ? (unsigned)wall_clock64() : 0u;
    } else {
      // one look at the conditions while F's stores drain, then publish F (the other roles' tile solves wait for it) before
      // any blocking wait: in the phases where the wide kernel is behind, that wait is long
      const bool ready = df_peek(w1, sh_ok);
      publish_f();
      pending = nullptr;
      if(a.dbg && p == 3 && j >= a.nchain / 2 && tid == 0) {   // profiling aid: which of the conditions is the late one?
        const unsigned t0 = (unsigned)wall_clock64();
        unsigned tw[3] = {0u, 0u, 0u};
        for(int q = 0; q < 3; ++q) {
          unsigned spins = 0;
          while(df_ld(w1.f[q]) < w1.v[q] && ++spins < 100000u) __builtin_amdgcn_s_sleep(4);
          tw[q] = (unsigned)wall_clock64() - t0;
        }
        atomicAdd(a.flags + a.off_ph + 22, tw[0]);
        atomicAdd(a.flags + a.off_ph + 23, tw[1] - tw[0]);
      }
      if(!ready && !df_wait(a.flags, w1, sh_ok, t_start, 100, j, DF_S, p, 1)) return false;
      ts2 = a.dbg ? (unsigned)wall_clock64() : 0u;
      solve_tile(prefx);
    }
  }
  double cv[2][2][4];
  if(!prefc) {
#pragma unroll
    for(int i = 0; i < 2; ++i)
#pragma unroll
      for(int q = 0; q < 2; ++q)
#pragma unroll
        for(int reg = 0; reg < 4; ++reg)
          cv[i][q][reg] = ldg_sc1(tcc.p + (int64_t)(32 * wr + 16 * i + lk + 4 * reg) * tcc.ld + 32 * wc + 16 * q + li);
  }
#pragma unroll
  for(int I = 0; I < 4; ++I)
#pragma unroll
    for(int r = 0; r < 4; ++r) {
      const int row = 16 * I + g + 4 * r;
      const double vv = vp[I][r], uu = vp[I][r] * dsc[I][r];
      stg_sc1(vt.p + (int64_t)row * vt.ld + cl, vv);
      stg_sc1(x.p + (int64_t)row * x.ld + cl, uu);
      L.Vl[row][cl] = vv;
      L.Ul[row][cl] = uu;
    }
  __syncthreads();   // V / U of the tile complete in LDS; every wave is done with L.S (nl) and L.sdinv
  // ---- U(p; c, c): tile (c, c) -= V^T U, operands from LDS; the result stays in L.S for the next step's F
  double4_t acc[2][2];
#pragma unroll
  for(int i = 0; i < 2; ++i)
#pragma unroll
    for(int q = 0; q < 2; ++q) acc[i][q] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for(int kk = 0; kk < 16; ++kk) {
    const int k = 4 * kk + lk;
    double av[2], bv[2];
#pragma unroll
    for(int i = 0; i < 2; ++i) av[i] = L.Vl[k][32 * wr + 16 * i + li];
#pragma unroll
    for(int q = 0; q < 2; ++q) bv[q] = L.Ul[k][32 * wc + 16 * q + li];
#pragma unroll
    for(int i = 0; i < 2; ++i)
#pragma unroll
      for(int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[q], acc[i][q], 0, 0, 0);
  }
  // V / U of tile (p, c) were stored before the products above: by now they have landed, publish them (the companion's
  // U(p; p+1, p+2) waits for exactly this) before the C tile goes out
  if(opt & 1) {
    pt.v = vpc;
    pt.f = cf + (c < 4 ? DF_CDONE : DF_HDONE);
    pt.j = j;
    pt.stamp = c >= 4;
  } else {
    df_drain();
    if(tid == 0 && c >= 4) df_stamp(a, j, 3);
    if(tid == 0) {
      df_add(vpc, 1u);
      df_add(cf + (c < 4 ? DF_CDONE : DF_HDONE), 1u);
    }
  }
  // the updated tile travels in LDS; memory needs it only for the hand-over to the stepwise kernels of a ragged order
  const bool to_memory = !(opt & 1) || (c == 4 && j + 1 >= a.nchain);
#pragma unroll
  for(int i = 0; i < 2; ++i)
#pragma unroll
    for(int q = 0; q < 2; ++q)
#pragma unroll
      for(int reg = 0; reg < 4; ++reg) {
        const int row = 32 * wr + 16 * i + lk + 4 * reg, col = 32 * wc + 16 * q + li;
        const double c0 = prefc ? L.Cl[row][col] : cv[i][q][reg];
        const double r = c0 - acc[i][q][reg];
        if(to_memory) stg_sc1(tcc.p + (int64_t)row * tcc.ld + col, r);
        L.S[row][col] = (col >= row) ? r : 0.0;
      }
  // the count of this update is published with the NEXT step's F (one drain less on the critical path): only tasks that
  // also wait for that F look at it — the tile itself travels in LDS
  __syncthreads();
  pending = vcc;
  if(a.dbg && tid == 0) {   // spine accounting: F part | wait for the tile's older updates | T + U part | steps
    const unsigned ts3 = (unsigned)wall_clock64();
    atomicAdd(a.flags + a.off_ph + 12, ts1 - ts0);
    atomicAdd(a.flags + a.off_ph + 13, ts2 - ts1);
    if(j >= a.nchain / 2) atomicAdd(a.flags + a.off_ph + 18 + p, ts2 - ts1);   // second half of the factorisation, by pivot
    atomicAdd(a.flags + a.off_ph + 14, ts3 - ts2);
    atomicAdd(a.flags + a.off_ph + 15, 1u);
  }
  return true;
}

// R(p): the spine's companion T(p, p+2) -> U(p; p+2, p+2) -> U(p; p+1, p+2) in one task (what S(p+1) and R(p+1) need next);
// only the last part waits for the spine's T(p, p+1)
__device__ __forceinline__ bool df_companion_step(const DfArgs& a, int j, int p, bool with_diag, int* sh_ok, long long t_start,
                                                  int tid)
{
  unsigned* cf = a.flags + a.off_chain + (int64_t)j * DF_CH;
  const int c = p + 2, m = p + 1;
  unsigned bpp, bpc, bpm, bmc, bcc;
  unsigned* vpp = df_ver(a, j, p, p, &bpp);
  unsigned* vpc = df_ver(a, j, p, c, &bpc);
  unsigned* vpm = df_ver(a, j, p, m, &bpm);
  unsigned* vmc = df_ver(a, j, m, c, &bmc);
  unsigned* vcc = df_ver(a, j, c, c, &bcc);
  {
    DfWait w0(a.flags + DF_ABORT);
    w0.set<0>(vpp, bpp + p + 1);   // F(p) published
    w0.set<1>(vpc, bpc + p);
    if(j >= a.nvb) w0.set<2>(a.flags + a.off_chain + (int64_t)(j - a.nvb) * DF_CH + DF_UPDONE, a.upcnt[j - a.nvb]);
    if(!df_wait(a.flags, w0, sh_ok, t_start, 101, j, DF_R, p, 0)) return false;
  }
  df_task_solve(a, j, p, c, tid);
  df_drain();
  if(tid == 0 && c >= 4) df_stamp(a, j, 3);
  if(tid == 0) {
    df_add(vpc, 1u);
    df_add(cf + (c < 4 ? DF_CDONE : DF_HDONE), 1u);
    if(c == 4) df_col4_stamp(a, j, p, 3);
  }
  const DfTile tmc = df_tile(a, j, m, c), tcc = df_tile(a, j, c, c);
  if(with_diag) {   // U(p; p+2, p+2) needs only this task's own V / U (drained above): before the wait for the spine
    DfWait w1(a.flags + DF_ABORT);
    w1.set<0>(vcc, bcc + p);
    if(!df_wait(a.flags, w1, sh_ok, t_start, 101, j, DF_R, p, 1)) return false;
    df_task_update(a, j, p, c, c, tcc, tcc, tid);
  }
  {
    DfWait w2(a.flags + DF_ABORT);
    w2.set<0>(vpm, bpm + p + 1);   // the spine's T(p, p+1) published
    w2.set<1>(vmc, bmc + p);
    if(!df_wait(a.flags, w2, sh_ok, t_start, 101, j, DF_R, p, 2)) return false;
  }
  df_task_update(a, j, p, m, c, tmc, tmc, tid);
  df_drain();
  if(tid == 0) {
    if(with_diag) df_add(vcc, 1u);
    df_add(vmc, 1u);
    if(c == 4) df_col4_stamp(a, j, m, p);
  }
  return true;
}

// C(p, c): a column step — T(p, c) followed by the updates U(p; a, c), a = p+1 .. min(3, c), of the tiles below it in the
// same window column, in ONE task.  As separate tasks the ten tasks of an H column (T(0,c), U(0;1..3,c), T(1,c), ...) were a
// serial chain of ~13 us links, as long as the spine's own period: the spine waited 29 us per super-panel at p = 3 for the
// last of them.  Inside one task only the first link pays the task overheads; the updates read this task's own U(p, c) back
// from L2 (already drained for the publish of T) and are drained together at the end.
__device__ __forceinline__ bool df_column_step(const DfArgs& a, int j, int p, int c, int amax_in, int role, int* sh_ok,
                                               long long t_start, int tid)
{
  unsigned* cf = a.flags + a.off_chain + (int64_t)j * DF_CH;
  unsigned bpp, bpc;
  unsigned* vpp = df_ver(a, j, p, p, &bpp);
  unsigned* vpc = df_ver(a, j, p, c, &bpc);
  {
    DfWait w(a.flags + DF_ABORT);
    w.set<0>(vpp, bpp + p + 1);
    w.set<1>(vpc, bpc + p);
    if(c >= 4 && p == 0) w.set<2>(df_wide_ver(a, j, p, c), (unsigned)j);
    if(j >= a.nvb) w.set<3>(a.flags + a.off_chain + (int64_t)(j - a.nvb) * DF_CH + DF_UPDONE, a.upcnt[j - a.nvb]);
    if(!df_wait(a.flags, w, sh_ok, t_start, 100 + role, j, DF_C, p, c)) return false;
  }
  df_task_solve(a, j, p, c, tid);
  df_drain();
  if(tid == 0 && c >= 4) df_stamp(a, j, 3);
  if(tid == 0) {
    df_add(vpc, 1u);
    df_add(cf + (c < 4 ? DF_CDONE : DF_HDONE), 1u);
    if(c == 4) df_col4_stamp(a, j, p, 3);      // T(p, 4) published
  }
  const int amax = amax_in > 0 ? amax_in : (c < 3 ? c : 3);   // (the first H column keeps only its next tile: see df_chain_tasks)
  for(int ta = p + 1; ta <= amax; ++ta) {
    unsigned ba, bt;
    unsigned* va = df_ver(a, j, p, ta, &ba);
    unsigned* vt = df_ver(a, j, ta, c, &bt);
    DfWait w(a.flags + DF_ABORT);
    w.set<0>(va, ba + p + 1);     // V(p, ta) published (this task's own when ta == c)
    w.set<1>(vt, bt + p);
    if(c >= 4 && p == 0) w.set<2>(df_wide_ver(a, j, ta, c), (unsigned)j);
    if(!df_wait(a.flags, w, sh_ok, t_start, 100 + role, j, 1000 * p + DF_C, ta, c)) return false;
    const DfTile dst = df_tile(a, j, ta, c);
    df_task_update(a, j, p, ta, c, dst, dst, tid);
    df_drain();                       // published tile by tile: the next pivot's tile solve of this column waits for the first one
    if(tid == 0) {
      df_add(vt, 1u);
      if(c == 4) df_col4_stamp(a, j, ta, p);   // update of pivot p on tile (ta, 4) published
    }
  }
  return true;
}

// LDS of a chain workgroup / of a wide workgroup as structs, so that the two bodies can also run as ONE dispatch
// (ldlt_df_one_kernel: the counter passes of rocprofv3 serialise dispatches, and the chain / wide pair wait for each other)
struct DfChainShared {
  DfChainLds L;
  int4 tasks[2][DF_MAXT];   // this role's task lists (with / without a next super-panel): no L2 round trip per task
  int ok;
};
__global__ __launch_bounds__(kBlock, 1) void ldlt_chain_kernel(const DfArgs a)
{
  __shared__ DfChainLds L;
  __shared__ int4 sh_tasks[2][DF_MAXT];   // this role's task lists (with / without a next super-panel): no L2 round trip per task
  __shared__ int sh_ok;
  const int role = blockIdx.x;
#include "ldlt_chain_body.inc"
}

// ---------------------------------------------------------------------------------------------------------------
// wide tasks (256 threads, LDS buffer shared by the two task kinds)
// ---------------------------------------------------------------------------------------------------------------
// TR(j, c16): DF_TRW columns starting at c16 of the tail of row panel j.  FOUR waves: wave I owns the 16-row sub-block I of
// every 64-row block row (algorithm of ldlt_headtrsm_kernel); every shared operand through sc1 loads.
// `early`: the task was taken before C_j was completely factored (see the selection loop): block row P then waits for the
// tiles of column P of C_j — F(P) and T(q, P), q < P — and the substitution advances in step with the chain kernel.
// `rowflags` != nullptr: after block row P is stored, word P is incremented (the head tiles of the update follow the
// substitution block row by block row, see DF_UPH).
// DF_TRW columns per task = DF_TRG column groups of 16 per wave: the operands that do not depend on the column (the rows of
// the factored diagonal block, the 16 x 16 inverses, 1/d) are loaded once for both groups and the two groups' MFMA chains
// interleave — the task is latency-bound (a 16-column task kept a workgroup slot for ~50 us: 18 % of the wide kernel's slot
// time at N = 8192), so twice the columns cost about the same time.
constexpr int DF_TRG = 2, DF_TRW = 16 * DF_TRG;
__device__ __forceinline__ bool df_task_trsm(const DfArgs& a, int j, int c16, double* smem, int tid, bool early, int* sh_ok,
                                             long long t_start, unsigned* rowflags, unsigned row_inc)
{
  double(*Vs)[DF_TRW + 1] = reinterpret_cast<double(*)[DF_TRW + 1]>(smem);   // 256 x 33
  const int lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int I = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K0 = LD_NB * j;
  int64_t col[DF_TRG], colc[DF_TRG];
  bool col_ok[DF_TRG];
#pragma unroll
  for(int gq = 0; gq < DF_TRG; ++gq) {
    col[gq] = (int64_t)c16 + 16 * gq + li;
    col_ok[gq] = col[gq] < a.N;
    colc[gq] = col_ok[gq] ? col[gq] : (int64_t)(a.N - 1);
  }
  double* Vb = a.V + (int64_t)(j % a.nvb) * LD_NB * a.ldv;
  const double* Cd = a.Cd + (int64_t)j * (LD_NB * LD_NB);
  const double* Dk_sp = a.Dblk + (int64_t)(K0 / LD_nb) * (LD_nb * LD_nb);
  const double* Li_sp = a.Li + (int64_t)(K0 / LD_nb) * (4 * LD_SB * LD_SB);
  double4_t t[DF_TRG][4];
#pragma unroll
  for(int gq = 0; gq < DF_TRG; ++gq)
#pragma unroll
    for(int P = 0; P < 4; ++P)
#pragma unroll
      for(int r = 0; r < 4; ++r) {
        const double v = ldg_sc1(a.A + (int64_t)(K0 + 64 * P + 16 * I + g + 4 * r) * a.lda + colc[gq]);
        t[gq][P][r] = col_ok[gq] ? v : 0.0;
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
      if(!df_wait(a.flags, w, sh_ok, t_start, 4, j, c16, P, 0)) return false;
    }
    const double* Dk = Dk_sp + P * (LD_nb * LD_nb);
    const double* Li = Li_sp + P * (4 * LD_SB * LD_SB);
    // the factored diagonal block, its compact tiles and the 16 x 16 inverses are write-once data read after their flag:
    // ordinary loads, issued unconditionally as one batch (a load per (J < I) test was a full round trip each)
    double nl[3][4], iv[4], dsc[4];
#pragma unroll
    for(int r = 0; r < 4; ++r) dsc[r] = ld_batch(a.dinv + K0 + 64 * P + 16 * I + g + 4 * r);
#pragma unroll
    for(int J = 0; J < 3; ++J)
#pragma unroll
      for(int kk = 0; kk < 4; ++kk) nl[J][kk] = -ld_batch(Dk + (16 * J + 4 * kk + g) * LD_nb + 16 * I + li);
#pragma unroll
    for(int kk = 0; kk < 4; ++kk) iv[kk] = ld_batch(Li + I * 256 + li * 16 + 4 * kk + g);
    double4_t u[DF_TRG];
#pragma unroll
    for(int gq = 0; gq < DF_TRG; ++gq) u[gq] = t[gq][P];
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
        for(int kk = 0; kk < 4; ++kk)
#pragma unroll
          for(int gq = 0; gq < DF_TRG; ++gq)
            u[gq] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lop[Jq][kk], Vs[64 * q + 16 * Jq + 4 * kk + g][16 * gq + li], u[gq], 0, 0, 0);
    }
#pragma unroll
    for(int J = 0; J < 4; ++J) {
      if(I == J) {   // wave-uniform
#pragma unroll
        for(int gq = 0; gq < DF_TRG; ++gq) {
          double4_t v = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for(int kk = 0; kk < 4; ++kk) v = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[kk], u[gq][kk], v, 0, 0, 0);
#pragma unroll
          for(int r = 0; r < 4; ++r) {
            const int row = 64 * P + 16 * I + g + 4 * r;
            Vs[row][16 * gq + li] = v[r];
            if(col_ok[gq]) {
              stg_sc1(Vb + (int64_t)row * a.ldv + col[gq], v[r]);
              stg_sc1(a.A + (int64_t)(K0 + row) * a.lda + col[gq], v[r] * dsc[r]);
            }
          }
        }
      }
      __syncthreads();
      if(I > J && J < 3) {
#pragma unroll
        for(int kk = 0; kk < 4; ++kk)
#pragma unroll
          for(int gq = 0; gq < DF_TRG; ++gq)
            u[gq] = __builtin_amdgcn_mfma_f64_16x16x4f64(nl[J][kk], Vs[64 * P + 16 * J + 4 * kk + g][16 * gq + li], u[gq], 0, 0, 0);
      }
    }
    if(rowflags) {   // uniform
      df_drain();
      if(tid == 0) df_add(rowflags + P, row_inc);
    }
  }
  return true;
}

// UP(j, I, J): the 128 x 128 tile (I, J) of the trailing matrix -= V_j[:, rows of I]^T U_j[:, columns of J], K = 256.
// The main loop of ldlt_update_db_kernel (double-buffered LDS stages of 16 k-rows, one barrier per stage, operand reads
// of the next k-step under the MFMAs of the current one); operand and C-tile traffic through sc1 loads / stores.
__device__ __forceinline__ void df_task_tile(const DfArgs& a, int j, int I, int J, double* smem, int tid)
{
  double(*Vs)[UD_KT][UD_LD] = reinterpret_cast<double(*)[UD_KT][UD_LD]>(smem);
  double(*Us)[UD_KT][UD_LD] = reinterpret_cast<double(*)[UD_KT][UD_LD]>(smem + 2 * UD_KT * UD_LD);
  const int N = a.N;
  const int r0 = UD_T * I, c0 = UD_T * J;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lk = lane >> 4, li = lane & 15;
  const double* V = a.V + (int64_t)(j % a.nvb) * LD_NB * a.ldv;
  const int urow0 = LD_NB * j;
  double4_t acc[4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int q = 0; q < 4; ++q) acc[i][q] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int lcol = tid & 127, lrow = tid >> 7;
  const bool vr_ok = (r0 + lcol) < N;
  const bool uc_ok = (c0 + lcol) < N;
  const double* Vp = V + (int64_t)lrow * a.ldv + (vr_ok ? (r0 + lcol) : 0);
  const double* Up = a.A + (int64_t)(urow0 + lrow) * a.lda + (uc_ok ? (c0 + lcol) : 0);
  double vreg[8], ureg[8];
  constexpr int nst = LD_NB / UD_KT;
  auto gload = [&](int st) {
#pragma unroll
    for(int q = 0; q < 8; ++q) {
      const int k = st * UD_KT + 2 * q;
      const double v = ldg_sc1(Vp + (int64_t)k * a.ldv);
      const double u = ldg_sc1(Up + (int64_t)k * a.lda);
      vreg[q] = vr_ok ? v : 0.0;
      ureg[q] = uc_ok ? u : 0.0;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for(int q = 0; q < 8; ++q) {
      Vs[buf][2 * q + lrow][lcol] = vreg[q];
      Us[buf][2 * q + lrow][lcol] = ureg[q];
    }
  };
  gload(0);
  __syncthreads();   // the LDS buffer may still be read by the previous task's waves
  lstore(0);
  gload(1);
  __syncthreads();
  const int arow = wr * 64 + li, bcol = wc * 64 + li;
  for(int st = 0; st < nst; ++st) {
    const int cur = st & 1;
    double av[2][4], bv[2][4];
#pragma unroll
    for(int i = 0; i < 4; ++i) av[0][i] = Vs[cur][lk][arow + 16 * i];
#pragma unroll
    for(int q = 0; q < 4; ++q) bv[0][q] = Us[cur][lk][bcol + 16 * q];
#pragma unroll
    for(int kk = 0; kk < UD_KT / 4; ++kk) {
      const int pb = kk & 1;
      if(kk + 1 < UD_KT / 4) {
#pragma unroll
        for(int i = 0; i < 4; ++i) av[pb ^ 1][i] = Vs[cur][4 * (kk + 1) + lk][arow + 16 * i];
#pragma unroll
        for(int q = 0; q < 4; ++q) bv[pb ^ 1][q] = Us[cur][4 * (kk + 1) + lk][bcol + 16 * q];
      }
      if(kk == 1 && st + 1 < nst) lstore(cur ^ 1);
      if(kk == 2 && st + 2 < nst) gload(st + 2);
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int q = 0; q < 4; ++q) acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb][i], bv[pb][q], acc[i][q], 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: C -= acc on the upper triangle, one 16-row group of the wave at a time (16 loads in flight, then 16 stores;
  // the other workgroup of the CU computes meanwhile)
#pragma unroll
  for(int i = 0; i < 4; ++i) {
    double cv[4][4];
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = r0 + wr * 64 + i * 16 + lk + 4 * reg;
      const double* Crow = a.A + (int64_t)(row < N ? row : (N - 1)) * a.lda;
#pragma unroll
      for(int q = 0; q < 4; ++q) {
        const int col = c0 + wc * 64 + q * 16 + li;
        cv[reg][q] = ldg_sc1(Crow + (col < N ? col : (N - 1)));
      }
    }
#pragma unroll
    for(int reg = 0; reg < 4; ++reg) {
      const int row = r0 + wr * 64 + i * 16 + lk + 4 * reg;
      double* Crow = a.A + (int64_t)row * a.lda;
#pragma unroll
      for(int q = 0; q < 4; ++q) {
        const int col = c0 + wc * 64 + q * 16 + li;
        if(row < N && col < N && col >= row) stg_sc1(Crow + col, cv[reg][q] - acc[i][q][reg]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Trailing-update tile, second form (even N, lda, ldv): the same 128 x 128 x 256 product, with everything around the MFMAs
// cut down —
//   * the C tile is loaded INTO the accumulators before the first stage and U is negated on its way into LDS:
//     acc = C + V^T (-U), so the epilogue is 32 stores per lane instead of four exposed load -> subtract -> store rounds;
//   * every global access is a 16-byte `sc1` buffer operation (8-byte sc1 accesses run at 0.54-0.70x (loads) and 1/2.7
//     (stores) of the 16-byte rate per byte, MI355X_MICROARCH.md): the rows / columns of the wave's 64 x 64 block are dealt
//     to the MFMA tiles so that tile pairs (2h, 2h+1) own ADJACENT rows (A operand, one ds_read_b128 per pair) and adjacent
//     columns (B operand and C: one 16-byte access per pair);
//   * addresses = descriptor (SGPR) + per-lane offset fixed for the whole tile + scalar offset: no 64-bit VALU in the loop.
// element (i, q, reg) of the accumulators  <->  row  wr*64 + 32*(i>>1) + 2*(lk + 4*reg) + (i&1),
//                                               col  wc*64 + 32*(q>>1) + 2*li + (q&1)   of the tile.
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned int df_u32x4 __attribute__((ext_vector_type(4)));
typedef double df_double2 __attribute__((ext_vector_type(2)));
typedef unsigned int df_u32x2 __attribute__((ext_vector_type(2)));
constexpr int DF_SC1 = 16;   // aux bit of the raw buffer builtins = `sc1` on gfx942 / gfx950

// (HIOPAMD_DF_OPAUX: experiment switch — the cache policy of the OPERAND loads of the update tile.  0 = plain loads, which may
//  hit stale lines of this XCD's L2: wrong factors, only good for timing what L2 reuse of the row panels would be worth.)
#ifndef HIOPAMD_DF_OPAUX
#define HIOPAMD_DF_OPAUX DF_SC1
#endif
template <int AUX = DF_SC1>
__device__ __forceinline__ df_double2 df_bload2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff)
{
  return __builtin_bit_cast(df_double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, AUX));
}
__device__ __forceinline__ void df_bstore2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, df_double2 v)
{
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(df_u32x4, v), rs, (int)voff, (int)soff, DF_SC1);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t df_rsrc(const double* base)
{
  // wave-uniform by construction; make that provable (two readfirstlanes per descriptor, not per access)
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0xffffffff, 0x00020000);
}

// Gate: DfNoGate for an ordinary tile.  A head tile (DF_UPH) passes DfRowGate: gate(P) blocks until block row P of both
// operand panels is stored (rows 64 P .. 64 P + 63 of V and U = stages 4 P .. 4 P + 3); it is called just before the first
// loads of such a stage are issued, two stages ahead of their use, with the C tile resident in the accumulators throughout.
struct DfNoGate {
  static constexpr bool active = false;
  __device__ __forceinline__ bool operator()(int) const { return true; }
};
template <class G>
struct DfRowGate {
  static constexpr bool active = true;
  G& g;
  __device__ __forceinline__ bool operator()(int P) const { return g(P); }
};
// Hook: called once per stage, before the stage's operand loads are issued (the selection ahead of ldlt_wide8_body.inc)
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
template <bool FULL, bool PROF, class Gate = DfNoGate, int PANELS = 1, class Hook = DfNoHook>
__device__ __forceinline__ bool df_task_tile2(const DfArgs& a, int j, int I, int J, double* smem, int tid, unsigned (&ph)[12],
                                              Gate gate = Gate(), Hook hook = Hook())
{
  static_assert(PANELS == 1 || !Gate::active, "the gated (head) tile is a single-panel task");
  bool gate_ok = true;
  const int dbg = PROF ? a.dbg : 0;
  const unsigned tp0 = dbg ? (unsigned)wall_clock64() : 0u;
  double(*Vs)[UD_KT][UD_LD] = reinterpret_cast<double(*)[UD_KT][UD_LD]>(smem);
  double(*Us)[UD_KT][UD_LD] = reinterpret_cast<double(*)[UD_KT][UD_LD]>(smem + 2 * UD_KT * UD_LD);
  const int N = a.N;
  const int r0 = UD_T * I, c0 = UD_T * J;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int lk = lane >> 4, li = lane & 15;
  const unsigned lda8 = (unsigned)a.lda * 8u, ldv8 = (unsigned)a.ldv * 8u;
  const __amdgpu_buffer_rsrc_t rsV = df_rsrc(a.V + (int64_t)(j % a.nvb) * LD_NB * a.ldv + r0);
  const __amdgpu_buffer_rsrc_t rsU = df_rsrc(a.A + (int64_t)(LD_NB * j) * a.lda + c0);
  // PANELS == 2: stages 16 .. 31 read the row panel and the factor rows of super-panel j + 1
  const __amdgpu_buffer_rsrc_t rsV2 = df_rsrc(a.V + (int64_t)((j + 1) % a.nvb) * LD_NB * a.ldv + r0);
  const __amdgpu_buffer_rsrc_t rsU2 = df_rsrc(a.A + (int64_t)(LD_NB * (j + 1)) * a.lda + c0);
  const __amdgpu_buffer_rsrc_t rsC = df_rsrc(a.A + (int64_t)r0 * a.lda + c0);
  const int rlim = N - r0, clim = N - c0;   // rows / columns of the tile inside the matrix (FULL: both >= 128)

  // ---- staging: pass p of a stage moves rows 4p + wave, two adjacent columns per lane
  const int col2 = 2 * lane;
  const unsigned vvoff = 8u * (unsigned)(FULL ? col2 : (col2 < rlim - 2 ? col2 : rlim - 2));
  const unsigned uvoff = 8u * (unsigned)(FULL ? col2 : (col2 < clim - 2 ? col2 : clim - 2));
  df_double2 vreg[4], ureg[4];
  auto gload = [&](int st) {
    const bool second = PANELS == 2 && st >= LD_NB / UD_KT;
    const int sl = second ? st - LD_NB / UD_KT : st;
#pragma unroll
    for(int p = 0; p < 4; ++p) {
      const unsigned k = (unsigned)(sl * UD_KT + 4 * p + wave);
      if(PANELS == 2 && second) {
        vreg[p] = df_bload2<HIOPAMD_DF_OPAUX>(rsV2, vvoff, k * ldv8);
        ureg[p] = df_bload2<HIOPAMD_DF_OPAUX>(rsU2, uvoff, k * lda8);
      } else {
        vreg[p] = df_bload2<HIOPAMD_DF_OPAUX>(rsV, vvoff, k * ldv8);
        ureg[p] = df_bload2<HIOPAMD_DF_OPAUX>(rsU, uvoff, k * lda8);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for(int p = 0; p < 4; ++p) {
      *reinterpret_cast<df_double2*>(&Vs[buf][4 * p + wave][col2]) = vreg[p];
      *reinterpret_cast<df_double2*>(&Us[buf][4 * p + wave][col2]) = -ureg[p];
    }
  };
  // ---- C addressing
  const unsigned cvoff_full = 8u * (unsigned)(2 * li) + (unsigned)(2 * lk) * lda8;
  auto crow = [&](int i, int reg) { return wr * 64 + 32 * (i >> 1) + 2 * (lk + 4 * reg) + (i & 1); };
  auto ccol = [&](int h) { return wc * 64 + 32 * h + 2 * li; };
  auto c_off = [&](int i, int reg, int h, unsigned& voff, unsigned& soff) {
    if constexpr(FULL) {
      voff = cvoff_full;
      soff = (unsigned)(wr * 64 + 32 * (i >> 1) + 8 * reg + (i & 1)) * lda8 + 8u * (unsigned)(wc * 64 + 32 * h);
    } else {
      const int R = crow(i, reg), Cc = ccol(h);
      voff = (unsigned)(R < rlim ? R : rlim - 1) * lda8 + 8u * (unsigned)(Cc < clim - 2 ? Cc : clim - 2);
      soff = 0u;
    }
  };

  // progress marker for the debug dump (second state word of the workgroup: 0x80000000 | stage << 16 | J; stage 100 = prologue,
  // 0 .. nst-1 = that stage of the loop, 101 = epilogue): one fire-and-forget store per stage by lane 0
  auto mark = [&](unsigned stg) {
    if(tid == 0) df_st(a.flags + a.off_wg + 2 * (int64_t)blockIdx.x + 1, 0x80000000u | (stg << 16) | ((unsigned)J & 0xffffu));
  };
  mark(100u);
  gload(0);
  double4_t acc[4][4];
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int reg = 0; reg < 4; ++reg)
#pragma unroll
      for(int h = 0; h < 2; ++h) {
        unsigned vo, so;
        c_off(i, reg, h, vo, so);
        const df_double2 c = df_bload2(rsC, vo, so);
        acc[i][2 * h][reg] = c.x;
        acc[i][2 * h + 1][reg] = c.y;
      }
  __syncthreads();   // the LDS buffers may still be read by the previous task's waves
  lstore(0);
  gload(1);
  __syncthreads();
  constexpr int nst = PANELS * (LD_NB / UD_KT);
  const int arow = wr * 64 + 2 * li, bcol = wc * 64 + 2 * li;
  const unsigned tp1 = dbg ? (unsigned)wall_clock64() : 0u;
  for(int st = 0; st < nst; ++st) {
    const int cur = st & 1;
    mark((unsigned)st);
    df_double2 av[2][2], bv[2][2];
#pragma unroll
    for(int h = 0; h < 2; ++h) {
      av[0][h] = *reinterpret_cast<const df_double2*>(&Vs[cur][lk][arow + 32 * h]);
      bv[0][h] = *reinterpret_cast<const df_double2*>(&Us[cur][lk][bcol + 32 * h]);
    }
#pragma unroll
    for(int kk = 0; kk < UD_KT / 4; ++kk) {
      const int pb = kk & 1;
      if(kk + 1 < UD_KT / 4) {
#pragma unroll
        for(int h = 0; h < 2; ++h) {
          av[pb ^ 1][h] = *reinterpret_cast<const df_double2*>(&Vs[cur][4 * (kk + 1) + lk][arow + 32 * h]);
          bv[pb ^ 1][h] = *reinterpret_cast<const df_double2*>(&Us[cur][4 * (kk + 1) + lk][bcol + 32 * h]);
        }
      }
      if constexpr(Hook::active) {
        if(kk == 0) hook(st, nst);   // (what it publishes in the last stage is behind that stage's closing barrier)
      }
      if(kk == 1 && st + 1 < nst) {
        lstore(cur ^ 1);
        if(st + 2 < nst) {
          if constexpr(Gate::active) {
            if(((st + 2) & 3) == 0) gate_ok = gate((st + 2) >> 2) && gate_ok;   // (aborted: the result is discarded anyway)
          }
          gload(st + 2);   // the staging registers are free again: three k-steps + a barrier of lead
        }
      }
#pragma unroll
      for(int i = 0; i < 4; ++i)
#pragma unroll
        for(int q = 0; q < 4; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb][i >> 1][i & 1], bv[pb][q >> 1][q & 1], acc[i][q], 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- epilogue: stores only
  mark(101u);
  if(dbg && (dbg == 1 || j + (PANELS - 1) == dbg - 2)) {   // (a fused task is accounted under the queue it was taken from)
    const unsigned tp2 = (unsigned)wall_clock64();
    ph[9] += tp1 - tp0;    // prologue (first operand stage + C tile in flight, two barriers)
    ph[10] += tp2 - tp1;   // the 16 stages
  }
  const bool diag = (I == J);
#pragma unroll
  for(int i = 0; i < 4; ++i)
#pragma unroll
    for(int reg = 0; reg < 4; ++reg)
#pragma unroll
      for(int h = 0; h < 2; ++h) {
        unsigned vo, so;
        c_off(i, reg, h, vo, so);
        const df_double2 v = df_double2{acc[i][2 * h][reg], acc[i][2 * h + 1][reg]};
        const int R = crow(i, reg), Cc = ccol(h);
        const bool inside = FULL || (R < rlim && Cc < clim);
        if(!diag) {
          if(inside) df_bstore2(rsC, vo, so, v);
        } else if(inside) {
          if(Cc >= R) df_bstore2(rsC, vo, so, v);
          else if(Cc + 1 == R)   // the pair straddles the diagonal: only its second element is in the upper triangle
          {
            const double second = acc[i][2 * h + 1][reg];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(df_u32x2, second), rsC, (int)(vo + 8u), (int)so, DF_SC1);
          }
        }
      }
  return gate_ok;
}

#ifndef HIOPAMD_DF_WIDE_WG_PER_CU
#define HIOPAMD_DF_WIDE_WG_PER_CU 2
#endif
constexpr int DF_WIDE_WG_PER_CU = HIOPAMD_DF_WIDE_WG_PER_CU;   // (register budget of the wide kernel: 256 per lane, so that a CU could hold a second workgroup — see DESIGN.md 3.1 on exactly-filled CUs)
constexpr int DF_TRQ = 40, DF_UPQ = 41, DF_UPQF = 42;   // per super-panel: TR / NEAR update / FAR update tasks of its queues handed out so far

// Two task queues per super-panel instead of one ticket list.  A ticket list must put TR(j+1, .) somewhere inside UP(j, .),
// and wherever it sits the ~500 substitution tasks are handed out in one burst: they all wait for the chain kernel to
// finish C_j+1 while occupying every resident workgroup, and the remaining tiles of UP(j) — ready, but with later tickets
// — do not run (measured at N = 8192: 156 us mean wait per TR task, 41 % of all workgroup time).  Here a workgroup TAKES
// a task only when the conditions that depend on UNTAKEN work or on the chain kernel's panel factorisation already hold:
//   TR(j, .)  : C_j factored (cdone), the first two tile rows of UP(j-1) all taken, UP(j-2) complete (V workspace parity)
//   UP(j, .)  : every TR(j, .) taken (and, by the order the queues are walked, every UP(j-1, .) taken)
// TR first (it feeds the next update), UP otherwise, sleep-poll when neither queue has an eligible head.  Inside a task
// every remaining wait is for a task that is already taken or for the chain kernel, so nothing can deadlock whatever the
// number of resident workgroups (tests/test_ldlt_dataflow_plan.py replays this policy).
// TILE_FORM 1: 8-byte accesses, any N;  2: df_task_tile2 (even N, lda, ldv).  PROF: with the phase accounting of
// HIOPAMD_DF_STAMPS (its counters cost registers: a separate instantiation, launched only when asked for)
struct DfWideShared {
  double smem[4 * UD_KT * UD_LD] __attribute__((aligned(16)));   // 73,728 B: the update's two double-buffered operand tile pairs / the substitution's V
  int4 task;
  int kind, ok;
};
template <int TILE_FORM, bool PROF>
__global__ __launch_bounds__(kBlock, DF_WIDE_WG_PER_CU) void ldlt_wide_kernel(const DfArgs a)
{
  __shared__ __attribute__((aligned(16))) double smem[4 * UD_KT * UD_LD];   // 73,728 B: the update's two double-buffered operand tile pairs / the substitution's V
  __shared__ int sh_kind, sh_ok;
  __shared__ int4 sh_task;
#include "ldlt_wide_body.inc"
}

// Both roles in ONE dispatch (HIOPAMD_DF_ONE=1; measurement aid): workgroups 0 .. DF_ROLES-1 run the chain, the others the
// wide task loop.  Every workgroup carries the chain's LDS (one workgroup per CU), so the wide part runs with ONE workgroup
// per CU instead of two: the schedule is slower, the work and its memory traffic are the same — this is the form the counter
// passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) profile, because rocprofv3 --pmc serialises dispatches and the two-kernel form
// needs both kernels running.  All DF_ROLES + wide workgroups must be resident together: the grid is at most one per CU.
__global__ __launch_bounds__(kBlock, 1) void ldlt_df_one_kernel(const DfArgs a)
{
  __shared__ union U {
    DfChainShared c;
    DfWideShared w;
    __device__ U() {}
  } u;
  if(blockIdx.x < (unsigned)DF_ROLES) {
    DfChainLds& L = u.c.L;
    int4(*sh_tasks)[DF_MAXT] = u.c.tasks;
    int& sh_ok = u.c.ok;
    const int role = blockIdx.x;
#include "ldlt_chain_body.inc"
  } else {
    constexpr int TILE_FORM = 2;
    constexpr bool PROF = false;
    double* smem = u.w.smem;
    int& sh_kind = u.w.kind;
    int& sh_ok = u.w.ok;
    int4& sh_task = u.w.task;
#include "ldlt_wide_body.inc"
  }
}

// Everything a dataflow factorisation needs before its two kernels start, in ONE launch (round 5; before: memset of the info words, the
// pack kernel, a memset of the flags — two fill kernels —, this kernel: five dependent launches, ~45 us of launch gaps per factorisation):
//   blocks 0 .. 255                 row blockIdx.x of the compact copy of diagonal block 0 (ldlt_pack_diag_kernel's work)
//   blocks 256 .. 256 + nzb - 1     zero the flag words (kBlock * 16 per block), with the pre-credits of super-panel 0's tile versions
//                                   (cv = 4) and the two header words (offsets of the state words / of the snapshot area) in place;
//                                   block 256 also zeroes the four info words
__global__ __launch_bounds__(kBlock) void ldlt_df_prep_kernel(const DfArgs a, int64_t nflags, int kbs0)
{
  if(blockIdx.x < (unsigned)LD_NB) {
    const int r = blockIdx.x, c = threadIdx.x;
    a.Cd[r * LD_NB + c] = (r < kbs0 && c < kbs0 && c >= r) ? a.A[(int64_t)r * a.lda + c] : 0.0;
    return;
  }
  const int64_t b = (int64_t)(blockIdx.x - LD_NB);
  if(b == 0 && threadIdx.x < 4) a.info[threadIdx.x] = 0;
  const int64_t cv0 = a.off_chain + DF_CV;
#pragma unroll 4
  for(int q = 0; q < 16; ++q) {
    const int64_t w = (b * 16 + q) * kBlock + threadIdx.x;
    if(w < nflags) {
      unsigned v = 0u;
      if(w >= cv0 && w < cv0 + 16) v = 4u;
      else if(w == 14) v = (unsigned)a.off_wg;
      else if(w == 15) v = (unsigned)a.off_snap;
      a.flags[w] = v;
    }
  }
}
