// hiop_amd — MI355X (gfx950) native KKT hot path behind HiOp's LinAlg / linear-solver plug points.
// Common device/host plumbing: the execution context (stream, reduction scratch, pinned result
// slots).  Plays the role of the reference's ExecSpace<MemBackendHip, ExecPolicyHip>
// (reference: src/ExecBackends/ExecSpace.hpp:345-405, MemBackendHipImpl.hpp:73-135) but is a
// runtime object because every kernel here is launched on an explicit HIP stream.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../../include/hiop_amd.h"

namespace hiopamd {

#define HIOPAMD_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if(e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "[hiop_amd] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_),    \
                   __FILE__, __LINE__, #expr);                                                \
      return HIOPAMD_ERR_HIP;                                                                 \
    }                                                                                         \
  } while(0)

#define HIOPAMD_CHECK_ABORT(expr)                                                             \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if(e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "[hiop_amd] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_),    \
                   __FILE__, __LINE__, #expr);                                                \
      std::abort();                                                                           \
    }                                                                                         \
  } while(0)

#ifdef HIOPAMD_POISON_ALLOC
// Test build (scripts/build_poison.sh, never the shipped library): every hipMalloc of the library is filled with 0xFF bytes — quiet
// NaNs as doubles, -1 as ints — so that a kernel that reads memory nobody wrote (0 x uninitialised in a padded tile, a flag word taken
// as initialised) fails in every run instead of only on a box whose fresh HBM happens to hold such a pattern.
inline hipError_t poison_malloc(void** p, size_t bytes)
{
  hipError_t e = hipMalloc(p, bytes);
  if(e == hipSuccess && bytes > 0) e = hipMemset(*p, 0xFF, bytes);
  if(e == hipSuccess) e = hipDeviceSynchronize();
  return e;
}
#define hipMalloc(p, bytes) ::hiopamd::poison_malloc((void**)(p), (bytes))
#endif

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kMaxGrid = 2048;       // 256 CUs x 8 blocks; grid-stride beyond
constexpr int kPartials = kMaxGrid;  // reduction partial slots
constexpr int kHostSlots = 256;      // doubles of pinned result space = 64 reduction results of up to 4 doubles

inline int grid_for(int64_t n, int per_thread = 1)
{
  int64_t g = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  if(g < 1) g = 1;
  if(g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

}  // namespace hiopamd

// The context is an opaque C struct at the ABI; defined here for the implementation files.
struct hiopamd_ctx {
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double* d_partials = nullptr;   // kPartials * 4 doubles of reduction scratch (device)
  double* d_result = nullptr;     // kHostSlots doubles (device) -- device-resident reduction results
  double* h_result = nullptr;     // kHostSlots doubles, pinned + device-mapped
  double* h_result_dev = nullptr; // device view of h_result
  int* d_iresult = nullptr;       // integer scratch (device)
  void* d_work = nullptr;         // grow-only general workspace
  size_t work_bytes = 0;
  // optional all-reduce hook for column-sharded (distributed) objects
  hiopamd_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int comm_rank = 0;
  int comm_size = 1;
  // collective statistics (hiopamd_ctx_collective_stats_*): every call of the hook goes through ctx_allreduce below
  long long coll_count = 0;
  bool coll_timed = false;
  std::vector<hipEvent_t> coll_ev;   // start / stop pairs on the context's stream (timing mode only)
  size_t coll_used = 0;
  // CU-masked streams + events for intra-operation concurrency (the LDL^T look-ahead); created lazily, always joined back
  // into `stream` before the operation returns
  hipStream_t diag_stream = nullptr;   // CU-masked: the reserved CUs (serial chain; its 1-workgroup kernel needs a whole CU's LDS)
  hipStream_t upd_stream = nullptr;    // CU-masked: every other CU
  int cu_split_state = 0;              // 0 not tried, 1 masked streams available, -1 unavailable
  int chain_cus = 0, wide_cus = 0;     // CUs behind diag_stream / upd_stream (from hipDeviceProp_t::multiProcessorCount and the reservation)
  hipEvent_t ev_pool[160] = {nullptr};
  hipEvent_t ev_info = nullptr;      // behind the read-back of a factorisation's info words (the host waits for THIS, not for the whole stream)
  hipEvent_t ev_pending = nullptr;   // behind the last deferred reduction (reduce_flush waits for this)
  int n_events = 0;
  void* spans = nullptr;               // hiopamd::SpanState (context.hip): KKT / linear-solver run-stats spans
  // deferred reductions (hiopamd_ctx_reduce_begin / _end): the reductions launched inside the bracket write their results to consecutive
  // pinned slots and are finished on the host — sqrt, Kahan fold, the copy to the caller's variable — after ONE synchronisation
  int defer_depth = 0;
  int n_pending = 0;
  std::vector<std::function<void()>> pending;
};

namespace hiopamd {
// Reserve CUs 0-1 of every XCD for `diag_stream` and give `upd_stream` the other 240 CUs.  Measured on MI355X
// (scripts/probes/cu_mask_probe.hip): mask bit i maps to XCD i%8, CU i/8; the workgroups of a masked queue are dealt
// round-robin over the XCDs, so every XCD must keep at least one enabled CU; a 160 KB-LDS workgroup on the reserved
// CUs starts within 8 us while the rest of the device is saturated, whereas on an unmasked high-priority stream it
// waits for the saturating kernel's tail (the dispatcher back-fills partially free CUs with the low-priority grid).
inline bool ctx_cu_split(hiopamd_ctx* ctx)
{
  if(ctx->cu_split_state != 0) return ctx->cu_split_state > 0;
  ctx->cu_split_state = -1;
  if(std::getenv("HIOPAMD_NO_CU_SPLIT")) return false;
  hipDeviceProp_t prop;
  int dev = 0;
  if(hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  const int ncu = prop.multiProcessorCount;
  if(ncu != 256) {   // the mapping above was verified for the 8 x 32 CU layout only
    std::fprintf(stderr, "[hiop_amd] device reports %d CUs (not 256): no CU-masked streams, the LDL^T runs its stepwise kernels instead of the dataflow pair\n", ncu);
    return false;
  }
  // HIOPAMD_SD_CUS = reserved CUs per XCD for the chain stream (default 2; bits 0..8k-1 of the mask = CUs 0..k-1 of every XCD).
  // Measured at N = 8192: 1 -> 9.53 ms per step, 2 -> 9.38 (the head substitution's 16 four-wave workgroups and the 10 tiles of
  // the diagonal-block update get a CU each), 3 -> 9.42; the update's time does not move with 8 or 16 CUs fewer.
  int per_xcd = std::getenv("HIOPAMD_SD_CUS") ? std::atoi(std::getenv("HIOPAMD_SD_CUS")) : 2;
  if(per_xcd < 1 || per_xcd > 3) per_xcd = 1;
  const uint32_t lowbits = (per_xcd == 1) ? 0xffu : (per_xcd == 2) ? 0xffffu : 0xffffffu;
  uint32_t m[8] = {lowbits, 0, 0, 0, 0, 0, 0, 0};
  uint32_t mc[8] = {~lowbits, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  if(hipExtStreamCreateWithCUMask(&ctx->diag_stream, 8, m) != hipSuccess) {
    ctx->diag_stream = nullptr;
    (void)hipGetLastError();
    return false;
  }
  if(hipExtStreamCreateWithCUMask(&ctx->upd_stream, 8, mc) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamDestroy(ctx->diag_stream);
    ctx->diag_stream = ctx->upd_stream = nullptr;
    return false;
  }
  ctx->chain_cus = 8 * per_xcd;
  ctx->wide_cus = ncu - ctx->chain_cus;
  ctx->cu_split_state = 1;
  return true;
}
// Run-stats spans = the reference's hiopRunStatsKKT / hiopLinSolStats timers (src/Utils/hiopRunStats.hpp:82-140, :244-300):
// every span is a roctx range (visible to `rocprofv3 --marker-trace`) and, when enabled with hiopamd_ctx_spans_enable, a
// pair of HIP events on the context's stream — no host synchronisation on the hot path; the elapsed times are summed
// when hiopamd_ctx_spans_read is called.  Ids = HIOPAMD_SPAN_* of include/hiop_amd.h.
void span_begin(hiopamd_ctx* ctx, int id);
void span_end(hiopamd_ctx* ctx, int id);
struct SpanScope {
  hiopamd_ctx* ctx;
  int id;
  SpanScope(hiopamd_ctx* c, int i) : ctx(c), id(i) { span_begin(ctx, id); }
  ~SpanScope() { span_end(ctx, id); }
  SpanScope(const SpanScope&) = delete;
  SpanScope& operator=(const SpanScope&) = delete;
};

inline hipEvent_t ctx_named_event(hipEvent_t& e)
{
  if(!e) HIOPAMD_CHECK_ABORT(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}
inline hipEvent_t ctx_event(hiopamd_ctx* ctx, int i)
{
  while(ctx->n_events <= i) {
    HIOPAMD_CHECK_ABORT(hipEventCreateWithFlags(&ctx->ev_pool[ctx->n_events], hipEventDisableTiming));
    ctx->n_events++;
  }
  return ctx->ev_pool[i];
}
// THE call of the all-reduce hook (RCCL, or whatever hiopamd_ctx_set_allreduce installed): counted, and in timing mode bracketed by a
// pair of HIP events on the context's stream.  0 on success.  Callers test ctx->allreduce themselves where "no hook" changes the algorithm.
inline int ctx_allreduce(hiopamd_ctx* ctx, double* buf, size_t count, int op)
{
  if(!ctx->allreduce) return 0;
  ctx->coll_count += 1;
  const bool timed = ctx->coll_timed;
  if(timed) {
    while(ctx->coll_ev.size() < ctx->coll_used + 2) {
      hipEvent_t e = nullptr;
      if(hipEventCreate(&e) != hipSuccess) return -1;
      ctx->coll_ev.push_back(e);
    }
    (void)hipEventRecord(ctx->coll_ev[ctx->coll_used], ctx->stream);
  }
  const int rc = ctx->allreduce(ctx->allreduce_user, buf, count, op, (void*)ctx->stream);
  if(timed) {
    (void)hipEventRecord(ctx->coll_ev[ctx->coll_used + 1], ctx->stream);
    ctx->coll_used += 2;
  }
  return rc;
}
// grow-only workspace (never shrinks; freed with the context)
inline void* ctx_workspace(hiopamd_ctx* ctx, size_t bytes)
{
  if(bytes <= ctx->work_bytes) return ctx->d_work;
  if(ctx->d_work) {
    HIOPAMD_CHECK_ABORT(hipStreamSynchronize(ctx->stream));
    HIOPAMD_CHECK_ABORT(hipFree(ctx->d_work));
  }
  size_t nb = bytes + bytes / 4 + 4096;
  HIOPAMD_CHECK_ABORT(hipMalloc(&ctx->d_work, nb));
  ctx->work_bytes = nb;
  return ctx->d_work;
}
// the deferred reductions of a context: one synchronisation, then every pending result is finished in launch order
static inline int reduce_flush(hiopamd_ctx* ctx)
{
  if(ctx->n_pending == 0) return HIOPAMD_OK;
  // wait for the LAST deferred reduction, not for whatever was queued behind it (a factorisation's epilogue kernels, for one): every
  // deferred launch re-records this event (device_utils.hpp, launch_reduce_fin)
  if(ctx->ev_pending) HIOPAMD_CHECK(hipEventSynchronize(ctx->ev_pending));
  else HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  for(auto& f : ctx->pending) f();
  ctx->pending.clear();
  ctx->n_pending = 0;
  return HIOPAMD_OK;
}

// Library code that calls the scalar-returning vector entry points and goes on with the values must not be caught by a caller's
// bracket.  ReduceNow: immediate mode for a scope.  ReduceBatch: a bracket of the library's own — `flush()` makes every pending
// result (the caller's included) valid; results still pending when the scope is left (an error return) are dropped, never written.
struct ReduceNow {
  hiopamd_ctx* c;
  int saved;
  explicit ReduceNow(hiopamd_ctx* ctx) : c(ctx), saved(ctx->defer_depth) { c->defer_depth = 0; }
  ~ReduceNow() { c->defer_depth = saved; }
  ReduceNow(const ReduceNow&) = delete;
  ReduceNow& operator=(const ReduceNow&) = delete;
};
struct ReduceBatch {
  hiopamd_ctx* c;
  int saved, base;
  explicit ReduceBatch(hiopamd_ctx* ctx) : c(ctx), saved(ctx->defer_depth), base(ctx->n_pending) { c->defer_depth = saved + 1; }
  int flush() { return reduce_flush(c); }
  ~ReduceBatch()
  {
    if(c->n_pending > base) {   // left without flush(): the locals those results were meant for are gone
      (void)hipStreamSynchronize(c->stream);
      c->pending.resize((size_t)base);
      c->n_pending = base;
    }
    c->defer_depth = saved;
  }
  ReduceBatch(const ReduceBatch&) = delete;
  ReduceBatch& operator=(const ReduceBatch&) = delete;
};

}  // namespace hiopamd
