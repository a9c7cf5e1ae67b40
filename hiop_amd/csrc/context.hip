// Execution context, memory entry points and the RCCL all-reduce hook.
//
// reference plug point: src/ExecBackends/ExecSpace.hpp:345-457 (alloc_array / dealloc_array / copy
// forwarded to AllocImpl / DeAllocImpl / TransferImpl) and the MPI_Allreduce call sites of the
// column-distributed objects.  Here the collective is RCCL over xGMI: one
// communicator per context, one rank per GPU, reductions issued on the context's stream so that
// they are ordered with the kernels that produce / consume the (device-resident) small blocks.
#include "common.hpp"

#include <rccl/rccl.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <cstring>
#include <vector>

namespace {
struct RcclState {
  ncclComm_t comm = nullptr;
};

int rccl_allreduce(void* user, double* buf, size_t count, int op, void* stream)
{
  RcclState* st = static_cast<RcclState*>(user);
  ncclRedOp_t rop = ncclSum;
  if(op == HIOPAMD_MIN) rop = ncclMin;
  if(op == HIOPAMD_MAX) rop = ncclMax;
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclDouble, rop, st->comm, (hipStream_t)stream);
  return r == ncclSuccess ? 0 : -1;
}
}  // namespace

namespace hiopamd {
static const char* kSpanNames[HIOPAMD_SPAN_COUNT] = {
    "kkt.tmUpdateInit",    "kkt.tmUpdateLinsys",  "kkt.tmUpdateInnerFact", "kkt.tmSolveRhsManip",
    "kkt.tmSolveInner",    "linsolv.tmFactTime",  "linsolv.tmInertiaComp", "linsolv.tmTriuSolves"};
struct SpanState {
  bool enabled = false;
  std::vector<hipEvent_t> pool;            // timing events, reused
  size_t used = 0;
  struct Rec {
    hipEvent_t b, e;
    int id;
    bool closed;
  };
  std::vector<Rec> recs;
  std::vector<size_t> open[HIOPAMD_SPAN_COUNT];
  double ms[HIOPAMD_SPAN_COUNT] = {0};
  int64_t cnt[HIOPAMD_SPAN_COUNT] = {0};
  hipEvent_t get()
  {
    if(used == pool.size()) {
      hipEvent_t e = nullptr;
      if(hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
  // fold the finished records into the sums (the caller has synchronised the stream)
  void collect()
  {
    for(const Rec& r : recs) {
      if(!r.closed || !r.b || !r.e) continue;
      float t = 0.f;
      if(hipEventElapsedTime(&t, r.b, r.e) == hipSuccess) {
        ms[r.id] += (double)t;
        cnt[r.id] += 1;
      }
    }
    recs.clear();
    for(auto& o : open) o.clear();
    used = 0;
  }
};
void span_begin(hiopamd_ctx* ctx, int id)
{
  (void)roctxRangePushA(kSpanNames[id]);
  SpanState* st = static_cast<SpanState*>(ctx->spans);
  if(!st || !st->enabled) return;
  if(st->recs.size() >= 8192) {   // bounded memory between reads: fold what is there (one stream sync every 8192 spans)
    (void)hipStreamSynchronize(ctx->stream);
    st->collect();
  }
  SpanState::Rec r{st->get(), nullptr, id, false};
  if(r.b) (void)hipEventRecord(r.b, ctx->stream);
  st->open[id].push_back(st->recs.size());
  st->recs.push_back(r);
}
void span_end(hiopamd_ctx* ctx, int id)
{
  (void)roctxRangePop();
  SpanState* st = static_cast<SpanState*>(ctx->spans);
  if(!st || !st->enabled || st->open[id].empty()) return;
  SpanState::Rec& r = st->recs[st->open[id].back()];
  st->open[id].pop_back();
  r.e = st->get();
  if(r.e) (void)hipEventRecord(r.e, ctx->stream);
  r.closed = true;
}
}  // namespace hiopamd

extern "C" {

const char* hiopamd_span_name(int id) { return (id >= 0 && id < HIOPAMD_SPAN_COUNT) ? hiopamd::kSpanNames[id] : nullptr; }

int hiopamd_ctx_spans_enable(hiopamd_ctx* c, int enable)
{
  if(!c) return HIOPAMD_ERR_ARG;
  hiopamd::SpanState* st = static_cast<hiopamd::SpanState*>(c->spans);
  if(!st) {
    st = new hiopamd::SpanState();
    c->spans = st;
  }
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  st->collect();
  for(int i = 0; i < HIOPAMD_SPAN_COUNT; ++i) {
    st->ms[i] = 0.0;
    st->cnt[i] = 0;
  }
  st->enabled = enable != 0;
  return HIOPAMD_OK;
}

int hiopamd_ctx_spans_read(hiopamd_ctx* c, double* ms_host, int64_t* count_host)
{
  if(!c) return HIOPAMD_ERR_ARG;
  hiopamd::SpanState* st = static_cast<hiopamd::SpanState*>(c->spans);
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  if(st) st->collect();
  for(int i = 0; i < HIOPAMD_SPAN_COUNT; ++i) {
    if(ms_host) ms_host[i] = st ? st->ms[i] : 0.0;
    if(count_host) count_host[i] = st ? st->cnt[i] : 0;
  }
  return HIOPAMD_OK;
}

const char* hiopamd_version(void) { return "hiop_amd 0.1.0 (gfx950)"; }

int hiopamd_device_info(char* name_host, size_t name_len, int* cu_count_host, size_t* hbm_bytes_host)
{
  int dev = 0;
  HIOPAMD_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t p;
  HIOPAMD_CHECK(hipGetDeviceProperties(&p, dev));
  if(name_host && name_len) {
    std::snprintf(name_host, name_len, "%s (%s)", p.name, p.gcnArchName);
  }
  if(cu_count_host) *cu_count_host = p.multiProcessorCount;
  if(hbm_bytes_host) *hbm_bytes_host = p.totalGlobalMem;
  return HIOPAMD_OK;
}

int hiopamd_ctx_create(hiopamd_ctx** out, void* hip_stream)
{
  if(!out) return HIOPAMD_ERR_ARG;
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "[hiop_amd] no HIP device visible: this library has no CPU path\n");
    return HIOPAMD_ERR_NODEVICE;
  }
  *out = nullptr;
  hiopamd_ctx* c = new hiopamd_ctx();
  bool ok = true;
  if(hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    c->own_stream = ok;
  }
  ok = ok && hipMalloc(&c->d_partials, sizeof(double) * 4 * hiopamd::kPartials) == hipSuccess;
  ok = ok && hipMalloc(&c->d_result, sizeof(double) * hiopamd::kHostSlots) == hipSuccess;
  ok = ok && hipMalloc(&c->d_iresult, 256) == hipSuccess;
  ok = ok && hipHostMalloc(&c->h_result, sizeof(double) * hiopamd::kHostSlots, hipHostMallocMapped) == hipSuccess;
  ok = ok && hipHostGetDevicePointer((void**)&c->h_result_dev, c->h_result, 0) == hipSuccess;
  if(!ok) {   // nothing leaks: the destroy tolerates the members that were never allocated
    std::fprintf(stderr, "[hiop_amd] hiopamd_ctx_create: HIP allocation failed (%s)\n", hipGetErrorName(hipGetLastError()));
    if(c->stream && c->own_stream) hiopamd_ctx_destroy(c);
    else {
      c->stream = nullptr;
      (void)hipFree(c->d_partials);
      (void)hipFree(c->d_result);
      (void)hipFree(c->d_iresult);
      if(c->h_result) (void)hipHostFree(c->h_result);
      delete c;
    }
    return HIOPAMD_ERR_HIP;
  }
  std::memset(c->h_result, 0, sizeof(double) * hiopamd::kHostSlots);
  *out = c;
  return HIOPAMD_OK;
}

int hiopamd_ctx_destroy(hiopamd_ctx* c)
{
  if(!c) return HIOPAMD_OK;
  hipStreamSynchronize(c->stream);
  if(c->allreduce == rccl_allreduce && c->allreduce_user) {
    RcclState* st = static_cast<RcclState*>(c->allreduce_user);
    if(st->comm) ncclCommDestroy(st->comm);
    delete st;
  }
  hipFree(c->d_partials);
  hipFree(c->d_result);
  hipFree(c->d_iresult);
  hipHostFree(c->h_result);
  if(c->d_work) hipFree(c->d_work);
  if(c->diag_stream) {
    hipStreamSynchronize(c->diag_stream);
    hipStreamDestroy(c->diag_stream);
  }
  if(c->upd_stream) {
    hipStreamSynchronize(c->upd_stream);
    hipStreamDestroy(c->upd_stream);
  }
  for(int i = 0; i < c->n_events; ++i) hipEventDestroy(c->ev_pool[i]);
  if(c->ev_info) hipEventDestroy(c->ev_info);
  if(c->ev_pending) hipEventDestroy(c->ev_pending);
  for(hipEvent_t e : c->coll_ev) hipEventDestroy(e);
  if(c->spans) {
    hiopamd::SpanState* st = static_cast<hiopamd::SpanState*>(c->spans);
    for(hipEvent_t e : st->pool) (void)hipEventDestroy(e);
    delete st;
  }
  if(c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return HIOPAMD_OK;
}

int hiopamd_ctx_sync(hiopamd_ctx* c)
{
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  return HIOPAMD_OK;
}

void* hiopamd_ctx_stream(hiopamd_ctx* c) { return (void*)c->stream; }

// Deferred reductions (csrc/device_utils.hpp::launch_reduce_fin): between _begin and _end the scalar-returning vector entry points
// (dot, norms, sum, min, log-barrier, linear damping, fraction-to-the-boundary) do not synchronise; their host results are written by
// _end after ONE synchronisation of the stream.  Brackets nest; the outermost _end flushes.
int hiopamd_ctx_reduce_begin(hiopamd_ctx* c)
{
  if(!c) return HIOPAMD_ERR_ARG;
  c->defer_depth += 1;
  return HIOPAMD_OK;
}
int hiopamd_ctx_reduce_end(hiopamd_ctx* c)
{
  if(!c || c->defer_depth <= 0) return HIOPAMD_ERR_STATE;
  c->defer_depth -= 1;
  if(c->defer_depth > 0 || c->n_pending == 0) return HIOPAMD_OK;
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  for(auto& f : c->pending) f();
  c->pending.clear();
  c->n_pending = 0;
  return HIOPAMD_OK;
}

int hiopamd_ctx_comm(const hiopamd_ctx* c, int* rank_host, int* size_host)
{
  if(!c || !rank_host || !size_host) return HIOPAMD_ERR_ARG;
  *rank_host = c->comm_rank;
  *size_host = c->comm_size > 0 ? c->comm_size : 1;
  return HIOPAMD_OK;
}

int hiopamd_ctx_set_allreduce(hiopamd_ctx* c, hiopamd_allreduce_fn fn, void* user, int rank, int size)
{
  if(!c || size < 1 || rank < 0 || rank >= size) return HIOPAMD_ERR_ARG;
  c->allreduce = fn;
  c->allreduce_user = user;
  c->comm_rank = rank;
  c->comm_size = size;
  return HIOPAMD_OK;
}

int hiopamd_ctx_collective_stats_begin(hiopamd_ctx* c, int timed)
{
  if(!c) return HIOPAMD_ERR_ARG;
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  c->coll_count = 0;
  c->coll_used = 0;
  c->coll_timed = timed != 0;
  return HIOPAMD_OK;
}

int hiopamd_ctx_collective_stats_read(hiopamd_ctx* c, int64_t* count_host, double* ms_host)
{
  if(!c) return HIOPAMD_ERR_ARG;
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  double ms = 0.0;
  for(size_t i = 0; i + 1 < c->coll_used; i += 2) {
    float t = 0.f;
    if(hipEventElapsedTime(&t, c->coll_ev[i], c->coll_ev[i + 1]) == hipSuccess) ms += t;
  }
  if(count_host) *count_host = (int64_t)c->coll_count;
  if(ms_host) *ms_host = c->coll_timed ? ms : -1.0;
  c->coll_used = 0;
  c->coll_timed = false;
  return HIOPAMD_OK;
}

int hiopamd_ctx_rccl_ranks(const hiopamd_ctx* c, int* ranks_host)
{
  if(!c || !ranks_host) return HIOPAMD_ERR_ARG;
  *ranks_host = 0;
  if(c->allreduce == rccl_allreduce && c->allreduce_user) {
    const RcclState* st = static_cast<const RcclState*>(c->allreduce_user);
    int n = 0;
    if(st->comm && ncclCommCount(st->comm, &n) == ncclSuccess) *ranks_host = n;
  }
  return HIOPAMD_OK;
}

int hiopamd_rccl_unique_id(unsigned char* unique_id_128_host)
{
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  if(ncclGetUniqueId(&id) != ncclSuccess) return HIOPAMD_ERR_HIP;
  std::memcpy(unique_id_128_host, &id, sizeof(id));
  return HIOPAMD_OK;
}

int hiopamd_ctx_init_rccl(hiopamd_ctx* c, const unsigned char* unique_id_128_host, int rank, int size)
{
  if(!c || !unique_id_128_host || size < 1 || rank < 0 || rank >= size) return HIOPAMD_ERR_ARG;
  ncclUniqueId id;
  std::memcpy(&id, unique_id_128_host, sizeof(id));
  RcclState* st = new RcclState();
  if(ncclCommInitRank(&st->comm, size, id, rank) != ncclSuccess) {
    delete st;
    return HIOPAMD_ERR_HIP;
  }
  return hiopamd_ctx_set_allreduce(c, rccl_allreduce, st, rank, size);
}

int hiopamd_alloc(void** dptr, size_t bytes)
{
  if(!dptr) return HIOPAMD_ERR_ARG;
  *dptr = nullptr;
  if(bytes == 0) return HIOPAMD_OK;
  HIOPAMD_CHECK(hipMalloc(dptr, bytes));
  return HIOPAMD_OK;
}
int hiopamd_free(void* dptr)
{
  if(dptr) HIOPAMD_CHECK(hipFree(dptr));
  return HIOPAMD_OK;
}
int hiopamd_copy_h2d(hiopamd_ctx* c, void* dst, const void* src_host, size_t bytes)
{
  if(bytes == 0) return HIOPAMD_OK;
  HIOPAMD_CHECK(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, c->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  return HIOPAMD_OK;
}
int hiopamd_copy_d2h(hiopamd_ctx* c, void* dst_host, const void* src, size_t bytes)
{
  if(bytes == 0) return HIOPAMD_OK;
  HIOPAMD_CHECK(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(c->stream));
  return HIOPAMD_OK;
}
int hiopamd_copy_d2d(hiopamd_ctx* c, void* dst, const void* src, size_t bytes)
{
  if(bytes == 0) return HIOPAMD_OK;
  HIOPAMD_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return HIOPAMD_OK;
}

}  // extern "C"
