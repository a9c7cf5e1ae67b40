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

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kMaxGrid = 2048;       // 256 CUs x 8 blocks; grid-stride beyond
constexpr int kPartials = kMaxGrid;  // reduction partial slots
constexpr int kHostSlots = 64;

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
};

namespace hiopamd {
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
}  // namespace hiopamd
