// hiop_amd adapters — HiOp-side glue for the `hip-native` LinAlg back-end (compile-checked against the reference headers by
// adapters/check_adapters.sh; nothing from the reference is copied or shipped).
//
// One process-wide execution context (HIP stream + reduction scratch of libhiopamd.so); every adapter object forwards to
// the C ABI of include/hiop_amd.h on it.  Plays the role ExecSpace<MemBackendHip, ExecPolicyHip> plays for hiopVectorHip
// (src/ExecBackends/ExecSpace.hpp:345-457).
#pragma once
#include "hiop_amd.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>

namespace hiop
{
inline hiopamd_ctx* hiopamd_default_ctx()
{
  static hiopamd_ctx* ctx = nullptr;
  if(!ctx) {
    if(hiopamd_ctx_create(&ctx, nullptr) != HIOPAMD_OK) {
      std::fprintf(stderr, "hiop_amd: no gfx950 device / context creation failed (the hip-native back-end has no CPU path)\n");
      std::abort();
    }
  }
  return ctx;
}
/// status -> assert, like the reference's back-ends treat a failed device operation
inline void hiopamd_ok(int rc)
{
  (void)rc;
  assert(rc == HIOPAMD_OK && "libhiopamd call failed");
}
inline double* hiopamd_new_array(size_t n)
{
  void* p = nullptr;
  hiopamd_ok(hiopamd_alloc(&p, sizeof(double) * (n ? n : 1)));
  return static_cast<double*>(p);
}
inline int* hiopamd_new_int_array(size_t n)
{
  void* p = nullptr;
  hiopamd_ok(hiopamd_alloc(&p, sizeof(int) * (n ? n : 1)));
  return static_cast<int*>(p);
}
}  // namespace hiop
