#include "hiopVectorIntHipNative.hpp"

namespace hiop
{
hiopVectorIntHipNative::hiopVectorIntHipNative(size_type sz)
    : hiopVectorInt(sz), ctx_(hiopamd_default_ctx()), data_(hiopamd_new_int_array((size_t)sz)), host_((size_t)sz, 0)
{
}
hiopVectorIntHipNative::~hiopVectorIntHipNative()
{
  hiopamd_ctx_sync(ctx_);
  hiopamd_free(data_);
}
void hiopVectorIntHipNative::copy_to_dev() { hiopamd_ok(hiopamd_copy_h2d(ctx_, data_, host_.data(), sizeof(index_type) * (size_t)sz_)); }
void hiopVectorIntHipNative::copy_from_dev() { hiopamd_ok(hiopamd_copy_d2h(ctx_, host_.data(), data_, sizeof(index_type) * (size_t)sz_)); }
void hiopVectorIntHipNative::copy_from(const index_type* v_local) { hiopamd_ok(hiopamd_ivec_copy(ctx_, sz_, data_, v_local)); }
void hiopVectorIntHipNative::copy_from_vectorseq(const hiopVectorIntSeq& src)
{
  assert(src.get_local_size() == sz_);
  hiopamd_ok(hiopamd_copy_h2d(ctx_, data_, src.local_data_const(), sizeof(index_type) * (size_t)sz_));
}
void hiopVectorIntHipNative::copy_to_vectorseq(hiopVectorIntSeq& dest) const
{
  assert(dest.get_local_size() == sz_);
  hiopamd_ok(hiopamd_copy_d2h(ctx_, dest.local_data(), data_, sizeof(index_type) * (size_t)sz_));
}
void hiopVectorIntHipNative::set_to_zero() { hiopamd_ok(hiopamd_ivec_set_to_constant(ctx_, sz_, data_, 0)); }
void hiopVectorIntHipNative::set_to_constant(const index_type c) { hiopamd_ok(hiopamd_ivec_set_to_constant(ctx_, sz_, data_, c)); }
void hiopVectorIntHipNative::linspace(const index_type& i0, const index_type& di) { hiopamd_ok(hiopamd_ivec_linspace(ctx_, sz_, data_, i0, di)); }
}  // namespace hiop
