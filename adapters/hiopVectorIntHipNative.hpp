// hiopVectorInt on libhiopamd.so: int32 index vector in device memory (src/LinAlg/hiopVectorInt.hpp:64-118) — what
// hiopVectorHipNative::copy_from_indexes / copy_from_two_vec_w_pattern / axpy(.., hiopVectorInt) and
// hiopMatrixSparseTriplet::copyRowsFrom receive as their index arguments.  local_data() is a DEVICE pointer (like every
// local_data() of this mem-space); local_data_host() is a host mirror refreshed on demand, as in hiopVectorIntRaja
// (src/LinAlg/hiopVectorIntRaja.hpp).  Created by HipNativeFactory::create_vector_int.
#pragma once
#include "hiopVectorInt.hpp"
#include "hiopVectorIntSeq.hpp"
#include "hiopamd_runtime.hpp"

#include <vector>

namespace hiop
{
class hiopVectorIntHipNative : public hiopVectorInt
{
public:
  explicit hiopVectorIntHipNative(size_type sz);
  virtual ~hiopVectorIntHipNative();

  index_type* local_data() override { return data_; }
  const index_type* local_data_const() const override { return data_; }
  /// host mirror; call copy_to_dev() after writing into it, copy_from_dev() before reading it
  index_type* local_data_host() override { return host_.data(); }
  const index_type* local_data_host_const() const override { return host_.data(); }
  void copy_to_dev();
  void copy_from_dev();

  /// v_local: a DEVICE array of this mem-space (the reference's device classes take device pointers here)
  void copy_from(const index_type* v_local) override;
  void copy_from_vectorseq(const hiopVectorIntSeq& src) override;
  void copy_to_vectorseq(hiopVectorIntSeq& dest) const override;
  void set_to_zero() override;
  void set_to_constant(const index_type c) override;
  void linspace(const index_type& i0, const index_type& di) override;

private:
  hiopamd_ctx* ctx_;
  index_type* data_;
  std::vector<index_type> host_;
};
}  // namespace hiop
