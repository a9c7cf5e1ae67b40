#include "LinAlgFactoryHipNative.hpp"

#include "hiopMatrixDenseHipNative.hpp"
#include "hiopMatrixSparseTripletHipNative.hpp"
#include "hiopVectorHipNative.hpp"
#include "hiopVectorIntHipNative.hpp"
#include "hiopamd_runtime.hpp"

#include <algorithm>
#include <cctype>

namespace hiop {

static std::string upper(std::string s)
{
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::toupper(c); });
  return s;
}

bool HipNativeFactory::handles(const std::string& mem_space) { return upper(mem_space) == "HIP-NATIVE"; }
bool HipNativeFactory::handles(const ExecSpaceInfo& hi) { return handles(hi.mem_space_); }

hiopVector* HipNativeFactory::create_vector(const ExecSpaceInfo& hi, const size_type& glob_n, index_type* col_part, MPI_Comm comm)
{
  return handles(hi) ? new hiopVectorHipNative(glob_n, col_part, comm) : nullptr;
}

hiopVectorInt* HipNativeFactory::create_vector_int(const ExecSpaceInfo& hi, size_type n)
{
  return handles(hi) ? new hiopVectorIntHipNative(n) : nullptr;
}

hiopMatrixDense* HipNativeFactory::create_matrix_dense(const ExecSpaceInfo& hi, const size_type& m, const size_type& glob_n,
                                                       index_type* col_part, MPI_Comm comm, const size_type& m_max_alloc)
{
  return handles(hi) ? new hiopMatrixDenseHipNative(m, glob_n, col_part, comm, m_max_alloc) : nullptr;
}

hiopMatrixSparse* HipNativeFactory::create_matrix_sparse(const ExecSpaceInfo& hi, size_type rows, size_type cols, size_type nnz)
{
  return handles(hi) ? new hiopMatrixSparseTripletHipNative((int)rows, (int)cols, (int)nnz) : nullptr;
}

hiopMatrixSparse* HipNativeFactory::create_matrix_sym_sparse(const ExecSpaceInfo& hi, size_type size, size_type nnz)
{
  return handles(hi) ? new hiopMatrixSymSparseTripletHipNative((int)size, (int)nnz) : nullptr;
}

double* HipNativeFactory::create_raw_array(const std::string& mem_space, size_type n)
{
  return handles(mem_space) ? hiopamd_new_array((size_t)n) : nullptr;
}

bool HipNativeFactory::delete_raw_array(const std::string& mem_space, double* a)
{
  if(!handles(mem_space)) return false;
  hiopamd_ok(hiopamd_free(a));
  return true;
}

}  // namespace hiop
