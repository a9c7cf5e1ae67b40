// Factory hook of the `hip-native` back-end: what LinearAlgebraFactory (src/LinAlg/LinAlgFactory.{hpp,cpp}) calls first in
// create_vector / create_vector_int / create_matrix_dense / create_matrix_sparse / create_matrix_sym_sparse / create_raw_array / delete_raw_array.
// Every function returns nullptr (false) when the execution space is not "HIP-NATIVE", so the hook is one line per factory
// method:   if(auto* p = HipNativeFactory::create_vector(hi, glob_n, col_part, comm)) return p;
// The signatures are the factory's own (ExecSpaceInfo of src/ExecBackends/ExecSpace.hpp:75-108; its constructor needs one more
// branch: mem_space_ == "HIP-NATIVE" -> mem_backend_ = "HIP", mem_backend_host_ = "STDCPP", exec_backend_ = "HIP").
// Compile-checked against the reference headers by adapters/check_adapters.sh.
#pragma once
#include "ExecSpace.hpp"
#include "hiopVector.hpp"
#include "hiopMatrixDense.hpp"
#include "hiopMatrixSparse.hpp"
#include "hiopVectorInt.hpp"

namespace hiop {

struct HipNativeFactory {
  static bool handles(const ExecSpaceInfo& hi);
  static bool handles(const std::string& mem_space);
  static hiopVector* create_vector(const ExecSpaceInfo& hi, const size_type& glob_n, index_type* col_part = nullptr,
                                   MPI_Comm comm = MPI_COMM_SELF);
  /// src/LinAlg/LinAlgFactory.cpp:182: device int32 index vector of the same mem-space
  static hiopVectorInt* create_vector_int(const ExecSpaceInfo& hi, size_type n);
  static hiopMatrixDense* create_matrix_dense(const ExecSpaceInfo& hi, const size_type& m, const size_type& glob_n,
                                              index_type* col_part = nullptr, MPI_Comm comm = MPI_COMM_SELF,
                                              const size_type& m_max_alloc = -1);
  static hiopMatrixSparse* create_matrix_sparse(const ExecSpaceInfo& hi, size_type rows, size_type cols, size_type nnz);
  static hiopMatrixSparse* create_matrix_sym_sparse(const ExecSpaceInfo& hi, size_type size, size_type nnz);
  /// nullptr when `mem_space` is not ours
  static double* create_raw_array(const std::string& mem_space, size_type n);
  /// false when `mem_space` is not ours
  static bool delete_raw_array(const std::string& mem_space, double* a);
};

}  // namespace hiop
