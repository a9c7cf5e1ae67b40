// `.iajaaa` linear-system dumps, byte-compatible with the reference's writer (src/Utils/hiopCSR_IO.hpp:44-152,
// format description src/LinAlg/csr_iajaaa.md:9-29): header m, nx, meq, mineq, nnz; 1-based CSR row pointers and
// column indices of the UPPER triangle (entries with |a| <= 1e-25 dropped); values, right-hand sides and solutions with
// "%.20f ".  The KKT matrix lives in HBM: it is copied to the host once per dump (this is a debugging/fixture path,
// the reference's `write_kkt yes` option, not part of the timed hot path).
#include "common.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

extern "C" {

int hiopamd_io_write_iajaaa_matrix(hiopamd_ctx* ctx, const char* path, int m, const double* M_dev, int64_t ld, int nx,
                                   int meq, int mineq)
{
  if(!ctx || !path || m < 0 || !M_dev || ld < m) return HIOPAMD_ERR_ARG;
  std::vector<double> M((size_t)m * (size_t)m);
  HIOPAMD_CHECK(hipMemcpy2DAsync(M.data(), sizeof(double) * (size_t)m, M_dev, sizeof(double) * (size_t)ld,
                                 sizeof(double) * (size_t)m, (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
  HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  FILE* f = std::fopen(path, "w+");   // hiopCSR_IO.hpp:104
  if(!f) return HIOPAMD_ERR_ARG;
  const double zero_tol = 1e-25;       // :110
  int nnz = 0;
  for(int i = 0; i < m; i++)
    for(int j = i; j < m; j++)
      if(std::fabs(M[(size_t)i * m + j]) > zero_tol) nnz++;
  std::fprintf(f, "%d\n%d\n%d\n%d\n%d\n", m, nx, meq, mineq, nnz);   // :121
  int offset = 1;
  std::fprintf(f, "%d ", offset);
  for(int i = 0; i < m; i++) {
    for(int j = i; j < m; j++)
      if(std::fabs(M[(size_t)i * m + j]) > zero_tol) offset++;
    std::fprintf(f, "%d ", offset);
  }
  std::fprintf(f, "\n");
  for(int i = 0; i < m; i++)
    for(int j = i; j < m; j++)
      if(std::fabs(M[(size_t)i * m + j]) > zero_tol) std::fprintf(f, "%d ", j + 1);
  std::fprintf(f, "\n");
  for(int i = 0; i < m; i++)
    for(int j = i; j < m; j++)
      if(std::fabs(M[(size_t)i * m + j]) > zero_tol) std::fprintf(f, "%.20f ", M[(size_t)i * m + j]);
  std::fprintf(f, "\n");
  std::fclose(f);
  return HIOPAMD_OK;
}

// writeRhsToFile / writeSolToFile (:44-76): append one vector as a line
int hiopamd_io_append_iajaaa_vector(hiopamd_ctx* ctx, const char* path, int m, const double* v_dev)
{
  if(!ctx || !path || m < 0 || (m > 0 && !v_dev)) return HIOPAMD_ERR_ARG;
  std::vector<double> v((size_t)m);
  if(m > 0) {
    HIOPAMD_CHECK(hipMemcpyAsync(v.data(), v_dev, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    HIOPAMD_CHECK(hipStreamSynchronize(ctx->stream));
  }
  FILE* f = std::fopen(path, "a+");   // :52
  if(!f) return HIOPAMD_ERR_ARG;
  for(int i = 0; i < m; i++) std::fprintf(f, "%.20f ", v[i]);
  std::fprintf(f, "\n");
  std::fclose(f);
  return HIOPAMD_OK;
}

}  // extern "C"
