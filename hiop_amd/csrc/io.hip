// `.iajaaa` linear-system dumps, byte-compatible with the reference's writer (src/Utils/hiopCSR_IO.hpp:44-152,
// format description src/LinAlg/csr_iajaaa.md:9-29): header m, nx, meq, mineq, nnz; 1-based CSR row pointers and
// column indices of the UPPER triangle (entries with |a| <= 1e-25 dropped); values, right-hand sides and solutions with
// "%.20f ".  The KKT matrix lives in HBM: it is copied to the host once per dump (this is a debugging/fixture path,
// the reference's `write_kkt yes` option, not part of the timed hot path).
#include "common.hpp"

#include <cctype>
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

// hiopAlgFilterIPMNewton::outputIteration (hiopAlgFilterIPM.cpp:2783-2812) / ...QuasiNewton::outputIteration (:1521-1549)
int hiopamd_io_iteration_header(char* buf, int buflen)
{
  if(!buf || buflen <= 0) return HIOPAMD_ERR_ARG;
  const int n = std::snprintf(buf, (size_t)buflen, "iter    objective     inf_pr     inf_du   lg(mu)  alpha_du   alpha_pr linesrch\n");
  return (n < 0 || n >= buflen) ? HIOPAMD_ERR_ARG : n;
}

int hiopamd_io_format_iteration(char* buf, int buflen, int quasi_newton, int iter, double obj, double inf_pr, double inf_du,
                                double mu, double alpha_du, double alpha_pr, int ls_status, int ls_num, int use_soc,
                                int use_fr)
{
  if(!buf || buflen <= 0) return HIOPAMD_ERR_ARG;
  int n;
  if(ls_status == -1) {
    n = std::snprintf(buf, (size_t)buflen, "%4d %14.7e %7.3e  %7.3e %6.2f  %7.3e  %7.3e  -(-)\n", iter, obj, inf_pr, inf_du,
                      std::log10(mu), alpha_du, alpha_pr);
  } else {
    char step[2] = {'?', 0};
    if(ls_status == 1) step[0] = 's';
    else if(ls_status == 2) step[0] = 'h';
    else if(ls_status == 3) step[0] = 'f';
    if(use_soc && ls_status >= 1 && ls_status <= 3) step[0] = (char)std::toupper(step[0]);
    if(use_fr) {
      if(!quasi_newton) ls_num = 0;   // only the Newton variant resets the count (:2803)
      step[0] = 'R';
    }
    n = std::snprintf(buf, (size_t)buflen, "%4d %14.7e %7.3e  %7.3e %6.2f  %7.3e  %7.3e  %d(%s)\n", iter, obj, inf_pr, inf_du,
                      std::log10(mu), alpha_du, alpha_pr, ls_num, step);
  }
  return (n < 0 || n >= buflen) ? HIOPAMD_ERR_ARG : n;
}

}  // extern "C"
