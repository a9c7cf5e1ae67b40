/* Test program for the C interface of the MDS solver (include/hiop_amd_interface.h): the reference's MdsEx1 problem
 * (src/Drivers/MDS/NlpMdsEx1.hpp:54-447) written as plain C callbacks, solved through hiop_mds_create/solve/destroy_problem —
 * what src/Drivers/MDS/NlpMdsEx1.c does against the reference's library.  That driver's acceptance check is the known answer:
 * objective -4.999509728895e+01 for ns = 400, nd = 100 to 1e-6 (NlpMdsEx1.c:376).
 *
 *   mds_c_interface host   [ns nd]    callbacks on host arrays (the reference's contract; the library stages over PCIe)
 *   mds_c_interface device [ns nd]    callbacks on DEVICE arrays: the library's device-resident example (hiopamd_mdsex1_*)
 * prints "obj=<%.15e> iters=<n> status=<s> nfact=<k>"; exit code 0 iff the solve succeeded (and, for 400 100, the check holds).
 *
 * Problem: variables (x[ns], s[ns], y[nd]);  min 1/2 sum x_i (x_i - 1) + 1/2 y'Qy + 1/2 s's
 *   x_i + s_i - sum(y) = 0 (i < ns);  -2 <= x_0 + sum(s) + sum(y) <= 2;  x_1 + sum(y) <= 2;  x_2 + sum(y) >= -2
 *   x <= 3, s >= 0, -4 <= y_0 <= 4;  Q = 2 on the diagonal, 1 on the first off-diagonals (rows 1..nd-2), 1e-8 elsewhere; x0 = 1. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hiop_amd.h"
#include "hiop_amd_interface.h"

typedef struct {
  int ns, nd;
  double* Q;
  hiopamd_mdsex1* dev; /* device mode only */
  hiopamd_ctx* ctx;
} prob_t;

static double Qentry(int nd, int i, int j)
{
  double v = 1e-8;
  if(i == j) v += 2.0;
  if(i >= 1 && i < nd - 1 && (j == i + 1)) v += 1.0; /* Q[i+1 + i*nd] and Q[i + (i+1)*nd] for i = 1..nd-2 */
  if(j >= 1 && j < nd - 1 && (i == j + 1)) v += 1.0;
  return v;
}

/* ------------------------------------------------------------ host callbacks */
static int sizes(hiop_size_type* n, hiop_size_type* m, void* u)
{
  prob_t* p = (prob_t*)u;
  *n = 2 * p->ns + p->nd;
  *m = p->ns + 3;
  return 0;
}
static int blocks(hiop_size_type* nxs, hiop_size_type* nxd, hiop_size_type* je, hiop_size_type* ji, hiop_size_type* hss,
                  hiop_size_type* hsd, void* u)
{
  prob_t* p = (prob_t*)u;
  *nxs = 2 * p->ns;
  *nxd = p->nd;
  *je = 2 * p->ns;
  *ji = p->ns + 3;
  *hss = 2 * p->ns;
  *hsd = 0;
  return 0;
}
static int start(hiop_size_type n, double* x0, void* u)
{
  (void)u;
  for(int i = 0; i < n; ++i) x0[i] = 1.0;
  return 0;
}
static int vars(hiop_size_type n, double* lo, double* up, void* u)
{
  prob_t* p = (prob_t*)u;
  const int ns = p->ns;
  for(int i = 0; i < n; ++i) {
    lo[i] = -1e20;
    up[i] = 1e20;
  }
  for(int i = 0; i < ns; ++i) up[i] = 3.0;
  for(int i = ns; i < 2 * ns; ++i) lo[i] = 0.0;
  lo[2 * ns] = -4.0;
  up[2 * ns] = 4.0;
  return 0;
}
/* a badly scaled variant (host callbacks only; MDS_KOBJ / MDS_KROW in the environment): objective x KOBJ, the first inequality
 * (row ns: x_0 + sum(s) + sum(y) in [-2, 2]) x KROW, body and bounds.  Same minimiser. */
static double KOBJ = 1.0, KROW = 1.0;
static int cons_info(hiop_size_type m, double* lo, double* up, void* u)
{
  (void)u;
  for(int i = 0; i < m; ++i) lo[i] = up[i] = 0.0;
  lo[m - 3] = -2.0 * KROW, up[m - 3] = 2.0 * KROW;
  lo[m - 2] = -1e20, up[m - 2] = 2.0;
  lo[m - 1] = -2.0, up[m - 1] = 1e20;
  return 0;
}
static int f_host(hiop_size_type n, double* x, int new_x, double* obj, void* u)
{
  prob_t* p = (prob_t*)u;
  (void)n, (void)new_x;
  const int ns = p->ns, nd = p->nd;
  const double *s = x + ns, *y = x + 2 * ns;
  double a = 0.0, b = 0.0, c = 0.0;
  for(int i = 0; i < ns; ++i) a += x[i] * (x[i] - 1.0);
  for(int i = 0; i < nd; ++i) {
    double qy = 0.0;
    for(int j = 0; j < nd; ++j) qy += p->Q[i * nd + j] * y[j];
    b += qy * y[i];
  }
  for(int i = 0; i < ns; ++i) c += s[i] * s[i];
  *obj = KOBJ * (0.5 * a + 0.5 * b + 0.5 * c);
  return 0;
}
static int g_host(hiop_size_type n, double* x, int new_x, double* g, void* u)
{
  prob_t* p = (prob_t*)u;
  (void)n, (void)new_x;
  const int ns = p->ns, nd = p->nd;
  for(int i = 0; i < ns; ++i) g[i] = KOBJ * (x[i] - 0.5);
  for(int i = 0; i < ns; ++i) g[ns + i] = KOBJ * x[ns + i];
  for(int i = 0; i < nd; ++i) {
    double qy = 0.0;
    for(int j = 0; j < nd; ++j) qy += p->Q[i * nd + j] * x[2 * ns + j];
    g[2 * ns + i] = KOBJ * qy;
  }
  return 0;
}
static int c_host(hiop_size_type n, hiop_size_type m, double* x, int new_x, double* c, void* u)
{
  prob_t* p = (prob_t*)u;
  (void)n, (void)m, (void)new_x;
  const int ns = p->ns, nd = p->nd;
  double ey = 0.0, es = 0.0;
  for(int j = 0; j < nd; ++j) ey += x[2 * ns + j];
  for(int i = 0; i < ns; ++i) es += x[ns + i];
  for(int i = 0; i < ns; ++i) c[i] = x[i] + x[ns + i] - ey;
  c[ns] = KROW * (x[0] + es + ey);
  c[ns + 1] = x[1] + ey;
  c[ns + 2] = x[2] + ey;
  return 0;
}
static int jac_host(hiop_size_type n, hiop_size_type m, double* x, int new_x, hiop_size_type nsp, hiop_size_type nde,
                    hiop_size_type nnz, hiop_index_type* iJ, hiop_index_type* jJ, double* MJ, double* JD, void* u)
{
  prob_t* p = (prob_t*)u;
  (void)n, (void)x, (void)new_x, (void)nsp, (void)nnz;
  const int ns = p->ns;
  int t = 0;
  for(int i = 0; i < ns; ++i) { /* x_i + s_i */
    if(iJ) iJ[t] = i, jJ[t] = i;
    if(MJ) MJ[t] = 1.0;
    ++t;
    if(iJ) iJ[t] = i, jJ[t] = ns + i;
    if(MJ) MJ[t] = 1.0;
    ++t;
  }
  if(iJ) iJ[t] = ns, jJ[t] = 0; /* x_0 + sum(s) */
  if(MJ) MJ[t] = KROW;
  ++t;
  for(int i = 0; i < ns; ++i, ++t) {
    if(iJ) iJ[t] = ns, jJ[t] = ns + i;
    if(MJ) MJ[t] = KROW;
  }
  if(iJ) iJ[t] = ns + 1, jJ[t] = 1;
  if(MJ) MJ[t] = 1.0;
  ++t;
  if(iJ) iJ[t] = ns + 2, jJ[t] = 2;
  if(MJ) MJ[t] = 1.0;
  ++t;
  if(JD) {
    for(int i = 0; i < m; ++i)
      for(int j = 0; j < nde; ++j) JD[(size_t)i * nde + j] = i < ns ? -1.0 : (i == ns ? KROW : 1.0);
  }
  return 0;
}
static int hess_host(hiop_size_type n, hiop_size_type m, double* x, int new_x, double obj_factor, double* lambda, int new_lambda,
                     hiop_size_type nsp, hiop_size_type nde, hiop_size_type nnzHSS, hiop_index_type* iH, hiop_index_type* jH,
                     double* MH, double* HDD, hiop_size_type nnzHSD, hiop_index_type* iSD, hiop_index_type* jSD, double* MSD, void* u)
{
  prob_t* p = (prob_t*)u;
  (void)n, (void)m, (void)x, (void)new_x, (void)lambda, (void)new_lambda, (void)nsp, (void)nnzHSD, (void)iSD, (void)jSD, (void)MSD;
  for(int t = 0; t < nnzHSS; ++t) {
    if(iH) iH[t] = jH[t] = t;
    if(MH) MH[t] = obj_factor * KOBJ;
  }
  if(HDD)
    for(int i = 0; i < nde * nde; ++i) HDD[i] = obj_factor * KOBJ * p->Q[i];
  return 0;
}

/* ------------------------------------------------------------ device callbacks: thin forwards to hiopamd_mdsex1_* */
static int f_dev(hiop_size_type n, double* x, int new_x, double* obj, void* u)
{
  (void)n, (void)new_x;
  return hiopamd_mdsex1_eval_f(((prob_t*)u)->dev, x, obj);
}
static int g_dev(hiop_size_type n, double* x, int new_x, double* g, void* u)
{
  prob_t* p = (prob_t*)u;
  (void)n, (void)new_x;
  int rc = hiopamd_mdsex1_eval_grad_f(p->dev, x, g);
  return rc ? rc : hiopamd_ctx_sync(p->ctx);
}
static int c_dev(hiop_size_type n, hiop_size_type m, double* x, int new_x, double* c, void* u)
{
  prob_t* p = (prob_t*)u;
  (void)n, (void)m, (void)new_x;
  int rc = hiopamd_mdsex1_eval_cons(p->dev, x, c);
  return rc ? rc : hiopamd_ctx_sync(p->ctx);
}
static int jac_dev(hiop_size_type n, hiop_size_type m, double* x, int new_x, hiop_size_type nsp, hiop_size_type nde,
                   hiop_size_type nnz, hiop_index_type* iJ, hiop_index_type* jJ, double* MJ, double* JD, void* u)
{
  prob_t* p = (prob_t*)u;
  if(iJ) return jac_host(n, m, NULL, new_x, nsp, nde, nnz, iJ, jJ, NULL, NULL, u); /* the pattern is asked for on the host */
  const int ns = p->ns;
  int rc = hiopamd_mdsex1_eval_Jac_cons_eq(p->dev, x, NULL, NULL, MJ, JD);
  if(!rc) rc = hiopamd_mdsex1_eval_Jac_cons_ineq(p->dev, x, ns, NULL, NULL, MJ ? MJ + 2 * ns : NULL, JD ? JD + (size_t)ns * nde : NULL);
  return rc ? rc : hiopamd_ctx_sync(p->ctx);
}
static int hess_dev(hiop_size_type n, hiop_size_type m, double* x, int new_x, double obj_factor, double* lambda, int new_lambda,
                    hiop_size_type nsp, hiop_size_type nde, hiop_size_type nnzHSS, hiop_index_type* iH, hiop_index_type* jH,
                    double* MH, double* HDD, hiop_size_type nnzHSD, hiop_index_type* iSD, hiop_index_type* jSD, double* MSD, void* u)
{
  prob_t* p = (prob_t*)u;
  if(iH) return hess_host(n, m, NULL, new_x, obj_factor, NULL, new_lambda, nsp, nde, nnzHSS, iH, jH, NULL, NULL, nnzHSD, iSD, jSD, MSD, u);
  int rc = hiopamd_mdsex1_eval_Hess_Lagr(p->dev, x, obj_factor, lambda, NULL, NULL, MH, HDD);
  return rc ? rc : hiopamd_ctx_sync(p->ctx);
}

int main(int argc, char** argv)
{
  const int device = argc > 1 && strcmp(argv[1], "device") == 0;
  prob_t P;
  memset(&P, 0, sizeof(P));
  P.ns = argc > 3 ? atoi(argv[2]) : 400;
  P.nd = argc > 3 ? atoi(argv[3]) : 100;
  const double tol = argc > 4 ? atof(argv[4]) : 0.0;
  const int n = 2 * P.ns + P.nd;
  if(getenv("MDS_KOBJ")) KOBJ = atof(getenv("MDS_KOBJ"));
  if(getenv("MDS_KROW")) KROW = atof(getenv("MDS_KROW"));
  P.Q = (double*)malloc(sizeof(double) * (size_t)P.nd * P.nd);
  for(int i = 0; i < P.nd; ++i)
    for(int j = 0; j < P.nd; ++j) P.Q[i * P.nd + j] = Qentry(P.nd, i, j);
  cHiopMDSProblem prob;
  memset(&prob, 0, sizeof(prob));
  prob.user_data = &P;
  prob.get_starting_point = start;
  prob.get_prob_sizes = sizes;
  prob.get_vars_info = vars;
  prob.get_cons_info = cons_info;
  prob.get_sparse_dense_blocks_info = blocks;
  prob.eval_f = device ? f_dev : f_host;
  prob.eval_grad_f = device ? g_dev : g_host;
  prob.eval_cons = device ? c_dev : c_host;
  prob.eval_Jac_cons = device ? jac_dev : jac_host;
  prob.eval_Hess_Lagr = device ? hess_dev : hess_host;
  prob.solution = (double*)calloc((size_t)n, sizeof(double));
  if(device) {
    if(hiopamd_ctx_create(&P.ctx, NULL) != 0 || hiopamd_mdsex1_create(&P.dev, P.ctx, P.ns, P.nd, 0) != 0) {
      fprintf(stderr, "cannot create the device-resident example problem\n");
      return 2;
    }
  }
  if(hiop_mds_create_problem(&prob) != 0) return 3;
  if(device && hiopamd_mds_set_callback_mem_space(&prob, 1) != 0) return 4;
  if(tol > 0) hiopamd_mds_set_numeric_option(&prob, "tolerance", tol);
  const int rc = hiop_mds_solve_problem(&prob);
  int status = 0, iters = 0, nfact = 0;
  hiopamd_mds_get_solve_info(&prob, &status, &iters, &nfact);
  double xsum = 0.0;
  for(int i = 0; i < n; ++i) xsum += prob.solution[i];
  printf("obj=%.15e iters=%d status=%d nfact=%d xsum=%.12e rc=%d\n", prob.obj_value, iters, status, nfact, xsum, rc);
  double t_total = 0.0, t_kkt = 0.0;
  hiopamd_mds_get_solve_times(&prob, &t_total, &t_kkt);
  printf("times: total=%.6f kkt=%.6f\n", t_total, t_kkt);
  int ret = rc != 0;
  if(!ret && P.ns == 400 && P.nd == 100 && tol == 0.0 && fabs(prob.obj_value - (-4.999509728895e+01)) > 1e-6) {
    printf("objective mismatch for MDS Ex1 C interface problem with 400 sparse variables and 100 dense variables\n");
    ret = 1;
  }
  if(getenv("HIOPAMD_TEST_RESOLVE")) {
    /* solve the SAME problem object again (the reference allows it: chiopInterface.cpp:79-87 builds a fresh solver per call) */
    const double obj1 = prob.obj_value;
    const int it1 = iters;
    for(int i = 0; i < n; ++i) prob.solution[i] = -7.0;
    const int rc2 = hiop_mds_solve_problem(&prob);
    int st2 = 0, it2 = 0, nf2 = 0;
    hiopamd_mds_get_solve_info(&prob, &st2, &it2, &nf2);
    double xsum2 = 0.0;
    for(int i = 0; i < n; ++i) xsum2 += prob.solution[i];
    printf("resolve: obj=%.15e iters=%d status=%d rc=%d same_obj=%d same_iters=%d same_x=%d\n", prob.obj_value, it2, st2, rc2, prob.obj_value == obj1,
           it2 == it1, xsum2 == xsum);
    if(rc2 != 0 || prob.obj_value != obj1 || it2 != it1 || xsum2 != xsum) ret = 1;
  }
  hiop_mds_destroy_problem(&prob);
  if(device) {
    hiopamd_mdsex1_destroy(P.dev);
    hiopamd_ctx_destroy(P.ctx);
  }
  free(prob.solution);
  free(P.Q);
  return ret;
}
