/*
 * oracle/osqp_shim.c -- TEST INFRASTRUCTURE ONLY.  Drives the reference's vendored OSQP 0.5.0
 * (compiled unmodified from /root/reference/src/osqp by oracle/Makefile into oracle/_ref/libosqp_ref.so)
 * exactly as SparseCMPC::runSolverOSQP does (src/MPC_Ctrl/OsqpTriples.cpp:57-142): default settings with
 * eps_abs = eps_rel = 1e-5 (both overridable for the accuracy studies in tests/), cold start.
 */
#include <stdlib.h>

#include "osqp.h"

/* P (upper triangle) and A in compressed-sparse-column form.  Returns the OSQP status value
 * (1 = solved), x[n] = workspace->solution->x. */
long long osqp_ref_solve(long long n, long long m, long long P_nnz, double* P_x, long long* P_i, long long* P_p, double* q,
                         long long A_nnz, double* A_x, long long* A_i, long long* A_p, double* l, double* u,
                         double eps_abs, double eps_rel, long long max_iter, int polish, double* x, long long* iters) {
  OSQPSettings* settings = (OSQPSettings*)malloc(sizeof(OSQPSettings));
  OSQPData* data = (OSQPData*)malloc(sizeof(OSQPData));
  data->n = n;
  data->m = m;
  data->P = csc_matrix(n, n, P_nnz, P_x, P_i, P_p);
  data->q = q;
  data->A = csc_matrix(m, n, A_nnz, A_x, A_i, A_p);
  data->l = l;
  data->u = u;
  osqp_set_default_settings(settings);
  settings->eps_abs = eps_abs; /* OsqpTriples.cpp:100-101: 1e-5 */
  settings->eps_rel = eps_rel;
  if (max_iter > 0) settings->max_iter = max_iter;
  settings->polish = polish;
  settings->verbose = 0;
  /* The reference leaves adaptive_rho_interval = 0, which in its PROFILING build means "every time the iterations
   * have taken 40 % of the setup time": the iterate sequence then depends on wall-clock timing and is not
   * reproducible from run to run (a loaded host changed the iteration count, once the status).  The checker pins
   * the interval; everything else is the reference's default. */
  settings->adaptive_rho_interval = 50;
  OSQPWorkspace* work = osqp_setup(data, settings);
  long long status = -100;
  if (work) {
    osqp_solve(work);
    for (long long k = 0; k < n; k++) x[k] = work->solution->x[k];
    status = work->info->status_val;
    if (iters) *iters = work->info->iter;
    osqp_cleanup(work);
  }
  free(data->P);
  free(data->A);
  free(data);
  free(settings);
  return status;
}
