// convex_mpc_shim.cpp -- the reference's single-robot MPC entry points
// (src/MPC_Ctrl/convexMPC_interface.h:40-48) as a batch-of-one client of the
// batched HIP solver.  Host C++ only; all compute goes through libqmpc.so.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/convexMPC_interface.h"
#include "../../include/qmpc.h"

namespace {

struct ShimState {
  qmpc_handle h = nullptr;
  int horizon = 0;
  bool has_solved = false;     // convexMPC_interface.cpp:81
  float x_drag = 0.f;          // update.x_drag
  int max_iter = 1000;
  double use_jcqp = 0.0;       // update.use_jcqp (convexMPC_interface.cpp:118)
  double rho = 1e-7, sigma = 1e-8, alpha = 1.5, terminate = 0.1;  // update.rho ... (:113-117)
  std::vector<double> q_soln;  // SolverMPC.cpp:45
  int status = -1, iters = 0;
} g;

bool ensure_handle() {
  if (g.h) return true;
  const int rc = qmpc_create(0, 1, QMPC_MAX_HORIZON, &g.h);
  if (rc != QMPC_OK) {
    std::fprintf(stderr, "[qmpc shim] qmpc_create failed (rc=%d): no usable HIP device\n", rc);
    g.h = nullptr;
    return false;
  }
  return true;
}

void solve_floats(const float* p, const float* v, const float* q, const float* w, const float* r,
                  float yaw, const float* weights, const float* traj, float alpha, const int* gait) {
  if (!g.h || g.horizon <= 0) {
    // no accepted setup_problem (never called, no device, or a horizon / mu / dt it refused): nothing is solved,
    // get_solution reads 0 and the failure is visible through qmpc_shim_last_status()
    std::fprintf(stderr, "[qmpc shim] update_problem_data without an accepted setup_problem\n");
    g.status = QMPC_SHIM_ERR_SETUP;
    g.has_solved = false;
    return;
  }
  const int h = g.horizon;
  std::vector<uint8_t> gt(4 * h);
  for (int i = 0; i < 4 * h; ++i) gt[i] = (uint8_t)gait[i];  // mint_to_u8, interface.cpp:76-79
  qmpc_inputs in;
  std::memset(&in, 0, sizeof(in));
  in.p = p; in.v = v; in.q = q; in.w = w; in.r = r; in.yaw = &yaw; in.traj = traj;
  in.gait = gt.data(); in.weights = weights; in.alpha = &alpha; in.x_drag = &g.x_drag;
  float grf[12];
  int32_t st = 0, it = 0;
  g.q_soln.assign(12 * h, 0.0);
  qmpc_outputs out;
  out.grf = grf; out.soln = g.q_soln.data(); out.status = &st; out.iters = &it;
  // use_jcqp = 1 / 2: the reference's JCQP/ADMM alternate (SolverMPC.cpp:400-414, :558-610), reproduced
  // on the GPU; otherwise the exact solve
  // (the double is thresholded exactly like convexMPC_interface.cpp:113-118: > 1.5 -> 2, > 0.5 -> 1, else 0)
  const int mode = (g.use_jcqp > 1.5) ? 2 : (g.use_jcqp > 0.5 ? 1 : 0);
  int rc = qmpc_settings_jcqp(g.h, mode, g.max_iter, g.rho, g.sigma, g.alpha, g.terminate);
  if (rc != QMPC_OK) {
    // e.g. rho <= 0: the reference would factor a singular KKT matrix; here the call is refused and reported
    std::fprintf(stderr, "[qmpc shim] update_solver_settings values rejected for use_jcqp=%d (rc=%d)\n", mode, rc);
    g.status = QMPC_SHIM_ERR_SETTINGS;
    g.q_soln.assign(12 * h, 0.0);
    return;
  }
  rc = qmpc_solve_host(g.h, 1, &in, &out);
  if (rc != QMPC_OK) {
    std::fprintf(stderr, "[qmpc shim] solve failed rc=%d %s\n", rc, qmpc_last_error(g.h));
    g.status = QMPC_SHIM_ERR_SOLVE;
    g.q_soln.assign(12 * h, 0.0);
    return;
  }
  g.status = st;
  g.iters = it;
  if (st & QMPC_ST_ERROR_MASK) std::printf("failed to solve! (status bits %d)\n", st);  // SolverMPC.cpp:541
  g.has_solved = true;
}

}  // namespace

extern "C" {

void setup_problem(double dt, int horizon, double mu, double f_max) {
  if (!ensure_handle()) {
    g.status = QMPC_SHIM_ERR_SETUP;
    return;
  }
  const int rc = qmpc_setup(g.h, dt, horizon, mu, f_max);
  if (rc != QMPC_OK) {
    std::fprintf(stderr, "[qmpc shim] setup_problem(dt=%g, horizon=%d, mu=%g, f_max=%g) rejected rc=%d\n",
                 dt, horizon, mu, f_max, rc);
    g.horizon = 0;
    g.status = QMPC_SHIM_ERR_SETUP;  // loud: not only stderr (the reference accepts up to K_MAX_GAIT_SEGMENTS)
    g.has_solved = false;            // get_solution reads 0 from now on, never a previous horizon's forces
    return;
  }
  g.horizon = horizon;
  qmpc_settings(g.h, g.max_iter, 1e-9);
}

void update_solver_settings(int max_iter, double rho, double sigma, double solver_alpha, double terminate,
                            double use_jcqp) {
  g.use_jcqp = use_jcqp;
  g.rho = rho;
  g.sigma = sigma;
  g.alpha = solver_alpha;
  g.terminate = terminate;
  // the reference's active path caps qpOASES at nWSR = 100 regardless of this
  // value (SolverMPC.cpp:435); max_iter (10000 in the caller) bounds ours.
  g.max_iter = max_iter > 0 ? max_iter : 1000;
  if (g.h) qmpc_settings(g.h, g.max_iter, 1e-9);
}

void update_problem_data_floats(float* p, float* v, float* q, float* w, float* r, float yaw,
                                float* weights, float* state_trajectory, float alpha, int* gait) {
  solve_floats(p, v, q, w, r, yaw, weights, state_trajectory, alpha, gait);
}

void update_problem_data(double* p, double* v, double* q, double* w, double* r, double yaw,
                         double* weights, double* state_trajectory, double alpha, int* gait) {
  // mfp_to_flt, convexMPC_interface.cpp:69-73 / :88-105
  const int h = g.horizon > 0 ? g.horizon : 0;
  float fp[3], fv[3], fq[4], fw[3], fr[12], fwt[12];
  std::vector<float> ft(12 * (h > 0 ? h : 1));
  for (int i = 0; i < 3; ++i) { fp[i] = (float)p[i]; fv[i] = (float)v[i]; fw[i] = (float)w[i]; }
  for (int i = 0; i < 4; ++i) fq[i] = (float)q[i];
  for (int i = 0; i < 12; ++i) { fr[i] = (float)r[i]; fwt[i] = (float)weights[i]; }
  for (int i = 0; i < 12 * h; ++i) ft[i] = (float)state_trajectory[i];
  solve_floats(fp, fv, fq, fw, fr, (float)yaw, fwt, ft.data(), (float)alpha, gait);
}

double get_solution(int index) {
  if (!g.has_solved) return 0.0;
  if (index < 0 || index >= (int)g.q_soln.size()) return 0.0;
  return g.q_soln[index];
}

int qmpc_shim_last_status(void) { return g.status; }
int qmpc_shim_last_iters(void) { return g.iters; }

}  // extern "C"

void update_x_drag(float x_drag) { g.x_drag = x_drag; }
