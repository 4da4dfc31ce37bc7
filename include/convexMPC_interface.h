/*
 * convexMPC_interface.h -- single-robot compatibility surface.
 *
 * Declares the entry points that the reference's MPC caller
 * (src/MPC_Ctrl/ConvexMPCLocomotion.cpp:630-674, solveDenseMPC) binds to,
 * with the signatures of the reference's src/MPC_Ctrl/convexMPC_interface.h
 * lines 40-48.  libconvexmpc_shim.so implements them as a batch-of-one client
 * of the batched HIP solver (include/qmpc.h), so the reference's
 * ConvexMPCLocomotion / GaitCtrller sources link against it unchanged in place
 * of SolverMPC.cpp + convexMPC_interface.cpp + qpOASES/JCQP.
 *
 * Behaviour kept from the reference: blocking solve inside
 * update_problem_data_floats / update_problem_data; get_solution(i) returns 0
 * until the first solve (convexMPC_interface.cpp:175-180); swing-foot entries
 * of the solution are 0; all state is process-global, one caller thread.
 * Added: qmpc_shim_last_status() because the reference reports solver failure
 * only by printing (SolverMPC.cpp:539-541).
 */
#ifndef QMPC_CONVEXMPC_INTERFACE_COMPAT_H
#define QMPC_CONVEXMPC_INTERFACE_COMPAT_H

#ifdef __cplusplus
#define QMPC_C_LINKAGE extern "C"
#else
#define QMPC_C_LINKAGE
#endif

/* reference :40  (horizon <= 36 = K_MAX_GAIT_SEGMENTS like the reference, any gait; see QMPC_MAX_HORIZON in qmpc.h for the
 * routes the long ones take) */
QMPC_C_LINKAGE void setup_problem(double dt, int horizon, double mu, double f_max);
/* reference :41  double-precision (MATLAB) twin of the floats entry */
QMPC_C_LINKAGE void update_problem_data(double* p, double* v, double* q, double* w,
                                        double* r, double yaw, double* weights,
                                        double* state_trajectory, double alpha, int* gait);
/* reference :42  q_soln[index], index < 12*horizon; 0.0 before the first solve */
QMPC_C_LINKAGE double get_solution(int index);
/* reference :43  use_jcqp = 0: exact solve, max_iter caps the active-set iterations; use_jcqp = 1 / 2:
 * the reference's JCQP/ADMM alternate with these very knobs (qmpc_settings_jcqp) */
QMPC_C_LINKAGE void update_solver_settings(int max_iter, double rho, double sigma,
                                           double solver_alpha, double terminate,
                                           double use_jcqp);
/* reference :44-46  q = (w,x,y,z); r axis-major r[axis*4+foot]; gait[4*horizon] */
QMPC_C_LINKAGE void update_problem_data_floats(float* p, float* v, float* q, float* w,
                                               float* r, float yaw, float* weights,
                                               float* state_trajectory, float alpha,
                                               int* gait);
/* reference :48  (C++ linkage in the reference, kept so the mangled name matches) */
#ifdef __cplusplus
void update_x_drag(float x_drag);
#endif

/* status of the most recent call: >= 0 = status bits (QMPC_ST_* of qmpc.h) of the most recent solve;
 * -1 = never solved; other negative values = the call was refused and get_solution reads 0:
 *   QMPC_SHIM_ERR_SETUP     setup_problem rejected (horizon beyond QMPC_MAX_HORIZON, mu / dt <= 0, no device)
 *                           or update_problem_data* without an accepted setup_problem
 *   QMPC_SHIM_ERR_SETTINGS  update_solver_settings values unusable for the selected use_jcqp (e.g. rho <= 0).  use_jcqp = 1 / 2
 *                           is honoured at every horizon setup_problem accepts, like the reference (SolverMPC.cpp:400-420,
 *                           :558-631): beyond 192 variables the ADMM runs on the large-problem path
 *   QMPC_SHIM_ERR_SOLVE     the batched solver returned an error code (see stderr)
 * use_jcqp is thresholded like the reference (convexMPC_interface.cpp:113-118): > 1.5 -> 2, > 0.5 -> 1, else 0. */
#define QMPC_SHIM_ERR_SETUP (-2)
#define QMPC_SHIM_ERR_SETTINGS (-3)
#define QMPC_SHIM_ERR_SOLVE (-4)
QMPC_C_LINKAGE int qmpc_shim_last_status(void);
/* active-set iterations of the most recent solve */
QMPC_C_LINKAGE int qmpc_shim_last_iters(void);

#endif
