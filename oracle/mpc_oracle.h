/*
 * oracle/mpc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, Eigen-free) of the reference's convex-MPC hot
 * path, used as the parity checker for the HIP solver.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (quadruped_ctrl_amd/, include/qmpc.h) never links or calls it.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/MPC_Ctrl unless stated).
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *   - QP solve: pinned.  The solver is the reference's own vendored
 *     qpOASES 3.2.0, compiled unmodified from /root/reference by
 *     oracle/Makefile into oracle/_ref/ and driven exactly as
 *     SolverMPC.cpp:527-541 drives it (oracle/qpoases_shim.cpp).
 *   - QP assembly (SolverMPC.cpp:296-525): PARITY UNPINNED by reference
 *     execution.  The reference's assembly needs Eigen 3 (un-vendored,
 *     unpinned, absent from this image), so it cannot be compiled here, and
 *     the reference ships no tests or golden vectors for it.  This file is a
 *     line-by-line restatement in the same precision (float), cross-checked
 *     analytically (scipy expm, fp64 twin, KKT residuals) in tests/.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_HORIZON 36 /* K_MAX_GAIT_SEGMENTS, convexMPC_interface.h:3 */

/* convexMPC_interface.h:13-19 */
typedef struct {
  float dt;
  float mu;
  float f_max;
  int horizon;
} oracle_setup_t;

/* convexMPC_interface.h:21-38 (payload fields only; gait sized 4*36 so that
 * h >= 10 does not overflow into hack_pad as the reference's u8 gait[36] does) */
typedef struct {
  float p[3];
  float v[3];
  float q[4]; /* w,x,y,z */
  float w[3];
  float r[12]; /* axis-major: r[axis*4 + foot] */
  float yaw;
  float weights[12];
  float traj[12 * ORACLE_MAX_HORIZON];
  float alpha;
  unsigned char gait[4 * ORACLE_MAX_HORIZON];
  float x_drag;
} oracle_update_t;

/* Signature of the QP back end (the real qpOASES via oracle/_ref, see
 * qpoases_shim.cpp).  Returns 0 on success. */
typedef int (*oracle_qp_fn)(int nv, int nc, const double* H, const double* g,
                            const double* A, const double* lb,
                            const double* ub, int nwsr_max, double* x,
                            int* nwsr_used);

/* SolverMPC.cpp:257-267.  rpy[0]=yaw, rpy[1]=pitch, rpy[2]=roll. */
void oracle_quat_to_rpy(const float q_wxyz[4], float rpy[3]);

/* SolverMPC.cpp:235-254 (+ :226-233 cross_mat, :319 I_world,
 * RobotState.cpp:25-40).  A is 13x13, B 13x12, row-major. */
void oracle_ct_ss_mats(const float r[12], float yaw, float x_drag, float* A,
                       float* B);

/* SolverMPC.cpp:87-95: Adt (13x13), Bdt (13x12) from expm of the 25x25
 * block matrix, float. */
void oracle_c2d(const float* A, const float* B, float dt, float* Adt,
                float* Bdt);

/* Full-size assembly, SolverMPC.cpp:298-399 + :423-429.
 * Outputs (row-major doubles, as handed to qpOASES by the reference):
 *   H[n*n], g[n], Acon[m*n], lb[m], ub[m]   n=12h, m=20h
 * x0_out[13] optional (may be NULL). */
void oracle_assemble(const oracle_update_t* u, const oracle_setup_t* s,
                     double* H, double* g, double* Acon, double* lb,
                     double* ub, float* x0_out);

/* Evaluation order of the float condensation (see mpc_oracle.c): 0 = default
 * restatement; 1..5 = alternative, equally legitimate orders, used only by
 * tests/golden/make_noise_floor.py to measure the reference's own fp32 noise. */
void oracle_set_accum_mode(int mode);
int oracle_get_accum_mode(void);

/* Swing elimination, SolverMPC.cpp:441-525.  Returns new_vars; writes
 * new_cons, var_elim[n] flags and the gathered reduced problem. */
int oracle_reduce(int n, int m, const double* H, const double* g,
                  const double* Acon, const double* lb, const double* ub,
                  char* var_elim, int* new_cons_out, double* H_red,
                  double* g_red, double* A_red, double* lb_red,
                  double* ub_red);

/* Whole solve_mpc, SolverMPC.cpp:296-557 with use_jcqp==0: assembly,
 * reduction, qp(...) with nWSR=100, scatter into q_soln[12h] (zeros for
 * swing).  Returns qp's return code (0 ok); nwsr_out optional. */
int oracle_solve_mpc(const oracle_update_t* u, const oracle_setup_t* s,
                     oracle_qp_fn qp, double* q_soln, int* nwsr_out);

/* Batched form of the above for baseline timing; returns #failures. */
int oracle_solve_mpc_batch(const oracle_update_t* u, int count,
                           const oracle_setup_t* s, oracle_qp_fn qp,
                           double* q_soln, int* nwsr_out);

/* Gait.cpp:142-166 OffsetDurationGait::getMpcTable with _iteration given
 * (Gait.cpp:189).  table[4*n_segments]. */
void oracle_mpc_table(int n_segments, const int offsets[4],
                      const int durations[4], int iteration, int* table);

/* ------------------------------------------------------------------------
 * Caller side of the solve (SURVEY.md row a12 / a11's consumer): what
 * ConvexMPCLocomotion::updateMPCIfNeeded (ConvexMPCLocomotion.cpp:498-577) and
 * ::solveDenseMPC (:592-665) compute around update_problem_data_floats.
 * PARITY UNPINNED like the assembly (these members need Eigen to compile);
 * plain float arithmetic restated operation by operation.
 */
typedef struct {
  /* StateEstimate seResult (read at :502, :594) */
  float position[3];
  float v_world[3];
  float omega_world[3];
  float orientation[4]; /* w,x,y,z */
  float rpy[3];
  float r_body[9];  /* rBody, row-major */
  float p_foot[12]; /* pFoot[leg][axis] (world), p_foot[3*leg + axis], :161 of the header */
  /* controller members */
  float vel_des[3]; /* _x_vel_des, _y_vel_des, _yaw_turn_rate */
  float yaw_des_true;
  float rpy_comp[2];
  float stand_traj[6];
  float rp_des[2]; /* _roll_des, _pitch_des */
  int gait_type;   /* current_gait; 4 = standing (:514) */
  int gait_offsets[4], gait_durations[4], gait_iteration;
  float body_height;
  int omni_mode;
} oracle_command_t;

/* updateMPCIfNeeded :498-577 (trajAll, world_position_desired clamp) +
 * solveDenseMPC :598-640 (Q, alpha, r, yaw, x_drag = x_comp_integral BEFORE
 * its update :632-640) + Gait.cpp:142-166 (contact table, n_segments =
 * horizon).  wpd[2] = world_position_desired x,y (in/out), xci =
 * x_comp_integral (in/out).  dt_mpc = float(dt * iterationsBetweenMPC). */
void oracle_pack_command(const oracle_command_t* c, float dt_mpc, int horizon,
                         float wpd[2], float* xci, oracle_update_t* u);

/* ConvexMPCLocomotion.cpp:672-680: f_ff[leg] = -rBody * f, f = float(q_soln[3 leg + axis]). */
void oracle_forces_to_body(const float r_body[9], const float f_world[12],
                           float f_ff[12]);

#ifdef __cplusplus
}
#endif
#endif
