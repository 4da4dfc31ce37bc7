/*
 * qmpc.h -- C ABI of the MI355X-native batched convex-MPC solver.
 *
 * This is the drop-in boundary for the MPC hot path of
 * Derek-TH-Wang/quadruped_ctrl.  Each entry point names the reference
 * interface it replaces (paths relative to the reference repository root).
 * Plain C: opaque handle, plain pointers and sizes, int return codes, no
 * exceptions, no globals, no framework types.  The single-robot reference
 * symbols of src/MPC_Ctrl/convexMPC_interface.h:40-48 are provided on top of
 * this ABI by include/convexMPC_interface.h (libconvexmpc_shim.so).
 *
 * Memory layout (all arrays one row per robot, row-major, batch-major):
 *   p[B][3] v[B][3] q[B][4] (w,x,y,z)  w[B][3]
 *   r[B][12]   axis-major foot offsets, r[axis*4 + foot]
 *              (src/MPC_Ctrl/RobotState.cpp:25-27)
 *   yaw[B]
 *   traj[B][12*h]   reference trajectory, 12 states per horizon step
 *   gait[B][4*h]    u8 contact table, gait[step*4 + foot]
 *                   (src/MPC_Ctrl/Gait.cpp:142-166)
 *   weights[12] or [B][12], alpha[1] or [B], x_drag[1] or [B]
 *                   (stride 0 = shared by the batch)
 * Outputs:
 *   grf[B][12]      float, world-frame ground reaction force of foot f at
 *                   horizon step 0: grf[3*f + axis]  == get_solution(0..11)
 *   soln[B][12*h]   optional double, the whole q_soln (zeros on swing feet,
 *                   src/MPC_Ctrl/SolverMPC.cpp:545-557)
 *   status[B]       per-robot status bits (QMPC_ST_*), 0 = solved
 *   iters[B]        optional, active-set iterations used
 * Every output row of every robot is written by every call: a robot that is not
 * solved (QMPC_ST_WS_FULL, QMPC_ST_INFEASIBLE) reads zero forces -- never a previous
 * call's values and never an abandoned, primal-infeasible iterate -- and in command
 * mode its controller state is advanced like everybody else's.
 *
 * Streams: a handle owns device state (coefficient tables, size-class work lists and
 * their counters) that ONE stream at a time orders.  Calls on the same stream need
 * nothing; a call that arrives on a different stream than the handle's previous call
 * is made to wait on the device (event + hipStreamWaitEvent, no host block) for that
 * previous stream's work, and qmpc_setup waits for solves in flight before it rewrites
 * the tables.  For independent batches in flight at the same time use one handle per
 * stream.  A handle is not thread-safe: serialise calls per handle.
 */
#ifndef QMPC_H
#define QMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMPC_MAX_HORIZON 36 /* K_MAX_GAIT_SEGMENTS (src/MPC_Ctrl/convexMPC_interface.h:3).  The reference's own gaits
                               use 10 .. 16 segments (ConvexMPCLocomotion.cpp:25,196,204): every size class takes those.
                               Longer horizons (17 .. 36) are assembled by the 192-row class (robots with at most 64
                               stance foot-steps in the horizon, n_r = 3 x stance foot-steps <= 192: trot up to 32
                               segments, gaits with a duty factor <= 0.44 up to 36) and, beyond that, by the
                               LARGE-PROBLEM path (n_r up to 432 = all four feet down for 36 segments: the Hessian in
                               global memory, a block sweep, a seven-block engine; 1e5 .. 4e5 QP solves/s instead of ~1e6 --
                               coverage of the interface, not a fast path; 1.6 MiB of device memory per robot of the
                               handle's max_batch, allocated on the first call at such a horizon) */
#define QMPC_LONG_HORIZON 16 /* horizons above this take the long-horizon routes described above */

/* return codes */
#define QMPC_OK 0
#define QMPC_ERR_ARG 1     /* bad argument (null, size, horizon > max) */
#define QMPC_ERR_DEVICE 2  /* HIP runtime error (see qmpc_last_error) */
#define QMPC_ERR_STATE 3   /* call order (solve before setup) */

/* per-robot status bits */
#define QMPC_ST_MAXITER 1    /* active-set iteration limit reached */
#define QMPC_ST_NOT_PD 2     /* condensed Hessian not positive definite */
#define QMPC_ST_INFEASIBLE 4 /* constraints inconsistent (cannot happen for
                                friction pyramids with f_max >= 0) */
#define QMPC_ST_WS_FULL 8    /* working-set capacity exceeded */
#define QMPC_ST_FALLBACK 16  /* informational, NOT an error: the robot was solved by a slower engine than
                                its class's first choice: the Schur-form engine (the fast engine's working-set
                                slots were exhausted), or the one-kernel path after the decoupled path's engine
                                kernel handed it back (more rank-1 events than its registers hold) */
#define QMPC_ST_NONFINITE 32 /* the result contains NaN / Inf (non-finite input) */
#define QMPC_ST_COMPACTED 64 /* informational, NOT an error: the fast engine ran out of pool with
                                constraints that had entered and left the working set again, and
                                rebuilt its records from the current working set (no second solve) */
#define QMPC_ST_SPILLED 128  /* informational, NOT an error: the fast engine's on-chip pool was full and the
                                robot continued on a slice of the handle's overflow pool in global memory (the
                                decoupled path's engine: on its workgroup's slice of that kernel's own pool) */
#define QMPC_ST_ERROR_MASK (15 | 32)

typedef struct qmpc_ctx* qmpc_handle;

/* Batched inputs.  Device pointers for qmpc_solve, host pointers for
 * qmpc_solve_host.  Replaces struct update_data_t
 * (src/MPC_Ctrl/convexMPC_interface.h:21-38). */
typedef struct {
  const float* p;
  const float* v;
  const float* q;
  const float* w;
  const float* r;
  const float* yaw;
  const float* traj;
  const uint8_t* gait;
  const float* weights;
  const float* alpha;
  const float* x_drag;
  int weights_stride; /* 0 (shared) or 12 */
  int alpha_stride;   /* 0 or 1 */
  int x_drag_stride;  /* 0 or 1 */
} qmpc_inputs;

typedef struct {
  float* grf;      /* [B][12]  required */
  double* soln;    /* [B][12h] optional (NULL to skip) */
  int32_t* status; /* [B]      required */
  int32_t* iters;  /* [B]      optional */
} qmpc_outputs;

/* Create a solver bound to HIP device `device_id`.  Allocates the per-call
 * work lists for up to max_batch robots and horizons up to max_horizon
 * (<= QMPC_MAX_HORIZON).  Replaces the file-scope globals of
 * src/MPC_Ctrl/convexMPC_interface.cpp:13-20 and SolverMPC.cpp:18-57. */
int qmpc_create(int device_id, int max_batch, int max_horizon,
                qmpc_handle* out);
int qmpc_destroy(qmpc_handle h);

/* Replaces setup_problem(dt, horizon, mu, f_max)
 * (src/MPC_Ctrl/convexMPC_interface.cpp:42-66).  dt, mu and f_max are
 * rounded to float exactly as struct problem_setup stores them. */
int qmpc_setup(qmpc_handle h, double dt, int horizon, double mu, double f_max);

/* Robot constants the reference hard-codes: mass 9 (RobotState.h:23),
 * I_body diag(.07,.26,.242) (RobotState.cpp:38), gravity state -9.8
 * (SolverMPC.cpp:318).  Optional; those literals are the defaults. */
int qmpc_set_robot(qmpc_handle h, double mass, const double ibody_diag[3],
                   double gravity);

/* Replaces update_solver_settings (convexMPC_interface.cpp:107-119): the
 * reference's JCQP knobs have no meaning for the exact active-set solve;
 * what remains is the iteration cap (the role of nWSR=100,
 * SolverMPC.cpp:435) and the constraint-violation tolerance [N]. */
int qmpc_settings(qmpc_handle h, int max_iter, double tol);

/* The reference's SPARSE formulation: SparseCMPC (src/MPC_Ctrl/SparseCMPC.cpp:31-73) keeps
 * the 12 states of every horizon step as variables, writes the dynamics as equality rows and hands the
 * sparse QP to OSQP (OsqpTriples.cpp:57-142).  Its discrete model differs from the dense path's:
 * A_d = expm(A dt) = I + A dt (A^2 = 0 without the gravity state), B_d = B dt (SparseCMPC_Math.cpp:25 --
 * not the matching block of the matrix exponential), gravity -9.81 added as g dt to the velocity rows once
 * per step (:41-42, :214, :268).  Eliminating the states through those equality rows gives a condensed QP
 * with the same block structure as the dense path's -- A_d^d B_d = dt B + d dt^2 A B, i.e. only the second
 * coefficient family and the free response change -- so QMPC_MODEL_SPARSE solves SparseCMPC's QP with the
 * same closed-form assembly and the same exact solver, and returns its exact minimiser (the reference's
 * OSQP run stops at eps = 1e-5, a few per cent from it on the small force components).  Set the gravity
 * with qmpc_set_robot(..., -9.81) and pass SparseCMPC's own mu / weights / alpha (initSparseMPC,
 * ConvexMPCLocomotion.cpp:732-756: mu 1.0, weights 0.25 0.25 10 2 2 20 0 0 0.3 0.2 0.2 0.2); x_drag
 * is ignored by this model; yaw is the quaternion's (SparseCMPC.cpp:99).  Horizons up to
 * QMPC_MAX_HORIZON like every other entry point (the long horizons that motivate a sparse solver on the
 * CPU are not what the reference uses it with: solveSparseMPC runs horizonLength = 10..16). */
#define QMPC_MODEL_DENSE 0
#define QMPC_MODEL_SPARSE 1
int qmpc_set_model(qmpc_handle h, int model);

/* The reference's alternate solver: update_solver_settings(..., use_jcqp) with
 * use_jcqp = 1 or 2 makes solve_mpc run JCQP's QpProblem<double>::runFromDense
 * (src/JCQP/QpProblem.cpp:178-269) -- an OSQP-style ADMM from a cold start, stopped when
 * (|A x - z|_inf + |P x + q + A^T y|_inf) / 4 < terminate (checked every 10 iterations) or after
 * max_iter iterations -- on the full 12h-variable problem (1, SolverMPC.cpp:400-414) or on the
 * swing-eliminated one (2, :558-610).  Its result is an APPROXIMATION of the minimiser (about 1e-3
 * relative with the caller's settings rho = 1e-7, sigma = 1e-8, alpha = 1.5, terminate = 0.1,
 * ConvexMPCLocomotion.cpp:644-648).  With use_jcqp != 0 qmpc_solve / qmpc_solve_host reproduce THAT
 * iteration on the GPU (same updates, same stopping rule; iters = ADMM iterations, QMPC_ST_MAXITER when
 * the residual test never passed); use_jcqp = 0 (default) is the exact active-set solve, which is what
 * the reference's qpOASES path returns.  qmpc_solve_commands always solves exactly.
 * At horizons above QMPC_LONG_HORIZON the alternate runs like the exact solve does: reduced sizes up to 192 variables in
 * the 192-row class's ADMM instantiation, larger ones -- every robot with use_jcqp = 1 (12 h variables), all feet down or
 * a trot beyond 32 segments with use_jcqp = 2 -- through the large-problem producer (which leaves
 * (P + sigma I + A^T R A)^-1 and the gradient in the robot's work item) and an ADMM kernel of its own
 * (qmpc_admm_big_kernel, one mat-vec over the 448 x 448 item per iteration: coverage of the interface, ~1e4 robots/s). */
int qmpc_settings_jcqp(qmpc_handle h, int use_jcqp, int max_iter, double rho, double sigma,
                       double solver_alpha, double terminate);

/* Optional size hint.  The kernels are specialised by reduced problem size
 * n_r = 3 * (stance foot-steps in the horizon) <= 64 / 96 / 128 / 192; without a
 * hint every class that the horizon allows is launched (the unused ones exit
 * at once).  A caller that knows its contact tables (e.g. trot: 2 feet x h)
 * states the bound and the larger classes are skipped (at horizons above 16: the large-problem stage behind the
 * 192-row class -- three launches per call that normally find nothing to do -- when the bound is at most 64); a robot that
 * exceeds it is reported with QMPC_ST_WS_FULL instead of being solved.  0 = no hint. */
int qmpc_set_max_stance(qmpc_handle h, int max_stance_footsteps);
/* Companion lower bound: when every robot has at least this many stance foot-steps, the
 * classes that are too small for all of them are not launched either (e.g. trot at
 * horizon 16: 32 foot-steps, n_r = 96 -> the 64-row class would only hand every robot on).
 * A robot below the bound is still solved correctly.  0 = no hint. */
int qmpc_set_min_stance(qmpc_handle h, int min_stance_footsteps);

/* Order hint.  A launch of several rounds of workgroups ends with whichever hard robot started last.  A controller solves
 * the SAME robots every MPC cycle, and a robot that needed many active-set iterations 26 ms ago needs many now: with
 * mode 1 (default) every one-kernel solve leaves its iteration count in a per-handle array, and a call of the same batch
 * size whose first size class is launched over more robots than it has resident workgroups takes the robots in the order
 * of those counts, longest first (the permutation is built inside the launch by its first workgroups since round 6: no kernel in
 * front of the call); a launch of ONE round
 * (everybody starts at once) uses them as issue priority instead: the few robots the previous call found hardest keep the
 * highest priority through their Gauss-Jordan sweep, so the launch no longer waits for them.  Scheduling only:
 * a robot's result does not depend on its place (bit-identical, tested; until round 4 there was one exception -- more robots of a
 * call outgrowing their on-chip event pool than the handle had overflow slices, 2048, at 65 536 mixed-gait robots on one GPU:
 * WHICH of them took the Schur-form fallback depended on the order they ran in -- closed since the slices are recycled within
 * a call); a stale or meaningless hint -- other robots in
 the same rows -- costs nothing but the benefit.  mode 0: off (robot = workgroup index).  Calls captured into a hipGraph
 * and the JCQP alternate do not use it.  Without usable counts (first call, another batch size, mode 0) the same machinery orders
 * a launch by what this call's own records say -- contact-table size and a tracking-error proxy --, see qmpc_set_size_order in
 * qmpc_expert.h (default on; scheduling only as well). */
int qmpc_set_order_hint(qmpc_handle h, int mode);

/* Everything else the library exports lives in two companion headers, so that this one is the surface a caller needs:
 *   include/qmpc_expert.h -- scheduling / memory knobs whose DEFAULT is the measured optimum (qmpc_set_split, qmpc_set_dense,
 *                            qmpc_set_chunks, qmpc_reserve, qmpc_set_block_start) and the warm start across MPC cycles, which is
 *                            correct but measured slower than the cold solve (qmpc_set_warm_start, _min_iters);
 *   include/qmpc_debug.h  -- test and profiling hooks (qmpc_set_debug_*, qmpc_debug_*): used by tests/ and tools/ only.
 * Same shared library, same ABI version. */

/* Solve `batch` independent MPC problems.  All pointers are DEVICE pointers
 * valid on the handle's device; the call only enqueues work on `stream`
 * (a hipStream_t, NULL = default stream) and returns; results are readable
 * once the stream has been synchronised.  Replaces the blocking
 * update_problem_data_floats(...) -> solve_mpc(...) -> get_solution(i)
 * sequence (convexMPC_interface.cpp:121-180, SolverMPC.cpp:296-639). */
int qmpc_solve(qmpc_handle h, int batch, const qmpc_inputs* in,
               const qmpc_outputs* out, void* stream);

/* Same with HOST pointers; returns when the results are in the caller's arrays.  The record is
 * gathered into one pinned, device-visible block: up to 64 robots the kernel reads and writes that
 * block in place over PCIe (no copy commands -- one launch, one event wait: the latency path of the
 * single-robot reference shim); larger batches use one H2D and one D2H copy. */
int qmpc_solve_host(qmpc_handle h, int batch, const qmpc_inputs* in,
                    const qmpc_outputs* out);

/* One host thread, several devices (SURVEY.md 8e / build plan step 4): `batch` robots given by
 * HOST pointers are split into contiguous shards of ceil(batch / n_handles) robots, shard k is
 * enqueued on handles[k] (its own device, its own stream: gather into that handle's pinned block,
 * H2D, solve, D2H), and only after every device has been given its work are the results collected
 * into the caller's arrays -- the devices run concurrently and no data-path collective is involved
 * (robots are independent).  Every handle must have been set up with the same problem; several
 * handles may share a device.  Returns the first error. */
int qmpc_solve_sharded(const qmpc_handle* handles, int n_handles, int batch,
                       const qmpc_inputs* in, const qmpc_outputs* out);

/* ---------------------------------------------------------------------------
 * Caller side of the solve, batched (SURVEY.md row a12 and the consumer of
 * a11).  One row per robot in every array, DEVICE pointers.
 *
 * qmpc_command: everything ConvexMPCLocomotion::updateMPCIfNeeded
 * (src/MPC_Ctrl/ConvexMPCLocomotion.cpp:498-577) and ::solveDenseMPC
 * (:592-640) read from the state estimate and from the controller's members.
 */
typedef struct {
  /* StateEstimate (seResult, :502 / :594) */
  const float* position;    /* [B][3] */
  const float* v_world;     /* [B][3] */
  const float* omega_world; /* [B][3] */
  const float* orientation; /* [B][4] w,x,y,z */
  const float* rpy;         /* [B][3] */
  const float* r_body;      /* [B][9] rBody, row-major */
  const float* p_foot;      /* [B][12] pFoot[leg][axis] (world): p_foot[3*leg + axis] */
  /* controller members */
  const float* vel_des;      /* [B][3] _x_vel_des, _y_vel_des, _yaw_turn_rate */
  const float* yaw_des_true; /* [B] */
  const float* rpy_comp;     /* [B][2] */
  const float* stand_traj;   /* [B][6]; may be NULL when no robot stands */
  const float* rp_des;       /* [B][2] _roll_des, _pitch_des; NULL = zeros */
  const int32_t* gait_type;  /* [B] current_gait, 4 = standing (:514); NULL = none stands */
  const int32_t* gait_offsets;   /* [B][4] OffsetDurationGait::_offsets   (MPC segments) */
  const int32_t* gait_durations; /* [B][4] OffsetDurationGait::_durations */
  const int32_t* gait_iteration; /* [B]    OffsetDurationGait::_iteration (Gait.cpp:189) */
  float* world_position_desired; /* [B][2] in/out (:535-545) */
  float* x_comp_integral;        /* [B]    in/out (:632-640) */
  float body_height;             /* _body_height */
  int omni_mode;                 /* omniMode (:507) */
} qmpc_command;

/* Writable view of the arrays qmpc_inputs points at (the update_data_t record). */
typedef struct {
  float* p;
  float* v;
  float* q;
  float* w;
  float* r;
  float* yaw;
  float* traj;
  uint8_t* gait;
  float* x_drag;  /* [B] */
  float* weights; /* [B][12] or NULL (caller keeps its own) */
  float* alpha;   /* [B]     or NULL */
} qmpc_record;

/* Build the MPC input record of `batch` robots on the GPU: reference
 * trajectory trajAll and the desired-position clamp (:534-576), standing
 * trajectory (:514-531), r = pFoot - position (:611-613), yaw, Q and alpha
 * (:598-604), x_drag = x_comp_integral followed by its integrator step
 * (:632-640), and the contact table of OffsetDurationGait::getMpcTable
 * (src/MPC_Ctrl/Gait.cpp:142-166) with n_segments = horizon.  dtMPC is the dt
 * given to qmpc_setup.  Enqueues on `stream`; feed `rec`'s arrays to
 * qmpc_solve through a qmpc_inputs with strides 12 / 1 / 1. */
int qmpc_pack(qmpc_handle h, int batch, const qmpc_command* cmd,
              const qmpc_record* rec, void* stream);

/* f_ff[B][12]: f_ff[3*leg + i] = (-rBody * f_leg)[i] with f_leg = grf[3*leg..]
 * (ConvexMPCLocomotion.cpp:672-680, the consumer of get_solution). */
int qmpc_forces_to_body(qmpc_handle h, int batch, const float* r_body,
                        const float* grf, float* f_ff, void* stream);

/* The three calls above fused into ONE launch: the record is generated inside
 * the solve kernel's first stage from `cmd` (never written to memory), the
 * controller state in `cmd` is updated by the workgroup that solves the robot,
 * and, when f_ff is non-NULL, the body-frame forces are written next to grf.
 * Q and alpha are the reference's literals (:598, :604).  Results are
 * bit-identical to qmpc_pack -> qmpc_solve -> qmpc_forces_to_body. */
int qmpc_solve_commands(qmpc_handle h, int batch, const qmpc_command* cmd,
                        const qmpc_outputs* out, float* f_ff, void* stream);

/* ---------------------------------------------------------------------------
 * Per-tick glue either side of the solve, batched.  One row per robot in every
 * array, DEVICE pointers, float arithmetic written operation by operation after the reference
 * (only the libm calls -- sin / cos / atan2 / sqrt -- may differ from the host's by an ulp).
 * Leg order 0..3 = FR, FL, RR, RL with side signs -1, +1, -1, +1 (src/Dynamics/Quadruped.h:85-89);
 * joint order abad, hip, knee; everything in the leg's hip frame like LegControllerData.
 */

/* Leg link lengths; defaults are the Mini Cheetah's (src/Dynamics/MiniCheetah.h:31-37):
 * abad 0.062, hip 0.209, knee 0.195, knee Y offset 0.004. */
int qmpc_set_leg_geometry(qmpc_handle h, double abad_link, double hip_link, double knee_link,
                          double knee_link_y_offset);

/* LegController::updateData (src/Controllers/LegController.cpp:89-110): for every leg the foot
 * position p[B][12] and Jacobian J[B][4][9] (row-major 3x3) of computeLegJacobianAndPosition
 * (:204-244) from the joint angles q[B][12] (q[3*leg + joint]), and, when v is non-NULL, the foot
 * velocity v = J * qd (:108). */
int qmpc_leg_kinematics(qmpc_handle h, int batch, const float* q, const float* qd, float* J,
                        float* p, float* v, void* stream);

/* What LegController::updateCommand (:116-160) reads. */
typedef struct {
  const float* tau_ff;   /* [B][12] commands[leg].tauFeedForward   (NULL = zeros) */
  const float* force_ff; /* [B][12] commands[leg].forceFeedForward (NULL = zeros) -- the f_ff of the MPC */
  const float* kp_cart;  /* [B][4][9] kpCartesian, row-major Mat3 per leg */
  const float* kd_cart;  /* [B][4][9] kdCartesian */
  const float* p_des;    /* [B][12] commands[leg].pDes */
  const float* v_des;    /* [B][12] commands[leg].vDes */
  const float* q;        /* [B][12] datas[leg].q  */
  const float* qd;       /* [B][12] datas[leg].qd */
  const float* J;        /* [B][4][9] datas[leg].J (qmpc_leg_kinematics) */
  const float* p;        /* [B][12] datas[leg].p */
  const float* v;        /* [B][12] datas[leg].v */
  float kp_joint;        /* crtlParam(2) (GaitCtrller.cpp:15,128) */
  float kd_joint;        /* crtlParam(3) */
} qmpc_leg_command;

/* LegController::updateCommand: footForce = forceFF + Kp (pDes - p) + Kd (vDes - v);
 * legTorque = tauFF + J^T footForce; tau[B][12] (tau[3*leg + joint] == JointEff::eff,
 * GaitCtrller.cpp:130-136) = kp_joint (0 - q) - kd_joint qd + legTorque.  q_des[B][12]
 * (optional) = computeLegIK(pDes) (:255-285). */
int qmpc_leg_torques(qmpc_handle h, int batch, const qmpc_leg_command* cmd, float* tau,
                     float* q_des, void* stream);

/* FootSwingTrajectory::computeSwingTrajectoryBezier (src/Controllers/FootSwingTrajectory.cpp:17-37)
 * for n_feet independent swing feet: start p0[n][3], landing pf[n][3], apex height[n], swing phase[n]
 * in [0, 1], swing_time[n] seconds -> position p, velocity v, acceleration a ([n][3] each). */
int qmpc_swing_trajectory(qmpc_handle h, int n_feet, const float* p0, const float* pf,
                          const float* height, const float* phase, const float* swing_time,
                          float* p, float* v, float* a, void* stream);

/* LinearKFPositionVelocityEstimator (src/Controllers/PositionVelocityEstimator.cpp:18-221): the
 * 18-state (body position, velocity, four foot positions) / 28-measurement Kalman filter of the state
 * estimator, one step per call for every robot.  The filter state lives in caller-owned DEVICE arrays. */
typedef struct {
  float* xhat;                /* [B][18]  _xhat, in/out */
  float* P;                   /* [B][324] _P (18x18 row-major), in/out */
  const float* r_body;        /* [B][9]   result->rBody */
  const float* a_world;       /* [B][3]   result->aWorld */
  const float* omega_body;    /* [B][3]   result->omegaBody */
  const float* contact_phase; /* [B][4]   result->contactEstimate */
  const float* leg_p;         /* [B][12]  legControllerData[leg].p (qmpc_leg_kinematics) */
  const float* leg_v;         /* [B][12]  legControllerData[leg].v */
  float* position;            /* [B][3]   out: result->position */
  float* v_world;             /* [B][3]   out: result->vWorld */
  float* v_body;              /* [B][3]   out: result->vBody (may be NULL) */
} qmpc_kf_state;
/* setup() (:18-63): xhat = 0, P = 100 I. */
int qmpc_kf_init(qmpc_handle h, int batch, float* xhat, float* P, void* stream);
/* run() (:66-221).  Hip locations are the Mini Cheetah's (MiniCheetah.h:25-26,105): (+-0.19, +-0.049, 0). */
int qmpc_kf_step(qmpc_handle h, int batch, const qmpc_kf_state* st, void* stream);

/* Last HIP error string for this handle ("" if none). */
const char* qmpc_last_error(qmpc_handle h);

/* Library/ABI version, bumped on any signature change. */
int qmpc_abi_version(void);
/* QMPC_MAX_HORIZON of the library that is loaded (for FFI callers that cannot read the macro). */
int qmpc_max_horizon(void);

#ifdef __cplusplus
}
#endif
#endif /* QMPC_H */
