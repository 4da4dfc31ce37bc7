/*
 * qmpc_debug.h -- TEST AND PROFILING HOOKS of libqmpc.so (companion of include/qmpc.h; same library, same ABI version).
 *
 * Used by tests/ and tools/ only: dumps of intermediate results, ways of forcing the rare paths (overflow pool, Schur-form
 * fallback, hand-back of the decoupled engine), shader-clock stamps, the handle's device counters.  Not part of the
 * drop-in boundary (INTEGRATION.md does not advertise them); they may change between ABI versions without notice.
 */
#ifndef QMPC_DEBUG_H
#define QMPC_DEBUG_H

#include "qmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook.  The 96-row class's solve kernels are launched with eight waves of which six stay, chosen so that the two
 * workgroups of a CU load its four SIMDs evenly.  mode 1: every workgroup makes the same choice (no per-CU
 * slot word); mode 2: the fallback "waves 0..5 stay"; mode 0 (default): balanced.  Results are bit-identical in all three. */
int qmpc_set_debug_balance(qmpc_handle h, int mode);

/* Test hook: when non-NULL, the next qmpc_solve calls also store the
 * assembled reduced QP of every robot (before the solve) into DEVICE
 * buffers H[B][ld*ld], g[B][ld] (doubles, row-major, ld = qmpc_debug_ld();
 * entries beyond n_r are padding).  Pass NULLs to switch off. */
int qmpc_set_debug(qmpc_handle h, double* H_dev, double* g_dev);

int qmpc_debug_ld(qmpc_handle h);

/* Test hook: DEVICE buffer aux[B][8] receiving, per robot, the float transcendentals exactly as
 * the kernel evaluated them -- cos(yaw), sin(yaw) (RobotState.cpp:30-35) and roll, pitch, yaw of
 * quat_to_rpy (SolverMPC.cpp:257-267) -- so that a test can separate "same libm bits" from
 * "same algebra" when it compares the assembled QP with an fp64 model.  NULL = off. */
int qmpc_set_debug_aux(qmpc_handle h, double* aux_dev);

/* Test hook: use only the first n slices (0 <= n <= min(max_batch, 2048)) of the handle's overflow event pool -- the
 * global-memory records a robot continues on when its on-chip event pool is full (QMPC_ST_SPILLED).  The slices are
 * RECYCLED within a call (one flag per slice, released when its robot finishes), so a handle's 2048 slices serve calls of
 * any size: the need is bounded by the robots in flight; a robot that finds every slice taken waits for one.  With n = 0,
 * or when the wait times out (qmpc_set_debug_overflow_spin), the robot is re-solved by the Schur-form engine
 * (QMPC_ST_FALLBACK).  Negative n restores the default. */
int qmpc_set_debug_overflow_slices(qmpc_handle h, int n);

/* Test hook: probes a robot makes for a free overflow slice before it gives up (default 2^22; negative restores it). */
int qmpc_set_debug_overflow_spin(qmpc_handle h, int probes);

/* Test hook: the decoupled path's engine kernel may hold at most n rank-1 events per robot (0 = its compiled
 * capacity); a robot that needs more is handed back to the one-kernel path (QMPC_ST_FALLBACK). */
int qmpc_set_debug_engine_events(qmpc_handle h, int n);

/* Test hook: on != 0 makes every slice of the 192-row class's global event pool look taken, so that every
 * workgroup of that class times out waiting for one: its robots must then be solved by the Schur-form engine
 * (QMPC_ST_FALLBACK set, same answer) instead of proceeding on a slice they do not own. */
int qmpc_set_debug_pool_busy(qmpc_handle h, int on);

/* Test hook: copies work item `item` of the decoupled path (which: 0 = 128-row class, 1 = 192-row class, 2 = large problems)
 * to the host after a solve: the inverse (ld x ld doubles, ld = 128 / 192 / 448; the 128- / 192-row classes write the lower block
 * triangle only), x_u (ld doubles), {rid, n, nst, status bits} (4 ints).  Any pointer may be NULL. */
int qmpc_debug_read_item(qmpc_handle h, int which, int item, double* hinv_host, double* xu_host, int* hdr4);

/* Test hook: the handle's three counter sets (3 x 256 ints: two ping-ponged by eager calls, one for captured calls) -> host. */
int qmpc_debug_read_counts(qmpc_handle h, int* host768);

/* Profiling hook: DEVICE buffer clk[B][16] receiving shader-clock stamps at
 * the kernel's phase boundaries (NULL = off). */
int qmpc_set_debug_clock(qmpc_handle h, long long* clk_dev);

/* What the scheduling of DESIGN.md 13 reads from a batch's records, as the kernels evaluate it (qmpc_robot_keys): per robot the
 * stance foot-steps, the score that orders launches and the force-scaled demand that gates the one-round staging.  DEVICE
 * pointers [batch]; enqueued on `stream`.  tests/test_gpu_hardening.py pins it to the numpy statement of the formula. */
int qmpc_debug_keys(qmpc_handle h, int batch, const qmpc_inputs* in, int32_t* nst_dev, float* score_dev, float* demand_dev,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif
