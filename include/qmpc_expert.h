/*
 * qmpc_expert.h -- EXPERT knobs of libqmpc.so (companion of include/qmpc.h; same library, same ABI version).
 *
 * Nothing here is needed to use the solver.  Every setter below changes HOW the same unique minimiser is computed --
 * which instantiation, which path, how much memory -- and its default is the setting that measured best on MI355X
 * (DESIGN.md 0 lists the numbers).  The warm start across MPC cycles is kept here as well: it is
 * correct and tested, and every variant of it measured SLOWER than the cold solve; it is off
 * unless a buffer is set.
 */
#ifndef QMPC_EXPERT_H
#define QMPC_EXPERT_H

#include "qmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Decoupled path of the 128- and 192-row size classes (n_r > 96: all four feet down, dense random contact tables).
 * on (default): the robot's condensed Hessian is inverted by a sweep kernel that leaves the inverse in a work item
 * in global memory (128 / 288 KiB per item, of which the lower block triangle is written; a bounded pool allocated by
 * qmpc_setup, see qmpc_set_chunks), and the active set is run by a second kernel,
 * one robot per small workgroup -- the robots with the most rows violated at the unconstrained minimiser first --
 * with the rank-1 events of the method in the register file of helper waves, instead of one workgroup pinning a whole
 * CU for the whole solve.  Same unique minimiser.  A robot whose history outgrows the engine's registers and LDS
 * continues with the excess in an overflow pool in global memory (QMPC_ST_SPILLED, informational; 160 events per engine
 * workgroup); one that outgrows that as well, or the engine's working-set slots, is re-solved by the one-kernel path
 * (QMPC_ST_FALLBACK).  mode 1 (default): used by handles created for at least 384 robots (128-row class) / 128 robots
 * (192-row class) -- smaller batches are latency-bound and the one-kernel path has one launch less on the critical
 * path; it is the handle's max_batch that decides, never the size of a call, so that a robot's result does not depend
 * on the batch it is solved in (the two paths agree to ~1e-14 relative, not bit for bit); mode 2:
 * always; mode 0: the one-kernel path for every class (QMPC_NO_SPLIT=1 in the environment selects that at
 * qmpc_create).  The JCQP alternate and warm-started solves always take the one-kernel path. */
int qmpc_set_split(qmpc_handle h, int mode);

/* Block start of the decoupled path's engine -- EXPERIMENTAL, default off.  The rows of the friction pyramids / force
 * limits that are violated at the unconstrained minimiser are, almost without exception, active at the solution; with
 * on != 0 the engine adds such candidate sets (one row per stance foot-step and round, up to four rounds) as forced
 * additions made by all threads of the workgroup with the records in LDS, removes the rows whose multiplier came out
 * negative and hands a valid Goldfarb-Idnani state to the normal iteration.  Same unique minimiser (tested), but as
 * measured on MI355X not faster than the iteration it replaces: ~3.1 k cycles per forced change against
 * ~4.9 k per iteration, and 20 % more changes.  `iters` counts every forced change like an iteration. */
int qmpc_set_block_start(qmpc_handle h, int on);

/* The 64-row size class has a second instantiation sized for FIVE workgroups per CU (96 VGPRs, a 16-event pool in LDS
 * instead of 28): a launch of several rounds of workgroups is bound by instruction issue, and a fifth wave per SIMD fills
 * the slots the other four leave (trot: +3.5 % at 2048 robots, +8 % at 4096, +14 % from 8192 on); a single round (1024 robots)
 * is bound by its slowest robot and would lose 1 - 11 %.  mode 1 (default): used by handles created for at least 2048 robots when the 64-row class is the
 * whole chain (qmpc_set_max_stance says every robot fits it) -- the handle's size decides, never a call's -- and, with larger
 * classes behind it in the chain as well (configs[4] + 2 %; its robots' overflow-pool slices are recycled within a call);
 * mode 0: never; mode 2: whenever the chain is that class alone.  Same arithmetic: bit-identical results (tested) -- with one
 * documented corner: the Schur-form FALLBACK engine of this instantiation (reached only when the fast engine runs
 * out of its 32 working-set slots, or numerically loses definiteness) shares a smaller LDS pool and holds 56 working-set slots at
 * n_r = 63, 59 at n_r = 60, all 64 up to n_r = 54, where the four-per-CU instantiation holds 64 throughout; a robot that needed
 * more -- nearly every variable pinned by an active row, i.e. every foot-step at a vertex of its friction pyramid and the force
 * limit at once -- would read QMPC_ST_WS_FULL here and be solved there.  Which instantiation runs is a property of the HANDLE
 * (its max_batch and stance hints), never of a call's size (since round 5 also for chains with larger classes behind). */
int qmpc_set_dense(qmpc_handle h, int mode);

/* Size order / proxy staging (default on).  A launch of several rounds of workgroups ends with whichever long robot
 * started last, a launch of one round with whichever robot iterates longest.  When no order hint is usable (first call, another
 * batch size, qmpc_set_order_hint off) the library schedules by what THIS call's input records say about a robot's cost: its
 * reduced size (3 x stance foot-steps: the sweep is that many half-steps long) and a score that follows the active-set iteration
 * count with a correlation of ~0.7 on the BASELINE workloads (0.2 - 0.5 on closed-loop rollouts it was not fitted on): the tracking
 * error the coasting state would have at the end of the horizon, times the early stance foot-steps, plus a term for a long first
 * support phase that cannot balance gravity's moment without friction near its limit (pacing, bounding).
 *   - several rounds: from the second round on (and within the last five rounds of the launch) the first class takes the robots that
 *     FIT it largest first, highest score first among equals; robots it only hands on keep to their own places and are ordered among
 *     themselves by the score (the next class's queue is filled in dispatch order).  The permutation is built inside the launch
 *     by its first workgroups while the first rounds are solved -- no kernel in front of the call, no host work;
 *   - one round with full CUs: the workgroups that share a CU post their score with one atomic maximum on the CU's word; the one
 *     whose entry stands keeps the highest issue priority through its sweep.
 * Scheduling only: bit-identical results (tested).  Not used by command mode (the contact table is generated in the kernel),
 * captured calls, the JCQP alternate, or a contact-table pointer that is not 8-byte aligned.  on = 0: robot = workgroup index,
 * everybody yields alike. */
int qmpc_set_size_order(qmpc_handle h, int on);

/* Work items are a BOUNDED pool per class -- min(max_batch, 4096 / 3072 / 1024) items of 128 KiB / 288 KiB / 1.53 MiB for
 * the 128-row class / 192-row class / large problems -- whatever max_batch is; a call with more robots than items runs the
 * class as consecutive chunks (producer kernel, engine kernel, producer kernel, ...) on the caller's stream, the pool reused
 * from chunk to chunk.  Results do not depend on the chunking (tested).  qmpc_set_chunks (test hook): at least n chunks
 * (0 / 1 = as few as the pool allows; at most 64). */
int qmpc_set_chunks(qmpc_handle h, int n);

/* Every device allocation of a handle happens in qmpc_create, qmpc_setup (what the horizon makes reachable), the stance
 * hints, qmpc_set_split and here -- NEVER inside a solve call: qmpc_solve / qmpc_solve_commands only enqueue kernels on the
 * caller's stream (no memset nodes either: they replay wrongly with ROCm 7.2), so they can be captured into a hipGraph and replayed
 * (tests/test_gpu_parity.py::test_solve_is_graph_capturable).  qmpc_reserve repeats the allocation step for the current
 * setup; it is implied by qmpc_setup and kept for callers of earlier versions. */
int qmpc_reserve(qmpc_handle h);

/* Warm start across MPC cycles (the reference cold-starts every solve,
 * SolverMPC.cpp:529-541).  ws_dev[max_batch][QMPC_WS_SLOTS] is a DEVICE buffer the caller keeps
 * between cycles, initialised to -1.  While it is set, every solve (a) reads robot b's previous
 * working set from row b -- global constraint ids 5 * (4 * step + foot) + type, type 0..3 = the
 * friction-pyramid rows of f_block (SolverMPC.cpp:366-370), 4 = fz <= f_max; -1 = empty -- slides
 * it by `shift_steps` horizon steps (1 when the contact table advanced by one MPC step since the
 * last solve; entries that fall off the front or land on a swing foot-step are discarded), adds
 * those constraints first without search, drops the ones whose multiplier comes out negative, and
 * continues with the normal dual active-set iteration; (b) writes the final working set back.
 * The result is the same unique minimiser as a cold solve (the QP is strictly convex); only the
 * iteration path is shorter.  Row order must follow the robots (row b belongs to robot b of every
 * call).  NULL switches warm starting off.  Warm-started solves take the one-kernel path in every size class (the
 * decoupled engine of the 128- / 192-row classes starts cold); qmpc_solve_commands always starts cold (warm starting is
 * wired into the record entry points). */
#define QMPC_WS_SLOTS 64
int qmpc_set_warm_start(qmpc_handle h, int32_t* ws_dev, int shift_steps);

/* Selective warm start: with min_iters > 0 only the robots that needed at least min_iters active-set
 * iterations in the handle's previous call (the counts the order hint keeps, qmpc_set_order_hint must be on) read their
 * previous working set; all others start cold -- a launch waits for its hardest robot, and the easy majority only pays for
 * wrong guesses.  0 (default): every robot starts warm while a buffer is set.  Same unique minimiser either way.
 * The counts exist while the order hint is on (QMPC_ERR_STATE otherwise), for eager calls, from the handle's previous call of
 * the SAME batch size; a call without them -- the first one, another batch size, a call captured into a hipGraph -- starts
 * every robot cold.
 * Measured on closed-loop rollouts (profiles/r05_b_warm_select.txt): NOT faster -- the launch's maximum iteration count
 * goes UP with a warm start (17 -> 22, 28 -> 38), whoever else starts cold; kept as an option, off by default. */
int qmpc_set_warm_start_min_iters(qmpc_handle h, int min_iters);

#ifdef __cplusplus
}
#endif
#endif
