// qmpc_device.h -- kernel parameter block shared by qmpc_kernels.hip (device)
// and qmpc_capi.cpp (host side of the C ABI in include/qmpc.h).
#ifndef QMPC_DEVICE_H
#define QMPC_DEVICE_H

#include <stdint.h>

// mirror of the QMPC_ST_* bits of include/qmpc.h
#define QMPC_DEV_ST_MAXITER 1
#define QMPC_DEV_ST_NOT_PD 2
#define QMPC_DEV_ST_INFEASIBLE 4
#define QMPC_DEV_ST_WS_FULL 8
#define QMPC_DEV_ST_FALLBACK 16
#define QMPC_DEV_ST_NONFINITE 32
#define QMPC_DEV_ST_COMPACTED 64
#define QMPC_DEV_ST_SPILLED 128

// warm start: working-set slots kept per robot between MPC cycles (QMPC_WS_SLOTS of qmpc.h)
#define QMPC_WS_STRIDE 64
// one slice of the overflow event pool: 96 events x (128 rows + 64 slots) doubles fit (class 2's record, the largest
// that spills; 96 x 256 allocated)
#define QMPC_OV_SLICE (96 * 256)
// ... and of the largest class's own pool: 160 events x (192 rows + 128 slots)
#define QMPC_EV_SLICE3 (160 * 320)

// per-call device counters, one set of QMPC_COUNTERS ints (two sets, ping-ponged between consecutive calls; the first
// kernel of a chain zeroes the NEXT call's set):
//   [0..2] list lengths of classes 4, 2, 3   [3] list length of the large problems (horizons > 16, n_r > 192)
//   [4..6] queue heads of the one-kernel list consumers   [7] overflow-pool slices handed out   [16] / [17] probes that
//   found a slice taken / robots that timed out waiting for one (QMPC_CNT_OV_*)
//   [8 + sk] robots the engine kernel of item class sk hands back (sk = 0: 128-row class, 1: 192-row class, 2: large
//   problems)   [12 + sk] queue heads of the launches that take them
//   [QMPC_CNT_GRP(sk, g)], g = 0 / 1: one GROUP of counters per chunk in flight of item class sk.  The work items of a class
//   are a bounded pool (qmpc_capi.cpp: ensure_pools); a call with more robots than items runs the class as consecutive
//   CHUNKS on the caller's stream -- sweep kernel, engine kernel, sweep kernel, ... -- chunk ch on group ch & 1; the engine
//   kernel of chunk ch zeroes the group chunk ch + 1 is going to use (its previous user, chunk ch - 1, has finished):
//   [+0] work items produced   [+1] engine queue head   [+2] the sweep kernel's list queue head
//   [+8 + b] work items in order bucket b (hardest robots first: the engine workgroups take the items bucket by bucket --
//   a launch ends with its slowest robot, which must not be the one that started last)
#define QMPC_ORDER_BUCKETS 16
#define QMPC_GRP_INTS 32
#define QMPC_CNT_BIGLIST 3
#define QMPC_CNT_OV 7          // robots that took a slice of the overflow pool in this call
#define QMPC_CNT_OV_PROBES 16  // ... compare-and-swap probes of theirs that found a slice taken (0 unless the pool is nearly full)
#define QMPC_CNT_OV_TIMEOUT 17 // ... robots that gave up after ov_spin probes (-> Schur-form fallback, QMPC_ST_FALLBACK)
#define QMPC_CNT_FB 8
#define QMPC_CNT_FBQ 12
#define QMPC_CNT_GRP(sk, g) (64 + QMPC_GRP_INTS * (2 * (sk) + (g)))
#define QMPC_COUNTERS (64 + QMPC_GRP_INTS * 6)
// work items per class (the pool is min(max_batch, this)): 128-row class 128 KiB each, 192-row class 288 KiB, large
// problems 1.53 MiB -- 0.5 / 0.84 / 1.5 GiB at most, whatever max_batch is
#define QMPC_ITEMS_C2 4096
#define QMPC_ITEMS_C3 3072
#define QMPC_ITEMS_BIG 1024

// decoupled path: a sweep workgroup (or the long-horizon producer) leaves its robot's explicit inverse,
// unconstrained minimiser and stance list in a work item; single-robot engine workgroups consume the items
#define QMPC_ENGINE_OVF_EVENTS 160 // events per engine workgroup in the overflow pool (beyond its registers and LDS)
#define QMPC_BIG_LD 448  // leading dimension of a large problem's work item (n_r <= 432 = 12 x 36; seven 64-row blocks)
#define QMPC_WK_SLOTS_MAX 192  // stance slots an item can describe (three 64-lane groups; trot at horizon 36 has 72)
struct QmpcWorkHdr {
  int rid, n, nst, status0;              // robot, reduced size 3 nst, stance foot-steps, status bits so far
  float fmaxk[QMPC_WK_SLOTS_MAX];        // f_max * gait of stance slot s (float product like SolverMPC.cpp:361)
  unsigned char sidx[QMPC_WK_SLOTS_MAX]; // foot-step 4 step + foot of stance slot s
};

// leading dimension of the debug dump (largest padded size, 3 * 64)
#define QMPC_DBG_LD 192

struct QmpcParams {
  // batched inputs (device pointers; layouts in include/qmpc.h)
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
  int weights_stride, alpha_stride, x_drag_stride;
  // outputs
  float* grf;
  double* soln;
  int32_t* status;
  int32_t* iters;
  // problem constants (problem_setup of convexMPC_interface.h:13-19, rounded
  // to float like the reference stores them, then widened)
  int batch, horizon;
  double dt, mu_inv, inv_fr_norm, f_max;
  double mass, ibody[3], gravity;
  double inv_mass, inv_ibody[3];  // host-side reciprocals (no fp64 divisions in the kernel prologue)
  // discretisation: 0 = the dense path's exact zero-order hold with the gravity state (SolverMPC.cpp:87-125),
  // 1 = SparseCMPC's (SparseCMPC_Math.cpp:6-28: A_d = expm(A dt) = I + A dt, B_d = B dt, g dt added per step)
  int model;
  // batch-constant tables built at qmpc_setup():
  //   coef[p][d]   p<3, d<h     dt, (2d+1)dt^2/2, ((d+1)^3-d^3)dt^3/6
  //   ctab[pq][i][j] pq<9       sum_{k>=max(i,j)} coef_p(k-i) coef_q(k-j), stored (h + 1) x (h + 1) with a ZERO last row and
  //                             column: rows / columns of the kernels' identity padding index them and vanish by themselves
  const double* coef;
  const double* ctab;
  // solver settings
  int max_iter;
  double tol;
  // JCQP alternate (src/JCQP/QpProblem.cpp:178-269; 0 = off -> exact active-set solve): use_jcqp value
  // (1 = full problem, 2 = swing-eliminated) and the caller's settings (ConvexMPCLocomotion.cpp:644-648)
  int admm_mode, admm_max_iter;
  double admm_rho, admm_sigma, admm_alpha, admm_term;
  // largest size class: event pool in global memory, ev_nslot slices of QMPC_EV_SLICE3 doubles, one flag each
  double* evpool;
  int* evflags;
  int ev_nslot;
  int ev_spin;  // bound of the wait for a slice's previous tenant (a workgroup that times out runs the Schur form)
  // classes 1, 4, 2: overflow pool in global memory for the robot whose LDS event pool is full: ov_nslice slices of
  // QMPC_OV_SLICE doubles, one flag each (ov_flags: 0 = free; taken with a compare-and-swap, released by the robot when it
  // is done -- RECYCLED within a call, so the need is bounded by the robots in flight); *ov_count only spreads the probe
  // sequences of concurrent robots; ov_spin bounds the wait of a robot that finds every slice taken
  double* ovpool;
  int* ov_count;
  int* ov_flags;
  int ov_nslice;
  int ov_spin;
  // decoupled path: work items [wk_cap] of this size class -- H^-1 (wk_ld x wk_ld doubles, row-major, symmetric),
  // x_u (wk_ld doubles), header; *wk_count = items produced so far (sweep workgroups take the next index),
  // *wk_qhead = queue head of the engine workgroups; fb_list / fb_count: robots the engine hands back to the
  // monolithic kernel of the class (event capacity exceeded, lost definiteness)
  double* wk_hinv;
  double* wk_xu;
  QmpcWorkHdr* wk_hdr;
  int* wk_count;
  int* wk_qhead;
  int wk_ld, wk_cap;
  int wk_base;  // first work item of this launch's chunk (items wk_base .. wk_base + *wk_count - 1; 0: the chunks of a call reuse the pool one after the other)
  int* wk_zero;  // engine kernel: the counter group (QMPC_GRP_INTS ints) of the NEXT chunk, zeroed by workgroup 0 (nullptr: none)
  // order in which the engine workgroups take the items: QMPC_ORDER_BUCKETS lists of item indices, bucket b of this chunk
  // at wk_order[b * wk_cap + wk_base ...], wk_bucket[b] entries; bucket 0 = most rows violated at x_u
  int* wk_order;
  int* wk_bucket;
  double* wk_ovf;  // engine kernel: overflow event pool, one slice of QMPC_ENGINE_OVF_EVENTS records per workgroup of the grid
  int rid0;     // sweep kernel: first robot (or list entry) of this launch's chunk
  int list_hi;  // ... and one past its last list entry (list-consuming launches)
  int wk_block;  // 1: the engine starts from a block-factorised candidate set (block_start, qmpc_engine.hip)
  int wk_kev;  // events the engine may hold per robot (test hook; the compiled capacity when larger)
  int* fb_list;
  int* fb_count;
  int status_or;  // bits OR-ed into the status of every robot this launch solves (the hand-back launch: QMPC_ST_FALLBACK)
  // warm start (nullptr = cold): [batch][QMPC_WS_STRIDE] working set of the previous cycle as global
  // constraint ids 5 * (4 step + foot) + type, -1 = empty; read slid by ws_shift horizon steps, rewritten
  // with this cycle's final working set
  int32_t* ws;
  int ws_shift;
  // selective warm start (qmpc_set_warm_start_min_iters): only robots that needed at least this many iterations in the
  // handle's previous call (hint_iters) read their previous working set; everybody else starts cold.  <= 0: every robot
  int ws_min_iters;
  // work lists: robots handed from one size class to the next
  const int* list;   // nullptr: robot = blockIdx.x
  int* count;        // entries in `list`
  int* qhead;        // queue head of `list`: entries past the grid size are handed out through it
  int* clear_counts; // first kernel of the chain: the counters and queue heads of the NEXT call's set (8 ints), zeroed here
  int* next_list;    // nullptr: no larger class available
  int* next_count;
  // command mode (qmpc_solve_commands; c_position == nullptr: off): the record is built in
  // stage 0 from the controller command instead of being loaded (include/qmpc.h qmpc_command)
  const float* c_position;
  const float* c_v_world;
  const float* c_omega_world;
  const float* c_orientation;
  const float* c_rpy;
  const float* c_r_body;
  const float* c_p_foot;
  const float* c_vel_des;
  const float* c_yaw_des_true;
  const float* c_rpy_comp;
  const float* c_stand_traj;
  const float* c_rp_des;
  const int32_t* c_gait_type;
  const int32_t* c_gait_offsets;
  const int32_t* c_gait_durations;
  const int32_t* c_gait_iteration;
  float* c_wpd;   // world_position_desired, in/out
  float* c_xci;   // x_comp_integral, in/out
  float c_body_height;
  int c_omni_mode;
  float* f_ff;    // optional output: -rBody * f per leg
  // debug dump (nullptr = off)
  double* dbg_H;
  double* dbg_g;
  double* dbg_aux;     // [batch][8]: cos/sin(yaw), roll, pitch, yaw as the kernel evaluated them (float transcendentals)
  long long* dbg_clk;  // [batch][16] shader-clock stamps per phase
  // order hint (qmpc_set_order_hint; both nullptr = off): the first class of the chain takes robot order[blockIdx.x]
  // instead of blockIdx.x -- the robots that iterated longest in the handle's previous call first, so that a launch of
  // several rounds of workgroups does not end with a hard robot that started last -- and every one-kernel solve
  // leaves its iteration count in hint_iters[robot] for the next call's order
  const int32_t* order;
  int32_t* hint_iters;
  // ... and in a launch of ONE round of workgroups (the order cannot matter: everybody starts at once) a robot the previous
  // call found hard keeps the highest issue priority through its sweep instead of yielding as it advances: of the workgroups
  // that share a CU, the one that will iterate longest finishes its fixed part first.  Hard = at least hint_hard iterations
  // (0 = off) and at least 3/5 of the previous call's maximum, *hint_max_r.  In a one-round launch the robots finish in the order
  // of their iteration counts, so every solve simply stores its count to *hint_max_w and the last store is the maximum (a
  // heuristic: no atomic); the first workgroup of the call clears *hint_max_z for the next call (three slots, rotated by the host)
  // the 96-row class's wave placement (qmpc_kernels.hip: balance_waves): one int per CU, [xcc 3 bits][se, sh, cu 8 bits]
  int* cu_slots;
  int bal_debug;  // test hook (qmpc_set_debug_balance): 1 = no slot word (every workgroup keeps its pairs on SIMDs 0 and 1), 2 = treat the placement as irregular (waves 0..5 stay)
  int hint_hard;
  const int32_t* hint_max_r;
  int32_t* hint_max_w;
  int32_t* hint_max_z;
  // one-round launches without a usable hint: one word per CU ([xcc 3 bits][se, sh, cu 8 bits]) on which the workgroups that share
  // it post (prio_tag << 32 | hardness by the tracking-error proxy << 24 | robot) with an atomic maximum; the one whose entry
  // stands keeps issue priority 3 through its sweep.  prio_tag: a per-handle call number (nullptr = off)
  unsigned long long* prio_cu;
  unsigned prio_tag;
  // size order (qmpc_kernels.hip: size_order_build; nullptr = off): in a launch of several rounds the workgroups from so_first on
  // take robot so_order[blockIdx.x] -- the robots that fit the class largest first (by their contact tables) within so_nseg
  // strided segments (robot so_first + j + so_nseg t: segment j), segment j built by workgroup j of the same launch; entries are
  // (so_tag << 32 | robot), so_tag = a per-handle call number
  unsigned long long* so_order;  // near copy (plain stores: stays in the builder's XCD's L2)
  unsigned long long* so_far;    // far copy (written through: what a reader polls when the near probe missed)
  const int32_t* so_hint;        // nullptr: keys from the contact tables; else the previous call's iteration counts (order hint)
  unsigned so_tag;
  int so_first, so_maxfit, so_nseg;
};

#endif
