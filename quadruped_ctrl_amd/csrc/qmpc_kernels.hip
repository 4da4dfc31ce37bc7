// qmpc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the batched
// convex-MPC hot path of quadruped_ctrl:
//   state -> SRB linearisation -> condensed QP (H_red, g_red) -> exact QP solve
//   -> first-step ground reaction forces
// replacing src/MPC_Ctrl/SolverMPC.cpp:296-639 (solve_mpc) of the reference,
// which runs it for ONE robot on one CPU core through Eigen + qpOASES.
//
// Design (DESIGN.md has the derivations and measurements):
//   * one workgroup per robot, matrix padded to NP = 64 / 96 / 128 / 192 rows (size classes
//     1 / 4 / 2 / 3, NT = 4 NP threads) by the reduced problem size n_r = 3 * (#stance foot-steps).
//   * the n_r x n_r Hessian lives in REGISTERS for the whole solve: thread
//     (row i, column group c) owns H[i][c*CW .. c*CW+CW-1].  LDS only carries
//     vectors (pivot columns, mat-vec operands) and the small working-set
//     state; per robot HBM traffic is the 728 B record in and 48 B out.
//   * assembly exploits A_ct^3 = 0 (SolverMPC.cpp:235-254): with
//     B0 = B, B1 = A B, B2 = A^2 B and batch-constant h x h tables C_pq,
//       qH = 2 sum_pq C_pq (x) (B_p^T W B_q) + 2 alpha I,
//     and every B_p^T W B_q has a closed form in the 3x3 blocks
//     M_b = I_w^-1 [r_b]x, N_b = R_yaw^T M_b -- no 13h x 12h B_qp is ever formed
//     (the reference multiplies it densely, SolverMPC.cpp:395).
//   * inversion by symmetric Gauss-Jordan sweeps, two pivots per barrier; the
//     pivot columns are held one value per lane and broadcast INSIDE the fp64
//     fmac (DP-ALU DPP, row_newbcast) instead of being re-read from LDS by every
//     lane; in class 1 the wave that owns the next pivot pair inverts the 2x2
//     pivot block once for the whole block.  Then a Goldfarb-Idnani dual
//     active-set on the explicit inverse, run by wave 0 out of registers
//     (lane = variable / stance foot-step / working-set slot) with the
//     projected inverse kept as a sum of rank-1 events (no in-place updates; no
//     barrier in the loop in class 1 -- in the larger classes waves 1..3 take a
//     share of the stored events once there are many).  A robot whose on-chip event
//     pool fills up continues on a slice of an overflow pool in global memory; the
//     Schur-form engine (second solve_one instantiation) is the last resort when the
//     working-set slots run out.  Swing feet are eliminated up front exactly like
//     SolverMPC.cpp:441-525.
//   * fp64 throughout the solve (the reference hands fp32-assembled data to a
//     double-precision qpOASES; fp64 assembly removes the fp32 rounding noise
//     instead of adding a second, uncorrelated copy of it).
//   * size classes chain 1 -> 4 -> 2 -> 3: the first is launched over the batch, a robot that does not fit appends
//     itself to a work list that the next class (LISTED instantiation) consumes as a queue.
//   * variants of the same template (own instantiations, the default kernel carries none of them):
//     WARM  -- warm start across MPC cycles from the previous working set (qmpc_set_warm_start);
//     ADMM  -- the reference's JCQP alternate, QpProblem::runFromDense (src/JCQP/QpProblem.cpp:178-269):
//              same assembly and sweep with the KKT system reduced to the x block, ADMM engine;
//     P.model = 1 -- SparseCMPC's discretisation (src/MPC_Ctrl/SparseCMPC_Math.cpp:6-28) through the same
//              closed forms: one coefficient family and the height row of the free response change.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "qmpc_device.h"
#include "qmpc_cmd.h"
#include "qmpc_wave.h"

namespace {
#ifndef QMPC_ENGINE_PRIO
#define QMPC_ENGINE_PRIO 3
#endif
#ifndef QMPC_SWEEP_PRIO
#define QMPC_SWEEP_PRIO 1
#endif
#ifndef QMPC_START_PRIO
#define QMPC_START_PRIO 1
#endif
#ifndef QMPC_PROD_PRIO
#define QMPC_PROD_PRIO 0  // 1: the wave that produces the next pivot pair runs at the highest issue priority while it does (measured, round 6)
#endif

// a[j] += c[lane j of this lane's row of 16] * u for j = 0..15: sixteen DP-ALU DPP
// fmacs (row_newbcast), i.e. the 16 wave-uniform pivot-column values are held one
// per lane and broadcast inside the instruction instead of being re-read from LDS
// by every lane.  One asm block: the leading s_nop covers the VALU-write -> DPP-read
// hazard that the compiler cannot see through inline asm.
__device__ __forceinline__ void fmac16_rowbcast(double (&a)[16], double c, double u) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
        "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
      : "v"(c), "v"(u));
}

// acc += c[lane N of this lane's row of 16] * u (one DP-ALU DPP fmac; N a compile-time lane)
template <int N>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double c, double u) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(u), "n"(N));
}

// the same for four accumulators a[J0 .. J0+3] (lets the producing wave of the
// sweep interleave its pivot arithmetic with the update)
template <int J0>
__device__ __forceinline__ void fmac4_rowbcast(double (&a)[16], double c, double u) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %4, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %4, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
      : "+v"(a[J0]), "+v"(a[J0 + 1]), "+v"(a[J0 + 2]), "+v"(a[J0 + 3])
      : "v"(c), "v"(u), "n"(J0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3));
}

// p[0 .. N-1] += c[lane J0 + k of this lane's row] * u for N = 2, 4 or 8 consecutive columns given by pointer
template <int J0, int N>
__device__ __forceinline__ void fmacn_rowbcast(double* p, double c, double u) {
  static_assert(N == 2 || N == 4 || N == 8, "block of 2, 4 or 8 columns");
  if constexpr (N == 2) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(p[0]), "+v"(p[1])
        : "v"(c), "v"(u), "n"(J0), "n"(J0 + 1));
  } else if constexpr (N == 4) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %4, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
        : "v"(c), "v"(u), "n"(J0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %8, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %8, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %8, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %8, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %8, %9 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %8, %9 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %8, %9 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %8, %9 row_newbcast:%17 row_mask:0xf bank_mask:0xf"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
        : "v"(c), "v"(u), "n"(J0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3), "n"(J0 + 4), "n"(J0 + 5), "n"(J0 + 6), "n"(J0 + 7));
  }
}
// ... for the columns LO .. HI-1 of a group (g points at the group's first column; LO, HI even): aligned blocks of 8 / 4 / 2
template <int LO, int HI>
__device__ __forceinline__ void fmac_range_rowbcast(double* g, double c, double u) {
  if constexpr (LO < HI) {
    constexpr int B = (LO % 8 == 0 && HI - LO >= 8) ? 8 : ((LO % 4 == 0 && HI - LO >= 4) ? 4 : 2);
    fmacn_rowbcast<LO, B>(g + LO, c, u);
    fmac_range_rowbcast<LO + B, HI>(g, c, u);
  }
}

// the same for eight accumulators (the 24-column groups of class 4: 16 + 8)
__device__ __forceinline__ void fmac8_rowbcast(double (&a)[8], double c, double u) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
      : "v"(c), "v"(u));
}

// ---- what a robot will cost, guessed from its record before anything is solved ----
// nst: stance foot-steps (the reduced size is 3 nst: the sweep is that many half-steps long).
// score: a proxy of the active-set iteration count,
//     score = first3 * sum_k Q_k |e_k + T de_k| / sum_k Q_k  +  0.15 * max(0, L1 * sa - 1.5)
//   first term: the tracking error the COASTING state would have at the end of the horizon (orientation and position rows,
//   T = h dt; small-angle roll / pitch: a proxy), relative to the weights, times the stance foot-steps of the first three
//   segments -- the forces that would have to correct it are the ones that run into their bounds;
//   second term: the FIRST support phase (leading flight skipped): L1 segments on one stance set whose centroid is off the
//   centre of mass by c; gravity's moment about it can only be balanced by tangential force, sat = |c| / (height mu) of the
//   friction limit (pacing 0.95, bounding 1.65, trot 0); sa = min(sat, 1) where sat > 0.6.  A long asymmetric first phase is
//   what makes a pacing / bounding robot iterate ten times where a trotting one needs two (configs[2], 30 contact tables).
//   Correlation with the iteration count: configs[1] 0.76, [2] 0.75, [3] 0.69, [4] 0.74 (the size alone: 0.03 ... 0.24).
//   One formula, constants fitted once on those four workloads (profiles/r06_z_proxy_order.md); it ORDERS work, nothing else.
__device__ __forceinline__ unsigned qmpc_stance_set(const uint32_t w) {  // the four contact bytes of a segment -> 4-bit set
  const uint32_t t = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;  // bit 7 of every nonzero byte
  return ((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u);
}
// demand (DEMAND = true only: the one-round staging's gate): the same error rows on a FORCE scale, sum_k sqrt(Q_k / alpha) |e_k + T de_k|
//   in units of the robot's weight, times first3 -- what the regulator would ask of the feet.  The score orders robots among
//   themselves; whether ANY of them will meet a bound it cannot say (SparseCMPC's parameters -- weights 5 - 25 x smaller, mu = 1 --
//   on configs[1]'s states: mean iteration count 0.01 instead of 2.1, the scores are the same).  On configs[1] every robot with six
//   or more iterations has demand >= 11.7; with the sparse model's parameters nobody exceeds 10.7.
struct QmpcKeys {
  int nst;
  float score, demand, pattern;
};
template <bool DEMAND = false>
__device__ __forceinline__ QmpcKeys qmpc_robot_keys(const QmpcParams& P, const int i) {
  const int h = P.horizon;
  // (4 h bytes per robot, the base 4-byte aligned: the host checks)
  const uint32_t* g4 = reinterpret_cast<const uint32_t*>(P.gait + (size_t)i * 4 * h);
  int nst = 0, first3 = 0, run = 0;
  unsigned set0 = 0u;
  bool open = true;
  for (int k = 0; k < h; ++k) {
    const unsigned m = qmpc_stance_set(g4[k]);
    const int c = __builtin_popcount(m);
    nst += c;
    first3 += (k < 3) ? c : 0;
    if (open) {
      if (set0 == 0u) {
        set0 = m;
        run = m ? 1 : 0;
      } else if (m == set0) {
        ++run;
      } else {
        open = false;
      }
    }
  }
  const float* q = P.q + (size_t)i * 4;
  const float* tr = P.traj + (size_t)i * 12 * h;
  const float* wt = P.weights + (size_t)i * P.weights_stride;
  const float T = (float)h * (float)P.dt;
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const float x0[3] = {2.f * (qw * qx + qy * qz), 2.f * (qw * qy - qz * qx), P.yaw[i]};
  float acc = 0.f, wsum = 1e-30f, dem = 0.f;
  const float ialpha = DEMAND ? 1.f / (P.alpha[(size_t)i * P.alpha_stride] + 1e-30f) : 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float er = (x0[k] - tr[k]) + T * (P.w[(size_t)i * 3 + k] - tr[6 + k]);
    const float ep = (P.p[(size_t)i * 3 + k] - tr[3 + k]) + T * (P.v[(size_t)i * 3 + k] - tr[9 + k]);
    acc += wt[k] * __builtin_fabsf(er) + wt[3 + k] * __builtin_fabsf(ep);
    wsum += wt[k] + wt[3 + k];
    if (DEMAND) dem += __builtin_sqrtf(wt[k] * ialpha) * __builtin_fabsf(er) + __builtin_sqrtf(wt[3 + k] * ialpha) * __builtin_fabsf(ep);
  }
  float score = acc / wsum * (float)first3;
  const float demand = DEMAND ? dem * (float)first3 / ((float)P.mass * __builtin_fabsf((float)P.gravity) + 1e-30f) : 0.f;
  float pattern = 0.f;
  if (set0 != 0u) {
    const float* r = P.r + (size_t)i * 12;  // r[axis * 4 + foot]
    float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float on = ((set0 >> f) & 1u) ? 1.f : 0.f;
      cx += on * r[f];
      cy += on * r[4 + f];
      cz += on * r[8 + f];
    }
    const float inv = 1.f / (float)__builtin_popcount(set0);
    cx *= inv;
    cy *= inv;
    cz = __builtin_fabsf(cz * inv);
    const float sat = __builtin_sqrtf(cx * cx + cy * cy) * (float)P.mu_inv / (cz > 1e-3f ? cz : 1e-3f);
    const float sa = sat > 0.6f ? (sat < 1.f ? sat : 1.f) : 0.f;
    const float pat = (float)run * sa - 1.5f;
    pattern = pat > 0.f ? pat : 0.f;
    score += 0.15f * pattern;
  }
  return {nst, score, demand, pattern};
}
// The same keys of ONE robot evaluated by a whole wave (the one-round staging: a single lane's 250 dependent instructions cost
// every workgroup 1.5 - 3 k cycles of stage 0, whichever wave they ran on): lane k takes segment k of the contact table, lanes 0..5
// one error row each, lanes 0..11 one foot coordinate each; ballots and three-step butterflies put them together.  All lanes of the
// wave must call it; the result is uniform.
__device__ __forceinline__ QmpcKeys qmpc_robot_keys_wave(const QmpcParams& P, const int i, const int lane) {
  const int h = P.horizon;
  const uint32_t* g4 = reinterpret_cast<const uint32_t*>(P.gait + (size_t)i * 4 * h);
  // (every load first: the feet's coordinates do not wait for the stance set they are going to be masked with)
  const uint32_t graw = (lane < h) ? g4[lane] : 0u;
  const float rraw = (lane < 12) ? P.r[(size_t)i * 12 + lane] : 0.f;  // r[axis * 4 + foot]
  const unsigned m = qmpc_stance_set(graw);
  int nst = 0, first3 = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const unsigned long long bf = __ballot((m >> f) & 1u);
    nst += __popcll(bf);
    first3 += __popcll(bf & 7ull);
  }
  const unsigned long long nz = __ballot(m != 0u);
  unsigned set0 = 0u;
  int run = 0;
  if (nz != 0ull) {  // (uniform)
    const int k0 = __builtin_ctzll(nz);
    set0 = (unsigned)__builtin_amdgcn_readlane((int)m, k0);
    const unsigned long long ne = ~(__ballot(m == set0 && lane < h) >> k0);
    run = ne ? __builtin_ctzll(ne) : 64;
    run = run < h - k0 ? run : h - k0;
  }
  // error rows: lane 0..2 orientation k, lane 3..5 position k
  const int k = (lane < 3) ? lane : (lane < 6 ? lane - 3 : 0);
  const bool rot = lane < 3, row = lane < 6;
  const float* q = P.q + (size_t)i * 4;
  const float* tr = P.traj + (size_t)i * 12 * h;
  const float T = (float)h * (float)P.dt;
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const float ang = (k == 0) ? 2.f * (qw * qx + qy * qz) : (k == 1 ? 2.f * (qw * qy - qz * qx) : P.yaw[i]);
  const float x0 = rot ? ang : P.p[(size_t)i * 3 + k];
  const float xd = rot ? P.w[(size_t)i * 3 + k] : P.v[(size_t)i * 3 + k];
  const float e = (x0 - tr[(rot ? 0 : 3) + k]) + T * (xd - tr[(rot ? 6 : 9) + k]);
  const float wk = P.weights[(size_t)i * P.weights_stride + (rot ? 0 : 3) + k];
  const float ialpha = 1.f / (P.alpha[(size_t)i * P.alpha_stride] + 1e-30f);
  // (the six rows and the twelve foot coordinates are put together with v_readlane -- uniform scalars, a few cycles each and
  //  independent of one another; butterflies through the LDS crossbar, ~17 dependent ds_bpermute, cost the stage ~1 k cycles)
  const float t_acc = wk * __builtin_fabsf(e), t_dem = __builtin_sqrtf(wk * ialpha) * __builtin_fabsf(e);
  float acc = 0.f, wsum = 1e-30f, dem = 0.f;
#pragma unroll
  for (int l = 0; l < 6; ++l) {
    acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_acc), l));
    wsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wk), l));
    dem += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_dem), l));
  }
  float score = acc / wsum * (float)first3;
  const float demand = dem * (float)first3 / ((float)P.mass * __builtin_fabsf((float)P.gravity) + 1e-30f);
  float pattern = 0.f;
  if (set0 != 0u) {  // (uniform)
    float c3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float rv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rraw), 4 * ax + f));
        c3[ax] += ((set0 >> f) & 1u) ? rv : 0.f;
      }
    const float inv = 1.f / (float)__builtin_popcount(set0);
    const float cx = c3[0] * inv, cy = c3[1] * inv, cz = __builtin_fabsf(c3[2] * inv);
    const float sat = __builtin_sqrtf(cx * cx + cy * cy) * (float)P.mu_inv / (cz > 1e-3f ? cz : 1e-3f);
    const float sa = sat > 0.6f ? (sat < 1.f ? sat : 1.f) : 0.f;
    const float pat = (float)run * sa - 1.5f;
    pattern = pat > 0.f ? pat : 0.f;
    score += 0.15f * pattern;
  }
  return {nst, score, demand, pattern};
}
// score -> `per` logarithmic levels per octave from 2^-6 on, `levels` of them, HARDEST = 0 (a NaN: the easiest)
__device__ __forceinline__ int qmpc_score_level(const float score, const float per, const int levels) {
  int b = (score > 0.f) ? (int)(per * (__log2f(score) + 6.f)) : 0;
  b = b < 0 ? 0 : (b > levels - 1 ? levels - 1 : b);
  return levels - 1 - b;
}

// (one-round staging: the other robots of a staged CU step back one level from the assembly on -- 0: only in the sweep)
#ifndef QMPC_STAGE_EARLY
#define QMPC_STAGE_EARLY 1
#endif

// Size classes.  RB names the class: 1, 2, 3 = 64 / 128 / 192 padded rows (four column
// groups of 16 RB columns, 256 RB threads); 4 = the 96-row class between 1 and 2 (four
// groups of 24 columns, 384 threads, two workgroups per CU) that catches n_r <= 96 --
// trot at horizon 16, most random-contact tables -- which class 2 can only run one
// workgroup per CU.
template <int RB>
struct Cfg {
  // (RB = 6: the 64-row class once more, sized for FIVE workgroups per CU -- 96 VGPRs, 32 KB of LDS with a 16-event pool --
  //  for handles of 4096 robots or more whose every robot fits the class: a large batch is bound by instruction issue, and a
  //  fifth wave per SIMD fills the slots the other four leave; qmpc_capi.cpp decides)
  static constexpr bool C1 = (RB == 1 || RB == 6);
  static constexpr int NP = (RB == 4) ? 96 : (RB == 6 ? 64 : 64 * RB);
  static constexpr int CW = NP / 4;
  static constexpr int NT = 4 * NP;
  // The 96-row class's workgroup is SIX waves, which the dispatcher lays (1,2,1,2)-style over the four SIMDs; two such
  // workgroups per CU never complement each other -- the per-CU totals are always a permutation of (4,3,3,2)
  // (tools/ubench/simd_place.hip) and the SIMD with four waves sets the pace of both workgroups (a barrier per pivot pair).
  // Its solve kernels are therefore LAUNCHED with eight waves (two per SIMD, always), look where they landed
  // (HW_REG_HW_ID), and two of them exit at once: the workgroup keeps both waves on two SIMDs and one on the other two,
  // WHICH two decided per CU (a two-bit slot word per CU in global memory) so that the two residents complement each
  // other: (3,3,3,3).  balance_waves() below
#ifndef QMPC_BALANCE4
#define QMPC_BALANCE4 1
#endif
  static constexpr bool BALANCE = QMPC_BALANCE4 && (RB == 4);
  static constexpr int NT_LAUNCH = BALANCE ? 512 : NT;
  static constexpr int RE = (NP + 63) / 64;  // 64-row blocks of an index-major engine vector
  // Schur-form engine (the fallback): working-set slots.  Every n_r <= 64 problem fits 64; in the largest
  // class the 160 KiB of LDS next to the packed inverse set the bound at run time (48 slots at
  // n_r = 192, all 96 at n_r <= 168)
  static constexpr int KMAX = C1 ? 64 : 96;
  static constexpr int KW = (KMAX + 63) / 64;  // slots per engine lane
  static constexpr int NS = KMAX * (KMAX + 1) / 2;
  static constexpr int NH = NP * (NP + 1) / 2;
  // doubles of active-set storage behind the packed inverse.  Event-form engine:
  // one (z~, g~) record of NP + KS doubles per working-set change; Schur-form
  // engine: packed S_W^-1 (front) and rows H^-1 c_w (back).  Sized so that
  // class 1 keeps 4 workgroups per CU and class 4 two
  static constexpr int POOL = (RB == 1) ? 2688 : (RB == 6 ? 1536 : (RB == 2 ? 11400 : (RB == 4 ? 5120 : 1176)));
  static constexpr int NPOOL = POOL;
  static constexpr int KS = C1 ? 32 : (RB == 3 ? 128 : 64);  // event-form engine: working-set slot capacity (class 3: two per lane)
  static constexpr bool EVENT_ENGINE = true;
  // class 3 has no LDS left for events: its event pool lives in global memory (one slice per workgroup in
  // flight, L2-resident: 2.5 KB per event), the LDS pool only serves the helper waves' partial sums and the
  // Schur-form fallback
  static constexpr bool GLOBAL_EVENTS = (RB == 3);
  // events a slice of a global pool holds.  Class 3's own pool: its 128 working-set slots plus room for drop events
  // before a compaction (QMPC_EV_SLICE3); the overflow pool of the other classes (at most 64 slots): QMPC_OV_SLICE
  static constexpr int KEV_GLOBAL = (RB == 3) ? 160 : 96;
  // waves 1..NHELP take a share of the stored events whenever there are enough of them to be worth two barriers
  // (the larger classes: long active-set runs, and seven or eleven waves with nothing else to do)
  static constexpr int NHELP = C1 ? 0 : 3;
  // event records with the lane's entries stored adjacently (16-byte loads): the classes whose robots hold many
  // events; the record size is the same (NP and KS are multiples of 64 there)
#ifndef QMPC_PAIRED
#define QMPC_PAIRED 1
#endif
  static constexpr bool PAIRED = QMPC_PAIRED && (RB == 2 || RB == 3);
#ifndef QMPC_HELP_MIN_TRIPS
#define QMPC_HELP_MIN_TRIPS 3
#endif
  static constexpr int HELP_MIN_TRIPS = QMPC_HELP_MIN_TRIPS;
  // horizons this class assembles: the reference's gaits use 10 .. 16 segments; its interface takes up to
  // K_MAX_GAIT_SEGMENTS = 36 (convexMPC_interface.h:3).  The long ones (h > 16) are assembled by the 192-row class only:
  // its 768 threads cover the 12 h <= 432 tracking-error entries one per thread, and it alone has the LDS for h x h tables
  static constexpr int HMAX = (RB == 3) ? 36 : 16;
  static constexpr int MIN_WAVES = (RB == 1 || RB == 4) ? 4 : (RB == 6 ? 5 : (RB == 2 ? 2 : 3));  // per SIMD (launch bounds)
  // ... of the producer half of the decoupled path (qmpc_sweep_kernel): without the packed inverse its LDS is the
  // assembly / sweep storage only, so the 128-row class fits two workgroups per CU if it stays within 128 VGPRs
#ifndef QMPC_SWEEP_WAVES2
#define QMPC_SWEEP_WAVES2 4
#endif
  static constexpr int MIN_WAVES_A = (RB == 2) ? QMPC_SWEEP_WAVES2 : MIN_WAVES;
};

template <int RB>
struct Smem {
  using C = Cfg<RB>;
  // ---- live for the whole solve
  QmpcParams par;  // kernel parameters parked in LDS (keeps ~45 uniforms out of SGPRs)
  // stance foot-steps a workgroup can list: 64 (n_r <= 192) -- 144 in the 192-row class's instantiations, whose producer
  // for the LARGE problems (n_r up to 432 = 12 x 36, matrix in global memory: big_tail) shares stages 0 - 1 with them
  static constexpr int SLOTS = (RB == 3) ? 144 : 64;
  double fmaxk[SLOTS];
  unsigned char sidx[SLOTS];
  unsigned char kslot[4 * C::HMAX];  // foot-step k -> stance slot (0xff = swing): inverse of sidx, for the warm start
  int nst, status;
  int mode;  // set by the engine wave: != 0 -> the robot must be re-run with the fallback engine
  int evslot;  // class 3: this workgroup's slice of the global event pool
  int qnext;   // next entry of the work list (classes launched after the first)
  int prio_rank;  // one-round launches without a hint: 0 = the hardest robot of its CU by the tracking-error proxy
  int bal_simd[8], bal_cu, bal_bit;  // balance_waves: where the launched waves landed, this CU's slot word, the bit taken
  // the engine wave's request to the helper waves (event-form engine, classes with NHELP > 0)
  struct Help {
    int cmd;  // 0 = the solve is over, 1 = accumulate your share of the events
    int pj1, pj2, neva, nevd, glob;
    double pa1, pa2;
    unsigned long long gptr;  // the global pool slice when glob != 0
  } hd;
  // ---- phase-local storage
  union U {
    struct AW {
      struct Asm {  // linearisation + assembly (stages 0-2)
        double Mb[4][9];  // M_b = I_world^-1 [r_b]x                 (B0 rows 6..8)
        double Nb[4][9];  // N_b = R_yaw^T M_b                       (B1 rows 0..2)
        double W[12];
        // coefficient tables, (h + 1) x (h + 1) with a ZERO last row and column (built by qmpc_setup): a row of the identity
        // padding reads row h, a column beyond the stance list column h, and their entries of H vanish by themselves
        static constexpr int TAB = (C::HMAX + 1) * (C::HMAX + 1);
        double ct0[TAB], ct4[TAB];  // C_00 (tau), C_11 (sigma)
        double ct1[TAB], ct5[TAB], ct8[TAB];  // C_01, C_12, C_22 (x_drag != 0 only)
        double E00[144], E11[144];
        double e[C::HMAX * 12];
        double s[3][C::HMAX * 12];
      } a;
      struct Swp {  // sweep + unconstrained minimiser (stages 2-4)
        alignas(16) double colbuf[2][2][C::NP + 2];  // [parity][column of the pair][row]
        double ubuf[2][2][C::NP];  // class 1: F columns (pivot-row correction folded in), same buffering
        double g[C::NP];
        double part[4][C::NP];
      } w;
    } aw;
    struct Slv {  // active-set solve (stage 5): the engine wave's working storage
      double Hp[C::NH];    // H^-1, packed lower triangle: (i,j), i >= j, at i(i+1)/2 + j
      // pool.  Event-form engine: records (z~[NP], g~[KS]), add events from the
      // front and drop events from the back.  Schur-form engine: (C_W^T H^-1 C_W)^-1
      // packed like Hp, growing from the front with the slot high-water mark, and
      // rows M[w] = H^-1 c_w (NP doubles each) from the back while they fit
      // (beyond that they are recomputed from Hp)
      double Sinv[C::NPOOL];
      double D[C::NP];  // event-form engine: diag(H^-1)
    } b;
  } u;
};

// phase timestamps (profiling hook; dbg_clk == nullptr in production)
#ifndef QMPC_DBG_ITER
#define QMPC_DBG_ITER 0  // the event-engine iteration the stamps 8 / 9 / 10 are taken in (> 0: stamp 14 = its start; tools/iter_phase.py)
#endif
// fine-grained stamps inside the FIRST active-set iteration
#define QMPC_TICK1(k)                                                \
  do {                                                               \
    if (dbg_clk && tid == 0 && iters == 0) dbg_clk[(k)] = clock64(); \
  } while (0)
#define QMPC_TICK(k)                                   \
  do {                                                 \
    if (dbg_clk && tid == 0) dbg_clk[(k)] = clock64(); \
  } while (0)

// profiling build (-DQMPC_SWEEP_STAMP=<k0>): shader-clock stamps INSIDE the sweep step that starts at pivot k0, taken by thread 0
// and stored in row rid + batch of the clock buffer (which the tool allocates twice as long), slots 0..7: tools/sweep_step_phase.py
#ifdef QMPC_SWEEP_STAMP
#define QMPC_STEP_TICK(k, dep)                                                                            \
  do {                                                                                                    \
    if (PK.dbg_clk && tid == 0 && k0 == QMPC_SWEEP_STAMP) {                         \
      double dep__ = (dep);                                                                               \
      asm volatile("" ::"v"(dep__));                                                                      \
      PK.dbg_clk[(size_t)(rid + PK.batch) * 16 + (k)] = clock64();                                    \
    }                                                                                                     \
  } while (0)
#else
#define QMPC_STEP_TICK(k, dep) do { } while (0)
#endif

// census builds (-DQMPC_STOP_AFTER=<k>, tools/census.sh; never in production): the workgroup leaves solve_one right after stage k
// (-1: at entry), after a dump that keeps everything the stage produced alive -- every byte of the workgroup's LDS and the given
// registers are summed into the robot's grf row -- so that the per-stage instruction counts are differences of PMC counters
// between consecutive builds (the dump is the same in all of them and cancels)
#ifdef QMPC_STOP_AFTER
template <int RB, int NV>
__device__ __forceinline__ void qmpc_census_dump(Smem<RB>& S, const QmpcParams& PK, int rid, int tid, const double (&regs)[NV]) {
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) acc += regs[k];
  const volatile double* lds = reinterpret_cast<const volatile double*>(&S);
  for (int k = tid; k < (int)(sizeof(Smem<RB>) / 8); k += Cfg<RB>::NT) acc += lds[k];
  PK.grf[(size_t)rid * 12 + (tid % 12)] = (float)acc;
  if (tid == 0) {
    PK.status[rid] = 0;
    if (PK.iters) PK.iters[rid] = 0;
  }
}
#define QMPC_STOP(k, regs)                                  \
  do {                                                      \
    if constexpr (QMPC_STOP_AFTER == (k)) {                 \
      __syncthreads();                                      \
      qmpc_census_dump<RB>(S, PK, rid, tid, regs);          \
      __syncthreads();                                      \
      return false;                                         \
    }                                                       \
  } while (0)
#else
#define QMPC_STOP(k, regs) do { } while (0)
#endif

// CMD selects where the input record comes from: false = loaded (qmpc_solve), true =
// generated in stage 0 from the controller command (qmpc_solve_commands).
// V5 selects the active-set engine: true = event form (projected inverse as a
// sum of rank-1 events; fastest, capacity-limited by the LDS pool), false = the
// Schur form that never overflows.  Both are run by wave 0 alone.  Returns true
// when the robot must be re-run with the other engine.
// PHA = true: the PRODUCER half of the decoupled path (qmpc_sweep_kernel): stages 0-3 and the unconstrained
// minimiser as below, then the inverse goes to the robot's work item in global memory (L2 / Infinity-Cache
// resident) instead of LDS and the workgroup moves on; the active set is run by qmpc_engine_kernel
// (qmpc_engine.hip) -- a workgroup of the large classes no longer pins a whole CU while one of its waves iterates.
template <int RB, bool V5, bool CMD, bool ADMM = false, bool WARM = false, bool PHA = false, bool BIG = false, bool PRIO = false>
__device__ __forceinline__ bool solve_one(const int rid, const int tid, Smem<RB>& S, const QmpcParams& PK) {
  using C = Cfg<RB>;
  constexpr int NP = C::NP, CW = C::CW, NT = C::NT, KMAX = C::KMAX, KW = C::KW, RE = C::RE;
  const QmpcParams& P = S.par;  // parked copy: everything after stage 0
  const int lane = tid & (WAVE - 1);
  const int i = tid % NP;  // matrix row owned by this thread
  const int c = tid / NP;  // column group (0..3): columns c*CW .. c*CW+CW-1
  const int h = PK.horizon;
  const int nfs = 4 * h;   // foot-steps in the horizon (<= 64; <= 144 in the 192-row class, horizons up to 36)
  constexpr int HMAX = C::HMAX;
  constexpr int NFG = (4 * HMAX + 63) / 64;  // 64-foot-step groups of the contact table
  long long* dbg_clk = PK.dbg_clk ? PK.dbg_clk + (size_t)rid * 16 : nullptr;
  QMPC_TICK(0);
  {
    const double none[1] = {0.0};
    (void)none;
    QMPC_STOP(-1, none);
  }
#if QMPC_SWEEP_PRIO && QMPC_START_PRIO
  __builtin_amdgcn_s_setprio(3);  // a workgroup that is just starting is behind everybody else on its CU
#endif
  // order hint, single-round launches: a robot the previous call found hard (wave-uniform: two scalar loads)
  bool hard = false;
  int hfloor = 0;  // the priority the robot does not fall below while it sweeps
  if (PK.hint_hard > 0) {
    // hard = at least hint_hard iterations, at least 3/5 of the previous call's maximum, AND among the top two of the sixteen
    // robots of its aligned group (one 64-byte scalar load): where every robot iterates about equally long nobody is
    // singled out and the staging below stays what it is -- everybody at the top priority is no staging at all (-13 %)
    const int32_t* grp = (const int32_t*)__builtin_assume_aligned(PK.hint_iters + (rid & ~15), 64);
    int mine = 0, ge = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) mine = ((rid & 15) == k) ? grp[k] : mine;
#pragma unroll
    for (int k = 0; k < 16; ++k) ge += (grp[k] >= mine) ? 1 : 0;
    mine = __builtin_amdgcn_readfirstlane(mine);
    ge = __builtin_amdgcn_readfirstlane(ge);
    const int top = __builtin_amdgcn_readfirstlane(*PK.hint_max_r);
    hard = mine >= PK.hint_hard && 5 * mine >= 3 * top && ge <= 2;
    hfloor = hard ? 3 : ((mine >= PK.hint_hard && 5 * mine >= 2 * top && ge <= 4) ? 2 : 0);
  }

  unsigned long long prio_mine = 0ull;
  unsigned long long* prio_word = nullptr;
  // ------------------------------------------------------------ stage 0
  // Every global load of the robot's record is issued up front (one memory
  // latency for the whole stage); then: contact table -> compact stance list
  // (SolverMPC.cpp:441-469 finds the same set by scanning for ub ~ 0 rows) and
  // everything that needs no LDS input.
  auto& Aa = S.u.aw.a;
  auto& Sw = S.u.aw.w;
  // one tracking-error entry per thread (12h <= NT), taken by the UPPER half of the block
  // first: waves 0-1 already carry the stance list, M_b / N_b and the tables, so the
  // transcendental-heavy error rows run beside them instead of after them
  const int eidx0 = (tid + NT / 2) % NT;
  const bool e_thr = eidx0 < 12 * h;
  const int ek = eidx0 / 12, erow = eidx0 - 12 * ek;
  const int mt = tid % 36, mb = mt / 9, ml = (mt % 9) / 3, max_ = mt % 3;  // (foot, row, axis) of M_b / N_b
  const int hs = h + 1, hh = hs * hs;  // the tables' row stride and size (zero last row / column)
  // ---- loads.  Command mode (qmpc_solve_commands): the record is generated here from the
  // controller command with the arithmetic of qmpc_cmd.h instead of being loaded.
  constexpr bool cmdm = CMD;  // compile-time: the record path carries no trace of the command mode
  // the parameter block is parked in LDS one dword per thread (read back as `P` after
  // barrier 1); its load goes out with all the others
  static_assert(sizeof(QmpcParams) % 4 == 0 && sizeof(QmpcParams) / 4 <= 256, "parameter block copy");
  uint32_t g_par = 0;
  if (tid < (int)(sizeof(QmpcParams) / 4)) g_par = reinterpret_cast<const uint32_t*>(&PK)[tid];
  float g_yaw = cmdm ? PK.c_rpy[(size_t)rid * 3 + 2] : PK.yaw[rid];
  float g_xdrag = cmdm ? PK.c_xci[rid] : PK.x_drag[(size_t)rid * PK.x_drag_stride];
  float c_p0 = 0.f, c_p1 = 0.f, c_p2 = 0.f;
  QmpcTrajGen tg;
  if (cmdm) {
    const float* pos = PK.c_position + (size_t)rid * 3;
    c_p0 = pos[0];
    c_p1 = pos[1];
    c_p2 = pos[2];
    const bool stand = PK.c_gait_type && PK.c_gait_type[rid] == 4;
    float vw0, vw1;
    qmpc_cmd_vdes_world(PK.c_r_body + (size_t)rid * 9, PK.c_vel_des[(size_t)rid * 3 + 0], PK.c_vel_des[(size_t)rid * 3 + 1],
                        PK.c_omni_mode, vw0, vw1);
    qmpc_cmd_traj_gen(tg, stand, stand ? PK.c_stand_traj + (size_t)rid * 6 : nullptr,
                      PK.c_rp_des ? PK.c_rp_des + (size_t)rid * 2 : nullptr, PK.c_rpy_comp + (size_t)rid * 2,
                      PK.c_yaw_des_true[rid], PK.c_wpd[(size_t)rid * 2 + 0], PK.c_wpd[(size_t)rid * 2 + 1], c_p0, c_p1,
                      PK.c_body_height, PK.c_vel_des[(size_t)rid * 3 + 2], vw0, vw1, (float)PK.dt);
  }
  // contact table: wave 0 takes foot-step lane + 64 g of every 64-foot-step group g (one group up to horizon 16)
  auto gait_at = [&](int fs) __attribute__((always_inline)) {
    if (cmdm) {
      const int leg = fs & 3;
      return (unsigned char)qmpc_cmd_gait_bit(fs >> 2, PK.c_gait_iteration[rid], PK.c_gait_offsets[(size_t)rid * 4 + leg],
                                              PK.c_gait_durations[(size_t)rid * 4 + leg], h);
    }
    return (unsigned char)PK.gait[(size_t)rid * nfs + fs];
  };
  unsigned char g_gait = 0;
  unsigned char g_gaitx[NFG > 1 ? NFG - 1 : 1] = {};
  if (tid < nfs) g_gait = gait_at(tid);
  if constexpr (NFG > 1) {
    if (tid < WAVE) {
#pragma unroll
      for (int g = 1; g < NFG; ++g)
        if (tid + 64 * g < nfs) g_gaitx[g - 1] = gait_at(tid + 64 * g);
    }
  }
  float g_r0 = 0.f, g_r1 = 0.f, g_r2 = 0.f;
  if (tid < 72) {  // r_feet(axis, foot) = r[axis*4 + foot], RobotState.cpp:25-27
    if (cmdm) {
      const float* pf = PK.c_p_foot + (size_t)rid * 12 + 3 * mb;
      g_r0 = qmpc_cmd_foot_offset(pf[0], c_p0);
      g_r1 = qmpc_cmd_foot_offset(pf[1], c_p1);
      g_r2 = qmpc_cmd_foot_offset(pf[2], c_p2);
    } else {
      const float* r = PK.r + (size_t)rid * 12;
      g_r0 = r[0 * 4 + mb];
      g_r1 = r[1 * 4 + mb];
      g_r2 = r[2 * 4 + mb];
    }
  }
  float g_q[4] = {1.f, 0.f, 0.f, 0.f}, g_w[3] = {0.f, 0.f, 0.f}, g_v[3] = {0.f, 0.f, 0.f};
  float g_p = 0.f, g_traj = 0.f, g_wt = 0.f;
  if (e_thr) {
    const float* q = (cmdm ? PK.c_orientation : PK.q) + (size_t)rid * 4;
    const float* om = (cmdm ? PK.c_omega_world : PK.w) + (size_t)rid * 3;
    const float* v = (cmdm ? PK.c_v_world : PK.v) + (size_t)rid * 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) g_q[k] = q[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g_w[k] = om[k];
      g_v[k] = v[k];
    }
    if (cmdm) {
      if (erow >= 3 && erow < 6) g_p = (erow == 3) ? c_p0 : (erow == 4 ? c_p1 : c_p2);
      g_traj = qmpc_cmd_traj_value(tg, ek, erow);
      g_wt = qmpc_cmd_weight(erow);
    } else {
      if (erow >= 3 && erow < 6) g_p = PK.p[(size_t)rid * 3 + (erow - 3)];
      g_traj = PK.traj[(size_t)rid * 12 * h + eidx0];
      g_wt = PK.weights[(size_t)rid * PK.weights_stride + erow];
    }
  }
  float g_w12 = 0.f;
  if (tid >= 96 && tid < 96 + 12)
    g_w12 = cmdm ? qmpc_cmd_weight(tid - 96) : PK.weights[(size_t)rid * PK.weights_stride + (tid - 96)];
  // (loaded here with everything else: a global load issued after barrier 1 would be waited
  //  for by barrier 2)
  float g_alpha = cmdm ? 4e-5f : PK.alpha[(size_t)rid * PK.alpha_stride];  // ConvexMPCLocomotion.cpp:604
  double g_ct0 = 0.0, g_ct4 = 0.0, g_ct1 = 0.0, g_ct5 = 0.0, g_ct8 = 0.0;
  // ((h + 1)^2 <= 256 == NT in the 64-row class up to h = 15; h = 16 there, and the 192-row class -- 768 threads, up to 37 x 37
  //  entries -- take a second entry)
  constexpr bool TAB2 = (HMAX + 1) * (HMAX + 1) > NT;
  double g2_ct0 = 0.0, g2_ct4 = 0.0, g2_ct1 = 0.0, g2_ct5 = 0.0, g2_ct8 = 0.0;
  if constexpr (TAB2) {
    if (tid + NT < hh) {
      g2_ct0 = PK.ctab[tid + NT];
      g2_ct4 = PK.ctab[4 * hh + tid + NT];
      g2_ct1 = PK.ctab[1 * hh + tid + NT];
      g2_ct5 = PK.ctab[5 * hh + tid + NT];
      g2_ct8 = PK.ctab[8 * hh + tid + NT];
    }
  }
  if (tid < hh) {
    // the x_drag tables are fetched unconditionally: making them wait for the x_drag
    // value would put a second memory round trip in front of them
    g_ct0 = PK.ctab[tid];
    g_ct4 = PK.ctab[4 * hh + tid];
    g_ct1 = PK.ctab[1 * hh + tid];
    g_ct5 = PK.ctab[5 * hh + tid];
    g_ct8 = PK.ctab[8 * hh + tid];
  }
  // ONE-ROUND launch without a usable hint (prio_cu != nullptr): the workgroups that share a CU compete for its issue
  // slots, and the CU is done when the LAST of them is -- the one that iterates longest.  Which one that will be is guessed from
  // the robot's keys (qmpc_robot_keys: correlation of the score with the iteration count 0.69 ... 0.76 -- configs[1]'s 13-iteration
  // robot is third of 1024): wave 1 (which only copies tables in stage 0: the stance list and M_b / N_b are wave 0's, the
  // transcendental-heavy error rows waves 2 - 3's) evaluates them HERE, behind the issue of its own loads of this stage (in front of them it put a second memory round trip
  // into the stage: +2.8 k cycles for everybody), all its lanes together
  // (qmpc_robot_keys_wave), and its last lane posts (call number, score level, robot) with ONE atomic maximum on its CU's word
  // (four arrivals per word, served by the XCD's L2; a newer call's number beats whatever an older call left: nothing to reset) ...
  if constexpr (PRIO && !CMD && !ADMM && !BIG && !PHA) {
    if (PK.prio_cu && (tid >> 6) == 1) {  // (wave 1, all lanes; its last lane posts)
      const QmpcKeys kk = qmpc_robot_keys_wave(PK, rid, lane);
      // (posted only by a robot that is likely to meet its bounds at all: where nobody does, lifting one robot of four over its
      //  neighbours only delays the other three -- SparseCMPC's parameters on configs[1]'s states: -9 %)
      if (tid == 127) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        prio_word = PK.prio_cu + (((xcc & 7u) << 8) | ((hwid >> 8) & 0xffu));
        if (kk.demand >= 12.f || kk.pattern > 0.5f) {
          const int pb = qmpc_score_level(kk.score, 6.f, 64);
          prio_mine = ((unsigned long long)PK.prio_tag << 32) | ((unsigned long long)(63 - pb) << 24) | (unsigned)(rid & 0xffffff);
          __hip_atomic_fetch_max(prio_word, prio_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  // Every load above is in flight now.  The compiler otherwise sinks the first use
  // of each value (a conversion) into the conditional block of its load and waits for
  // memory there -- one full round trip per block, eight in a row.  Touching all the
  // values here, after the last load has been issued, leaves one wait for the lot.
  {
    int gg = g_gait;
    asm volatile("" : "+v"(g_par), "+v"(g_yaw), "+v"(g_xdrag), "+v"(g_alpha), "+v"(gg), "+v"(g_r0), "+v"(g_r1),
                 "+v"(g_r2), "+v"(g_q[0]), "+v"(g_q[1]), "+v"(g_q[2]), "+v"(g_q[3]), "+v"(g_w[0]), "+v"(g_w[1]),
                 "+v"(g_w[2]), "+v"(g_v[0]), "+v"(g_v[1]), "+v"(g_v[2]), "+v"(g_p), "+v"(g_traj), "+v"(g_wt),
                 "+v"(g_w12), "+v"(g_ct0), "+v"(g_ct4), "+v"(g_ct1), "+v"(g_ct5), "+v"(g_ct8));
    g_gait = (unsigned char)gg;
    if constexpr (TAB2) asm volatile("" : "+v"(g2_ct0), "+v"(g2_ct4), "+v"(g2_ct1), "+v"(g2_ct5), "+v"(g2_ct8));
  }
  if (tid < (int)(sizeof(QmpcParams) / 4)) reinterpret_cast<uint32_t*>(&S.par)[tid] = g_par;
  const double x_drag = (double)g_xdrag;
  const bool drag = (x_drag != 0.0);

  if (dbg_clk && tid == 0) {
    float sink = g_yaw + g_traj + (float)g_gait + (float)g_ct0;
    asm volatile("" ::"v"(sink));
    dbg_clk[11] = clock64();
  }
  // ---- stance list (wave 0; group after group when the horizon has more than 64 foot-steps)
  if (tid < WAVE) {
    int base = 0;
#pragma unroll
    for (int g = 0; g < NFG; ++g) {
      const int fs = tid + 64 * g;
      const float fm = (float)(g == 0 ? g_gait : g_gaitx[g > 0 ? g - 1 : 0]) * (float)PK.f_max;  // :361
      // (use_jcqp == 1 hands JCQP the FULL problem, swing foot-steps included with u = 0: SolverMPC.cpp:400-407)
      const bool st = fs < nfs && (((ADMM || BIG) && PK.admm_mode == 1) ? true : !(fm < 0.01f && fm > -.01f));  // :64-67
      const unsigned long long mask = __ballot(st);
      const int pos = base + __popcll(mask & ((1ull << tid) - 1ull));
      if (st && pos < Smem<RB>::SLOTS) {  // (beyond 64 stance foot-steps, n_r > 192: the large-problem producer's robots)
        S.sidx[pos] = (unsigned char)fs;
        S.fmaxk[pos] = (double)fm;
      }
      if (fs < 4 * HMAX) S.kslot[fs] = (st && pos < Smem<RB>::SLOTS) ? (unsigned char)pos : (unsigned char)0xff;
      base += __popcll(mask);
    }
    if (tid == 0) {
      S.nst = base;
      S.status = 0;
    }
  }
  if (PK.soln)  // q_soln is zero on swing feet (SolverMPC.cpp:545-551)
    for (int k = tid; k < 12 * h; k += NT) PK.soln[(size_t)rid * 12 * h + k] = 0.0;

  const double inv_m = PK.inv_mass;
  {
    // yaw rotation (RobotState.cpp:30-35); float transcendentals like the reference
    float syf, cyf;
    sincosf(g_yaw, &syf, &cyf);
    const double cy = cyf, sy = syf;
    if (PK.dbg_aux && tid == 0) {  // test hook: the float transcendentals as evaluated here
      PK.dbg_aux[(size_t)rid * 8 + 0] = cy;
      PK.dbg_aux[(size_t)rid * 8 + 1] = sy;
    }
    if (tid < 72) {
      // M_b = I_w^-1 [r_b]x with I_w^-1 = R diag(1/I) R^T (closed form of
      // I_world.inverse(), SolverMPC.cpp:319,:247), N_b = R^T M_b.
      const int b = mb, l = ml, ax = max_;
      const double ix = PK.inv_ibody[0], iy = PK.inv_ibody[1], iz = PK.inv_ibody[2];
      // cy, sy are float evaluations, so cy^2 + sy^2 = 1 + e with |e| ~ 1e-7 and R is not exactly
      // orthogonal: the exact inverse of R I_body R^T is the closed form divided by (1 + e)^2
      // (1 - 2e + 3e^2 to 1e-20), which is what I_world.inverse() returns up to its own rounding
      const double en = __builtin_fma(cy, cy, sy * sy) - 1.0;
      const double isc = 1.0 - 2.0 * en + 3.0 * en * en;
      const double I00 = (cy * cy * ix + sy * sy * iy) * isc, I01 = (cy * sy * (ix - iy)) * isc,
                   I11 = (sy * sy * ix + cy * cy * iy) * isc;
      const double rx = g_r0, ry = g_r1, rz = g_r2;
      // column ax of [r]x  (cross_mat, SolverMPC.cpp:226-233)
      const double cm0 = (ax == 0) ? 0.0 : (ax == 1 ? -rz : ry);
      const double cm1 = (ax == 0) ? rz : (ax == 1 ? 0.0 : -rx);
      const double cm2 = (ax == 0) ? -ry : (ax == 1 ? rx : 0.0);
      const double m0 = I00 * cm0 + I01 * cm1;  // M_b[0][ax]
      const double m1 = I01 * cm0 + I11 * cm1;  // M_b[1][ax]
      const double m2 = iz * cm2;               // M_b[2][ax]
      if (tid < 36) {
        Aa.Mb[b][3 * l + ax] = (l == 0) ? m0 : (l == 1 ? m1 : m2);
      } else {
        // R^T = [[c, s, 0], [-s, c, 0], [0, 0, 1]]   (A(0:3,6:9), SolverMPC.cpp:244)
        Aa.Nb[b][3 * l + ax] = (l == 0) ? (cy * m0 + sy * m1) : (l == 1 ? (-sy * m0 + cy * m1) : m2);
      }
    } else if (tid >= 96 && tid < 96 + 12) {
      Aa.W[tid - 96] = (double)g_w12;
    }
    // weighted tracking error of the free response at step k (k < h):
    //   e_k = W .* (x0 + A x0 t + A^2 x0 t^2/2 - xd_k),  t = (k+1) dt
    // ( = S (A_qp x0 - X_d), SolverMPC.cpp:399 ), closed form per state row.
    if (e_thr) {
      const int row = erow;
      const double t = (double)(ek + 1) * PK.dt;
      double val;
      if (row < 3) {
        // (float products and sums evaluated one by one, like the reference's x86 build: whether the compiler fuses them
        //  otherwise depends on the shape of the surrounding code, and two instantiations of this template can disagree
        //  in the last float bit of the angles -- 2e-8 in g)
#pragma clang fp contract(off)
        // x0(0..2) = roll, pitch, yaw from the quaternion (SolverMPC.cpp:257-267, :318)
        const float w = g_q[0], x = g_q[1], y = g_q[2], z = g_q[3];
        // roll and yaw share ONE atan2f evaluation (the rows diverge inside a wave, so two
        // calls would simply run one after the other); pitch replaces its lane's value
        const float a_num = (row == 0) ? 2.f * (y * z + w * x) : 2.f * (x * y + w * z);
        const float a_den = (row == 0) ? (w * w - x * x - y * y + z * z) : (w * w + x * x - y * y - z * z);
        float ang = atan2f(a_num, a_den);
        if (row == 1) {
          double asd = -2. * (double)(x * z - w * y);
          if (!(asd < .99999)) asd = .99999;
          ang = asinf((float)asd);
        }
        const double o0 = g_w[0], o1 = g_w[1], o2 = g_w[2];
        const double rate = (row == 0) ? (cy * o0 + sy * o1) : (row == 1 ? (-sy * o0 + cy * o1) : o2);
        if (PK.dbg_aux && ek == 0) PK.dbg_aux[(size_t)rid * 8 + 2 + row] = (double)ang;  // roll, pitch, yaw as evaluated
        val = (double)ang + rate * t;  // Theta' = R_yaw^T omega
      } else if (row < 6) {
        val = (double)g_p + (double)(row == 3 ? g_v[0] : (row == 4 ? g_v[1] : g_v[2])) * t;
        if (row == 5) {
          // ZOH model: 1/2 g t^2 (A(11,12), A(11,9)); SparseCMPC's model adds g dt to the velocity once per step
          // and integrates positions with explicit Euler: g dt^2 n (n - 1) / 2 = 1/2 g (t^2 - t dt)
          val += (PK.model == 1) ? 0.5 * PK.gravity * (t * t - t * PK.dt)
                                 : 0.5 * (PK.gravity + x_drag * (double)g_v[0]) * t * t;
        }
      } else if (row < 9) {
        val = (double)(row == 6 ? g_w[0] : (row == 7 ? g_w[1] : g_w[2]));
      } else {
        val = (double)(row == 9 ? g_v[0] : (row == 10 ? g_v[1] : g_v[2]));
        if (row == 11) val += (PK.gravity + x_drag * (double)g_v[0]) * t;
      }
      Aa.e[eidx0] = (double)g_wt * (val - (double)g_traj);
    }
    if (tid < hh) {
      Aa.ct0[tid] = g_ct0;
      Aa.ct4[tid] = g_ct4;
      if (drag) {
        Aa.ct1[tid] = g_ct1;
        Aa.ct5[tid] = g_ct5;
        Aa.ct8[tid] = g_ct8;
      }
    }
    if constexpr (TAB2) {
      if (tid + NT < hh) {
        Aa.ct0[tid + NT] = g2_ct0;
        Aa.ct4[tid + NT] = g2_ct4;
        if (drag) {
          Aa.ct1[tid + NT] = g2_ct1;
          Aa.ct5[tid + NT] = g2_ct5;
          Aa.ct8[tid + NT] = g2_ct8;
        }
      }
    }
  }
  if (dbg_clk && tid == 0) dbg_clk[12] = clock64();
  __syncthreads();  // ---- barrier 1
  QMPC_TICK(1);
  // ... and a stage later reads the word back: the robot whose entry stands is the CU's hardest and keeps the top priority through
  // its sweep (S.prio_rank = 0), the others yield as they advance, as ever.  The barriers of stage 1 publish it
  if constexpr (PRIO && !CMD && !ADMM && !BIG && !PHA) {
    if (PK.prio_cu && tid == 127) {  // 0: this robot's entry stands; 1: another robot's of this call; 2: nobody on this CU posted
      const unsigned long long pv = __hip_atomic_load(prio_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S.prio_rank = (prio_mine != 0ull && pv == prio_mine) ? 0 : ((unsigned)(pv >> 32) == PK.prio_tag ? 1 : 2);
    }
  }
  {
    const double keep0[2] = {(double)g_alpha, x_drag};
    (void)keep0;
    QMPC_STOP(0, keep0);
  }
#if QMPC_SWEEP_PRIO && QMPC_START_PRIO
  // one-round launch with the order hint: everybody started at the highest priority (a tie); from here to the sweep the
  // robots the previous call did not find hard step back one level (the hint's two scalar loads have landed with the record)
  if (PK.hint_hard > 0 && !hard) __builtin_amdgcn_s_setprio(2);
#endif
  const int nst = S.nst;
  const int n = 3 * nst;
  // command mode: what the thread that finalises a robot does besides the outputs --
  // the controller state owned by the packer (updated once, by the run that produces
  // the result, so a fallback re-run still reads the old values) ...
  auto cmd_finish_state = [&]() __attribute__((always_inline)) {
    // recomputed from memory here rather than carried in registers through the whole solve
    const float* pos = P.c_position + (size_t)rid * 3;
    if (!(P.c_gait_type && P.c_gait_type[rid] == 4)) {
      P.c_wpd[(size_t)rid * 2 + 0] = qmpc_cmd_clamp(P.c_wpd[(size_t)rid * 2 + 0], pos[0]);  // ConvexMPCLocomotion.cpp:534-545
      P.c_wpd[(size_t)rid * 2 + 1] = qmpc_cmd_clamp(P.c_wpd[(size_t)rid * 2 + 1], pos[1]);
    }
    P.c_xci[rid] = qmpc_cmd_xci_next(P.c_xci[rid], pos[2], P.c_body_height, (float)P.dt, P.c_v_world[(size_t)rid * 3 + 0]);
  };
  // ... and, optionally, the body-frame forces f_ff[leg] = -rBody * f (:672-680) from the
  // twelve floats staged in `fb` (lanes 0..11 of one wave)
  auto cmd_finish_forces = [&](const float* fb, int l12) __attribute__((always_inline)) {
    const int leg = l12 / 3, ii = l12 - 3 * leg;
    P.f_ff[(size_t)rid * 12 + l12] =
        qmpc_cmd_f2b(P.c_r_body + (size_t)rid * 9 + 3 * ii, fb[3 * leg], fb[3 * leg + 1], fb[3 * leg + 2]);
  };
  static_assert(!BIG || (RB == 3 && PHA && !ADMM && !WARM), "large-problem producer: an instantiation of the 192-row class");
  if (nst == 0 || n > (BIG ? 3 * Smem<RB>::SLOTS : NP)) {
    // all-swing: q_soln is all zeros (SolverMPC.cpp:545-551).  Too large for
    // this instantiation: hand the robot to the next size class.
    if (nst == 0) {
      if (tid < 12) P.grf[(size_t)rid * 12 + tid] = 0.f;
      if (cmdm && P.f_ff && tid < 12) P.f_ff[(size_t)rid * 12 + tid] = 0.f;
      if (tid == 0) {
        P.status[rid] = 0;
        if (P.iters) P.iters[rid] = 0;
        if (cmdm) cmd_finish_state();
      }
      if (WARM && P.ws && tid < QMPC_WS_STRIDE) P.ws[(size_t)rid * QMPC_WS_STRIDE + tid] = -1;
    } else if (P.next_list) {
      if (tid == 0) {
        const int slot = atomicAdd(P.next_count, 1);
        // (a list holds one entry per robot of the call at most; a slot beyond that means the host's counters are
        //  corrupt -- report the robot instead of writing past the list)
        if (slot < P.batch) {
          P.next_list[slot] = rid;
        } else {
          P.status[rid] = QMPC_DEV_ST_WS_FULL;
          if (P.iters) P.iters[rid] = 0;
          for (int k = 0; k < 12; ++k) P.grf[(size_t)rid * 12 + k] = 0.f;
        }
      }
    } else {
      // larger than the caller's size hint allows and no class left to take it: reported, with
      // every output of the robot defined (zero forces, never a previous call's values) and
      // the controller state advanced like everybody else's
      if (tid < 12) P.grf[(size_t)rid * 12 + tid] = 0.f;
      if (cmdm && P.f_ff && tid < 12) P.f_ff[(size_t)rid * 12 + tid] = 0.f;
      if (tid == 0) {
        P.status[rid] = QMPC_DEV_ST_WS_FULL;
        if (P.iters) P.iters[rid] = 0;
        if (cmdm) cmd_finish_state();
      }
    }
    __syncthreads();
    return false;
  }
  const double alpha = (double)g_alpha;
  if constexpr (PHA) {
    if (tid == 0) S.evslot = P.wk_base + atomicAdd(P.wk_count, 1);  // this robot's work item (read after barrier 2)
  }
  // ------------------------------------------------------------ stage 1
  // E_00 = B0^T W B0, E_11 = B1^T W B1 in closed form, and the weighted sums
  // s_p[st] = sum_{k>=st} coef_p(k-st) e_k.
  if (tid < 144) {
    const int u = tid / 12, v = tid - 12 * u;
    const int bu = u / 3, au = u - 3 * bu, bv = v / 3, av = v - 3 * bv;
    double e00 = 0.0, e11 = 0.0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {  // W * (x * y): E[u][v] == E[v][u] bitwise
      e00 += Aa.W[6 + l] * (Aa.Mb[bu][3 * l + au] * Aa.Mb[bv][3 * l + av]);
      e11 += Aa.W[l] * (Aa.Nb[bu][3 * l + au] * Aa.Nb[bv][3 * l + av]);
    }
    if (au == av) {
      e00 += Aa.W[9 + au] * (inv_m * inv_m);
      e11 += Aa.W[3 + au] * (inv_m * inv_m);
      if (drag && au == 0) e11 += Aa.W[11] * ((x_drag * inv_m) * (x_drag * inv_m));  // B1 row 11
    }
    Aa.E00[tid] = e00;
    Aa.E11[tid] = e11;
  }
  // s_p[st] = sum_{k>=st} coef_p(k-st) e_k for the three coefficient families
  //   coef_0(d) = dt,  coef_1(d) = (2d+1) dt^2/2,  coef_2(d) = (3d^2+3d+1) dt^3/6
  // by ONE backward scan per state row over the moments
  //   S0[st] = sum e_k,  S1[st] = sum (k-st) e_k,  S2[st] = sum (k-st)^2 e_k :
  //   S2[st] = S2[st+1] + 2 S1[st+1] + S0[st+1],  S1[st] = S1[st+1] + S0[st+1],  S0[st] = S0[st+1] + e_st
  // -- O(h) work on twelve threads of wave 3, concurrent with E_00 / E_11 on waves 0-2,
  // instead of O(h^2) LDS-bound dot products on every thread.
  if constexpr (HMAX <= 16) {
    // (round 6) The same three moments as SUFFIX SCANS inside 16-lane rows: state row r = one DPP row, lane = horizon step.
    //   S0 = suffix sum of e;  S1[st] = sum_{j > st} S0[j];  T[st] = sum_{j > st} S1[j] = sum_k C(k - st, 2) e_k;  S2 = 2 T + S1
    // -- four shift-and-add steps per scan (row_shl with zeros shifted in) on 192 threads instead of a 16-step dependent
    // recurrence on 12: the stage was as long as that chain (3.9 k cycles for 330 vector instructions).  Summation order
    // differs from the recurrence's: g changes in the last bits (1e-16 relative).
    if (tid >= 64 && tid < 256) {
      const int row = (tid - 64) >> 4, st = tid & 15;
      auto shl = [](double x, auto nc) __attribute__((always_inline)) {  // value of lane + N of this 16-lane row, 0 beyond it
        constexpr int N = decltype(nc)::value;
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, 0x100 + N, 0xf, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), 0x100 + N, 0xf, 0xf, true);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      };
      auto suffix = [&](double x) __attribute__((always_inline)) {
        x += shl(x, std::integral_constant<int, 1>{});
        x += shl(x, std::integral_constant<int, 2>{});
        x += shl(x, std::integral_constant<int, 4>{});
        x += shl(x, std::integral_constant<int, 8>{});
        return x;
      };
      const double e = (st < h) ? Aa.e[st * 12 + row] : 0.0;
      const double S0 = suffix(e);
      const double S1 = suffix(shl(S0, std::integral_constant<int, 1>{}));
      const double T2 = suffix(shl(S1, std::integral_constant<int, 1>{}));
      const double S2 = 2.0 * T2 + S1;
      const double dt1 = P.dt, dt2 = dt1 * dt1, dt3 = dt2 * dt1;
      if (st < h) {
        Aa.s[0][st * 12 + row] = dt1 * S0;
        // coef_1(d) = (2d + 1) dt^2 / 2 (exact zero-order hold) or d dt^2 (SparseCMPC: B_d = B dt, c2d)
        Aa.s[1][st * 12 + row] = (P.model == 1) ? dt2 * S1 : dt2 * S1 + (0.5 * dt2) * S0;
        Aa.s[2][st * 12 + row] = (0.5 * dt3) * (S2 + S1) + (dt3 / 6.0) * S0;
      }
    }
  } else if (tid >= 192 && tid < 204) {
    const int row = tid - 192;
    double ek[HMAX];
#pragma unroll
    for (int k = 0; k < HMAX; ++k) ek[k] = Aa.e[(k < h ? k : 0) * 12 + row];
    const double dt1 = P.dt, dt2 = dt1 * dt1, dt3 = dt2 * dt1;
    double S0 = 0.0, S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int st = HMAX - 1; st >= 0; --st) {
      if (st < h) {
        S2 = S2 + 2.0 * S1 + S0;
        S1 = S1 + S0;
        S0 = S0 + ek[st];
        Aa.s[0][st * 12 + row] = dt1 * S0;
        // coef_1(d) = (2d + 1) dt^2 / 2 (exact zero-order hold) or d dt^2 (SparseCMPC: B_d = B dt, c2d)
        Aa.s[1][st * 12 + row] = (P.model == 1) ? dt2 * S1 : dt2 * S1 + (0.5 * dt2) * S0;
        Aa.s[2][st * 12 + row] = (0.5 * dt3) * (S2 + S1) + (dt3 / 6.0) * S0;
      }
    }
  }
  __syncthreads();  // ---- barrier 2
  QMPC_TICK(2);
#if QMPC_STAGE_EARLY
  // (the robots of a staged CU that are not its hardest step back one level from the assembly on, like the hint's robots after
  //  stage 0: configs[1] 2.73e7 -> 2.76e7; one level lower still throughout their sweep: no change -- tools/dbg/run_stage_variants.sh)
  if constexpr (PRIO && !CMD && !ADMM && !BIG && !PHA) {
    if (PK.prio_cu && __builtin_amdgcn_readfirstlane(S.prio_rank) == 1) __builtin_amdgcn_s_setprio(2);
  }
#endif
  {
    const double keep1[2] = {alpha, x_drag};
    (void)keep1;
    QMPC_STOP(1, keep1);
  }
  if constexpr (PHA) {
    // (the pool of work items is bounded and the host never launches more robots per chunk than it holds; an index
    //  beyond it means corrupt counters: the robot is reported, nothing is written out of bounds)
    if (S.evslot - P.wk_base >= P.wk_cap) {
      if (tid < 12) P.grf[(size_t)rid * 12 + tid] = 0.f;
      if (cmdm && P.f_ff && tid < 12) P.f_ff[(size_t)rid * 12 + tid] = 0.f;
      if (tid == 0) {
        P.status[rid] = QMPC_DEV_ST_WS_FULL;
        if (P.iters) P.iters[rid] = 0;
        if (cmdm) cmd_finish_state();
      }
      __syncthreads();
      return false;
    }
  }

#ifdef QMPC_BIG_STAMP  // (profiling build, tools/big_phase.py: shader-clock stamps of block step QMPC_BIG_STAMP and of the stages around the sweep)
#define QMPC_BIG_TICK(k) do { if (dbg_clk && tid == 0) dbg_clk[(k)] = clock64(); } while (0)
#define QMPC_BIG_STEP_TICK(k) do { if (dbg_clk && tid == 0 && k0 == NB * QMPC_BIG_STAMP) dbg_clk[(k)] = clock64(); } while (0)
#else
#define QMPC_BIG_TICK(k) do { } while (0)
#define QMPC_BIG_STEP_TICK(k) do { } while (0)
#endif
  if constexpr (BIG) {
    // ------------------------------------------------------------ the large problems (192 < n_r <= 432)
    // The reference's interface takes up to K_MAX_GAIT_SEGMENTS = 36 segments: all four feet down at 36 segments is
    // n_r = 432, a trot 216.  Nothing of that size fits the register file: H is written to the robot's work item in
    // global memory (448 x 448), inverted there by a symmetric BLOCK sweep -- 16 pivots per step: the pivot block
    // factorised in LDS, the pivot columns C and F = C P^-1 (through the factor) staged in LDS, A <- A - F C^T through L2, the same sweep
    // operator as stage 3 (A ends as -H^-1) -- and handed to the engine kernel like every other work item.  This path is
    // about coverage of the interface, not speed: ~3 ms per robot and CU (the reference's dense qpOASES needs seconds).
    constexpr int LDB = QMPC_BIG_LD, NB = 16, PD = NB + 1, NMAX = 3 * Smem<RB>::SLOTS;
    const int item = S.evslot;
    GlobalF64* const A = (GlobalF64*)P.wk_hinv + (size_t)item * ((size_t)LDB * LDB);
    // (the item is this workgroup's alone until the kernel ends: workgroup scope -- the L2 may serve it; at agent scope every
    //  load went out to the fabric)
    auto ldA = [&](int r, int cidx) __attribute__((always_inline)) {
      return __hip_atomic_load(A + (size_t)r * LDB + cidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const int wv = tid >> 6;
    constexpr int NWV = NT / 64;
    const double dm2 = x_drag * inv_m * inv_m;

    QMPC_BIG_TICK(8);
    // ---- g (one variable per thread: n <= 432 < 768), as in stage 2
    double gmine = 0.0;
    if (tid < n) {
      const int ki = S.sidx[tid / 3], ax = tid % 3;
      const int st = ki >> 2, b = ki & 3;
      const double* s0 = &Aa.s[0][st * 12];
      const double* s1 = &Aa.s[1][st * 12];
      double acc = s0[9 + ax] * inv_m + s1[3 + ax] * inv_m;
#pragma unroll
      for (int l = 0; l < 3; ++l) acc += Aa.Mb[b][3 * l + ax] * s0[6 + l] + Aa.Nb[b][3 * l + ax] * s1[l];
      if (drag && ax == 0) acc += (x_drag * inv_m) * (s1[11] + Aa.s[2][st * 12 + 5]);
      gmine = 2.0 * acc;
    }
    // ---- H = 2 (tau (x) E_00 + sigma (x) E_11 + x_drag terms + alpha I) element by element (SolverMPC.cpp:395): a wave per
    // row, lanes along the columns
    {
      const double w11 = Aa.W[11] * dm2, w5 = Aa.W[5] * dm2, w5x = Aa.W[5] * (x_drag * dm2);
      for (int r = wv; r < n; r += NWV) {
        const int ki = S.sidx[r / 3], ai = r % 3, si = ki >> 2, u = 3 * (ki & 3) + ai;
        for (int j = lane; j < n; j += 64) {
          const int kj = S.sidx[j / 3], cax = j % 3, sj = kj >> 2;
          const int cidx = si * hs + sj, tidx = sj * hs + si, eidx = u * 12 + 3 * (kj & 3) + cax;
          double v = Aa.ct0[cidx] * Aa.E00[eidx] + Aa.ct4[cidx] * Aa.E11[eidx];
          if (drag) {
            double add = 0.0;
            if (ai == 2 && cax == 0) add = Aa.ct1[cidx] * w11 + Aa.ct5[cidx] * w5;
            if (ai == 0 && cax == 2) add = Aa.ct1[tidx] * w11 + Aa.ct5[tidx] * w5;
            if (ai == 0 && cax == 0) add = Aa.ct8[cidx] * w5x;
            v += add;
          }
          double hv = 2.0 * (v + ((r == j) ? alpha : 0.0));
          if (P.admm_mode != 0 && r == j) {
            // JCQP alternate on a large problem (use_jcqp = 1 / 2 at horizons above 16): the KKT matrix reduced to the x
            // block, M = P + sigma I + A^T R A with its DIAGONAL friction part -- as in stage 2 of the ADMM instantiations
            const double fk = S.fmaxk[r / 3];
            const double rho4 = (__builtin_fabs(fk) < 1e-10) ? P.admm_rho * 1e3 : (fk > 1e10 ? 1e-6 : P.admm_rho);
            hv += P.admm_sigma + ((ai < 2) ? 2.0 * 1e-6 * P.mu_inv * P.mu_inv : 4.0 * 1e-6 + rho4);
          }
          A[(size_t)r * LDB + j] = hv;
        }
      }
    }
    __syncthreads();  // (the tables are dead: the union becomes scratch; the rows of H are in L2)
    QMPC_BIG_TICK(9);
    double* const Cp = reinterpret_cast<double*>(&S.u);  // C[NMAX][PD]: the pivot columns
    double* const Fp = Cp + NMAX * PD;                    // F[NMAX][PD] = C P^-1
    double* const Pm = Fp + NMAX * PD;                    // the pivot block, then its Cholesky factor L [NB][PD]
    double* const Qm = Pm + NB * PD;                      // L^-1
    double* const Rm = Qm + NB * PD;                      // P^-1 = L^-T L^-1
    double* const gl = Rm + NB * PD;                      // g [NMAX]
    constexpr int NTLMAX = (NMAX + 15) / 16, TLW = (NTLMAX * (NTLMAX + 1) / 2 + NWV - 1) / NWV;
    unsigned short* const tl = reinterpret_cast<unsigned short*>(gl + NMAX);  // the waves' tile lists [NWV][TLW]
    static_assert(sizeof(double) * (2 * NMAX * PD + 3 * NB * PD + NMAX) + 2 * NWV * TLW <= sizeof(S.u), "scratch of the block sweep");
    if (tid < NMAX) gl[tid] = gmine;
    // the 16 x 16 tiles (ti, tj <= ti) of the lower triangle, dealt round-robin to the waves ONCE per robot: tile p -> wave
    // p mod NWV, position p / NWV (every step walks the same lists and skips the tiles of its pivot row / column block)
    {
      const int ntl0 = (n + 15) >> 4, ntiles = ntl0 * (ntl0 + 1) / 2;
      for (int pq = tid; pq < ntiles; pq += NT) {
        int ti0 = (int)((__builtin_sqrtf(8.f * (float)pq + 1.f) - 1.f) * 0.5f);
        while (ti0 * (ti0 + 1) / 2 > pq) --ti0;
        while ((ti0 + 1) * (ti0 + 2) / 2 <= pq) ++ti0;
        tl[(pq % NWV) * TLW + pq / NWV] = (unsigned short)((ti0 << 8) | (pq - ti0 * (ti0 + 1) / 2));
      }
    }
    bool bad = false;
#pragma unroll 1
    for (int k0 = 0; k0 < n; k0 += NB) {
      const int bsz = (n - k0 < NB) ? n - k0 : NB;
      QMPC_BIG_STEP_TICK(0);
      // 1. pivot block (identity-padded) and pivot columns -> LDS
      if (tid < NB * NB) {
        const int pi = tid / NB, pj = tid % NB;
        // (only the LOWER triangle of the sweep state is kept current: the state is symmetric, and the update streams the
        //  matrix through L2 / HBM once per step -- that traffic is what this path costs)
        Pm[pi * PD + pj] = (pi < bsz && pj < bsz) ? ldA(k0 + (pi > pj ? pi : pj), k0 + (pi > pj ? pj : pi)) : ((pi == pj) ? 1.0 : 0.0);
      }
      for (int e = tid; e < n * NB; e += NT) {
        const int r = e / NB, kk = e % NB, cc = k0 + kk;
        Cp[r * PD + kk] = (kk < bsz) ? (r >= cc ? ldA(r, cc) : ldA(cc, r)) : 0.0;
      }
      __syncthreads();
      QMPC_BIG_STEP_TICK(1);
      // 2. P = L L^T in place (the lower triangle of Pm becomes L).  NOT an explicit Gauss-Jordan inverse of the block: the
      // four feet of a step push the body almost identically, a 16 x 16 block of the sweep state has eigenvalues down to
      // the regulariser alpha, and F = C P^-1 through an unpivoted explicit inverse loses the digits the whole sweep
      // needs (measured on all feet down at 36 segments: H^-1 to 1e-1 -- through the factor: 4e-12; oracle-side study)
      // ONE wave, in registers: lane i (of every 16) holds row i of the block; a column's entries reach the other rows
      // through DPP broadcasts (row_newbcast), 1 / sqrt(d) is a seed and two Newton steps -- straight-line code, ~5 k
      // cycles with L^-1, where 256 threads with two workgroup barriers per column took 13.5 k and one wave going through
      // LDS 15 k (a dozen predicated LDS accesses per column, each its own branch)
      if (wv == 0) {
        const int i16 = lane & 15;
        double pr[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) pr[j] = Pm[i16 * PD + j];
        StaticFor<0, NB>::run([&](auto ppc) __attribute__((always_inline)) {
          constexpr int pp = decltype(ppc)::value;
          double d = 0.0;
          fmac_rowbcast<pp>(d, pr[pp], 1.0);
          bad |= !(d > 0.0);
          double rl = __builtin_amdgcn_rsq(d);
          double e = __builtin_fma(-d * rl, rl, 1.0);
          rl = __builtin_fma(0.5 * rl, e, rl);
          e = __builtin_fma(-d * rl, rl, 1.0);
          rl = __builtin_fma(0.5 * rl, e, rl);
          const double l = pr[pp] * rl;  // L_ip; row pp keeps 1 / L_pp instead (all that is needed of the diagonal)
          pr[pp] = (i16 == pp) ? rl : l;
          const double cl = (i16 > pp) ? l : 0.0;
          fmac16_rowbcast(pr, cl, -cl);  // P_ij -= L_ip L_jp for i, j > pp (columns <= pp: the broadcast value is 0)
        });
        if (lane < NB) {
#pragma unroll
          for (int j = 0; j < NB; ++j) Pm[lane * PD + j] = pr[j];  // (row i: L_i0 .. L_i,i-1, 1 / L_ii)
        }
        __builtin_amdgcn_wave_barrier();
        QMPC_BIG_STEP_TICK(7);
        // 2b. L^-1, column c on lane c (of every 16): forward substitution, the entries of L broadcast from LDS row by row
        // (entries above the diagonal of L^-1 come out as exact zeros: no predicates; a scheduling barrier per row, or the
        //  136 loads are hoisted to the top and spill)
        {
          double x[NB];
          StaticFor<0, NB>::run([&](auto ic) __attribute__((always_inline)) {
            constexpr int i2 = decltype(ic)::value;
            double acc = (i2 == i16) ? 1.0 : 0.0;
            StaticFor<0, i2>::run([&](auto mc) __attribute__((always_inline)) {
              constexpr int m = decltype(mc)::value;
              acc = __builtin_fma(-Pm[i2 * PD + m], x[m], acc);
            });
            x[i2] = acc * Pm[i2 * PD + i2];
            if (lane < NB) Qm[i2 * PD + lane] = x[i2];
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        __builtin_amdgcn_wave_barrier();
        {
          // P^-1 = L^-T L^-1: one 16 x 16 x 16 product, both operands the same LDS entry (A = Li^T[row lc][k], B = Li[k][col lc])
          typedef double v4d __attribute__((ext_vector_type(4)));
          const int lc = lane & 15, rq = lane >> 4;
          v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            const double v = Qm[(4 * kc + rq) * PD + lc];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) Rm[(rq + 4 * g) * PD + lc] = acc[g];
        }
      }
      __syncthreads();
      QMPC_BIG_STEP_TICK(2);
      // 3. F = C P^-1 = (C L^-T) L^-1: two products with the explicit inverse of the FACTOR (as accurate as the two
      // substitutions per row they replace -- oracle-side study: 5e-12 either way at n_r = 432, where C (L^-T L^-1) as one
      // product loses four digits --, and on the matrix cores: 16 rows per wave at a time, the intermediate through LDS)
      {
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int lc = lane & 15, rq = lane >> 4;
        const int ntl = (n + 15) >> 4;
        for (int t = wv; t < ntl; t += NWV) {
          const int R0 = 16 * t;
          v4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            const double aop = (R0 + lc < n) ? Cp[(R0 + lc) * PD + 4 * kc + rq] : 0.0;
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Qm[lc * PD + 4 * kc + rq], y, 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) Fp[(R0 + rq + 4 * g) * PD + lc] = y[g];
          __builtin_amdgcn_wave_barrier();
          v4d f = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kc = 0; kc < 4; ++kc)
            f = __builtin_amdgcn_mfma_f64_16x16x4f64(Fp[(R0 + lc) * PD + 4 * kc + rq], Qm[(4 * kc + rq) * PD + lc], f, 0, 0, 0);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int g = 0; g < 4; ++g) Fp[(R0 + rq + 4 * g) * PD + lc] = f[g];
        }
      }
      __syncthreads();
      QMPC_BIG_STEP_TICK(3);
      // 4. the sweep step on the lower triangle.  (a) The pivot columns / rows <- F and the pivot block <- -P^-1: n x 16
      // entries, one thread each.  (b) Everything else, A <- A - F C^T, IS a GEMM (rank 16 per step) and runs on the matrix
      // cores: 16 x 16 tiles of the triangle dealt to the waves, four v_mfma_f64_16x16x4 per tile -- accumulator lane
      // (lc = lane & 15, rq = lane >> 4), register g = element (row rq + 4 g, column lc); A operand = -F[row lc][k = rq],
      // B operand = C[column lc][k = rq] -- i.e. 8 LDS reads and 4 matrix instructions per lane and tile where the vector
      // version had 64 reads and 64 multiply-adds
      for (int e = tid; e < n * NB; e += NT) {
        const int r = e / NB, m = e % NB;
        if (m < bsz) {
          if (r >= k0 + bsz) A[(size_t)r * LDB + k0 + m] = Fp[r * PD + m];
          else if (r < k0) A[(size_t)(k0 + m) * LDB + r] = Fp[r * PD + m];
          else if (m <= r - k0) A[(size_t)r * LDB + k0 + m] = -Rm[(r - k0) * PD + m];
        }
      }
      {
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int ntl = (n + 15) >> 4, kb = k0 >> 4;
        const int lc = lane & 15, rq = lane >> 4;
        const int ntiles = ntl * (ntl + 1) / 2, cntw = (ntiles > wv) ? (ntiles - wv + NWV - 1) / NWV : 0;
        const unsigned short* const mytl = tl + wv * TLW;
        int idx = 0;
        // the next tile of this wave outside the pivot row / column block (false: none left)
        auto next_tile = [&](int& ti, int& tj) __attribute__((always_inline)) {
          while (idx < cntw) {
            const int e = mytl[idx++];
            ti = e >> 8;
            tj = e & 255;
            if (ti != kb && tj != kb) return true;
          }
          return false;
        };
        auto fetch = [&](int ti, int tj) __attribute__((always_inline)) {
          v4d acc;
          const int col = 16 * tj + lc;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * ti + rq + 4 * g;
            acc[g] = (row < n && col < n) ? ldA(row, col) : 0.0;
          }
          return acc;
        };
        // (the NEXT tile's entries are requested before the current tile is multiplied: a load is an HBM round trip)
        int ti = 0, tj = 0, ni = 0, nj = 0;
        bool have = next_tile(ti, tj);
        v4d acc_n = {0.0, 0.0, 0.0, 0.0};
        if (have) acc_n = fetch(ti, tj);
#pragma unroll 1
        while (have) {
          v4d acc = acc_n;
          const bool more = next_tile(ni, nj);
          if (more) acc_n = fetch(ni, nj);
          const int R0 = 16 * ti, col = 16 * tj + lc, fr_row = R0 + lc;
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            const double aop = (fr_row < n) ? -Fp[fr_row * PD + 4 * kc + rq] : 0.0;
            const double bop = (col < n) ? Cp[col * PD + 4 * kc + rq] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = R0 + rq + 4 * g;
            if (row < n && col <= row) A[(size_t)row * LDB + col] = acc[g];
          }
          ti = ni;
          tj = nj;
          have = more;
        }
      }
      QMPC_BIG_STEP_TICK(4);
      __syncthreads();  // (every store of the step is in L2 before the next step reads)
      QMPC_BIG_STEP_TICK(5);
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
      if (tid == 0) S.status |= QMPC_DEV_ST_NOT_PD;
    }
    QMPC_BIG_TICK(10);
    // ---- A = -H^-1 (lower triangle): the work item gets +H^-1, mirrored into the full matrix (the engine's reads find it
    // whichever way they go); then x_u = -H^-1 g over full rows
    // Tile by tile (the waves' lists again), two tiles per trip with every load issued before the first use; the mirrored
    // tile goes through a 16 x 17 LDS patch of the wave so that both copies leave as whole 128-byte rows (written entry
    // by entry down a column -- 64 partial lines per store -- this pass took 0.45 M of 5.2 M cycles at n_r = 432)
    constexpr int NQ = (NMAX + 63) / 64;
    {
      typedef double v4d __attribute__((ext_vector_type(4)));
      const int lc = lane & 15, rq = lane >> 4;
      const int ntl = (n + 15) >> 4, ntiles = ntl * (ntl + 1) / 2, cntw = (ntiles > wv) ? (ntiles - wv + NWV - 1) / NWV : 0;
      const unsigned short* const mytl = tl + wv * TLW;
      double* const T = Cp + wv * (2 * 16 * PD);  // (the pivot columns are dead)
      for (int i0 = 0; i0 < cntw; i0 += 2) {
        v4d v[2];
        int ti[2], tj[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int e = (i0 + w < cntw) ? mytl[i0 + w] : 0xffff;
          ti[w] = e >> 8;
          tj[w] = e & 255;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * ti[w] + rq + 4 * g, col = 16 * tj[w] + lc;
            v[w][g] = (row < n && col <= row) ? -ldA(row, col) : 0.0;
          }
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * ti[w] + rq + 4 * g, col = 16 * tj[w] + lc;
            if (row < n && col <= row) A[(size_t)row * LDB + col] = v[w][g];
            T[w * 16 * PD + (rq + 4 * g) * PD + lc] = v[w][g];
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int w = 0; w < 2; ++w) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // entry (row', col') of the mirrored tile = entry (col', row') of the tile
            const int row = 16 * tj[w] + rq + 4 * g, col = 16 * ti[w] + lc;
            if (col < n && row < col) A[(size_t)row * LDB + col] = T[w * 16 * PD + lc * PD + rq + 4 * g];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    QMPC_BIG_TICK(14);
    for (int r = wv; r < n; r += 2 * NWV) {
      double v[2][NQ], gq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) gq[q] = (lane + 64 * q < n) ? gl[lane + 64 * q] : 0.0;
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int rr = r + w * NWV, j = lane + 64 * q;
          v[w][q] = (rr < n && j < n) ? ldA(rr, j) : 0.0;
        }
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc = __builtin_fma(-v[w][q], gq[q], acc);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft);
        if (lane == 0 && r + w * NWV < n) Fp[r + w * NWV] = acc;  // (x_u, parked where F was)
      }
    }
    __syncthreads();
    QMPC_BIG_TICK(15);
    constexpr int LDX = QMPC_BIG_LD;
    // (JCQP alternate: the ADMM starts cold and needs the gradient, not the unconstrained minimiser)
    if (tid < LDX) P.wk_xu[(size_t)item * LDX + tid] = (tid < n) ? (P.admm_mode != 0 ? gl[tid] : Fp[tid]) : 0.0;
    QmpcWorkHdr* const hd = P.wk_hdr + item;
    if (tid < QMPC_WK_SLOTS_MAX) {
      hd->sidx[tid] = (tid < Smem<RB>::SLOTS) ? S.sidx[tid] : (unsigned char)0;
      hd->fmaxk[tid] = (tid < nst) ? (float)S.fmaxk[tid] : 0.f;
    }
    if (tid == 0) {
      hd->rid = rid;
      hd->n = n;
      hd->nst = nst;
      hd->status0 = S.status;
    }
    if (tid < WAVE) {
      // (hardest robots first, as in the other producers: rows violated at x_u)
      int viol = 0;
      for (int sl = tid; sl < nst; sl += 64) {
        const double x0 = Fp[3 * sl], x1 = Fp[3 * sl + 1], x2 = Fp[3 * sl + 2];
        const double fx = P.mu_inv * x0, fy = P.mu_inv * x1, nt = -P.tol, ifr = P.inv_fr_norm;
        viol += ((fx + x2) * ifr < nt) + ((x2 - fx) * ifr < nt) + ((fy + x2) * ifr < nt) + ((x2 - fy) * ifr < nt) +
                (S.fmaxk[sl] - x2 < nt);
      }
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) viol += __shfl_xor(viol, sft);
      if (tid == 0) {
        const int lvl = viol >> 4;
        const int b = QMPC_ORDER_BUCKETS - 1 - (lvl < QMPC_ORDER_BUCKETS - 1 ? lvl : QMPC_ORDER_BUCKETS - 1);
        const int pos = atomicAdd(P.wk_bucket + b, 1);
        P.wk_order[(size_t)b * P.wk_cap + P.wk_base + pos] = item;
      }
    }
    __syncthreads();
    return false;
  }

  // ------------------------------------------------------------ stage 2
  // gradient g_red -> LDS, Hessian rows straight into registers.
  if (tid < NP) {
    double gv = 0.0;
    if (tid < n) {
      const int ki = S.sidx[tid / 3], ax = tid % 3;
      const int st = ki >> 2, b = ki & 3;
      const double* s0 = &Aa.s[0][st * 12];
      const double* s1 = &Aa.s[1][st * 12];
      // g = 2 sum_p B_p^T s_p :  B0 rows 6..8 = M_b, 9..11 = I/m ;
      //                          B1 rows 0..2 = N_b, 3..5 = I/m, row 11 = x_drag/m on fx ;
      //                          B2 row 5 = x_drag/m on fx
      double acc = s0[9 + ax] * inv_m + s1[3 + ax] * inv_m;
#pragma unroll
      for (int l = 0; l < 3; ++l)
        acc += Aa.Mb[b][3 * l + ax] * s0[6 + l] + Aa.Nb[b][3 * l + ax] * s1[l];
      if (drag && ax == 0) acc += (x_drag * inv_m) * (s1[11] + Aa.s[2][st * 12 + 5]);
      gv = 2.0 * acc;
    }
    Sw.g[tid] = gv;
  }

  if (dbg_clk && tid == 0) dbg_clk[13] = clock64();
  double a[CW];
  {
    const bool rowok = i < n;
    const int ki = rowok ? (int)S.sidx[(i / 3) & 63] : 0;  // (i/3 < 64 always; the load is unconditional)
    const int ai = rowok ? i % 3 : 0;
    const int si = rowok ? (ki >> 2) : h;  // (identity padding: the tables' zero row)
    const int u = 3 * (ki & 3) + ai;
    const double dm2 = x_drag * inv_m * inv_m;
    // JCQP alternate: A^T R A is DIAGONAL for the friction block (its columns are orthogonal):
    // diag(2 rho_inf / mu^2, 2 rho_inf / mu^2, 4 rho_inf + rho_4) per foot-step, rho_inf for the four rows
    // whose upper bound is BIG_NUMBER (QpProblem.cpp:278-280), rho_4 for fz <= f_max (equality when u = 0)
    double admm_diag = 0.0;
    if constexpr (ADMM) {
      const double fk = rowok ? S.fmaxk[(i / 3) & 63] : 1.0;
      const double rho4 = (__builtin_fabs(fk) < 1e-10) ? P.admm_rho * 1e3 : (fk > 1e10 ? 1e-6 : P.admm_rho);
      admm_diag = P.admm_sigma + ((ai < 2) ? 2.0 * 1e-6 * P.mu_inv * P.mu_inv : 4.0 * 1e-6 + rho4);
    }
    // the CW columns of this thread walk at most CW/3 + 2 stance slots: fetch
    // their foot-step ids in one batch, then the table / E loads in groups of 4
    constexpr int NSL = CW / 3 + 2;
    const int cslot0 = (c * CW) / 3, cax0 = (c * CW) % 3;
    // (the largest class packs the ids four to a register: it has no VGPRs to spare)
    constexpr bool PACK = (RB == 3);
    int kjs[PACK ? (NSL + 3) / 4 : NSL];
    if constexpr (PACK) {
#pragma unroll
      for (int q4 = 0; q4 < (NSL + 3) / 4; ++q4) {
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int sl = cslot0 + 4 * q4 + b;
          const unsigned kv = S.sidx[sl < 63 ? sl : 63];
          w |= ((sl < nst) ? kv : (unsigned)(4 * h)) << (8 * b);  // (beyond the stance list: the tables' zero column)
        }
        kjs[q4] = (int)w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NSL; ++q) {  // unconditional (clamped) loads: all in flight together
        const int sl = cslot0 + q;
        const int kv = (int)S.sidx[sl < 63 ? sl : 63];
        kjs[q] = (sl < nst) ? kv : 4 * h;  // (beyond the stance list: the tables' zero column)
      }
    }
    auto kj_at = [&](int q) __attribute__((always_inline)) {  // q is a compile-time constant at every call site
      if constexpr (PACK) return (int)(((unsigned)kjs[q >> 2] >> (8 * (q & 3))) & 0xffu);
      else return kjs[q];
    };
    // branch-free element loop (indices are always in range: out-of-range rows /
    // columns read slot 0 and are overwritten by the padding select), so the LDS
    // loads of a group of elements are in flight together.  The (slot, axis) walk
    // of the columns is compile-time once the start axis (wave-uniform) is fixed.
    auto fill = [&](auto cax0c) __attribute__((always_inline)) {
      constexpr int CAX0 = decltype(cax0c)::value;
      if constexpr (RB == 3) {
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) {
          const int kj = kj_at((CAX0 + jj) / 3), cax = (CAX0 + jj) % 3;
          const int cidx = si * hs + (kj >> 2), eidx = u * 12 + 3 * (kj & 3) + cax;
          // H = 2 (tau (x) E_00 + sigma (x) E_11 + x_drag terms + alpha I), SolverMPC.cpp:395
          a[jj] = Aa.ct0[cidx] * Aa.E00[eidx] + Aa.ct4[cidx] * Aa.E11[eidx];
          if ((jj & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // bound the load hoisting (VGPR pressure)
        }
      } else {
        // the table entries tau, sigma depend on the column's FOOT-STEP only: one pair of loads per stance slot
        // the thread's columns walk, not per column (the stage is bound by LDS bytes: 64 -> 46 doubles per thread
        // at CW = 16); same products, same sums
        constexpr int NS = (CAX0 + CW - 1) / 3 + 1;  // slots touched
        double t0[NS], t4[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          // (identity padding: row h / column h of the tables are zero, so the products of a row beyond n_r or of a slot
          //  beyond the stance list vanish by themselves -- no compare and no select per element at the end)
          const int cidx = si * hs + (kj_at(q) >> 2);
          t0[q] = Aa.ct0[cidx];
          t4[q] = Aa.ct4[cidx];
        }
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) {
          constexpr int dummy = 0;
          (void)dummy;
          const int q = (CAX0 + jj) / 3, cax = (CAX0 + jj) % 3;
          const int eidx = u * 12 + 3 * (kj_at(q) & 3) + cax;
          // H = 2 (tau (x) E_00 + sigma (x) E_11 + x_drag terms + alpha I), SolverMPC.cpp:395
          a[jj] = t0[q] * Aa.E00[eidx] + t4[q] * Aa.E11[eidx];
          if ((jj & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the load hoisting (VGPR pressure)
        }
      }
      // (uniform, and rare.  The 128- / 192-row classes are short of registers in this stage: marked cold, the block is laid out
      //  of line and its spills stay inside it; the smaller classes allocate better without the hint)
      if ((RB == 2 || RB == 3) ? __builtin_expect(drag, false) : drag) {
        // E_01 / E_12 couple (z of foot-step i, x of foot-step j); E_10 / E_21 the
        // transposed pair (C_qp[i][j] == C_pq[j][i]); E_22 couples x with x.
        const double w11 = Aa.W[11] * dm2, w5 = Aa.W[5] * dm2, w5x = Aa.W[5] * (x_drag * dm2);
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) {
          const int kj = kj_at((CAX0 + jj) / 3), cax = (CAX0 + jj) % 3;
          const int sj = kj >> 2, cidx = si * hs + sj, tidx = sj * hs + si;
          double add = 0.0;
          if (ai == 2 && cax == 0) add = Aa.ct1[cidx] * w11 + Aa.ct5[cidx] * w5;
          if (ai == 0 && cax == 2) add = Aa.ct1[tidx] * w11 + Aa.ct5[tidx] * w5;
          if (ai == 0 && cax == 0) add = Aa.ct8[cidx] * w5x;
          a[jj] += add;
          if (RB == 3 || (jj & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (as above: the largest class has no VGPRs to spare)
        }
      }
    };
    if (cax0 == 0) fill(std::integral_constant<int, 0>{});
    else if (cax0 == 1) fill(std::integral_constant<int, 1>{});
    else fill(std::integral_constant<int, 2>{});
    if constexpr (RB != 3) {
      // the padding entries are exact zeros already; its diagonal gets 2 (0 + 1/2) = 1, the real rows' 2 (a + alpha): one
      // compare, one select and two operations per element (same expression as before on every real entry: same bits)
      const int dd = i - c * CW;
      const double dval = rowok ? alpha : 0.5;
      const double dadm = (ADMM && rowok) ? admm_diag : 0.0;
#pragma unroll
      for (int jj = 0; jj < CW; ++jj) {
        double v = 2.0 * (a[jj] + ((dd == jj) ? dval : 0.0));
        if constexpr (ADMM) v += (dd == jj) ? dadm : 0.0;  // the KKT matrix reduced to the x block
        a[jj] = v;
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < CW; ++jj) {
        const int j = c * CW + jj;
        double v = 2.0 * (a[jj] + ((i == j) ? alpha : 0.0));
        if constexpr (ADMM) v += (i == j) ? admm_diag : 0.0;  // the KKT matrix reduced to the x block
        a[jj] = (rowok && j < n) ? v : ((i == j) ? 1.0 : 0.0);  // identity padding
      }
    }
  }
  if (P.dbg_H) {
    double* Hd = P.dbg_H + (size_t)rid * QMPC_DBG_LD * QMPC_DBG_LD;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) Hd[(size_t)i * QMPC_DBG_LD + c * CW + jj] = a[jj];
    if (tid < NP) P.dbg_g[(size_t)rid * QMPC_DBG_LD + tid] = Sw.g[tid];
  }
  QMPC_TICK(3);
  QMPC_STOP(2, a);

  // ------------------------------------------------------------ stage 3
  // Symmetric Gauss-Jordan sweeps, TWO pivots per barrier: a <- -H^-1.
  // Pivot columns k0 = 2m, k1 = k0+1 live in column group k0 / CW, registers
  // k0 % CW and +1, and are broadcast through a double-buffered pair of LDS
  // vectors (row k == column k by symmetry).  Every thread applies pivot k0 to
  // its copy of column k1 locally (c1' = c1 - c0 e/d0), so the second pivot
  // needs no second broadcast.  With C = A[:, {k0,k1}] and P = A[{k0,k1},{k0,k1}]:
  //     A <- A - F C^T ,  F = C P^-1 ;  pivot columns <- F ;  pivot block <- -P^-1
  // i.e. two fma per element on the ORIGINAL columns.  Pivot rows need
  // (P^-1 C^T); since a_kj == c_j they get it from the same update with
  // F_k = I - P^-1, so the update has no row special-casing.
  bool notpd = false;
  bool staged = PK.hint_hard > 0;  // one-round staging of the sweep's issue priority: by the hint, or by the proxy's rank
  if constexpr (PRIO && !CMD && !ADMM && !BIG && !PHA) {
    if (PK.prio_cu) {  // (S.prio_rank: written before stage 1's barriers)
      const int rank = __builtin_amdgcn_readfirstlane(S.prio_rank);
      if (QMPC_DBG_ITER == 0 && dbg_clk && tid == 0) dbg_clk[14] = rank;  // (profiling hook)
      hard = rank == 0;
      hfloor = hard ? 3 : 0;
      staged = rank < 2;  // (nobody posted: everybody yields as in an unstaged launch)
    }
  }
  if constexpr (C::C1) {
    // Class 1: a column group is exactly one wave (lane == row), so the wave that
    // owns the NEXT pivot pair reads its 2x2 pivot block with readlane, inverts it
    // once, and publishes the pivot columns and F = C P^-1 for everybody.  The other
    // three waves only load and do the 2 fma per element: the reciprocal and the F
    // arithmetic are not replicated 4x.
    // SYMMETRIC form (round 6).  With C' = C - [e_k0 e_k1] (the pivot columns with 1 subtracted from their
    // own diagonal entries) used BOTH as the broadcast vector and in F' = C' P^-1,
    //     A - F' C'^T = A - C' P^-1 C'^T
    // IS the sweep result in the pivot rows (C_j - P P^-1 C_j + P^-1 C_j), in the pivot columns (their transpose) and
    // everywhere else; only the 2 x 2 pivot block comes out as 2 I - P^-1 instead of -P^-1, and the producing wave
    // subtracts 2 from those two diagonal entries when it reads the pair (the update is additive: any time will do).
    // Until round 5 the pivot rows took a corrected F (twelve selects in the producing wave) and the pivot columns were
    // written back explicitly by their owner -- four selects and four moves in EVERY wave and pair, because the owner
    // test is a vector compare: 22 + 8 of the ~85 vector instructions of the producing wave's pair, 8 of the other
    // waves' 44.  A launch of several rounds is paced by exactly that instruction stream (five workgroups share a CU:
    // a wave issues every ~12 cycles, whatever it issues).  Same arithmetic otherwise; results agree with the previous
    // form to rounding (1e-13 relative in H^-1), not bit for bit.
    // the pivot arithmetic in stages, so that the producing wave can slot the rest
    // of its update into the latencies of this dependent chain
    double pd0, pe, pd1, pdet, px, pc0, pc1, pm0, pm1;
    auto prod_read = [&](double c0n, double c1n, int k0n) __attribute__((always_inline)) {
      pd0 = readlane_f64(c0n, k0n);
      pe = readlane_f64(c0n, k0n + 1);   // A[k1][k0]   (k1 == n: identity padding column, a no-op pivot)
      pd1 = readlane_f64(c1n, k0n + 1);
      pm0 = (i == k0n) ? 1.0 : 0.0;
      pm1 = (i == k0n + 1) ? 1.0 : 0.0;
      pc0 = c0n - pm0;  // C'
      pc1 = c1n - pm1;
    };
    // 2x2 pivot block P = [[d0, e], [e, d1]] inverted through its determinant:
    // ONE reciprocal on the critical path.  P^-1 = idet [[d1, -e], [-e, d0]].
    auto prod_det = [&]() __attribute__((always_inline)) {
      pdet = __builtin_fma(pd0, pd1, -pe * pe);
      notpd |= !(pd0 > 0.0) | !(pdet > 0.0);
    };
    auto prod_rcp = [&]() __attribute__((always_inline)) { px = __builtin_amdgcn_rcp(pdet); };
    auto prod_newton = [&]() __attribute__((always_inline)) {
      const double e1 = __builtin_fma(-pdet, px, 1.0);
      px = __builtin_fma(px, e1, px);
    };
    double pnf0, pnf1;  // -F'_i0, -F'_i1 (negated: the update is a += c' * (-F'))
    double pt0, pt1;    // the adjugate's part of it, which does not wait for the reciprocal
    // F' = C' P^-1 = C' adj(P) / det:  F'_i0 = (d1 c0'_i - e c1'_i) / det ,  F'_i1 = (d0 c1'_i - e c0'_i) / det.  The two
    // numerators are formed while the reciprocal's Newton steps are in flight; one multiplication each is all that follows it
    auto prod_adj = [&]() __attribute__((always_inline)) {
      pt0 = __builtin_fma(pe, pc1, -pd1 * pc0);
      pt1 = __builtin_fma(pe, pc0, -pd0 * pc1);
    };
    auto prod_fg = [&]() __attribute__((always_inline)) {
      pnf0 = pt0 * px;
      pnf1 = pt1 * px;
    };
    auto prod_store = [&](int mn) __attribute__((always_inline)) {
      Sw.colbuf[mn & 1][0][i] = pc0;
      Sw.colbuf[mn & 1][1][i] = pc1;
      Sw.ubuf[mn & 1][0][i] = pnf0;
      Sw.ubuf[mn & 1][1][i] = pnf1;
    };
    if (c == 0) {
      prod_read(a[0], a[1], 0);
      a[0] = __builtin_fma(-2.0, pm0, a[0]);  // the pivot block's diagonal: 2 I - P^-1 -> -P^-1
      a[1] = __builtin_fma(-2.0, pm1, a[1]);
      prod_det();
      prod_rcp();
      prod_adj();
      prod_newton();
      prod_newton();
      prod_fg();
      prod_store(0);
    }
    __syncthreads();
#define QMPC_PIN __builtin_amdgcn_sched_barrier(0)
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
#if QMPC_SWEEP_PRIO
      // Issue priority falls as the sweep advances.  Among equal priorities the SIMD
      // favours its OLDEST wave, so of the four workgroups of a CU the first one
      // dispatched used to sweep at full speed (24k cycles) and the last one at half
      // (50k) -- and a launch ends with its slowest workgroup.  A wave that is ahead now
      // yields to the ones behind it.
      if (hard || (kb == 0 && !staged)) __builtin_amdgcn_s_setprio(3);
      else if (kb <= 1 || hfloor >= 2) __builtin_amdgcn_s_setprio(2);
      else if (kb == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
#endif
      StaticFor<0, CW / 2>::run([&](auto pc) __attribute__((always_inline)) {
        constexpr int r0 = 2 * decltype(pc)::value;
        constexpr int rn0 = (r0 + 2 < CW) ? r0 + 2 : 0, rn1 = rn0 + 1;
        constexpr int G0 = 4 * (rn0 / 4), G1 = (G0 + 4) % 16, G2 = (G0 + 8) % 16, G3 = (G0 + 12) % 16;
        const int k0 = kb * CW + r0;
        if (k0 < n) {
          const int m = k0 >> 1;
          // lane l holds pivot-column entry c*16 + l%16 (the 16 columns of this wave)
          const double cv0 = Sw.colbuf[m & 1][0][c * CW + (lane & 15)];
          const double cv1 = Sw.colbuf[m & 1][1][c * CW + (lane & 15)];
          const double nu0 = Sw.ubuf[m & 1][0][i], nu1 = Sw.ubuf[m & 1][1][i];  // -F'_i0, -F'_i1
          const int kbn = (r0 + 2 < CW) ? kb : kb + 1;
          if (k0 + 2 < n && c == kbn) {
            // this wave owns the next pivot pair: its two columns first, then the
            // pivot-block inverse interleaved with the other twelve columns
#if QMPC_PROD_PRIO
            __builtin_amdgcn_s_setprio(3);  // (experiment: the producing wave's stream paces the pair)
#endif
            fmac4_rowbcast<G0>(a, cv0, nu0);
            fmac4_rowbcast<G0>(a, cv1, nu1);
            QMPC_PIN;
            prod_read(a[rn0], a[rn1], k0 + 2);
            a[rn0] = __builtin_fma(-2.0, pm0, a[rn0]);
            a[rn1] = __builtin_fma(-2.0, pm1, a[rn1]);
            QMPC_PIN;
            fmac4_rowbcast<G1>(a, cv0, nu0);
            QMPC_PIN;
            prod_det();
            prod_rcp();
            prod_adj();
            QMPC_PIN;
            fmac4_rowbcast<G1>(a, cv1, nu1);
            QMPC_PIN;
            prod_newton();
            QMPC_PIN;
            fmac4_rowbcast<G2>(a, cv0, nu0);
            QMPC_PIN;
            prod_newton();
            QMPC_PIN;
            fmac4_rowbcast<G2>(a, cv1, nu1);
            QMPC_PIN;
            prod_fg();
            QMPC_PIN;
            fmac4_rowbcast<G3>(a, cv0, nu0);
            QMPC_PIN;
            prod_store(m + 1);
            QMPC_PIN;
            fmac4_rowbcast<G3>(a, cv1, nu1);
#if QMPC_PROD_PRIO
            if (hard || (kb == 0 && !staged)) __builtin_amdgcn_s_setprio(3);
            else if (kb <= 1 || hfloor >= 2) __builtin_amdgcn_s_setprio(2);
            else if (kb == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
          } else if (c * CW < n) {
            // (a column group that lies entirely in the identity padding never changes:
            //  its pivot-column entries are zero)
            fmac16_rowbcast(a, cv0, nu0);
            fmac16_rowbcast(a, cv1, nu1);
          }
          __syncthreads();
        }
      });
    }
#undef QMPC_PIN
  } else {
    // Larger classes: a column group spans several waves (and in the 96-row class waves straddle two
    // groups), so every thread still derives F for its own row -- but the 2x2 pivot block is inverted ONCE,
    // by the thread that owns the first pivot row of the NEXT pair, as soon as its two pivot columns carry
    // this step's update: it takes A[k1][k0], A[k1][k1] from its neighbour lane (the pair's rows are
    // adjacent lanes of one 16-lane row), inverts through the determinant and publishes the three entries of
    // P^-1 next to the pivot columns.  Every other thread used to repeat the determinant, the reciprocal and
    // its two Newton steps itself: ~22 of ~95 instructions per thread and pair, in a loop that is bound by
    // single-wave issue rate (same values, same operations: results are bit-identical).
    auto lane_next = [](double x) __attribute__((always_inline)) {  // value of lane + 1 (DPP row_shl:1)
      const unsigned long long b = (unsigned long long)__double_as_longlong(x);
      // (bound_ctrl: no `old` operand to initialise -- two moves less per value on the step's critical chain)
      const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, 0x101, 0xf, 0xf, true);
      const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), 0x101, 0xf, 0xf, true);
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    auto publish_pinv = [&](double d0, double e, double d1p, int par) __attribute__((always_inline)) {
      // 2x2 pivot block P = [[d0, e], [e, d1p]] inverted through its determinant:
      // ONE reciprocal on the critical path.  P^-1 = idet [[d1p, -e], [-e, d0]].
      const double det = __builtin_fma(d0, d1p, -e * e);
      const double idet = fast_rcp(det);
      double* pv = Sw.ubuf[par][0];
      st2(pv, d1p * idet, e * idet);  // +-entries of P^-1: i11, i01, i00
      // ... and ONE number whose sign says whether both pivots were positive (a NaN fails the readers' test as well)
      st2(pv + 2, d0 * idet, (d0 < det) ? d0 : det);  // (ordered compare: a NaN determinant is what gets stored)
    };
    // SYMMETRIC form (round 6; the class-1 branch above has the derivation): the pivot columns are published with 1 subtracted
    // from their own diagonal entries, C' = C - [e_k0 e_k1]; every thread's F' = C' P^-1 then carries the pivot rows' correction
    // by itself, the update A - F' C'^T leaves F in the pivot columns by itself, and the publishing thread subtracts 2 from the
    // pivot block's two diagonal entries (2 I - P^-1 -> -P^-1).  No pivot-row branch, no pivot-column write-back.
    if (c == 0) {
      Sw.colbuf[0][0][i] = a[0];
      Sw.colbuf[0][1][i] = a[1];
      const double e_n = lane_next(a[0]), d1_n = lane_next(a[1]);
      if (i == 0) {
        publish_pinv(a[0], e_n, d1_n, 0);
        Sw.colbuf[0][0][i] = a[0] - 1.0;
        a[0] -= 2.0;
      }
      if (i == 1) {
        Sw.colbuf[0][1][i] = a[1] - 1.0;
        a[1] -= 2.0;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
#if QMPC_SWEEP_PRIO
      // see the class-1 loop: a wave that is ahead yields issue slots to the ones behind it
      if (hard || (kb == 0 && !staged)) __builtin_amdgcn_s_setprio(3);
      else if (kb <= 1 || hfloor >= 2) __builtin_amdgcn_s_setprio(2);
      else if (kb == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
#endif
      StaticFor<0, CW / 2>::run([&](auto pc) __attribute__((always_inline)) {
        constexpr int r0 = 2 * decltype(pc)::value;
        constexpr int rn0 = (r0 + 2 < CW) ? r0 + 2 : 0, rn1 = rn0 + 1;
        const int k0 = kb * CW + r0;  // (k0 + 1 == n: identity padding column, a no-op pivot)
        if (k0 < n) {
          const int m = k0 >> 1;
          QMPC_STEP_TICK(0, 0.0);
          const double* cb0 = Sw.colbuf[m & 1][0];
          const double* cb1 = Sw.colbuf[m & 1][1];
          const double* pv = Sw.ubuf[m & 1][0];
          const double c0i = cb0[i], c1i = cb1[i];
          const double i11 = pv[0], i01 = pv[1], i00 = pv[2];  // +-entries of P^-1
          // (the pivot-column values of the next pair's group: loaded HERE, with the row's own entries -- one LDS round
          //  trip for the step's critical chain; further down they would sit behind the pivot rows' branch, whose memory
          //  clobber keeps loads from moving up)
          constexpr bool TAIL8 = (CW % 16 == 8);
          constexpr int NG = (CW + 15) / 16;                         // groups (the last one 8 wide when TAIL8)
          constexpr int GN = (rn0 / 16 < NG) ? rn0 / 16 : NG - 1;    // group of the next pivot pair
          constexpr bool GN_TAIL = TAIL8 && GN == NG - 1;
          constexpr int GN0 = 16 * GN, JN = rn0 - GN0, WN = GN_TAIL ? 8 : 16;
          // ... and of every other group: ALL the step's LDS loads are in flight together (they used to be three round
          // trips in a row -- row entries, next pair's group, the other groups -- in EVERY wave: 0.35 k of a 1.2 k-cycle step)
          double cvg0[NG], cvg1[NG];
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const bool tl = TAIL8 && g == NG - 1;
            cvg0[g] = cb0[c * CW + 16 * g + (tl ? (lane & 7) : (lane & 15))];
            cvg1[g] = cb1[c * CW + 16 * g + (tl ? (lane & 7) : (lane & 15))];
          }
          const double cvn0 = cvg0[GN], cvn1 = cvg1[GN];
          notpd |= !(pv[3] > 0.0);
          // F = C P^-1 for this row:  F_i0 = i11 c0_i - i01 c1_i ,  F_i1 = i00 c1_i - i01 c0_i ; kept NEGATED (the
          // update is a += c * (-F): the sign goes into the fma instead of into extra instructions)
          // (c0i, c1i are C': the pivot rows' correction is in F' already)
          const double nu0 = __builtin_fma(-i11, c0i, i01 * c1i);
          const double nu1 = __builtin_fma(-i00, c1i, i01 * c0i);
          QMPC_STEP_TICK(1, nu0 + nu1);
          // The thread's columns in groups of 16 (class 4: 16 + 8): a group's pivot-column values are held one per lane of
          // a row of 16 (8) and broadcast inside the DPP fmac.  Only the TWO columns of the next pivot pair are updated
          // before their owner publishes them and the inverse of their 2 x 2 block -- four fmacs on the step's critical
          // chain instead of the whole group's 32 (the step of these classes is bound by that chain, not by issue:
          // tools/sweep_step_phase.py) -- then everybody, the owner included, updates the rest; the order in which a thread
          // updates its independent columns does not change a single bit of the result
          fmacn_rowbcast<JN, 2>(&a[rn0], cvn0, nu0);
          fmacn_rowbcast<JN, 2>(&a[rn0], cvn1, nu1);
          QMPC_STEP_TICK(2, a[rn0] + a[rn1]);
          const int kbn = (r0 + 2 < CW) ? kb : kb + 1;
          if (k0 + 2 < n && c == kbn) {
            // C' = C - [e_k0' e_k1']: every thread stores its entries of the two columns as they are; the two threads that hold
            // the pivot block's diagonal store theirs again with 1 subtracted (same lane, same address: LDS keeps a wave's
            // stores in order) and take 2 off their own copy (2 I - P^-1 -> -P^-1).  Branches, not masks: the 96-row class
            // has no register to spare here, and only one wave takes them
            Sw.colbuf[(m + 1) & 1][0][i] = a[rn0];
            Sw.colbuf[(m + 1) & 1][1][i] = a[rn1];
            const double e_n = lane_next(a[rn0]), d1_n = lane_next(a[rn1]);  // A[k1'][k0'], A[k1'][k1'] of the next pair
            if (i == k0 + 2) {
              publish_pinv(a[rn0], e_n, d1_n, (m + 1) & 1);
              Sw.colbuf[(m + 1) & 1][0][i] = a[rn0] - 1.0;
              a[rn0] -= 2.0;
            }
            if (i == k0 + 3) {
              asm volatile("" ::: "memory");  // (keeps the block a branch)
              Sw.colbuf[(m + 1) & 1][1][i] = a[rn1] - 1.0;
              a[rn1] -= 2.0;
            }
          }
          QMPC_STEP_TICK(3, 0.0);
          __builtin_amdgcn_sched_barrier(0);
          // the rest of the next pair's group: both pivot columns over the columns before the pair, then after it (the
          // same two updates, in the same order, for every element as fmac16 + fmac16)
          fmac_range_rowbcast<0, JN>(&a[GN0], cvn0, nu0);
          fmac_range_rowbcast<JN + 2, WN>(&a[GN0], cvn0, nu0);
          fmac_range_rowbcast<0, JN>(&a[GN0], cvn1, nu1);
          fmac_range_rowbcast<JN + 2, WN>(&a[GN0], cvn1, nu1);
          StaticFor<0, NG>::run([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g != GN) {
              constexpr bool TAIL = TAIL8 && g == NG - 1;
              const double cv0 = cvg0[g], cv1 = cvg1[g];
              if constexpr (TAIL) {
                double(&ag)[8] = *reinterpret_cast<double(*)[8]>(&a[16 * g]);
                fmac8_rowbcast(ag, cv0, nu0);
                fmac8_rowbcast(ag, cv1, nu1);
              } else {
                double(&ag)[16] = *reinterpret_cast<double(*)[16]>(&a[16 * g]);
                fmac16_rowbcast(ag, cv0, nu0);
                fmac16_rowbcast(ag, cv1, nu1);
              }
            }
          });
          QMPC_STEP_TICK(4, a[0] + a[CW - 1] + a[CW / 2]);
          __syncthreads();
          QMPC_STEP_TICK(5, 0.0);
        }
      });
    }
  }
  if (notpd) S.status = QMPC_DEV_ST_NOT_PD;  // benign race: same value from every thread
  QMPC_TICK(4);
  QMPC_STOP(3, a);

  auto& Sb = S.u.b;
  // ------------------------------------------------------------ stage 4
  // unconstrained minimiser x = -H^-1 g = a * g   (distributed mat-vec), then
  // the inverse leaves the registers: packed lower triangle in LDS.
  {
    double acc = 0.0;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) acc = __builtin_fma(a[jj], Sw.g[c * CW + jj], acc);
    Sw.part[c][i] = acc;
  }
  __syncthreads();
  if constexpr (PHA) {
    // ---- producer: H^-1 (= -a) row i, columns of this thread's group -> the work item, full rows (the padding is the
    // identity), 16-byte stores; x_u; the stance list.  Row-major and symmetric: the engine reads COLUMN j as row j,
    // two or three coalesced loads.
    const int item = S.evslot;
    constexpr int LD = NP;
    // Only the LOWER BLOCK TRIANGLE (64-row blocks; the diagonal blocks in full) of the first n rows and columns is
    // written: the engine takes the part of a column that lies above the diagonal block from the row (symmetry) and
    // the part below it with strided loads.  All CUs dump at the same time; what does not fit the L2 (4 MB per XCD, 32
    // CUs) drains at the XCD's write bandwidth while every wave of the chip waits in its stores (46 k of 248 k cycles
    // per robot with full 192 x 192 matrices, 9.2 MB per XCD and round) -- 152 KB instead of 288 KB per robot at n = 168,
    // 87 instead of 128 KB at n = 120.  (A wave is one 64-row block and one column group: the bound is wave-uniform.)
    GlobalF64* const Hrow = (GlobalF64*)P.wk_hinv + (size_t)item * (LD * LD) + (size_t)i * LD + c * CW;
    {
      const int nev = (n + 1) & ~1;  // (16-byte stores)
      int cend = 64 * (i / 64 + 1);
      cend = cend < nev ? cend : nev;
      const int cnt = (i < n) ? cend - c * CW : 0;  // columns of this thread's group to write
#pragma unroll
      for (int jj = 0; jj < CW; jj += 2)
        if (jj < cnt) st2(Hrow + jj, -a[jj], -a[jj + 1]);
    }
    if (tid < NP)
      P.wk_xu[(size_t)item * LD + tid] = (tid < n) ? Sw.part[0][tid] + Sw.part[1][tid] + Sw.part[2][tid] + Sw.part[3][tid] : 0.0;
    QmpcWorkHdr* const hd = P.wk_hdr + item;
    if (tid < 64) {
      hd->sidx[tid] = S.sidx[tid];
      hd->fmaxk[tid] = (tid < nst) ? (float)S.fmaxk[tid] : 0.f;
    }
    if (tid == 0) {
      hd->rid = rid;
      hd->n = n;
      hd->nst = nst;
      hd->status0 = S.status;
    }
    if (tid < WAVE) {
      // how hard this robot is going to be: rows violated at x_u (nearly all of them end up in the working set, and
      // every one costs an iteration).  The engine workgroups take the hardest robots first -- a launch ends with its
      // slowest robot, and that one must not be the one that started last
      int viol = 0;
      if (tid < nst) {
        double xs[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const int j = 3 * tid + ax;
          xs[ax] = Sw.part[0][j] + Sw.part[1][j] + Sw.part[2][j] + Sw.part[3][j];
        }
        const double fx = P.mu_inv * xs[0], fy = P.mu_inv * xs[1], nt = -P.tol, ifr = P.inv_fr_norm;
        viol = ((fx + xs[2]) * ifr < nt) + ((xs[2] - fx) * ifr < nt) + ((fy + xs[2]) * ifr < nt) + ((xs[2] - fy) * ifr < nt) +
               (S.fmaxk[tid] - xs[2] < nt);
      }
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) viol += __shfl_xor(viol, sft);
      if (tid == 0) {
        const int lvl = viol >> (RB == 2 ? 2 : 3);
        const int b = QMPC_ORDER_BUCKETS - 1 - (lvl < QMPC_ORDER_BUCKETS - 1 ? lvl : QMPC_ORDER_BUCKETS - 1);
        const int pos = atomicAdd(P.wk_bucket + b, 1);
        P.wk_order[(size_t)b * P.wk_cap + P.wk_base + pos] = item;
      }
    }
    QMPC_TICK(5);
    __syncthreads();  // (the next robot of a list-consuming workgroup reuses the LDS)
    return false;
  }
  const bool engine = tid < WAVE;
  double xv[RE];  // engine lane: x[lane + 64 q]
#pragma unroll
  for (int q = 0; q < RE; ++q) xv[q] = 0.0;
  if (engine) {
#pragma unroll
    for (int q = 0; q < RE; ++q) {
      const int j = lane + 64 * q;
      if (j < n) xv[q] = Sw.part[0][j] + Sw.part[1][j] + Sw.part[2][j] + Sw.part[3][j];
    }
  }
  double gq[RE];  // ADMM engine lane: gradient q[lane + 64 q]
#pragma unroll
  for (int q = 0; q < RE; ++q) gq[q] = 0.0;
  if constexpr (ADMM) {
    if (engine) {
#pragma unroll
      for (int q = 0; q < RE; ++q)
        if (lane + 64 * q < n) gq[q] = Sw.g[lane + 64 * q];
    }
  }
  const double fmx = (engine && lane < nst) ? S.fmaxk[lane] : 0.0;  // f_max of stance slot `lane`
  __syncthreads();  // sweep storage (and the assembly storage under it) is dead: Slv may overwrite it
  if (i < n) {
    // opaque copies of the indices: keeps the compiler from carrying values that
    // it pre-computed during assembly across the whole sweep (they were spilled)
    int io = i, co = c * CW;
    asm volatile("" : "+v"(io), "+v"(co));
    // (one compare against a literal per element -- d >= jj -- and one base address: the dump is ~4 % of the kernel's vector
    //  instructions; the registers are negated in place, nothing reads them afterwards)
    const int dd = io - co;
    double* const hrow = Sb.Hp + (io * (io + 1) / 2 + co);
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) {
      if (dd >= jj) hrow[jj] = -a[jj];
      if ((jj & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (V5) {
    if constexpr (!C::GLOBAL_EVENTS) {
      static_assert(C::NPOOL % 2 == 0 && C::NH % 2 == 0, "16-byte stores into the event pool");
      for (int k = 2 * tid; k < C::NPOOL; k += 2 * NT) st2(&Sb.Sinv[k], 0.0, 0.0);  // event rows start out zero
    }
  } else {
    // packed S_W^-1 behind the n(n+1)/2 doubles of the inverse (see stage 5)
    const int nh0 = n * (n + 1) / 2, cap0 = C::NH + C::NPOOL - nh0;
    for (int k = tid; k < (cap0 < C::NS ? cap0 : C::NS); k += NT) Sb.Hp[nh0 + k] = 0.0;
  }
  // (diag(H^-1), which the event engine scales its dependence test with, is read from the packed inverse where it is needed --
  //  entry (j, j) at j (j + 3) / 2, a scalar address -- instead of being collected here through a sixteen-way register select in
  //  every wave: 190 vector instructions per robot; Sb.D stays as the ADMM engine's / the output stage's scratch)
  __syncthreads();
  QMPC_TICK(5);
  QMPC_STOP(4, xv);

  // ------------------------------------------------------------ stage 5 (event form)
  // Goldfarb-Idnani with the projected inverse kept as a SUM OF EVENTS.  Every
  // change of the working set W is a rank-1 event (z~, g~):
  //   add  p -> slot q : z~ = P c_p / sqrt(delta),  g~_w = -r_w / sqrt(delta), g~_q = 1 / sqrt(delta)
  //   drop slot l      : z~ = N*_l / sqrt(gamma),    g~_w = -S^-1[w][l] / sqrt(gamma)   (gamma = S^-1[l][l])
  // and the three matrices of the method are
  //   P    = H^-1 - sum_add z~ z~^T + sum_drop z~ z~^T     (projected inverse, n x n)
  //   N*   = sum_all z~ g~^T                              (H^-1 C_W S^-1, n x slots)
  //   S^-1 = sum_add g~ g~^T - sum_drop g~ g~^T           ((C_W^T H^-1 C_W)^-1)
  // (column l of every g~ is zeroed when slot l is dropped).  Nothing is ever
  // updated in place: a step costs two columns of the packed H^-1 plus, per event,
  // two broadcast loads and two FMAs per lane -- no block barrier, no rank-1
  // sweep over n^2 entries, and the other waves of the block are not needed.
  // Events live in the LDS pool: add events from the front, drop events from the
  // back (so neither loop needs a sign), rows in between stay zero.
  int iters = 0;
  bool retry = false;
  if constexpr (ADMM) {
    // ---------------------------------------------------------- stage 5 (JCQP alternate)
    // QpProblem<double>::runFromDense (src/JCQP/QpProblem.cpp:178-269): OSQP-style ADMM from a cold start
    // with the constant KKT matrix [[P + sigma I, A^T], [A, -R^-1]].  Eliminating the constraint block,
    //   (P + sigma I + A^T R A) x~ = sigma x - q + A^T (R z - y),    z~ = A x~,
    // and A^T R A is diagonal here, so the matrix swept above IS that system's and one iteration is one
    // mat-vec with its explicit inverse (packed in LDS) plus element-wise updates of the five rows of
    // every foot-step: x <- alpha x~ + (1 - alpha) x, z <- clamp(alpha z~ + (1 - alpha) z + y / rho),
    // y <- y + rho (alpha z~ + (1 - alpha) z_prev - z); every tenth iteration the residual
    // (|A x - z_prev|_inf + |P x + q + A^T y|_inf) / 4 is compared with `terminate` (:238-247, :381-407).
    // P x is carried along as M x - diag(M - P) x with M x = alpha rhs + (1 - alpha) M x_prev (exact
    // recurrence: M x~ = rhs), A x likewise.  Run by wave 0 alone: lane = variable, lane = foot-step.
    if (engine) {
      const double mi = P.mu_inv, sig = P.admm_sigma, al = P.admm_alpha, rho = P.admm_rho, rinf = 1e-6;
      const double big = (double)5e10f;  // BIG_NUMBER through float (SolverMPC.cpp:15, :356)
      const int max_it = P.admm_max_iter;
      double r4 = rho;  // rho of the fz <= f_max row of this lane's foot-step (computeConstraintInfos :276-291)
      if (__builtin_fabs(fmx) < 1e-10) r4 = rho * 1e3;
      else if (fmx > 1e10) r4 = rinf;
      const double ir4 = 1.0 / r4, irinf = 1.0 / rinf;
      double dj[RE], mxv[RE];  // diag(M - P) and M x of this lane's variables
#pragma unroll
      for (int q = 0; q < RE; ++q) {
        const int j = lane + 64 * q;
        const double f3 = (j < n) ? S.fmaxk[(j / 3) & 63] : 1.0;
        const double rr = (__builtin_fabs(f3) < 1e-10) ? rho * 1e3 : (f3 > 1e10 ? rinf : rho);
        dj[q] = sig + ((j % 3 < 2) ? 2.0 * rinf * mi * mi : 4.0 * rinf + rr);
        mxv[q] = 0.0;
        xv[q] = 0.0;  // cold start (the unconstrained minimiser computed above is not used)
      }
      double z[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0}, ax[5] = {0, 0, 0, 0, 0};
      auto gather = [&](const double (&v)[RE], int j) __attribute__((always_inline)) {
        double out = 0.0;
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          const double cand = __shfl(v[q], j & 63);
          if ((j >> 6) == q) out = cand;
        }
        return out;
      };
      // slot quantities (cx, cy, cz) -> this lane's variables: variable j takes component j % 3 of slot j / 3
      auto scatter3 = [&](double cx, double cy, double cz, double (&out)[RE]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          const int j = lane + 64 * q, sl = (j / 3) & 63, ax3 = j % 3;
          const double a0 = __shfl(cx, sl), a1 = __shfl(cy, sl), a2 = __shfl(cz, sl);
          out[q] = (j < n) ? (ax3 == 0 ? a0 : (ax3 == 1 ? a1 : a2)) : 0.0;
        }
      };
      auto Hel = [&](int row, int j) __attribute__((always_inline)) {
        return Sb.Hp[(j <= row) ? row * (row + 1) / 2 + j : j * (j + 1) / 2 + row];
      };
      double resid = __builtin_inf();
      double* const rv = Sb.D;  // rhs, broadcast through LDS
      __builtin_amdgcn_s_setprio(QMPC_ENGINE_PRIO);
      for (int it = 1; it <= max_it; ++it) {
        // rhs = sigma x - q + A^T (R z - y)                                       (solveLinearSystem :315-323)
        const double w0 = rinf * z[0] - y[0], w1 = rinf * z[1] - y[1], w2 = rinf * z[2] - y[2],
                     w3 = rinf * z[3] - y[3], w4 = r4 * z[4] - y[4];
        double cw[RE], rhs[RE];
        scatter3(mi * (w0 - w1), mi * (w2 - w3), (w0 + w1) + (w2 + w3) + w4, cw);
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          rhs[q] = sig * xv[q] - gq[q] + cw[q];
          if (lane + 64 * q < NP) rv[lane + 64 * q] = (lane + 64 * q < n) ? rhs[q] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        // x~ = M^-1 rhs
        double xt[RE];
#pragma unroll
        for (int q = 0; q < RE; ++q) xt[q] = 0.0;
        for (int j0 = 0; j0 < n; j0 += 4) {
          double rj[4], hv[4][RE];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = (j0 + u < n) ? j0 + u : 0;
            rj[u] = (j0 + u < n) ? rv[j] : 0.0;
#pragma unroll
            for (int q = 0; q < RE; ++q) hv[u][q] = (lane + 64 * q < n) ? Hel(lane + 64 * q, j) : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < RE; ++q) xt[q] = __builtin_fma(hv[u][q], rj[u], xt[q]);
        }
        __builtin_amdgcn_wave_barrier();
        // z~ = A x~ on the foot-step lanes (f_block rows, SolverMPC.cpp:366-370)
        const int j3 = 3 * (lane < nst ? lane : 0);
        const double t0 = gather(xt, j3), t1 = gather(xt, j3 + 1), t2 = gather(xt, j3 + 2);
        const double zt[5] = {mi * t0 + t2, -mi * t0 + t2, mi * t1 + t2, -mi * t1 + t2, t2};
        // x, M x (stepX :340-347)
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          xv[q] = al * xt[q] + (1.0 - al) * xv[q];
          mxv[q] = al * rhs[q] + (1.0 - al) * mxv[q];
        }
        // z, y, A x (stepZ :349-358, stepY :360-367)
        double pmax = 0.0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const double rr = (r < 4) ? rinf : r4, ir = (r < 4) ? irinf : ir4, ub = (r < 4) ? big : fmx;
          const double zr = al * zt[r] + (1.0 - al) * z[r];
          double zn = zr + ir * y[r];
          zn = zn < 0.0 ? 0.0 : zn;
          zn = zn > ub ? ub : zn;
          y[r] = y[r] + rr * (zr - zn);
          ax[r] = al * zt[r] + (1.0 - al) * ax[r];  // A x of the relaxed iterate
          const double pr = __builtin_fabs(ax[r] - z[r]);  // ... against the PREVIOUS z (:388)
          pmax = pr > pmax ? pr : pmax;
          z[r] = zn;
        }
        iters = it;
        if (it % 10 == 0) {  // residual check (:238-247)
          double ey[RE];
          scatter3(mi * (y[0] - y[1]), mi * (y[2] - y[3]), (y[0] + y[1]) + (y[2] + y[3]) + y[4], ey);
          double dmax = 0.0;
#pragma unroll
          for (int q = 0; q < RE; ++q) {
            const double dv = __builtin_fabs((mxv[q] - dj[q] * xv[q]) + gq[q] + ey[q]);
            if (lane + 64 * q < n) dmax = dv > dmax ? dv : dmax;
          }
          if (!(lane < nst)) pmax = 0.0;
          // max over the wave of non-negative doubles: bit pattern order == value order
          const double pm = wave_max_pos_f64(pmax), dm = wave_max_pos_f64(dmax);
          resid = (dm + pm) * 0.25;
          if (resid < P.admm_term || it >= max_it) break;
        }
      }
      __builtin_amdgcn_s_setprio(0);
      QMPC_TICK(6);
      // outputs: q_soln = the ADMM iterate (SolverMPC.cpp:598-602 / :613-617), not an exact minimiser
      if (lane < 12) P.grf[(size_t)rid * 12 + lane] = 0.f;
      __builtin_amdgcn_wave_barrier();
      bool nf = false;
#pragma unroll
      for (int q = 0; q < RE; ++q) {
        const int j = lane + 64 * q;
        if (j < n) {
          const int k = S.sidx[j / 3], ax3 = j % 3;
          if (k < 4) P.grf[(size_t)rid * 12 + 3 * k + ax3] = (float)xv[q];
          if (P.soln) P.soln[(size_t)rid * 12 * h + 3 * k + ax3] = xv[q];
          nf |= !(__builtin_fabs(xv[q]) < __builtin_inf());
        }
      }
      int status = (resid < P.admm_term) ? 0 : QMPC_DEV_ST_MAXITER;
      if (__ballot(nf)) status |= QMPC_DEV_ST_NONFINITE;
      if (lane == 0) {
        P.status[rid] = S.status | status;
        if (P.iters) P.iters[rid] = iters;
        S.mode = 0;
      }
    }
  } else if constexpr (V5) {
    constexpr int NPE = NP, KS = C::KS, EV = NPE + KS;
    constexpr int KQ = (KS + 63) / 64;  // working-set slots per lane: slot s lives in lane s % 64, entry s / 64
    // event capacity: LDS pool of this class / a slice of a global pool (class 3: its only pool; the other
    // classes: where a robot continues when its LDS pool is full)
    constexpr int NHELP = C::NHELP;
    // (LDS pool: the last NHELP records' worth of it holds the helper waves' partial sums)
    constexpr int KEV_L = ((C::NPOOL - (C::GLOBAL_EVENTS ? 0 : NHELP * EV)) / EV) & ~3, KEV_G = C::KEV_GLOBAL;
#ifndef QMPC_TR_G
#define QMPC_TR_G 4
#endif
    // events per trip on a global pool.  (8 -- twice the loads in flight per wait -- measured: no faster, the
    // accumulation pays ~22 cycles per load instruction whatever the trip length, and it costs 20 VGPRs)
    constexpr int TR_G = QMPC_TR_G;
    // this lane's entries of an index-major vector stored NP long: lanes past row NP (class 4:
    // 96 rows in two 64-lane blocks) read entry 0 -- harmless, those rows are never used -- and
    // do not write
    int zo[RE];
    bool zw[RE];
#pragma unroll
    for (int q = 0; q < RE; ++q) {
      zw[q] = lane + 64 * q < NP;
      zo[q] = zw[q] ? lane + 64 * q : 0;
    }
    int gl_off[KQ];  // this lane's entries of an event's g~
#pragma unroll
    for (int k = 0; k < KQ; ++k) gl_off[k] = NPE + ((lane + 64 * k) & (KS - 1));
    // ---- layout of an event record (z~[NP], g~[KS]).  Plain: z~ then g~, one entry per lane and 64-row block.
    // PAIRED: what ONE LANE reads of a record is stored adjacently, so that it takes 16-byte loads -- the
    // accumulation over the events is bound by the number of load instructions the CU's address unit takes
    // (~22 cycles each, whichever wave issues them), not by bytes:
    //   128-row class:  [lane](z[l], z[l+64])  |  [lane] g[l]                             2 loads instead of 3
    //   192-row class:  [lane](z[l], z[l+64])  |  [lane](z[l+128], g[l])  |  [lane] g[l+64]      3 instead of 5
    constexpr bool PAIRED = C::PAIRED;
    static_assert(!PAIRED || (C::NH % 2 == 0 && NP % 64 == 0 && C::KS % 64 == 0), "paired records: 16-byte aligned pool, whole 64-lane blocks");
    auto off_z = [](int j) __attribute__((always_inline)) {  // row j of z~
      if constexpr (!PAIRED) return j;
      else if constexpr (RE == 2) return 2 * (j & 63) + (j >> 6);
      else return (j < 128) ? 2 * (j & 63) + (j >> 6) : 128 + 2 * (j & 63);
    };
    auto off_g = [](int sl) __attribute__((always_inline)) {  // slot sl of g~
      if constexpr (!PAIRED) return NPE + sl;
      else if constexpr (RE == 2) return 128 + sl;
      else return (sl < 64) ? 128 + 2 * sl + 1 : 256 + (sl - 64);
    };
    // this lane's entries of a record: read ...
    auto rec_load = [&](const auto eu, double (&zv)[RE], double (&gv)[KQ]) __attribute__((always_inline)) {
      if constexpr (!PAIRED) {
#pragma unroll
        for (int q = 0; q < RE; ++q) zv[q] = eu[zo[q]];
#pragma unroll
        for (int k = 0; k < KQ; ++k) gv[k] = eu[gl_off[k]];
      } else {
        const F64x2 a = ld2(eu + 2 * lane);
        zv[0] = a.x;
        zv[1] = a.y;
        if constexpr (RE == 2) {
          gv[0] = eu[128 + lane];
        } else {
          const F64x2 b2 = ld2(eu + 128 + 2 * lane);
          zv[2] = b2.x;
          gv[0] = b2.y;
          gv[1] = eu[256 + lane];
        }
      }
    };
    // ... and written (rows past NP and slots past KS do not exist in the plain layout; PAIRED: NP, KS multiples of 64)
    auto rec_store = [&](const auto en, const double (&zv)[RE], const double (&gv)[KQ]) __attribute__((always_inline)) {
      if constexpr (!PAIRED) {
#pragma unroll
        for (int q = 0; q < RE; ++q)
          if (zw[q]) en[zo[q]] = zv[q];
#pragma unroll
        for (int k = 0; k < KQ; ++k)
          if (lane + 64 * k < KS) en[NPE + lane + 64 * k] = gv[k];
      } else {
        st2(en + 2 * lane, zv[0], zv[1]);
        if constexpr (RE == 2) {
          en[128 + lane] = gv[0];
        } else {
          st2(en + 128 + 2 * lane, zv[2], gv[0]);
          en[256 + lane] = gv[1];
        }
      }
    };
    // where the helper waves leave their partial sums: NHELP records of (z[NP], r[KS]) behind the LDS event pool (the
    // largest class keeps no events in LDS: the front of the pool)
    double* const hpart = Sb.Sinv + (C::GLOBAL_EVENTS ? 0 : KEV_L * EV);
    // z -= +-y z~ , r += y g~ with y = z~^T c_p over the stored events, four (TR) per trip; rows past the last event of
    // a trip are zero.  Wave w of nw takes the trips w, w + nw, ... of the add events (front of the pool, DIR = +1)
    // and then of the drop events (back of the pool, DIR = -1)
    auto ev_part = [&](auto gpc, const auto pool, int pj1, int pj2, double pa1, double pa2, int neva, int nevd, int w, int nw,
                       double (&z)[RE], double (&rw)[KQ]) __attribute__((always_inline)) {
      constexpr bool GPOOL = decltype(gpc)::value;
      constexpr int KEV = GPOOL ? KEV_G : KEV_L, TR = GPOOL ? TR_G : 4;
      const int oj1 = off_z(pj1), oj2 = off_z(pj2);
      auto part = [&](auto dirc, int base, int cnt, int w0) __attribute__((always_inline)) {
        constexpr int DIR = decltype(dirc)::value;
#pragma unroll 1
        for (int t0 = TR * w0; t0 < cnt; t0 += TR * nw) {
          const auto ev = pool + (base + DIR * t0) * EV;
          double ya[TR], yb[TR], zl[TR][RE], gl[TR][KQ];
#pragma unroll
          for (int u = 0; u < TR; ++u) {
            const auto eu = ev + DIR * u * EV;
            ya[u] = eu[oj1];
            yb[u] = eu[oj2];
            rec_load(eu, zl[u], gl[u]);
          }
#pragma unroll
          for (int u = 0; u < TR; ++u) {
            const double y = __builtin_fma(pa2, yb[u], pa1 * ya[u]);
#pragma unroll
            for (int q = 0; q < RE; ++q) z[q] = __builtin_fma(DIR > 0 ? -y : y, zl[u][q], z[q]);
#pragma unroll
            for (int k = 0; k < KQ; ++k) rw[k] = __builtin_fma(y, gl[u][k], rw[k]);
          }
        }
      };
      part(std::integral_constant<int, 1>{}, 0, neva, w);
      if (nevd > 0) {
        // (the drop trips continue the round-robin where the add trips stopped)
        const int ta = (neva + TR - 1) / TR;
        part(std::integral_constant<int, -1>{}, KEV - 1, nevd, nw == 1 ? 0 : (w + nw - ta % nw) % nw);
      }
    };
    if (engine) {
      const double mi = P.mu_inv, inv_fr = P.inv_fr_norm, tol = P.tol;
      const int max_iter = __builtin_amdgcn_readfirstlane(P.max_iter);
      // wave-uniform predicate -> scalar branch (the operands are uniform but live in
      // VGPRs; a ballot gives the compiler an SGPR condition, so the loop state
      // below stays in SGPRs instead of being carried through exec masks)
      auto uni = [](bool cnd) __attribute__((always_inline)) { return __builtin_amdgcn_ballot_w64(cnd) != 0ull; };
      unsigned amask = 0;  // stance-slot lane: bit ty = constraint (slot, ty) is in the working set
      int wcid[KQ];        // working-slot lane: constraint id in slot lane + 64 k, -1 = free
      double lam[KQ];      // ... and its multiplier
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        wcid[k] = -1;
        lam[k] = 0.0;
      }
      int khw = 0, status = 0, neva = 0, nevd = 0;
      bool need_p0 = true;  // (carried between the two runs; each run works on its own copy)
      unsigned long long rbm[KQ];  // compaction in progress: working-set slots whose add event is still to be rebuilt
#pragma unroll
      for (int k = 0; k < KQ; ++k) rbm[k] = 0ull;
      auto rb_any = [&]() __attribute__((always_inline)) {
        unsigned long long m = rbm[0];
#pragma unroll
        for (int k = 1; k < KQ; ++k) m |= rbm[k];
        return m != 0ull;
      };
      bool spill = false;             // the LDS pool is full: continue on a slice of the overflow pool
      // ---- warm start (qmpc_set_warm_start): the previous cycle's working set, slid by `ws_shift`
      // horizon steps and mapped onto this cycle's stance slots, one candidate per lane.  The
      // candidates are ADDED FIRST, without search or ratio test (a full step onto each, whatever the
      // sign of the step): that lands on the minimiser of the equality-constrained problem on that
      // set.  Candidates whose multiplier comes out negative are then dropped one by one; what
      // remains is a genuine Goldfarb-Idnani state (x optimal on W, multipliers >= 0) and the normal
      // iteration takes over.  The answer is the same unique minimiser; only the path is shorter.
      int cand = -1;
      // (selective: a robot the previous call found easy starts cold -- a cold dual active set needs ~|W*| iterations
      //  anyway, and a wrong guess costs two events; wave-uniform scalar load)
      bool ws_take = WARM && P.ws != nullptr;
      if (WARM && ws_take && P.ws_min_iters > 0) ws_take = P.hint_iters != nullptr && P.hint_iters[rid] >= P.ws_min_iters;
      if (WARM && ws_take && lane < (KS < QMPC_WS_STRIDE ? KS : QMPC_WS_STRIDE)) {
        const int eg = P.ws[(size_t)rid * QMPC_WS_STRIDE + lane];  // global id 5 * (4 step + foot) + type
        const int kg = (eg >= 0 ? eg / 5 : 0) - 4 * P.ws_shift;
        if (eg >= 0 && kg >= 0 && kg < nfs) {
          const int sl = S.kslot[kg];
          if (sl != 0xff) cand = 5 * sl + (eg - 5 * (eg / 5));
        }
      }
      // (WARM is a separate instantiation: the cold kernel carries none of this)
      unsigned long long cmask = WARM ? __ballot(cand >= 0) : 0ull;
      bool forced = false, fixneg = WARM && (cmask != 0ull);
      int p_e = 0, psl = 0, pty = 0, pj1 = 0, pj2 = 0;
      double pa1 = 0.0, pa2 = 0.0, p_rhs = 0.0, lp = 0.0;
      int rbl[RE];
#pragma unroll
      for (int q = 0; q < RE; ++q) rbl[q] = (lane + 64 * q) * (lane + 64 * q + 1) / 2;
      auto gather = [&](const double (&v)[RE], int j) __attribute__((always_inline)) {
        double out = 0.0;
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          const double cand = __shfl(v[q], j & 63);
          if ((j >> 6) == q) out = cand;
        }
        return out;
      };
      auto bcast = [&](const double (&v)[RE], int j) __attribute__((always_inline)) {
        return lane_elem<RE>(v, j);
      };
      // element (row = lane + 64 q, column j) of H^-1 for a wave-uniform j
      auto Hcol = [&](int q, int j) __attribute__((always_inline)) {
        const int row = lane + 64 * q;
        const int tj = j * (j + 1) / 2;
        return Sb.Hp[(j <= row) ? rbl[q] + j : tj + row];
      };
      // 1/sqrt(d), d > 0, to full double precision: v_rsq_f64 seed + two Newton steps
      auto rsqrt_full = [&](double d) __attribute__((always_inline)) {
        double y = __builtin_amdgcn_rsq(d);
        double e = __builtin_fma(-d * y, y, 1.0);
        y = __builtin_fma(0.5 * y, e, y);
        e = __builtin_fma(-d * y, y, 1.0);
        y = __builtin_fma(0.5 * y, e, y);
        return y;
      };
      // ---- the iteration, on the LDS pool (gpc = false) or on a slice of a global pool (true).  Everything it
      // carries from one trip to the next lives outside, so that a robot can leave the first and continue in
      // the second
      auto run = [&](auto gpc, GlobalF64* const gpool) __attribute__((always_inline)) {
        constexpr bool GPOOL = decltype(gpc)::value;
        constexpr int KEV = GPOOL ? KEV_G : KEV_L;
        constexpr int TR = GPOOL ? TR_G : 4;  // events per trip of the accumulation loops
        // (locals, not captures: a flag that nested lambdas reach through two closures ends up in scratch)
        bool need_p = need_p0, retry = false;
        const auto pool = [&]() __attribute__((always_inline)) {
          if constexpr (GPOOL) return gpool; else return (double*)Sb.Sinv;
        }();
        // rows past the last event of a group of four must read zero (the accumulation loops take four events
        // per trip).  The LDS pool is zeroed wholesale beforehand; the global pool lazily, one group ahead
        auto zero_group = [&](int first) __attribute__((always_inline)) {
          if constexpr (GPOOL) {
            if (first >= 0 && first + TR <= KEV)
              for (int idx = lane; idx < TR * EV; idx += 64) pool[(size_t)first * EV + idx] = 0.0;
          }
        };
        // (global pool: called by the READERS of the pool, right before they start -- the stores of the previous
        //  step complete while the next constraint is being selected, not while the wave waits for them)
        auto pool_sync = [&]() __attribute__((always_inline)) {
          if constexpr (GPOOL) {  // event rows written by some lanes are read by others of this wave through L1 / L2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          }
        };
        // Remove working-set slot l: one drop event.  u = N*_l (index-major lanes), sc = S^-1[:, l]
        // (slot lanes), gamma = S^-1[l][l].  With `repair` (warm start) the iterate also moves to the
        // minimiser of the problem WITHOUT that constraint: x -= (lam_l / gamma) u, lam -= (lam_l / gamma) sc.
        // Returns false when the projected inverse has lost definiteness numerically (retry is set).
        auto drop_slot = [&](int l, bool repair) __attribute__((always_inline)) {
          double u[RE], sc[KQ];
#pragma unroll
          for (int k = 0; k < KQ; ++k) sc[k] = 0.0;
          pool_sync();
#pragma unroll
          for (int q = 0; q < RE; ++q) u[q] = 0.0;
          const int ogl = off_g(l);
          auto dacc = [&](auto dirc, int base, int cnt) __attribute__((always_inline)) {
            constexpr int DIR = decltype(dirc)::value;
#pragma unroll 1
            for (int t0 = 0; t0 < cnt; t0 += TR) {
              const auto ev = pool + (base + DIR * t0) * EV;
              double gll[TR], zl[TR][RE], gw[TR][KQ];
#pragma unroll
              for (int u4 = 0; u4 < TR; ++u4) {
                const auto eu = ev + DIR * u4 * EV;
                gll[u4] = eu[ogl];
                rec_load(eu, zl[u4], gw[u4]);
              }
#pragma unroll
              for (int u4 = 0; u4 < TR; ++u4) {
#pragma unroll
                for (int q = 0; q < RE; ++q) u[q] = __builtin_fma(gll[u4], zl[u4][q], u[q]);
#pragma unroll
                for (int k = 0; k < KQ; ++k) sc[k] = __builtin_fma(DIR > 0 ? gll[u4] : -gll[u4], gw[u4][k], sc[k]);
              }
            }
          };
          dacc(std::integral_constant<int, 1>{}, 0, neva);
          if (nevd > 0) dacc(std::integral_constant<int, -1>{}, KEV - 1, nevd);
          const double gamma = lane_elem<KQ>(sc, l);
          if (uni(!(gamma > 0.0))) {
            retry = true;  // numerically lost S^-1[l][l] > 0: start over with the other engine
            return false;
          }
          if (repair) {
            const double coef = lane_elem<KQ>(lam, l) * fast_rcp(gamma);
#pragma unroll
            for (int q = 0; q < RE; ++q) xv[q] = __builtin_fma(-coef, u[q], xv[q]);
#pragma unroll
            for (int k = 0; k < KQ; ++k) lam[k] = __builtin_fma(-coef, sc[k], lam[k]);
          }
          const double sg = rsqrt_full(gamma);
          const int de = lane_elem<KQ>(wcid, l);
          const auto en = pool + (KEV - 1 - nevd) * EV;
          {
            double zv[RE], gv[KQ];
#pragma unroll
            for (int q = 0; q < RE; ++q) zv[q] = u[q] * sg;
#pragma unroll
            for (int k = 0; k < KQ; ++k) gv[k] = (lane + 64 * k == l || wcid[k] < 0) ? 0.0 : -sc[k] * sg;
            rec_store(en, zv, gv);
          }
          // slot l leaves: column l of every earlier g~ is cleared (N*_l = 0, S^-1[l][:] = 0)
          for (int e = lane; e < neva; e += 64) pool[e * EV + ogl] = 0.0;
          for (int e = lane; e < nevd; e += 64) pool[(KEV - 1 - e) * EV + ogl] = 0.0;
#pragma unroll
          for (int k = 0; k < KQ; ++k)
            if (lane + 64 * k == l) {
              wcid[k] = -1;
              lam[k] = 0.0;
            }
          if (lane == de / 5) amask &= ~(1u << (de % 5));
          nevd += 1;
          if (GPOOL && (nevd & (TR - 1)) == 0 && ((neva + TR - 1) & ~(TR - 1)) + nevd + TR <= KEV) zero_group(KEV - nevd - TR);
          __builtin_amdgcn_wave_barrier();
          return true;
        };

        while (true) {
          // loop-carried counters are wave-uniform: keep them in SGPRs
          iters = __builtin_amdgcn_readfirstlane(iters);
          khw = __builtin_amdgcn_readfirstlane(khw);
          neva = __builtin_amdgcn_readfirstlane(neva);
          nevd = __builtin_amdgcn_readfirstlane(nevd);
          status = __builtin_amdgcn_readfirstlane(status);
          p_e = __builtin_amdgcn_readfirstlane(p_e);
          psl = __builtin_amdgcn_readfirstlane(psl);
          pty = __builtin_amdgcn_readfirstlane(pty);
          pj1 = __builtin_amdgcn_readfirstlane(pj1);
          pj2 = __builtin_amdgcn_readfirstlane(pj2);
          if (QMPC_DBG_ITER > 0 && dbg_clk && lane == 0 && iters == QMPC_DBG_ITER) dbg_clk[14] = clock64();
          // ---- room for one more event of either kind?  Checked here, between iterations, where the state is
          // consistent.  LDS pool: no -> leave the loop, the events move to this robot's slice of the overflow pool
          // in global memory and the iteration continues there (`spill`).  Global pool: no -> compaction.  Every
          // working-set change costs an event; a constraint that entered and left again holds two records that
          // cancel.  With drop events in the pool: forget all records and rebuild the projected inverse from H^-1
          // with one add event per constraint that is in the working set NOW (x, the multipliers and the pending
          // constraint are untouched: the operators are the same, only their representation is shorter).
          if (!rb_any() && ((neva + TR) & ~(TR - 1)) + ((nevd + TR) & ~(TR - 1)) > KEV) {
            if constexpr (!GPOOL) {
              spill = true;
              break;
            } else {
              if (nevd == 0) {
                retry = true;  // the working set alone fills the pool: the robot is re-run with the Schur-form engine
                break;
              }
#pragma unroll
              for (int k = 0; k < KQ; ++k) rbm[k] = __ballot(lane + 64 * k < KS && wcid[k] >= 0);
              status |= QMPC_DEV_ST_COMPACTED;  // informational
              neva = 0;
              nevd = 0;
              zero_group(0);
              zero_group(KEV - TR);
              __builtin_amdgcn_wave_barrier();
            }
          }
          const bool rebuild = GPOOL && rb_any();
          int rl = 0;
          if (rebuild) {
            bool took = false;
#pragma unroll
            for (int k = 0; k < KQ; ++k)
              if (!took && rbm[k] != 0ull) {
                rl = 64 * k + __ffsll((long long)rbm[k]) - 1;
                rbm[k] &= rbm[k] - 1ull;
                took = true;
              }
            con_coefs(lane_elem<KQ>(wcid, rl), mi, pj1, pj2, pa1, pa2);
          } else {
          if (WARM && uni(need_p) && cmask != 0ull) {
            // ---- warm start: next candidate of the previous working set, forced
            const int cl = __ffsll((long long)cmask) - 1;
            cmask &= cmask - 1ull;
            p_e = __builtin_amdgcn_readlane(cand, cl);
            psl = p_e / 5;
            pty = p_e - 5 * psl;
            con_coefs(p_e, mi, pj1, pj2, pa1, pa2);
            p_rhs = (pty == 4) ? -readlane_f64(fmx, psl) : 0.0;
            lp = 0.0;
            need_p = false;
            forced = true;
          } else if (WARM && uni(need_p) && fixneg) {
            // ---- warm start, second phase: a candidate whose multiplier is negative does not belong to
            // the working set -- remove it (one drop event) and move to the minimiser without it
            int l = -1;
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
              const unsigned long long nm = __ballot(wcid[k] >= 0 && lam[k] < 0.0);
              if (l < 0 && nm != 0ull) l = 64 * k + __ffsll((long long)nm) - 1;
            }
            if (l < 0) {
              fixneg = false;
              continue;
            }
            if (iters >= max_iter) {
              retry = true;  // the guess cannot be repaired within the iteration limit: start over, cold, with the other engine
              break;
            }
            if (!drop_slot(l, true)) break;
            iters += 1;
            continue;
          }
          if (uni(need_p)) {
            // ---- most violated constraint outside the working set (normalised), or done
            unsigned key = 0;
            const int j0 = 3 * (lane < nst ? lane : 0);
            const double x0 = gather(xv, j0), x1 = gather(xv, j0 + 1), x2 = gather(xv, j0 + 2);
            if (lane < nst) {
              const double fx = mi * x0, fy = mi * x1;
              double vmin = 0.0;
              int tmin = -1;
              const double sv[5] = {(fx + x2) * inv_fr, (x2 - fx) * inv_fr, (fy + x2) * inv_fr, (x2 - fy) * inv_fr, fmx - x2};
#pragma unroll
              for (int ty = 0; ty < 5; ++ty) {
                const bool cand = !((amask >> ty) & 1u) && sv[ty] < vmin;
                vmin = cand ? sv[ty] : vmin;
                tmin = cand ? ty : tmin;
              }
              if (vmin < -tol) key = (__float_as_uint((float)(-vmin)) & ~0x1FFu) | (unsigned)(5 * lane + tmin);
            }
            const unsigned best = wave_max_u32(key);
            if (best == 0u) break;
            if (iters >= max_iter) {
              status |= QMPC_DEV_ST_MAXITER;
              break;
            }
            p_e = (int)(best & 0x1FFu);
            psl = p_e / 5;
            pty = p_e - 5 * psl;
            con_coefs(p_e, mi, pj1, pj2, pa1, pa2);
            p_rhs = (pty == 4) ? -readlane_f64(fmx, psl) : 0.0;
            lp = 0.0;
            need_p = false;
            forced = false;
          }
          }  // (!rebuild)
          if (dbg_clk && lane == 0 && iters == QMPC_DBG_ITER) dbg_clk[8] = clock64();
          // ---- z = P c_p (index-major lanes), r = N*^T c_p (slot lanes)
          double z[RE];
#pragma unroll
          for (int q = 0; q < RE; ++q) {
            const int row = lane + 64 * q;
            z[q] = (row < n) ? __builtin_fma(pa2, Hcol(q, pj2), pa1 * Hcol(q, pj1)) : 0.0;
          }
          double rw[KQ];
#pragma unroll
          for (int k = 0; k < KQ; ++k) rw[k] = 0.0;
          pool_sync();
          if (QMPC_DBG_ITER > 0 && dbg_clk && lane == 0 && iters == QMPC_DBG_ITER) {
            double zs = 0.0;  // (the stamp waits for z)
#pragma unroll
            for (int q = 0; q < RE; ++q) zs += z[q];
            asm volatile("" ::"v"(zs));
            dbg_clk[15] = clock64();
          }
          // (measured, not kept: computing every event's y first -- one lane per event -- and streaming the rows
          //  afterwards saves two loads and two multiply-adds per event; +5 % for the 128-row class, -1 % for the
          //  64- and 96-row classes whose robots hold few events)
          bool helped = false;
          if constexpr (NHELP > 0) {
            // enough events for two barriers to pay: waves 1..NHELP take their share (fixed split, fixed order of the
            // final sum: the result does not depend on timing)
            if ((neva + TR - 1) / TR + (nevd + TR - 1) / TR >= C::HELP_MIN_TRIPS) {
              helped = true;
              if (lane == 0) {
                S.hd.cmd = 1;
                S.hd.pj1 = pj1;
                S.hd.pj2 = pj2;
                S.hd.neva = neva;
                S.hd.nevd = nevd;
                S.hd.glob = GPOOL ? 1 : 0;
                S.hd.pa1 = pa1;
                S.hd.pa2 = pa2;
                if constexpr (GPOOL) S.hd.gptr = (unsigned long long)(size_t)pool;
              }
              __syncthreads();  // (A) the request is up (and, global pool: this wave's event stores are drained)
              ev_part(gpc, pool, pj1, pj2, pa1, pa2, neva, nevd, 0, NHELP + 1, z, rw);
              __syncthreads();  // (B) the partial sums are in LDS
#pragma unroll
              for (int w = 0; w < NHELP; ++w) {
#pragma unroll
                for (int q = 0; q < RE; ++q) z[q] += hpart[w * EV + zo[q]];
#pragma unroll
                for (int k = 0; k < KQ; ++k) rw[k] += hpart[w * EV + gl_off[k]];
              }
            }
          }
          if (!helped) ev_part(gpc, pool, pj1, pj2, pa1, pa2, neva, nevd, 0, 1, z, rw);
          const double delta = __builtin_fma(pa2, bcast(z, pj2), pa1 * bcast(z, pj1));
          const double cn = __builtin_fma(pa2 * pa2, Sb.Hp[pj2 * (pj2 + 3) / 2], pa1 * pa1 * Sb.Hp[pj1 * (pj1 + 3) / 2]);  // scale of c_p^T H^-1 c_p: diag(H^-1)
          const double sp = __builtin_fma(pa2, bcast(xv, pj2), pa1 * bcast(xv, pj1)) - p_rhs;
          if (dbg_clk && lane == 0 && iters == QMPC_DBG_ITER) dbg_clk[9] = clock64();
          const bool dep = uni(!(delta > 1e-11 * cn));
          if (rebuild) {
            // the add event of slot rl, exactly as in the full step below (r is zero for the slots not rebuilt yet)
            if (uni(!(delta > 0.0))) {
              retry = true;
              break;
            }
            const double s = rsqrt_full(delta);
            const auto en = pool + neva * EV;
            {
              double zv[RE], gv[KQ];
#pragma unroll
              for (int q = 0; q < RE; ++q) zv[q] = z[q] * s;
#pragma unroll
              for (int k = 0; k < KQ; ++k) gv[k] = (lane + 64 * k == rl) ? s : ((wcid[k] >= 0) ? -rw[k] * s : 0.0);
              rec_store(en, zv, gv);
            }
            neva += 1;
            if (GPOOL && (neva & (TR - 1)) == 0 && neva + TR <= KEV) zero_group(neva);
            if (!rb_any() && !need_p) con_coefs(p_e, mi, pj1, pj2, pa1, pa2);  // the pending constraint's coefficients again
            __builtin_amdgcn_wave_barrier();
            continue;
          }
          if (WARM && forced && dep) {  // a candidate that depends on the ones already added: skip it
            need_p = true;
            continue;
          }
          const double t2 = dep ? __builtin_inf() : -sp * fast_rcp(dep ? 1.0 : delta);
          double ratio[KQ], rmin = __builtin_inf();
#pragma unroll
          for (int k = 0; k < KQ; ++k) {
            ratio[k] = __builtin_inf();
            if (!(WARM && forced) && wcid[k] >= 0 && rw[k] > 0.0) {
              const double qv = lam[k] * fast_rcp(rw[k]);
              ratio[k] = qv > 0.0 ? qv : 0.0;
            }
            rmin = (k == 0 || ratio[k] < rmin) ? ratio[k] : rmin;
          }
          double t1 = __builtin_inf();
          int l = -1;
          if (khw > 0 && !(WARM && forced)) {
            t1 = wave_min_pos_f64(rmin);
            if (uni(t1 < __builtin_inf())) {
#pragma unroll
              for (int k = 0; k < KQ; ++k) {
                const unsigned long long hit = __ballot(ratio[k] == t1);
                if (l < 0 && hit != 0ull) l = 64 * k + __ffsll((long long)hit) - 1;
              }
            }
          }
          const double t = (t2 <= t1) ? t2 : t1;
          if (uni(!(t < __builtin_inf()))) {
            status |= QMPC_DEV_ST_INFEASIBLE;
            break;
          }
          if (!dep) {
#pragma unroll
            for (int q = 0; q < RE; ++q) xv[q] = __builtin_fma(t, z[q], xv[q]);
          }
#pragma unroll
          for (int k = 0; k < KQ; ++k) lam[k] -= t * rw[k];
          lp += t;
          iters += 1;
          if (uni(t2 <= t1)) {
            // ---- full step: p joins the working set in the first free slot (an add event)
            int qslot = -1;
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
              const unsigned long long fm = __ballot(lane + 64 * k < KS && wcid[k] < 0);
              if (qslot < 0 && fm != 0ull) qslot = 64 * k + __ffsll((long long)fm) - 1;
            }
            if (qslot < 0) {
              retry = true;  // out of working-set slots: the robot is re-run with the Schur-form engine
              break;
            }
            const double s = rsqrt_full(delta);
            const auto en = pool + neva * EV;
            {
              double zv[RE], gv[KQ];
#pragma unroll
              for (int q = 0; q < RE; ++q) zv[q] = z[q] * s;
#pragma unroll
              for (int k = 0; k < KQ; ++k) gv[k] = (lane + 64 * k == qslot) ? s : ((wcid[k] >= 0) ? -rw[k] * s : 0.0);
              rec_store(en, zv, gv);
            }
#pragma unroll
            for (int k = 0; k < KQ; ++k)
              if (lane + 64 * k == qslot) {
                wcid[k] = p_e;
                lam[k] = lp;
              }
            if (lane == psl) amask |= (1u << pty);
            khw = (qslot + 1 > khw) ? qslot + 1 : khw;
            neva += 1;
            if (GPOOL && (neva & (TR - 1)) == 0 && neva + TR + ((nevd + TR - 1) & ~(TR - 1)) <= KEV) zero_group(neva);
            need_p = true;
          } else {
            // ---- partial step: the multiplier of slot l reached zero -> drop it (a drop event)
            if (!drop_slot(l, false)) break;
          }
          __builtin_amdgcn_wave_barrier();
          if (dbg_clk && lane == 0 && iters == QMPC_DBG_ITER + 1) dbg_clk[10] = clock64();
        }
        need_p0 = need_p;
        return retry;
      };
      __builtin_amdgcn_s_setprio(QMPC_ENGINE_PRIO);  // the serial part of the workgroup: win issue arbitration
      if constexpr (C::GLOBAL_EVENTS) {
        // (this workgroup's slice of the class's pool, taken in the kernel prologue)
        GlobalF64* const gpool = (GlobalF64*)P.evpool + (size_t)S.evslot * ((size_t)KEV_G * EV);
        for (int idx = lane; idx < TR_G * EV; idx += 64) {
          gpool[idx] = 0.0;
          gpool[(size_t)(KEV_G - TR_G) * EV + idx] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        retry = run(std::true_type{}, gpool);
      } else {
        retry = run(std::false_type{}, nullptr);
        if (spill && !retry) {
          // ---- LDS pool full: take a slice of the overflow pool, move the records there -- add events from the bottom,
          // drop events from the top, the rest of the group each side is in zeroed -- and go on.  Slices are RECYCLED: a flag
          // per slice, taken here (compare-and-swap, the probe sequence started at a per-call counter so that concurrent
          // robots spread out) and released when this robot is done with it, so the need is bounded by the robots in flight
          // (at most the resident workgroups, fewer than the 2048 slices of a handle), not by the robots of a call: no
          // Schur-form fallback for want of a slice, whatever the batch size and the launch order.  A robot that finds
          // every slice taken waits for one (their holders depend on nobody); the wait is bounded (ov_spin probes), and a
          // robot that times out -- or a handle without slices -- is re-run with the Schur-form engine, loudly
          int slice = -1;
          if (lane == 0 && P.ov_flags && P.ov_nslice > 0) {
            const unsigned start = P.ov_count ? (unsigned)atomicAdd(P.ov_count, 1) : (unsigned)rid;
            const unsigned ns = (unsigned)P.ov_nslice;
            unsigned idx = start % ns;
            int probe = 0;
            for (; probe < P.ov_spin; ++probe) {
              if (atomicCAS(&P.ov_flags[idx], 0, 1) == 0) {
                slice = (int)idx;
                break;
              }
              idx = (idx + 1u == ns) ? 0u : idx + 1u;
              if (idx == start % ns) __builtin_amdgcn_s_sleep(32);  // once round: everything is taken, give the holders time
            }
            // (statistics of the call, next to the spill counter: probes that found a slice taken, robots that timed out --
            //  qmpc_debug_read_counts; off the hot path: only a robot that spills gets here)
            if (P.ov_count && probe > 0) atomicAdd(P.ov_count + (QMPC_CNT_OV_PROBES - QMPC_CNT_OV), probe);
            if (P.ov_count && slice < 0) atomicAdd(P.ov_count + (QMPC_CNT_OV_TIMEOUT - QMPC_CNT_OV), 1);
          }
          slice = __builtin_amdgcn_readfirstlane(slice);
          // the slice's previous tenant may have run on another XCD: pair its release (agent-scope fence + flag store) with an
          // agent-scope acquire, so that nothing this wave reads from the slice later can come from a stale L1 / L2 line
          // (today every byte of a slice is written by its tenant before it is read -- copy + zero_group -- this keeps it
          //  correct if that ever changes; spill path only)
          if (slice >= 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          if (slice < 0) {
            retry = true;  // no slice (none configured / timed out): the robot is re-run with the Schur-form engine
          } else {
            GlobalF64* const gpool = (GlobalF64*)P.ovpool + (size_t)slice * QMPC_OV_SLICE;
            const double* lp_ = Sb.Sinv;
            for (int e = 0; e < neva; ++e)
              for (int idx = lane; idx < EV; idx += 64) gpool[(size_t)e * EV + idx] = lp_[e * EV + idx];
            for (int e = 0; e < nevd; ++e)
              for (int idx = lane; idx < EV; idx += 64) gpool[(size_t)(KEV_G - 1 - e) * EV + idx] = lp_[(KEV_L - 1 - e) * EV + idx];
            for (int e = neva; e < (neva & ~(TR_G - 1)) + TR_G; ++e)
              for (int idx = lane; idx < EV; idx += 64) gpool[(size_t)e * EV + idx] = 0.0;
            for (int e = (KEV_G - 1 - nevd) & ~(TR_G - 1); e <= KEV_G - 1 - nevd; ++e)
              for (int idx = lane; idx < EV; idx += 64) gpool[(size_t)e * EV + idx] = 0.0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            status |= QMPC_DEV_ST_SPILLED;  // informational
            retry = run(std::true_type{}, gpool);
            // done with the slice (the iterate lives in registers): every store to it reaches memory before the flag falls --
            // the next tenant may sit on another XCD, whose L2 is not coherent with this one's inside a kernel
            // (helper waves: their last read of the slice lies before the barrier at which run() received their partial sums)
            __threadfence();
            if (lane == 0) atomicExch(&P.ov_flags[slice], 0);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (NHELP > 0) {  // the helper waves leave their loop
        if (lane == 0) S.hd.cmd = 0;
        __syncthreads();
      }
      QMPC_TICK(6);
      if (!retry) {
        // outputs: get_solution(0..11) = forces of the four feet at horizon step 0
        // (convexMPC_interface.cpp:175-180, ConvexMPCLocomotion.cpp:672-685)
        // an iterate the method abandoned (constraints found inconsistent) is not a solution:
        // the robot is reported and its forces read zero rather than a primal-infeasible point
        const bool dead = (status & (QMPC_DEV_ST_INFEASIBLE | QMPC_DEV_ST_WS_FULL)) != 0;
        if (lane < 12) P.grf[(size_t)rid * 12 + lane] = 0.f;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          const int j = lane + 64 * q;
          if (j < n && !dead) {
            const int k = S.sidx[j / 3], ax = j % 3;  // foot-step of this variable
            if (k < 4) P.grf[(size_t)rid * 12 + 3 * k + ax] = (float)xv[q];
            if (P.soln) P.soln[(size_t)rid * 12 * h + 3 * k + ax] = xv[q];
          }
        }
        {
          bool nf = false;  // NaN / Inf anywhere in the result (non-finite input): report it
#pragma unroll
          for (int q = 0; q < RE; ++q) nf |= (lane + 64 * q < n) && !(__builtin_fabs(xv[q]) < __builtin_inf());
          if (__ballot(nf)) status |= QMPC_DEV_ST_NONFINITE;
        }
        if (lane == 0) {
          P.status[rid] = S.status | status;
          if (P.iters) P.iters[rid] = iters;
          if (P.hint_iters) P.hint_iters[rid] = iters;
          if (P.hint_max_w) *P.hint_max_w = iters;  // (one round: the robot that finishes last is the hardest; a plain store, no atomic on anybody's critical path)
          if (cmdm) cmd_finish_state();
        }
        if (WARM && P.ws)  // the final working set, as global ids, for the next cycle's warm start
          P.ws[(size_t)rid * QMPC_WS_STRIDE + lane] =  // (the first QMPC_WS_STRIDE slots; a guess need not be complete)
              (lane < KS && wcid[0] >= 0 && !dead) ? 5 * (int)S.sidx[wcid[0] / 5] + (wcid[0] % 5) : -1;
        if (cmdm && P.f_ff) {
          float* fb = reinterpret_cast<float*>(Sb.D);  // diag(H^-1) is dead now
          if (lane < 12) fb[lane] = 0.f;
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < RE; ++q) {
            const int j = lane + 64 * q;
            if (j < n && !dead && S.sidx[j / 3] < 4) fb[3 * S.sidx[j / 3] + j % 3] = (float)xv[q];
          }
          __builtin_amdgcn_wave_barrier();
          if (lane < 12) cmd_finish_forces(fb, lane);
        }
      }
      if (lane == 0) S.mode = retry ? 1 : 0;
    } else if constexpr (NHELP > 0) {
      // ---- every other wave: wait for a request; waves 1..NHELP accumulate their share of the events into a
      // partial sum of their own (the remaining waves only keep the barrier count)
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      while (true) {
        __syncthreads();  // (A)
        if (__builtin_amdgcn_readfirstlane(S.hd.cmd) == 0) break;
        if (wv <= NHELP) {
          const int hj1 = __builtin_amdgcn_readfirstlane(S.hd.pj1), hj2 = __builtin_amdgcn_readfirstlane(S.hd.pj2);
          const int hna = __builtin_amdgcn_readfirstlane(S.hd.neva), hnd = __builtin_amdgcn_readfirstlane(S.hd.nevd);
          const double ha1 = S.hd.pa1, ha2 = S.hd.pa2;
          double zp[RE], rp[KQ];
#pragma unroll
          for (int q = 0; q < RE; ++q) zp[q] = 0.0;
#pragma unroll
          for (int k = 0; k < KQ; ++k) rp[k] = 0.0;
          if (__builtin_amdgcn_readfirstlane(S.hd.glob) != 0)
            ev_part(std::true_type{}, (GlobalF64*)(size_t)S.hd.gptr, hj1, hj2, ha1, ha2, hna, hnd, wv, NHELP + 1, zp, rp);
          else
            ev_part(std::false_type{}, (double*)Sb.Sinv, hj1, hj2, ha1, ha2, hna, hnd, wv, NHELP + 1, zp, rp);
          double* const mine = hpart + (wv - 1) * EV;
#pragma unroll
          for (int q = 0; q < RE; ++q)
            if (zw[q]) mine[zo[q]] = zp[q];
#pragma unroll
          for (int k = 0; k < KQ; ++k)
            if (lane + 64 * k < KS) mine[NPE + lane + 64 * k] = rp[k];
        }
        __syncthreads();  // (B)
      }
    }
  } else {
  // ------------------------------------------------------------ stage 5 (Schur form)
  // Goldfarb-Idnani dual active set on the explicit inverse, run by wave 0
  // alone: no block barrier inside the loop.  Engine state lives in registers:
  //   lane = variable index i (+64q)   : x_i, (H^-1 c_p)_i, z_i
  //   lane = stance slot sl            : working-set membership of its 5 rows
  //   lane = working-set slot w (+64q) : constraint id, multiplier, r_w
  // Cross-lane traffic is DPP / readlane / bpermute; matrix data comes from the
  // packed H^-1, S_W^-1 and the pooled rows H^-1 c_w in LDS.
  if (engine) {
    const double mi = P.mu_inv;
    const double inv_fr = P.inv_fr_norm;
    const double tol = P.tol;
    const int max_iter = P.max_iter;
    __builtin_amdgcn_s_setprio(QMPC_ENGINE_PRIO);  // the serial part of the workgroup: win issue arbitration
    auto Hinv = [&](int r, int cidx) __attribute__((always_inline)) {
      const int hi = r > cidx ? r : cidx, lo = r > cidx ? cidx : r;
      return Sb.Hp[hi * (hi + 1) / 2 + lo];
    };
    int rbl[RE];  // packed-row base of this lane's variables
#pragma unroll
    for (int q = 0; q < RE; ++q) rbl[q] = (lane + 64 * q) * (lane + 64 * q + 1) / 2;
    // element (row = lane + 64 q, column j) of H^-1 for a wave-uniform j
    auto Hcol = [&](int q, int j) __attribute__((always_inline)) {
      const int row = lane + 64 * q;
      const int tj = j * (j + 1) / 2;  // scalar
      return Sb.Hp[(j <= row) ? rbl[q] + j : tj + row];
    };
    // value of the index-major vector v at variable j (per-lane j): a bpermute per 64-block
    auto gather = [&](const double (&v)[RE], int j) __attribute__((always_inline)) {
      double out = 0.0;
#pragma unroll
      for (int q = 0; q < RE; ++q) {
        const double cand = __shfl(v[q], j & 63);
        if ((j >> 6) == q) out = cand;
      }
      return out;
    };
    // the same for a wave-uniform j
    auto bcast = [&](const double (&v)[RE], int j) __attribute__((always_inline)) {
      return lane_elem<RE>(v, j);
    };
    // row w of M = H^-1 C_W lives at the back of the pool while it does not collide with S_W^-1
    // The packed inverse only occupies n(n+1)/2 doubles of Hp: the working-set storage
    // starts right behind it, so a smaller problem gets room for more constraints
    // (class 3: 48 slots at n = 192, 96 at n <= 168).
    const int nh = n * (n + 1) / 2;
    double* const swp = Sb.Hp + nh;                      // packed S_W^-1, growing from the front
    const int cap = C::NH + C::NPOOL - nh;              // doubles available
    constexpr int NPE = 64 * RE;  // row stride of M: a lane never indexes past its row
    auto m_row = [&](int w) __attribute__((always_inline)) { return swp + cap - NPE * (w + 1); };

    unsigned amask = 0;  // stance-slot lane: bit ty = constraint (sl, ty) is in the working set
    int wcid[KW];        // working-slot lane: constraint id in slot w + 64 q, -1 = free
    int wj1[KW], wj2[KW];   // ... and its row c_w = wa1 e_wj1 + wa2 e_wj2 (cached)
    double wa1[KW], wa2[KW];
    double lam[KW], rw[KW];
    int khw = 0;         // high-water mark of used working-set slots (uniform)
    int mvalid = 0;      // rows M[0..mvalid) are stored (uniform)
    int status = 0;
#pragma unroll
    for (int q = 0; q < KW; ++q) {
      wcid[q] = -1;
      wj1[q] = wj2[q] = 0;
      wa1[q] = wa2[q] = 0.0;
      lam[q] = 0.0;
      rw[q] = 0.0;
    }

    while (true) {
      // ---- pick the most violated constraint outside the working set
      //      (normalised by its row norm); none -> optimal
      unsigned key = 0;
      {
        // slot lane sl needs x[3sl..3sl+2] of the index-major x
        const int j0 = 3 * (lane < nst ? lane : 0);
        const double x0 = gather(xv, j0), x1 = gather(xv, j0 + 1), x2 = gather(xv, j0 + 2);
        if (lane < nst) {
          // most violated of this slot's rows outside the working set (normalised)
          const double fx = mi * x0, fy = mi * x1;
          double vmin = 0.0;
          int tmin = -1;
          const double sv[5] = {(fx + x2) * inv_fr, (x2 - fx) * inv_fr, (fy + x2) * inv_fr, (x2 - fy) * inv_fr, fmx - x2};
#pragma unroll
          for (int ty = 0; ty < 5; ++ty) {
            const bool cand = !((amask >> ty) & 1u) && sv[ty] < vmin;
            vmin = cand ? sv[ty] : vmin;
            tmin = cand ? ty : tmin;
          }
          if (vmin < -tol)  // more negative -> larger float magnitude -> larger key; low 9 bits = id
            key = (__float_as_uint((float)(-vmin)) & ~0x1FFu) | (unsigned)(5 * lane + tmin);
        }
      }
      const unsigned best = wave_max_u32(key);
      if (best == 0u) break;
      if (iters >= max_iter) {
        status |= QMPC_DEV_ST_MAXITER;
        break;
      }
      // uniform description of the constraint p being added: c_p = pa1 e_pj1 + pa2 e_pj2
      const int p_e = (int)(best & 0x1FFu);
      const int psl = p_e / 5, pty = p_e - 5 * psl;
      int pj1, pj2;
      double pa1, pa2;
      con_coefs(p_e, mi, pj1, pj2, pa1, pa2);
      const bool two = (pa2 != 0.0);
      const double p_rhs = (pty == 4) ? -readlane_f64(fmx, psl) : 0.0;
      if (dbg_clk && lane == 0 && iters == 0) dbg_clk[8] = clock64();

      // hc = H^-1 c_p, index-major
      double hc[RE];
#pragma unroll
      for (int q = 0; q < RE; ++q) {
        const int row = lane + 64 * q;
        hc[q] = (row < n) ? pa1 * Hcol(q, pj1) + (two ? pa2 * Hcol(q, pj2) : 0.0) : 0.0;
      }
      const double hcn = pa1 * bcast(hc, pj1) + (two ? pa2 * bcast(hc, pj2) : 0.0);  // c_p^T H^-1 c_p
      // d = C_W^T H^-1 c_p does not change while p is being added (only r does)
      double dw[KW];
#pragma unroll
      for (int q = 0; q < KW; ++q) {
        const double h1 = gather(hc, wj1[q]), h2 = gather(hc, wj2[q]);
        dw[q] = (wcid[q] >= 0) ? wa1[q] * h1 + wa2[q] * h2 : 0.0;
      }
      double lp = 0.0;  // multiplier of p
      bool done = false;

      // ---- inner loop: one pass per (partial or full) step
      while (true) {
        // r = S_W^-1 d  (slots that were dropped hold d-contributions of 0 rows/cols)
#pragma unroll
        for (int q = 0; q < KW; ++q) rw[q] = 0.0;
        for (int v0 = 0; v0 < khw; v0 += 4) {
          double sv[KW][4];  // 4 entries of row w per lane, loaded back-to-back (one LDS latency)
#pragma unroll
          for (int q = 0; q < KW; ++q) {
            const int w = lane + 64 * q;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int v = v0 + u;
              sv[q][u] = (w < khw && v < khw) ? swp[sym_idx(w, v)] : 0.0;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int v = v0 + u;
            double dv = lane_elem<KW>(dw, v);
            const int ev = lane_elem<KW>(wcid, v);
            if (v >= khw || ev < 0) dv = 0.0;
#pragma unroll
            for (int q = 0; q < KW; ++q) rw[q] = __builtin_fma(sv[q][u], dv, rw[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < KW; ++q)
          if (wcid[q] < 0) rw[q] = 0.0;
        // z = H^-1 (c_p - C_W r) = hc - sum_w r_w (H^-1 c_w), index-major
        double z[RE];
#pragma unroll
        for (int q = 0; q < RE; ++q) z[q] = hc[q];
        {
          const int wfast = khw < mvalid ? khw : mvalid;
          for (int w0 = 0; w0 < wfast; w0 += 4) {  // stored rows: one load per lane and slot
            double mv[4][RE];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int q = 0; q < RE; ++q) mv[u][q] = (w0 + u < wfast) ? m_row(w0 + u)[lane + 64 * q] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int w = w0 + u;
              const double rv = (w < wfast) ? lane_elem<KW>(rw, w) : 0.0;
#pragma unroll
              for (int q = 0; q < RE; ++q) z[q] = __builtin_fma(-rv, mv[u][q], z[q]);
            }
          }
          for (int w = wfast; w < khw; ++w) {  // rows that did not fit the pool: recompute from Hp
            const int e = lane_elem<KW>(wcid, w);
            if (e < 0) continue;  // uniform
            const double rv = lane_elem<KW>(rw, w);
            int j1, j2;
            double a1, a2;
            con_coefs(e, mi, j1, j2, a1, a2);
#pragma unroll
            for (int q = 0; q < RE; ++q) {
              const int row = lane + 64 * q;
              if (row < n) {
                const double hw = a1 * Hcol(q, j1) + (a2 != 0.0 ? a2 * Hcol(q, j2) : 0.0);
                z[q] = __builtin_fma(-rv, hw, z[q]);
              }
            }
          }
        }
        if (dbg_clk && lane == 0 && iters == 0) dbg_clk[9] = clock64();
        // delta = c_p^T z, current violation of p, step lengths
        const double delta = pa1 * bcast(z, pj1) + (two ? pa2 * bcast(z, pj2) : 0.0);
        const double sp = pa1 * bcast(xv, pj1) + (two ? pa2 * bcast(xv, pj2) : 0.0) - p_rhs;
        const bool dep = !(delta > 1e-12 * hcn);
        const double rdelta = fast_rcp(dep ? 1.0 : delta);
        const double t2 = dep ? __builtin_inf() : -sp * rdelta;
        // t1: largest dual step keeping the working-set multipliers >= 0
        double ratio = __builtin_inf();
        int lq = 0;
#pragma unroll
        for (int q = 0; q < KW; ++q) {
          if (wcid[q] >= 0 && rw[q] > 0.0) {
            double qv = lam[q] * fast_rcp(rw[q]);
            qv = qv > 0.0 ? qv : 0.0;
            if (qv < ratio) {
              ratio = qv;
              lq = q;
            }
          }
        }
        double t1 = __builtin_inf();
        int l = -1;
        if (khw > 0) {
          t1 = wave_min_pos_f64(ratio);
          if (t1 < __builtin_inf()) {
            const unsigned long long m = __ballot(ratio == t1);
            const int ll = __ffsll((long long)m) - 1;
            l = ll + 64 * __builtin_amdgcn_readlane(lq, ll);
          }
        }
        const double t = (t2 <= t1) ? t2 : t1;
        if (!(t < __builtin_inf())) {
          status |= QMPC_DEV_ST_INFEASIBLE;
          done = true;
          break;
        }
        if (!dep) {
#pragma unroll
          for (int q = 0; q < RE; ++q) xv[q] = __builtin_fma(t, z[q], xv[q]);
        }
#pragma unroll
        for (int q = 0; q < KW; ++q) lam[q] -= t * rw[q];
        lp += t;
        iters += 1;
        if (t2 <= t1) {
          // full step: constraint p joins the working set in a free slot
          int qslot = -1;
#pragma unroll
          for (int q = 0; q < KW; ++q) {
            const bool fr = (lane + 64 * q < KMAX) && (wcid[q] < 0);
            const unsigned long long m = __ballot(fr);
            if (qslot < 0 && m) qslot = 64 * q + __ffsll((long long)m) - 1;
          }
          if (qslot < 0 || ((qslot + 1 > khw) ? (qslot + 1) * (qslot + 2) / 2 : 0) > cap) {
            status |= QMPC_DEV_ST_WS_FULL;
            done = true;
            break;
          }
          const int kn = (qslot + 1 > khw) ? qslot + 1 : khw;
          // the packed S_W^-1 grows with the high-water mark: rows of M it would
          // overwrite are given up first
          {
            const int room = (cap - kn * (kn + 1) / 2) / NPE;  // rows of M that still fit behind it
            const int want = (qslot == mvalid) ? mvalid + 1 : mvalid;
            mvalid = want < room ? want : (mvalid < room ? mvalid : room);
            if (mvalid < 0) mvalid = 0;
          }
          const double dinv = rdelta;
          // bordered-inverse update of S_W^-1 (free slots have r = 0):
          //   S[a][b] += r_a r_b / delta ; S[q][a] = -r_a / delta ; S[q][q] = 1/delta
          for (int h0 = 0; h0 < kn; h0 += 4) {
            double old[KW][4];
#pragma unroll
            for (int q = 0; q < KW; ++q)
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int hi = h0 + u, lo = lane + 64 * q;
                old[q][u] = (hi < kn && lo <= hi) ? swp[hi * (hi + 1) / 2 + lo] : 0.0;
              }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int hi = h0 + u;
              const double rhi = lane_elem<KW>(rw, hi);
#pragma unroll
              for (int q = 0; q < KW; ++q) {
                const int lo = lane + 64 * q;
                if (hi < kn && lo <= hi) {
                  double v;
                  if (hi == qslot && lo == qslot) v = dinv;
                  else if (hi == qslot) v = -rw[q] * dinv;
                  else if (lo == qslot) v = -rhi * dinv;
                  else v = __builtin_fma(rhi * dinv, rw[q], old[q][u]);
                  swp[hi * (hi + 1) / 2 + lo] = v;
                }
              }
            }
          }
          if (qslot < mvalid) {
#pragma unroll
            for (int q = 0; q < RE; ++q) m_row(qslot)[lane + 64 * q] = hc[q];
          }
#pragma unroll
          for (int q = 0; q < KW; ++q)
            if (lane + 64 * q == qslot) {
              wcid[q] = p_e;
              wj1[q] = pj1;
              wj2[q] = pj2;
              wa1[q] = pa1;
              wa2[q] = pa2;
              lam[q] = lp;
              rw[q] = 0.0;
            }
          if (lane == psl) amask |= (1u << pty);
          khw = kn;
          if (dbg_clk && lane == 0 && iters == 1) dbg_clk[10] = clock64();
          break;  // next violated constraint
        }
        // partial step: the multiplier of slot l hit zero -> drop it.
        //   S' = S - S[:,l] S[l,:] / S[l,l] on the other slots; row/col l is
        //   only read here and zeroed afterwards, so in place is safe.
        const double il = fast_rcp(swp[sym_idx(l, l)]);
        for (int hi = 0; hi < khw; ++hi) {
          if (hi == l) continue;
          const double shl = swp[sym_idx(hi, l)] * il;
#pragma unroll
          for (int q = 0; q < KW; ++q) {
            const int lo = lane + 64 * q;
            if (lo <= hi && lo != l) {
              const int idx = hi * (hi + 1) / 2 + lo;
              swp[idx] = __builtin_fma(-shl, swp[sym_idx(l, lo)], swp[idx]);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < KW; ++q) {
          const int w = lane + 64 * q;
          if (w < khw) swp[sym_idx(w, l)] = 0.0;
        }
        // the dropped constraint leaves its slot: clear lane-w and lane-sl state
        const int de = lane_elem<KW>(wcid, l);
#pragma unroll
        for (int q = 0; q < KW; ++q)
          if (lane + 64 * q == l) {
            wcid[q] = -1;
            wj1[q] = wj2[q] = 0;
            wa1[q] = wa2[q] = 0.0;
            lam[q] = 0.0;
            rw[q] = 0.0;
            dw[q] = 0.0;
          }
        if (lane == de / 5) amask &= ~(1u << (de % 5));
      }
      if (done) break;
    }
    __builtin_amdgcn_s_setprio(0);
    QMPC_TICK(6);

    // ---------------------------------------------------------- outputs
    // get_solution(0..11): forces of the four feet at horizon step 0
    // (convexMPC_interface.cpp:175-180, ConvexMPCLocomotion.cpp:672-685);
    // feet in swing at step 0 read 0.
    // (an abandoned iterate -- working-set storage exhausted, constraints inconsistent -- is
    //  not a solution: the robot is reported and its forces read zero)
    const bool dead = (status & (QMPC_DEV_ST_INFEASIBLE | QMPC_DEV_ST_WS_FULL)) != 0;
    if (lane < 12) P.grf[(size_t)rid * 12 + lane] = 0.f;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < RE; ++q) {
      const int j = lane + 64 * q;
      if (j < n && !dead) {
        const int k = S.sidx[j / 3], ax = j % 3;  // foot-step of this variable
        if (k < 4) P.grf[(size_t)rid * 12 + 3 * k + ax] = (float)xv[q];
        if (P.soln) P.soln[(size_t)rid * 12 * h + 3 * k + ax] = xv[q];
      }
    }
    {
      bool nf = false;  // NaN / Inf anywhere in the result (non-finite input): report it
#pragma unroll
      for (int q = 0; q < RE; ++q) nf |= (lane + 64 * q < n) && !(__builtin_fabs(xv[q]) < __builtin_inf());
      if (__ballot(nf)) status |= QMPC_DEV_ST_NONFINITE;
    }
    if (lane == 0) {
      P.status[rid] = S.status | status;
      if (P.iters) P.iters[rid] = iters;
      if (P.hint_iters) P.hint_iters[rid] = iters;
      if (P.hint_max_w) *P.hint_max_w = iters;  // (one round: the robot that finishes last is the hardest; a plain store, no atomic on anybody's critical path)
      if (cmdm) cmd_finish_state();
    }
    if (WARM && P.ws)  // (this engine always starts cold; it still leaves its working set for the next cycle)
      P.ws[(size_t)rid * QMPC_WS_STRIDE + lane] =
          (wcid[0] >= 0 && !dead) ? 5 * (int)S.sidx[wcid[0] / 5] + (wcid[0] % 5) : -1;
    if (cmdm && P.f_ff) {
      float* fb = reinterpret_cast<float*>(Sb.D);  // not used by this engine
      if (lane < 12) fb[lane] = 0.f;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < RE; ++q) {
        const int j = lane + 64 * q;
        if (j < n && !dead && S.sidx[j / 3] < 4) fb[3 * S.sidx[j / 3] + j % 3] = (float)xv[q];
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 12) cmd_finish_forces(fb, lane);
    }
  }
    if (engine && lane == 0) S.mode = 0;
  }
  __syncthreads();  // the other waves wait here for the engine
  QMPC_TICK(7);
  return S.mode != 0;
}

}  // namespace

// one robot with the engine pair of its class: event-form engine first (it continues in global memory when its
// on-chip pool fills up); the (rare) robot that runs out of working-set SLOTS there is solved again from scratch
// with the Schur-form engine, which has more of them
template <int RB, bool CMD, bool WARM, bool PRIO = false>
__device__ __forceinline__ void solve_robot(const int rid, const int tid, Smem<RB>& S, const QmpcParams& P) {
  if constexpr (Cfg<RB>::EVENT_ENGINE) {
    bool again;
    if (Cfg<RB>::GLOBAL_EVENTS && S.evslot < 0) again = true;  // no pool slice (pool_acquire timed out): Schur form only
    else again = solve_one<RB, true, CMD, false, WARM, false, false, PRIO>(rid, tid, S, P);
    if (again) {
      __syncthreads();
      // opaque thread id: without it the compiler keeps per-thread values of the
      // first run alive (spilled to scratch by EVERY workgroup) for this rare second run
      int tid2 = tid;
      asm volatile("" : "+v"(tid2));
      solve_one<RB, false, CMD, false, WARM>(rid, tid2, S, P);
      __syncthreads();
      if (tid2 == 0) P.status[rid] |= QMPC_DEV_ST_FALLBACK;  // informational
    }
  } else {
    solve_one<RB, false, CMD, false, WARM>(rid, tid, S, P);
  }
  // (thread 0 wrote the robot's status; the launch that re-solves what the decoupled engine handed back says so)
  if (P.status_or != 0 && tid == 0) P.status[rid] |= P.status_or;
}

// The 96-row class (Cfg<4>::BALANCE): eight waves launched, six stay.  Returns the LOGICAL thread index (wave rank among
// the survivors x 64 + lane), or -1 for a wave that must exit.  One barrier, taken by all eight waves; the hardware drops
// a terminated wave from every later barrier.  P.cu_slots: one int per CU (xcc, se, sh, cu), bit 0 / bit 1 = a resident
// workgroup keeps its pairs on SIMDs {0,1} / {2,3}; taken with atomicOr here, released by balance_release at the end.
// Anything unexpected (a placement other than two waves per SIMD, no free bit, no slot array) falls back to "waves 0..5
// stay", which is what a 384-thread launch does.
template <int RB>
__device__ __forceinline__ int balance_waves(Smem<RB>& S, const QmpcParams& P) {
  if constexpr (!Cfg<RB>::BALANCE) {
    return (int)threadIdx.x;
  } else {
    const int pw = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 0) S.bal_simd[pw] = (int)((hwid >> 4) & 3u);
    if (threadIdx.x == 0) {
      int bit = 0;
      const int cu = (int)(((xcc & 7u) << 8) | ((hwid >> 8) & 0xffu));
      if (P.cu_slots && P.bal_debug == 0) {
        if (!(atomicOr(&P.cu_slots[cu], 1) & 1)) bit = 1;
        else if (!(atomicOr(&P.cu_slots[cu], 2) & 2)) bit = 2;
      }
      S.bal_cu = cu;
      S.bal_bit = bit;
    }
    __syncthreads();
    const int pairs_hi = (S.bal_bit == 2) ? 1 : 0;  // 1: both waves stay on SIMDs 2 and 3, one each on 0 and 1
    int seen[4] = {0, 0, 0, 0};
    unsigned keep = 0u;
    bool regular = true;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int sd = S.bal_simd[w] & 3;
      int cnt = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) cnt = (sd == q) ? seen[q] : cnt;
      const bool paired = ((sd >> 1) == pairs_hi);
      if (cnt == 0 || (cnt == 1 && paired)) keep |= 1u << w;
      if (cnt >= 2) regular = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) seen[q] += (sd == q) ? 1 : 0;
    }
    if (!regular || __builtin_popcount(keep) != 6 || P.bal_debug == 2) keep = 0x3fu;
    if (!((keep >> pw) & 1u)) return -1;
    return 64 * __builtin_popcount(keep & ((1u << pw) - 1u)) + lane;
  }
}
template <int RB>
__device__ __forceinline__ void balance_release(int tid, Smem<RB>& S, const QmpcParams& P) {
  if constexpr (Cfg<RB>::BALANCE) {
    if (tid == 0 && S.bal_bit != 0) atomicAnd(&P.cu_slots[S.bal_cu], ~S.bal_bit);
  }
}

// class 3: take a slice of the global event pool for the lifetime of the workgroup: slot = workgroup index
// modulo the slice count, guarded by a flag (a workgroup whose predecessor on that slice is still running --
// it would have to be ~ev_nslot / 256 times slower than average -- waits for it).  The wait is bounded (a flag
// left behind by an aborted launch must not hang this one); a workgroup that does NOT get its slice never
// touches it: S.evslot = -1, its robots are solved by the Schur-form engine (no global pool) and carry
// QMPC_ST_FALLBACK -- slower, never silently wrong.
template <int RB>
__device__ __forceinline__ void pool_acquire(Smem<RB>& S, const QmpcParams& P) {
  if constexpr (Cfg<RB>::GLOBAL_EVENTS) {
    if (threadIdx.x == 0) {
      const int slot = (int)(blockIdx.x % (unsigned)P.ev_nslot);
      bool mine = false;
      for (int spin = 0; spin < P.ev_spin; ++spin) {
        if (atomicCAS(&P.evflags[slot], 0, 1) == 0) {
          mine = true;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      S.evslot = mine ? slot : -1;
    }
    __syncthreads();
  }
}
template <int RB>
__device__ __forceinline__ void pool_release(Smem<RB>& S, const QmpcParams& P) {
  if constexpr (Cfg<RB>::GLOBAL_EVENTS) {
    __syncthreads();
    if (threadIdx.x == 0 && S.evslot >= 0) {
      __threadfence();
      atomicExch(&P.evflags[S.evslot], 0);
    }
  }
}

// SIZE ORDER: a launch of several rounds of workgroups ends with whichever long robot started last.  What a
// robot will cost is not known before its inverse exists -- but its record says a good deal (qmpc_robot_keys above): the
// reduced size n_r = 3 x stance foot-steps (the sweep is n_r / 2 steps long) and a score that follows the iteration count
// with a correlation of 0.7.  So the workgroups beyond so_first (the host's choice: the second round on) take the robots THAT FIT THIS CLASS largest
// first and, among equals, highest score first; the robots the class only hands on stay among their own places (bunching those
// costs more than any order of theirs gives: they hide behind their neighbours' sweeps) and are ordered among themselves by
// the score, for the NEXT class's sake: its queue is filled in dispatch order, and that launch (two workgroups per CU, robots
// that iterate up to forty times) ends with whichever long robot it took last.
// The permutation is built INSIDE the launch while the first rounds are being solved -- a sort kernel in front of the launch
// would cost what the order gives (4 us of 126): the robots from so_first on are dealt to so_nseg segments of at most 4096
// (robot so_first + j + nseg t belongs to segment j), workgroup j sorts segment j WITHIN itself before it solves its own robot
// (one workgroup sorting everything took 5 ns per robot, too long from 8192 robots on) with keys, positions and the sorted
// list in LDS -- the solve's phase-local storage, not in use yet -- and hands the
// result over through 8-byte entries (call tag << 32 | robot): a workgroup reads its own entry until the tag is this call's.
// Nobody waits in practice (the first entry is needed ~40 us into the launch, a segment is sorted in well under 10); the wait
// is bounded all the same, and a workgroup that gives up takes robot = blockIdx.x -- the identity, which is what every reader
// of the segment gets if its builder never ran.  Results do not depend on the order (a robot is solved by one workgroup from
// its own record).
#define QMPC_SO_SEG 4096
#define QMPC_SO_HEAD 16  // robots / places of the first round per segment (qmpc_capi.cpp sizes the segments: QMPC_SO_SEG - QMPC_SO_HEAD strided robots at most)
// The same builder serves the ORDER HINT (so_hint != nullptr: the iteration counts the handle's previous call left, one per
// robot): key = the count, every robot takes part (a robot that is handed on carries the count of the class that solved it,
// so the next class's queue comes out hardest first as well) -- in place of a sort kernel in front of the call.
// keys of the builder: a robot that FITS this class 16 * size level + score level (size first: 32 levels, largest = 0; then 16
// score levels, two per octave, hardest = 0) -- host emulation of the orders on configs[2]: by size 3.54e7, by score 3.70e7, size
// then score 3.96e7 QP/s (plain 3.22e7; profiles/r06_z_proxy_order.md) --; a robot that is only handed on 512 + one of 64 score
// levels (six per octave); with the order hint 8 * (63 - the previous call's count).
#define QMPC_SO_BINS 576
template <int RB, bool CMD>
__device__ __forceinline__ void size_order_build(const int tid, Smem<RB>& S, const QmpcParams& P) {
  constexpr int NT = Cfg<RB>::NT, NW = NT / 64;
  constexpr int BPT = (QMPC_SO_BINS + NT - 1) / NT;  // bins per thread of the prefix
  static_assert(sizeof(S.u) >= QMPC_SO_BINS * 4 + 48 * 4 + QMPC_SO_SEG * 6, "size order: scratch in the phase-local storage");
  int* hist = reinterpret_cast<int*>(&S.u);  // QMPC_SO_BINS counts, then offsets
  int* wcnt = hist + QMPC_SO_BINS;           // [0..15] fitting robots per wave, [16..31] handed on, [32..47] the prefix's wave totals
  unsigned short* keys = reinterpret_cast<unsigned short*>(wcnt + 48);  // [seg] bin (0xffff: nobody)
  unsigned short* place = keys + QMPC_SO_SEG;    // [seg] the fitting robots' places in index order, then the others'
  unsigned short* sorted = place + QMPC_SO_SEG;  // [seg] the robots in bin order
  const int lane = tid & 63, wv = tid >> 6;
  const int maxfit = P.so_maxfit;
  // this workgroup's segment: the robots (and places) s0 + nseg t, t < n -- STRIDED, so that every segment is a sample of the
  // whole batch and the t-th largest robots of all segments land next to each other: the launch as a whole runs from large to
  // small without the builders exchanging a word
  const int nseg = P.so_nseg, s0 = P.so_first + (int)blockIdx.x;
  // ... preceded by QMPC_SO_HEAD robots (and places) of the launch's FIRST round, nseg + j + nseg t: the workgroups right behind the
  // builders wait the few microseconds a builder takes and start the segment's largest / hardest robots at once -- a robot that
  // will iterate thirty times must not start a round and a half into the launch (closed-loop rollouts: a handful of such robots,
  // mean count 1.6)
  const int n = QMPC_SO_HEAD + (P.batch - s0 + nseg - 1) / nseg;
  auto gidx = [&](int t) __attribute__((always_inline)) {
    return t < QMPC_SO_HEAD ? nseg + (int)blockIdx.x + nseg * t : s0 + nseg * (t - QMPC_SO_HEAD);
  };
  const unsigned long long tag = (unsigned long long)P.so_tag << 32;
  // an entry is stored TWICE: plainly into the near copy (the line stays in this XCD's L2: workgroup b runs on XCD b % 8 and
  // so_nseg is a multiple of 8, so a segment's readers sit on its builder's XCD and their one probe is an L2 hit) and written
  // through (sc1) into the far copy, which is what a reader polls if the near probe did not show this call's tag -- whatever
  // the placement, a tag that matches was written in this call, and the far copy is the plain tagged-granule hand-off
  auto so_put = [&](int at, int robot) __attribute__((always_inline)) {
    const unsigned long long v = tag | (unsigned)robot;
    P.so_order[at] = v;
    __hip_atomic_store(P.so_far + at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  __builtin_amdgcn_s_setprio(3);
  for (int k = tid; k < QMPC_SO_BINS + 48; k += NT) hist[k] = 0;
  __syncthreads();
  // wave wv: the robots [lo, hi) of the segment, 128 at a time (two per lane: their loads in flight together)
  const int per = ((n + NW - 1) / NW + 63) & ~63;
  const int lo = wv * per, hi = (lo + per < n) ? lo + per : n;
  int nfit = 0, nhand = 0;
  for (int base = lo; base < hi; base += 128) {
    int key[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = base + 64 * u + lane;
      key[u] = -1;
      if (i < hi) {
        if (CMD || P.so_hint) {
          // (command mode: the contact table exists in registers only -- the count or nothing.  The count ALONE also where the
          //  table is there: joined with the size -- added to it in the sweep's unit, or behind it in the two-level key -- the
          //  exact hint on configs[2] gives 3.90e7 / 3.94e7 instead of 4.03e7, on configs[4] 2.02e7 instead of 2.17e7)
          const int d = 63 - P.so_hint[gidx(i)];
          key[u] = 8 * (d < 0 ? 0 : d);
        } else {
          const QmpcKeys kk = qmpc_robot_keys(P, gidx(i));
          const int d = maxfit - kk.nst;
          if (d >= 0) key[u] = 16 * (d > 31 ? 31 : d) + qmpc_score_level(kk.score, 2.f, 16);
          else key[u] = 512 + qmpc_score_level(kk.score, 6.f, 64);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = base + 64 * u + lane;
      if (key[u] >= 0) {
        keys[i] = (unsigned short)key[u];
        atomicAdd(&hist[key[u]], 1);
      }
      nfit += __popcll(__ballot(key[u] >= 0 && key[u] < 512));
      nhand += __popcll(__ballot(key[u] >= 512));
    }
  }
  if (lane == 0) {
    wcnt[wv] = nfit;
    wcnt[16 + wv] = nhand;
  }
  __syncthreads();
  int fbase = 0, hbase = 0, nfit_all = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int cf = wcnt[w], ch = wcnt[16 + w];
    fbase += (w < wv) ? cf : 0;
    hbase += (w < wv) ? ch : 0;
    nfit_all += cf;
  }
  // exclusive prefix over the bins: BPT consecutive bins per thread, a shuffle scan per wave, the waves' totals through LDS
  int cnt[BPT], mysum = 0;
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int bin = BPT * tid + q;
    cnt[q] = bin < QMPC_SO_BINS ? hist[bin] : 0;
    mysum += cnt[q];
  }
  int acc = mysum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(acc, d);
    if (lane >= d) acc += up;
  }
  if (lane == 63) wcnt[32 + wv] = acc;
  __syncthreads();  // (everybody has read the counts; the waves' totals are there)
  int off = acc - mysum;
#pragma unroll
  for (int w = 0; w < NW; ++w) off += (w < wv) ? wcnt[32 + w] : 0;
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int bin = BPT * tid + q;
    if (bin < QMPC_SO_BINS) hist[bin] = off;
    off += cnt[q];
  }
  __syncthreads();
  int runf = fbase, runh = nfit_all + hbase;
  for (int base = lo; base < hi; base += 64) {
    const int i = base + lane;
    const int key = (i < hi) ? (int)keys[i] : 0xffff;
    const bool fit = key < 512, hand = key >= 512 && key < QMPC_SO_BINS;
    const unsigned long long mf = __ballot(fit), mh = __ballot(hand);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (fit) place[runf + __popcll(mf & below)] = (unsigned short)i;
    if (hand) place[runh + __popcll(mh & below)] = (unsigned short)i;
    if (fit || hand) sorted[atomicAdd(&hist[key], 1)] = (unsigned short)i;
    runf += __popcll(mf);
    runh += __popcll(mh);
  }
  __syncthreads();
  // k < nfit_all: the k-th place of a fitting robot takes the k-th fitting robot in bin order; beyond: the same among the others
  for (int k = tid; k < n; k += NT) so_put(gidx(place[k]), gidx(sorted[k]));
  __syncthreads();  // (the scratch is the solve's from here on)
  __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ int size_order_take(const QmpcParams& P) {
  // one probe of the near copy (bypassing this CU's L1: an L2 hit when the builder ran on this XCD) ...
  unsigned long long v = __hip_atomic_load(P.so_order + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((unsigned)(v >> 32) == P.so_tag) return __builtin_amdgcn_readfirstlane((int)(unsigned)v);
  // ... then the far copy until it carries this call's tag
  const unsigned long long* e = P.so_far + blockIdx.x;
  for (int spin = 0; spin < (1 << 18); ++spin) {  // (bounded: ~1 s)
    v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(v >> 32) == P.so_tag) return __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    __builtin_amdgcn_s_sleep(32);
  }
  return (int)blockIdx.x;
}

#ifndef QMPC_LISTED_VARIANT
#define QMPC_LISTED_VARIANT 2
#endif

// LISTED = false: the first class launched: robot = blockIdx.x; clears the NEXT call's list counters and queue
// heads.  LISTED = true: every later class, launched with at most one workgroup per resident slot; it consumes
// the list the previous classes filled as a queue: entry blockIdx.x first, then whatever entry the head counter
// hands out.  (A grid of `batch` workgroups of which most find no entry costs more than the few solves
// themselves: the no-op workgroups are dispatched at this class's LDS-limited occupancy.)
template <int RB, bool CMD, bool WARM = false, bool LISTED = false>
__global__ __launch_bounds__(Cfg<RB>::NT_LAUNCH, Cfg<RB>::MIN_WAVES) void qmpc_solve_kernel(const QmpcParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char qmpc_smem[];
  Smem<RB>& S = *reinterpret_cast<Smem<RB>*>(qmpc_smem);
  // (the 96-row class: eight waves launched, six stay -- the logical thread index from here on)
  const int tid0 = balance_waves<RB>(S, P);
  if (tid0 < 0) return;
  if constexpr (!LISTED) {
    // (block index first: only block 0 waits for the kernel argument)
    if (blockIdx.x == 0 && P.clear_counts)
      for (int k = tid0; k < QMPC_COUNTERS; k += Cfg<RB>::NT) P.clear_counts[k] = 0;  // 3 counters + 3 heads
    if (blockIdx.x == 0 && tid0 == 0 && P.hint_max_z) *P.hint_max_z = 0;
    pool_acquire<RB>(S, P);
    // (order hint: the previous call's hardest robots first; results do not depend on the order)
    int rid = P.order ? __builtin_amdgcn_readfirstlane(P.order[blockIdx.x]) : (int)blockIdx.x;
    // (size order / order hint: the first workgroups build the permutation, a segment each, before they solve their own robots;
    //  the workgroups of the later rounds read it)
    if (P.so_order) {
      if ((int)blockIdx.x < P.so_nseg) size_order_build<RB, CMD>(tid0, S, P);
      else if ((int)blockIdx.x >= P.so_first || (int)blockIdx.x < P.so_nseg * (QMPC_SO_HEAD + 1)) rid = size_order_take(P);
    }
    __builtin_assume(tid0 >= 0 && tid0 < Cfg<RB>::NT);
    // (the first class of a chain: the only launch that can be one round.  Not the five-per-CU instantiation: it exists for launches
    //  of several rounds, and at 96 registers the staging's code in stage 0 costs it spills on the main path)
    solve_robot<RB, CMD, WARM, RB != 6>(rid, tid0, S, P);
    pool_release<RB>(S, P);
    balance_release<RB>(tid0, S, P);
  } else {
    const int nlist = *P.count;
    if ((int)blockIdx.x >= nlist) {  // uniform
      balance_release<RB>(tid0, S, P);
      return;
    }
    pool_acquire<RB>(S, P);
    for (int idx = (int)blockIdx.x;;) {
      const int rid = P.list[idx];
      // opaque thread id per robot: nothing derived from it is loop-invariant, so the compiler cannot hoist
      // per-thread values out of the loop (and spill them across the whole solve)
      int tid1 = tid0;
#if QMPC_LISTED_VARIANT >= 1
      asm volatile("" : "+v"(tid1));
      __builtin_assume(tid1 >= 0 && tid1 < Cfg<RB>::NT);
#endif
#if QMPC_LISTED_VARIANT >= 2
      // ... and an opaque kernel-argument pointer: the ~30 scalar loads of stage 0 stay inside the loop instead
      // of being hoisted into SGPRs that then spill (the parameter block is the only explicit kernel argument,
      // so it sits at offset 0 of the kernarg segment)
      typedef const __attribute__((address_space(4))) QmpcParams* KernargPtr;
      KernargPtr pk = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(pk));
      const QmpcParams& PK = *(const QmpcParams*)pk;
      solve_robot<RB, CMD, WARM>(rid, tid1, S, PK);
#else
      solve_robot<RB, CMD, WARM>(rid, tid1, S, P);
#endif
      __syncthreads();  // every wave is done with this robot's LDS state
      if (tid0 == 0) S.qnext = (int)gridDim.x + atomicAdd(P.qhead, 1);
      __syncthreads();
      idx = S.qnext;
      if (idx >= nlist) break;  // uniform
    }
    pool_release<RB>(S, P);
    balance_release<RB>(tid0, S, P);
  }
}

// JCQP alternate (update_solver_settings' use_jcqp = 1 / 2): same assembly and sweep, ADMM instead of the
// active set; record mode only, same size-class chain
template <int RB, bool LISTED = false>
__global__ __launch_bounds__(Cfg<RB>::NT, Cfg<RB>::MIN_WAVES) void qmpc_admm_kernel(const QmpcParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char qmpc_smem[];
  Smem<RB>& S = *reinterpret_cast<Smem<RB>*>(qmpc_smem);
  if constexpr (!LISTED) {
    if (blockIdx.x == 0 && P.clear_counts)
      for (int k = threadIdx.x; k < QMPC_COUNTERS; k += blockDim.x) P.clear_counts[k] = 0;
    solve_one<RB, false, false, true>((int)blockIdx.x, (int)threadIdx.x, S, P);
  } else {  // list = queue, as in qmpc_solve_kernel
    const int nlist = *P.count;
    if ((int)blockIdx.x >= nlist) return;  // uniform
    for (int idx = (int)blockIdx.x;;) {
      const int rid = P.list[idx];
      int tid1 = (int)threadIdx.x;
      asm volatile("" : "+v"(tid1));
      __builtin_assume(tid1 >= 0 && tid1 < Cfg<RB>::NT);
      solve_one<RB, false, false, true>(rid, tid1, S, P);
      __syncthreads();
      if (threadIdx.x == 0) S.qnext = (int)gridDim.x + atomicAdd(P.qhead, 1);
      __syncthreads();
      idx = S.qnext;
      if (idx >= nlist) break;  // uniform
    }
  }
}

// Producer half of the decoupled path (classes 2 and 3): assembly, sweep, x_u, inverse -> work item.  Same launch
// geometry rules as qmpc_solve_kernel (first of the chain: robot = blockIdx.x; LISTED: the list as a queue).
template <int RB, bool CMD, bool LISTED = false>
__global__ __launch_bounds__(Cfg<RB>::NT, Cfg<RB>::MIN_WAVES_A) void qmpc_sweep_kernel(const QmpcParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char qmpc_smem[];
  Smem<RB>& S = *reinterpret_cast<Smem<RB>*>(qmpc_smem);
  // (a launch covers one CHUNK of the class: robots / list entries rid0 .. list_hi - 1, at most as many as the class has work items;
  //  the chunks of a call run one after the other on the caller's stream)
  if constexpr (!LISTED) {
    if (blockIdx.x == 0 && P.clear_counts)
      for (int k = threadIdx.x; k < QMPC_COUNTERS; k += blockDim.x) P.clear_counts[k] = 0;
    solve_one<RB, true, CMD, false, false, true>(P.rid0 + (int)blockIdx.x, (int)threadIdx.x, S, P);
  } else {
    int nlist = *P.count;
    nlist = nlist < P.list_hi ? nlist : P.list_hi;
    if (P.rid0 + (int)blockIdx.x >= nlist) return;  // uniform
    for (int idx = P.rid0 + (int)blockIdx.x;;) {
      const int rid = P.list[idx];
      int tid1 = (int)threadIdx.x;
      asm volatile("" : "+v"(tid1));
      __builtin_assume(tid1 >= 0 && tid1 < Cfg<RB>::NT);
      typedef const __attribute__((address_space(4))) QmpcParams* KernargPtr;
      KernargPtr pk = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(pk));
      const QmpcParams& PK = *(const QmpcParams*)pk;
      solve_one<RB, true, CMD, false, false, true>(rid, tid1, S, PK);
      __syncthreads();
      if (threadIdx.x == 0) S.qnext = P.rid0 + (int)gridDim.x + atomicAdd(P.qhead, 1);
      __syncthreads();
      idx = S.qnext;
      if (idx >= nlist) break;  // uniform
    }
  }
}

// Producer of the LARGE problems (192 < n_r <= 432; solve_one<..., BIG>): list-consuming, one workgroup per CU (its block
// sweep uses the whole union as scratch).  Built with the 192-row class's translation unit.
template <bool CMD>
__global__ __launch_bounds__(Cfg<3>::NT, Cfg<3>::MIN_WAVES) void qmpc_big_kernel(const QmpcParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char qmpc_smem[];
  Smem<3>& S = *reinterpret_cast<Smem<3>*>(qmpc_smem);
  // (a launch covers one CHUNK of the list: entries rid0 .. list_hi - 1, at most as many as there are work items)
  int nlist = *P.count;
  nlist = nlist < P.list_hi ? nlist : P.list_hi;
  if (P.rid0 + (int)blockIdx.x >= nlist) return;  // uniform
  for (int idx = P.rid0 + (int)blockIdx.x;;) {
    const int rid = P.list[idx];
    int tid1 = (int)threadIdx.x;
    asm volatile("" : "+v"(tid1));
    __builtin_assume(tid1 >= 0 && tid1 < Cfg<3>::NT);
    typedef const __attribute__((address_space(4))) QmpcParams* KernargPtr;
    KernargPtr pk = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pk));
    const QmpcParams& PK = *(const QmpcParams*)pk;
    solve_one<3, true, CMD, false, false, true, true>(rid, tid1, S, PK);
    __syncthreads();
    if (threadIdx.x == 0) S.qnext = P.rid0 + (int)gridDim.x + atomicAdd(P.qhead, 1);
    __syncthreads();
    idx = S.qnext;
    if (idx >= nlist) break;  // uniform
  }
}

namespace {
// LDS of the producer: everything but the active-set storage
template <int RB>
constexpr size_t sweep_smem() {
  return sizeof(Smem<RB>) - sizeof(typename Smem<RB>::U) + sizeof(typename Smem<RB>::U::AW);
}
template <typename K>
hipError_t set_smem(K kernel, size_t bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <int RB>
hipError_t prepare_class() {
  hipError_t e;
  const size_t n = sizeof(Smem<RB>);
  if ((e = set_smem(qmpc_solve_kernel<RB, false, false, false>, n)) != hipSuccess) return e;
  if ((e = set_smem(qmpc_solve_kernel<RB, true, false, false>, n)) != hipSuccess) return e;
  if ((e = set_smem(qmpc_solve_kernel<RB, false, true, false>, n)) != hipSuccess) return e;
  if ((e = set_smem(qmpc_admm_kernel<RB, false>, n)) != hipSuccess) return e;
  if constexpr (!Cfg<RB>::C1) {  // (class 1 is only ever launched first)
    if ((e = set_smem(qmpc_solve_kernel<RB, false, false, true>, n)) != hipSuccess) return e;
    if ((e = set_smem(qmpc_solve_kernel<RB, true, false, true>, n)) != hipSuccess) return e;
    if ((e = set_smem(qmpc_solve_kernel<RB, false, true, true>, n)) != hipSuccess) return e;
    if ((e = set_smem(qmpc_admm_kernel<RB, true>, n)) != hipSuccess) return e;
  }
  if constexpr (RB == 2 || RB == 3) {
    const size_t na = sweep_smem<RB>();
    if ((e = set_smem(qmpc_sweep_kernel<RB, false, false>, na)) != hipSuccess) return e;
    if ((e = set_smem(qmpc_sweep_kernel<RB, true, false>, na)) != hipSuccess) return e;
    if ((e = set_smem(qmpc_sweep_kernel<RB, false, true>, na)) != hipSuccess) return e;
    if ((e = set_smem(qmpc_sweep_kernel<RB, true, true>, na)) != hipSuccess) return e;
  }
  return hipSuccess;
}
template <int RB>
void launch_sweep(bool cmd, const QmpcParams* P, int grid, hipStream_t stream) {
  if constexpr (RB == 2 || RB == 3) {
    const dim3 g(grid), b(Cfg<RB>::NT);
    const size_t n = sweep_smem<RB>();
    if (P->list) {
      if (cmd) hipLaunchKernelGGL((qmpc_sweep_kernel<RB, true, true>), g, b, n, stream, *P);
      else hipLaunchKernelGGL((qmpc_sweep_kernel<RB, false, true>), g, b, n, stream, *P);
    } else {
      if (cmd) hipLaunchKernelGGL((qmpc_sweep_kernel<RB, true, false>), g, b, n, stream, *P);
      else hipLaunchKernelGGL((qmpc_sweep_kernel<RB, false, false>), g, b, n, stream, *P);
    }
  }
}
template <int RB>
int resident_sweep() {
  if constexpr (RB == 2 || RB == 3) {
    static int cached = 0;
    if (cached) return cached;
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, qmpc_sweep_kernel<RB, false, true>, Cfg<RB>::NT,
                                                                      sweep_smem<RB>());
    if (e != hipSuccess || per < 1 || cus < 1) return 0;
    return cached = per * cus;
  } else {
    return 0;
  }
}
template <int RB, bool LISTED>
void launch_variant(bool cmd, const QmpcParams* P, int grid, hipStream_t stream) {
  const dim3 g(grid), b(Cfg<RB>::NT), bs(Cfg<RB>::NT_LAUNCH);  // (the solve kernels of the 96-row class: eight waves launched, six stay)
  const size_t n = sizeof(Smem<RB>);
  if (P->admm_mode)
    hipLaunchKernelGGL((qmpc_admm_kernel<RB, LISTED>), g, b, n, stream, *P);
  else if (P->ws && !cmd)  // warm start across cycles: its own instantiation (record mode)
    hipLaunchKernelGGL((qmpc_solve_kernel<RB, false, true, LISTED>), g, bs, n, stream, *P);
  else if (cmd)
    hipLaunchKernelGGL((qmpc_solve_kernel<RB, true, false, LISTED>), g, bs, n, stream, *P);
  else
    hipLaunchKernelGGL((qmpc_solve_kernel<RB, false, false, LISTED>), g, bs, n, stream, *P);
}
template <int RB>
void launch_one(bool cmd, const QmpcParams* P, int grid, hipStream_t stream) {
  if constexpr (!Cfg<RB>::C1) {
    if (P->list) return launch_variant<RB, true>(cmd, P, grid, stream);
  }
  launch_variant<RB, false>(cmd, P, grid, stream);
}
}  // namespace

// workgroups of the class that fit on the device at once (the grid of a list-consuming launch); the
// instantiations of a class share LDS size and launch bounds, so the record-mode kernel stands for all of them
template <int RB>
int resident_class() {
  static int cached = 0;
  if (cached) return cached;
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, qmpc_solve_kernel<RB, false, false, !Cfg<RB>::C1>,
                                                                    Cfg<RB>::NT_LAUNCH, sizeof(Smem<RB>));
  if (e != hipSuccess || per < 1 || cus < 1) return 0;
  return cached = per * cus;
}

// Host entry points, one set per size class (qmpc_capi.cpp dispatches on the class).  The file is compiled once
// per class with -DQMPC_RB=<class> (four translation units built in parallel by __graft_entry__.build()), or
// once without it for all four.  The command-mode instantiation is selected by P->c_position != nullptr.
#define QMPC_DEFINE_CLASS(RB)                                                                              \
  extern "C" size_t qmpc_c##RB##_smem(void) { return sizeof(Smem<RB>); }                                   \
  extern "C" hipError_t qmpc_c##RB##_prepare(void) { return prepare_class<RB>(); }                         \
  extern "C" int qmpc_c##RB##_resident(void) { return resident_class<RB>(); }                              \
  extern "C" hipError_t qmpc_c##RB##_launch(const QmpcParams* P, int grid, hipStream_t stream) {           \
    launch_one<RB>(P->c_position != nullptr, P, grid, stream);                                             \
    return hipGetLastError();                                                                              \
  }                                                                                                        \
  extern "C" int qmpc_c##RB##_resident_sweep(void) { return resident_sweep<RB>(); }                        \
  extern "C" hipError_t qmpc_c##RB##_launch_sweep(const QmpcParams* P, int grid, hipStream_t stream) {     \
    launch_sweep<RB>(P->c_position != nullptr, P, grid, stream);                                           \
    return hipGetLastError();                                                                              \
  }
#if !defined(QMPC_RB) || QMPC_RB == 1
// test hook (qmpc_debug_keys): what qmpc_robot_keys says about every robot of a batch -- the size order's and the one-round
// staging's keys, as the kernels evaluate them
__global__ void qmpc_keys_kernel(const QmpcParams P, int* __restrict__ nst, float* __restrict__ score, float* __restrict__ demand) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= P.batch) return;
  const QmpcKeys k = qmpc_robot_keys<true>(P, i);
  nst[i] = k.nst;
  score[i] = k.score;
  demand[i] = k.demand;
}
extern "C" hipError_t qmpc_launch_keys(const QmpcParams* P, int* nst, float* score, float* demand, hipStream_t stream) {
  hipLaunchKernelGGL(qmpc_keys_kernel, dim3((P->batch + 255) / 256), dim3(256), 0, stream, *P, nst, score, demand);
  return hipGetLastError();
}
QMPC_DEFINE_CLASS(1)
#endif
#if !defined(QMPC_RB) || QMPC_RB == 2
QMPC_DEFINE_CLASS(2)
#endif
#if !defined(QMPC_RB) || QMPC_RB == 3
QMPC_DEFINE_CLASS(3)
extern "C" hipError_t qmpc_big_prepare(void) {
  hipError_t e = set_smem(qmpc_big_kernel<false>, sizeof(Smem<3>));
  if (e != hipSuccess) return e;
  return set_smem(qmpc_big_kernel<true>, sizeof(Smem<3>));
}
extern "C" hipError_t qmpc_big_launch(const QmpcParams* P, int grid, hipStream_t stream) {
  const dim3 g(grid), b(Cfg<3>::NT);
  if (P->c_position != nullptr) hipLaunchKernelGGL((qmpc_big_kernel<true>), g, b, sizeof(Smem<3>), stream, *P);
  else hipLaunchKernelGGL((qmpc_big_kernel<false>), g, b, sizeof(Smem<3>), stream, *P);
  return hipGetLastError();
}
#endif
#if !defined(QMPC_RB) || QMPC_RB == 4
QMPC_DEFINE_CLASS(4)
#endif
#if !defined(QMPC_RB) || QMPC_RB == 6
static_assert(sizeof(Smem<6>) <= 32768, "64-row class at five workgroups per CU");
QMPC_DEFINE_CLASS(6)
#endif
