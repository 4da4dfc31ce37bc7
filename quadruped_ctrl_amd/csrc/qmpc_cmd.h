// qmpc_cmd.h -- the float arithmetic of the caller side, shared by the
// stand-alone pack kernel (qmpc_pack.hip) and the fused command mode of the solve
// kernel (qmpc_kernels.hip stage 0), so both produce the same record bit for bit.
// Every helper restates one piece of ConvexMPCLocomotion::updateMPCIfNeeded
// (src/MPC_Ctrl/ConvexMPCLocomotion.cpp:498-577) or ::solveDenseMPC (:592-680)
// operation by operation; fp contraction is switched off inside each body because
// the reference's host code has no fma.
#ifndef QMPC_CMD_H
#define QMPC_CMD_H

#include <hip/hip_runtime.h>

struct QmpcTrajGen {
  float init[12];     // trajInitial
  float inc[3];       // per-step increments of rows 2, 3, 4 (dtMPC * yaw rate / v_des_world)
  float xs, ys;       // clamped world_position_desired (unchanged when standing)
  bool stand;
};

// :505-507  v_des_world = omniMode ? v_des_robot : rBody^T v_des_robot
__device__ __forceinline__ void qmpc_cmd_vdes_world(const float* R, float vx_r, float vy_r, int omni, float& vw0,
                                                    float& vw1) {
#pragma clang fp contract(off)
  vw0 = vx_r;
  vw1 = vy_r;
  if (!omni) {
    vw0 = ((R[0] * vx_r) + (R[3] * vy_r)) + (R[6] * 0.f);
    vw1 = ((R[1] * vx_r) + (R[4] * vy_r)) + (R[7] * 0.f);
  }
}

// :534-545  pull the desired position to within 0.1 m of the estimate
// ("p[0] + 0.1": double literal, result stored to float)
__device__ __forceinline__ float qmpc_cmd_clamp(float start, float p) {
#pragma clang fp contract(off)
  const float max_pos_error = .1f;
  if (start - p > max_pos_error) start = (float)((double)p + 0.1);
  if (p - start > max_pos_error) start = (float)((double)p - 0.1);
  return start;
}

// trajInitial (:514-531 standing, :547-561 moving) and the increments of :566-573
__device__ __forceinline__ void qmpc_cmd_traj_gen(QmpcTrajGen& g, bool stand, const float* stand_traj, const float* rp_des,
                                                  const float* rpy_comp, float yaw_des_true, float wpd_x, float wpd_y,
                                                  float p0, float p1, float body_height, float yaw_rate, float vw0,
                                                  float vw1, float dt_mpc) {
#pragma clang fp contract(off)
  g.stand = stand;
  g.xs = wpd_x;
  g.ys = wpd_y;
  g.inc[0] = g.inc[1] = g.inc[2] = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) g.init[j] = 0.f;
  if (stand) {
    g.init[0] = rp_des ? rp_des[0] : 0.f;
    g.init[1] = rp_des ? rp_des[1] : 0.f;
    g.init[2] = stand_traj[5];
    g.init[3] = stand_traj[0];
    g.init[4] = stand_traj[1];
    g.init[5] = body_height;
  } else {
    g.xs = qmpc_cmd_clamp(wpd_x, p0);
    g.ys = qmpc_cmd_clamp(wpd_y, p1);
    g.init[0] = rpy_comp[0];
    g.init[1] = rpy_comp[1];
    g.init[2] = yaw_des_true;
    g.init[3] = g.xs;
    g.init[4] = g.ys;
    g.init[5] = body_height;
    g.init[8] = yaw_rate;
    g.init[9] = vw0;
    g.init[10] = vw1;
    g.inc[0] = dt_mpc * yaw_rate;
    g.inc[1] = dt_mpc * vw0;
    g.inc[2] = dt_mpc * vw1;
  }
}

// trajAll[12 k + j] (:563-576): rows 2, 3, 4 are running float sums, step by step
__device__ __forceinline__ float qmpc_cmd_traj_value(const QmpcTrajGen& g, int k, int j) {
#pragma clang fp contract(off)
  float val = g.init[0];
#pragma unroll
  for (int q = 1; q < 12; ++q) {
    float t = g.init[q];
    asm volatile("" : "+v"(t));  // keeps the select chain from becoming an indexed load of a scratch copy
    val = (j == q) ? t : val;
  }
  if (!g.stand && j >= 2 && j <= 4) {
    float i0 = g.inc[0], i1 = g.inc[1], i2 = g.inc[2];
    asm volatile("" : "+v"(i0), "+v"(i1), "+v"(i2));
    const float inc = (j == 2) ? i0 : (j == 3 ? i1 : i2);
    for (int s = 0; s < k; ++s) val = val + inc;
  }
  return val;
}

// OffsetDurationGait::getMpcTable (src/MPC_Ctrl/Gait.cpp:142-166), entry (step i, leg)
__device__ __forceinline__ int qmpc_cmd_gait_bit(int i, int iteration, int offset, int duration, int n_segments) {
  const int iter = (i + iteration + 1) % n_segments;
  int progress = iter - offset;
  if (progress < 0) progress += n_segments;
  return (progress < duration) ? 1 : 0;
}

// :625, :636-640  x_comp_integral after update_x_drag(x_comp_integral) (:632)
__device__ __forceinline__ float qmpc_cmd_xci_next(float xci, float p2, float body_height, float dt_mpc, float vx) {
#pragma clang fp contract(off)
  const float pz_err = p2 - body_height;
  if ((double)vx > 0.3 || (double)vx < -0.3) xci = xci + __fdiv_rn(((3.0f * pz_err) * dt_mpc), vx);
  return xci;
}

// :672-680  (f_ff[leg])[i] = (-rBody * f)[i]
__device__ __forceinline__ float qmpc_cmd_f2b(const float* Rrow, float f0, float f1, float f2) {
#pragma clang fp contract(off)
  return (((-Rrow[0]) * f0) + ((-Rrow[1]) * f1)) + ((-Rrow[2]) * f2);
}

// :611-613  r[axis*4 + foot] = pFoot[foot][axis] - position[axis]
__device__ __forceinline__ float qmpc_cmd_foot_offset(float pf, float p) {
#pragma clang fp contract(off)
  return pf - p;
}

// :598
__device__ __forceinline__ float qmpc_cmd_weight(int j) {
  const float Q[12] = {2.5f, 2.5f, 10.f, 50.f, 50.f, 100.f, 0.f, 0.f, 0.5f, 0.2f, 0.2f, 0.1f};
  float qv = Q[0];
#pragma unroll
  for (int q = 1; q < 12; ++q) qv = (j == q) ? Q[q] : qv;
  return qv;
}

#endif
