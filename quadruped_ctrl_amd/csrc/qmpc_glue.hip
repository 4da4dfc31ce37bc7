// qmpc_glue.hip -- per-tick glue either side of the MPC solve, batched on the GPU:
//   qmpc_leg_kin_kernel    LegController::updateData (src/Controllers/LegController.cpp:89-110):
//                          foot position and Jacobian of every leg from the joint angles
//                          (computeLegJacobianAndPosition, :204-244) and foot velocity v = J qd
//   qmpc_leg_cmd_kernel    LegController::updateCommand (:116-160): Cartesian PD on the foot,
//                          tau = tauFF + J^T (forceFF + Kp (pDes - p) + Kd (vDes - v)), the joint PD
//                          of GaitCtrller's ctrlParam(2..3), and the desired joint angles (computeLegIK)
//   qmpc_swing_kernel      FootSwingTrajectory::computeSwingTrajectoryBezier
//                          (src/Controllers/FootSwingTrajectory.cpp:17-37)
// One thread per (robot, leg): small fixed-size float algebra, pure streaming of narrow rows
// (kinematics 96 B in / 240 B out per robot).  The arithmetic lives in qmpc_glue.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qmpc.h"

#include "qmpc_glue.h"

namespace {

__global__ __launch_bounds__(256) void qmpc_leg_kin_kernel(const QmpcLegGeom g, const float* __restrict__ q,
                                                           const float* __restrict__ qd, float* __restrict__ J,
                                                           float* __restrict__ p, float* __restrict__ v, const int n) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * 256 + threadIdx.x;  // robot * 4 + leg
  if (t >= n) return;
  const int leg = t & 3;
  const float q0 = q[(size_t)t * 3 + 0], q1 = q[(size_t)t * 3 + 1], q2 = q[(size_t)t * 3 + 2];
  float Jl[9], pl[3];
  qmpc_leg_fk(g, leg, q0, q1, q2, Jl, pl);
#pragma unroll
  for (int k = 0; k < 9; ++k) J[(size_t)t * 9 + k] = Jl[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[(size_t)t * 3 + k] = pl[k];
  if (v) {
    const float d0 = qd[(size_t)t * 3 + 0], d1 = qd[(size_t)t * 3 + 1], d2 = qd[(size_t)t * 3 + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[(size_t)t * 3 + k] = qmpc_row3(Jl + 3 * k, d0, d1, d2);  // datas[leg].v = J * qd  (:108)
  }
}

__global__ __launch_bounds__(256) void qmpc_leg_cmd_kernel(const QmpcLegGeom g, const qmpc_leg_command c,
                                                           float* __restrict__ tau, float* __restrict__ q_des,
                                                           const int n) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * 256 + threadIdx.x;  // robot * 4 + leg
  if (t >= n) return;
  const int leg = t & 3;
  const size_t o3 = (size_t)t * 3, o9 = (size_t)t * 9;
  float lt[3], ff[3], dp[3], dv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lt[k] = c.tau_ff ? c.tau_ff[o3 + k] : 0.f;      // legTorque = tauFeedForward            (:121)
    ff[k] = c.force_ff ? c.force_ff[o3 + k] : 0.f;  // footForce = forceFeedForward          (:124)
    dp[k] = c.p_des[o3 + k] - c.p[o3 + k];
    dv[k] = c.v_des[o3 + k] - c.v[o3 + k];
  }
  // footForce += kpCartesian * (pDes - p); footForce += kdCartesian * (vDes - v)             (:128-131)
  float add[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) add[k] = qmpc_row3(c.kp_cart + o9 + 3 * k, dp[0], dp[1], dp[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) ff[k] = ff[k] + add[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) add[k] = qmpc_row3(c.kd_cart + o9 + 3 * k, dv[0], dv[1], dv[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) ff[k] = ff[k] + add[k];
  // legTorque += J^T * footForce                                                             (:134)
  const float* Jl = c.J + o9;
#pragma unroll
  for (int k = 0; k < 3; ++k) lt[k] = lt[k] + ((Jl[k] * ff[0] + Jl[3 + k] * ff[1]) + Jl[6 + k] * ff[2]);
  // tau_*_ff[leg] = crtlParam(2) * (0 - q) - crtlParam(3) * qd + legTorque                   (:138-158)
#pragma unroll
  for (int k = 0; k < 3; ++k)
    tau[o3 + k] = c.kp_joint * (0.0f - c.q[o3 + k]) - c.kd_joint * c.qd[o3 + k] + lt[k];
  if (q_des) {  // computeLegIK(_quadruped, commands[leg].pDes, &qDes, leg)                  (:137)
    float qd3[3];
    qmpc_leg_ik(g, leg, c.p_des[o3 + 0], c.p_des[o3 + 1], c.p_des[o3 + 2], qd3);
#pragma unroll
    for (int k = 0; k < 3; ++k) q_des[o3 + k] = qd3[k];
  }
}

__global__ __launch_bounds__(256) void qmpc_swing_kernel(const float* __restrict__ p0, const float* __restrict__ pf,
                                                         const float* __restrict__ height, const float* __restrict__ phase,
                                                         const float* __restrict__ swing_time, float* __restrict__ p,
                                                         float* __restrict__ v, float* __restrict__ a, const int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // foot index (robot * 4 + foot), one thread per axis triple
  if (t >= n) return;
  const size_t o = (size_t)t * 3;
  const float a0 = p0[o], a1 = p0[o + 1], a2 = p0[o + 2], b0 = pf[o], b1 = pf[o + 1], b2 = pf[o + 2];
  const float hgt = height[t], ph = phase[t], st = swing_time[t];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    float pp, vv, aa;
    qmpc_swing_axis(ax, ax == 0 ? a0 : a1, ax == 0 ? b0 : b1, a2, b2, hgt, ph, st, pp, vv, aa);
    p[o + ax] = pp;
    v[o + ax] = vv;
    a[o + ax] = aa;
  }
}

}  // namespace

extern "C" hipError_t qmpc_launch_leg_kin(const float geom[4], const float* q, const float* qd, float* J, float* p,
                                          float* v, int batch, hipStream_t stream) {
  const QmpcLegGeom g{geom[0], geom[1], geom[2], geom[3]};
  const int n = batch * 4;
  hipLaunchKernelGGL(qmpc_leg_kin_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, g, q, qd, J, p, v, n);
  return hipGetLastError();
}

extern "C" hipError_t qmpc_launch_leg_cmd(const float geom[4], const qmpc_leg_command* c, float* tau, float* q_des,
                                          int batch, hipStream_t stream) {
  const QmpcLegGeom g{geom[0], geom[1], geom[2], geom[3]};
  const int n = batch * 4;
  hipLaunchKernelGGL(qmpc_leg_cmd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, g, *c, tau, q_des, n);
  return hipGetLastError();
}

extern "C" hipError_t qmpc_launch_swing(const float* p0, const float* pf, const float* height, const float* phase,
                                        const float* swing_time, float* p, float* v, float* a, int n_feet,
                                        hipStream_t stream) {
  hipLaunchKernelGGL(qmpc_swing_kernel, dim3((n_feet + 255) / 256), dim3(256), 0, stream, p0, pf, height, phase,
                     swing_time, p, v, a, n_feet);
  return hipGetLastError();
}
