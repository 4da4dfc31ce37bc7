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
    // (LegController.cpp:147-154 writes `crtlParam(2) * (0.0 - q)` with a DOUBLE literal: the joint-PD term and the sum
    //  are evaluated in double there and rounded to float once; kd * qd is a float product)
    tau[o3 + k] = (float)((double)c.kp_joint * (0.0 - (double)c.q[o3 + k]) - (double)(c.kd_joint * c.qd[o3 + k]) + (double)lt[k]);
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

// ---------------------------------------------------------------------------------------------------
// LinearKFPositionVelocityEstimator<float>::run (src/Controllers/PositionVelocityEstimator.cpp:66-221):
// the 18-state / 28-measurement Kalman filter that turns leg kinematics + IMU into body position and
// velocity, one WAVE per robot, every matrix of the filter in LDS (12 KB).  The measurement matrix C is
// never stored: each of its rows has one +1 and at most one -1 (:28-46), so C Pm, C Pm C^T and Pm C^T
// are differences of rows / columns of Pm.  The two S.lu().solve() calls (:183, :186) are ONE LU with
// partial pivoting on the augmented matrix [S | ey | C].  Every output element is computed by one lane
// with its inner sums in index order and no fma, i.e. the same float operations in the same order as the
// restatement (oracle/glue_oracle.c) -- bit-identical results.
constexpr int KF_N = 18, KF_M = 28, KF_W = KF_M + 1 + KF_N;

__device__ __forceinline__ int kf_ca(int r) { return r < 12 ? r % 3 : (r < 24 ? 3 + r % 3 : 8 + 3 * (r - 24)); }
__device__ __forceinline__ int kf_cb(int r) { return r < 12 ? 6 + r : -1; }

__global__ __launch_bounds__(64) void qmpc_kf_kernel(const qmpc_kf_state s, const float hx, const float hy, const float hz,
                                                     const int batch) {
#pragma clang fp contract(off)
  __shared__ float Pm[KF_N * KF_N], AP[KF_N * KF_N], K1[KF_N * KF_M], CP[KF_M * KF_N], Sa[KF_M * KF_W];
  __shared__ float xh[KF_N], Q[KF_N], R[KF_M], y[KF_M];
  __shared__ int pivrow;
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= batch) return;
  const float dt = 0.002f;
  float* Pg = s.P + (size_t)b * KF_N * KF_N;
  if (lane < KF_N) {
    xh[lane] = s.xhat[(size_t)b * KF_N + lane];
    // run() :73-76 with _Q0 of setup() :56-60
    Q[lane] = lane < 3 ? (dt / 20.f) * 0.02f : (lane < 6 ? (dt * 9.8f / 20.f) * 0.02f : dt * 0.002f);
  }
  if (lane < KF_M) R[lane] = 1.f * (lane < 12 ? 0.001f : (lane < 24 ? 0.1f : 0.001f));
  __syncthreads();
  if (lane < 4) {  // per leg :118-166
    const int i = lane;
    const float* rB = s.r_body + (size_t)b * 9;
    const float* om = s.omega_body + (size_t)b * 3;
    const float ph[3] = {(i == 0 || i == 1) ? hx : -hx, (i == 1 || i == 3) ? hy : -hy, hz};
    float p_rel[3], dp_rel[3], w[3], p_f[3], dp_f[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p_rel[k] = ph[k] + s.leg_p[(size_t)b * 12 + 3 * i + k];
      dp_rel[k] = s.leg_v[(size_t)b * 12 + 3 * i + k];
    }
    w[0] = (om[1] * p_rel[2] - om[2] * p_rel[1]) + dp_rel[0];
    w[1] = (om[2] * p_rel[0] - om[0] * p_rel[2]) + dp_rel[1];
    w[2] = (om[0] * p_rel[1] - om[1] * p_rel[0]) + dp_rel[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // Rbod = rBody^T
      p_f[k] = (rB[0 * 3 + k] * p_rel[0] + rB[1 * 3 + k] * p_rel[1]) + rB[2 * 3 + k] * p_rel[2];
      dp_f[k] = (rB[0 * 3 + k] * w[0] + rB[1 * 3 + k] * w[1]) + rB[2 * 3 + k] * w[2];
    }
    float trust = 1.f;
    const float phase = fminf(s.contact_phase[(size_t)b * 4 + i], 1.f);
    const float trust_window = 0.2f;
    if (phase < trust_window) trust = phase / trust_window;
    else if (phase > (1.f - trust_window)) trust = (1.f - phase) / trust_window;
    const float factor = 1.f + (1.f - trust) * 100.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Q[6 + 3 * i + k] = factor * Q[6 + 3 * i + k];
      R[3 * i + k] = 1 * R[3 * i + k];
      R[12 + 3 * i + k] = factor * R[12 + 3 * i + k];
      y[3 * i + k] = -p_f[k];
      y[12 + 3 * i + k] = (1.0f - trust) * xh[3 + k] + trust * (-dp_f[k]);
    }
    R[24 + i] = factor * R[24 + i];
    y[24 + i] = (1.0f - trust) * (xh[2] + p_f[2]);
  }
  __syncthreads();
  if (lane < 3) {  // _xhat = _A * _xhat + _B * a  (:170), a = aWorld + (0, 0, -9.81)  (:95)
    const float ak = s.a_world[(size_t)b * 3 + lane] + (lane == 2 ? -9.81f : 0.f);
    const float xp = xh[lane], xv = xh[3 + lane];
    xh[lane] = xp + dt * xv;
    xh[3 + lane] = xv + dt * ak;
  }
  for (int e = lane; e < KF_N * KF_N; e += 64) {  // A P
    const int i = e / KF_N, j = e - KF_N * i;
    AP[e] = (i < 3) ? Pg[e] + dt * Pg[(i + 3) * KF_N + j] : Pg[e];
  }
  __syncthreads();
  for (int e = lane; e < KF_N * KF_N; e += 64) {  // Pm = (A P) A^T + Q  (:172)
    const int i = e / KF_N, j = e - KF_N * i;
    float v = (j < 3) ? AP[e] + dt * AP[e + 3] : AP[e];
    if (i == j) v = v + Q[i];
    Pm[e] = v;
  }
  __syncthreads();
  for (int e = lane; e < KF_M * KF_N; e += 64) {  // C Pm and Pm C^T
    const int r = e / KF_N, j = e - KF_N * r;
    CP[e] = (kf_cb(r) >= 0) ? Pm[kf_ca(r) * KF_N + j] - Pm[kf_cb(r) * KF_N + j] : Pm[kf_ca(r) * KF_N + j];
    const int i = e / KF_M, c = e - KF_M * i;
    K1[e] = (kf_cb(c) >= 0) ? Pm[i * KF_N + kf_ca(c)] - Pm[i * KF_N + kf_cb(c)] : Pm[i * KF_N + kf_ca(c)];
  }
  __syncthreads();
  for (int e = lane; e < KF_M * KF_W; e += 64) {  // [ S | ey | C ],  S = C Pm C^T + R (:177), ey = y - C xhat (:175-176)
    const int r = e / KF_W, c = e - KF_W * r;
    float v;
    if (c < KF_M) {
      v = (kf_cb(c) >= 0) ? CP[r * KF_N + kf_ca(c)] - CP[r * KF_N + kf_cb(c)] : CP[r * KF_N + kf_ca(c)];
      if (r == c) v = v + R[r];
    } else if (c == KF_M) {
      const float ym = (kf_cb(r) >= 0) ? xh[kf_ca(r)] - xh[kf_cb(r)] : xh[kf_ca(r)];
      v = y[r] - ym;
    } else {
      const int j = c - KF_M - 1;
      v = (j == kf_ca(r)) ? 1.f : ((j == kf_cb(r)) ? -1.f : 0.f);
    }
    Sa[e] = v;
  }
  __syncthreads();
  for (int k = 0; k < KF_M; ++k) {  // LU with partial pivoting (first largest wins), every right-hand side carried along
    if (lane == 0) {
      int piv = k;
      float best = fabsf(Sa[k * KF_W + k]);
      for (int r = k + 1; r < KF_M; ++r) {
        const float a = fabsf(Sa[r * KF_W + k]);
        if (a > best) {
          best = a;
          piv = r;
        }
      }
      pivrow = piv;
    }
    __syncthreads();
    const int piv = pivrow;
    if (piv != k && lane < KF_W) {
      const float t = Sa[k * KF_W + lane];
      Sa[k * KF_W + lane] = Sa[piv * KF_W + lane];
      Sa[piv * KF_W + lane] = t;
    }
    __syncthreads();
    const float dkk = Sa[k * KF_W + k];
    const int ncol = KF_W - (k + 1), nel = (KF_M - (k + 1)) * ncol;
    for (int e = lane; e < nel; e += 64) {
      const int r = k + 1 + e / ncol, c = k + 1 + e % ncol;
      const float l = Sa[r * KF_W + k] / dkk;
      Sa[r * KF_W + c] = Sa[r * KF_W + c] - l * Sa[k * KF_W + c];
    }
    __syncthreads();
  }
  if (lane < KF_W - KF_M) {  // back substitution: one right-hand side per lane
    const int c = KF_M + lane;
    for (int r = KF_M - 1; r >= 0; --r) {
      float acc = Sa[r * KF_W + c];
      for (int j = r + 1; j < KF_M; ++j) acc = acc - Sa[r * KF_W + j] * Sa[j * KF_W + c];
      Sa[r * KF_W + c] = acc / Sa[r * KF_W + r];
    }
  }
  __syncthreads();
  if (lane < KF_N) {  // _xhat += Pm C^T S_ey  (:184)
    float acc = 0.f;
    for (int c = 0; c < KF_M; ++c) acc = acc + K1[lane * KF_M + c] * Sa[c * KF_W + KF_M];
    xh[lane] = xh[lane] + acc;
  }
  for (int e = lane; e < KF_N * KF_N; e += 64) {  // T1 = I - Pm C^T S_C
    const int i = e / KF_N, j = e - KF_N * i;
    float acc = 0.f;
    for (int c = 0; c < KF_M; ++c) acc = acc + K1[i * KF_M + c] * Sa[c * KF_W + KF_M + 1 + j];
    AP[e] = ((i == j) ? 1.f : 0.f) - acc;
  }
  __syncthreads();
  for (int e = lane; e < KF_N * KF_N; e += 64) {  // _P = T1 Pm  (:187)
    const int i = e / KF_N, j = e - KF_N * i;
    float acc = 0.f;
    for (int k = 0; k < KF_N; ++k) acc = acc + AP[i * KF_N + k] * Pm[k * KF_N + j];
    CP[e] = acc;  // (C Pm is dead)
  }
  __syncthreads();
  for (int e = lane; e < KF_N * KF_N; e += 64) {  // (_P + _P^T) / 2  (:189-190)
    const int i = e / KF_N, j = e - KF_N * i;
    Pm[e] = (CP[e] + CP[j * KF_N + i]) / 2.f;
  }
  __syncthreads();
  const bool reset = Pm[0] * Pm[KF_N + 1] - Pm[1] * Pm[KF_N] > 0.000001f;  // :192-196
  for (int e = lane; e < KF_N * KF_N; e += 64) {
    const int i = e / KF_N, j = e - KF_N * i;
    float v = Pm[e];
    if (reset) {
      if ((i < 2) != (j < 2)) v = 0.f;
      else if (i < 2 && j < 2) v = v / 10.f;
    }
    Pg[e] = v;
  }
  if (lane < KF_N) s.xhat[(size_t)b * KF_N + lane] = xh[lane];
  if (lane < 3) {
    s.position[(size_t)b * 3 + lane] = xh[lane];
    s.v_world[(size_t)b * 3 + lane] = xh[3 + lane];
    const float* rB = s.r_body + (size_t)b * 9;  // vBody = rBody * vWorld  (:212-214)
    if (s.v_body) s.v_body[(size_t)b * 3 + lane] = (rB[3 * lane] * xh[3] + rB[3 * lane + 1] * xh[4]) + rB[3 * lane + 2] * xh[5];
  }
}

__global__ __launch_bounds__(256) void qmpc_kf_init_kernel(float* __restrict__ xhat, float* __restrict__ P, const int batch) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // setup() :22-24, :52-53: xhat = 0, P = 100 I
  if (t >= batch * KF_N * KF_N) return;
  const int e = t % (KF_N * KF_N), b = t / (KF_N * KF_N);
  P[t] = (e / KF_N == e % KF_N) ? 100.f : 0.f;
  if (e < KF_N) xhat[(size_t)b * KF_N + e] = 0.f;
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

extern "C" hipError_t qmpc_launch_kf(const qmpc_kf_state* st, const float hip[3], int batch, hipStream_t stream) {
  hipLaunchKernelGGL(qmpc_kf_kernel, dim3(batch), dim3(64), 0, stream, *st, hip[0], hip[1], hip[2], batch);
  return hipGetLastError();
}

extern "C" hipError_t qmpc_launch_kf_init(float* xhat, float* P, int batch, hipStream_t stream) {
  const int n = batch * 18 * 18;
  hipLaunchKernelGGL(qmpc_kf_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xhat, P, batch);
  return hipGetLastError();
}
