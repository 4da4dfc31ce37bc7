// qmpc_pack.hip -- the caller side of the solve, batched on the GPU:
//   qmpc_pack_kernel       what ConvexMPCLocomotion::updateMPCIfNeeded
//                          (src/MPC_Ctrl/ConvexMPCLocomotion.cpp:498-577) and
//                          ::solveDenseMPC (:592-640) do before
//                          update_problem_data_floats: reference trajectory,
//                          desired-position clamp, foot offsets, drag integrator,
//                          weights, and the contact table of
//                          OffsetDurationGait::getMpcTable (Gait.cpp:142-166)
//   qmpc_f2b_kernel        what solveDenseMPC does after get_solution (:672-680):
//                          f_ff[leg] = -rBody * f
// One wave per robot (four robots per workgroup); every array one row per
// robot.  The float arithmetic is written operation by operation
// (fp contraction off: no fma), so the record equals the
// reference's float code bit for bit; the only HBM traffic is the command row
// in (~250 B) and the record row out (728 B at h = 10).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qmpc.h"

// hipcc contracts a*b+c into fma by default; the reference's host code does not
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float fmul(float a, float b) { return a * b; }
__device__ __forceinline__ float fadd(float a, float b) { return a + b; }

__global__ __launch_bounds__(256) void qmpc_pack_kernel(const qmpc_command c, const qmpc_record rec, const int batch,
                                                        const int h, const float dt_mpc) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= batch) return;  // wave-uniform
  const float* pos = c.position + (size_t)b * 3;
  const float p0 = pos[0], p1 = pos[1], p2 = pos[2];
  const bool stand = c.gait_type && c.gait_type[b] == 4;  // :514
  // ---- :505-507  v_des_world = omniMode ? v_des_robot : rBody^T v_des_robot
  const float vx_r = c.vel_des[(size_t)b * 3 + 0], vy_r = c.vel_des[(size_t)b * 3 + 1];
  const float yaw_rate = c.vel_des[(size_t)b * 3 + 2];
  float vw0 = vx_r, vw1 = vy_r;
  if (!c.omni_mode) {
    const float* R = c.r_body + (size_t)b * 9;
    vw0 = fadd(fadd(fmul(R[0], vx_r), fmul(R[3], vy_r)), fmul(R[6], 0.f));
    vw1 = fadd(fadd(fmul(R[1], vx_r), fmul(R[4], vy_r)), fmul(R[7], 0.f));
  }
  // ---- trajInitial and the per-step increments of rows 2, 3, 4
  float init[12];
  float inc2 = 0.f, inc3 = 0.f, inc4 = 0.f;
  if (stand) {  // :514-531
    const float* st = c.stand_traj + (size_t)b * 6;
    init[0] = c.rp_des ? c.rp_des[(size_t)b * 2 + 0] : 0.f;
    init[1] = c.rp_des ? c.rp_des[(size_t)b * 2 + 1] : 0.f;
    init[2] = st[5];
    init[3] = st[0];
    init[4] = st[1];
    init[5] = c.body_height;
#pragma unroll
    for (int j = 6; j < 12; ++j) init[j] = 0.f;
  } else {  // :534-561
    const float max_pos_error = .1f;
    float xs = c.world_position_desired[(size_t)b * 2 + 0], ys = c.world_position_desired[(size_t)b * 2 + 1];
    // "p[0] + 0.1": double literal, result stored to float (:538-542)
    if (fadd(xs, -p0) > max_pos_error) xs = (float)((double)p0 + 0.1);
    if (fadd(p0, -xs) > max_pos_error) xs = (float)((double)p0 - 0.1);
    if (fadd(ys, -p1) > max_pos_error) ys = (float)((double)p1 + 0.1);
    if (fadd(p1, -ys) > max_pos_error) ys = (float)((double)p1 - 0.1);
    __builtin_amdgcn_wave_barrier();  // every lane has read the old value
    if (lane == 0) {
      c.world_position_desired[(size_t)b * 2 + 0] = xs;  // :544-545
      c.world_position_desired[(size_t)b * 2 + 1] = ys;
    }
    init[0] = c.rpy_comp[(size_t)b * 2 + 0];
    init[1] = c.rpy_comp[(size_t)b * 2 + 1];
    init[2] = c.yaw_des_true[b];
    init[3] = xs;
    init[4] = ys;
    init[5] = c.body_height;
    init[6] = 0.f;
    init[7] = 0.f;
    init[8] = yaw_rate;
    init[9] = vw0;
    init[10] = vw1;
    init[11] = 0.f;
    inc2 = fmul(dt_mpc, yaw_rate);  // :566-573
    inc3 = fmul(dt_mpc, vw0);
    inc4 = fmul(dt_mpc, vw1);
  }
  // ---- trajAll (:563-576): rows 2,3,4 are running float sums, step by step
  for (int idx = lane; idx < 12 * h; idx += 64) {
    const int k = idx / 12, j = idx - 12 * k;
    float val = init[0];
#pragma unroll
    for (int q = 1; q < 12; ++q) val = (j == q) ? init[q] : val;
    const float inc = (j == 2) ? inc2 : (j == 3 ? inc3 : inc4);
    if (!stand && j >= 2 && j <= 4)
      for (int s = 0; s < k; ++s) val = fadd(val, inc);
    rec.traj[(size_t)b * 12 * h + idx] = val;
  }
  // ---- contact table, Gait.cpp:142-166 with _nIterations = horizon
  if (lane < 4 * h) {
    const int i = lane >> 2, leg = lane & 3;
    const int iter = (i + c.gait_iteration[b] + 1) % h;
    int progress = iter - c.gait_offsets[(size_t)b * 4 + leg];
    if (progress < 0) progress += h;
    rec.gait[(size_t)b * 4 * h + lane] = (progress < c.gait_durations[(size_t)b * 4 + leg]) ? 1 : 0;
  }
  // ---- solveDenseMPC :598-613
  if (lane < 12) {
    // r[i] = pFoot[i%4][i/4] - position[i/4]   (axis-major, :611-613)
    const int leg = lane & 3, ax = lane >> 2;
    rec.r[(size_t)b * 12 + lane] = fadd(c.p_foot[(size_t)b * 12 + 3 * leg + ax], -(ax == 0 ? p0 : (ax == 1 ? p1 : p2)));
    if (rec.weights) {
      const float Q[12] = {2.5f, 2.5f, 10.f, 50.f, 50.f, 100.f, 0.f, 0.f, 0.5f, 0.2f, 0.2f, 0.1f};  // :598
      float qv = Q[0];
#pragma unroll
      for (int q = 1; q < 12; ++q) qv = (lane == q) ? Q[q] : qv;
      rec.weights[(size_t)b * 12 + lane] = qv;
    }
  }
  if (lane < 3) {
    rec.p[(size_t)b * 3 + lane] = pos[lane];
    rec.v[(size_t)b * 3 + lane] = c.v_world[(size_t)b * 3 + lane];
    rec.w[(size_t)b * 3 + lane] = c.omega_world[(size_t)b * 3 + lane];
  }
  if (lane < 4) rec.q[(size_t)b * 4 + lane] = c.orientation[(size_t)b * 4 + lane];
  if (lane == 0) {
    rec.yaw[b] = c.rpy[(size_t)b * 3 + 2];  // :602
    if (rec.alpha) rec.alpha[b] = 4e-5f;    // :604
    // update_x_drag(x_comp_integral) (:632) comes BEFORE the integrator step (:636-640)
    const float xci = c.x_comp_integral[b];
    rec.x_drag[b] = xci;
    const float pz_err = fadd(p2, -c.body_height);  // :625
    const float vx = c.v_world[(size_t)b * 3 + 0];
    if ((double)vx > 0.3 || (double)vx < -0.3)
      c.x_comp_integral[b] = fadd(xci, __fdiv_rn(fmul(fmul(3.0f, pz_err), dt_mpc), vx));  // correctly rounded division
  }
}

// f_ff[leg] = -rBody * f   (:672-680); one thread per (robot, leg, axis)
__global__ __launch_bounds__(256) void qmpc_f2b_kernel(const float* __restrict__ r_body, const float* __restrict__ grf,
                                                       float* __restrict__ f_ff, const int batch) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= batch * 12) return;
  const int b = t / 12, e = t - 12 * b, leg = e / 3, i = e - 3 * leg;
  const float* R = r_body + (size_t)b * 9 + 3 * i;
  const float* f = grf + (size_t)b * 12 + 3 * leg;
  f_ff[t] = fadd(fadd(fmul(-R[0], f[0]), fmul(-R[1], f[1])), fmul(-R[2], f[2]));
}

}  // namespace

extern "C" hipError_t qmpc_launch_pack(const qmpc_command* c, const qmpc_record* rec, int batch, int horizon, float dt_mpc,
                                       hipStream_t stream) {
  hipLaunchKernelGGL(qmpc_pack_kernel, dim3((batch + 3) / 4), dim3(256), 0, stream, *c, *rec, batch, horizon, dt_mpc);
  return hipGetLastError();
}

extern "C" hipError_t qmpc_launch_f2b(const float* r_body, const float* grf, float* f_ff, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(qmpc_f2b_kernel, dim3((batch * 12 + 255) / 256), dim3(256), 0, stream, r_body, grf, f_ff, batch);
  return hipGetLastError();
}
