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

#include "qmpc_cmd.h"

namespace {

__global__ __launch_bounds__(256) void qmpc_pack_kernel(const qmpc_command c, const qmpc_record rec, const int batch,
                                                        const int h, const float dt_mpc) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= batch) return;  // wave-uniform
  const float* pos = c.position + (size_t)b * 3;
  const float p0 = pos[0], p1 = pos[1], p2 = pos[2];
  const bool stand = c.gait_type && c.gait_type[b] == 4;  // :514
  const float yaw_rate = c.vel_des[(size_t)b * 3 + 2];
  float vw0, vw1;
  qmpc_cmd_vdes_world(c.r_body + (size_t)b * 9, c.vel_des[(size_t)b * 3 + 0], c.vel_des[(size_t)b * 3 + 1], c.omni_mode,
                      vw0, vw1);
  QmpcTrajGen g;
  qmpc_cmd_traj_gen(g, stand, stand ? c.stand_traj + (size_t)b * 6 : nullptr, c.rp_des ? c.rp_des + (size_t)b * 2 : nullptr,
                    c.rpy_comp + (size_t)b * 2, c.yaw_des_true[b], c.world_position_desired[(size_t)b * 2 + 0],
                    c.world_position_desired[(size_t)b * 2 + 1], p0, p1, c.body_height, yaw_rate, vw0, vw1, dt_mpc);
  __builtin_amdgcn_wave_barrier();  // every lane has read the old desired position
  if (lane == 0 && !stand) {
    c.world_position_desired[(size_t)b * 2 + 0] = g.xs;  // :544-545
    c.world_position_desired[(size_t)b * 2 + 1] = g.ys;
  }
  for (int idx = lane; idx < 12 * h; idx += 64) {
    const int k = idx / 12, j = idx - 12 * k;
    rec.traj[(size_t)b * 12 * h + idx] = qmpc_cmd_traj_value(g, k, j);
  }
  for (int fs = lane; fs < 4 * h; fs += 64) {  // (more than 64 foot-steps beyond horizon 16)
    const int leg = fs & 3;
    rec.gait[(size_t)b * 4 * h + fs] = (uint8_t)qmpc_cmd_gait_bit(fs >> 2, c.gait_iteration[b], c.gait_offsets[(size_t)b * 4 + leg],
                                                                   c.gait_durations[(size_t)b * 4 + leg], h);
  }
  if (lane < 12) {
    const int leg = lane & 3, ax = lane >> 2;  // axis-major (:611-613)
    rec.r[(size_t)b * 12 + lane] = qmpc_cmd_foot_offset(c.p_foot[(size_t)b * 12 + 3 * leg + ax], ax == 0 ? p0 : (ax == 1 ? p1 : p2));
    if (rec.weights) rec.weights[(size_t)b * 12 + lane] = qmpc_cmd_weight(lane);
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
    c.x_comp_integral[b] = qmpc_cmd_xci_next(xci, p2, c.body_height, dt_mpc, c.v_world[(size_t)b * 3 + 0]);
  }
}

// f_ff[leg] = -rBody * f   (:672-680); one thread per (robot, leg, axis)
__global__ __launch_bounds__(256) void qmpc_f2b_kernel(const float* __restrict__ r_body, const float* __restrict__ grf,
                                                       float* __restrict__ f_ff, const int batch) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= batch * 12) return;
  const int b = t / 12, e = t - 12 * b, leg = e / 3, i = e - 3 * leg;
  const float* f = grf + (size_t)b * 12 + 3 * leg;
  f_ff[t] = qmpc_cmd_f2b(r_body + (size_t)b * 9 + 3 * i, f[0], f[1], f[2]);
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
