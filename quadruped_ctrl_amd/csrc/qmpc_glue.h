// qmpc_glue.h -- float arithmetic of the per-tick glue either side of the MPC solve
// (SURVEY.md 8f-2), written operation by operation after the reference:
//   leg forward kinematics + Jacobian   src/Controllers/LegController.cpp:204-244, :89-110
//   leg command (Cartesian PD, J^T f)   src/Controllers/LegController.cpp:116-160
//   leg inverse kinematics              src/Controllers/LegController.cpp:255-285
//   swing-foot Bezier trajectory        src/Controllers/FootSwingTrajectory.cpp:17-37,
//                                       src/Utilities/Interpolation.h:27-67
// fp contraction is off inside every body (the reference's host code has no fma), so everything
// except the libm calls (sin / cos / atan2 / sqrt) rounds exactly like the reference's float code.
#ifndef QMPC_GLUE_H
#define QMPC_GLUE_H

#include <hip/hip_runtime.h>

struct QmpcLegGeom {
  float abad, hip, knee, knee_y;  // _abadLinkLength, _hipLinkLength, _kneeLinkLength, _kneeLinkY_offset
};

// Quadruped::getSideSign (src/Dynamics/Quadruped.h:85-89)
__device__ __forceinline__ float qmpc_side_sign(int leg) { return (leg & 1) ? 1.f : -1.f; }

// computeLegJacobianAndPosition (:204-244).  J row-major 3x3.
__device__ __forceinline__ void qmpc_leg_fk(const QmpcLegGeom& g, int leg, float q0, float q1, float q2, float* J,
                                            float* p) {
#pragma clang fp contract(off)
  const float l1 = g.abad, l2 = g.hip, l3 = g.knee, l4 = g.knee_y;
  const float sideSign = qmpc_side_sign(leg);
  const float s1 = sinf(q0), s2 = sinf(q1), s3 = sinf(q2);
  const float c1 = cosf(q0), c2 = cosf(q1), c3 = cosf(q2);
  const float c23 = c2 * c3 - s2 * s3;
  const float s23 = s2 * c3 + c2 * s3;
  J[0] = 0.f;
  J[1] = l3 * c23 + l2 * c2;
  J[2] = l3 * c23;
  J[3] = l3 * c1 * c23 + l2 * c1 * c2 - (l1 + l4) * sideSign * s1;
  J[4] = -l3 * s1 * s23 - l2 * s1 * s2;
  J[5] = -l3 * s1 * s23;
  J[6] = l3 * s1 * c23 + l2 * c2 * s1 + (l1 + l4) * sideSign * c1;
  J[7] = l3 * c1 * s23 + l2 * c1 * s2;
  J[8] = l3 * c1 * s23;
  p[0] = l3 * s23 + l2 * s2;
  p[1] = (l1 + l4) * sideSign * c1 + l3 * (s1 * c23) + l2 * c2 * s1;
  p[2] = (l1 + l4) * sideSign * s1 - l3 * (c1 * c23) - l2 * c1 * c2;
}

// 3x3 row-major times 3-vector, accumulated left to right like Eigen's fixed-size product
__device__ __forceinline__ float qmpc_row3(const float* r, float x0, float x1, float x2) {
#pragma clang fp contract(off)
  return (r[0] * x0 + r[1] * x1) + r[2] * x2;
}

// computeLegIK (:255-285)
__device__ __forceinline__ void qmpc_leg_ik(const QmpcLegGeom& g, int leg, float px, float py, float pz, float* qdes) {
#pragma clang fp contract(off)
  const float l1 = g.abad + g.knee_y, l2 = g.hip, l3 = g.knee;
  const float sideSign = qmpc_side_sign(leg);
  float D = (px * px + py * py + pz * pz - l1 * l1 - l2 * l2 - l3 * l3) / (2 * l2 * l3);
  // ("D > 1.00001": double literals compared with the float widened)
  if ((double)D > 1.00001) D = (float)0.99999;
  if ((double)D < -1.00001) D = (float)-0.99999;
  const float gamma = atan2f(-sqrtf(1 - D * D), D);
  const float rad = sqrtf(py * py + pz * pz - l1 * l1);
  const float tetta = -atan2f(pz, py) - atan2f(rad, sideSign * l1);
  const float alpha = atan2f(-px, rad) - atan2f(l3 * sinf(gamma), l2 + l3 * cosf(gamma));
  qdes[0] = -tetta;
  qdes[1] = alpha;
  qdes[2] = gamma;
}

// Interpolate::cubicBezier and derivatives (Interpolation.h:27-67)
__device__ __forceinline__ float qmpc_bez(float y0, float yf, float x) {
#pragma clang fp contract(off)
  const float yDiff = yf - y0;
  const float bezier = x * x * x + 3.f * (x * x * (1.f - x));
  return y0 + bezier * yDiff;
}
__device__ __forceinline__ float qmpc_bez_d1(float y0, float yf, float x) {
#pragma clang fp contract(off)
  const float yDiff = yf - y0;
  const float bezier = 6.f * x * (1.f - x);
  return bezier * yDiff;
}
__device__ __forceinline__ float qmpc_bez_d2(float y0, float yf, float x) {
#pragma clang fp contract(off)
  const float yDiff = yf - y0;
  const float bezier = 6.f - 12.f * x;
  return bezier * yDiff;
}

// FootSwingTrajectory::computeSwingTrajectoryBezier (:17-37) for one axis
__device__ __forceinline__ void qmpc_swing_axis(int axis, float p0, float pf, float p0z, float pfz, float height,
                                                float phase, float swingTime, float& p, float& v, float& a) {
#pragma clang fp contract(off)
  if (axis < 2) {
    p = qmpc_bez(p0, pf, phase);
    v = qmpc_bez_d1(p0, pf, phase) / swingTime;
    a = qmpc_bez_d2(p0, pf, phase) / (swingTime * swingTime);
  } else if (phase < 0.5f) {
    p = qmpc_bez(p0z, p0z + height, phase * 2);
    v = qmpc_bez_d1(p0z, p0z + height, phase * 2) * 2 / swingTime;
    a = qmpc_bez_d2(p0z, p0z + height, phase * 2) * 4 / (swingTime * swingTime);
  } else {
    p = qmpc_bez(p0z + height, pfz, phase * 2 - 1);
    v = qmpc_bez_d1(p0z + height, pfz, phase * 2 - 1) * 2 / swingTime;
    a = qmpc_bez_d2(p0z + height, pfz, phase * 2 - 1) * 4 / (swingTime * swingTime);
  }
}

#endif
