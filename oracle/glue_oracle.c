/*
 * oracle/glue_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as mpc_oracle.h: only tests/,
 * smoke() and bench.py's cpu_baseline may load it).
 *
 * Plain-C float restatement of the reference's per-tick glue either side of the MPC solve
 * (SURVEY.md 8f-2), one function per reference function, paths relative to /root/reference/src:
 *   Controllers/LegController.cpp:204-244   computeLegJacobianAndPosition
 *   Controllers/LegController.cpp:89-110    LegController::updateData      (v = J qd)
 *   Controllers/LegController.cpp:116-160   LegController::updateCommand
 *   Controllers/LegController.cpp:255-285   computeLegIK
 *   Controllers/FootSwingTrajectory.cpp:17-37 + Utilities/Interpolation.h:27-67
 * PARITY UNPINNED by reference execution like the MPC assembly (these sources need Eigen for
 * Vec3 / Mat3); the arithmetic is scalar float, restated operation by operation.
 * Build with -ffp-contract=off.
 */
#include <math.h>

/* Quadruped::getSideSign, Dynamics/Quadruped.h:85-89 */
static float side_sign(int leg) {
  static const float s[4] = {-1, 1, -1, 1};
  return s[leg];
}

/* geom = {abad, hip, knee, knee_y_offset}; J row-major 3x3 */
void oracle_leg_fk(const float geom[4], int leg, const float q[3], float J[9], float p[3]) {
  const float l1 = geom[0], l2 = geom[1], l3 = geom[2], l4 = geom[3];
  const float sideSign = side_sign(leg);
  const float s1 = sinf(q[0]), s2 = sinf(q[1]), s3 = sinf(q[2]);
  const float c1 = cosf(q[0]), c2 = cosf(q[1]), c3 = cosf(q[2]);
  const float c23 = c2 * c3 - s2 * s3;
  const float s23 = s2 * c3 + c2 * s3;
  J[0] = 0;
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

static float row3(const float* r, const float* x) { return (r[0] * x[0] + r[1] * x[1]) + r[2] * x[2]; }

/* updateData :89-110 for one leg */
void oracle_leg_update(const float geom[4], int leg, const float q[3], const float qd[3], float J[9], float p[3],
                       float v[3]) {
  oracle_leg_fk(geom, leg, q, J, p);
  for (int k = 0; k < 3; k++) v[k] = row3(J + 3 * k, qd);
}

/* computeLegIK :255-285 */
void oracle_leg_ik(const float geom[4], int leg, const float pDes[3], float qDes[3]) {
  const float l1 = geom[0] + geom[3], l2 = geom[1], l3 = geom[2];
  const float sideSign = side_sign(leg);
  float D = (pDes[0] * pDes[0] + pDes[1] * pDes[1] + pDes[2] * pDes[2] - l1 * l1 - l2 * l2 - l3 * l3) / (2 * l2 * l3);
  if (D > 1.00001 || D < -1.00001) {
    if (D > 1.00001) D = 0.99999;
    if (D < -1.00001) D = -0.99999;
  }
  const float gamma = atan2f(-sqrtf(1 - D * D), D);
  const float tetta = -atan2f(pDes[2], pDes[1]) - atan2f(sqrtf(pDes[1] * pDes[1] + pDes[2] * pDes[2] - l1 * l1), sideSign * l1);
  const float alpha = atan2f(-pDes[0], sqrtf(pDes[1] * pDes[1] + pDes[2] * pDes[2] - l1 * l1)) -
                      atan2f(l3 * sinf(gamma), l2 + l3 * cosf(gamma));
  qDes[0] = -tetta;
  qDes[1] = alpha;
  qDes[2] = gamma;
}

/* updateCommand :116-160 for one leg (both branches of the leg == 1 || leg == 3 test are the same
 * arithmetic: "1*crtlParam(2)" is exact) */
void oracle_leg_command(const float geom[4], int leg, const float tauFF[3], const float forceFF[3], const float Kp[9],
                        const float Kd[9], const float pDes[3], const float vDes[3], const float q[3],
                        const float qd[3], const float J[9], const float p[3], const float v[3], float kp_joint,
                        float kd_joint, float tau[3], float qDes[3]) {
  float legTorque[3], footForce[3], dp[3], dv[3], add[3];
  for (int k = 0; k < 3; k++) {
    legTorque[k] = tauFF[k];
    footForce[k] = forceFF[k];
    dp[k] = pDes[k] - p[k];
    dv[k] = vDes[k] - v[k];
  }
  for (int k = 0; k < 3; k++) add[k] = row3(Kp + 3 * k, dp);
  for (int k = 0; k < 3; k++) footForce[k] = footForce[k] + add[k];
  for (int k = 0; k < 3; k++) add[k] = row3(Kd + 3 * k, dv);
  for (int k = 0; k < 3; k++) footForce[k] = footForce[k] + add[k];
  for (int k = 0; k < 3; k++) /* J^T * footForce */
    legTorque[k] = legTorque[k] + ((J[k] * footForce[0] + J[3 + k] * footForce[1]) + J[6 + k] * footForce[2]);
  oracle_leg_ik(geom, leg, pDes, qDes);
  for (int k = 0; k < 3; k++) tau[k] = kp_joint * (0.0f - q[k]) - kd_joint * qd[k] + legTorque[k];
}

/* Interpolation.h:27-67 */
static float bez(float y0, float yf, float x) {
  const float yDiff = yf - y0;
  const float bezier = x * x * x + 3.f * (x * x * (1.f - x));
  return y0 + bezier * yDiff;
}
static float bez_d1(float y0, float yf, float x) {
  const float yDiff = yf - y0;
  const float bezier = 6.f * x * (1.f - x);
  return bezier * yDiff;
}
static float bez_d2(float y0, float yf, float x) {
  const float yDiff = yf - y0;
  const float bezier = 6.f - 12.f * x;
  return bezier * yDiff;
}

/* FootSwingTrajectory::computeSwingTrajectoryBezier :17-37 */
void oracle_swing(const float p0[3], const float pf[3], float height, float phase, float swingTime, float p[3],
                  float v[3], float a[3]) {
  for (int k = 0; k < 3; k++) {
    p[k] = bez(p0[k], pf[k], phase);
    v[k] = bez_d1(p0[k], pf[k], phase) / swingTime;
    a[k] = bez_d2(p0[k], pf[k], phase) / (swingTime * swingTime);
  }
  float zp, zv, za;
  if (phase < 0.5f) {
    zp = bez(p0[2], p0[2] + height, phase * 2);
    zv = bez_d1(p0[2], p0[2] + height, phase * 2) * 2 / swingTime;
    za = bez_d2(p0[2], p0[2] + height, phase * 2) * 4 / (swingTime * swingTime);
  } else {
    zp = bez(p0[2] + height, pf[2], phase * 2 - 1);
    zv = bez_d1(p0[2] + height, pf[2], phase * 2 - 1) * 2 / swingTime;
    za = bez_d2(p0[2] + height, pf[2], phase * 2 - 1) * 4 / (swingTime * swingTime);
  }
  p[2] = zp;
  v[2] = zv;
  a[2] = za;
}

/* batched drivers (so that no Python sits in the loops) */
void oracle_leg_update_batch(const float geom[4], int batch, const float* q, const float* qd, float* J, float* p, float* v) {
  for (int t = 0; t < 4 * batch; t++) oracle_leg_update(geom, t & 3, q + 3 * t, qd + 3 * t, J + 9 * t, p + 3 * t, v + 3 * t);
}
void oracle_leg_command_batch(const float geom[4], int batch, const float* tauFF, const float* forceFF, const float* Kp,
                              const float* Kd, const float* pDes, const float* vDes, const float* q, const float* qd,
                              const float* J, const float* p, const float* v, float kp_joint, float kd_joint, float* tau,
                              float* qDes) {
  for (int t = 0; t < 4 * batch; t++)
    oracle_leg_command(geom, t & 3, tauFF + 3 * t, forceFF + 3 * t, Kp + 9 * t, Kd + 9 * t, pDes + 3 * t, vDes + 3 * t,
                       q + 3 * t, qd + 3 * t, J + 9 * t, p + 3 * t, v + 3 * t, kp_joint, kd_joint, tau + 3 * t, qDes + 3 * t);
}
void oracle_swing_batch(int n, const float* p0, const float* pf, const float* height, const float* phase,
                        const float* swingTime, float* p, float* v, float* a) {
  for (int t = 0; t < n; t++) oracle_swing(p0 + 3 * t, pf + 3 * t, height[t], phase[t], swingTime[t], p + 3 * t, v + 3 * t, a + 3 * t);
}
