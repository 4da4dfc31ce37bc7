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
  /* LegController.cpp:147-154: `crtlParam(2) * (0.0 - q)` carries a double literal, so the term and the sum are double,
   * rounded to float once on assignment; crtlParam(3) * qd is a float product */
  for (int k = 0; k < 3; k++) tau[k] = (float)((double)kp_joint * (0.0 - (double)q[k]) - (double)(kd_joint * qd[k]) + (double)legTorque[k]);
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

/* ------------------------------------------------------------------------------------------------
 * LinearKFPositionVelocityEstimator<float> (Controllers/PositionVelocityEstimator.cpp:18-221): one
 * run() of the 18-state / 28-measurement Kalman filter (body position, velocity, four foot positions;
 * measurements: foot positions and velocities relative to the body from the leg kinematics, and the
 * foot heights), float arithmetic, dense like the reference.  setup() (:18-63): xhat = 0, P = 100 I,
 * dt = 0.002.  The two S.lu().solve(...) calls (:183, :186) are restated as ONE unblocked LU with
 * partial pivoting applied to both right-hand sides (Eigen's PartialPivLU is blocked, so its rounding
 * differs in the last bits; parity with Eigen is unpinned -- Eigen is absent -- like every other
 * Eigen-backed piece).
 */
#define KF_N 18
#define KF_M 28

static float kf_q_process(int k, float dt) { /* run() :73-76 with _Q0 of setup() :56-60 */
  const float process_noise_pimu = 0.02f, process_noise_vimu = 0.02f, process_noise_pfoot = 0.002f;
  if (k < 3) return (dt / 20.f) * process_noise_pimu;
  if (k < 6) return (dt * 9.8f / 20.f) * process_noise_vimu;
  return dt * process_noise_pfoot;
}

/* hip = {abadLocation x, y, z} (Dynamics/MiniCheetah.h:25-26,105: 0.19, 0.049, 0) */
void oracle_kf_step(const float hip[3], float xhat[KF_N], float P[KF_N * KF_N], const float rBody[9],
                    const float aWorld[3], const float omegaBody[3], const float contact[4], const float legp[12],
                    const float legv[12], float position[3], float vWorld[3], float vBody[3]) {
  const float dt = 0.002f;
  const float sensor_noise_pimu_rel_foot = 0.001f, sensor_noise_vimu_rel_foot = 0.1f, sensor_noise_zfoot = 0.001f;
  float Q[KF_N], R[KF_M]; /* both stay diagonal */
  for (int k = 0; k < KF_N; k++) Q[k] = kf_q_process(k, dt);
  for (int k = 0; k < KF_M; k++)
    R[k] = 1.f * (k < 12 ? sensor_noise_pimu_rel_foot : (k < 24 ? sensor_noise_vimu_rel_foot : sensor_noise_zfoot));
  const float a[3] = {aWorld[0] + 0.f, aWorld[1] + 0.f, aWorld[2] + -9.81f}; /* :95 */
  float y[KF_M];
  const float p0[3] = {xhat[0], xhat[1], xhat[2]}, v0[3] = {xhat[3], xhat[4], xhat[5]};
  for (int i = 0; i < 4; i++) { /* :118-166 */
    const float ph[3] = {(i == 0 || i == 1) ? hip[0] : -hip[0], (i == 1 || i == 3) ? hip[1] : -hip[1], hip[2]};
    float p_rel[3], dp_rel[3], w[3], p_f[3], dp_f[3];
    for (int k = 0; k < 3; k++) {
      p_rel[k] = ph[k] + legp[3 * i + k];
      dp_rel[k] = legv[3 * i + k];
    }
    /* omegaBody.cross(p_rel) + dp_rel */
    w[0] = (omegaBody[1] * p_rel[2] - omegaBody[2] * p_rel[1]) + dp_rel[0];
    w[1] = (omegaBody[2] * p_rel[0] - omegaBody[0] * p_rel[2]) + dp_rel[1];
    w[2] = (omegaBody[0] * p_rel[1] - omegaBody[1] * p_rel[0]) + dp_rel[2];
    for (int k = 0; k < 3; k++) { /* Rbod = rBody^T */
      p_f[k] = (rBody[0 * 3 + k] * p_rel[0] + rBody[1 * 3 + k] * p_rel[1]) + rBody[2 * 3 + k] * p_rel[2];
      dp_f[k] = (rBody[0 * 3 + k] * w[0] + rBody[1 * 3 + k] * w[1]) + rBody[2 * 3 + k] * w[2];
    }
    float trust = 1.f;
    const float phase = fminf(contact[i], 1.f);
    const float trust_window = 0.2f;
    if (phase < trust_window) trust = phase / trust_window;
    else if (phase > (1.f - trust_window)) trust = (1.f - phase) / trust_window;
    const float high_suspect_number = 100.f;
    const float factor = 1.f + (1.f - trust) * high_suspect_number;
    for (int k = 0; k < 3; k++) {
      Q[6 + 3 * i + k] = factor * Q[6 + 3 * i + k];
      R[3 * i + k] = 1 * R[3 * i + k];
      R[12 + 3 * i + k] = factor * R[12 + 3 * i + k];
      y[3 * i + k] = -p_f[k];                                        /* _ps */
      y[12 + 3 * i + k] = (1.0f - trust) * v0[k] + trust * (-dp_f[k]); /* _vs */
    }
    R[24 + i] = factor * R[24 + i];
    y[24 + i] = (1.0f - trust) * (p0[2] + p_f[2]); /* pzs */
  }
  /* _xhat = _A * _xhat + _B * a  (:170) */
  for (int k = 0; k < 3; k++) {
    xhat[k] = xhat[k] + dt * xhat[3 + k];
    xhat[3 + k] = xhat[3 + k] + dt * a[k];
  }
  /* Pm = A P A^T + Q (:172): A = [[I, dt I, 0], [0, I, 0], [0, 0, I]] */
  static float AP[KF_N * KF_N], Pm[KF_N * KF_N], CP[KF_M * KF_N], K1[KF_N * KF_M], Saug[KF_M * (KF_M + 1 + KF_N)];
  static float T1[KF_N * KF_N];
  for (int i = 0; i < KF_N; i++)
    for (int j = 0; j < KF_N; j++) AP[i * KF_N + j] = (i < 3) ? P[i * KF_N + j] + dt * P[(i + 3) * KF_N + j] : P[i * KF_N + j];
  for (int i = 0; i < KF_N; i++)
    for (int j = 0; j < KF_N; j++) {
      float v = (j < 3) ? AP[i * KF_N + j] + dt * AP[i * KF_N + j + 3] : AP[i * KF_N + j];
      if (i == j) v = v + Q[i];
      Pm[i * KF_N + j] = v;
    }
  /* C row r: position rows r = 3i+k: e_k - e_{6+3i+k}; velocity rows 12+3i+k: e_{3+k}; height rows 24+i: e_{8+3i} */
#define CROW_A(r) ((r) < 12 ? (r) % 3 : ((r) < 24 ? 3 + (r) % 3 : 8 + 3 * ((r) - 24)))
#define CROW_B(r) ((r) < 12 ? 6 + (r) : -1)
  float ey[KF_M];
  for (int r = 0; r < KF_M; r++) { /* yModel = C xhat; ey = y - yModel (:175-176) */
    const float ym = (CROW_B(r) >= 0) ? xhat[CROW_A(r)] - xhat[CROW_B(r)] : xhat[CROW_A(r)];
    ey[r] = y[r] - ym;
  }
  for (int r = 0; r < KF_M; r++) /* C Pm */
    for (int j = 0; j < KF_N; j++)
      CP[r * KF_N + j] = (CROW_B(r) >= 0) ? Pm[CROW_A(r) * KF_N + j] - Pm[CROW_B(r) * KF_N + j] : Pm[CROW_A(r) * KF_N + j];
  const int W = KF_M + 1 + KF_N; /* [ S | ey | C ] */
  for (int r = 0; r < KF_M; r++) {
    for (int c = 0; c < KF_M; c++) { /* S = C Pm C^T + R (:177) */
      float v = (CROW_B(c) >= 0) ? CP[r * KF_N + CROW_A(c)] - CP[r * KF_N + CROW_B(c)] : CP[r * KF_N + CROW_A(c)];
      if (r == c) v = v + R[r];
      Saug[r * W + c] = v;
    }
    Saug[r * W + KF_M] = ey[r];
    for (int j = 0; j < KF_N; j++) Saug[r * W + KF_M + 1 + j] = (j == CROW_A(r)) ? 1.f : ((j == CROW_B(r)) ? -1.f : 0.f);
  }
  for (int i = 0; i < KF_N; i++) /* Pm C^T */
    for (int c = 0; c < KF_M; c++)
      K1[i * KF_M + c] = (CROW_B(c) >= 0) ? Pm[i * KF_N + CROW_A(c)] - Pm[i * KF_N + CROW_B(c)] : Pm[i * KF_N + CROW_A(c)];
  /* LU with partial pivoting on the augmented matrix, then back substitution (:183, :186) */
  for (int k = 0; k < KF_M; k++) {
    int piv = k;
    float best = fabsf(Saug[k * W + k]);
    for (int r = k + 1; r < KF_M; r++)
      if (fabsf(Saug[r * W + k]) > best) {
        best = fabsf(Saug[r * W + k]);
        piv = r;
      }
    if (piv != k)
      for (int c = 0; c < W; c++) {
        const float t = Saug[k * W + c];
        Saug[k * W + c] = Saug[piv * W + c];
        Saug[piv * W + c] = t;
      }
    for (int r = k + 1; r < KF_M; r++) {
      const float l = Saug[r * W + k] / Saug[k * W + k];
      for (int c = k + 1; c < W; c++) Saug[r * W + c] = Saug[r * W + c] - l * Saug[k * W + c];
    }
  }
  for (int c = KF_M; c < W; c++) /* back substitution, every right-hand side */
    for (int r = KF_M - 1; r >= 0; r--) {
      float acc = Saug[r * W + c];
      for (int j = r + 1; j < KF_M; j++) acc = acc - Saug[r * W + j] * Saug[j * W + c];
      Saug[r * W + c] = acc / Saug[r * W + r];
    }
  /* _xhat += Pm C^T S_ey (:184) */
  for (int i = 0; i < KF_N; i++) {
    float acc = 0.f;
    for (int c = 0; c < KF_M; c++) acc = acc + K1[i * KF_M + c] * Saug[c * W + KF_M];
    xhat[i] = xhat[i] + acc;
  }
  /* _P = (I - Pm C^T S_C) Pm (:187), symmetrised (:189-190) */
  for (int i = 0; i < KF_N; i++)
    for (int j = 0; j < KF_N; j++) {
      float acc = 0.f;
      for (int c = 0; c < KF_M; c++) acc = acc + K1[i * KF_M + c] * Saug[c * W + KF_M + 1 + j];
      T1[i * KF_N + j] = ((i == j) ? 1.f : 0.f) - acc;
    }
  for (int i = 0; i < KF_N; i++)
    for (int j = 0; j < KF_N; j++) {
      float acc = 0.f;
      for (int k = 0; k < KF_N; k++) acc = acc + T1[i * KF_N + k] * Pm[k * KF_N + j];
      AP[i * KF_N + j] = acc;
    }
  for (int i = 0; i < KF_N; i++)
    for (int j = 0; j < KF_N; j++) P[i * KF_N + j] = (AP[i * KF_N + j] + AP[j * KF_N + i]) / 2.f;
  if (P[0] * P[KF_N + 1] - P[1] * P[KF_N] > 0.000001f) { /* :192-196 */
    for (int i = 0; i < 2; i++)
      for (int j = 2; j < KF_N; j++) {
        P[i * KF_N + j] = 0.f;
        P[j * KF_N + i] = 0.f;
      }
    P[0] = P[0] / 10.f;
    P[1] = P[1] / 10.f;
    P[KF_N] = P[KF_N] / 10.f;
    P[KF_N + 1] = P[KF_N + 1] / 10.f;
  }
  for (int k = 0; k < 3; k++) {
    position[k] = xhat[k];
    vWorld[k] = xhat[3 + k];
  }
  for (int k = 0; k < 3; k++) /* vBody = rBody * vWorld (:212-214) */
    vBody[k] = (rBody[3 * k] * xhat[3] + rBody[3 * k + 1] * xhat[4]) + rBody[3 * k + 2] * xhat[5];
}

void oracle_kf_step_batch(const float hip[3], int batch, float* xhat, float* P, const float* rBody, const float* aWorld,
                          const float* omegaBody, const float* contact, const float* legp, const float* legv,
                          float* position, float* vWorld, float* vBody) {
  for (int b = 0; b < batch; b++)
    oracle_kf_step(hip, xhat + 18 * b, P + 324 * b, rBody + 9 * b, aWorld + 3 * b, omegaBody + 3 * b, contact + 4 * b,
                   legp + 12 * b, legv + 12 * b, position + 3 * b, vWorld + 3 * b, vBody + 3 * b);
}
