/*
 * oracle/mpc_oracle.c -- TEST INFRASTRUCTURE ONLY (see mpc_oracle.h header).
 *
 * Plain-C, float-precision restatement of the reference's dense convex-MPC
 * assembly.  Deliberately written the way the reference computes it (dense
 * 13h x 12h B_qp, dense S, dense fmat, O(m*n) elimination scan) so that it
 * is an independent check on the structure-exploiting HIP kernels.
 *
 * PARITY: solver pinned by the real qpOASES (oracle/_ref); assembly UNPINNED
 * by reference execution (Eigen absent) -- see mpc_oracle.h.
 *
 * Build with -ffp-contract=off so float products/sums round individually as
 * an SSE2 Eigen build would.
 */
#include "mpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BIG_NUMBER 5e10 /* SolverMPC.cpp:15 */

/* SolverMPC.cpp:64-72 */
static int near_zero(float a) { return (a < 0.01 && a > -.01); }
static int near_one(float a) { return near_zero(a - 1); }

/* SolverMPC.cpp:257-267 */
void oracle_quat_to_rpy(const float q[4], float rpy[3]) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  /* t_min(-2.*(...), .99999): evaluated in double, stored to float (fpt as) */
  double asd = -2. * (double)(x * z - w * y);
  if (!(asd < .99999)) asd = .99999;
  const float as = (float)asd;
  rpy[0] = atan2f(2.f * (x * y + w * z), w * w + x * x - y * y - z * z);
  rpy[1] = asinf(as);
  rpy[2] = atan2f(2.f * (y * z + w * x), w * w - x * x - y * y + z * z);
}

/* 3x3 helpers, float, row-major */
static void mat3_mul(const float* a, const float* b, float* c) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = 0.f;
      for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
      c[3 * i + j] = s;
    }
}
static void mat3_T(const float* a, float* t) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * j + i] = a[3 * i + j];
}
/* cofactor inverse (what Eigen's fixed-size 3x3 inverse() evaluates) */
static void mat3_inv(const float* a, float* inv) {
  const float c00 = a[4] * a[8] - a[5] * a[7];
  const float c01 = a[5] * a[6] - a[3] * a[8];
  const float c02 = a[3] * a[7] - a[4] * a[6];
  const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const float id = 1.f / det;
  inv[0] = c00 * id;
  inv[1] = (a[2] * a[7] - a[1] * a[8]) * id;
  inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  inv[3] = c01 * id;
  inv[4] = (a[0] * a[8] - a[2] * a[6]) * id;
  inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  inv[6] = c02 * id;
  inv[7] = (a[1] * a[6] - a[0] * a[7]) * id;
  inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

/* SolverMPC.cpp:235-254, :226-233, :319; RobotState.cpp:25-40 */
void oracle_ct_ss_mats(const float r[12], float yaw, float x_drag, float* A,
                       float* B) {
  /* RobotState.cpp:30-35 */
  const float yc = cosf(yaw), ys = sinf(yaw);
  const float R_yaw[9] = {yc, -ys, 0, ys, yc, 0, 0, 0, 1};
  /* RobotState.cpp:37-40 */
  const float I_body[9] = {.07f, 0, 0, 0, 0.26f, 0, 0, 0, 0.242f};
  const float m = 9; /* RobotState.h:23 */
  float R_yawT[9], tmp[9], I_world[9], I_inv[9];
  mat3_T(R_yaw, R_yawT);
  /* SolverMPC.cpp:319  I_world = R_yaw * I_body * R_yaw^T */
  mat3_mul(R_yaw, I_body, tmp);
  mat3_mul(tmp, R_yawT, I_world);

  memset(A, 0, sizeof(float) * 13 * 13);
  A[3 * 13 + 9] = 1.f;
  A[11 * 13 + 9] = x_drag;
  A[4 * 13 + 10] = 1.f;
  A[5 * 13 + 11] = 1.f;
  A[11 * 13 + 12] = 1.f;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i * 13 + 6 + j] = R_yawT[3 * i + j];

  memset(B, 0, sizeof(float) * 13 * 12);
  mat3_inv(I_world, I_inv);
  for (int b = 0; b < 4; b++) {
    /* RobotState.cpp:25-27  r_feet(row, foot) = r[row*4 + foot] */
    const float rx = r[0 * 4 + b], ry = r[1 * 4 + b], rz = r[2 * 4 + b];
    const float cm[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float blk[9];
    mat3_mul(I_inv, cm, blk);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        B[(6 + i) * 12 + b * 3 + j] = blk[3 * i + j];
        B[(9 + i) * 12 + b * 3 + j] = (i == j) ? 1.f / m : 0.f;
      }
  }
}

/* dense float matmul C(rxc) = A(rxk) B(kxc), row-major */
static void matmul_f(const float* A, const float* B, float* C, int r, int k,
                     int c) {
  /* i-l-j loop order: every C[i][j] still accumulates its k products in
   * sequence l = 0..k-1 (identical rounding to the textbook dot-product
   * loop), but the inner loop is unit-stride so gcc can vectorise it. */
  for (int i = 0; i < r; i++) {
    float* Ci = C + (size_t)i * c;
    for (int j = 0; j < c; j++) Ci[j] = 0.f;
    for (int l = 0; l < k; l++) {
      const float a = A[i * k + l];
      const float* Bl = B + (size_t)l * c;
      for (int j = 0; j < c; j++) Ci[j] += a * Bl[j];
    }
  }
}

/* SolverMPC.cpp:89-95.  The reference evaluates Eigen's float Pade expm of
 * M = dt*[[A,B],[0,0]] (25x25).  For this model A^3 = 0, hence M^4 = 0 and
 * exp(M) = I + M + M^2/2 + M^3/6 exactly; evaluated here in float.  Both are
 * exp(M) to float roundoff (tests cross-check against scipy.linalg.expm). */
void oracle_c2d(const float* A, const float* B, float dt, float* Adt,
                float* Bdt) {
  enum { N = 25 };
  float M[N * N], M2[N * N], M3[N * N], E[N * N]; /* (automatic: the restatement is re-entrant) */
  memset(M, 0, sizeof(M));
  for (int i = 0; i < 13; i++) {
    for (int j = 0; j < 13; j++) M[i * N + j] = dt * A[i * 13 + j];
    for (int j = 0; j < 12; j++) M[i * N + 13 + j] = dt * B[i * 12 + j];
  }
  matmul_f(M, M, M2, N, N, N);
  matmul_f(M2, M, M3, N, N, N);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      float e = (i == j) ? 1.f : 0.f;
      e += M[i * N + j];
      e += M2[i * N + j] / 2.f;
      e += M3[i * N + j] / 6.f;
      E[i * N + j] = e;
    }
  for (int i = 0; i < 13; i++) {
    for (int j = 0; j < 13; j++) Adt[i * 13 + j] = E[i * N + j];
    for (int j = 0; j < 12; j++) Bdt[i * 12 + j] = E[i * N + 13 + j];
  }
}


/* ---- evaluation order of the float condensation (SolverMPC.cpp:395, :399).
 * The reference writes qH = 2*(B_qp^T * S * B_qp + alpha*I) and leaves the
 * order of the float operations to Eigen (un-vendored, unpinned): which
 * product is formed first, how the length-13h inner sums are blocked into SIMD
 * partial sums, and whether multiply-adds are fused all depend on the Eigen
 * version and the compiler flags.  Mode 0 is the restatement's default; the
 * other modes are equally legitimate evaluations of the same expression and
 * are used ONLY to measure the reference pipeline's own float noise floor.
 *   0  B^T (S B), sequential sum over k               (default)
 *   1  (B^T S) B, sequential sum over k     -- left-to-right as written
 *   2  B^T (S B), 4 interleaved partial sums  (SSE packet reduction)
 *   3  B^T (S B), 8 interleaved partial sums  (AVX)
 *   4  B^T (S B), sequential, fused multiply-add (-mfma builds)
 *   5  (B^T S) B, 16 interleaved partial sums, fused (AVX-512 + fma)        */
static int g_accum_mode = 0;
void oracle_set_accum_mode(int mode) { g_accum_mode = mode; }
int oracle_get_accum_mode(void) { return g_accum_mode; }

/* sum_k a[k*sa] * s[k] * b[k*sb] in the selected order */
static float accum_dot(const float* a, int sa, const float* b, int sb,
                       const float* s, int len) {
  const int mode = g_accum_mode;
  const int lanes = (mode == 2) ? 4 : (mode == 3 ? 8 : (mode == 5 ? 16 : 1));
  const int left = (mode == 1 || mode == 5), fused = (mode == 4 || mode == 5);
  float part[16];
  for (int l = 0; l < lanes; l++) part[l] = 0.f;
  for (int k = 0; k < len; k++) {
    const float av = a[(size_t)k * sa], bv = b[(size_t)k * sb];
    float x, y;
    if (left) {
      x = av * s[k];
      y = bv;
    } else {
      x = av;
      y = s[k] * bv;
    }
    float* p = &part[k % lanes];
    *p = fused ? fmaf(x, y, *p) : (*p + x * y);
  }
  /* pairwise reduction of the partial sums, like a SIMD horizontal add */
  for (int w = lanes / 2; w >= 1; w /= 2)
    for (int l = 0; l < w; l++) part[l] = part[l] + part[l + w];
  return part[0];
}

/* SolverMPC.cpp:298-399, :423-429 */
void oracle_assemble(const oracle_update_t* u, const oracle_setup_t* s,
                     double* H, double* g, double* Acon, double* lb,
                     double* ub, float* x0_out) {
  const int h = s->horizon;
  const int n = 12 * h, m = 20 * h, ns = 13 * h;

  /* :315-318 */
  float rpy[3], x0[13];
  oracle_quat_to_rpy(u->q, rpy);
  x0[0] = rpy[2];
  x0[1] = rpy[1];
  x0[2] = rpy[0];
  for (int i = 0; i < 3; i++) {
    x0[3 + i] = u->p[i];
    x0[6 + i] = u->w[i];
    x0[9 + i] = u->v[i];
  }
  x0[12] = -9.8f;
  if (x0_out) memcpy(x0_out, x0, sizeof(x0));

  /* :319-322 */
  float A_ct[13 * 13], B_ct[13 * 12], Adt[13 * 13], Bdt[13 * 12];
  oracle_ct_ss_mats(u->r, u->yaw, u->x_drag, A_ct, B_ct);

  /* c2qp :87-125 */
  oracle_c2d(A_ct, B_ct, s->dt, Adt, Bdt);
  float* pw = (float*)calloc((size_t)(h + 1) * 169, sizeof(float));
  for (int i = 0; i < 13; i++) pw[i * 13 + i] = 1.f;
  for (int i = 1; i <= h; i++) /* powerMats[i] = Adt * powerMats[i-1] */
    matmul_f(Adt, pw + (i - 1) * 169, pw + i * 169, 13, 13, 13);
  float* A_qp = (float*)calloc((size_t)ns * 13, sizeof(float));
  float* B_qp = (float*)calloc((size_t)ns * n, sizeof(float));
  float blk[13 * 12];
  for (int r = 0; r < h; r++) {
    memcpy(A_qp + (size_t)13 * r * 13, pw + (r + 1) * 169,
           169 * sizeof(float));
    for (int c = 0; c <= r; c++) {
      matmul_f(pw + (r - c) * 169, Bdt, blk, 13, 13, 12);
      for (int i = 0; i < 13; i++)
        for (int j = 0; j < 12; j++)
          B_qp[(size_t)(13 * r + i) * n + 12 * c + j] = blk[i * 12 + j];
    }
  }

  /* :335-346  S diagonal (dense in the reference), X_d */
  float* Sd = (float*)calloc(ns, sizeof(float));
  float* X_d = (float*)calloc(ns, sizeof(float));
  for (int i = 0; i < h; i++)
    for (int j = 0; j < 12; j++) {
      Sd[13 * i + j] = u->weights[j];
      X_d[13 * i + j] = u->traj[12 * i + j];
    }

  /* :395  qH = 2*(B^T S B + alpha I)   (float) */
  float* SB = (float*)malloc(sizeof(float) * (size_t)ns * n);
  for (int i = 0; i < ns; i++)
    for (int j = 0; j < n; j++)
      SB[(size_t)i * n + j] = Sd[i] * B_qp[(size_t)i * n + j];
  if (g_accum_mode == 0) {
    /* k-outer order: per element the same sequential sum over k, unit-stride
     * inner loop (see matmul_f). */
    float* Hf = (float*)calloc((size_t)n * n, sizeof(float));
    for (int k = 0; k < ns; k++) {
      const float* Bk = B_qp + (size_t)k * n;
      const float* SBk = SB + (size_t)k * n;
      for (int i = 0; i < n; i++) {
        const float bki = Bk[i];
        float* Hi = Hf + (size_t)i * n;
        for (int j = 0; j < n; j++) Hi[j] += bki * SBk[j];
      }
    }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        float acc = Hf[(size_t)i * n + j];
        if (i == j) acc += u->alpha;
        H[(size_t)i * n + j] = (double)(2.f * acc); /* :423 matrix_to_real */
      }
    free(Hf);
  } else {
    /* alternative evaluation orders of the same float expression (noise-floor
     * measurement, tests/golden/make_noise_floor.py) */
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        float acc = accum_dot(B_qp + i, n, B_qp + j, n, Sd, ns);
        if (i == j) acc += u->alpha;
        H[(size_t)i * n + j] = (double)(2.f * acc);
      }
  }
  /* :399  qg = 2 B^T S (A_qp x0 - X_d) */
  float* t = (float*)malloc(sizeof(float) * ns);
  for (int i = 0; i < ns; i++) {
    float acc = 0.f;
    for (int k = 0; k < 13; k++) acc += A_qp[i * 13 + k] * x0[k];
    t[i] = (acc - X_d[i]);
    if (g_accum_mode == 0) t[i] = Sd[i] * t[i];
  }
  for (int j = 0; j < n; j++) {
    float acc = 0.f;
    if (g_accum_mode == 0)
      for (int k = 0; k < ns; k++) acc += B_qp[(size_t)k * n + j] * t[k];
    else
      acc = accum_dot(B_qp + j, n, t, 1, Sd, ns);
    g[j] = (double)(2.f * acc);
  }

  /* :352-364  U_b ; :428 lb = 0 */
  int k = 0;
  for (int i = 0; i < h; i++)
    for (int j = 0; j < 4; j++) {
      ub[5 * k + 0] = (double)(float)BIG_NUMBER;
      ub[5 * k + 1] = (double)(float)BIG_NUMBER;
      ub[5 * k + 2] = (double)(float)BIG_NUMBER;
      ub[5 * k + 3] = (double)(float)BIG_NUMBER;
      ub[5 * k + 4] = (double)((float)u->gait[i * 4 + j] * s->f_max);
      k++;
    }
  for (int i = 0; i < m; i++) lb[i] = 0.0;

  /* :366-378  fmat */
  const float mu = 1.f / s->mu;
  const float f_block[15] = {mu, 0, 1.f, -mu, 0, 1.f, 0,  mu,
                             1.f, 0, -mu, 1.f, 0, 0,  1.f};
  memset(Acon, 0, sizeof(double) * (size_t)m * n);
  for (int i = 0; i < 4 * h; i++)
    for (int a = 0; a < 5; a++)
      for (int b = 0; b < 3; b++)
        Acon[(size_t)(5 * i + a) * n + 3 * i + b] = (double)f_block[3 * a + b];

  free(pw);
  free(A_qp);
  free(B_qp);
  free(Sd);
  free(X_d);
  free(SB);
  free(t);
}

/* SolverMPC.cpp:431-525 */
int oracle_reduce(int n, int m, const double* H, const double* g,
                  const double* Acon, const double* lb, const double* ub,
                  char* var_elim, int* new_cons_out, double* H_red,
                  double* g_red, double* A_red, double* lb_red,
                  double* ub_red) {
  int new_vars = n, new_cons = m;
  char* con_elim = (char*)calloc(m, 1);
  memset(var_elim, 0, n);
  for (int i = 0; i < m; i++) { /* :448-469 */
    if (!(near_zero((float)lb[i]) && near_zero((float)ub[i]))) continue;
    const double* c_row = &Acon[(size_t)i * n];
    for (int j = 0; j < n; j++) {
      if (near_one((float)c_row[j])) {
        new_vars -= 3;
        new_cons -= 5;
        const int cs = (j * 5) / 3 - 3;
        var_elim[j - 2] = 1;
        var_elim[j - 1] = 1;
        var_elim[j] = 1;
        con_elim[cs] = con_elim[cs + 1] = con_elim[cs + 2] =
            con_elim[cs + 3] = con_elim[cs + 4] = 1;
      }
    }
  }
  int* var_ind = (int*)malloc(sizeof(int) * (n + 1));
  int* con_ind = (int*)malloc(sizeof(int) * (m + 1));
  int vc = 0;
  for (int i = 0; i < n; i++)
    if (!var_elim[i]) var_ind[vc++] = i;
  vc = 0;
  for (int i = 0; i < m; i++)
    if (!con_elim[i]) con_ind[vc++] = i;
  for (int i = 0; i < new_vars; i++) { /* :501-510 */
    const int olda = var_ind[i];
    g_red[i] = g[olda];
    for (int j = 0; j < new_vars; j++)
      H_red[(size_t)i * new_vars + j] = H[(size_t)olda * n + var_ind[j]];
  }
  for (int con = 0; con < new_cons; con++) /* :512-519 (through float cval) */
    for (int st = 0; st < new_vars; st++) {
      const float cval = (float)Acon[(size_t)n * con_ind[con] + var_ind[st]];
      A_red[(size_t)con * new_vars + st] = cval;
    }
  for (int i = 0; i < new_cons; i++) { /* :520-525 */
    ub_red[i] = ub[con_ind[i]];
    lb_red[i] = lb[con_ind[i]];
  }
  free(con_elim);
  free(var_ind);
  free(con_ind);
  *new_cons_out = new_cons;
  return new_vars;
}

/* SolverMPC.cpp:296-557, use_jcqp == 0 */
int oracle_solve_mpc(const oracle_update_t* u, const oracle_setup_t* s,
                     oracle_qp_fn qp, double* q_soln, int* nwsr_out) {
  const int h = s->horizon, n = 12 * h, m = 20 * h;
  double* H = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* g = (double*)malloc(sizeof(double) * n);
  double* Ac = (double*)malloc(sizeof(double) * (size_t)m * n);
  double* lb = (double*)malloc(sizeof(double) * m);
  double* ub = (double*)malloc(sizeof(double) * m);
  double* Hr = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* gr = (double*)malloc(sizeof(double) * n);
  double* Ar = (double*)malloc(sizeof(double) * (size_t)m * n);
  double* lr = (double*)malloc(sizeof(double) * m);
  double* ur = (double*)malloc(sizeof(double) * m);
  double* qr = (double*)calloc(n, sizeof(double));
  char* ve = (char*)malloc(n);
  oracle_assemble(u, s, H, g, Ac, lb, ub, NULL);
  int nc = 0;
  const int nv = oracle_reduce(n, m, H, g, Ac, lb, ub, ve, &nc, Hr, gr, Ar, lr, ur);
  int rc = 0, nwsr = 0;
  if (nv > 0) rc = qp(nv, nc, Hr, gr, Ar, lr, ur, 100 /* :435 */, qr, &nwsr);
  int vc = 0;
  for (int i = 0; i < n; i++) /* :545-557 */
    q_soln[i] = ve[i] ? 0.0 : qr[vc++];
  if (nwsr_out) *nwsr_out = nwsr;
  free(H); free(g); free(Ac); free(lb); free(ub);
  free(Hr); free(gr); free(Ar); free(lr); free(ur); free(qr); free(ve);
  return rc;
}

/* The same, over `count` packed records; used for CPU-baseline timing so that
 * no Python sits inside the timed loop.  q_soln is [count][12h]. */
int oracle_solve_mpc_batch(const oracle_update_t* u, int count,
                           const oracle_setup_t* s, oracle_qp_fn qp,
                           double* q_soln, int* nwsr_out) {
  int bad = 0;
  for (int i = 0; i < count; i++) {
    int nw = 0;
    bad += oracle_solve_mpc(&u[i], s, qp, q_soln + (size_t)i * 12 * s->horizon, &nw) != 0;
    if (nwsr_out) nwsr_out[i] = nw;
  }
  return bad;
}

/* Gait.cpp:142-166 */
void oracle_mpc_table(int n_segments, const int offsets[4],
                      const int durations[4], int iteration, int* table) {
  for (int i = 0; i < n_segments; i++) {
    const int iter = (i + iteration + 1) % n_segments;
    for (int j = 0; j < 4; j++) {
      int progress = iter - offsets[j];
      if (progress < 0) progress += n_segments;
      table[i * 4 + j] = (progress < durations[j]) ? 1 : 0;
    }
  }
}

/* ------------------------------------------------------------------------
 * Caller side: ConvexMPCLocomotion.cpp:498-577 (updateMPCIfNeeded) and
 * :592-665 (solveDenseMPC), Gait.cpp:142-166.
 */
void oracle_pack_command(const oracle_command_t* c, float dt_mpc, int horizon,
                         float wpd[2], float* xci, oracle_update_t* u) {
  const float* p = c->position; /* :503 */
  float traj_init[12];
  int k, j;
  /* :505-507  v_des_world = omniMode ? v_des_robot : rBody^T * v_des_robot */
  const float vr[3] = {c->vel_des[0], c->vel_des[1], 0.f};
  float vw[3];
  if (c->omni_mode) {
    vw[0] = vr[0];
    vw[1] = vr[1];
    vw[2] = vr[2];
  } else {
    for (k = 0; k < 3; k++) /* (R^T)(k,:) . v, accumulated left to right */
      vw[k] = (c->r_body[0 * 3 + k] * vr[0] + c->r_body[1 * 3 + k] * vr[1]) +
              c->r_body[2 * 3 + k] * vr[2];
  }
  if (c->gait_type == 4) { /* :514-531 */
    const float t0[12] = {c->rp_des[0], c->rp_des[1], c->stand_traj[5],
                          c->stand_traj[0], c->stand_traj[1], c->body_height,
                          0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (k = 0; k < horizon; k++)
      for (j = 0; j < 12; j++) u->traj[12 * k + j] = t0[j];
  } else { /* :534-576 */
    const float max_pos_error = .1f;
    float x_start = wpd[0], y_start = wpd[1];
    /* "p[0] + 0.1": double literal, result stored to float */
    if (x_start - p[0] > max_pos_error) x_start = (float)((double)p[0] + 0.1);
    if (p[0] - x_start > max_pos_error) x_start = (float)((double)p[0] - 0.1);
    if (y_start - p[1] > max_pos_error) y_start = (float)((double)p[1] + 0.1);
    if (p[1] - y_start > max_pos_error) y_start = (float)((double)p[1] - 0.1);
    wpd[0] = x_start;
    wpd[1] = y_start;
    traj_init[0] = c->rpy_comp[0];
    traj_init[1] = c->rpy_comp[1];
    traj_init[2] = c->yaw_des_true;
    traj_init[3] = x_start;
    traj_init[4] = y_start;
    traj_init[5] = c->body_height;
    traj_init[6] = 0.f;
    traj_init[7] = 0.f;
    traj_init[8] = c->vel_des[2];
    traj_init[9] = vw[0];
    traj_init[10] = vw[1];
    traj_init[11] = 0.f;
    for (k = 0; k < horizon; k++) {
      for (j = 0; j < 12; j++) u->traj[12 * k + j] = traj_init[j];
      if (k == 0) {
        u->traj[2] = c->yaw_des_true;
      } else { /* :566-573 */
        u->traj[12 * k + 3] = u->traj[12 * (k - 1) + 3] + dt_mpc * vw[0];
        u->traj[12 * k + 4] = u->traj[12 * (k - 1) + 4] + dt_mpc * vw[1];
        u->traj[12 * k + 2] = u->traj[12 * (k - 1) + 2] + dt_mpc * c->vel_des[2];
      }
    }
  }
  /* solveDenseMPC :598-613 */
  {
    static const float Q[12] = {2.5f, 2.5f, 10.f, 50.f, 50.f, 100.f,
                                0.f,  0.f,  0.5f, 0.2f, 0.2f, 0.1f};
    for (j = 0; j < 12; j++) u->weights[j] = Q[j];
    u->alpha = 4e-5f; /* :604 */
    u->yaw = c->rpy[2]; /* :602 */
    for (j = 0; j < 3; j++) {
      u->p[j] = c->position[j];
      u->v[j] = c->v_world[j];
      u->w[j] = c->omega_world[j];
    }
    for (j = 0; j < 4; j++) u->q[j] = c->orientation[j];
    for (j = 0; j < 12; j++) /* :611-613  r[i] = pFoot[i%4][i/4] - position[i/4] */
      u->r[j] = c->p_foot[3 * (j % 4) + j / 4] - c->position[j / 4];
  }
  /* :632 update_x_drag(x_comp_integral) comes BEFORE the integrator step :636-640 */
  u->x_drag = *xci;
  {
    const float pz_err = p[2] - c->body_height; /* :625 */
    const float vx = c->v_world[0];
    const float cmpc_x_drag = 3.0f;
    if ((double)vx > 0.3 || (double)vx < -0.3)
      *xci += cmpc_x_drag * pz_err * dt_mpc / vx;
  }
  /* contact table, Gait.cpp:142-166 with _nIterations = horizon */
  {
    int table[4 * ORACLE_MAX_HORIZON];
    oracle_mpc_table(horizon, c->gait_offsets, c->gait_durations, c->gait_iteration, table);
    for (j = 0; j < 4 * horizon; j++) u->gait[j] = (unsigned char)table[j];
  }
}

/* ConvexMPCLocomotion.cpp:672-680 */
void oracle_forces_to_body(const float r_body[9], const float f_world[12],
                           float f_ff[12]) {
  int leg, i;
  for (leg = 0; leg < 4; leg++) {
    const float* f = &f_world[3 * leg];
    for (i = 0; i < 3; i++) /* (-rBody)(i,:) . f, accumulated left to right */
      f_ff[3 * leg + i] = ((-r_body[3 * i + 0]) * f[0] + (-r_body[3 * i + 1]) * f[1]) +
                          (-r_body[3 * i + 2]) * f[2];
  }
}
