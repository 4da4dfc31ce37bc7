// qmpc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the batched
// convex-MPC hot path of quadruped_ctrl:
//   state -> SRB linearisation -> condensed QP (H_red, g_red) -> exact QP solve
//   -> first-step ground reaction forces
// replacing src/MPC_Ctrl/SolverMPC.cpp:296-639 (solve_mpc) of the reference,
// which runs it for ONE robot on one CPU core through Eigen + qpOASES.
//
// Design (DESIGN.md has the derivations):
//   * one workgroup per robot, NT = 256*RB threads, RB in {1,2,3} selected by
//     the reduced problem size n_r = 3 * (#stance foot-steps) <= 64*RB.
//   * the n_r x n_r Hessian lives in REGISTERS for the whole solve: thread
//     (row i, column group c) owns H[i][c*CW .. c*CW+CW-1].  LDS only carries
//     vectors (pivot columns, matvec operands) and the small working-set
//     inverse; per robot HBM traffic is the 728 B record in and 48 B out.
//   * assembly exploits A_ct^3 = 0 (SolverMPC.cpp:235-254):
//       qH = 2 sum_pq C_pq (x) (B_p^T W B_q) + 2 alpha I
//     with batch-constant h x h tables C_pq -- no 13h x 12h B_qp is ever
//     formed (the reference multiplies it densely, SolverMPC.cpp:395).
//   * inversion by n_r symmetric Gauss-Jordan sweeps (one barrier each), then a
//     Goldfarb-Idnani dual active-set on the explicit inverse; swing feet are
//     eliminated up front exactly like SolverMPC.cpp:441-525.
//   * fp64 throughout the solve (the reference hands fp32-assembled data to a
//     double-precision qpOASES; fp64 assembly removes the fp32 rounding noise
//     instead of adding a second, uncorrelated copy of it).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmpc_device.h"

namespace {

constexpr int WAVE = 64;

__device__ __forceinline__ int sym_idx(int a, int b) {
  const int hi = a > b ? a : b, lo = a > b ? b : a;
  return hi * (hi + 1) / 2 + lo;
}

// Constraint e = 5*slot + ty on stance slot `slot` (reduced vars 3*slot..+2):
//   ty 0:  fx/mu + fz >= 0     ty 1: -fx/mu + fz >= 0
//   ty 2:  fy/mu + fz >= 0     ty 3: -fy/mu + fz >= 0
//   ty 4: -fz >= -fmax_k
// (fmat / U_b of SolverMPC.cpp:352-378; the BIG_NUMBER uppers can never be
//  active and fz >= 0 is implied by rows 0+1, so neither is instantiated.)
struct Con {
  int j1, j2;
  double a1, a2, rhs, inv_norm;
};
__device__ __forceinline__ Con make_con(int e, double mi, double inv_fr_norm,
                                        const double* fmaxk) {
  Con c;
  const int slot = e / 5, ty = e - 5 * slot, j0 = 3 * slot;
  c.j2 = j0 + 2;
  if (ty < 4) {
    c.j1 = j0 + (ty >> 1);
    c.a1 = (ty & 1) ? -mi : mi;
    c.a2 = 1.0;
    c.rhs = 0.0;
    c.inv_norm = inv_fr_norm;
  } else {
    c.j1 = j0 + 2;
    c.a1 = -1.0;
    c.a2 = 0.0;
    c.rhs = -fmaxk[slot];
    c.inv_norm = 1.0;
  }
  return c;
}

template <int RB>
struct Smem {
  static constexpr int NP = 64 * RB;
  static constexpr int KMAX = (RB == 1) ? 64 : 96;
  static constexpr int NS = KMAX * (KMAX + 1) / 2;
  // ---- live for the whole solve
  alignas(16) double colbuf[2][NP + 2];
  double x[NP];
  double g[NP];
  double fmaxk[64];
  unsigned char sidx[64];
  unsigned char actf[320];
  unsigned char slotOf[320];
  int nst, state, status, iters, khw, p;
  int cj1, cj2;
  double ca1, ca2, lp;
  // kernel parameters, parked in LDS so that 45 uniform values do not stay
  // live in SGPRs across the whole solve (they were being spilled to scratch)
  QmpcParams par;
  // ---- phase-local storage
  union U {
    struct Asm {  // linearisation + assembly
      double A[169];
      double B[3][156];
      double W[13], x0[13], Ax[13], AAx[13];
      double E[9][144];
      double e[16 * 13];
      double s[3][16 * 13];
      double Rt[9], Iinv[9];
      double ct0[256], ct4[256];  // C_00 (tau) and C_11 (sigma), h x h
    } a;
    struct Slv {  // active-set solve
      double Sinv[NS];
      alignas(16) double y[NP];
      double part[4][NP];
      double rowA[NP], rowB[NP], hc[NP], z[NP];
      double lam[KMAX], r[KMAX], d[KMAX];
      int wcid[KMAX];
    } b;
  } u;
};

enum { ST_NEXT = 0, ST_INNER = 1, ST_DONE = 2 };

// phase timestamps (test/profiling hook; P.dbg_clk == nullptr in production)
#define QMPC_TICK(k)                                                        \
  do {                                                                     \
    if (P.dbg_clk && tid == 0) P.dbg_clk[(size_t)rid * 16 + (k)] = clock64(); \
  } while (0)

template <int RB>
__device__ void solve_one(const QmpcParams& P, const int rid, Smem<RB>& S) {
  constexpr int NP = 64 * RB, CW = 16 * RB, NT = 256 * RB;
  constexpr int KMAX = Smem<RB>::KMAX;
  const int tid = threadIdx.x;
  const int i = tid % NP;   // matrix row owned by this thread
  const int c = tid / NP;   // column group (0..3): columns c*CW .. c*CW+CW-1
  const int h = P.horizon;
  const int nfs = 4 * h;    // foot-steps in the horizon (<= 64)

  QMPC_TICK(0);
  // ------------------------------------------------------------ phase 0a
  // contact table -> compact stance list (SolverMPC.cpp:441-469 finds the same
  // set by scanning for ub ~ 0 rows).
  if (tid < WAVE) {
    float fm = 0.f;
    if (tid < nfs)
      fm = (float)P.gait[(size_t)rid * nfs + tid] * (float)P.f_max;  // :361
    const bool st = !(fm < 0.01f && fm > -.01f);                      // :64-67
    const unsigned long long mask = __ballot(st);
    const int pos = __popcll(mask & ((1ull << tid) - 1ull));
    if (st) {
      S.sidx[pos] = (unsigned char)tid;
      S.fmaxk[pos] = (double)fm;
    }
    if (tid == 0) {
      S.nst = __popcll(mask);
      S.status = 0;
      S.iters = 0;
    }
  }
  __syncthreads();
  const int nst = S.nst;
  const int n = 3 * nst;
  if (nst == 0 || n > NP) {
    // all-swing: q_soln is all zeros (SolverMPC.cpp:545-551).  Too large for
    // this instantiation: hand the robot to the next size class.
    if (nst == 0) {
      if (tid < 12) P.grf[(size_t)rid * 12 + tid] = 0.f;
      if (P.soln)
        for (int k = tid; k < 12 * h; k += NT) P.soln[(size_t)rid * 12 * h + k] = 0.0;
      if (tid == 0) {
        P.status[rid] = 0;
        if (P.iters) P.iters[rid] = 0;
      }
    } else if (tid == 0) {
      if (P.next_list) {
        const int slot = atomicAdd(P.next_count, 1);
        P.next_list[slot] = rid;
      } else {
        P.status[rid] = QMPC_DEV_ST_WS_FULL;
      }
    }
    __syncthreads();
    return;
  }

  auto& Aa = S.u.a;
  const double x_drag = (double)P.x_drag[(size_t)rid * P.x_drag_stride];
  const double alpha = (double)P.alpha[(size_t)rid * P.alpha_stride];

  QMPC_TICK(1);
  // ------------------------------------------------------------ phase 0b
  // scalars: yaw rotation, world inertia inverse, x0 (SolverMPC.cpp:315-319,
  // RobotState.cpp:30-40).  Transcendentals in float like the reference
  // (cos/sin/atan2/asin on fpt); four waves take one each.
  if (tid < 169) Aa.A[tid] = 0.0;
  if (tid < 13) Aa.W[tid] = (tid < 12) ? (double)P.weights[(size_t)rid * P.weights_stride + tid] : 0.0;
  __syncthreads();
  {
    const float* q = P.q + (size_t)rid * 4;
    if (tid == 0) {
      float sy, cy;
      sincosf(P.yaw[rid], &sy, &cy);
      const double cd = cy, sd = sy;
      // R_yaw^T (A(0:3,6:9), SolverMPC.cpp:244)
      const double Rt[9] = {cd, sd, 0, -sd, cd, 0, 0, 0, 1};
      for (int k = 0; k < 9; ++k) Aa.Rt[k] = Rt[k];
      // I_world^-1 = R diag(1/I) R^T  (closed form of :247 I_world.inverse())
      const double ix = 1.0 / P.ibody[0], iy = 1.0 / P.ibody[1], iz = 1.0 / P.ibody[2];
      Aa.Iinv[0] = cd * cd * ix + sd * sd * iy;
      Aa.Iinv[1] = cd * sd * (ix - iy);
      Aa.Iinv[2] = 0;
      Aa.Iinv[3] = Aa.Iinv[1];
      Aa.Iinv[4] = sd * sd * ix + cd * cd * iy;
      Aa.Iinv[5] = 0;
      Aa.Iinv[6] = 0;
      Aa.Iinv[7] = 0;
      Aa.Iinv[8] = iz;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Aa.A[a * 13 + 6 + b] = Rt[3 * a + b];
      Aa.A[3 * 13 + 9] = 1.0;
      Aa.A[4 * 13 + 10] = 1.0;
      Aa.A[5 * 13 + 11] = 1.0;
      Aa.A[11 * 13 + 9] = x_drag;  // :239
      Aa.A[11 * 13 + 12] = 1.0;    // :243
      Aa.x0[12] = P.gravity;
      for (int k = 0; k < 3; ++k) {
        Aa.x0[3 + k] = (double)P.p[(size_t)rid * 3 + k];
        Aa.x0[6 + k] = (double)P.w[(size_t)rid * 3 + k];
        Aa.x0[9 + k] = (double)P.v[(size_t)rid * 3 + k];
      }
    } else if (tid == WAVE) {  // roll  (rpy(2), :265)
      const float w = q[0], x = q[1], y = q[2], z = q[3];
      Aa.x0[0] = (double)atan2f(2.f * (y * z + w * x), w * w - x * x - y * y + z * z);
    } else if (tid == 2 * WAVE) {  // pitch (rpy(1), :262-264)
      const float w = q[0], x = q[1], y = q[2], z = q[3];
      double asd = -2. * (double)(x * z - w * y);
      if (!(asd < .99999)) asd = .99999;
      Aa.x0[1] = (double)asinf((float)asd);
    } else if (tid == 3 * WAVE) {  // yaw   (rpy(0), :263)
      const float w = q[0], x = q[1], y = q[2], z = q[3];
      Aa.x0[2] = (double)atan2f(2.f * (x * y + w * z), w * w + x * x - y * y - z * z);
    }
  }
  __syncthreads();

  QMPC_TICK(2);
  // ------------------------------------------------------------ phase 0c
  // B0 = B_ct (SolverMPC.cpp:246-253); free-response pieces A x0, A^2 x0.
  if (tid < 156) {
    const int row = tid / 12, col = tid - 12 * row;
    const int b = col / 3, jj = col - 3 * b;
    double v = 0.0;
    if (row >= 6 && row < 9) {
      const float* r = P.r + (size_t)rid * 12;
      const double rx = r[0 * 4 + b], ry = r[1 * 4 + b], rz = r[2 * 4 + b];
      // column jj of [r]x
      const double cm0 = (jj == 0) ? 0.0 : (jj == 1 ? -rz : ry);
      const double cm1 = (jj == 0) ? rz : (jj == 1 ? 0.0 : -rx);
      const double cm2 = (jj == 0) ? -ry : (jj == 1 ? rx : 0.0);
      const double* I = &Aa.Iinv[3 * (row - 6)];
      v = I[0] * cm0 + I[1] * cm1 + I[2] * cm2;
    } else if (row >= 9 && row < 12) {
      v = (row - 9 == jj) ? 1.0 / P.mass : 0.0;
    }
    Aa.B[0][tid] = v;
  } else if (tid >= 192 && tid < 192 + 13) {
    const int row = tid - 192;
    double s = 0.0;
    for (int k = 0; k < 13; ++k) s += Aa.A[row * 13 + k] * Aa.x0[k];
    Aa.Ax[row] = s;
  }
  __syncthreads();
  if (tid < 156) {  // B1 = A B0
    const int row = tid / 12, col = tid - 12 * row;
    double s = 0.0;
    for (int k = 0; k < 13; ++k) s += Aa.A[row * 13 + k] * Aa.B[0][k * 12 + col];
    Aa.B[1][tid] = s;
  } else if (tid >= 192 && tid < 192 + 13) {
    const int row = tid - 192;
    double s = 0.0;
    for (int k = 0; k < 13; ++k) s += Aa.A[row * 13 + k] * Aa.Ax[k];
    Aa.AAx[row] = s;
  }
  __syncthreads();
  if (tid < 156) {  // B2 = A B1   (A^3 = 0 ends the series)
    const int row = tid / 12, col = tid - 12 * row;
    double s = 0.0;
    for (int k = 0; k < 13; ++k) s += Aa.A[row * 13 + k] * Aa.B[1][k * 12 + col];
    Aa.B[2][tid] = s;
  }
  // weighted tracking error of the free response at step k (k < h):
  //   e_k = W .* (x0 + A x0 t + A^2 x0 t^2/2 - xd_k),  t = (k+1) dt
  // ( = S (A_qp x0 - X_d), SolverMPC.cpp:399 )
  for (int idx = tid; idx < 13 * h; idx += NT) {
    const int k = idx / 13, row = idx - 13 * k;
    const double t = (double)(k + 1) * P.dt;
    double v = Aa.x0[row] + Aa.Ax[row] * t + Aa.AAx[row] * (0.5 * t * t);
    if (row < 12) v -= (double)P.traj[(size_t)rid * 12 * h + 12 * k + row];
    Aa.e[idx] = Aa.W[row] * v;
  }
  __syncthreads();

  QMPC_TICK(3);
  // ------------------------------------------------------------ phase 0d
  // E_pq = B_p^T W B_q  and  s_p,i = sum_{k>=i} coef_p(k-i) e_k
  for (int idx = tid; idx < 9 * 144; idx += NT) {
    const int pq = idx / 144, uv = idx - 144 * pq;
    const int pp = pq / 3, qq = pq - 3 * pp, u = uv / 12, v = uv - 12 * u;
    double s = 0.0;
#pragma unroll
    for (int row = 0; row < 12; ++row)
      s += Aa.W[row] * (Aa.B[pp][row * 12 + u] * Aa.B[qq][row * 12 + v]);  // E_qp = E_pq^T bitwise
    Aa.E[pq][uv] = s;
  }
  for (int idx = tid; idx < h * h; idx += NT) {
    Aa.ct0[idx] = P.ctab[idx];
    Aa.ct4[idx] = P.ctab[4 * h * h + idx];
  }
  for (int idx = tid; idx < 3 * 13 * h; idx += NT) {
    const int pp = idx / (13 * h), rem = idx - pp * 13 * h;
    const int st = rem / 13, row = rem - 13 * st;
    double s = 0.0;
    for (int k = st; k < h; ++k) s += P.coef[pp * h + (k - st)] * Aa.e[k * 13 + row];
    Aa.s[pp][st * 13 + row] = s;
  }
  __syncthreads();

  QMPC_TICK(4);
  // ------------------------------------------------------------ phase 1
  // gradient g_red and Hessian rows straight into registers.
  if (tid < n) {
    const int ki = S.sidx[tid / 3], ax = tid % 3;
    const int st = ki >> 2, u = 3 * (ki & 3) + ax;
    double s = 0.0;
    for (int pp = 0; pp < 3; ++pp)
      for (int row = 0; row < 12; ++row)
        s += Aa.B[pp][row * 12 + u] * Aa.s[pp][st * 13 + row];
    S.g[tid] = 2.0 * s;
  } else if (tid < NP) {
    S.g[tid] = 0.0;
  }

  double a[CW];
  {
    int si = 0, u = 0;
    const bool rowok = i < n;
    if (rowok) {
      const int ki = S.sidx[i / 3];
      si = ki >> 2;
      u = 3 * (ki & 3) + (i % 3);
    }
    const bool drag = (x_drag != 0.0);
    const int hh = h * h;
    if (!drag) {
      // x_drag == 0: only (p,q) = (0,0) and (1,1) survive:
      //   H = 2 (tau (x) E_00 + sigma (x) E_11 + alpha I)
#pragma unroll
      for (int jj = 0; jj < CW; ++jj) {
        const int j = c * CW + jj;
        double val = (i == j) ? 1.0 : 0.0;  // identity padding
        if (rowok && j < n) {
          const int kj = S.sidx[j / 3];
          const int cidx = si * h + (kj >> 2), eidx = u * 12 + 3 * (kj & 3) + (j % 3);
          double acc = Aa.ct0[cidx] * Aa.E[0][eidx] + Aa.ct4[cidx] * Aa.E[4][eidx];
          if (i == j) acc += alpha;
          val = 2.0 * acc;  // qH = 2 (B^T S B + alpha I), SolverMPC.cpp:395
        }
        a[jj] = val;
        if ((jj & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < CW; ++jj) {
        const int j = c * CW + jj;
        double val = (i == j) ? 1.0 : 0.0;
        if (rowok && j < n) {
          const int kj = S.sidx[j / 3];
          const int cidx = si * h + (kj >> 2), eidx = u * 12 + 3 * (kj & 3) + (j % 3);
          // summed so that H[i][j] == H[j][i] bitwise: (p,q) and (q,p) terms
          // are paired, and ctab[qp][sj][si] == ctab[pq][si][sj], E_qp = E_pq^T
          auto term = [&](int pq) { return P.ctab[pq * hh + cidx] * Aa.E[pq][eidx]; };
          double acc = term(0) + term(4) + term(8);
          acc += term(1) + term(3);
          acc += term(2) + term(6);
          acc += term(5) + term(7);
          if (i == j) acc += alpha;
          val = 2.0 * acc;
        }
        a[jj] = val;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (P.dbg_H) {
    double* Hd = P.dbg_H + (size_t)rid * QMPC_DBG_LD * QMPC_DBG_LD;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) Hd[(size_t)i * QMPC_DBG_LD + c * CW + jj] = a[jj];
    if (tid < NP) P.dbg_g[(size_t)rid * QMPC_DBG_LD + tid] = (tid < n) ? S.g[tid] : 0.0;
  }
  __syncthreads();  // Asm storage dead from here on (g, sidx, fmaxk are not in the union)

  QMPC_TICK(5);
  // ------------------------------------------------------------ phase 2
  // n symmetric Gauss-Jordan sweeps: a <- -H^-1, one barrier per pivot.
  // Pivot column k is owned by column group k / CW in register k % CW; it is
  // broadcast through a double-buffered LDS vector (row k == column k).
  {
    bool notpd = false;
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int r = 0; r < CW; ++r) {
        const int k = kb * CW + r;
        if (k < n) {
          double* cb = S.colbuf[k & 1];
          if (c == kb) cb[i] = a[r];
          __syncthreads();
          double d = cb[k];
          if (!(d > 1e-300)) {
            notpd = true;
            d = 1e-300;
          }
          const double dinv = 1.0 / d;
          const double ci = cb[i];
          const double f = ci * dinv;
          const bool prow = (i == k);
#pragma unroll
          for (int jj = 0; jj < CW; ++jj) {
            const double cj = cb[c * CW + jj];
            const double upd = __builtin_fma(-f, cj, a[jj]);
            a[jj] = prow ? cj * dinv : upd;
          }
          if (c == kb) a[r] = prow ? -dinv : f;
        }
      }
    }
    if (notpd && tid == 0) S.status |= QMPC_DEV_ST_NOT_PD;
  }

  auto& Sb = S.u.b;
  QMPC_TICK(6);
  // ------------------------------------------------------------ phase 3
  // unconstrained minimiser x = -H^-1 g = a * g   (distributed mat-vec)
  {
    double acc = 0.0;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) acc = __builtin_fma(a[jj], S.g[c * CW + jj], acc);
    Sb.part[c][i] = acc;
    for (int k = tid; k < KMAX; k += NT) {
      Sb.wcid[k] = -1;
      Sb.lam[k] = 0.0;
      Sb.r[k] = 0.0;
      Sb.d[k] = 0.0;
    }
    for (int k = tid; k < Smem<RB>::NS; k += NT) Sb.Sinv[k] = 0.0;
    for (int k = tid; k < 320; k += NT) {
      S.actf[k] = 0;
      S.slotOf[k] = 0xFF;
    }
    if (tid == 0) S.khw = 0;
  }
  __syncthreads();
  if (tid < NP) S.x[tid] = Sb.part[0][tid] + Sb.part[1][tid] + Sb.part[2][tid] + Sb.part[3][tid];
  __syncthreads();

  QMPC_TICK(7);
  // ------------------------------------------------------------ phase 4
  // Goldfarb-Idnani dual active set on the explicit inverse.  Wave 0 is the
  // engine (all small algebra, no block barriers inside a step); the other
  // waves serve row extractions and mat-vecs of the register-resident -H^-1.
  const int lane = tid & (WAVE - 1);
  const bool engine = tid < WAVE;
  const double mi = P.mu_inv;
  const double inv_fr = P.inv_fr_norm;
  const int ncon = 5 * nst;

  // engine: pick the most violated inactive constraint; publish it or DONE
  auto select = [&]() {
    double best = 0.0;
    int beste = -1;
    for (int e = lane; e < ncon; e += WAVE) {
      if (S.actf[e]) continue;
      const Con cn = make_con(e, mi, inv_fr, S.fmaxk);
      const double s = (cn.a1 * S.x[cn.j1] + cn.a2 * S.x[cn.j2] - cn.rhs) * cn.inv_norm;
      if (s < best) {
        best = s;
        beste = e;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off);
      const int oe = __shfl_xor(beste, off);
      if (ob < best || (ob == best && oe >= 0 && (beste < 0 || oe < beste))) {
        best = ob;
        beste = oe;
      }
    }
    if (lane == 0) {
      if (beste < 0 || best >= -P.tol) {
        S.state = ST_DONE;
      } else if (S.iters >= P.max_iter) {
        S.status |= QMPC_DEV_ST_MAXITER;
        S.state = ST_DONE;
      } else {
        const Con cn = make_con(beste, mi, inv_fr, S.fmaxk);
        S.p = beste;
        S.cj1 = cn.j1;
        S.cj2 = cn.j2;
        S.ca1 = cn.a1;
        S.ca2 = cn.a2;
        S.lp = 0.0;
        S.state = ST_NEXT;
      }
    }
  };

  if (engine) select();
  __syncthreads();

  while (S.state != ST_DONE) {
    // --- rows j1, j2 of H^-1 (= -a) out of the registers
    {
      const int j1 = S.cj1, j2 = S.cj2;
      if (i == j1) {
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) Sb.rowA[c * CW + jj] = -a[jj];
      }
      if (i == j2) {
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) Sb.rowB[c * CW + jj] = -a[jj];
      }
    }
    __syncthreads();
    if (engine) {
      const double a1 = S.ca1, a2 = S.ca2;
      for (int k = lane; k < NP; k += WAVE) Sb.hc[k] = a1 * Sb.rowA[k] + a2 * Sb.rowB[k];
    }
    // --- inner loop: one pass per (partial or full) step
    while (true) {
      if (engine) {
        __builtin_amdgcn_wave_barrier();
        const int khw = S.khw;
        // d = C_W^T H^-1 c_p ; r = S_W^-1 d
        for (int w = lane; w < khw; w += WAVE) {
          double dv = 0.0;
          const int e = Sb.wcid[w];
          if (e >= 0) {
            const Con cw = make_con(e, mi, inv_fr, S.fmaxk);
            dv = cw.a1 * Sb.hc[cw.j1] + cw.a2 * Sb.hc[cw.j2];
          }
          Sb.d[w] = dv;
        }
        __builtin_amdgcn_wave_barrier();
        for (int w = lane; w < khw; w += WAVE) {
          double rv = 0.0;
          for (int v = 0; v < khw; ++v) rv = __builtin_fma(Sb.Sinv[sym_idx(w, v)], Sb.d[v], rv);
          Sb.r[w] = (Sb.wcid[w] >= 0) ? rv : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        // y = c_p - C_W r, gathered per stance slot (no scatter conflicts)
        for (int k = lane; k < NP; k += WAVE) Sb.y[k] = 0.0;
        __builtin_amdgcn_wave_barrier();
        for (int sl = lane; sl < nst; sl += WAVE) {
          double y0 = 0.0, y1 = 0.0, y2 = 0.0;
          const int e0 = 5 * sl;
          int q;
          q = S.slotOf[e0 + 0]; if (q != 0xFF) { y0 -= mi * Sb.r[q]; y2 -= Sb.r[q]; }
          q = S.slotOf[e0 + 1]; if (q != 0xFF) { y0 += mi * Sb.r[q]; y2 -= Sb.r[q]; }
          q = S.slotOf[e0 + 2]; if (q != 0xFF) { y1 -= mi * Sb.r[q]; y2 -= Sb.r[q]; }
          q = S.slotOf[e0 + 3]; if (q != 0xFF) { y1 += mi * Sb.r[q]; y2 -= Sb.r[q]; }
          q = S.slotOf[e0 + 4]; if (q != 0xFF) { y2 += Sb.r[q]; }
          Sb.y[3 * sl + 0] = y0;
          Sb.y[3 * sl + 1] = y1;
          Sb.y[3 * sl + 2] = y2;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
          Sb.y[S.cj1] += S.ca1;
          if (S.ca2 != 0.0) Sb.y[S.cj2] += S.ca2;
        }
      }
      __syncthreads();
      // --- z = H^-1 y : distributed mat-vec on the register matrix
      {
        double acc = 0.0;
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) acc = __builtin_fma(a[jj], Sb.y[c * CW + jj], acc);
        Sb.part[c][i] = acc;
      }
      __syncthreads();
      if (engine) {
        for (int k = lane; k < NP; k += WAVE)
          Sb.z[k] = -(Sb.part[0][k] + Sb.part[1][k] + Sb.part[2][k] + Sb.part[3][k]);
        __builtin_amdgcn_wave_barrier();
        const int khw = S.khw;
        const int j1 = S.cj1, j2 = S.cj2;
        const double a1 = S.ca1, a2 = S.ca2;
        const Con cp = make_con(S.p, mi, inv_fr, S.fmaxk);
        const double delta = a1 * Sb.z[j1] + a2 * Sb.z[j2];
        const double hcn = a1 * Sb.hc[j1] + a2 * Sb.hc[j2];
        const double sp = a1 * S.x[j1] + a2 * S.x[j2] - cp.rhs;
        const bool dep = !(delta > 1e-12 * hcn);
        const double t2 = dep ? __builtin_inf() : -sp / delta;
        // t1: largest dual step keeping the working-set multipliers >= 0
        double t1 = __builtin_inf();
        int l = -1;
        for (int w = lane; w < khw; w += WAVE) {
          const double rv = Sb.r[w];
          if (Sb.wcid[w] >= 0 && rv > 0.0) {
            const double q = Sb.lam[w] / rv;
            if (q < t1) {
              t1 = q;
              l = w;
            }
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double ot = __shfl_xor(t1, off);
          const int ol = __shfl_xor(l, off);
          if (ot < t1 || (ot == t1 && ol >= 0 && (l < 0 || ol < l))) {
            t1 = ot;
            l = ol;
          }
        }
        const double t = (t2 <= t1) ? t2 : t1;
        if (!(t < __builtin_inf())) {
          if (lane == 0) {
            S.status |= QMPC_DEV_ST_INFEASIBLE;
            S.state = ST_DONE;
          }
        } else {
          if (!dep)
            for (int k = lane; k < n; k += WAVE) S.x[k] = __builtin_fma(t, Sb.z[k], S.x[k]);
          for (int w = lane; w < khw; w += WAVE) Sb.lam[w] -= t * Sb.r[w];
          if (lane == 0) {
            S.lp += t;
            S.iters += 1;
          }
          __builtin_amdgcn_wave_barrier();
          if (t2 <= t1) {
            // full step: constraint p joins the working set in a free slot
            int q = -1;
            for (int base = 0; base < KMAX && q < 0; base += WAVE) {
              const int w = base + lane;
              const bool fr = (w < KMAX) && (Sb.wcid[w] < 0);
              const unsigned long long m = __ballot(fr);
              if (m) q = base + __ffsll((long long)m) - 1;
            }
            if (q < 0) {
              if (lane == 0) {
                S.status |= QMPC_DEV_ST_WS_FULL;
                S.state = ST_DONE;
              }
            } else {
              const int kn = (q + 1 > khw) ? q + 1 : khw;
              const double dinv = 1.0 / delta;
              // bordered-inverse update of S_W^-1 (inactive slots have r = 0)
              for (int hi = 0; hi < kn; ++hi)
                for (int lo = lane; lo <= hi; lo += WAVE) {
                  const int idx = hi * (hi + 1) / 2 + lo;
                  double v;
                  if (hi == q && lo == q) v = dinv;
                  else if (hi == q) v = -Sb.r[lo] * dinv;
                  else if (lo == q) v = -Sb.r[hi] * dinv;
                  else v = __builtin_fma(Sb.r[hi] * dinv, Sb.r[lo], Sb.Sinv[idx]);
                  Sb.Sinv[idx] = v;
                }
              if (lane == 0) {
                Sb.wcid[q] = S.p;
                Sb.lam[q] = S.lp;
                Sb.r[q] = 0.0;
                S.slotOf[S.p] = (unsigned char)q;
                S.actf[S.p] = 1;
                S.khw = kn;
              }
              __builtin_amdgcn_wave_barrier();
              select();  // next violated constraint, or DONE
            }
          } else {
            // partial step: multiplier of slot l hit zero -> drop it
            const double sll = Sb.Sinv[sym_idx(l, l)];
            const double il = 1.0 / sll;
            // S' = S - S[:,l] S[l,:] / S[l,l] on the other slots; row/col l is
            // only read here and zeroed afterwards, so in place is safe.
            for (int hi = 0; hi < khw; ++hi)
              for (int lo = lane; lo <= hi; lo += WAVE)
                if (hi != l && lo != l) {
                  const int idx = hi * (hi + 1) / 2 + lo;
                  Sb.Sinv[idx] = __builtin_fma(-Sb.Sinv[sym_idx(hi, l)] * il, Sb.Sinv[sym_idx(l, lo)], Sb.Sinv[idx]);
                }
            __builtin_amdgcn_wave_barrier();
            for (int w = lane; w < khw; w += WAVE) Sb.Sinv[sym_idx(w, l)] = 0.0;
            if (lane == 0) {
              const int e = Sb.wcid[l];
              Sb.wcid[l] = -1;
              Sb.lam[l] = 0.0;
              Sb.r[l] = 0.0;
              S.slotOf[e] = 0xFF;
              S.actf[e] = 0;
              S.state = ST_INNER;
            }
          }
        }
      }
      __syncthreads();
      if (S.state != ST_INNER) break;
    }
  }

  QMPC_TICK(8);
  // ------------------------------------------------------------ outputs
  // get_solution(0..11): forces of the four feet at horizon step 0
  // (convexMPC_interface.cpp:175-180, ConvexMPCLocomotion.cpp:672-685).
  if (tid < 12) {
    const int foot = tid / 3, ax = tid - 3 * foot;
    float f = 0.f;
    for (int sl = 0; sl < nst && sl < 4; ++sl)
      if (S.sidx[sl] == foot) f = (float)S.x[3 * sl + ax];
    P.grf[(size_t)rid * 12 + tid] = f;
  }
  if (P.soln) {
    double* so = P.soln + (size_t)rid * 12 * h;
    for (int k = tid; k < 12 * h; k += NT) so[k] = 0.0;
    __syncthreads();
    if (tid < n) so[3 * S.sidx[tid / 3] + (tid % 3)] = S.x[tid];
  }
  if (tid == 0) {
    P.status[rid] = S.status;
    if (P.iters) P.iters[rid] = S.iters;
  }
  __syncthreads();
  QMPC_TICK(9);
}

}  // namespace

// One workgroup per robot (list == nullptr: robot = blockIdx.x, grid covers the
// batch) or a persistent stride over a deferred-robot list (larger classes).
template <int RB>
__global__ __launch_bounds__(256 * RB, (RB == 1) ? 4 : (RB == 2 ? 2 : 3)) void qmpc_solve_kernel(const QmpcParams P) {
  __shared__ Smem<RB> S;
  if (threadIdx.x == 0) S.par = P;
  __syncthreads();
  const int cnt = S.par.list ? *S.par.count : S.par.batch;
#pragma unroll 1
  for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
    const int rid = S.par.list ? S.par.list[it] : it;
    solve_one<RB>(S.par, rid, S);
  }
}

extern "C" hipError_t qmpc_launch(int rb, const QmpcParams* P, int grid, hipStream_t stream) {
  switch (rb) {
    case 1: hipLaunchKernelGGL(qmpc_solve_kernel<1>, dim3(grid), dim3(256), 0, stream, *P); break;
    case 2: hipLaunchKernelGGL(qmpc_solve_kernel<2>, dim3(grid), dim3(512), 0, stream, *P); break;
    case 3: hipLaunchKernelGGL(qmpc_solve_kernel<3>, dim3(grid), dim3(768), 0, stream, *P); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
