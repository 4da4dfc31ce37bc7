// mfma_sweep.hip -- round-6 prototype: stage 3 of the 64-row class (the symmetric Gauss-Jordan inverse) two ways, on
// the same synthetic SPD matrices, at the occupancy of the five-per-CU instantiation (256 threads, <= 32 KB of LDS):
//
//   A  the shipped form (copy of qmpc_kernels.hip, class-1 branch of stage 3): thread (row = lane, column group = wave)
//      holds 16 doubles, two pivots per barrier, 32 DPP fp64 fmacs per wave and pair = n^3 multiply-adds in all;
//   B  the matrix as the LOWER BLOCK TRIANGLE of 16 x 16 tiles in v_mfma_f64_16x16x4 accumulator layout (10 of 16 tiles),
//      FOUR pivots per step: one wave inverts the 4 x 4 pivot block (2 x 2 Schur complement) and forms F = C P^-1 for all
//      64 rows, every tile then takes ONE matrix instruction (T -= F_I C_J^T).  10 MFMAs per step and workgroup instead of
//      256 vector fmacs per pair-of-pairs; half the barriers.
//
// Prints: max |H Hinv - I| of both, max |A - B|, per-workgroup shader-clock cycles of the stage (median / max) and the
// kernel time at 1 280 / 8 192 / 16 384 workgroups.
//    hipcc -O3 --offload-arch=gfx950 mfma_sweep.hip -o /tmp/mfma_sweep && /tmp/mfma_sweep
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../quadruped_ctrl_amd/csrc/qmpc_wave.h"

namespace {
constexpr int NP = 64, CW = 16;
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fmac16_rowbcast(double (&a)[16], double c, double u) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
        "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
      : "v"(c), "v"(u));
}
template <int J0>
__device__ __forceinline__ void fmac4_rowbcast(double (&a)[16], double c, double u) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %4, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %4, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
      : "+v"(a[J0]), "+v"(a[J0 + 1]), "+v"(a[J0 + 2]), "+v"(a[J0 + 3])
      : "v"(c), "v"(u), "n"(J0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3));
}

// the synthetic matrix: rank-8 Gram matrix of hashed vectors + a graded diagonal (SPD, condition ~1e3), identity padding
__host__ __device__ inline double vhash(int i, int k, int seed) {
  const unsigned x = (unsigned)(i * 37 + k * 101 + seed * 13 + 7) * 2654435761u;
  return (double)((x >> 8) & 0xffff) / 65536.0 - 0.5;
}
__host__ __device__ inline double hmat(int i, int j, int n, int seed) {
  if (i >= n || j >= n) return i == j ? 1.0 : 0.0;
  double s = (i == j) ? 0.02 + 0.001 * i : 0.0;
  for (int k = 0; k < 8; ++k) s += vhash(i, k, seed) * vhash(j, k, seed);
  return s;
}

constexpr int LDS_PAD = 30 * 1024;  // what the five-per-CU instantiation of the real kernel holds

// ------------------------------------------------------------------ A: the shipped sweep
struct SmemA {
  alignas(16) double colbuf[2][2][NP + 2];
  double ubuf[2][2][NP];
  double Hp[NP * (NP + 1) / 2];
  char pad[LDS_PAD - (2 * 2 * (NP + 2) + 2 * 2 * NP + NP * (NP + 1) / 2) * 8];
};

__global__ __launch_bounds__(256, 5) void sweep_a(long long* clk, double* out, int n, int nout) {
  __shared__ SmemA S;
  const int tid = threadIdx.x, lane = tid & 63, i = tid % NP, c = tid / NP;
  const int seed = blockIdx.x & 1023;
  double a[CW];
#pragma unroll
  for (int jj = 0; jj < CW; ++jj) a[jj] = hmat(i, c * CW + jj, n, seed);
  if (tid == 0) S.pad[0] = 0;
  __syncthreads();
  const long long t0 = clock64();
  bool notpd = false;
  {
    double sv0 = 0.0, sv1 = 0.0;
    double pd0, pe, pd1, pdet, px, pc0, pc1;
    auto prod_read = [&](double c0n, double c1n, int k0n) __attribute__((always_inline)) {
      pc0 = c0n;
      pc1 = c1n;
      pd0 = readlane_f64(c0n, k0n);
      pe = readlane_f64(c0n, k0n + 1);
      pd1 = readlane_f64(c1n, k0n + 1);
    };
    auto prod_det = [&]() __attribute__((always_inline)) {
      pdet = __builtin_fma(pd0, pd1, -pe * pe);
      notpd |= !(pd0 > 0.0) | !(pdet > 0.0);
    };
    auto prod_rcp = [&]() __attribute__((always_inline)) { px = __builtin_amdgcn_rcp(pdet); };
    auto prod_newton = [&]() __attribute__((always_inline)) {
      const double e1 = __builtin_fma(-pdet, px, 1.0);
      px = __builtin_fma(px, e1, px);
    };
    double pfg0, pfg1, pq0, pq1;
    auto prod_fg = [&](int k0n) __attribute__((always_inline)) {
      const double i11 = pd1 * px, i01 = pe * px, i00 = pd0 * px;
      pfg0 = __builtin_fma(i11, pc0, -i01 * pc1);
      pfg1 = __builtin_fma(i00, pc1, -i01 * pc0);
      const bool p0 = (i == k0n), p1 = (i == k0n + 1);
      pq0 = p0 ? -i11 : (p1 ? i01 : 0.0);
      pq1 = p0 ? i01 : (p1 ? -i00 : 0.0);
    };
    auto prod_store = [&](int k0n, int mn) __attribute__((always_inline)) {
      const bool pr = (i == k0n) | (i == k0n + 1);
      sv0 = pr ? pq0 : pfg0;
      sv1 = pr ? pq1 : pfg1;
      S.colbuf[mn & 1][0][i] = pc0;
      S.colbuf[mn & 1][1][i] = pc1;
      S.ubuf[mn & 1][0][i] = -(pfg0 + pq0);
      S.ubuf[mn & 1][1][i] = -(pfg1 + pq1);
    };
    if (c == 0) {
      prod_read(a[0], a[1], 0);
      prod_det();
      prod_rcp();
      prod_newton();
      prod_newton();
      prod_fg(0);
      prod_store(0, 0);
    }
    __syncthreads();
#define QMPC_PIN __builtin_amdgcn_sched_barrier(0)
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
      StaticFor<0, CW / 2>::run([&](auto pc) __attribute__((always_inline)) {
        constexpr int r0 = 2 * decltype(pc)::value, r1 = r0 + 1;
        constexpr int rn0 = (r0 + 2 < CW) ? r0 + 2 : 0, rn1 = rn0 + 1;
        constexpr int G0 = 4 * (rn0 / 4), G1 = (G0 + 4) % 16, G2 = (G0 + 8) % 16, G3 = (G0 + 12) % 16;
        const int k0 = kb * CW + r0;
        if (k0 < n) {
          const int m = k0 >> 1;
          const double cv0 = S.colbuf[m & 1][0][c * CW + (lane & 15)];
          const double cv1 = S.colbuf[m & 1][1][c * CW + (lane & 15)];
          const double nu0 = S.ubuf[m & 1][0][i], nu1 = S.ubuf[m & 1][1][i];
          const double so0 = sv0, so1 = sv1;
          const int kbn = (r0 + 2 < CW) ? kb : kb + 1;
          if (k0 + 2 < n && c == kbn) {
            fmac4_rowbcast<G0>(a, cv0, nu0);
            fmac4_rowbcast<G0>(a, cv1, nu1);
            QMPC_PIN;
            prod_read(a[rn0], a[rn1], k0 + 2);
            QMPC_PIN;
            fmac4_rowbcast<G1>(a, cv0, nu0);
            QMPC_PIN;
            prod_det();
            prod_rcp();
            QMPC_PIN;
            fmac4_rowbcast<G1>(a, cv1, nu1);
            QMPC_PIN;
            prod_newton();
            QMPC_PIN;
            fmac4_rowbcast<G2>(a, cv0, nu0);
            QMPC_PIN;
            prod_newton();
            QMPC_PIN;
            fmac4_rowbcast<G2>(a, cv1, nu1);
            QMPC_PIN;
            prod_fg(k0 + 2);
            QMPC_PIN;
            fmac4_rowbcast<G3>(a, cv0, nu0);
            QMPC_PIN;
            prod_store(k0 + 2, m + 1);
            QMPC_PIN;
            fmac4_rowbcast<G3>(a, cv1, nu1);
          } else if (c * CW < n) {
            fmac16_rowbcast(a, cv0, nu0);
            fmac16_rowbcast(a, cv1, nu1);
          }
          if (c == kb) {
            a[r0] = so0;
            a[r1] = so1;
          }
          __syncthreads();
        }
      });
    }
#undef QMPC_PIN
  }
  // the inverse leaves the registers: packed lower triangle in LDS (stage 4 of the real kernel, without x_u)
  if (i < n) {
    const int rb = i * (i + 1) / 2;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) {
      const int j = c * CW + jj;
      if (j <= i) S.Hp[rb + j] = -a[jj];
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
  if ((int)blockIdx.x < nout)
    for (int k = tid; k < n * (n + 1) / 2; k += 256) out[(size_t)blockIdx.x * (NP * (NP + 1) / 2) + k] = S.Hp[k];
  if (notpd && tid == 0) clk[blockIdx.x] = -1;
}

// ------------------------------------------------------------------ A2: the shipped sweep, symmetric pivot handling
__global__ __launch_bounds__(256, 5) void sweep_a2(long long* clk, double* out, int n, int nout) {
  __shared__ SmemA S;
  const int tid = threadIdx.x, lane = tid & 63, i = tid % NP, c = tid / NP;
  const int seed = blockIdx.x & 1023;
  double a[CW];
#pragma unroll
  for (int jj = 0; jj < CW; ++jj) a[jj] = hmat(i, c * CW + jj, n, seed);
  if (tid == 0) S.pad[0] = 0;
  __syncthreads();
  const long long t0 = clock64();
  bool notpd = false;
  {
    // SYMMETRIC form (round 6): with C' = C - [e_k0 e_k1] used BOTH as the broadcast vector and in F' = C' P^-1,
    //   A - F' C'^T  is the sweep result everywhere except the 2 x 2 pivot block, which comes out 2 I - P^-1: the producer
    //   subtracts 2 from its two diagonal entries when it reads the pair.  No pivot-row selects, no pivot-column write-back.
    double pd0, pe, pd1, pdet, px, pc0, pc1, pm0, pm1;
    auto prod_read = [&](double c0n, double c1n, int k0n) __attribute__((always_inline)) {
      pd0 = readlane_f64(c0n, k0n);
      pe = readlane_f64(c0n, k0n + 1);
      pd1 = readlane_f64(c1n, k0n + 1);
      pm0 = (i == k0n) ? 1.0 : 0.0;
      pm1 = (i == k0n + 1) ? 1.0 : 0.0;
      pc0 = c0n - pm0;
      pc1 = c1n - pm1;
    };
    auto prod_det = [&]() __attribute__((always_inline)) {
      pdet = __builtin_fma(pd0, pd1, -pe * pe);
      notpd |= !(pd0 > 0.0) | !(pdet > 0.0);
    };
    auto prod_rcp = [&]() __attribute__((always_inline)) { px = __builtin_amdgcn_rcp(pdet); };
    auto prod_newton = [&]() __attribute__((always_inline)) {
      const double e1 = __builtin_fma(-pdet, px, 1.0);
      px = __builtin_fma(px, e1, px);
    };
    double nf0, nf1;  // -F'_i0, -F'_i1
    auto prod_fg = [&](int k0n) __attribute__((always_inline)) {
      const double i11 = pd1 * px, i01 = pe * px, i00 = pd0 * px;
      nf0 = __builtin_fma(-i11, pc0, i01 * pc1);
      nf1 = __builtin_fma(-i00, pc1, i01 * pc0);
    };
    auto prod_store = [&](int k0n, int mn) __attribute__((always_inline)) {
      S.colbuf[mn & 1][0][i] = pc0;
      S.colbuf[mn & 1][1][i] = pc1;
      S.ubuf[mn & 1][0][i] = nf0;
      S.ubuf[mn & 1][1][i] = nf1;
    };
    if (c == 0) {
      prod_read(a[0], a[1], 0);
      a[0] = __builtin_fma(-2.0, pm0, a[0]);
      a[1] = __builtin_fma(-2.0, pm1, a[1]);
      prod_det();
      prod_rcp();
      prod_newton();
      prod_newton();
      prod_fg(0);
      prod_store(0, 0);
    }
    __syncthreads();
#define QMPC_PIN __builtin_amdgcn_sched_barrier(0)
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
      StaticFor<0, CW / 2>::run([&](auto pc) __attribute__((always_inline)) {
        constexpr int r0 = 2 * decltype(pc)::value, r1 = r0 + 1;
        constexpr int rn0 = (r0 + 2 < CW) ? r0 + 2 : 0, rn1 = rn0 + 1;
        constexpr int G0 = 4 * (rn0 / 4), G1 = (G0 + 4) % 16, G2 = (G0 + 8) % 16, G3 = (G0 + 12) % 16;
        const int k0 = kb * CW + r0;
        if (k0 < n) {
          const int m = k0 >> 1;
          const double cv0 = S.colbuf[m & 1][0][c * CW + (lane & 15)];
          const double cv1 = S.colbuf[m & 1][1][c * CW + (lane & 15)];
          const double nu0 = S.ubuf[m & 1][0][i], nu1 = S.ubuf[m & 1][1][i];
          const int kbn = (r0 + 2 < CW) ? kb : kb + 1;
          if (k0 + 2 < n && c == kbn) {
            fmac4_rowbcast<G0>(a, cv0, nu0);
            fmac4_rowbcast<G0>(a, cv1, nu1);
            QMPC_PIN;
            prod_read(a[rn0], a[rn1], k0 + 2);
            a[rn0] = __builtin_fma(-2.0, pm0, a[rn0]);
            a[rn1] = __builtin_fma(-2.0, pm1, a[rn1]);
            QMPC_PIN;
            fmac4_rowbcast<G1>(a, cv0, nu0);
            QMPC_PIN;
            prod_det();
            prod_rcp();
            QMPC_PIN;
            fmac4_rowbcast<G1>(a, cv1, nu1);
            QMPC_PIN;
            prod_newton();
            QMPC_PIN;
            fmac4_rowbcast<G2>(a, cv0, nu0);
            QMPC_PIN;
            prod_newton();
            QMPC_PIN;
            fmac4_rowbcast<G2>(a, cv1, nu1);
            QMPC_PIN;
            prod_fg(k0 + 2);
            QMPC_PIN;
            fmac4_rowbcast<G3>(a, cv0, nu0);
            QMPC_PIN;
            prod_store(k0 + 2, m + 1);
            QMPC_PIN;
            fmac4_rowbcast<G3>(a, cv1, nu1);
          } else if (c * CW < n) {
            fmac16_rowbcast(a, cv0, nu0);
            fmac16_rowbcast(a, cv1, nu1);
          }
          __syncthreads();
        }
      });
    }
#undef QMPC_PIN
  }
  // the inverse leaves the registers: packed lower triangle in LDS (stage 4 of the real kernel, without x_u)
  if (i < n) {
    const int rb = i * (i + 1) / 2;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) {
      const int j = c * CW + jj;
      if (j <= i) S.Hp[rb + j] = -a[jj];
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
  if ((int)blockIdx.x < nout)
    for (int k = tid; k < n * (n + 1) / 2; k += 256) out[(size_t)blockIdx.x * (NP * (NP + 1) / 2) + k] = S.Hp[k];
  if (notpd && tid == 0) clk[blockIdx.x] = -1;
}


// ------------------------------------------------------------------ A3: A2 with ROTATING producers and deferred bulk updates
// Column pair m = (2m, 2m+1) belongs to wave m % 4 (registers 2 (m / 4), +1): the producing role moves to the next wave every
// pair.  The wave that produces pair s applies pair s - 1's update only to the block of four registers that holds its new pivot
// columns (8 fmacs) next to the pivot arithmetic and DEFERS the other three blocks, one to each of its next three steps (the
// updates are additive and commute; a column is complete again long before it becomes a pivot, four pairs later).  Every wave
// then issues ~40 vector instructions per step (32 + 8 deferred, or 8 + the pivot arithmetic) where the block mapping has
// 58 in the producer and 36 in the others -- and a step is as long as its longest wave.  Pivot columns and F live in a ring of
// five step slots (a deferred block reads the pair of four steps ago).
constexpr int RING = 5;
struct SmemA3 {
  alignas(16) double colbuf[RING][2][NP + 2];
  double ubuf[RING][2][NP];
  double Hp[NP * (NP + 1) / 2];
  char pad[LDS_PAD - (RING * 2 * (NP + 2) + RING * 2 * NP + NP * (NP + 1) / 2) * 8];
};
// (every control path "defines" all sixteen accumulators, or the register coalescer keeps two copies of the array and moves all
//  of it at every join)
__device__ __forceinline__ void touch16(double (&a)[16]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
               "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}
template <int B>
__device__ __forceinline__ void fmac_block(double (&a)[16], double c0, double u0, double c1, double u1) {
  fmac4_rowbcast<4 * B>(a, c0, u0);
  fmac4_rowbcast<4 * B>(a, c1, u1);
}
__device__ __forceinline__ void fmac_block_rt(int b, double (&a)[16], double c0, double u0, double c1, double u1) {
  if (b == 0) fmac_block<0>(a, c0, u0, c1, u1);
  else if (b == 1) fmac_block<1>(a, c0, u0, c1, u1);
  else if (b == 2) fmac_block<2>(a, c0, u0, c1, u1);
  else fmac_block<3>(a, c0, u0, c1, u1);
}

__global__ __launch_bounds__(256, 5) void sweep_a3(long long* clk, double* out, int n, int nout) {
  __shared__ SmemA3 S;
  const int tid = threadIdx.x, lane = tid & 63, i = tid % NP;
  const int c = __builtin_amdgcn_readfirstlane(tid / NP);
  const int seed = blockIdx.x & 1023;
  auto col = [&](int r) __attribute__((always_inline)) { return 8 * (r >> 1) + 2 * c + (r & 1); };
  double a[CW];
#pragma unroll
  for (int r = 0; r < CW; ++r) a[r] = hmat(i, col(r), n, seed);
  if (tid == 0) S.pad[0] = 0;
  __syncthreads();
  const long long t0 = clock64();
  bool notpd = false;
  const int M = (n + 1) >> 1;          // pivot pairs
  const int mycol = col(lane & 15);    // the column whose pivot-column entry this lane holds for the broadcast
  double pd0, pe, pd1, pdet, px, pc0, pc1, pm0, pm1, pt0, pt1, pnf0, pnf1;
  auto prod_read = [&](double c0n, double c1n, int k0n) __attribute__((always_inline)) {
    pd0 = readlane_f64(c0n, k0n);
    pe = readlane_f64(c0n, k0n + 1);
    pd1 = readlane_f64(c1n, k0n + 1);
    pm0 = (i == k0n) ? 1.0 : 0.0;
    pm1 = (i == k0n + 1) ? 1.0 : 0.0;
    pc0 = c0n - pm0;
    pc1 = c1n - pm1;
  };
  auto prod_math = [&]() __attribute__((always_inline)) {
    pdet = __builtin_fma(pd0, pd1, -pe * pe);
    notpd |= !(pd0 > 0.0) | !(pdet > 0.0);
    px = __builtin_amdgcn_rcp(pdet);
    pt0 = __builtin_fma(pe, pc1, -pd1 * pc0);
    pt1 = __builtin_fma(pe, pc0, -pd0 * pc1);
    double e1 = __builtin_fma(-pdet, px, 1.0);
    px = __builtin_fma(px, e1, px);
    e1 = __builtin_fma(-pdet, px, 1.0);
    px = __builtin_fma(px, e1, px);
    pnf0 = pt0 * px;
    pnf1 = pt1 * px;
  };
  auto prod_store = [&](int slot) __attribute__((always_inline)) {
    S.colbuf[slot][0][i] = pc0;
    S.colbuf[slot][1][i] = pc1;
    S.ubuf[slot][0][i] = pnf0;
    S.ubuf[slot][1][i] = pnf1;
  };
  // steps s = 0 .. M + 3.  Every wave runs ITS OWN straight-line program (role = wave index, known at compile time inside the
  // program): no branch on the role inside a step, so the sixteen accumulators keep their registers from step to step.
  auto program = [&](auto rolec) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(rolec)::value;
    StaticFor<0, 36>::run([&](auto sc) __attribute__((always_inline)) {
      constexpr int sst = decltype(sc)::value;
      constexpr int Q = sst >> 2, w = sst & 3;
      constexpr int PB = (Q >> 1) & 3, DB = w;
      constexpr int t = (w - ROLE) & 3;
      if (sst <= M + 3) {
        const bool have = sst >= 1 && sst - 1 < M;
        constexpr int slot_p = ((sst - 1) % RING + RING) % RING;
        double cv0 = 0.0, cv1 = 0.0, nu0 = 0.0, nu1 = 0.0;
        if (have) {
          cv0 = S.colbuf[slot_p][0][mycol];
          cv1 = S.colbuf[slot_p][1][mycol];
          nu0 = S.ubuf[slot_p][0][i];
          nu1 = S.ubuf[slot_p][1][i];
        }
        if constexpr (t == 0) {
          if constexpr (Q < 8) {
            fmac_block<PB>(a, cv0, nu0, cv1, nu1);
            if constexpr (DB != PB) fmac_block<DB>(a, cv0, nu0, cv1, nu1);
            // (beyond the last pair this runs on identity padding rows: harmless, nothing of it is read)
            prod_read(a[2 * Q], a[2 * Q + 1], 2 * sst);
            a[2 * Q] = __builtin_fma(-2.0, pm0, a[2 * Q]);
            a[2 * Q + 1] = __builtin_fma(-2.0, pm1, a[2 * Q + 1]);
            const bool np0 = notpd;
            prod_math();
            if (sst >= M) notpd = np0;
            prod_store(sst % RING);
          }
        } else {
          fmac_block<0>(a, cv0, nu0, cv1, nu1);
          fmac_block<1>(a, cv0, nu0, cv1, nu1);
          fmac_block<2>(a, cv0, nu0, cv1, nu1);
          fmac_block<3>(a, cv0, nu0, cv1, nu1);
          constexpr int s_mine = sst - t, pd = s_mine - 1;
          if constexpr (s_mine >= 0 && pd >= 0 && (((s_mine >> 3) & 3) != DB)) {
            constexpr int slot_d = pd % RING;
            double dv0 = 0.0, dv1 = 0.0, du0 = 0.0, du1 = 0.0;
            if (pd < M) {
              dv0 = S.colbuf[slot_d][0][mycol];
              dv1 = S.colbuf[slot_d][1][mycol];
              du0 = S.ubuf[slot_d][0][i];
              du1 = S.ubuf[slot_d][1][i];
            }
            fmac_block<DB>(a, dv0, du0, dv1, du1);
          }
        }
        if (sst <= M) __syncthreads();
      }
    });
  };
  if (c == 0) program(std::integral_constant<int, 0>{});
  else if (c == 1) program(std::integral_constant<int, 1>{});
  else if (c == 2) program(std::integral_constant<int, 2>{});
  else program(std::integral_constant<int, 3>{});
  __syncthreads();
  if (i < n) {
    const int rb = i * (i + 1) / 2;
#pragma unroll
    for (int r = 0; r < CW; ++r) {
      const int j = col(r);
      if (j <= i) S.Hp[rb + j] = -a[r];
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
  if ((int)blockIdx.x < nout)
    for (int k = tid; k < n * (n + 1) / 2; k += 256) out[(size_t)blockIdx.x * (NP * (NP + 1) / 2) + k] = S.Hp[k];
  if (__syncthreads_or(notpd ? 1 : 0) && tid == 0) clk[blockIdx.x] = -1;
}

// ------------------------------------------------------------------ B: tiles on the matrix cores, four pivots per step
constexpr int TS = 16 * 17;  // a staged tile: column-major with a stride of 17 (conflict-free for writers AND readers)
struct SmemB {
  union {
    double stage[10 * TS];          // assembly layout -> accumulator layout (21.8 KB, dead once the tiles are loaded)
    struct {
      double Cb[2][4][NP];          // pivot columns of the step (by parity): [q][row]
      double nF[4][NP];             // -F (pivot rows: -(I - P^-1)): the A operand
      double Ft[4][NP];             // F (pivot rows: -P^-1): what the pivot columns become
      double Hp[NP * (NP + 1) / 2];
    } w;
  } u;
  char pad[LDS_PAD - ((2 * 4 + 4 + 4) * NP + NP * (NP + 1) / 2) * 8];
};

template <int VARIANT>
__global__ __launch_bounds__(256, 5) void sweep_b(long long* clk, double* out, int n, int nout) {
  __shared__ SmemB S;
  const int tid = threadIdx.x, lane = tid & 63, i = tid % NP, c = tid / NP;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, rq = lane >> 4;
  const int seed = blockIdx.x & 1023;
  double a[CW];
#pragma unroll
  for (int jj = 0; jj < CW; ++jj) a[jj] = hmat(i, c * CW + jj, n, seed);
  if (tid == 0) S.pad[0] = 0;
  __syncthreads();
  const long long t0 = clock64();
  // ---- tiles of this wave: wave 0 = (0,0) + the pivot-block duty; 1 = column block 0; 2 = column block 1; 3 = the rest
  //      slot s of wave w: (I, J)
  int TI[3], TJ[3];
  if (wv == 0) { TI[0] = 0; TJ[0] = 0; TI[1] = -1; TJ[1] = 0; TI[2] = -1; TJ[2] = 0; }
  else if (wv == 1) { TI[0] = 1; TJ[0] = 0; TI[1] = 2; TJ[1] = 0; TI[2] = 3; TJ[2] = 0; }
  else if (wv == 2) { TI[0] = 1; TJ[0] = 1; TI[1] = 2; TJ[1] = 1; TI[2] = 3; TJ[2] = 1; }
  else { TI[0] = 2; TJ[0] = 2; TI[1] = 3; TJ[1] = 2; TI[2] = 3; TJ[2] = 3; }
  const int nblk = (n + 15) >> 4;
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (TI[s] >= nblk) TI[s] = -1;  // a tile that lies in the identity padding never changes
  // ---- assembly layout -> staging (lower block triangle, tile (I, J) at (I (I + 1) / 2 + J) TS, element (r, cc) at 17 cc + r)
  {
    const int I = i >> 4, r = i & 15;
    if (c <= I) {
      double* const tb = S.u.stage + (I * (I + 1) / 2 + c) * TS + r;
#pragma unroll
      for (int jj = 0; jj < CW; ++jj) tb[17 * jj] = a[jj];
    }
  }
  __syncthreads();
  v4d t[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[s] = v4d{0.0, 0.0, 0.0, 0.0};
    if (TI[s] >= 0) {
      const double* const tb = S.u.stage + (TI[s] * (TI[s] + 1) / 2 + TJ[s]) * TS + 17 * lc + rq;
#pragma unroll
      for (int g = 0; g < 4; ++g) t[s][g] = tb[4 * g];
    }
  }
  __syncthreads();  // staging is dead: the sweep buffers may overwrite it
  // publish the pivot columns of step sn (pivots 4 sn .. 4 sn + 3) from this wave's tiles: column form from the tiles of
  // that block column, row form (symmetry) from the tiles of that block row left of the diagonal
  auto publish = [&](auto gsc, int Kn, int par) __attribute__((always_inline)) {
    constexpr int GS = decltype(gsc)::value;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (TI[s] < 0) continue;
      if (TJ[s] == Kn) {
        if ((lc >> 2) == GS) {
          double* const cb = S.u.w.Cb[par][lc & 3] + 16 * TI[s] + rq;
#pragma unroll
          for (int g = 0; g < 4; ++g) cb[4 * g] = t[s][g];
        }
      } else if (TI[s] == Kn) {  // (TJ < Kn)
        S.u.w.Cb[par][rq][16 * TJ[s] + lc] = t[s][GS];
      }
    }
  };
  bool notpd = false;
  publish(std::integral_constant<int, 0>{}, 0, 0);
  const int nsteps = (n + 3) >> 2;
#pragma unroll 1
  for (int Kb = 0; Kb < 4; ++Kb) {
    StaticFor<0, 4>::run([&](auto gsc) __attribute__((always_inline)) {
      constexpr int GS = decltype(gsc)::value;
      constexpr int GSN = (GS + 1) & 3;
      const int st = 4 * Kb + GS;
      if (st < nsteps) {
        const int par = st & 1, k0 = 4 * st;
        __syncthreads();  // (A) the pivot columns of this step are in Cb[par]
        if (wv == 0) {
          // ---- pivot-block duty: P^-1 (4 x 4, through the Schur complement of its leading 2 x 2 block), F = C P^-1
          const double* const cb = S.u.w.Cb[par][0];
          const double c0 = cb[i], c1 = cb[NP + i], c2 = cb[2 * NP + i], c3 = cb[3 * NP + i];
          // P[a][b] = C_b[k0 + a]  (uniform: broadcast reads)
          const double a00 = cb[k0], a10 = cb[k0 + 1], b00 = cb[k0 + 2], b10 = cb[k0 + 3];
          const double a11 = cb[NP + k0 + 1], b01 = cb[NP + k0 + 2], b11 = cb[NP + k0 + 3];
          const double d00 = cb[2 * NP + k0 + 2], d10 = cb[2 * NP + k0 + 3], d11 = cb[3 * NP + k0 + 3];
          const double detA = __builtin_fma(a00, a11, -a10 * a10);
          const double rA = fast_rcp(detA);
          const double A00 = a11 * rA, A10 = -a10 * rA, A11 = a00 * rA;  // A^-1
          // W = B A^-1
          const double w00 = __builtin_fma(b00, A00, b01 * A10), w01 = __builtin_fma(b00, A10, b01 * A11);
          const double w10 = __builtin_fma(b10, A00, b11 * A10), w11 = __builtin_fma(b10, A10, b11 * A11);
          // S = D - W B^T
          const double s00 = d00 - __builtin_fma(w00, b00, w01 * b01);
          const double s10 = d10 - __builtin_fma(w10, b00, w11 * b01);
          const double s11 = d11 - __builtin_fma(w10, b10, w11 * b11);
          const double detS = __builtin_fma(s00, s11, -s10 * s10);
          const double rS = fast_rcp(detS);
          notpd |= !(a00 > 0.0) | !(detA > 0.0) | !(s00 > 0.0) | !(detS > 0.0);
          const double S00 = s11 * rS, S10 = -s10 * rS, S11 = s00 * rS;  // S^-1 = the (2,2) block of P^-1
          // M = -(S^-1 W) = the (2,1) block
          const double m00 = -__builtin_fma(S00, w00, S10 * w10), m01 = -__builtin_fma(S00, w01, S10 * w11);
          const double m10 = -__builtin_fma(S10, w00, S11 * w10), m11 = -__builtin_fma(S10, w01, S11 * w11);
          // N = A^-1 - W^T M = the (1,1) block
          const double n00 = A00 - __builtin_fma(w00, m00, w10 * m10);
          const double n10 = A10 - __builtin_fma(w01, m00, w11 * m10);
          const double n11 = A11 - __builtin_fma(w01, m01, w11 * m11);
          // P^-1 = [[n00 n10 m00 m10], [n10 n11 m01 m11], [m00 m01 S00 S10], [m10 m11 S10 S11]]  (symmetric)
          // F_i = C_i P^-1
          double f0 = __builtin_fma(c0, n00, __builtin_fma(c1, n10, __builtin_fma(c2, m00, c3 * m10)));
          double f1 = __builtin_fma(c0, n10, __builtin_fma(c1, n11, __builtin_fma(c2, m01, c3 * m11)));
          double f2 = __builtin_fma(c0, m00, __builtin_fma(c1, m01, __builtin_fma(c2, S00, c3 * S10)));
          double f3 = __builtin_fma(c0, m10, __builtin_fma(c1, m11, __builtin_fma(c2, S10, c3 * S11)));
          double g0 = f0, g1 = f1, g2 = f2, g3 = f3;  // Fmod
          const int ar = i - k0;
          if (ar >= 0 && ar < 4) {
            asm volatile("" ::: "memory");
            // pivot row a: F <- -P^-1[a][:],  Fmod <- e_a - P^-1[a][:]
            const double p0 = ar == 0 ? n00 : (ar == 1 ? n10 : (ar == 2 ? m00 : m10));
            const double p1 = ar == 0 ? n10 : (ar == 1 ? n11 : (ar == 2 ? m01 : m11));
            const double p2 = ar == 0 ? m00 : (ar == 1 ? m01 : (ar == 2 ? S00 : S10));
            const double p3 = ar == 0 ? m10 : (ar == 1 ? m11 : (ar == 2 ? S10 : S11));
            f0 = -p0; f1 = -p1; f2 = -p2; f3 = -p3;
            g0 = (ar == 0 ? 1.0 : 0.0) - p0;
            g1 = (ar == 1 ? 1.0 : 0.0) - p1;
            g2 = (ar == 2 ? 1.0 : 0.0) - p2;
            g3 = (ar == 3 ? 1.0 : 0.0) - p3;
          }
          S.u.w.Ft[0][i] = f0; S.u.w.Ft[1][i] = f1; S.u.w.Ft[2][i] = f2; S.u.w.Ft[3][i] = f3;
          S.u.w.nF[0][i] = -g0; S.u.w.nF[1][i] = -g1; S.u.w.nF[2][i] = -g2; S.u.w.nF[3][i] = -g3;
        }
        __syncthreads();  // (B) F is up
        // ---- every tile: T += (-Fmod_I) C_J^T, one matrix instruction; the tiles of the pivot block column then take F
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (TI[s] < 0) continue;
          const double aop = S.u.w.nF[rq][16 * TI[s] + lc];
          const double bop = S.u.w.Cb[par][rq][16 * TJ[s] + lc];
          t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, t[s], 0, 0, 0);
          if (TJ[s] == Kb && (lc >> 2) == GS) {
            const double* const ft = S.u.w.Ft[lc & 3] + 16 * TI[s] + rq;
#pragma unroll
            for (int g = 0; g < 4; ++g) t[s][g] = ft[4 * g];
          }
        }
        // ---- the next step's pivot columns
        if (st + 1 < nsteps) publish(std::integral_constant<int, GSN>{}, GS == 3 ? Kb + 1 : Kb, par ^ 1);
      }
    });
  }
  // ---- the inverse leaves the registers: packed lower triangle in LDS
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (TI[s] < 0) continue;
    const int cc = 16 * TJ[s] + lc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int r = 16 * TI[s] + rq + 4 * g;
      if (cc <= r && r < n) S.u.w.Hp[r * (r + 1) / 2 + cc] = -t[s][g];
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
  if ((int)blockIdx.x < nout)
    for (int k = tid; k < n * (n + 1) / 2; k += 256) out[(size_t)blockIdx.x * (NP * (NP + 1) / 2) + k] = S.u.w.Hp[k];
  if (notpd && tid == 0) clk[blockIdx.x] = -1;
}


// ------------------------------------------------------------------ C: B with the pivot-block duty ONE STEP AHEAD
// Wave 0 holds no tile.  While the tile waves apply step s (one matrix instruction per tile), it brings the NEXT step's four
// pivot columns up to date itself (lane = row: c' = c_old - Fmod_s C_s[pivot rows of s+1], 16 multiply-adds), inverts their
// 4 x 4 block and forms F for step s + 1: the 4 x 4 inverse is off the tile waves' path, and a step has ONE barrier.
// The tile waves publish the (pre-update) pivot columns two steps ahead.  Tiles dealt along the anti-diagonals so that the
// tiles of a block column (the ones that publish, and take F) belong to different waves.
struct SmemC {
  union {
    double stage[10 * TS];
    struct {
      double Cn[2][4][NP];  // pivot columns of the step, up to date (by parity): the B operand
      double Co[2][4][NP];  // pivot columns as the tiles hold them two steps earlier (by parity)
      double nF[2][4][NP];  // -Fmod: the A operand
      double Ft[2][4][NP];  // F: what the pivot columns become
    } w;
    double Hp[NP * (NP + 1) / 2];
  } u;
  char pad[LDS_PAD - 10 * TS * 8];
};

__global__ __launch_bounds__(256, 5) void sweep_c(long long* clk, double* out, int n, int nout) {
  __shared__ SmemC S;
  const int tid = threadIdx.x, lane = tid & 63, i = tid % NP, c = tid / NP;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, rq = lane >> 4;
  const int seed = blockIdx.x & 1023;
  double a[CW];
#pragma unroll
  for (int jj = 0; jj < CW; ++jj) a[jj] = hmat(i, c * CW + jj, n, seed);
  if (tid == 0) S.pad[0] = 0;
  __syncthreads();
  const long long t0 = clock64();
  constexpr int NS = 4;
  int TI[NS], TJ[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { TI[s] = -1; TJ[s] = 0; }
  if (wv == 1) { TI[0] = 0; TJ[0] = 0; TI[1] = 1; TJ[1] = 1; TI[2] = 2; TJ[2] = 2; TI[3] = 3; TJ[3] = 3; }
  else if (wv == 2) { TI[0] = 1; TJ[0] = 0; TI[1] = 2; TJ[1] = 1; TI[2] = 3; TJ[2] = 2; }
  else if (wv == 3) { TI[0] = 2; TJ[0] = 0; TI[1] = 3; TJ[1] = 1; TI[2] = 3; TJ[2] = 0; }
  const int nblk = (n + 15) >> 4;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (TI[s] >= nblk) TI[s] = -1;
  {
    const int I = i >> 4, r = i & 15;
    if (c <= I) {
      double* const tb = S.u.stage + (I * (I + 1) / 2 + c) * TS + r;
#pragma unroll
      for (int jj = 0; jj < CW; ++jj) tb[17 * jj] = a[jj];
    }
  }
  __syncthreads();
  v4d t[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    t[s] = v4d{0.0, 0.0, 0.0, 0.0};
    if (TI[s] >= 0) {
      const double* const tb = S.u.stage + (TI[s] * (TI[s] + 1) / 2 + TJ[s]) * TS + 17 * lc + rq;
#pragma unroll
      for (int g = 0; g < 4; ++g) t[s][g] = tb[4 * g];
    }
  }
  __syncthreads();
  const int nsteps = (n + 3) >> 2;
  auto publish = [&](auto gsc, int Kn, int par) __attribute__((always_inline)) {
    constexpr int GS = decltype(gsc)::value;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (TI[s] < 0) continue;
      if (TJ[s] == Kn) {
        if ((lc >> 2) == GS) {
          double* const cb = S.u.w.Co[par][lc & 3] + 16 * TI[s] + rq;
#pragma unroll
          for (int g = 0; g < 4; ++g) cb[4 * g] = t[s][g];
        }
      } else if (TI[s] == Kn) {
        S.u.w.Co[par][rq][16 * TJ[s] + lc] = t[s][GS];
      }
    }
  };
  bool notpd = false;
  double nfp0 = 0.0, nfp1 = 0.0, nfp2 = 0.0, nfp3 = 0.0;  // wave 0: -Fmod of the step in flight (lane = row)
  // the pivot-block duty of step st (wave 0): pivot columns brought up to date, P^-1, F
  auto duty = [&](int st, bool update) __attribute__((always_inline)) {
    const int par = st & 1, k0 = 4 * st;
    const double* const co = S.u.w.Co[par][0];
    double c0 = co[i], c1 = co[NP + i], c2 = co[2 * NP + i], c3 = co[3 * NP + i];
    if (update) {
      // c_q' += sum_q (-Fmod_prev[i][q]) C_prev[q][k0 + q']
      const double* const cp = S.u.w.Cn[par ^ 1][0] + k0;
      const F64x2 u0a = ld2(cp), u0b = ld2(cp + 2), u1a = ld2(cp + NP), u1b = ld2(cp + NP + 2);
      const F64x2 u2a = ld2(cp + 2 * NP), u2b = ld2(cp + 2 * NP + 2), u3a = ld2(cp + 3 * NP), u3b = ld2(cp + 3 * NP + 2);
      c0 = __builtin_fma(nfp0, u0a.x, __builtin_fma(nfp1, u1a.x, __builtin_fma(nfp2, u2a.x, __builtin_fma(nfp3, u3a.x, c0))));
      c1 = __builtin_fma(nfp0, u0a.y, __builtin_fma(nfp1, u1a.y, __builtin_fma(nfp2, u2a.y, __builtin_fma(nfp3, u3a.y, c1))));
      c2 = __builtin_fma(nfp0, u0b.x, __builtin_fma(nfp1, u1b.x, __builtin_fma(nfp2, u2b.x, __builtin_fma(nfp3, u3b.x, c2))));
      c3 = __builtin_fma(nfp0, u0b.y, __builtin_fma(nfp1, u1b.y, __builtin_fma(nfp2, u2b.y, __builtin_fma(nfp3, u3b.y, c3))));
    }
    // P[a][b] = c_b at lane k0 + a
    const double a00 = readlane_f64(c0, k0), a10 = readlane_f64(c0, k0 + 1), b00 = readlane_f64(c0, k0 + 2), b10 = readlane_f64(c0, k0 + 3);
    const double a11 = readlane_f64(c1, k0 + 1), b01 = readlane_f64(c1, k0 + 2), b11 = readlane_f64(c1, k0 + 3);
    const double d00 = readlane_f64(c2, k0 + 2), d10 = readlane_f64(c2, k0 + 3), d11 = readlane_f64(c3, k0 + 3);
    const double detA = __builtin_fma(a00, a11, -a10 * a10);
    const double rA = fast_rcp(detA);
    const double A00 = a11 * rA, A10 = -a10 * rA, A11 = a00 * rA;
    const double w00 = __builtin_fma(b00, A00, b01 * A10), w01 = __builtin_fma(b00, A10, b01 * A11);
    const double w10 = __builtin_fma(b10, A00, b11 * A10), w11 = __builtin_fma(b10, A10, b11 * A11);
    const double s00 = d00 - __builtin_fma(w00, b00, w01 * b01);
    const double s10 = d10 - __builtin_fma(w10, b00, w11 * b01);
    const double s11 = d11 - __builtin_fma(w10, b10, w11 * b11);
    const double detS = __builtin_fma(s00, s11, -s10 * s10);
    const double rS = fast_rcp(detS);
    notpd |= !(a00 > 0.0) | !(detA > 0.0) | !(s00 > 0.0) | !(detS > 0.0);
    const double S00 = s11 * rS, S10 = -s10 * rS, S11 = s00 * rS;
    const double m00 = -__builtin_fma(S00, w00, S10 * w10), m01 = -__builtin_fma(S00, w01, S10 * w11);
    const double m10 = -__builtin_fma(S10, w00, S11 * w10), m11 = -__builtin_fma(S10, w01, S11 * w11);
    const double n00 = A00 - __builtin_fma(w00, m00, w10 * m10);
    const double n10 = A10 - __builtin_fma(w01, m00, w11 * m10);
    const double n11 = A11 - __builtin_fma(w01, m01, w11 * m11);
    double f0 = __builtin_fma(c0, n00, __builtin_fma(c1, n10, __builtin_fma(c2, m00, c3 * m10)));
    double f1 = __builtin_fma(c0, n10, __builtin_fma(c1, n11, __builtin_fma(c2, m01, c3 * m11)));
    double f2 = __builtin_fma(c0, m00, __builtin_fma(c1, m01, __builtin_fma(c2, S00, c3 * S10)));
    double f3 = __builtin_fma(c0, m10, __builtin_fma(c1, m11, __builtin_fma(c2, S10, c3 * S11)));
    double g0 = f0, g1 = f1, g2 = f2, g3 = f3;
    const int ar = i - k0;
    if (ar >= 0 && ar < 4) {
      asm volatile("" ::: "memory");
      const double p0 = ar == 0 ? n00 : (ar == 1 ? n10 : (ar == 2 ? m00 : m10));
      const double p1 = ar == 0 ? n10 : (ar == 1 ? n11 : (ar == 2 ? m01 : m11));
      const double p2 = ar == 0 ? m00 : (ar == 1 ? m01 : (ar == 2 ? S00 : S10));
      const double p3 = ar == 0 ? m10 : (ar == 1 ? m11 : (ar == 2 ? S10 : S11));
      f0 = -p0; f1 = -p1; f2 = -p2; f3 = -p3;
      g0 = (ar == 0 ? 1.0 : 0.0) - p0;
      g1 = (ar == 1 ? 1.0 : 0.0) - p1;
      g2 = (ar == 2 ? 1.0 : 0.0) - p2;
      g3 = (ar == 3 ? 1.0 : 0.0) - p3;
    }
    nfp0 = -g0; nfp1 = -g1; nfp2 = -g2; nfp3 = -g3;
    double* const cn = S.u.w.Cn[par][0];
    cn[i] = c0; cn[NP + i] = c1; cn[2 * NP + i] = c2; cn[3 * NP + i] = c3;
    double* const nf = S.u.w.nF[par][0];
    nf[i] = nfp0; nf[NP + i] = nfp1; nf[2 * NP + i] = nfp2; nf[3 * NP + i] = nfp3;
    double* const ft = S.u.w.Ft[par][0];
    ft[i] = f0; ft[NP + i] = f1; ft[2 * NP + i] = f2; ft[3 * NP + i] = f3;
  };
  // prologue: the tile waves publish the pivot columns of steps 0 and 1 as they stand; wave 0 does step 0's duty
  if (wv != 0) {
    publish(std::integral_constant<int, 0>{}, 0, 0);
    if (nsteps > 1) publish(std::integral_constant<int, 1>{}, 0, 1);
  }
  __syncthreads();
  if (wv == 0) duty(0, false);
#pragma unroll 1
  for (int Kb = 0; Kb < 4; ++Kb) {
    StaticFor<0, 4>::run([&](auto gsc) __attribute__((always_inline)) {
      constexpr int GS = decltype(gsc)::value;
      constexpr int GS2 = (GS + 2) & 3;
      const int st = 4 * Kb + GS;
      if (st < nsteps) {
        const int par = st & 1;
        __syncthreads();  // F, C of step st and the stale pivot columns of step st + 1 are up
        if (wv == 0) {
          if (st + 1 < nsteps) duty(st + 1, true);
        } else {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (TI[s] < 0) continue;
            const double aop = S.u.w.nF[par][rq][16 * TI[s] + lc];
            const double bop = S.u.w.Cn[par][rq][16 * TJ[s] + lc];
            t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, t[s], 0, 0, 0);
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (TI[s] < 0) continue;
            if (TJ[s] == Kb && (lc >> 2) == GS) {
              const double* const ft = S.u.w.Ft[par][lc & 3] + 16 * TI[s] + rq;
#pragma unroll
              for (int g = 0; g < 4; ++g) t[s][g] = ft[4 * g];
            }
          }
          if (st + 2 < nsteps) publish(std::integral_constant<int, GS2>{}, GS >= 2 ? Kb + 1 : Kb, par);
        }
      }
    });
  }
  __syncthreads();  // the sweep buffers are dead: the packed inverse may overwrite them
  if (wv != 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (TI[s] < 0) continue;
      const int cc = 16 * TJ[s] + lc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * TI[s] + rq + 4 * g;
        if (cc <= r && r < n) S.u.Hp[r * (r + 1) / 2 + cc] = -t[s][g];
      }
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
  if ((int)blockIdx.x < nout)
    for (int k = tid; k < n * (n + 1) / 2; k += 256) out[(size_t)blockIdx.x * (NP * (NP + 1) / 2) + k] = S.u.Hp[k];
  if (__syncthreads_or(notpd ? 1 : 0) && tid == 0) clk[blockIdx.x] = -1;
}

double check(const std::vector<double>& hp, int n, int seed) {
  // max |H Hinv - I|
  std::vector<double> H(n * n), X(n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      H[i * n + j] = hmat(i, j, n, seed);
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      X[i * n + j] = hp[hi * (hi + 1) / 2 + lo];
    }
  double worst = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += H[i * n + k] * X[k * n + j];
      worst = std::max(worst, std::fabs(s - (i == j ? 1.0 : 0.0)));
    }
  return worst;
}
}  // namespace

template <class K>
void bench(const char* name, K kern, int n, long long* dclk, double* dout, std::vector<double>* keep) {
  constexpr int NOUT = 8, NH = NP * (NP + 1) / 2;
  for (int grid : {1280, 8192, 16384}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dclk, dout, n, NOUT);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dclk, dout, n, NOUT);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), dclk, 8 * grid, hipMemcpyDeviceToHost);
    int bad = 0;
    for (auto v : h) bad += v < 0;
    std::sort(h.begin(), h.end());
    printf("%-10s n=%d grid=%5d  kernel %8.2f us  (%.3f us/robot-slot)  stage cycles med %lld  p90 %lld  max %lld  notpd %d\n", name, n, grid,
           ms * 1e3 / reps, ms * 1e3 / reps / grid * 1280, h[grid / 2], h[grid * 9 / 10], h[grid - 1], bad);
  }
  std::vector<double> ho((size_t)NOUT * NH);
  hipMemcpy(ho.data(), dout, ho.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int b = 0; b < NOUT; ++b) {
    std::vector<double> one(ho.begin() + (size_t)b * NH, ho.begin() + (size_t)(b + 1) * NH);
    worst = std::max(worst, check(one, n, b & 1023));
  }
  printf("%-10s n=%d max |H Hinv - I| over %d matrices = %.3e\n", name, n, NOUT, worst);
  if (keep) *keep = ho;
}

int main() {
  long long* dclk;
  double* dout;
  hipMalloc(&dclk, 8 * 16384);
  hipMalloc(&dout, 8 * 8 * (NP * (NP + 1) / 2));
  for (int n : {60, 48, 57}) {
    std::vector<double> ka, kb;
    hipMemset(dout, 0, 8 * 8 * (NP * (NP + 1) / 2));
    bench("A:shipped", sweep_a, n, dclk, dout, &ka);
    hipMemset(dout, 0, 8 * 8 * (NP * (NP + 1) / 2));
    std::vector<double> ka2;
    hipMemset(dout, 0, 8 * 8 * (NP * (NP + 1) / 2));
    bench("A2:symm", sweep_a2, n, dclk, dout, &ka2);
    {
      double da = 0.0;
      for (size_t k = 0; k < ka.size(); ++k) da = std::max(da, std::fabs(ka[k] - ka2[k]));
      printf("n=%d  max |A - A2| = %.3e\n", n, da);
    }
    {
      std::vector<double> ka3;
      hipMemset(dout, 0, 8 * 8 * (NP * (NP + 1) / 2));
      bench("A3:rotate", sweep_a3, n, dclk, dout, &ka3);
      double da = 0.0;
      for (size_t k = 0; k < ka.size(); ++k) da = std::max(da, std::fabs(ka[k] - ka3[k]));
      printf("n=%d  max |A - A3| = %.3e\n", n, da);
    }
    bench("B:mfma", sweep_b<0>, n, dclk, dout, &kb);
    std::vector<double> kc;
    hipMemset(dout, 0, 8 * 8 * (NP * (NP + 1) / 2));
    bench("C:mfma+1", sweep_c, n, dclk, dout, &kc);
    {
      double dc = 0.0;
      for (size_t k = 0; k < ka.size(); ++k) dc = std::max(dc, std::fabs(ka[k] - kc[k]));
      printf("n=%d  max |A - C| = %.3e\n", n, dc);
    }
    double d = 0.0, m = 0.0;
    for (size_t k = 0; k < ka.size(); ++k) {
      d = std::max(d, std::fabs(ka[k] - kb[k]));
      m = std::max(m, std::fabs(ka[k]));
    }
    printf("n=%d  max |A - B| = %.3e (largest entry %.3e)\n", n, d, m);
  }
  return 0;
}
