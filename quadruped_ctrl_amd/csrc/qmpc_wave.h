// qmpc_wave.h -- wave-level helpers shared by the solve kernels (qmpc_kernels.hip) and the decoupled
// active-set engine (qmpc_engine.hip): DPP reductions, readlane / lane-element access, reciprocals,
// the compile-time loop, and the sparse form of the friction / force-limit rows.  gfx950 only.
#ifndef QMPC_WAVE_H
#define QMPC_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace {
typedef __attribute__((address_space(1))) double GlobalF64;  // a double known to live in global memory (global_load, not flat_load)
typedef double __attribute__((ext_vector_type(2))) F64x2;
typedef __attribute__((address_space(1))) F64x2 GlobalF64x2;
// two adjacent doubles with one 16-byte access (p 16-byte aligned)
__device__ __forceinline__ F64x2 ld2(const double* p) { return *reinterpret_cast<const F64x2*>(p); }
__device__ __forceinline__ F64x2 ld2(const GlobalF64* p) { return *reinterpret_cast<const GlobalF64x2*>(p); }
__device__ __forceinline__ void st2(double* p, double a, double b) { *reinterpret_cast<F64x2*>(p) = F64x2{a, b}; }
__device__ __forceinline__ void st2(GlobalF64* p, double a, double b) { *reinterpret_cast<GlobalF64x2*>(p) = F64x2{a, b}; }

constexpr int WAVE = 64;
// ----------------------------------------------------------------- wave helpers
// DPP control words (gfx9): row_shr:n = 0x110+n, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143.  After the six steps lane 63 holds the reduction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned identity, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// max over the 64 lanes of a wave, result uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  unsigned t;
  t = dpp_u32<0x111, 0xf>(0u, v); v = v > t ? v : t;
  t = dpp_u32<0x112, 0xf>(0u, v); v = v > t ? v : t;
  t = dpp_u32<0x114, 0xf>(0u, v); v = v > t ? v : t;
  t = dpp_u32<0x118, 0xf>(0u, v); v = v > t ? v : t;
  t = dpp_u32<0x142, 0xa>(0u, v); v = v > t ? v : t;
  t = dpp_u32<0x143, 0xc>(0u, v); v = v > t ? v : t;
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// min over the 64 lanes of a wave, result uniform
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  unsigned t;
  t = dpp_u32<0x111, 0xf>(~0u, v); v = v < t ? v : t;
  t = dpp_u32<0x112, 0xf>(~0u, v); v = v < t ? v : t;
  t = dpp_u32<0x114, 0xf>(~0u, v); v = v < t ? v : t;
  t = dpp_u32<0x118, 0xf>(~0u, v); v = v < t ? v : t;
  t = dpp_u32<0x142, 0xa>(~0u, v); v = v < t ? v : t;
  t = dpp_u32<0x143, 0xc>(~0u, v); v = v < t ? v : t;
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// min over the wave of NON-NEGATIVE doubles (bit pattern order == value order):
// two 32-bit reductions (high words, then the low words of the lanes that tie on
// the high word) instead of one 64-bit compare/select ladder
__device__ __forceinline__ double wave_min_pos_f64(double x) {
  const unsigned long long v = (unsigned long long)__double_as_longlong(x);
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mhi = wave_min_u32(hi);
  const unsigned mlo = wave_min_u32(hi == mhi ? lo : ~0u);
  return __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
}
// max over the wave of NON-NEGATIVE doubles: the min reduction on the bitwise complement (larger
// value <=> smaller complement)
__device__ __forceinline__ double wave_max_pos_f64(double x) {
  const unsigned long long v = ~(unsigned long long)__double_as_longlong(x);
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mhi = wave_min_u32(hi);
  const unsigned mlo = wave_min_u32(hi == mhi ? lo : ~0u);
  return __longlong_as_double((long long)~(((unsigned long long)mhi << 32) | mlo));
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// element q (uniform) of a small per-lane register array without dynamic indexing
template <int KW, typename T>
__device__ __forceinline__ T pick(const T (&arr)[KW], int q) {
  T v = arr[0];
#pragma unroll
  for (int k = 1; k < KW; ++k) v = (q == k) ? arr[k] : v;
  return v;
}
// element idx (wave-uniform) of a vector stored 64 entries per register: arr[idx >> 6] at lane
// idx & 63.  One readlane per register and a scalar select -- a select over the REGISTERS would be
// turned into a dynamically indexed private array (scratch) by the compiler
template <int KW>
__device__ __forceinline__ double lane_elem(const double (&arr)[KW], int idx) {
  const int q = idx >> 6, l = idx & 63;
  double out = readlane_f64(arr[0], l);
#pragma unroll
  for (int k = 1; k < KW; ++k) {
    const double t = readlane_f64(arr[k], l);
    out = (q == k) ? t : out;
  }
  return out;
}
template <int KW>
__device__ __forceinline__ int lane_elem(const int (&arr)[KW], int idx) {
  const int q = idx >> 6, l = idx & 63;
  int out = __builtin_amdgcn_readlane(arr[0], l);
#pragma unroll
  for (int k = 1; k < KW; ++k) {
    const int t = __builtin_amdgcn_readlane(arr[k], l);
    out = (q == k) ? t : out;
  }
  return out;
}
// 1/d to full double precision: v_rcp_f64 seed + two Newton steps
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  return x;
}

// compile-time loop: f(integral_constant<int, R>) for R = 0 .. N-1, so that
// register-array indices derived from R are constant expressions
template <int R, int N>
struct StaticFor {
  template <class F>
  static __device__ __forceinline__ void run(F&& f) {
    f(std::integral_constant<int, R>{});
    StaticFor<R + 1, N>::run(f);
  }
};
template <int N>
struct StaticFor<N, N> {
  template <class F>
  static __device__ __forceinline__ void run(F&&) {}
};

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
// As a sparse row c = a1 e_{j1} + a2 e_{j2}:
__device__ __forceinline__ void con_coefs(int e, double mi, int& j1, int& j2, double& a1, double& a2) {
  const int slot = e / 5, ty = e - 5 * slot, j0 = 3 * slot;
  j2 = j0 + 2;
  if (ty < 4) {
    j1 = j0 + (ty >> 1);
    a1 = (ty & 1) ? -mi : mi;
    a2 = 1.0;
  } else {
    j1 = j0 + 2;
    a1 = -1.0;
    a2 = 0.0;
  }
}

}  // namespace

#endif
