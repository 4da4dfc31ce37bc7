// Isolated timing of the rank-2 Gauss-Jordan sweep step (same code shape as
// qmpc_kernels.hip stage 3) with ablations, to see where the cycles go.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>
template <int R, int N> struct StaticFor {
  template <class F> static __device__ __forceinline__ void run(F&& f) { f(std::integral_constant<int, R>{}); StaticFor<R + 1, N>::run(f); }
};
template <int N> struct StaticFor<N, N> { template <class F> static __device__ __forceinline__ void run(F&&) {} };
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, x, 1.0); x = __builtin_fma(x, e, x);
  e = __builtin_fma(-d, x, 1.0); x = __builtin_fma(x, e, x);
  return x;
}
constexpr int NP = 64, CW = 16;
// ABL bit0: no rcp (constant)  bit1: no barrier  bit2: no LDS column reads (use constants)  bit3: no fma update
template <int ABL>
__global__ __launch_bounds__(256, 4) void k(long long* out, double* sink, int n) {
  __shared__ __attribute__((aligned(16))) double colbuf[2][2][NP + 2];
  const int tid = threadIdx.x, i = tid % NP, c = tid / NP;
  double a[CW];
  for (int jj = 0; jj < CW; ++jj) { int j = c * CW + jj; a[jj] = (i == j) ? 4.0 + 0.01 * i : 1.0 / (1.0 + i + j); }
  if (c == 0) { colbuf[0][0][i] = a[0]; colbuf[0][1][i] = a[1]; }
  __syncthreads();
  long long t0 = clock64();
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    StaticFor<0, CW / 2>::run([&](auto pc) __attribute__((always_inline)) {
      constexpr int r0 = 2 * decltype(pc)::value, r1 = r0 + 1;
      constexpr int rn0 = (r0 + 2 < CW) ? r0 + 2 : 0, rn1 = rn0 + 1;
      const int k0 = kb * CW + r0, k1 = k0 + 1;
      if (k0 < n) {
        const int m = k0 >> 1;
        const double* cb0 = colbuf[m & 1][0];
        const double* cb1 = colbuf[m & 1][1];
        double d0 = (ABL & 4) ? 4.0 : cb0[k0];
        const double e = (ABL & 4) ? 0.1 : cb0[k1];
        const double d1p = (ABL & 4) ? 4.0 : cb1[k1];
        const double c0i = (ABL & 4) ? 0.2 : cb0[i], c1i = (ABL & 4) ? 0.3 : cb1[i];
        const double dinv0 = (ABL & 1) ? 0.25 * d0 : fast_rcp(d0);
        const double g = e * dinv0;
        double d1 = __builtin_fma(-e, g, d1p);
        const double dinv1 = (ABL & 1) ? 0.25 * d1 : fast_rcp(d1);
        const bool p0 = (i == k0), p1 = (i == k1);
        const double f0 = (p0 ? d0 - 1.0 : c0i) * dinv0;
        const double c1pi = __builtin_fma(-f0, e, c1i);
        const double f1 = (p1 ? d1 - 1.0 : c1pi) * dinv1;
        const double f0g = __builtin_fma(-f1, g, f0);
        if (!(ABL & 8)) {
#pragma unroll
          for (int jj = 0; jj < CW; ++jj) {
            const double x0 = (ABL & 4) ? 0.01 * jj : cb0[c * CW + jj];
            const double x1 = (ABL & 4) ? 0.02 * jj : cb1[c * CW + jj];
            a[jj] = __builtin_fma(-f1, x1, __builtin_fma(-f0g, x0, a[jj]));
          }
        }
        if (c == kb) {
          a[r0] = __builtin_fma(-f1, g, p0 ? -dinv0 : f0);
          a[r1] = p1 ? -dinv1 : f1;
        }
        const int kbn = (r0 + 2 < CW) ? kb : kb + 1;
        if (k0 + 2 < n && c == kbn) { colbuf[(m + 1) & 1][0][i] = a[rn0]; colbuf[(m + 1) & 1][1][i] = a[rn1]; }
        if (!(ABL & 2)) __syncthreads();
      }
    });
  }
  long long t1 = clock64();
  double s = 0; for (int jj = 0; jj < CW; ++jj) s += a[jj];
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 256 + tid] = s;
}
template <int ABL> void run(const char* name, long long* d, double* s) {
  for (int grid : {64, 1024}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<ABL>), dim3(grid), dim3(256), 0, 0, d, s, 60);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-40s grid %4d: %7.0f cycles/pair (total %7.0f)\n", name, grid, (double)h[grid / 2] / 30, (double)h[grid / 2]);
  }
}
int main() {
  long long* d; double* s;
  hipMalloc(&d, 8 * 2048); hipMalloc(&s, 8 * 2048 * 256);
  run<0>("full", d, s);
  run<1>("no rcp", d, s);
  run<2>("no barrier", d, s);
  run<4>("no LDS reads", d, s);
  run<8>("no fma update", d, s);
  run<1 | 8>("no rcp, no fma", d, s);
  run<1 | 2 | 4 | 8>("nothing but control flow + writes", d, s);
  return 0;
}
