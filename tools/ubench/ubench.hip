// Micro-benchmarks (cycles via s_memtime) for the latency model of the solve
// kernel: barrier+LDS round trip, fp64 rcp/fma chains, readlane, LDS broadcast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, x, 1.0); x = __builtin_fma(x, e, x);
  e = __builtin_fma(-d, x, 1.0); x = __builtin_fma(x, e, x);
  return x;
}
constexpr int N = 256;
__global__ void k(long long* out, double* sink, int mode) {
  __shared__ double buf[2][80];
  const int tid = threadIdx.x;
  double acc = 1.0 + tid * 1e-9, b = 1.0000001, c2 = 0.999999;
  double a[16];
  for (int j = 0; j < 16; ++j) a[j] = 1.0 + j;
  buf[0][tid & 63] = acc; buf[1][tid & 63] = acc;
  __syncthreads();
  long long t0 = clock64();
  if (mode == 0) {  // barrier + LDS write + read round trip (one wave writes, all read)
    for (int it = 0; it < N; ++it) {
      if ((tid >> 6) == (it & 3)) buf[it & 1][tid & 63] = acc;
      __syncthreads();
      acc += buf[it & 1][(tid + 1) & 63];
    }
  } else if (mode == 1) {  // dependent fast_rcp chain
    for (int it = 0; it < N; ++it) acc = fast_rcp(acc) + 0.5;
  } else if (mode == 2) {  // dependent fma chain
    for (int it = 0; it < N; ++it) acc = __builtin_fma(acc, b, c2);
  } else if (mode == 3) {  // 16 independent fma per iteration (2 dependent rounds)
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(-b, acc, __builtin_fma(-c2, b, a[j]));
      acc += 1e-9;
    }
    for (int j = 0; j < 16; ++j) acc += a[j];
  } else if (mode == 4) {  // bare barrier
    for (int it = 0; it < N; ++it) __syncthreads();
  } else if (mode == 5) {  // LDS broadcast read of 16 doubles + dependent use
    for (int it = 0; it < N; ++it) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += buf[it & 1][(j + (int)acc) & 63];
      acc = s * 1e-3 + 1.0;
    }
  } else if (mode == 6) {  // readlane f64 dependent
    for (int it = 0; it < N; ++it) {
      unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(acc), it & 63);
      unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(((unsigned long long)__double_as_longlong(acc)) >> 32), it & 63);
      acc = acc * 0.5 + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) * 0.5;
    }
  } else if (mode == 7) {  // dependent LDS write->read same wave (no barrier)
    for (int it = 0; it < N; ++it) {
      buf[0][tid & 63] = acc;
      __builtin_amdgcn_wave_barrier();
      acc = buf[0][(tid + 1) & 63] + 1e-9;
    }
  } else if (mode == 8) {  // v_rcp_f64 alone (dependent)
    for (int it = 0; it < N; ++it) acc = __builtin_amdgcn_rcp(acc) + 0.5;
  } else if (mode == 9) {  // int ALU dependent chain
    int x = tid;
    for (int it = 0; it < N; ++it) x = x * 3 + (x >> 2);
    acc += x;
  }
  long long t1 = clock64();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = acc;
}
int main() {
  long long* d; double* s;
  hipMalloc(&d, 8 * 2048); hipMalloc(&s, 8 * 2048 * 256);
  const char* names[] = {"barrier+LDS wr/rd", "fast_rcp chain", "fma chain", "16 indep fma x2", "bare barrier",
                         "16 LDS bcast reads + dep", "readlane f64 dep", "LDS wr->rd same wave", "v_rcp_f64 + add", "int mul-add chain"};
  for (int grid : {64, 1024}) {
    printf("grid=%d blocks of 256 threads: cycles per iteration (median over blocks)\n", grid);
    for (int mode = 0; mode < 10; ++mode) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, s, mode);
      hipDeviceSynchronize();
      std::vector<long long> h(grid);
      hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      printf("  %-28s %8.1f\n", names[mode], (double)h[grid / 2] / N);
    }
  }
  return 0;
}
