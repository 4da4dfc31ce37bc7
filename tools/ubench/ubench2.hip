// LDS read micro-benchmarks: cost of K ds_read_b128 per iteration in the access
// patterns the sweep uses (wave-uniform "broadcast" address vs per-lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int N = 256;
template <int K, int MODE>
__global__ void k(long long* out, double* sink) {
  __shared__ __attribute__((aligned(16))) double buf[2][264];
  const int tid = threadIdx.x;
  for (int j = tid; j < 2 * 264; j += blockDim.x) (&buf[0][0])[j] = 1.0 + 1e-6 * j;
  __syncthreads();
  double acc[2 * K];
  for (int j = 0; j < 2 * K; ++j) acc[j] = 0;
  const int c = tid >> 6;
  long long t0 = clock64();
  for (int it = 0; it < N; ++it) {
    const double* b = buf[it & 1];
    const int base = (MODE == 0) ? c * 32 : ((MODE == 1) ? (tid & 63) * 2 : c * 32);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const double2 v = *reinterpret_cast<const double2*>(&b[(base + 2 * j) % 256]);
      acc[2 * j] += v.x; acc[2 * j + 1] += v.y;
    }
    if (MODE == 2) __syncthreads();
  }
  long long t1 = clock64();
  double s = 0; for (int j = 0; j < 2 * K; ++j) s += acc[j];
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = s;
}
template <int K, int MODE> void run(const char* name, long long* d, double* s) {
  for (int grid : {64, 1024}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<K, MODE>), dim3(grid), dim3(256), 0, 0, d, s);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-44s grid %4d: %8.1f cycles/iter\n", name, grid, (double)h[grid / 2] / N);
  }
}
int main() {
  long long* d; double* s;
  hipMalloc(&d, 8 * 2048); hipMalloc(&s, 8 * 2048 * 256);
  run<16, 0>("16 x ds_read_b128 wave-uniform address", d, s);
  run<8, 0>(" 8 x ds_read_b128 wave-uniform address", d, s);
  run<4, 0>(" 4 x ds_read_b128 wave-uniform address", d, s);
  run<16, 1>("16 x ds_read_b128 per-lane addresses", d, s);
  run<8, 1>(" 8 x ds_read_b128 per-lane addresses", d, s);
  run<16, 2>("16 x ds_read_b128 uniform + barrier", d, s);
  return 0;
}
