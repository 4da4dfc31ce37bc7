// How long does the dispatcher take to start the 1024 workgroups of a one-round launch (256 threads, 39 KB of LDS: four
// per CU)?  Every workgroup records the constant-rate wall clock (100 MHz, one time base for the whole chip) when it starts.
//   hipcc --offload-arch=gfx950 -O2 dispatch_spread.hip -o dispatch_spread && ./dispatch_spread [blocks] [threads] [lds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long* out, int spin) {
  extern __shared__ char lds[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t0;
  while (wall_clock64() - t0 < (unsigned long long)spin) lds[threadIdx.x] = (char)spin;
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 256, ldsb = argc > 3 ? atoi(argv[3]) : 39000;
  unsigned long long* d;
  hipMalloc(&d, blocks * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  std::vector<unsigned long long> h(blocks);
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), ldsb, 0, d, 3000);  // 30 us
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    std::vector<double> s(blocks);
    for (int b = 0; b < blocks; ++b) s[b] = (h[b] - t0) * 0.01;  // us
    std::vector<double> q = s;
    std::sort(q.begin(), q.end());
    printf("rep %d: start of workgroup b after the first one (us): median %.2f  90%% %.2f  99%% %.2f  max %.2f | b=0 %.2f b=%d %.2f b=%d %.2f\n", rep,
           q[blocks / 2], q[blocks * 9 / 10], q[blocks * 99 / 100], q[blocks - 1], s[0], blocks / 2, s[blocks / 2], blocks - 1, s[blocks - 1]);
  }
  return 0;
}
