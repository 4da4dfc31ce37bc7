// How long does the dispatcher take to start the 1024 workgroups of a one-round launch (256 threads, 39 KB of LDS: four
// per CU)?  Every workgroup records the constant-rate wall clock (100 MHz, one time base for the whole chip) when it starts.
//   hipcc --offload-arch=gfx950 -O2 dispatch_spread.hip -o dispatch_spread && ./dispatch_spread [blocks] [threads] [lds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long* out, unsigned* xout, int spin) {
  extern __shared__ char lds[];
  const unsigned long long t0 = wall_clock64();
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[blockIdx.x] = t0; xout[blockIdx.x] = xcc & 15u; }
  while (wall_clock64() - t0 < (unsigned long long)spin) lds[threadIdx.x] = (char)spin;
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 256, ldsb = argc > 3 ? atoi(argv[3]) : 39000;
  unsigned long long* d;
  hipMalloc(&d, blocks * 8);
  unsigned* dx;
  hipMalloc(&dx, blocks * 4);
  std::vector<unsigned> hx(blocks);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  std::vector<unsigned long long> h(blocks);
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), ldsb, 0, d, dx, 3000);  // 30 us
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), dx, blocks * 4, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    std::vector<double> s(blocks);
    for (int b = 0; b < blocks; ++b) s[b] = (h[b] - t0) * 0.01;  // us
    std::vector<double> q = s;
    std::sort(q.begin(), q.end());
    printf("rep %d: start of workgroup b after the first one (us): median %.2f  90%% %.2f  99%% %.2f  max %.2f | b=0 %.2f b=%d %.2f b=%d %.2f\n", rep,
           q[blocks / 2], q[blocks * 9 / 10], q[blocks * 99 / 100], q[blocks - 1], s[0], blocks / 2, s[blocks / 2], blocks - 1, s[blocks - 1]);
    if (rep == 3) {  // per XCD: first and last start, and which block indices it got
      for (unsigned x = 0; x < 8; ++x) {
        double lo = 1e9, hi = 0; int n = 0, bmin = 1 << 30, bmax = -1;
        for (int b = 0; b < blocks; ++b) if (hx[b] == x) { lo = std::min(lo, s[b]); hi = std::max(hi, s[b]); ++n; bmin = std::min(bmin, b); bmax = std::max(bmax, b); }
        printf("  XCD %u: %d workgroups (block %d .. %d), starts %.2f .. %.2f us\n", x, n, bmin, bmax, lo, hi);
      }
    }
  }
  return 0;
}
