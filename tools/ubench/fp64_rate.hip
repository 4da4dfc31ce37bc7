// Aggregate issue rate of fp64 FMA per SIMD: plain v_fmac_f64 vs v_fmac_f64_dpp row_newbcast with
// 1 / 2 / 4 waves per SIMD.  Every wave stamps its own start and end; the figure printed is
//   (latest end - earliest start on the CU) / (fmacs issued per SIMD)
// i.e. cycles per wave-instruction as the SIMD sees it (4.0 = the 16 lanes/cycle vector fp64 peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
template <int DPP>
__global__ void k(double* out, long long* clk, int iters) {
  double a[16]; for (int j = 0; j < 16; ++j) a[j] = threadIdx.x + j;
  double c = out[threadIdx.x & 63], u = out[64 + (threadIdx.x & 63)];
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (DPP)
    asm volatile(
      "v_fmac_f64_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
        "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(c), "v"(u));
    else
    asm volatile(
      "v_fmac_f64 %0, %16, %17\n v_fmac_f64 %1, %16, %17\n v_fmac_f64 %2, %16, %17\n v_fmac_f64 %3, %16, %17\n"
      "v_fmac_f64 %4, %16, %17\n v_fmac_f64 %5, %16, %17\n v_fmac_f64 %6, %16, %17\n v_fmac_f64 %7, %16, %17\n"
      "v_fmac_f64 %8, %16, %17\n v_fmac_f64 %9, %16, %17\n v_fmac_f64 %10, %16, %17\n v_fmac_f64 %11, %16, %17\n"
      "v_fmac_f64 %12, %16, %17\n v_fmac_f64 %13, %16, %17\n v_fmac_f64 %14, %16, %17\n v_fmac_f64 %15, %16, %17\n"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
        "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(c), "v"(u));
  }
  long long t1 = clock64();
  double s = 0; for (int j = 0; j < 16; ++j) s += a[j];
  out[128 + blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    clk[2 * w] = t0; clk[2 * w + 1] = t1;
  }
}
int main() {
  double* out; long long* clk; hipMalloc(&out, 8 * (128 + 1024 * 1024)); hipMalloc(&clk, 16 * 8192);
  hipMemset(out, 0, 8 * 128);
  const int iters = 2000;
  static long long h[2 * 8192];
  for (int threads : {256, 512, 1024}) {
    for (int which = 0; which < 2; ++which) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (which == 0) k<0><<<256, threads>>>(out, clk, iters); else k<1><<<256, threads>>>(out, clk, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      const int wpb = threads / 64;
      hipMemcpy(h, clk, 16 * 256 * wpb, hipMemcpyDeviceToHost);
      double avg = 0, slow = 0;
      for (int b = 0; b < 256; ++b) {
        long long lo = h[2 * b * wpb], hi = h[2 * b * wpb + 1];
        for (int w = 0; w < wpb; ++w) { lo = std::min(lo, h[2 * (b * wpb + w)]); hi = std::max(hi, h[2 * (b * wpb + w) + 1]); }
        avg += double(hi - lo);
        slow = std::max(slow, double(hi - lo));
      }
      avg /= 256;
      const double per_simd = 16.0 * iters * (wpb / 4.0);
      printf("%s waves/SIMD=%d: %.2f cycles per wave-fmac per SIMD (CU span avg; worst %.2f); kernel %.3f ms -> %.1f TFLOP/s fp64\n",
             which ? "dpp  " : "plain", wpb / 4, avg / per_simd, slow / per_simd, ms,
             2.0 * 64 * 16.0 * iters * wpb * 256 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
