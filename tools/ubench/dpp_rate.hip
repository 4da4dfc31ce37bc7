// throughput of v_fmac_f64 (plain) vs v_fmac_f64_dpp row_newbcast, one wave / four waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_plain(double* out, long long* clk, int iters) {
  double a[16]; for (int j = 0; j < 16; ++j) a[j] = threadIdx.x + j;
  double c = out[threadIdx.x & 63], u = out[64 + (threadIdx.x & 63)];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
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
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
__global__ void k_dpp(double* out, long long* clk, int iters) {
  double a[16]; for (int j = 0; j < 16; ++j) a[j] = threadIdx.x + j;
  double c = out[threadIdx.x & 63], u = out[64 + (threadIdx.x & 63)];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
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
  }
  long long t1 = clock64();
  double s = 0; for (int j = 0; j < 16; ++j) s += a[j];
  out[128 + blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* clk; hipMalloc(&out, 8 * (128 + 1024 * 1024)); hipMalloc(&clk, 8 * 4096);
  hipMemset(out, 0, 8 * 128);
  const int iters = 1000;
  for (int threads : {64, 256, 1024}) {
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        if (which == 0) k_plain<<<256, threads>>>(out, clk, iters); else k_dpp<<<256, threads>>>(out, clk, iters);
        hipDeviceSynchronize();
      }
      long long h[256]; hipMemcpy(h, clk, 8 * 256, hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
      printf("%s threads/WG=%d (waves/SIMD=%d): %.2f cycles per fmac per wave\n", which ? "dpp  " : "plain", threads, threads / 256 ? threads / 256 : 1, avg / (16.0 * iters));
    }
  }
  return 0;
}
