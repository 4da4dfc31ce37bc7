#include <hip/hip_runtime.h>
template <int N>
__device__ __forceinline__ void fmac_bcast(double& a, double c, double u) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c), "v"(u), "n"(N));
}
__global__ void k(const double* c, const double* u, double* out) {
  const int l = threadIdx.x;
  double cv = c[l], uv = u[l];
  double a0 = 0, a1 = 0, a5 = 0, a15 = 0;
  fmac_bcast<0>(a0, cv, uv);
  fmac_bcast<1>(a1, cv, uv);
  fmac_bcast<5>(a5, cv, uv);
  fmac_bcast<15>(a15, cv, uv);
  out[l * 4 + 0] = a0; out[l * 4 + 1] = a1; out[l * 4 + 2] = a5; out[l * 4 + 3] = a15;
}
int main() {
  double *c, *u, *o; hipMalloc(&c, 512); hipMalloc(&u, 512); hipMalloc(&o, 2048);
  double hc[64], hu[64], ho[256];
  for (int i = 0; i < 64; ++i) { hc[i] = 100 + i; hu[i] = 1.0 + 0.01 * i; }
  hipMemcpy(c, hc, 512, hipMemcpyHostToDevice); hipMemcpy(u, hu, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(c, u, o); hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) { int rb = l & ~15; int idx[4] = {0, 1, 5, 15};
    for (int q = 0; q < 4; ++q) { double e = hc[rb + idx[q]] * hu[l]; if (ho[l * 4 + q] != e) { bad++; if (bad < 5) printf("lane %d q %d got %f exp %f\n", l, q, ho[l*4+q], e); } } }
  printf("bad=%d\n", bad); return bad != 0;
}
