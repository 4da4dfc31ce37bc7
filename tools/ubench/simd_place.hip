// Where do the six waves of a 384-thread workgroup land?  Two workgroups per CU (80 KB of LDS each), every wave reports
// its HW_ID (SIMD, CU, SE) and XCC_ID while all of them are resident.   hipcc --offload-arch=gfx950 -O2 simd_place.hip -o simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6;
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64();
  while (clock64() - t0 < spin) { lds[threadIdx.x] = (char)spin; }
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 8 + wave) * 2 + 0] = hwid;
    out[(blockIdx.x * 8 + wave) * 2 + 1] = xcc;
  }
}
int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 384, blocks = argc > 2 ? atoi(argv[2]) : 512, ldsb = argc > 3 ? atoi(argv[3]) : 80 * 1024;
  unsigned* d;
  hipMalloc(&d, blocks * 8 * 2 * 4);
  hipMemset(d, 0xff, blocks * 8 * 2 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), ldsb, 0, d, 2000000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 16);
  hipMemcpy(h.data(), d, blocks * 16 * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;  // (xcc, se, sh, cu) -> waves per SIMD
  std::map<unsigned, int> nb;
  std::map<std::vector<int>, int> pat;
  for (int b = 0; b < blocks; ++b) {
    std::vector<int> mine(4, 0);
    unsigned key = 0;
    for (int w = 0; w < threads / 64; ++w) {
      const unsigned id = h[(b * 8 + w) * 2], xcc = h[(b * 8 + w) * 2 + 1] & 0xf;
      const unsigned simd = (id >> 4) & 3, cuid = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
      key = (xcc << 16) | (se << 8) | (sh << 4) | cuid;
      if (cu[key].empty()) cu[key].assign(4, 0);
      cu[key][simd]++;
      mine[simd]++;
    }
    nb[key]++;
    pat[mine]++;
  }
  {  // which SIMD does each wave INDEX land on?  (the engine wave of the solve kernels is wave 0)
    int hist[8][4] = {};
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < threads / 64; ++w) hist[w][(h[(b * 8 + w) * 2] >> 4) & 3]++;
    printf("wave index -> SIMD histogram:\n");
    for (int w = 0; w < threads / 64; ++w) printf("  wave %d: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  }
  printf("per-workgroup SIMD patterns (waves on SIMD0..3 -> count):\n");
  for (auto& p : pat) printf("  %d %d %d %d : %d\n", p.first[0], p.first[1], p.first[2], p.first[3], p.second);
  std::map<std::vector<int>, int> cupat;
  for (auto& c : cu) { std::vector<int> v = c.second; v.push_back(nb[c.first]); cupat[v]++; }
  printf("per-CU totals (waves on SIMD0..3, workgroups -> CUs):\n");
  for (auto& p : cupat) printf("  %d %d %d %d (%d wgs) : %d\n", p.first[0], p.first[1], p.first[2], p.first[3], p.first[4], p.second);
  return 0;
}
