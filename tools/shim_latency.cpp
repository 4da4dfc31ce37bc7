// shim_latency.cpp -- what a drop-in user of the reference's six-symbol interface gets per MPC
// cycle: setup_problem + update_x_drag + update_solver_settings + update_problem_data_floats +
// 12 x get_solution, exactly the call sequence of ConvexMPCLocomotion.cpp:630-685, on ONE robot.
//   build: g++ -O2 -std=c++17 tools/shim_latency.cpp -Iinclude -Lquadruped_ctrl_amd -lconvexmpc_shim -lqmpc
//          -Wl,-rpath,$PWD/quadruped_ctrl_amd -o tools/shim_latency
//   run:   tools/shim_latency record.bin [cycles]     (record.bin written by tools/shim_latency.py)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "convexMPC_interface.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int h = 0, nrec = 0;
  float dt, mu, fmax;
  if (std::fread(&h, 4, 1, f) != 1 || std::fread(&nrec, 4, 1, f) != 1) return 2;
  if (std::fread(&dt, 4, 1, f) != 1 || std::fread(&mu, 4, 1, f) != 1 || std::fread(&fmax, 4, 1, f) != 1) return 2;
  const int nf = 3 + 3 + 4 + 3 + 12 + 1 + 12 + 12 * h + 1;
  std::vector<float> rec((size_t)nrec * nf);
  std::vector<int> gait((size_t)nrec * 4 * h);
  for (int i = 0; i < nrec; ++i) {
    if (std::fread(&rec[(size_t)i * nf], 4, nf, f) != (size_t)nf) return 2;
    if (std::fread(&gait[(size_t)i * 4 * h], 4, 4 * h, f) != (size_t)(4 * h)) return 2;
  }
  std::fclose(f);
  const int cycles = argc > 2 ? std::atoi(argv[2]) : 2000;
  std::vector<double> us;
  double sink = 0;
  int bad = 0;
  for (int it = 0; it < cycles + 50; ++it) {
    float* r = &rec[(size_t)(it % nrec) * nf];
    int* g = &gait[(size_t)(it % nrec) * 4 * h];
    const auto t0 = std::chrono::steady_clock::now();
    setup_problem(dt, h, mu, fmax);
    update_x_drag(0.f);
    update_solver_settings(10000, 1e-7, 1e-8, 1.5, 0.1, 0.0);
    update_problem_data_floats(r, r + 3, r + 6, r + 10, r + 13, r[25], r + 26, r + 38, r[38 + 12 * h], g);
    for (int k = 0; k < 12; ++k) sink += get_solution(k);
    const auto t1 = std::chrono::steady_clock::now();
    if (it >= 50) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    if (qmpc_shim_last_status() != 0) ++bad;
  }
  std::sort(us.begin(), us.end());
  double mean = 0;
  for (double v : us) mean += v;
  mean /= us.size();
  std::printf("{\"horizon\": %d, \"cycles\": %d, \"mean_us\": %.2f, \"median_us\": %.2f, \"p10_us\": %.2f, \"p99_us\": %.2f, "
              "\"iters_last\": %d, \"status_nonzero\": %d, \"sink\": %.3f}\n",
              h, cycles, mean, us[us.size() / 2], us[us.size() / 10], us[us.size() * 99 / 100], qmpc_shim_last_iters(), bad, sink);
  return 0;
}
