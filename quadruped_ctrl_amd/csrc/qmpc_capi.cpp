// qmpc_capi.cpp -- host side of the C ABI declared in include/qmpc.h.
//
// Owns the per-handle device state that the reference keeps in file-scope
// globals (src/MPC_Ctrl/convexMPC_interface.cpp:13-20, SolverMPC.cpp:18-57):
// problem constants, the batch-constant coefficient tables, the size-class
// work lists, and staging buffers for the host-pointer entry point.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qmpc.h"
#include "../../include/qmpc_debug.h"   // test / profiling hooks (same library)
#include "../../include/qmpc_expert.h"  // tuning knobs whose default is the measured optimum, warm start
#include "qmpc_device.h"

// per-class entry points of qmpc_kernels.hip (one translation unit per size class)
#define QMPC_DECLARE_CLASS(RB)                                                                  \
  extern "C" size_t qmpc_c##RB##_smem(void);                                                    \
  extern "C" hipError_t qmpc_c##RB##_prepare(void);                                             \
  extern "C" int qmpc_c##RB##_resident(void);                                                   \
  extern "C" hipError_t qmpc_c##RB##_launch(const QmpcParams* P, int grid, hipStream_t stream);   \
  extern "C" int qmpc_c##RB##_resident_sweep(void);                                              \
  extern "C" hipError_t qmpc_c##RB##_launch_sweep(const QmpcParams* P, int grid, hipStream_t stream);
QMPC_DECLARE_CLASS(1)
QMPC_DECLARE_CLASS(2)
QMPC_DECLARE_CLASS(3)
QMPC_DECLARE_CLASS(4)
QMPC_DECLARE_CLASS(6)

// the decoupled path's consumer (qmpc_engine.hip)
extern "C" hipError_t qmpc_engine_prepare(void);
extern "C" hipError_t qmpc_big_prepare(void);
extern "C" hipError_t qmpc_big_launch(const QmpcParams* P, int grid, hipStream_t stream);
extern "C" int qmpc_engine_resident(int rb);
extern "C" int qmpc_engine_capacity(int rb);
extern "C" hipError_t qmpc_engine_launch(int rb, const QmpcParams* P, int grid, hipStream_t stream);
extern "C" hipError_t qmpc_admm_big_launch(const QmpcParams* P, int grid, hipStream_t stream);

extern "C" size_t qmpc_smem_bytes(int rb) {
  switch (rb) {
    case 1: return qmpc_c1_smem();
    case 2: return qmpc_c2_smem();
    case 3: return qmpc_c3_smem();
    case 4: return qmpc_c4_smem();
    case 6: return qmpc_c6_smem();
  }
  return 0;
}
extern "C" int qmpc_resident_blocks(int rb) {
  switch (rb) {
    case 1: return qmpc_c1_resident();
    case 2: return qmpc_c2_resident();
    case 3: return qmpc_c3_resident();
    case 4: return qmpc_c4_resident();
    case 6: return qmpc_c6_resident();
  }
  return 0;
}
static hipError_t qmpc_prepare(void) {
  hipError_t e;
  if ((e = qmpc_c1_prepare()) != hipSuccess) return e;
  if ((e = qmpc_c4_prepare()) != hipSuccess) return e;
  if ((e = qmpc_c6_prepare()) != hipSuccess) return e;
  if ((e = qmpc_c2_prepare()) != hipSuccess) return e;
  if ((e = qmpc_c3_prepare()) != hipSuccess) return e;
  if ((e = qmpc_big_prepare()) != hipSuccess) return e;
  return qmpc_engine_prepare();
}
static hipError_t qmpc_launch_sweep(int rb, const QmpcParams* P, int grid, hipStream_t stream) {
  return rb == 2 ? qmpc_c2_launch_sweep(P, grid, stream) : (rb == 3 ? qmpc_c3_launch_sweep(P, grid, stream) : hipErrorInvalidValue);
}
static int qmpc_resident_sweep(int rb) { return rb == 2 ? qmpc_c2_resident_sweep() : (rb == 3 ? qmpc_c3_resident_sweep() : 0); }
static hipError_t qmpc_launch(int rb, const QmpcParams* P, int grid, hipStream_t stream) {
  switch (rb) {
    case 1: return qmpc_c1_launch(P, grid, stream);
    case 2: return qmpc_c2_launch(P, grid, stream);
    case 3: return qmpc_c3_launch(P, grid, stream);
    case 4: return qmpc_c4_launch(P, grid, stream);
    case 6: return qmpc_c6_launch(P, grid, stream);
  }
  return hipErrorInvalidValue;
}
extern "C" hipError_t qmpc_launch_keys(const QmpcParams* P, int* nst, float* score, float* demand, hipStream_t stream);
extern "C" hipError_t qmpc_launch_pack(const qmpc_command* c, const qmpc_record* rec, int batch, int horizon, float dt_mpc,
                                       hipStream_t stream);
extern "C" hipError_t qmpc_launch_f2b(const float* r_body, const float* grf, float* f_ff, int batch, hipStream_t stream);
extern "C" hipError_t qmpc_launch_leg_kin(const float geom[4], const float* q, const float* qd, float* J, float* p,
                                          float* v, int batch, hipStream_t stream);
extern "C" hipError_t qmpc_launch_leg_cmd(const float geom[4], const qmpc_leg_command* c, float* tau, float* q_des,
                                          int batch, hipStream_t stream);
extern "C" hipError_t qmpc_launch_kf(const qmpc_kf_state* st, const float hip[3], int batch, hipStream_t stream);
extern "C" hipError_t qmpc_launch_kf_init(float* xhat, float* P, int batch, hipStream_t stream);
extern "C" hipError_t qmpc_launch_swing(const float* p0, const float* pf, const float* height, const float* phase,
                                        const float* swing_time, float* p, float* v, float* a, int n_feet,
                                        hipStream_t stream);

// Small device arrays are (re)set by this kernel, never by hipMemsetAsync: a memset node captured into a hipGraph writes
// garbage on replay with this runtime (ROCm 7.2: the counters came back as {512, 24111, 1, 24106, ...}; measured with
// tools/dbg/graph_probe.py), a kernel node replays correctly
__global__ void qmpc_fill_ints_kernel(int* p, int n, int v) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) p[k] = v;
}
static hipError_t fill_ints(int* p, int n, int v, hipStream_t stream) {
  hipLaunchKernelGGL(qmpc_fill_ints_kernel, dim3((n + 255) / 256 < 64 ? (n + 255) / 256 : 64), dim3(256), 0, stream, p, n, v);
  return hipGetLastError();
}

// Order hint: robots sorted by the iteration count of the handle's previous call, longest first (counting sort, one
// workgroup; the order inside a bin is whatever the atomics make it -- a robot's result does not depend on its place)
#define QMPC_HINT_BINS 64
__global__ __launch_bounds__(1024) void qmpc_order_kernel(const int* __restrict__ it, int* __restrict__ order, int n) {
  __shared__ int hist[QMPC_HINT_BINS];
  if (threadIdx.x < QMPC_HINT_BINS) hist[threadIdx.x] = 0;
  __syncthreads();
  auto bin = [](int v) { return QMPC_HINT_BINS - 1 - (v < 0 ? 0 : (v > QMPC_HINT_BINS - 1 ? QMPC_HINT_BINS - 1 : v)); };
  for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[bin(it[i])], 1);
  __syncthreads();
  if (threadIdx.x < QMPC_HINT_BINS) {  // exclusive prefix over the 64 bins: one wave, six shuffle steps
    const int cnt = hist[threadIdx.x];
    int acc = cnt;
#pragma unroll
    for (int d = 1; d < QMPC_HINT_BINS; d <<= 1) {
      const int up = __shfl_up(acc, d);
      if ((int)threadIdx.x >= d) acc += up;
    }
    hist[threadIdx.x] = acc - cnt;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) order[atomicAdd(&hist[bin(it[i])], 1)] = i;
}

struct qmpc_ctx {
  int device = 0;
  int max_batch = 0, max_horizon = 0;
  bool is_setup = false;
  int horizon = 0;
  double dt = 0, mu = 0, f_max = 0;
  double mass = 9.0;                       // RobotState.h:23
  double ibody[3] = {.07f, 0.26f, 0.242f}; // RobotState.cpp:38 (float literals)
  double gravity = -9.8f;                  // SolverMPC.cpp:318
  int max_iter = 1000;
  double tol = 1e-9;
  float leg_geom[4] = {0.062f, 0.209f, 0.195f, 0.004f};  // MiniCheetah.h:31-37 (abad, hip, knee, knee Y offset)
  double* d_tables = nullptr;  // coef[3][H] then ctab[9][H + 1][H + 1] (zero last row and column: the kernels' identity padding reads them)
  int* d_lists = nullptr;      // [4][max_batch] robot ids handed to classes 4, 2 and 3, and to the large-problem producer
  int* d_counts = nullptr;     // [3 sets][QMPC_COUNTERS] (layout in qmpc_device.h); sets 0 / 1 ping-ponged between calls, set 2: calls captured into a graph
  // decoupled path (sweep kernel -> work items -> engine kernel) of the 128- and 192-row classes: [0] class 2, [1] class 3
  int split = 1;               // qmpc_set_split: 0 off, 1 automatic (by batch size), 2 always; QMPC_NO_SPLIT=1 in the environment: 0 at creation
  double* d_wk_hinv[3] = {nullptr, nullptr, nullptr};
  double* d_wk_xu[3] = {nullptr, nullptr, nullptr};
  QmpcWorkHdr* d_wk_hdr[3] = {nullptr, nullptr, nullptr};  // [2]: the large problems (192 < n_r <= 432)
  int* d_wk_order[3] = {nullptr, nullptr, nullptr};
  double* d_wk_ovf[3] = {nullptr, nullptr, nullptr};  // engine kernels' overflow event pools (one slice per resident workgroup)  // [QMPC_ORDER_BUCKETS][max_batch] item indices, hardest robots first
  int wk_cap[3] = {0, 0, 0};    // work items per class: a bounded pool, min(max_batch, QMPC_ITEMS_*) -- ensure_pools
  int* d_fb_lists = nullptr;   // [3][max_batch] robots the engine kernels hand back (128-row, 192-row, large problems)
  bool block = false;          // qmpc_set_block_start (experimental, off: measured no faster); QMPC_BLOCK=1 in the environment switches it on at creation
  int dense = 1;               // qmpc_set_dense: the 64-row class at five workgroups per CU -- 0 never, 1 automatic (by the handle's size), 2 whenever the chain is that class alone
  int chunks = 0;              // qmpc_set_chunks (test hook): run the item classes in at least this many chunks (0 / 1: as few as the pools allow)
  int dbg_engine_events = 0;   // test hook: events the engine may hold per robot (0 = the compiled capacity)
  unsigned call_no = 0;
  // order hint (qmpc_set_order_hint): 0 off, 1 automatic -- a call whose first class is launched over more robots than it has
  // resident workgroups takes them in the order of the previous call's iteration counts (same batch size), longest first
  int order_hint = 1;
  int* d_hint_iters = nullptr;  // [max_batch] iteration counts the one-kernel classes left in the previous call
  int* d_order = nullptr;       // [max_batch] the permutation of this call
  // size order (qmpc_set_size_order, default on): no hint usable (first call, another batch size, hint off) -> the first class of a
  // chain, launched over several rounds, takes the robots that fit it largest first by their contact tables; the permutation is
  // built inside the launch (qmpc_kernels.hip: size_order_build)
  int size_order = 1;
  unsigned long long* d_so_order = nullptr;  // [max_batch] (call number << 32 | robot)
  unsigned so_call = 0;
  unsigned long long* d_prio_cu = nullptr;  // [2048] one-round launches without a hint: one word per CU (qmpc_device.h: prio_cu)
  unsigned prio_call = 0;
  int so_first_pct = 0, so_first_pct_hint = 50;  // the unsorted head beyond the first round, % of a round: keys from the records / from the hint
  int so_min_div = 8;     // at least a round / so_min_div robots to order (measured 2 / 4 / 8 on batches of 1.1 ... 2.5 rounds: no loss anywhere, +4 ... +13 % at 1.4 rounds)
  int so_tail_rounds = 5;
  int hint_prepass = 0;  // 1: the order hint's permutation by a sort kernel in front of the call (as until round 6)
  int hint_batch = 0;           // batch size of the call that wrote d_hint_iters (0: none yet)
  int hint_hard = 5;            // single-round launches: iterations in the previous call from which a robot may keep the highest issue priority (0 = off)
  int* d_hint_max = nullptr;    // [3] largest iteration count of the last calls (slots rotated by hint_call: read / fold / clear)
  unsigned hint_call = 0;
  int* d_cu_slots = nullptr;
  int bal_debug = 0;  // qmpc_set_debug_balance
  int max_stance = 0;          // caller's bound on stance foot-steps per robot (0 = unknown)
  int min_stance = 0;          // ... and lower bound (0 = unknown)
  int admm_mode = 0, admm_max_iter = 10000;  // JCQP alternate, see qmpc_settings_jcqp
  double admm_rho = 1e-7, admm_sigma = 1e-8, admm_alpha = 1.5, admm_term = 0.1;
  double* d_evpool = nullptr;  // largest size class: global event pool (allocated on first use)
  double* d_ovpool = nullptr;  // the other classes: overflow event pool, ov_nslice slices, recycled within a call (one flag per slice)
  int* d_ovflags = nullptr;    // [ov_nslice_alloc] 0 = free
  int ov_nslice = 0;
  int ov_spin = 1 << 22;       // probes of a robot that finds every slice taken before it gives up (test hook: qmpc_set_debug_overflow_slices)
  bool device_error = false;   // a HIP call of this handle failed since the last successful solve (fail()): see solve_impl
  int* d_evflags = nullptr;
  int ev_nslot = 0;
  bool dbg_pool_busy = false;  // test hook: every slice of the 192-row class's pool looks taken (qmpc_set_debug_pool_busy)
  int32_t* ws = nullptr;  // warm-start buffer (device), see qmpc_set_warm_start
  int ws_shift = 1;
  int ws_min_iters = 0;   // qmpc_set_warm_start_min_iters: only robots with at least this many iterations in the previous call start warm
  double* dbg_H = nullptr;
  double* dbg_g = nullptr;
  double* dbg_aux = nullptr;
  long long* dbg_clk = nullptr;
  // staging for qmpc_solve_host
  void* d_stage = nullptr;
  size_t stage_bytes = 0;
  // the handle's device state (tables, work lists, counters) is ordered by ONE stream at a
  // time: a call that arrives on a different stream is made to wait for the previous one
  hipStream_t last_stream = nullptr;
  bool has_last = false;
  hipEvent_t order_ev = nullptr;
  // host-pointer entry point: its own stream, one pinned (device-visible) staging block
  hipStream_t host_stream = nullptr;
  hipEvent_t host_ev = nullptr;
  void* h_pin = nullptr;
  size_t pin_bytes = 0;
  double tab_dt = -1.0;  // (dt, horizon, model) the tables in d_tables were built for
  int tab_h = -1, tab_model = -1;
  int model = 0;         // QMPC_MODEL_*
  std::string err;
};

namespace {

struct DeviceGuard {
  int prev = 0;
  explicit DeviceGuard(int dev) {
    hipGetDevice(&prev);
    if (prev != dev) hipSetDevice(dev);
  }
  ~DeviceGuard() { hipSetDevice(prev); }
};

int fail(qmpc_ctx* c, hipError_t e, const char* what) {
  c->err = std::string(what) + ": " + hipGetErrorString(e);
  c->device_error = true;  // (the next solve re-arms the device state a failed launch chain may have left behind)
  return QMPC_ERR_DEVICE;
}

#define HIP_TRY(ctx, call)                              \
  do {                                                  \
    hipError_t e__ = (call);                            \
    if (e__ != hipSuccess) return fail(ctx, e__, #call); \
  } while (0)

// One stream at a time orders the handle's device state.  A call on a different stream than the
// previous one waits (on the device, no host block) for everything the previous stream was given.
// (a stream that is being captured into a graph: the work becomes graph nodes; the legacy null stream cannot capture)
bool is_capturing(hipStream_t stream) {
  if (!stream) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}

int order_after_previous(qmpc_ctx* c, hipStream_t stream) {
  // (captured calls: ordering against earlier work of the handle on OTHER streams is the caller's business -- an event
  //  recorded outside the capture cannot be waited for inside it)
  if (is_capturing(stream)) return QMPC_OK;
  if (c->has_last && c->last_stream != stream) {
    if (!c->order_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->order_ev, c->last_stream));
    HIP_TRY(c, hipStreamWaitEvent(stream, c->order_ev, 0));
  }
  c->last_stream = stream;
  c->has_last = true;
  return QMPC_OK;
}

}  // namespace

namespace {
int ensure_pools(qmpc_ctx* c);
}

extern "C" {

int qmpc_abi_version(void) { return 21; }  // 21: qmpc_set_size_order (expert), qmpc_debug_keys (debug) added; no signature changed
int qmpc_max_horizon(void) { return QMPC_MAX_HORIZON; }

const char* qmpc_last_error(qmpc_handle h) { return h ? h->err.c_str() : "null handle"; }

int qmpc_create(int device_id, int max_batch, int max_horizon, qmpc_handle* out) {
  if (!out || max_batch <= 0 || max_horizon <= 0 || max_horizon > QMPC_MAX_HORIZON)
    return QMPC_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
    return QMPC_ERR_DEVICE;
  qmpc_ctx* c = new qmpc_ctx();
  c->device = device_id;
  c->max_batch = max_batch;
  c->max_horizon = max_horizon;
  DeviceGuard g(device_id);
  const size_t H = (size_t)max_horizon;
  hipError_t e = hipMalloc(&c->d_tables, sizeof(double) * (3 * H + 9 * (H + 1) * (H + 1)));
  if (e == hipSuccess) e = hipMalloc(&c->d_lists, sizeof(int) * 4 * (size_t)max_batch);  // ([3]: the large problems, horizons > 16)
  if (e == hipSuccess) e = hipMalloc(&c->d_counts, sizeof(int) * 3 * QMPC_COUNTERS);
  if (e == hipSuccess) e = hipMemset(c->d_counts, 0, sizeof(int) * 3 * QMPC_COUNTERS);
  if (e == hipSuccess) e = hipMalloc(&c->d_fb_lists, sizeof(int) * 3 * (size_t)max_batch);
  if (e == hipSuccess) e = hipMalloc(&c->d_hint_iters, sizeof(int) * (2 * (size_t)max_batch + 4 + 2048));
  if (e == hipSuccess) e = hipMemset(c->d_hint_iters, 0, sizeof(int) * (2 * (size_t)max_batch + 4 + 2048));
  if (e == hipSuccess) c->d_order = c->d_hint_iters + max_batch;
  if (e == hipSuccess) e = hipMalloc(&c->d_prio_cu, sizeof(unsigned long long) * 2048);
  if (e == hipSuccess) e = hipMemset(c->d_prio_cu, 0, sizeof(unsigned long long) * 2048);
  if (e == hipSuccess) e = hipMalloc(&c->d_so_order, sizeof(unsigned long long) * 2 * (size_t)max_batch);  // near copy, far copy
  if (e == hipSuccess) e = hipMemset(c->d_so_order, 0, sizeof(unsigned long long) * 2 * (size_t)max_batch);  // (call numbers start at 1)
  if (e == hipSuccess) c->d_hint_max = c->d_hint_iters + 2 * (size_t)max_batch;
  if (e == hipSuccess) c->d_cu_slots = c->d_hint_max + 4;  // [2048] the 96-row class's per-CU placement words (zero whenever no kernel runs)
  {
    const char* ns = std::getenv("QMPC_NO_SPLIT");
    c->split = (ns && ns[0] == '1') ? 0 : 1;
    const char* sm = std::getenv("QMPC_SO_MIN_DIV");
    if (sm && std::atoi(sm) > 0) c->so_min_div = std::atoi(sm);
    const char* st = std::getenv("QMPC_SO_TAIL_ROUNDS");
    if (st && std::atoi(st) > 0) c->so_tail_rounds = std::atoi(st);
    const char* sf = std::getenv("QMPC_SO_FIRST_PCT");  // (measurement knob: the unsorted head of a size-ordered launch, % of a round beyond the first)
    if (sf) c->so_first_pct = c->so_first_pct_hint = std::atoi(sf);
    const char* hp = std::getenv("QMPC_HINT_PREPASS");
    if (hp) c->hint_prepass = std::atoi(hp);
    const char* hh = std::getenv("QMPC_HINT_HARD");
    if (hh) c->hint_hard = std::atoi(hh);
    const char* nb = std::getenv("QMPC_BLOCK");
    c->block = nb && nb[0] == '1';
  }
  if (e == hipSuccess) {
    // a robot whose on-chip event pool fills up continues here.  The slices are RECYCLED within a call (one flag per slice,
    // taken with a compare-and-swap, released when its robot is done): ov_nslice bounds the robots that hold one AT THE SAME
    // TIME, not the robots of a call; a robot that finds every slice taken waits (ov_spin probes), then falls back to the
    // Schur-form engine.  Every flag is 0 between calls; after a call that ended in a device error they are re-armed
    // (solve_impl) -- a flag left at 1 would otherwise be lost for the life of the handle
    // (2048 slices of 192 KiB: what a call of 8192 robots needs when the 64-row class runs five per CU ahead of larger
    //  classes -- see the launch loop of solve_impl)
    c->ov_nslice = max_batch < 2048 ? max_batch : 2048;
    e = hipMalloc(&c->d_ovpool, sizeof(double) * (size_t)c->ov_nslice * QMPC_OV_SLICE);
    if (e == hipSuccess) e = hipMalloc(&c->d_ovflags, sizeof(int) * (size_t)c->ov_nslice);
    if (e == hipSuccess) e = hipMemset(c->d_ovflags, 0, sizeof(int) * (size_t)c->ov_nslice);
  }
  if (e == hipSuccess) e = qmpc_prepare();
  if (e == hipSuccess) {
    // the runtime loads a translation unit's code object at the first launch of one of its kernels (2 MiB of device
    // memory): this file's small kernels (fill, order hint) are first launched HERE, not inside some later solve call
    e = fill_ints(c->d_hint_max, 4, 0, nullptr);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(qmpc_order_kernel, dim3(1), dim3(1024), 0, nullptr, (const int*)c->d_hint_iters, c->d_order, 1);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  }
  if (e != hipSuccess) {
    qmpc_destroy(c);
    return QMPC_ERR_DEVICE;
  }
  *out = c;
  return QMPC_OK;
}

int qmpc_destroy(qmpc_handle h) {
  if (!h) return QMPC_ERR_ARG;
  {
    DeviceGuard g(h->device);
    if (h->d_tables) hipFree(h->d_tables);
    if (h->d_lists) hipFree(h->d_lists);
    if (h->d_counts) hipFree(h->d_counts);
    if (h->d_stage) hipFree(h->d_stage);
    if (h->d_evpool) hipFree(h->d_evpool);
    if (h->d_ovpool) hipFree(h->d_ovpool);
    if (h->d_ovflags) hipFree(h->d_ovflags);
    if (h->d_evflags) hipFree(h->d_evflags);
    if (h->d_fb_lists) hipFree(h->d_fb_lists);
    if (h->d_hint_iters) hipFree(h->d_hint_iters);
    if (h->d_so_order) hipFree(h->d_so_order);
    if (h->d_prio_cu) hipFree(h->d_prio_cu);
    for (int k = 0; k < 3; ++k) {
      if (h->d_wk_hinv[k]) hipFree(h->d_wk_hinv[k]);
      if (h->d_wk_xu[k]) hipFree(h->d_wk_xu[k]);
      if (h->d_wk_hdr[k]) hipFree(h->d_wk_hdr[k]);
      if (h->d_wk_order[k]) hipFree(h->d_wk_order[k]);
      if (h->d_wk_ovf[k]) hipFree(h->d_wk_ovf[k]);
    }
    if (h->h_pin) hipHostFree(h->h_pin);
    if (h->order_ev) hipEventDestroy(h->order_ev);
    if (h->host_ev) hipEventDestroy(h->host_ev);
    if (h->host_stream) hipStreamDestroy(h->host_stream);
  }
  delete h;
  return QMPC_OK;
}

int qmpc_setup(qmpc_handle c, double dt, int horizon, double mu, double f_max) {
  if (!c) return QMPC_ERR_ARG;
  if (horizon <= 0 || horizon > c->max_horizon || !(mu > 0) || !(dt > 0)) return QMPC_ERR_ARG;
  // struct problem_setup stores floats (convexMPC_interface.h:13-19)
  c->dt = (double)(float)dt;
  c->mu = (double)(float)mu;
  c->f_max = (double)(float)f_max;
  c->horizon = horizon;
  const int h = horizon;
  if (12 * h > 128 && !c->d_evpool) {
    // the 192-row class keeps its rank-1 events in global memory: one 400 KiB slice (QMPC_EV_SLICE3) per workgroup in
    // flight (256 CUs x 1 workgroup); 1024 flag-guarded slices, so that a slice's previous tenant has usually finished
    // (a workgroup whose predecessor on its slice is still running waits for it).  Allocated here, where the
    // horizon that makes the class reachable is announced -- never inside a solve call (a hipMalloc there
    // synchronises the device and cannot be captured into a graph)
    DeviceGuard g0(c->device);
    const int nslot = c->max_batch < 1024 ? c->max_batch : 1024;
    HIP_TRY(c, hipMalloc(&c->d_evpool, sizeof(double) * (size_t)nslot * QMPC_EV_SLICE3));
    HIP_TRY(c, hipMalloc(&c->d_evflags, sizeof(int) * (size_t)nslot));
    HIP_TRY(c, hipMemset(c->d_evflags, 0, sizeof(int) * (size_t)nslot));
    c->ev_nslot = nslot;
  }
  // the tables depend on (dt, horizon) only: the reference's caller repeats setup_problem with the
  // same values before every solve (ConvexMPCLocomotion.cpp:630), which costs nothing here
  if (c->is_setup && c->tab_dt == c->dt && c->tab_h == h && c->tab_model == c->model) return ensure_pools(c);
  // coefficient tables (see qmpc_device.h); A_ct^3 = 0 makes
  // Adt^d Bdt = dt B + c_d A B + e_d A^2 B exact.
  const int hs = h + 1;  // row stride of the C_pq tables: row h and column h stay zero
  std::vector<double> t(3 * h + 9 * hs * hs, 0.0);
  double* coef = t.data();
  double* ctab = t.data() + 3 * h;
  const double d1 = c->dt;
  for (int d = 0; d < h; ++d) {
    coef[0 * h + d] = d1;
    const double dp = d + 1.0;
    if (c->model == QMPC_MODEL_SPARSE) {
      // SparseCMPC: A_d^d B_d = (I + d dt A) B dt = dt B + d dt^2 A B  (A^2 = 0; no drag term)
      coef[1 * h + d] = (double)d * d1 * d1;
      coef[2 * h + d] = 0.0;
    } else {
      coef[1 * h + d] = (2.0 * d + 1.0) * d1 * d1 / 2.0;
      coef[2 * h + d] = (dp * dp * dp - (double)d * d * d) * d1 * d1 * d1 / 6.0;
    }
  }
  for (int p = 0; p < 3; ++p)
    for (int q = 0; q < 3; ++q)
      for (int i = 0; i < h; ++i)
        for (int j = 0; j < h; ++j) {
          double s = 0.0;
          for (int k = (i > j ? i : j); k < h; ++k) s += coef[p * h + (k - i)] * coef[q * h + (k - j)];
          ctab[((p * 3 + q) * hs + i) * hs + j] = s;
        }
  DeviceGuard g(c->device);
  // solves still in flight read the old tables: wait for the stream that orders this handle
  if (c->has_last) HIP_TRY(c, hipStreamSynchronize(c->last_stream));
  HIP_TRY(c, hipMemcpy(c->d_tables, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
  c->tab_dt = c->dt;
  c->tab_h = h;
  c->tab_model = c->model;
  c->is_setup = true;
  // every pool a solve at this horizon can need is allocated now: no solve call allocates (a hipMalloc synchronises
  // the device and cannot be captured into a graph)
  return ensure_pools(c);
}

int qmpc_set_robot(qmpc_handle c, double mass, const double ibody_diag[3], double gravity) {
  if (!c || !ibody_diag || !(mass > 0)) return QMPC_ERR_ARG;
  for (int k = 0; k < 3; ++k)
    if (!(ibody_diag[k] > 0)) return QMPC_ERR_ARG;
  c->mass = mass;
  for (int k = 0; k < 3; ++k) c->ibody[k] = ibody_diag[k];
  c->gravity = gravity;
  return QMPC_OK;
}

int qmpc_settings(qmpc_handle c, int max_iter, double tol) {
  if (!c || max_iter <= 0 || !(tol >= 0)) return QMPC_ERR_ARG;
  c->max_iter = max_iter;
  c->tol = tol;
  return QMPC_OK;
}

int qmpc_set_max_stance(qmpc_handle c, int max_stance_footsteps) {
  if (!c || max_stance_footsteps < 0) return QMPC_ERR_ARG;
  c->max_stance = max_stance_footsteps;
  return ensure_pools(c);  // (a wider hint can make a larger class reachable)
}

int qmpc_set_min_stance(qmpc_handle c, int min_stance_footsteps) {
  if (!c || min_stance_footsteps < 0) return QMPC_ERR_ARG;
  c->min_stance = min_stance_footsteps;
  return ensure_pools(c);
}

int qmpc_set_model(qmpc_handle c, int model) {
  if (!c || (model != QMPC_MODEL_DENSE && model != QMPC_MODEL_SPARSE)) return QMPC_ERR_ARG;
  if (model != c->model) {
    c->model = model;
    if (c->is_setup) return qmpc_setup(c, c->dt, c->horizon, c->mu, c->f_max);  // rebuild the tables
  }
  return QMPC_OK;
}

int qmpc_settings_jcqp(qmpc_handle c, int use_jcqp, int max_iter, double rho, double sigma, double solver_alpha,
                       double terminate) {
  if (!c || use_jcqp < 0 || use_jcqp > 2) return QMPC_ERR_ARG;
  if (use_jcqp && (max_iter <= 0 || !(rho > 0) || !(sigma >= 0) || !(solver_alpha > 0) || !(terminate >= 0)))
    return QMPC_ERR_ARG;
  c->admm_mode = use_jcqp;
  if (use_jcqp) {
    c->admm_max_iter = max_iter;
    c->admm_rho = rho;
    c->admm_sigma = sigma;
    c->admm_alpha = solver_alpha;
    c->admm_term = terminate;
  }
  return ensure_pools(c);  // (the alternate's launch plan can reach the large-problem pool where the exact solve's does not)
}

int qmpc_set_warm_start(qmpc_handle c, int32_t* ws_dev, int shift_steps) {
  if (!c || shift_steps < 0) return QMPC_ERR_ARG;
  c->ws = ws_dev;
  c->ws_shift = shift_steps;
  return QMPC_OK;
}

int qmpc_set_warm_start_min_iters(qmpc_handle c, int min_iters) {
  if (!c || min_iters < 0) return QMPC_ERR_ARG;
  if (min_iters > 0 && !c->order_hint) return QMPC_ERR_STATE;  // the selection reads the counts the order hint keeps
  c->ws_min_iters = min_iters;
  return QMPC_OK;
}

int qmpc_set_debug(qmpc_handle c, double* H_dev, double* g_dev) {
  if (!c) return QMPC_ERR_ARG;
  c->dbg_H = H_dev;
  c->dbg_g = g_dev;
  return QMPC_OK;
}

int qmpc_set_debug_overflow_spin(qmpc_handle c, int probes) {
  if (!c) return QMPC_ERR_ARG;
  c->ov_spin = probes < 0 ? (1 << 22) : probes;
  return QMPC_OK;
}

int qmpc_set_debug_overflow_slices(qmpc_handle c, int n) {
  if (!c) return QMPC_ERR_ARG;
  const int all = c->max_batch < 2048 ? c->max_batch : 2048;  // what qmpc_create allocated
  if (n > all) return QMPC_ERR_ARG;  // more slices than allocated
  c->ov_nslice = n < 0 ? all : n;
  return QMPC_OK;
}

int qmpc_set_split(qmpc_handle c, int on) {
  if (!c) return QMPC_ERR_ARG;
  if (on < 0 || on > 2) return QMPC_ERR_ARG;
  c->split = on;
  return ensure_pools(c);
}

int qmpc_set_block_start(qmpc_handle c, int on) {
  if (!c) return QMPC_ERR_ARG;
  c->block = on != 0;
  return QMPC_OK;
}

int qmpc_set_dense(qmpc_handle c, int mode) {
  if (!c || mode < 0 || mode > 2) return QMPC_ERR_ARG;
  c->dense = mode;
  return QMPC_OK;
}

int qmpc_set_debug_balance(qmpc_handle c, int mode) {
  if (!c || mode < 0 || mode > 2) return QMPC_ERR_ARG;
  c->bal_debug = mode;
  return QMPC_OK;
}

int qmpc_debug_keys(qmpc_handle c, int batch, const qmpc_inputs* in, int32_t* nst_dev, float* score_dev, float* demand_dev,
                    void* stream_) {
  if (!c || !in || !nst_dev || !score_dev || !demand_dev) return QMPC_ERR_ARG;
  if (!c->is_setup) return QMPC_ERR_STATE;
  if (batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  if (!in->p || !in->v || !in->q || !in->w || !in->r || !in->yaw || !in->traj || !in->gait || !in->weights || !in->alpha ||
      ((uintptr_t)in->gait & 3u))
    return QMPC_ERR_ARG;
  DeviceGuard g(c->device);
  QmpcParams P;
  std::memset(&P, 0, sizeof(P));
  P.p = in->p; P.v = in->v; P.q = in->q; P.w = in->w; P.r = in->r; P.yaw = in->yaw;
  P.traj = in->traj; P.gait = in->gait; P.weights = in->weights; P.alpha = in->alpha;
  P.weights_stride = in->weights_stride;
  P.alpha_stride = in->alpha_stride;
  P.batch = batch; P.horizon = c->horizon;
  P.dt = c->dt;
  P.mu_inv = (double)(1.f / (float)c->mu);
  P.mass = c->mass;
  P.gravity = c->gravity;
  HIP_TRY(c, qmpc_launch_keys(&P, nst_dev, score_dev, demand_dev, (hipStream_t)stream_));
  return QMPC_OK;
}

int qmpc_set_size_order(qmpc_handle c, int on) {
  if (!c || on < 0 || on > 1) return QMPC_ERR_ARG;
  c->size_order = on;
  return QMPC_OK;
}

int qmpc_set_order_hint(qmpc_handle c, int mode) {
  if (!c || mode < 0 || mode > 1) return QMPC_ERR_ARG;
  c->order_hint = mode;
  c->hint_batch = 0;
  return QMPC_OK;
}

int qmpc_set_chunks(qmpc_handle c, int n) {
  if (!c || n < 0 || n > 64) return QMPC_ERR_ARG;
  c->chunks = n;
  return QMPC_OK;
}

int qmpc_set_debug_engine_events(qmpc_handle c, int n) {
  if (!c || n < 0) return QMPC_ERR_ARG;
  c->dbg_engine_events = n;
  return QMPC_OK;
}

int qmpc_reserve(qmpc_handle c) {
  if (!c) return QMPC_ERR_ARG;
  if (!c->is_setup) return QMPC_ERR_STATE;
  DeviceGuard g(c->device);
  return ensure_pools(c);  // (qmpc_setup and the hint setters have done this already: kept for callers of earlier versions)
}

int qmpc_set_debug_pool_busy(qmpc_handle c, int on) {
  if (!c) return QMPC_ERR_ARG;
  c->dbg_pool_busy = on != 0;
  return QMPC_OK;
}

int qmpc_set_debug_aux(qmpc_handle c, double* aux_dev) {
  if (!c) return QMPC_ERR_ARG;
  c->dbg_aux = aux_dev;
  return QMPC_OK;
}

int qmpc_debug_ld(qmpc_handle) { return QMPC_DBG_LD; }

// test hook: a work item of the decoupled path after a solve -- which: 0 = 128-row class, 1 = 192-row class, 2 = large
// problems; hinv_host: ld x ld doubles (ld = 128 / 192 / 448), xu_host: ld doubles, hdr4: {rid, n, nst, status0}
int qmpc_debug_read_item(qmpc_handle c, int which, int item, double* hinv_host, double* xu_host, int* hdr4) {
  if (!c || which < 0 || which > 2 || c->wk_cap[which] <= 0 || item < 0 || item >= c->wk_cap[which]) return QMPC_ERR_ARG;
  DeviceGuard g(c->device);
  const size_t ld = which == 0 ? 128 : (which == 1 ? 192 : QMPC_BIG_LD);
  HIP_TRY(c, hipDeviceSynchronize());
  if (hinv_host) HIP_TRY(c, hipMemcpy(hinv_host, c->d_wk_hinv[which] + (size_t)item * ld * ld, sizeof(double) * ld * ld, hipMemcpyDeviceToHost));
  if (xu_host) HIP_TRY(c, hipMemcpy(xu_host, c->d_wk_xu[which] + (size_t)item * ld, sizeof(double) * ld, hipMemcpyDeviceToHost));
  if (hdr4) HIP_TRY(c, hipMemcpy(hdr4, c->d_wk_hdr[which] + item, sizeof(int) * 4, hipMemcpyDeviceToHost));
  return QMPC_OK;
}

int qmpc_debug_read_counts(qmpc_handle c, int* host768) {
  if (!c || !host768) return QMPC_ERR_ARG;
  DeviceGuard g(c->device);
  HIP_TRY(c, hipDeviceSynchronize());
  HIP_TRY(c, hipMemcpy(host768, c->d_counts, sizeof(int) * 3 * QMPC_COUNTERS, hipMemcpyDeviceToHost));
  return QMPC_OK;
}

int qmpc_set_debug_clock(qmpc_handle c, long long* clk_dev) {
  if (!c) return QMPC_ERR_ARG;
  c->dbg_clk = clk_dev;
  return QMPC_OK;
}

}  // extern "C"

namespace {

bool command_ok(const qmpc_command* cmd) {
  return cmd->position && cmd->v_world && cmd->omega_world && cmd->orientation && cmd->rpy && cmd->r_body &&
         cmd->p_foot && cmd->vel_des && cmd->yaw_des_true && cmd->rpy_comp && cmd->gait_offsets &&
         cmd->gait_durations && cmd->gait_iteration && cmd->world_position_desired && cmd->x_comp_integral &&
         !(cmd->gait_type && !cmd->stand_traj);  // a standing robot needs its stand_traj row
}

// Which size classes a solve launches, and which of them through work items (the decoupled path): ONE rule, used by
// the solve and by the allocation of the pools.
//   chain 1 -> 4 -> 2 -> 3 (64 / 96 / 128 / 192 padded rows), n_r = 3 * stance foot-steps; classes k0 .. k1 - 1 are
//   launched; long_h: the large-problem stage behind the 192-row class (192 < n_r <= 432, horizons above 16)
struct ClassPlan {
  int k0 = 0, k1 = 0;
  bool long_h = false;
  bool split[4] = {false, false, false, false};
};
const int kChain[4] = {1, 4, 2, 3};
const int kRows[4] = {64, 96, 128, 192};

ClassPlan plan_classes(const qmpc_ctx* c, int admm_mode, bool warm) {
  ClassPlan pl;
  const bool full_problem = (admm_mode == 1), exact_cold = !admm_mode && !warm;
  const int h = c->horizon;
  const int nmax = 12 * h;
  int nclass = 4;
  for (int k = 0; k < 4; ++k)
    if (nmax <= kRows[k]) { nclass = k + 1; break; }
  // a caller that knows its gaits can bound the reduced size (qmpc_set_max_stance):
  // larger classes are then not even launched; violators are flagged WS_FULL
  pl.k1 = nclass;
  if (c->max_stance > 0 && !full_problem) {  // (use_jcqp = 1: every foot-step is a variable block, n_r = 12 h for all robots)
    const int nb = 3 * c->max_stance;
    int hc = 4;
    for (int k = 0; k < 4; ++k)
      if (nb <= kRows[k]) { hc = k + 1; break; }
    if (hc < pl.k1) pl.k1 = hc;
  }
  // ... and with a lower bound the classes that are too small for every robot are skipped
  while (pl.k0 + 1 < pl.k1 && 3 * (full_problem ? 4 * h : c->min_stance) > kRows[pl.k0]) ++pl.k0;
  if (h > QMPC_LONG_HORIZON) {
    // long horizons (up to K_MAX_GAIT_SEGMENTS = 36): the 192-row class alone has the threads (12 h tracking-error
    // entries, one per thread) and the LDS (h x h coefficient tables) to assemble them -- it takes every robot; one with
    // more than 64 stance foot-steps goes on to the large-problem path
    pl.k0 = 3;
    pl.k1 = 4;
    // (the large-problem stage: skipped when the caller's size hint, qmpc_set_max_stance, rules such robots out -- a
    //  violator is reported like any other; use_jcqp = 1 makes every robot a large problem, 12 h variables)
    pl.long_h = full_problem || !(c->max_stance > 0 && 3 * c->max_stance <= 192);
  }
  // decoupled path (128- and 192-row classes, exact solve, cold start): sweep kernel -> work items -> engine kernel ->
  // (rarely) the monolithic kernel on the robots the engine handed back.  Automatic: a small batch is latency-bound --
  // one workgroup per CU either way -- and the one-kernel path has one launch and no trip through L2 on it: measured
  // break-even ~300 robots in the 128-row class, below 128 in the 192-row class.  Decided by the HANDLE's size, not
  // the call's: a robot's result does not depend on the batch it is solved in -- the two paths agree to ~1e-14
  // relative, not bit for bit
  for (int k = pl.k0; k < pl.k1; ++k) {
    const bool big = c->split == 2 || c->max_batch >= (kChain[k] == 2 ? 384 : 128);
    pl.split[k] = c->split && big && (kChain[k] == 2 || kChain[k] == 3) && exact_cold;
  }
  return pl;
}

// one pool of work items: sk 0 = 128-row class, 1 = 192-row class, 2 = large problems (448-row items)
int ensure_items(qmpc_ctx* c, int sk) {
  if (c->wk_cap[sk] > 0) return QMPC_OK;  // (set last: a pool counts as present only when ALL its arrays are)
  static const int lim[3] = {QMPC_ITEMS_C2, QMPC_ITEMS_C3, QMPC_ITEMS_BIG};
  const int rb = sk == 0 ? 2 : (sk == 1 ? 3 : 5);
  const size_t ld = sk == 0 ? 128 : (sk == 1 ? 192 : QMPC_BIG_LD);
  const size_t cap = (size_t)(c->max_batch < lim[sk] ? c->max_batch : lim[sk]);
  // (the engine grid never exceeds the resident workgroups: slice = blockIdx.x)
  size_t wgs = (size_t)qmpc_engine_resident(rb);
  if (wgs == 0 || wgs > cap) wgs = cap;
  const size_t ev = sk == 0 ? 128 + 64 : (sk == 1 ? 192 + 128 : QMPC_BIG_LD + 192);
  // all or nothing: an allocation that fails half-way (the pools are up to 1.5 GiB) must not leave a pool that LOOKS
  // present -- the next solve would divide by wk_cap == 0 or launch on null arrays
  hipError_t e = hipMalloc(&c->d_wk_hinv[sk], sizeof(double) * cap * ld * ld);
  if (e == hipSuccess) e = hipMalloc(&c->d_wk_xu[sk], sizeof(double) * cap * ld);
  if (e == hipSuccess) e = hipMalloc(&c->d_wk_hdr[sk], sizeof(QmpcWorkHdr) * cap);
  if (e == hipSuccess) e = hipMalloc(&c->d_wk_order[sk], sizeof(int) * QMPC_ORDER_BUCKETS * cap);
  if (e == hipSuccess) e = hipMalloc(&c->d_wk_ovf[sk], sizeof(double) * wgs * QMPC_ENGINE_OVF_EVENTS * ev);
  if (e != hipSuccess) {
    if (c->d_wk_hinv[sk]) hipFree(c->d_wk_hinv[sk]);
    if (c->d_wk_xu[sk]) hipFree(c->d_wk_xu[sk]);
    if (c->d_wk_hdr[sk]) hipFree(c->d_wk_hdr[sk]);
    if (c->d_wk_order[sk]) hipFree(c->d_wk_order[sk]);
    if (c->d_wk_ovf[sk]) hipFree(c->d_wk_ovf[sk]);
    c->d_wk_hinv[sk] = nullptr; c->d_wk_xu[sk] = nullptr; c->d_wk_hdr[sk] = nullptr;
    c->d_wk_order[sk] = nullptr; c->d_wk_ovf[sk] = nullptr;
    (void)hipGetLastError();
    return fail(c, e, "work-item pool allocation");
  }
  c->wk_cap[sk] = (int)cap;
  return QMPC_OK;
}

// Every pool the current setup, stance hints and split mode can reach, allocated NOW (qmpc_setup, the hint setters,
// qmpc_set_split, qmpc_reserve call this): a solve call never allocates.  The pools are bounded by QMPC_ITEMS_*,
// not by max_batch: a call with more robots than items runs the class in consecutive chunks.
int ensure_pools(qmpc_ctx* c) {
  if (!c->is_setup) return QMPC_OK;
  DeviceGuard g(c->device);
  // the UNION of the plans a solve on this handle can make: the exact solve, and -- when the JCQP alternate is selected --
  // its own plan (use_jcqp = 1 makes every robot of a long horizon a large problem whatever the stance hint says)
  const int modes[2] = {0, c->admm_mode};
  for (int m = 0; m < (c->admm_mode ? 2 : 1); ++m) {
    const ClassPlan pl = plan_classes(c, modes[m], false);
    for (int k = pl.k0; k < pl.k1; ++k)
      if (pl.split[k])
        if (const int rc = ensure_items(c, kChain[k] == 2 ? 0 : 1)) return rc;
    if (pl.long_h)
      if (const int rc = ensure_items(c, 2)) return rc;
  }
  return QMPC_OK;
}

// one solve: inputs either as the record (`in`) or as the controller command (`cmd`, record built in stage 0)
int solve_impl(qmpc_ctx* c, int batch, const qmpc_inputs* in, const qmpc_command* cmd, const qmpc_outputs* out,
               float* f_ff, void* stream_) {
  if (!c || (!in && !cmd) || !out) return QMPC_ERR_ARG;
  if (!c->is_setup) return QMPC_ERR_STATE;
  if (batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  if (!out->grf || !out->status) return QMPC_ERR_ARG;
  if (in) {
    if (!in->p || !in->v || !in->q || !in->w || !in->r || !in->yaw || !in->traj || !in->gait ||
        !in->weights || !in->alpha || !in->x_drag)
      return QMPC_ERR_ARG;
    if ((in->weights_stride != 0 && in->weights_stride != 12) || (in->alpha_stride & ~1) ||
        (in->x_drag_stride & ~1))
      return QMPC_ERR_ARG;
  } else if (!command_ok(cmd)) {
    return QMPC_ERR_ARG;
  }
  hipStream_t stream = (hipStream_t)stream_;
  DeviceGuard g(c->device);
  const int h = c->horizon;
  const bool capturing = is_capturing(stream);
  if (const int rc = order_after_previous(c, stream)) return rc;
  if (c->device_error && !capturing) {
    // the previous call (or set-up step) of this handle failed on the device: whatever its kernels left in the
    // recycled-slice flags and the per-CU placement words is void.  Re-armed with the fill kernel (no memset node: they
    // replay wrongly when captured, and this path must not differ from the normal one in kind)
    if (c->d_ovflags && c->ov_nslice > 0) HIP_TRY(c, fill_ints(c->d_ovflags, c->max_batch < 2048 ? c->max_batch : 2048, 0, stream));
    c->device_error = false;
  }

  QmpcParams P;
  std::memset(&P, 0, sizeof(P));
  if (in) {
    P.p = in->p; P.v = in->v; P.q = in->q; P.w = in->w; P.r = in->r; P.yaw = in->yaw;
    P.traj = in->traj; P.gait = in->gait; P.weights = in->weights; P.alpha = in->alpha;
    P.x_drag = in->x_drag;
    P.weights_stride = in->weights_stride;
    P.alpha_stride = in->alpha_stride;
    P.x_drag_stride = in->x_drag_stride;
  } else {
    P.c_position = cmd->position; P.c_v_world = cmd->v_world; P.c_omega_world = cmd->omega_world;
    P.c_orientation = cmd->orientation; P.c_rpy = cmd->rpy; P.c_r_body = cmd->r_body; P.c_p_foot = cmd->p_foot;
    P.c_vel_des = cmd->vel_des; P.c_yaw_des_true = cmd->yaw_des_true; P.c_rpy_comp = cmd->rpy_comp;
    P.c_stand_traj = cmd->stand_traj; P.c_rp_des = cmd->rp_des; P.c_gait_type = cmd->gait_type;
    P.c_gait_offsets = cmd->gait_offsets; P.c_gait_durations = cmd->gait_durations;
    P.c_gait_iteration = cmd->gait_iteration; P.c_wpd = cmd->world_position_desired;
    P.c_xci = cmd->x_comp_integral; P.c_body_height = cmd->body_height; P.c_omni_mode = cmd->omni_mode;
    P.f_ff = f_ff;
  }
  P.grf = out->grf; P.soln = out->soln; P.status = out->status; P.iters = out->iters;
  P.batch = batch; P.horizon = h;
  P.dt = c->dt;
  // fpt mu = 1.f/setup->mu  (SolverMPC.cpp:366), float arithmetic
  const float mi = 1.f / (float)c->mu;
  P.mu_inv = (double)mi;
  P.inv_fr_norm = 1.0 / std::sqrt(P.mu_inv * P.mu_inv + 1.0);
  P.f_max = c->f_max;
  P.mass = c->mass;
  for (int k = 0; k < 3; ++k) P.ibody[k] = c->ibody[k];
  P.inv_mass = 1.0 / c->mass;
  for (int k = 0; k < 3; ++k) P.inv_ibody[k] = 1.0 / c->ibody[k];
  P.gravity = c->gravity;
  P.model = c->model;
  P.coef = c->d_tables;
  P.ctab = c->d_tables + 3 * h;
  P.max_iter = c->max_iter;
  P.tol = c->tol;
  P.ws = c->ws;
  P.ws_shift = c->ws_shift;
  // selective warm start: the selection reads the previous call's iteration counts, which exist only while the order hint is on,
  // the call is eager (not captured) and the previous call had this batch size; without them NOBODY qualifies (every robot
  // starts cold) -- eager and captured calls, first and later calls behave alike
  P.ws_min_iters = (c->ws_min_iters > 0 && !(c->order_hint && !capturing && c->hint_batch == batch)) ? 0x7fffffff : c->ws_min_iters;
  if (c->admm_mode && in) {  // (record mode only: the command mode always solves exactly)
    P.admm_mode = c->admm_mode;
    P.admm_max_iter = c->admm_max_iter;
    P.admm_rho = c->admm_rho;
    P.admm_sigma = c->admm_sigma;
    P.admm_alpha = c->admm_alpha;
    P.admm_term = c->admm_term;
  }
  P.dbg_H = c->dbg_H;
  P.dbg_g = c->dbg_g;
  P.dbg_aux = c->dbg_aux;
  P.dbg_clk = c->dbg_clk;
  P.hint_iters = (c->order_hint && !capturing) ? c->d_hint_iters : nullptr;
  // (selective warm start without usable counts -- hint off, captured call: the kernel compares the count array against a
  //  threshold nobody reaches, so it needs the array; the counts it leaves there are not used while the hint is off)
  if (c->ws && c->ws_min_iters > 0 && !P.hint_iters) P.hint_iters = c->d_hint_iters;
  P.cu_slots = c->d_cu_slots;
  P.bal_debug = c->bal_debug;

  P.ovpool = c->d_ovpool;
  P.ov_nslice = c->ov_nslice;
  P.ov_flags = c->d_ovflags;
  P.ov_spin = c->ov_spin;
  P.evpool = c->d_evpool;
  P.evflags = c->d_evflags;
  P.ev_nslot = c->ev_nslot;
  P.ev_spin = c->dbg_pool_busy ? 4 : (1 << 16);

  // ---- the launch plan, and everything that can refuse the call, BEFORE the call counter moves: the counter sets
  // ping-pong between calls and each call's first kernel zeroes the NEXT call's set, so a call that took a set without
  // launching would leave the following call on counters nobody cleared
  const ClassPlan pl = plan_classes(c, P.admm_mode, P.ws != nullptr);
  for (int k = pl.k0; k < pl.k1; ++k)
    if (pl.split[k] && c->wk_cap[kChain[k] == 2 ? 0 : 1] <= 0) {
      c->err = "work-item pool missing (qmpc_setup / qmpc_reserve allocate it)";
      return QMPC_ERR_STATE;
    }
  if (pl.long_h && c->wk_cap[2] <= 0) {
    c->err = "large-problem pool missing (qmpc_setup / qmpc_reserve allocate it)";
    return QMPC_ERR_STATE;
  }
  int* cnt;       // this call's counters
  int* cnt_next;  // the set this call's first kernel clears (the next call's)
  if (capturing) {
    // a call captured into a hipGraph is replayed with the SAME kernel arguments every time: it cannot take part in
    // the ping-pong (its set would be dirty from the previous replay).  Captured calls use a set of their own, cleared
    // by a small kernel node in front of the call's kernels; the eager calls' two sets are not touched
    cnt = c->d_counts + QMPC_COUNTERS * 2;
    cnt_next = nullptr;
    HIP_TRY(c, fill_ints(cnt, QMPC_COUNTERS, 0, stream));
  } else {
    const unsigned set = c->call_no & 1u;
    c->call_no++;
    cnt = c->d_counts + QMPC_COUNTERS * set;
    cnt_next = c->d_counts + QMPC_COUNTERS * (set ^ 1u);
  }
  P.ov_count = cnt + QMPC_CNT_OV;                         // slices of the overflow pool handed out in this call
  bool first = true;                            // the next launch is the first of the call: it carries clear_counts

  // one item class of the decoupled path (sk 0 / 1: sweep kernel of class rb; sk 2: the large-problem producer): the
  // robots [0, batch) -- or the entries of `list` -- in consecutive chunks of at most wk_cap[sk], every chunk a
  // producer launch and an engine launch on the caller's stream, the pool reused from chunk to chunk
  auto run_items = [&](const QmpcParams& base, int sk, int rb, const int* list, int* count) -> int {
    QmpcParams A = base;
    A.wk_hinv = c->d_wk_hinv[sk];
    A.wk_xu = c->d_wk_xu[sk];
    A.wk_hdr = c->d_wk_hdr[sk];
    A.wk_order = c->d_wk_order[sk];
    A.wk_ovf = c->d_wk_ovf[sk];
    A.wk_ld = sk == 0 ? 128 : (sk == 1 ? 192 : QMPC_BIG_LD);
    A.wk_cap = c->wk_cap[sk];
    A.wk_base = 0;
    A.wk_kev = c->dbg_engine_events > 0 ? c->dbg_engine_events : (1 << 20);
    A.wk_block = (sk < 2 && c->block) ? 1 : 0;
    A.fb_list = c->d_fb_lists + (size_t)sk * c->max_batch;
    A.fb_count = cnt + QMPC_CNT_FB + sk;
    A.list = list;
    A.count = count;
    int nch = (batch + c->wk_cap[sk] - 1) / c->wk_cap[sk];
    if (c->chunks > nch) nch = c->chunks < batch ? c->chunks : batch;
    const int per = (batch + nch - 1) / nch;  // (<= wk_cap[sk])
    for (int ch = 0; ch < nch; ++ch) {
      const int lo = ch * per, hi = (lo + per < batch) ? lo + per : batch;
      if (lo >= hi) break;
      int* grp = cnt + QMPC_CNT_GRP(sk, ch & 1);
      A.wk_count = grp;
      A.wk_qhead = grp + 1;
      A.qhead = list ? grp + 2 : nullptr;
      A.wk_bucket = grp + 8;
      A.wk_zero = nullptr;
      A.rid0 = lo;
      A.list_hi = hi;
      A.clear_counts = first ? cnt_next : nullptr;
      first = false;
      int grid = hi - lo;
      if (list || sk == 2) {
        const int res = sk == 2 ? qmpc_resident_blocks(3) : qmpc_resident_sweep(rb);  // (the large-problem producer has the 192-row class's footprint)
        if (res > 0 && res < grid) grid = res;
      }
      if (sk == 2) HIP_TRY(c, qmpc_big_launch(&A, grid, stream));
      else HIP_TRY(c, qmpc_launch_sweep(rb, &A, grid, stream));
      QmpcParams B = A;  // the engine: one robot per workgroup, the chunk's items as a queue
      B.list = nullptr; B.count = nullptr; B.qhead = nullptr; B.clear_counts = nullptr;
      B.next_list = nullptr; B.next_count = nullptr;
      B.wk_zero = (ch + 1 < nch) ? cnt + QMPC_CNT_GRP(sk, (ch + 1) & 1) : nullptr;
      int gb = hi - lo;
      {
        const int res = qmpc_engine_resident(sk == 2 ? 5 : rb);
        if (res > 0 && res < gb) gb = res;
      }
      if (sk == 2 && base.admm_mode) {
        // JCQP alternate on the large problems: the producer left M^-1 and the gradient; the ADMM kernel consumes the items
        gb = (hi - lo) < 2048 ? (hi - lo) : 2048;
        HIP_TRY(c, qmpc_admm_big_launch(&B, gb, stream));
      } else {
        HIP_TRY(c, qmpc_engine_launch(sk == 2 ? 5 : rb, &B, gb, stream));
      }
    }
    if (sk == 2 && base.admm_mode) return QMPC_OK;  // (the ADMM hands nothing back)
    // robots handed back (event capacity exceeded, lost definiteness): the monolithic kernel, list-consuming.  The
    // large problems have no class to fall back to: the 192-row class's stage 0 REPORTS them (QMPC_ST_WS_FULL)
    QmpcParams F = base;
    F.list = A.fb_list; F.count = A.fb_count; F.qhead = cnt + QMPC_CNT_FBQ + sk; F.clear_counts = nullptr;
    F.list_hi = 0x7fffffff;
    F.rid0 = 0;
    F.next_list = nullptr; F.next_count = nullptr;
    F.status_or = sk == 2 ? 0 : QMPC_DEV_ST_FALLBACK;
    const int frb = sk == 0 ? 2 : 3;
    if (frb == 3 && c->d_evflags)
      HIP_TRY(c, fill_ints(c->d_evflags, c->ev_nslot, c->dbg_pool_busy ? 1 : 0, stream));
    int gf = batch;
    {
      const int res = qmpc_resident_blocks(frb);
      if (res > 0 && res < gf) gf = res;
    }
    HIP_TRY(c, qmpc_launch(frb, &F, gf, stream));
    return QMPC_OK;
  };

  // size classes by padded rows: 64 (kernel class 1), 96 (class 4), 128 (class 2), 192 (class 3);
  // n_r = 3 * stance foot-steps.  The first class is launched over the whole batch; a robot that
  // does not fit appends itself to the list of the next one.
  for (int k = pl.k0; k < pl.k1; ++k) {
    const bool listed = k > pl.k0;
    P.list = listed ? c->d_lists + (size_t)(k - 1) * c->max_batch : nullptr;
    P.count = listed ? cnt + (k - 1) : nullptr;
    P.qhead = listed ? cnt + 4 + (k - 1) : nullptr;
    const bool more = k + 1 < pl.k1;
    P.next_list = more ? c->d_lists + (size_t)k * c->max_batch : nullptr;
    P.next_count = more ? cnt + k : nullptr;
    if (pl.long_h && kChain[k] == 3) {  // robots beyond 192 rows go on to the large-problem producer
      P.next_list = c->d_lists + (size_t)3 * c->max_batch;
      P.next_count = cnt + QMPC_CNT_BIGLIST;
    }
    if (pl.split[k]) {
      if (const int rc = run_items(P, kChain[k] == 2 ? 0 : 1, kChain[k], P.list, P.count)) return rc;
      continue;
    }
    // the first class of the chain: one workgroup per robot; the later ones: one per resident slot, the list
    // is consumed as a queue (no workgroup is dispatched only to find its list entry missing)
    if (kChain[k] == 3 && c->d_evflags)  // no kernel of this handle is in flight on another stream (ordered above)
      HIP_TRY(c, fill_ints(c->d_evflags, c->ev_nslot, c->dbg_pool_busy ? 1 : 0, stream));
    P.clear_counts = first ? cnt_next : nullptr;
    first = false;
    int grid = batch;
    if (listed) {
      const int res = qmpc_resident_blocks(kChain[k]);
      if (res > 0 && res < grid) grid = res;
    }
    // the 64-row class alone in the chain (the stance hint says every robot fits it), on a handle made for large batches:
    // its five-workgroups-per-CU instantiation (same arithmetic, bit-identical results: tests) -- a launch of several rounds
    // is bound by instruction issue, and the fifth wave per SIMD fills what four leave (trot: +3.5 % at 2048 robots, +8 % at 4096, +14 % from
    // 8192 on: 3.76e7 -> 4.30e7 QP/s at 16384; mixed gaits +3 / +7 / +11 %: tools/dense_threshold.py); one round of workgroups
    // (batch 1024) is bound by its slowest robot and loses 1 - 11 % to the 96-VGPR code.  By the
    // HANDLE's size, never the call's.  Not for chains with larger classes: their robots iterate long, and 16 events in LDS
    // instead of 28 send a thousand of them to the overflow pool (configs[4]: -40 %)
    int kcls = kChain[k];
    // Chains with larger classes behind it (configs[4]: random contact tables): their robots iterate longer, and 16 events
    // in LDS instead of 28 send about one in eight of them to the overflow pool; the launch still gains (configs[4] 492 -> 482 us
    // per 8192 robots, +2 %) as long as every one of them finds a slice there -- which, the slices being RECYCLED within a call
    // since round 5 (a flag per slice, released when its robot finishes: the need is bounded by the robots in flight), holds for
    // any batch size.  So this choice, too, is made by the HANDLE's size alone (until round 4 it also looked at the call's:
    // batch <= 4 x the slices)
    if (kcls == 1 && pl.k1 - pl.k0 > 1 && !pl.long_h && c->dense == 1 && c->max_batch >= 2048) kcls = 6;
    if (kcls == 1 && pl.k1 - pl.k0 == 1 && !pl.long_h && (c->dense == 2 || (c->dense == 1 && c->max_batch >= 2048))) kcls = 6;
    // order hint: the first class of the chain, launched over more robots than it has resident workgroups (several rounds:
    // the launch ends with whichever hard robot started last), takes the robots in the order of their iteration counts in the
    // previous call -- the same robots one MPC cycle earlier -- longest first.  Results do not depend on the order.
    // A launch of ONE round (the order cannot matter) uses the counts differently: the robots the previous call found hard
    // keep the highest issue priority through their sweep (qmpc_device.h: hint_hard) -- batch 1024, trot: 2.42e7 -> 2.75e7 QP/s.
    // (Only there: in a launch of many rounds it costs 3 %, measured at 16384 robots.)
    P.order = nullptr;
    P.hint_hard = 0;
    bool use_hint_keys = false;
    if (!listed && c->order_hint && !capturing && !P.admm_mode) {
      if (batch > qmpc_resident_blocks(kcls)) {
        if (c->hint_batch == batch) {
          if (c->hint_prepass) {  // (until round 6, kept for comparison: QMPC_HINT_PREPASS=1)
            hipLaunchKernelGGL(qmpc_order_kernel, dim3(1), dim3(1024), 0, stream, (const int*)c->d_hint_iters, c->d_order, batch);
            HIP_TRY(c, hipGetLastError());
            P.order = c->d_order;
          } else {
            use_hint_keys = true;  // the permutation is built inside the launch (below), keys = the previous call's counts
          }
        }
      } else if (2 * batch > qmpc_resident_blocks(kcls)) {  // (workgroups share CUs: below that priority has nobody to act on)
        // the largest count of the previous one-round call / of this one / cleared for the next: three slots in rotation
        const unsigned hc = c->hint_call++;
        P.hint_max_r = c->d_hint_max + (hc + 2) % 3;
        P.hint_max_w = c->d_hint_max + hc % 3;
        P.hint_max_z = c->d_hint_max + (hc + 1) % 3;
        if (c->hint_batch == batch) P.hint_hard = c->hint_hard;
      }
    }
    // size order: no usable hint, several rounds, contact tables in memory (record mode) and 8-byte aligned.  The first round keeps
    // robot = workgroup index (its workgroups start before anything can be known); the builders (the first workgroups, one
    // segment of the rest each) need a few microseconds, for which the workgroups that follow robots only handed on may have to wait
    // (measured: no loss on configs[4], where a third of the first round is handed on)
    P.so_order = nullptr;
    const bool by_size = c->size_order && !cmd && P.gait && ((uintptr_t)P.gait & 7u) == 0;
    if (!listed && !capturing && !P.admm_mode && !P.order && (use_hint_keys || by_size)) {
      const int res = qmpc_resident_blocks(kcls);
      if (res > 0 && batch > res) {
        // (the unsorted head beyond the first round: none by size -- measured 0 / 15 / 30 / 50 % of a round: 0 is best or equal
        //  everywhere, +2.6 % on configs[2] --, half a round by the hint's counts: 0 / 25 / 50 %: 3.92e7 / 3.96e7 / 4.05e7 on
        //  configs[2] with exact counts, equal elsewhere -- robots of one COUNT side by side run their engine phases together)
        const int head = (int)((long long)res * (use_hint_keys ? c->so_first_pct_hint : c->so_first_pct) / 100);
        const int half = (batch - res) / 2 < head ? (batch - res) / 2 : head;
        P.so_first = (res + half + 7) & ~7;
        // (a launch of many rounds: only its last five are ordered -- a robot lasts three or four rounds at most, so nothing that
        //  starts earlier can end the launch, and every reader pays a memory round trip for its entry: trot, 16384 robots, -3.9 %
        //  with everything ordered)
        if (batch - c->so_tail_rounds * res > P.so_first) P.so_first = (batch - c->so_tail_rounds * res + 7) & ~7;
        const int n = batch - P.so_first;
        // (next to nothing to order: the builders and the sixteen first-round places per segment cost more than the order gives)
        if (c->so_min_div * n >= res && 17 * 8 * ((n + 8 * 4080 - 1) / (8 * 4080)) <= res) {
          // strided segments of at most 4096 robots (QMPC_SO_SEG of qmpc_kernels.hip: the builder's LDS scratch)
          // (a multiple of 8, and so_first too: place b of segment j has b % 8 == j % 8 -- readers and builder on one XCD)
          // (QMPC_SO_HEAD = 16 places of the first round per segment on top: 17 nseg workgroups in front of so_first)
          P.so_nseg = 8 * ((n + 8 * 4080 - 1) / (8 * 4080));
          P.so_order = c->d_so_order;
          P.so_far = c->d_so_order + c->max_batch;
          P.so_tag = ++c->so_call;
          if (P.so_tag == 0) P.so_tag = ++c->so_call;  // (0 is what fresh memory holds)
          P.so_maxfit = kRows[k] / 3;
          P.so_hint = use_hint_keys ? c->d_hint_iters : nullptr;
        }
      }
    }
    // ONE round, full CUs, no usable hint: the sweep's issue priority is staged per CU by the robots' scores (one atomic maximum on
    // the CU's word: qmpc_kernels.hip, stage 0) -- same conditions as the size order (record mode: the proxy reads the record)
    P.prio_cu = nullptr;
    if (!listed && !capturing && !P.admm_mode && c->size_order && !cmd && P.gait && ((uintptr_t)P.gait & 3u) == 0 && P.hint_hard <= 0) {
      const int res = qmpc_resident_blocks(kcls);
      // (only where the CUs are full: at three workgroups per CU -- 768 robots on 1024 slots -- the staging costs 3 %)
      if (res > 0 && batch <= res && 8 * batch > 7 * res) {
        P.prio_cu = c->d_prio_cu;
        P.prio_tag = ++c->prio_call;
        if (P.prio_tag == 0) {  // (the call number wrapped: the words start again)
          HIP_TRY(c, fill_ints(reinterpret_cast<int*>(c->d_prio_cu), 2 * 2048, 0, stream));
          P.prio_tag = ++c->prio_call;
        }
      }
    }
    HIP_TRY(c, qmpc_launch(kcls, &P, grid, stream));
    P.prio_cu = nullptr;
    P.so_order = nullptr;
    P.order = nullptr;
    P.hint_hard = 0;
    P.hint_max_z = nullptr;
    P.hint_max_w = nullptr;  // (the later classes of a chain are queues of several rounds)
  }
  if (c->order_hint && !capturing && !P.admm_mode) c->hint_batch = batch;
  if (pl.long_h) {
    // ---- the large problems (192 < n_r <= 432: all feet down beyond 16 segments, a trot beyond 32): H in global memory,
    // block sweep, the seven-block engine; what that engine cannot hold is REPORTED.  Normally the list is empty: launches
    // that find nothing to do
    QmpcParams A = P;
    A.next_list = nullptr; A.next_count = nullptr;
    A.status_or = 0;
    A.clear_counts = nullptr;
    if (const int rc = run_items(A, 2, 3, c->d_lists + (size_t)3 * c->max_batch, cnt + QMPC_CNT_BIGLIST)) return rc;
  }
  return QMPC_OK;
}

}  // namespace

extern "C" {

int qmpc_solve(qmpc_handle c, int batch, const qmpc_inputs* in, const qmpc_outputs* out, void* stream) {
  if (!in) return QMPC_ERR_ARG;
  return solve_impl(c, batch, in, nullptr, out, nullptr, stream);
}

int qmpc_solve_commands(qmpc_handle c, int batch, const qmpc_command* cmd, const qmpc_outputs* out, float* f_ff,
                        void* stream) {
  if (!cmd) return QMPC_ERR_ARG;
  return solve_impl(c, batch, nullptr, cmd, out, f_ff, stream);
}

int qmpc_pack(qmpc_handle c, int batch, const qmpc_command* cmd, const qmpc_record* rec, void* stream_) {
  if (!c || !cmd || !rec) return QMPC_ERR_ARG;
  if (!c->is_setup) return QMPC_ERR_STATE;
  if (batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  if (!command_ok(cmd)) return QMPC_ERR_ARG;
  if (!rec->p || !rec->v || !rec->q || !rec->w || !rec->r || !rec->yaw || !rec->traj || !rec->gait || !rec->x_drag)
    return QMPC_ERR_ARG;
  DeviceGuard g(c->device);
  if (const int rc = order_after_previous(c, (hipStream_t)stream_)) return rc;
  HIP_TRY(c, qmpc_launch_pack(cmd, rec, batch, c->horizon, (float)c->dt, (hipStream_t)stream_));
  return QMPC_OK;
}

int qmpc_forces_to_body(qmpc_handle c, int batch, const float* r_body, const float* grf, float* f_ff, void* stream_) {
  if (!c || !r_body || !grf || !f_ff) return QMPC_ERR_ARG;
  if (batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  DeviceGuard g(c->device);
  if (const int rc = order_after_previous(c, (hipStream_t)stream_)) return rc;
  HIP_TRY(c, qmpc_launch_f2b(r_body, grf, f_ff, batch, (hipStream_t)stream_));
  return QMPC_OK;
}

}  // extern "C"

namespace {

// host-pointer solve in two halves, so that one host thread can keep several devices busy:
// enqueue (gather into the pinned block, copies / launch on the handle's own stream) and finish
// (wait for the event, scatter the results into the caller's arrays)
struct HostJob {
  size_t B = 0, h = 0;
  size_t o_grf = 0, o_soln = 0, o_st = 0, o_it = 0;
  qmpc_outputs out{};
  bool active = false;
};

int host_enqueue(qmpc_ctx* c, int batch, const qmpc_inputs* in, const qmpc_outputs* out, HostJob& job) {
  if (!c || !in || !out) return QMPC_ERR_ARG;
  if (!c->is_setup) return QMPC_ERR_STATE;
  if (batch <= 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (!in->p || !in->v || !in->q || !in->w || !in->r || !in->yaw || !in->traj || !in->gait ||
      !in->weights || !in->alpha || !in->x_drag || !out->grf || !out->status)
    return QMPC_ERR_ARG;
  DeviceGuard g(c->device);
  const size_t B = (size_t)batch, h = (size_t)c->horizon;
  // ONE pinned host block holds the whole record and the results:  [ inputs | outputs ].
  //   * small batches (the single-robot shim): the kernel reads and writes the pinned block in
  //     place over PCIe (hipHostMalloc memory is device-visible and coherent) -- no copy
  //     commands at all, one launch and one event wait;
  //   * large batches: ONE H2D copy of the input part into a device mirror, ONE D2H copy of the
  //     output part back (PCIe bandwidth, not latency, matters there).
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 63) & ~(size_t)63;
    return o;
  };
  const size_t wN = in->weights_stride ? 12 * B : 12, aN = in->alpha_stride ? B : 1,
               xN = in->x_drag_stride ? B : 1;
  const size_t o_p = carve(4 * 3 * B), o_v = carve(4 * 3 * B), o_q = carve(4 * 4 * B),
               o_w = carve(4 * 3 * B), o_r = carve(4 * 12 * B), o_yaw = carve(4 * B),
               o_traj = carve(4 * 12 * h * B), o_gait = carve(4 * h * B), o_wt = carve(4 * wN),
               o_al = carve(4 * aN), o_xd = carve(4 * xN);
  const size_t in_bytes = (off + 255) & ~(size_t)255;
  off = in_bytes;
  const size_t o_grf = carve(4 * 12 * B), o_soln = carve(out->soln ? 8 * 12 * h * B : 0), o_st = carve(4 * B),
               o_it = carve(out->iters ? 4 * B : 0);
  const size_t total = off, out_bytes = total - in_bytes;
  if (total > c->pin_bytes) {
    if (c->h_pin) hipHostFree(c->h_pin);
    c->h_pin = nullptr;
    c->pin_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_pin, total, hipHostMallocDefault));
    c->pin_bytes = total;
  }
  if (!c->host_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->host_stream, hipStreamNonBlocking));
  if (!c->host_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->host_ev, hipEventDisableTiming));
  const bool in_place = batch <= 64;
  if (!in_place && total > c->stage_bytes) {
    if (c->d_stage) hipFree(c->d_stage);
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->d_stage, total));
    c->stage_bytes = total;
  }
  char* hb = (char*)c->h_pin;
  std::memcpy(hb + o_p, in->p, 4 * 3 * B);
  std::memcpy(hb + o_v, in->v, 4 * 3 * B);
  std::memcpy(hb + o_q, in->q, 4 * 4 * B);
  std::memcpy(hb + o_w, in->w, 4 * 3 * B);
  std::memcpy(hb + o_r, in->r, 4 * 12 * B);
  std::memcpy(hb + o_yaw, in->yaw, 4 * B);
  std::memcpy(hb + o_traj, in->traj, 4 * 12 * h * B);
  std::memcpy(hb + o_gait, in->gait, 4 * h * B);
  std::memcpy(hb + o_wt, in->weights, 4 * wN);
  std::memcpy(hb + o_al, in->alpha, 4 * aN);
  std::memcpy(hb + o_xd, in->x_drag, 4 * xN);
  hipStream_t s = c->host_stream;
  char* base = in_place ? hb : (char*)c->d_stage;
  if (!in_place) {
    if (const int rc = order_after_previous(c, s)) return rc;  // the mirror may still be read by an earlier call
    HIP_TRY(c, hipMemcpyAsync(base, hb, in_bytes, hipMemcpyHostToDevice, s));
  }
  qmpc_inputs din = *in;
  din.p = (const float*)(base + o_p); din.v = (const float*)(base + o_v);
  din.q = (const float*)(base + o_q); din.w = (const float*)(base + o_w);
  din.r = (const float*)(base + o_r); din.yaw = (const float*)(base + o_yaw);
  din.traj = (const float*)(base + o_traj); din.gait = (const uint8_t*)(base + o_gait);
  din.weights = (const float*)(base + o_wt); din.alpha = (const float*)(base + o_al);
  din.x_drag = (const float*)(base + o_xd);
  qmpc_outputs dout;
  dout.grf = (float*)(base + o_grf);
  dout.soln = out->soln ? (double*)(base + o_soln) : nullptr;
  dout.status = (int32_t*)(base + o_st);
  dout.iters = out->iters ? (int32_t*)(base + o_it) : nullptr;
  const int rc = qmpc_solve(c, batch, &din, &dout, s);
  if (rc != QMPC_OK) return rc;
  if (!in_place) HIP_TRY(c, hipMemcpyAsync(hb + in_bytes, base + in_bytes, out_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipEventRecord(c->host_ev, s));
  job.B = B; job.h = h;
  job.o_grf = o_grf; job.o_soln = o_soln; job.o_st = o_st; job.o_it = o_it;
  job.out = *out;
  job.active = true;
  return QMPC_OK;
}

int host_finish(qmpc_ctx* c, HostJob& job) {
  if (!job.active) return QMPC_OK;
  job.active = false;
  DeviceGuard g(c->device);
  // latency path: poll the event for a short while before handing the thread to the OS
  hipError_t q = hipErrorNotReady;
  for (int spin = 0; spin < 20000 && q == hipErrorNotReady; ++spin) q = hipEventQuery(c->host_ev);
  if (q == hipErrorNotReady) q = hipEventSynchronize(c->host_ev);
  if (q != hipSuccess) return fail(c, q, "qmpc_solve_host wait");
  const char* hb = (const char*)c->h_pin;
  const size_t B = job.B, h = job.h;
  std::memcpy(job.out.grf, hb + job.o_grf, 4 * 12 * B);
  if (job.out.soln) std::memcpy(job.out.soln, hb + job.o_soln, 8 * 12 * h * B);
  std::memcpy(job.out.status, hb + job.o_st, 4 * B);
  if (job.out.iters) std::memcpy(job.out.iters, hb + job.o_it, 4 * B);
  return QMPC_OK;
}

}  // namespace

extern "C" {

int qmpc_set_leg_geometry(qmpc_handle c, double abad, double hip, double knee, double knee_y) {
  if (!c || !(abad >= 0) || !(hip > 0) || !(knee > 0)) return QMPC_ERR_ARG;
  c->leg_geom[0] = (float)abad;
  c->leg_geom[1] = (float)hip;
  c->leg_geom[2] = (float)knee;
  c->leg_geom[3] = (float)knee_y;
  return QMPC_OK;
}

int qmpc_leg_kinematics(qmpc_handle c, int batch, const float* q, const float* qd, float* J, float* p, float* v,
                        void* stream) {
  if (!c || !q || !J || !p || (v && !qd)) return QMPC_ERR_ARG;
  if (batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  DeviceGuard g(c->device);
  HIP_TRY(c, qmpc_launch_leg_kin(c->leg_geom, q, qd, J, p, v, batch, (hipStream_t)stream));
  return QMPC_OK;
}

int qmpc_leg_torques(qmpc_handle c, int batch, const qmpc_leg_command* cmd, float* tau, float* q_des, void* stream) {
  if (!c || !cmd || !tau) return QMPC_ERR_ARG;
  if (!cmd->kp_cart || !cmd->kd_cart || !cmd->p_des || !cmd->v_des || !cmd->q || !cmd->qd || !cmd->J || !cmd->p ||
      !cmd->v)
    return QMPC_ERR_ARG;
  if (batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  DeviceGuard g(c->device);
  HIP_TRY(c, qmpc_launch_leg_cmd(c->leg_geom, cmd, tau, q_des, batch, (hipStream_t)stream));
  return QMPC_OK;
}

int qmpc_swing_trajectory(qmpc_handle c, int n_feet, const float* p0, const float* pf, const float* height,
                          const float* phase, const float* swing_time, float* p, float* v, float* a, void* stream) {
  if (!c || !p0 || !pf || !height || !phase || !swing_time || !p || !v || !a) return QMPC_ERR_ARG;
  if (n_feet < 0 || n_feet > 4 * c->max_batch) return QMPC_ERR_ARG;
  if (n_feet == 0) return QMPC_OK;
  DeviceGuard g(c->device);
  HIP_TRY(c, qmpc_launch_swing(p0, pf, height, phase, swing_time, p, v, a, n_feet, (hipStream_t)stream));
  return QMPC_OK;
}

int qmpc_kf_init(qmpc_handle c, int batch, float* xhat, float* P, void* stream) {
  if (!c || !xhat || !P || batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  DeviceGuard g(c->device);
  HIP_TRY(c, qmpc_launch_kf_init(xhat, P, batch, (hipStream_t)stream));
  return QMPC_OK;
}

int qmpc_kf_step(qmpc_handle c, int batch, const qmpc_kf_state* st, void* stream) {
  if (!c || !st || batch < 0 || batch > c->max_batch) return QMPC_ERR_ARG;
  if (!st->xhat || !st->P || !st->r_body || !st->a_world || !st->omega_body || !st->contact_phase || !st->leg_p ||
      !st->leg_v || !st->position || !st->v_world)
    return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  DeviceGuard g(c->device);
  static const float hip[3] = {0.19f, 0.049f, 0.f};  // _abadLocation (MiniCheetah.h:25-26,105)
  HIP_TRY(c, qmpc_launch_kf(st, hip, batch, (hipStream_t)stream));
  return QMPC_OK;
}

int qmpc_solve_host(qmpc_handle c, int batch, const qmpc_inputs* in, const qmpc_outputs* out) {
  HostJob job;
  const int rc = host_enqueue(c, batch, in, out, job);
  if (rc != QMPC_OK) return rc;
  return host_finish(c, job);
}

int qmpc_solve_sharded(const qmpc_handle* handles, int n_handles, int batch, const qmpc_inputs* in,
                       const qmpc_outputs* out) {
  if (!handles || n_handles <= 0 || n_handles > 64 || !in || !out || batch < 0) return QMPC_ERR_ARG;
  for (int k = 0; k < n_handles; ++k)
    if (!handles[k]) return QMPC_ERR_ARG;
  if (batch == 0) return QMPC_OK;
  if (!in->p || !in->v || !in->q || !in->w || !in->r || !in->yaw || !in->traj || !in->gait || !in->weights ||
      !in->alpha || !in->x_drag || !out->grf || !out->status)
    return QMPC_ERR_ARG;
  const int h = handles[0]->horizon;
  for (int k = 1; k < n_handles; ++k)
    if (handles[k]->horizon != h) return QMPC_ERR_STATE;  // every device must be set up for the same problem
  const int per = (batch + n_handles - 1) / n_handles;  // contiguous shards
  HostJob jobs[64];
  int rc_all = QMPC_OK;
  // enqueue on every device first (each call returns as soon as its copies / launches are queued) ...
  for (int k = 0; k < n_handles; ++k) {
    const int lo = k * per, hi = (lo + per < batch) ? lo + per : batch;
    if (lo >= hi) break;
    const size_t o = (size_t)lo;
    qmpc_inputs si = *in;
    si.p = in->p + 3 * o; si.v = in->v + 3 * o; si.q = in->q + 4 * o; si.w = in->w + 3 * o;
    si.r = in->r + 12 * o; si.yaw = in->yaw + o; si.traj = in->traj + (size_t)12 * h * o;
    si.gait = in->gait + (size_t)4 * h * o;
    si.weights = in->weights + (size_t)in->weights_stride * o;
    si.alpha = in->alpha + (size_t)in->alpha_stride * o;
    si.x_drag = in->x_drag + (size_t)in->x_drag_stride * o;
    qmpc_outputs so;
    so.grf = out->grf + 12 * o;
    so.soln = out->soln ? out->soln + (size_t)12 * h * o : nullptr;
    so.status = out->status + o;
    so.iters = out->iters ? out->iters + o : nullptr;
    const int rc = host_enqueue(handles[k], hi - lo, &si, &so, jobs[k]);
    if (rc != QMPC_OK && rc_all == QMPC_OK) rc_all = rc;
  }
  // ... then collect: the devices run concurrently, one host thread drives them all
  for (int k = 0; k < n_handles; ++k) {
    const int rc = host_finish(handles[k], jobs[k]);
    if (rc != QMPC_OK && rc_all == QMPC_OK) rc_all = rc;
  }
  return rc_all;
}

}  // extern "C"
