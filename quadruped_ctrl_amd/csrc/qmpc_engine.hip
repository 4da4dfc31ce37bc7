// qmpc_engine.hip -- the CONSUMER half of the decoupled path: the dual active set (Goldfarb-Idnani, event form)
// on an explicit inverse that lives in global memory, for the size classes whose robots iterate long
// (128- and 192-row classes, horizons up to 36 segments in the latter).
//
// What it replaces in the reference: the qpOASES call of solve_mpc (src/MPC_Ctrl/SolverMPC.cpp:527-557: cold
// QProblem::init on H_red, g_red, A_red, then q_soln scatter) -- same unique minimiser (H is positive definite).
//
// Why a second engine.  In the monolithic kernel (qmpc_kernels.hip) a robot of the large classes holds
// a whole CU -- the packed inverse fills its LDS -- while ONE of its 8 / 12 waves iterates (73 % of the wave-cycles
// parked, PMC).  Here the producer (qmpc_sweep_kernel) writes H^-1 to an L2 / Infinity-Cache resident work item and
// leaves; this kernel runs one robot per small workgroup (two per CU in the 128-row class):
//   * wave 0 is the ENGINE: x, multipliers, working set in registers (lane = variable / stance slot / working slot),
//     the two columns of H^-1 an iteration needs are two or three coalesced row loads (the matrix is symmetric);
//   * waves 1..NH are event HOLDERS: the rank-1 events (z~, g~) that represent the projected inverse
//         P = H^-1 - sum_add z~ z~^T + sum_drop z~ z~^T,   N* = sum z~ g~^T,   S^-1 = sum_add g~ g~^T - sum_drop g~ g~^T
//     never leave the REGISTER FILE: a new event goes to whoever holds the fewest -- a holder (MAXL statically indexed
//     register slots of RE + KQ doubles per lane) or the engine wave itself (MAXE records in LDS: it is idle while the
//     holders work).  An iteration's accumulation z -= +-y z~, r += y g~ (y = z~^T c_p) runs in all waves at once on
//     their own events, the holders' operands by readlane -- no event pool in global memory, no spill / compaction
//     machinery.  Two LDS-only workgroup barriers per iteration carry the request and the partial sums (fixed split,
//     fixed order of the final sum: results do not depend on timing).
// A robot that needs more events than registers + LDS hold (NH * MAXL + MAXE), or whose projected inverse loses
// definiteness numerically, is handed back to the monolithic kernel of its class through a list (QMPC_ST_FALLBACK).
// block_start (experimental, off by default): forced additions of candidate sets by all threads before the iteration.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "qmpc_device.h"
#include "qmpc_cmd.h"
#include "qmpc_wave.h"

namespace {

// RE: 64-row blocks of a variable-indexed vector (LD = 64 RE padded rows); KQ: working-set slots per lane (64 KQ
// slots); SQ: stance slots per lane (64 SQ foot-steps in stance at most); NW: waves per workgroup (1 engine + NH
// holders); MAXL: events per holder.
// MAXE: events the ENGINE wave itself holds, in LDS (it is idle while the holders accumulate: it takes a share of the
// events like they do, and everything beyond the holders' registers).
template <int RE_, int KQ_, int SQ_, int NW_, int MAXL_, int MAXE_>
struct ECfg {
  static constexpr int RE = RE_, KQ = KQ_, SQ = SQ_, NW = NW_, MAXL = MAXL_, MAXE = MAXE_;
  static constexpr int NH = NW - 1, NP = 64 * RE, KS = 64 * KQ, NSL = 64 * SQ, NT = 64 * NW;
  static constexpr int NRR = (NH + 1) * MAXL;     // events dealt round-robin over engine + holders
  static constexpr int KEV = NH * MAXL + MAXE;    // capacity of registers + LDS
  // ... and beyond that the engine wave keeps events in this workgroup's slice of an overflow pool in global memory
  // (L2): slower per event, but a robot with an unusually long active-set history continues instead of being handed back
  static constexpr int MAXG = QMPC_ENGINE_OVF_EVENTS;
  static constexpr int BIAS = (RE == 2) ? 5 : 7;  // events the engine wave is dealt fewer than a holder (measured)
  static constexpr int EV = NP + KS;
  static_assert(MAXE >= MAXL && MAXE <= 64, "engine-held events");
  // block start: records it may leave -- what the holders' registers take afterwards (its LDS is the engine wave's pool)
  static constexpr int MAXB = (NH * MAXL < MAXE ? NH * MAXL : MAXE) & ~1;
  static_assert(NSL <= QMPC_WK_SLOTS_MAX, "stance slots of a work item");
  static_assert(5 * NSL <= 1024, "constraint id in ten bits of the selection key");
};

enum { CMD_DONE = 0, CMD_ACC = 1, CMD_DROPACC = 2 };

// profiling hook (qmpc_set_debug_clock): shader-clock stamps of ONE iteration of the engine wave (slots 0..7) and of
// holder 1 (slots 8..11); tools/engine_phase.py
#ifndef QMPC_EDBG_ITER
#define QMPC_EDBG_ITER 20
#endif
#define QMPC_ESTAMP(k)                                                                   \
  do {                                                                                   \
    if (dbg_clk && lane == 0 && iters == QMPC_EDBG_ITER) dbg_clk[(k)] = clock64();      \
  } while (0)

template <class C>
struct ESmem {
  QmpcParams par;
  int rid, n, nst, status0, qnext;
  float fmaxk[C::NSL];
  unsigned char sidx[C::NSL];
  // the engine's request to the holders, and the event it staged for one of them in the previous round
  // (three 16-byte words: written with three stores, read by every holder with three broadcast loads in flight together)
  struct alignas(16) Req {
    int cmd, pj1, pj2, l;
    double pa1, pa2;
    int ing_valid, ing_owner, ing_li, ing_flags;  // ing_flags: bit 0 = drop event, bits 8.. = 1 + slot to clear
    int nglv, pad0, pad1, pad2;                   // overflow events (global memory) this round's accumulation covers
  } rq;
  alignas(16) double stage[C::EV];       // the new event (z~[NP], g~[KS])
  alignas(16) double epool[C::MAXE][C::EV];  // the engine wave's own events
  double part[C::NH][C::EV];             // the holders' partial sums (z[NP], r[KS])
  double xl[C::NP];                      // x, variable-indexed, for the stance-slot lanes of the selection
  double D[C::NP];                       // diag(H^-1)
  float fb[12];
  signed char gsign[C::MAXG];            // overflow events in global memory: +1 add, -1 drop
  // block start (see block_start): working set and multipliers by slot, the signs of the records in the event pool
  struct Blk {
    int ne, nadd, nc, fail, dl, pad0, pad1, pad2;
    int cand[C::NSL];          // this round's candidates (constraint ids)
    int wcid[C::KS];           // working-set slot -> constraint id (-1 = free)
    int sign[C::MAXE];         // record e: +1 add event, -1 drop event
    unsigned bmask[C::NSL];    // stance slot -> rows in the working set
    double lam[C::KS];         // multipliers by slot
    alignas(16) double Y[C::MAXE];
    alignas(16) double Ys[C::MAXE];
  } bk;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence over ALL address
// spaces, i.e. it waits for the wave's outstanding GLOBAL loads too (one counter for loads and stores on gfx9) -- the
// engine wave's H^-1 column loads would be waited for at (A) with every other wave behind them.  The data the waves of
// this kernel exchange lives in LDS; global memory is only read (work item) or written at the very end (results).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// f(integral_constant<LI>) for LI = 0 .. count-1 (count <= N, wave-uniform) as NESTED ifs: straight-line code with one
// not-taken branch per event and a single exit -- a flat chain of `if (LI < count)` was laid out by the compiler as
// blocks scattered over the kernel, a taken branch (instruction refetch) per event
template <int LI, int N>
struct Upto {
  template <class F>
  static __device__ __forceinline__ void run(int count, F&& f) {
    if (LI < count) {
      f(std::integral_constant<int, LI>{});
      Upto<LI + 1, N>::run(count, f);
    }
  }
};
template <int N>
struct Upto<N, N> {
  template <class F>
  static __device__ __forceinline__ void run(int, F&&) {}
};

// f(q1c, q2c) with the 64-row blocks of the two variables of a constraint row as compile-time constants: j2 - j1 <= 2
// (same stance slot), so q2 is q1 or q1 + 1
template <int RE, class F>
__device__ __forceinline__ void dispatch_blocks(int q1, int q2, F&& f) {
  StaticFor<0, RE>::run([&](auto qa) __attribute__((always_inline)) {
    constexpr int QA = decltype(qa)::value;
    if (q1 == QA) {
      if (q2 == QA) f(std::integral_constant<int, QA>{}, std::integral_constant<int, QA>{});
      else if constexpr (QA + 1 < RE) f(std::integral_constant<int, QA>{}, std::integral_constant<int, QA + 1>{});
    }
  });
}

// f(integral_constant<V>) for the run-time value v in [LO, HI): a binary tree of wave-uniform branches (log2 levels)
template <int LO, int HI>
struct Pick {
  template <class F>
  static __device__ __forceinline__ void run(int v, F&& f) {
    if constexpr (HI - LO == 1) {
      f(std::integral_constant<int, LO>{});
    } else {
      constexpr int MID = (LO + HI) / 2;
      if (v < MID) Pick<LO, MID>::run(v, f);
      else Pick<MID, HI>::run(v, f);
    }
  }
};

// ---------------------------------------------------------------------------------------------------- block start
// The dual active set adds ONE constraint per iteration, and a robot braking to a stand ends with 30+ rows at a bound:
// a serial chain of 30+ iterations of ~5 k cycles, each with a selection, a ratio test and two barrier rounds with the
// holders.  Most of those rows are known early: the rows violated at the unconstrained minimiser x_u are, almost
// without exception, active at the solution (precision 0.96 - 1.00).  So the iteration starts from a BLOCK
// of forced changes made by ALL threads of the workgroup:
//   round r = 1 .. QMPC_BLK_ROUNDS: the most violated row of every stance foot-step at the current x (not in W yet),
//   each added as in a full Goldfarb-Idnani step but WITHOUT search and ratio test:
//       y_e = z~_e^T c,  z = H^-1 c - sum y z~,  r = sum y g~,  delta = c^T z,  t = -(c^T x - rhs) / delta,
//       x += t z,  lambda_W -= t r,  lambda_p = t,  new record (z, -r, 1) / sqrt(delta);
//   x is then the minimiser on W as equalities; afterwards the rows whose multiplier came out negative leave, most
//   negative first (drop records with the repair step: x -= (lambda_l / gamma) N*_l, lambda -= (lambda_l / gamma) S^-1[:, l]).
// Data layout of the phase: thread t owns ENTRY t of every record (t < NP: variable t of z~; t >= NP: slot t - NP of g~)
// in REGISTERS (`col`, statically indexed), and a TRANSPOSED copy of the records sits in LDS (`Rt[t][e]`: entry t of all
// records contiguous) -- so y for all records is two contiguous broadcast vectors (16-byte loads), and the accumulation
// over the records is one fma per record on the thread's own registers: no pass over record-major data in LDS (measured
// before: ~95 cycles per record and forced change, latency-bound).  Two LDS-only barriers per forced change.
// What comes out is a genuine Goldfarb-Idnani state (x optimal on W, multipliers >= 0, the projected inverse as rank-1
// records): the holders take the records into their registers and the normal iteration finishes the job -- same unique
// minimiser (tools/block_proto.py: braking at horizon 10, 34 iterations -> 4 rounds, ~5 removals, ~4 iterations).
#ifndef QMPC_BLK_ROUNDS
#define QMPC_BLK_ROUNDS 4
#endif
template <class C>
__device__ __forceinline__ void block_start(const int tid, ESmem<C>& S, const GlobalF64* const Hi, const GlobalF64* const xu,
                                            const int n, const int nst, const int max_changes) {
  constexpr int SQ = C::SQ, NP = C::NP, LD = C::NP, NT = C::NT, EV = C::EV, KS = C::KS, NH = C::NH;
  constexpr int MAXB = C::MAXB;
  constexpr int NCMAX = 12;  // candidates added per round
  static_assert(EV <= NT, "one thread per record entry");
  const QmpcParams& P = S.par;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto& B = S.bk;
  double(*const Rt)[MAXB] = reinterpret_cast<double(*)[MAXB]>(&S.epool[0][0]);  // Rt[entry][record]
  static_assert(sizeof(double) * EV * MAXB <= sizeof(S.epool), "transposed records fit the engine wave's pool");
  double* const T = S.stage;  // the vector being built (z | r)
  const double mi = P.mu_inv;
  const bool isvar = tid < NP, isslot = tid >= NP && tid < EV, isent = tid < EV;
  const int w = tid - NP;  // slot of a slot thread
  if (tid == 0) {
    B.ne = 0;
    B.nadd = 0;
    B.fail = 0;
  }
  for (int k = tid; k < KS; k += NT) {
    B.wcid[k] = -1;
    B.lam[k] = 0.0;
  }
  if (isent) {
#pragma unroll
    for (int e = 0; e < MAXB; e += 2) st2(&Rt[tid][e], 0.0, 0.0);
  }
  double col[MAXB];  // this thread's entry of every record
#pragma unroll
  for (int e = 0; e < MAXB; ++e) col[e] = 0.0;
  __syncthreads();
  // the new record's entry goes into register [ne] (a run-time index: one case per register, kept apart by a dummy
  // instruction -- see the holders' ingest) and into the transposed copy
  auto put_entry = [&](int ne, double v) __attribute__((always_inline)) {
    Pick<0, MAXB>::run(ne, [&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      col[e] = v;
      asm volatile("; record register %0" ::"n"(e));
    });
    Rt[tid][ne] = v;
  };
  bool stop = false;
  for (int round = 0; round < QMPC_BLK_ROUNDS && !stop; ++round) {
    if (wv == 0) {
      // ---- this round's candidates: the most violated row of every stance foot-step at the current x, not in W
      const double inv_fr = P.inv_fr_norm, tol = P.tol;
#pragma unroll
      for (int s = 0; s < SQ; ++s) B.bmask[lane + 64 * s] = 0u;
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < KS; k += 64) {
        const int id = B.wcid[k];
        if (id >= 0) atomicOr(&B.bmask[id / 5], 1u << (id % 5));
      }
      __builtin_amdgcn_wave_barrier();
      int nc = 0;
#pragma unroll
      for (int s = 0; s < SQ; ++s) {
        const int sl = lane + 64 * s;
        const int sc = sl < nst ? sl : 0;
        const double x0 = S.xl[3 * sc], x1 = S.xl[3 * sc + 1], x2 = S.xl[3 * sc + 2];
        const unsigned am = B.bmask[sc];
        const double fmx = (double)S.fmaxk[sc];
        int tmin = -1;
        if (sl < nst) {
          const double fx = mi * x0, fy = mi * x1;
          double vmin = 0.0;
          const double sv[5] = {(fx + x2) * inv_fr, (x2 - fx) * inv_fr, (fy + x2) * inv_fr, (x2 - fy) * inv_fr, fmx - x2};
#pragma unroll
          for (int ty = 0; ty < 5; ++ty) {
            const bool cand = !((am >> ty) & 1u) && sv[ty] < vmin;
            vmin = cand ? sv[ty] : vmin;
            tmin = cand ? ty : tmin;
          }
          if (!(vmin < -tol)) tmin = -1;
        }
        const unsigned long long vm = __ballot(tmin >= 0);
        if (tmin >= 0) B.cand[nc + __popcll(vm & ((1ull << lane) - 1ull))] = 5 * sl + tmin;
        nc += __popcll(vm);
      }
      if (lane == 0) B.nc = nc;
    }
    __syncthreads();
    int nc = B.nc;
    if (nc == 0) break;  // uniform: nothing violated (optimal) -- or nothing new to add
    nc = nc < NCMAX ? nc : NCMAX;  // (the rest of a longer list is picked up by the next round)
    // ---- H^-1 c for EVERY candidate of the round first: all loads in flight together, one L2 / HBM round trip per round
    // (the inverse was written by another kernel: every column is a first touch, ~2 k cycles each if waited for singly)
    double zc[NCMAX];
    Upto<0, NCMAX>::run(nc, [&](auto cic) __attribute__((always_inline)) {
      constexpr int ci = decltype(cic)::value;
      const int id = B.cand[ci];
      const int slot = id / 5, ty = id - 5 * slot, j0 = 3 * slot;
      const int j1 = (ty < 4) ? j0 + (ty >> 1) : j0 + 2, j2 = j0 + 2;
      const double a1 = (ty < 4) ? ((ty & 1) ? -mi : mi) : -1.0, a2 = (ty < 4) ? 1.0 : 0.0;
      // (lower block triangle: above the column's diagonal block by symmetry from the row, below it from the column)
      const bool up1 = (tid >> 6) <= (j1 >> 6), up2 = (tid >> 6) <= (j2 >> 6);
      zc[ci] = (isvar && tid < n) ? __builtin_fma(a2, up2 ? Hi[(size_t)j2 * LD + tid] : Hi[(size_t)tid * LD + j2],
                                                  a1 * (up1 ? Hi[(size_t)j1 * LD + tid] : Hi[(size_t)tid * LD + j1]))
                                  : 0.0;
    });
#pragma unroll 1
    for (int ci = 0; ci < nc && !stop; ++ci) {
      const int ne = B.ne, q = B.nadd;
      if (ne >= MAXB || q >= KS || ne >= max_changes) {  // uniform: no room / iteration limit -- the normal iteration goes on
        stop = true;
        break;
      }
#ifdef QMPC_BLK_STAMP
      long long* const bclk = (P.dbg_clk && tid == 0 && ne == QMPC_BLK_STAMP) ? P.dbg_clk + (size_t)S.rid * 16 : nullptr;
      if (bclk) bclk[0] = clock64();
#endif
      // ---- forced addition of row id into slot q
      const int id = B.cand[ci];
      const int slot = id / 5, ty = id - 5 * slot, j0 = 3 * slot;
      const int j1 = (ty < 4) ? j0 + (ty >> 1) : j0 + 2, j2 = j0 + 2;
      const double a1 = (ty < 4) ? ((ty & 1) ? -mi : mi) : -1.0, a2 = (ty < 4) ? 1.0 : 0.0;
      const double rhs = (ty == 4) ? -(double)S.fmaxk[slot] : 0.0;
      const double sp = __builtin_fma(a2, S.xl[j2], a1 * S.xl[j1]) - rhs;  // (x is stable until step 2)
      // (every record is an ADD record while the rounds last -- removals come after them)
      double zc_ci = 0.0;
      StaticFor<0, NCMAX>::run([&](auto cc) __attribute__((always_inline)) {
        if (decltype(cc)::value == ci) zc_ci = zc[decltype(cc)::value];
      });
      if (isent) {
        const double sg = isvar ? -1.0 : 1.0;
        double acc = isvar ? zc_ci : 0.0, acc2 = 0.0;
        const double* const r1 = Rt[j1];
        const double* const r2 = Rt[j2];
        // eight records per group: their 16-byte broadcast loads in flight together, ONE wait (behind a wave-uniform
        // branch per record the loads cannot be hoisted and every step waits its own LDS round trip: 1.8 k cycles at 20 records)
        Upto<0, (MAXB + 7) / 8>::run((ne + 7) >> 3, [&](auto gc) __attribute__((always_inline)) {
          constexpr int e0 = 8 * decltype(gc)::value;
          constexpr int NPR = (MAXB - e0 >= 8) ? 4 : (MAXB - e0) / 2;
          F64x2 p[NPR], qq[NPR];
#pragma unroll
          for (int u = 0; u < NPR; ++u) {
            p[u] = ld2(r1 + e0 + 2 * u);  // (records past the last one are zero)
            qq[u] = ld2(r2 + e0 + 2 * u);
          }
#pragma unroll
          for (int u = 0; u < NPR; ++u) {
            acc = __builtin_fma(sg * __builtin_fma(a2, qq[u].x, a1 * p[u].x), col[e0 + 2 * u], acc);
            acc2 = __builtin_fma(sg * __builtin_fma(a2, qq[u].y, a1 * p[u].y), col[e0 + 2 * u + 1], acc2);
          }
        });
        T[tid] = acc + acc2;
      }
#ifdef QMPC_BLK_STAMP
      if (bclk) bclk[3] = clock64();
#endif
      lds_barrier();
#ifdef QMPC_BLK_STAMP
      if (bclk) bclk[4] = clock64();
#endif
      const double delta = __builtin_fma(a2, T[j2], a1 * T[j1]);
      const double cn = __builtin_fma(a2 * a2, S.D[j2], a1 * a1 * S.D[j1]);
      if (delta > 1e-11 * cn) {  // uniform (else: the row depends on the ones in W, e.g. the pyramid's apex: skipped)
        const double tt = -sp * fast_rcp(delta);
        double sq = __builtin_amdgcn_rsq(delta);
        {
          double e = __builtin_fma(-delta * sq, sq, 1.0);
          sq = __builtin_fma(0.5 * sq, e, sq);
          e = __builtin_fma(-delta * sq, sq, 1.0);
          sq = __builtin_fma(0.5 * sq, e, sq);
        }
        if (isvar) {
          const double z = T[tid];
          put_entry(ne, z * sq);
          if (tid < n) S.xl[tid] = __builtin_fma(tt, z, S.xl[tid]);
        } else if (isslot) {
          const double r = T[tid];
          const bool active = B.wcid[w] >= 0;
          put_entry(ne, (w == q) ? sq : (active ? -r * sq : 0.0));
          if (active) B.lam[w] = __builtin_fma(-tt, r, B.lam[w]);
          if (w == q) {
            B.lam[w] = tt;
            B.wcid[w] = id;
          }
        }
        if (tid == 0) {
          B.sign[ne] = 1;
          B.ne = ne + 1;
          B.nadd = q + 1;
        }
      }
#ifdef QMPC_BLK_STAMP
      if (bclk) bclk[5] = clock64();
#endif
      lds_barrier();
#ifdef QMPC_BLK_STAMP
      if (bclk) bclk[7] = clock64();
#endif
    }
  }
  // ---- rows whose multiplier came out negative leave, most negative first
  while (true) {
    if (wv == 0) {
      double worst = 0.0;
      for (int k = lane; k < KS; k += 64) {
        const double v = (B.wcid[k] >= 0 && B.lam[k] < 0.0) ? -B.lam[k] : 0.0;
        worst = v > worst ? v : worst;
      }
      const double wmax = wave_max_pos_f64(worst);
      int l = -1;
      if (wmax > 0.0) {
        for (int k0 = 0; k0 < KS && l < 0; k0 += 64) {
          const int k = k0 + lane;
          const unsigned long long hit = __ballot(B.wcid[k] >= 0 && B.lam[k] < 0.0 && -B.lam[k] == wmax);
          if (hit != 0ull) l = k0 + __ffsll((long long)hit) - 1;
        }
      }
      if (lane == 0) B.dl = l;
    }
    lds_barrier();
    const int l = B.dl, ne = B.ne;
    if (l < 0 || ne >= MAXB || ne >= max_changes) break;  // uniform (rows still negative are dropped by the engine wave's own loop)
    const double laml = B.lam[l];
    if (isent) {
      // u = N*_l = sum g~_e[l] z~_e (variable entries), sc = S^-1[:, l] = sum +-g~_e[l] g~_e (slot entries; - for drop records)
      double acc = 0.0, acc2 = 0.0;
      const double* const rl = Rt[NP + l];
      Upto<0, (MAXB + 7) / 8>::run((ne + 7) >> 3, [&](auto gc) __attribute__((always_inline)) {
        constexpr int e0 = 8 * decltype(gc)::value;
        constexpr int NPR = (MAXB - e0 >= 8) ? 4 : (MAXB - e0) / 2;
        F64x2 y[NPR];
        int sgn[2 * NPR];
#pragma unroll
        for (int u = 0; u < NPR; ++u) {
          y[u] = ld2(rl + e0 + 2 * u);
          sgn[2 * u] = B.sign[e0 + 2 * u];
          sgn[2 * u + 1] = B.sign[e0 + 2 * u + 1];
        }
#pragma unroll
        for (int u = 0; u < NPR; ++u) {
          const double s0 = (isslot && sgn[2 * u] < 0) ? -y[u].x : y[u].x, s1 = (isslot && sgn[2 * u + 1] < 0) ? -y[u].y : y[u].y;
          acc = __builtin_fma(s0, col[e0 + 2 * u], acc);
          acc2 = __builtin_fma(s1, col[e0 + 2 * u + 1], acc2);
        }
      });
      T[tid] = acc + acc2;
    }
    lds_barrier();
    const double gamma = T[NP + l];
    if (!(gamma > 0.0)) {  // uniform: numerically lost S^-1[l][l] > 0 -- the robot is handed back
      if (tid == 0) B.fail = 1;
      break;
    }
    const double coef = laml * fast_rcp(gamma);
    double sg = __builtin_amdgcn_rsq(gamma);
    {
      double e = __builtin_fma(-gamma * sg, sg, 1.0);
      sg = __builtin_fma(0.5 * sg, e, sg);
      e = __builtin_fma(-gamma * sg, sg, 1.0);
      sg = __builtin_fma(0.5 * sg, e, sg);
    }
    if (isvar) {
      const double u = T[tid];
      put_entry(ne, u * sg);
      if (tid < n) S.xl[tid] = __builtin_fma(-coef, u, S.xl[tid]);
    } else if (isslot) {
      const double sc = T[tid];
      const bool active = B.wcid[w] >= 0 && w != l;
      if (w == l) {
        // slot l leaves: its column of every earlier g~ is cleared -- this thread's registers and its row of the copy
#pragma unroll
        for (int e = 0; e < MAXB; ++e) col[e] = 0.0;
#pragma unroll
        for (int e = 0; e < MAXB; e += 2) st2(&Rt[tid][e], 0.0, 0.0);
        B.wcid[w] = -1;
        B.lam[w] = 0.0;
      } else {
        put_entry(ne, active ? -sc * sg : 0.0);
        if (active) B.lam[w] = __builtin_fma(-coef, sc, B.lam[w]);
      }
    }
    if (tid == 0) {
      B.sign[ne] = -1;
      B.ne = ne + 1;
    }
    lds_barrier();
  }
  __syncthreads();
}

// +-1.0 in a scalar register pair, opaque to the optimiser (which would turn y * (c ? 1 : -1) back into a negation and a
// select).  acc: the sign of an event's term in z (-1 add, +1 drop); otherwise its sign in S^-1 (+1 add, -1 drop)
__device__ __forceinline__ double event_sign(unsigned long long isdrop, bool acc) {
  unsigned hi = ((isdrop != 0ull) == acc) ? 0x3FF00000u : 0xBFF00000u;
  asm volatile("" : "+s"(hi));
  return __longlong_as_double((long long)((unsigned long long)hi << 32));
}

// A wave's share of an accumulation over the OVERFLOW events (the workgroup's slice in global memory): events first,
// first + stride, ... below count, four per trip.  The records were written by the engine wave in earlier rounds (its stores
// drained before the barrier that published `count`); they are read past the L1 (agent-scope loads: the slice is reused from
// robot to robot, and an L1 line of this CU may hold an older robot's record).  Same sums and signs as own_events
template <class C, bool ACC>
__device__ __forceinline__ void ovf_accumulate(const GlobalF64* const gov, const signed char* const gsign, const int first, const int stride,
                                               const int count, const int j1, const int j2, const int l, const double a1,
                                               const double a2, const int lane, double (&zs)[C::RE], double (&rs)[C::KQ]) {
  constexpr int RE = C::RE, KQ = C::KQ, NP = C::NP, EV = C::EV;
  auto ld = [](const GlobalF64* p) __attribute__((always_inline)) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
#pragma unroll 1
  for (int g0 = first; g0 < count; g0 += 4 * stride) {
    double ya[4], yb[4], zl[4][RE], gl[4][KQ];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = g0 + u * stride;
      const GlobalF64* ev = gov + (size_t)(g < count ? g : first) * EV;
      ya[u] = ACC ? ld(ev + j1) : ld(ev + NP + l);
      yb[u] = ACC ? ld(ev + j2) : 0.0;
#pragma unroll
      for (int q = 0; q < RE; ++q) zl[u][q] = ld(ev + lane + 64 * q);
#pragma unroll
      for (int k = 0; k < KQ; ++k) gl[u][k] = ld(ev + NP + lane + 64 * k);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = g0 + u * stride;
      const bool isdrop = gsign[g < count ? g : first] < 0;
      double y = ACC ? __builtin_fma(a2, yb[u], a1 * ya[u]) : ya[u];
      if (!(g < count)) y = 0.0;
      const double yz = ACC ? (isdrop ? y : -y) : y;
      const double yr = ACC ? y : (isdrop ? -y : y);
#pragma unroll
      for (int q = 0; q < RE; ++q) zs[q] = __builtin_fma(yz, zl[u][q], zs[q]);
#pragma unroll
      for (int k = 0; k < KQ; ++k) rs[k] = __builtin_fma(yr, gl[u][k], rs[k]);
    }
  }
}

template <class C, bool WARM>
__device__ __forceinline__ void engine_item(const int item, const int tid, ESmem<C>& S, const QmpcParams& PK) {
  constexpr int RE = C::RE, KQ = C::KQ, SQ = C::SQ, NH = C::NH, NP = C::NP, KS = C::KS, MAXL = C::MAXL, LD = C::NP;
  constexpr int MAXE = C::MAXE, EV = C::EV;
  using Req = typename ESmem<C>::Req;
  const QmpcParams& P = S.par;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (work item of the robot: `item` counts within this launch's chunk)
  const GlobalF64* const Hi = (const GlobalF64*)PK.wk_hinv + (size_t)item * (LD * LD);
  const GlobalF64* const xu = (const GlobalF64*)PK.wk_xu + (size_t)item * LD;
  const QmpcWorkHdr* const hd = PK.wk_hdr + item;
  // ---- the item's header and the diagonal of H^-1 -> LDS; the parameter block is parked once per workgroup (kernel)
  if (tid == 0) {
    S.rid = hd->rid;
    S.n = hd->n;
    S.nst = hd->nst;
    S.status0 = hd->status0;
    S.rq.ing_valid = 0;
    S.rq.cmd = CMD_DONE;
  }
  for (int k = tid; k < C::NSL; k += C::NT) {
    S.fmaxk[k] = hd->fmaxk[k < QMPC_WK_SLOTS_MAX ? k : 0];
    S.sidx[k] = hd->sidx[k < QMPC_WK_SLOTS_MAX ? k : 0];
  }
  {
    const int n0 = hd->n;  // (rows and columns past n are not part of the item)
    for (int k = tid; k < NP; k += C::NT) {
      S.D[k] = (k < n0) ? Hi[(size_t)k * LD + k] : 1.0;
      S.xl[k] = xu[k];
    }
  }
  __syncthreads();
  const int n = S.n, nst = S.nst, rid = S.rid;
  const int h = P.horizon;
  long long* dbg_clk = P.dbg_clk ? P.dbg_clk + (size_t)rid * 16 : nullptr;
  if (dbg_clk && tid == 0) dbg_clk[12] = clock64();
  // (the experimental block start needs one thread per record entry: the 128-row class's engine has them)
  const bool blk = C::EV <= C::NT && P.wk_block != 0;
  if constexpr (C::EV <= C::NT) {
    if (blk) block_start<C>(tid, S, Hi, xu, n, nst, P.max_iter);
  }
  if (!blk && tid == 0) {
    S.bk.ne = 0;
    S.bk.fail = 0;
  }
  __syncthreads();
  const int bkev = S.bk.ne;  // records the block start left in the engine wave's pool (record e = event e)
  if (dbg_clk && tid == 0) dbg_clk[13] = clock64();
  // Events are dealt to whoever holds the fewest: the engine wave (owner 0, its LDS pool, MAXE records) or a holder
  // (owners 1..NH, MAXL register slots each); the engine wave keeps the counts.  The block start's records all go to the
  // holders (record e -> holder 1 + e mod NH, registers [e / NH]): its LDS is the engine wave's pool.
  if (wv == 0) {
    // =============================================================== the engine wave
    const double mi = P.mu_inv, inv_fr = P.inv_fr_norm, tol = P.tol;
    const int max_iter = __builtin_amdgcn_readfirstlane(P.max_iter);
    const int kev = __builtin_amdgcn_readfirstlane(P.wk_kev < C::KEV + C::MAXG ? P.wk_kev : C::KEV + C::MAXG);  // (test hook)
    auto uni = [](bool cnd) __attribute__((always_inline)) { return __builtin_amdgcn_ballot_w64(cnd) != 0ull; };
    double xv[RE];  // (x_u, or the block start's minimiser on its working set)
#pragma unroll
    for (int q = 0; q < RE; ++q) xv[q] = (lane + 64 * q < n) ? S.xl[lane + 64 * q] : 0.0;
    double fmx[SQ];
    unsigned amask[SQ];
#pragma unroll
    for (int s = 0; s < SQ; ++s) {
      fmx[s] = (lane + 64 * s < nst) ? (double)S.fmaxk[lane + 64 * s] : 0.0;
      amask[s] = 0u;
    }
    int wcid[KQ];
    double lam[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      wcid[k] = -1;
      lam[k] = 0.0;
    }
    int khw = 0, status = 0, nev = 0, iters = 0;
    int nle = 0;                        // events in the engine wave's own LDS pool
    int ngl = 0;                        // ... and in its slice of the overflow pool in global memory
    int nglv = 0;                       // (how many of them the current round's request announced)
    GlobalF64* const gov = P.wk_ovf ? (GlobalF64*)P.wk_ovf + (size_t)blockIdx.x * ((size_t)C::MAXG * EV) : nullptr;
    unsigned long long dropme = 0ull;   // ... that are drop events
    int cnt[NH + 1];                    // events held by owner o (0 = this wave's pool = nle)
#pragma unroll
    for (int o = 0; o <= NH; ++o) cnt[o] = 0;
    bool retry = false;
    if (bkev > 0) {
      // ---- state left by the block start: x (already in xl), the working set and its multipliers by slot, membership
      // masks, and the engine wave's share of the records (event e belongs to owner e mod (NH + 1)), compacted to the
      // front of its pool once the holders have taken theirs (barrier)
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        wcid[k] = S.bk.wcid[lane + 64 * k];
        lam[k] = S.bk.lam[lane + 64 * k];
        const unsigned long long used = __ballot(wcid[k] >= 0);
        if (used != 0ull) khw = 64 * k + 64 - __builtin_clzll(used);
      }
#pragma unroll
      for (int s = 0; s < SQ; ++s) S.bk.bmask[lane + 64 * s] = 0u;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < KQ; ++k)
        if (wcid[k] >= 0) atomicOr(&S.bk.bmask[wcid[k] / 5], 1u << (wcid[k] % 5));
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < SQ; ++s) amask[s] = S.bk.bmask[lane + 64 * s];
      nev = bkev;
      iters = bkev;  // (every forced addition / removal of the block start is a working-set change like an iteration's)
      if (S.bk.fail != 0) retry = true;
#pragma unroll
      for (int o = 1; o <= NH; ++o) cnt[o] = (bkev > o - 1) ? (bkev - (o - 1) + NH - 1) / NH : 0;
      lds_barrier();  // the holders have taken the records (the phase's LDS is this wave's event pool from here on)
    }
    int p_e = 0, psl = 0, pty = 0, pj1 = 0, pj2 = 0;
    double pa1 = 0.0, pa2 = 0.0, p_rhs = 0.0, lp = 0.0;
    auto rsqrt_full = [&](double d) __attribute__((always_inline)) {
      double y = __builtin_amdgcn_rsq(d);
      double e = __builtin_fma(-d * y, y, 1.0);
      y = __builtin_fma(0.5 * y, e, y);
      e = __builtin_fma(-d * y, y, 1.0);
      y = __builtin_fma(0.5 * y, e, y);
      return y;
    };
    // the engine wave's share of an accumulation, over its own events in LDS, four per trip:
    //   ACC:     y = z~^T c_p          zs += -+y z~ (add / drop event)   rs += y g~
    //   DROPACC: y = g~[l]             zs += y z~                        rs += +-y g~
    auto own_events = [&](auto accc, int l, double (&zs)[RE], double (&rs)[KQ]) __attribute__((always_inline)) {
      constexpr bool ACC = decltype(accc)::value;
#pragma unroll 1
      for (int t0 = 0; t0 < nle; t0 += 4) {
        double ya[4], yb[4], zl[4][RE], gl[4][KQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double* ev = S.epool[(t0 + u < nle) ? t0 + u : 0];
          ya[u] = ACC ? ev[pj1] : ev[NP + l];
          yb[u] = ACC ? ev[pj2] : 0.0;
#pragma unroll
          for (int q = 0; q < RE; ++q) zl[u][q] = ev[lane + 64 * q];
#pragma unroll
          for (int k = 0; k < KQ; ++k) gl[u][k] = ev[NP + lane + 64 * k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool isdrop = ((dropme >> (t0 + u)) & 1ull) != 0ull;
          double y = ACC ? __builtin_fma(pa2, yb[u], pa1 * ya[u]) : ya[u];
          if (!(t0 + u < nle)) y = 0.0;
          const double yz = ACC ? (isdrop ? y : -y) : y;
          const double yr = ACC ? y : (isdrop ? -y : y);
#pragma unroll
          for (int q = 0; q < RE; ++q) zs[q] = __builtin_fma(yz, zl[u][q], zs[q]);
#pragma unroll
          for (int k = 0; k < KQ; ++k) rs[k] = __builtin_fma(yr, gl[u][k], rs[k]);
        }
      }
      // ... and its share of the overflow events that were there when the request went up (every wave takes one in NW)
      if (nglv > 0)
        ovf_accumulate<C, ACC>(gov, S.gsign, 0, C::NW, nglv, pj1, pj2, l, pa1, pa2, lane, zs, rs);
    };
    // one round with the holders: the request goes up, (A), everybody accumulates over the events it holds, (B), the
    // partial sums come back and are added in a fixed order.  Between the barriers the engine wave does everything that
    // does not need the sums: it PLACES THE EVENT OF THE PREVIOUS ROUND (reciprocal square root, scaling, staging it for
    // its owner -- none of that is on the holders' critical path any more) and adds that event's term itself, from
    // registers; the holders accumulate over the events they already have and take the staged one after (B)
    auto post = [&](int cmd, int l) __attribute__((always_inline)) {
      nglv = ngl;
      if (lane == 0) {
        *reinterpret_cast<int4*>(&S.rq.cmd) = int4{cmd, pj1, pj2, l};
        st2(&S.rq.pa1, pa1, pa2);
        S.rq.nglv = nglv;
      }
      lds_barrier();  // (A)
    };
    auto finish = [&](double (&zs)[RE], double (&rs)[KQ]) __attribute__((always_inline)) {
      lds_barrier();  // (B)
#pragma unroll
      for (int w = 0; w < NH; ++w) {
#pragma unroll
        for (int q = 0; q < RE; ++q) zs[q] += S.part[w][lane + 64 * q];
#pragma unroll
        for (int k = 0; k < KQ; ++k) rs[k] += S.part[w][NP + lane + 64 * k];
      }
    };
    // the new event goes to its owner: straight into the engine wave's LDS pool, or staged for a holder (which takes it
    // into its registers at the next (A)); clear_slot >= 0: that working-set slot was dropped -- its column of every
    // g~ is cleared (the holders do theirs when they see the flag)
    auto has_room = [&]() __attribute__((always_inline)) {
      bool room = false;
#pragma unroll
      for (int o = 0; o <= NH; ++o) room |= cnt[o] < ((o == 0) ? MAXE : MAXL);
      room |= gov != nullptr && ngl < C::MAXG;
      return room && nev < kev;
    };
    auto place_event = [&](const double (&zv)[RE], const double (&gv)[KQ], bool is_drop, int clear_slot) __attribute__((always_inline)) -> int {
      // the owner with the fewest events that still has room (ties: the lowest index); has_room() was checked before
      int owner = -1, best = 1 << 30;
#pragma unroll
      for (int o = 0; o <= NH; ++o) {
        // (the engine wave places the events and folds in the columns while the holders accumulate, and its own events
        //  are in LDS, not in registers: it is dealt BIAS events fewer than a holder)
        const int cap = (o == 0) ? MAXE : MAXL, load = cnt[o] + (o == 0 ? C::BIAS : 0);
        if (cnt[o] < cap && load < best) {
          best = load;
          owner = o;
        }
      }
      int li = 0;
#pragma unroll
      for (int o = 0; o <= NH; ++o)
        if (o == owner) li = cnt[o];
#pragma unroll
      for (int o = 0; o <= NH; ++o)
        if (o == owner) cnt[o] += 1;
      if (clear_slot >= 0) {
        if (lane < nle) S.epool[lane][NP + clear_slot] = 0.0;
        for (int e = lane; e < ngl; e += 64) gov[(size_t)e * EV + NP + clear_slot] = 0.0;
      }
      if (owner < 0) {
        // registers and LDS are full: the event goes to the overflow pool
        GlobalF64* const dst = gov + (size_t)ngl * EV;
#pragma unroll
        for (int q = 0; q < RE; ++q) dst[lane + 64 * q] = zv[q];
#pragma unroll
        for (int k = 0; k < KQ; ++k) dst[NP + lane + 64 * k] = gv[k];
        if (lane == 0) S.gsign[ngl] = is_drop ? -1 : 1;
        ngl += 1;
        status |= QMPC_DEV_ST_SPILLED;  // informational
      }
      if (ngl > 0 && (owner < 0 || clear_slot >= 0)) {
        // (the record and the cleared entries are in L2 before the next request announces them to the other waves)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      } else {
        double* const dst = (owner == 0) ? S.epool[li] : S.stage;
#pragma unroll
        for (int q = 0; q < RE; ++q) dst[lane + 64 * q] = zv[q];
#pragma unroll
        for (int k = 0; k < KQ; ++k) dst[NP + lane + 64 * k] = gv[k];
        if (owner == 0) {
          if (is_drop) dropme |= (1ull << li);
          nle = li + 1;
        }
      }
      // (always written: the holders read this word after (B) of every round)
      if (lane == 0)
        *reinterpret_cast<int4*>(&S.rq.ing_valid) =
            int4{(owner > 0 || clear_slot >= 0) ? 1 : 0, owner, li, (is_drop ? 1 : 0) | ((clear_slot + 1) << 8)};
      nev += 1;
      return owner;
    };
    // the event the last working-set change produced, not placed yet: unscaled vectors (z or u; -r or -S^-1[:, l] on the
    // slots that were in use), the quantity whose inverse square root scales them (delta or gamma), the slot that gets
    // the scale itself (an add event) or is cleared (a drop event)
    bool pend = false, pdrop = false;
    int pone = -1, pzero = -1;
    double parg = 1.0, pz[RE], pg[KQ];
#pragma unroll
    for (int q = 0; q < RE; ++q) pz[q] = 0.0;
#pragma unroll
    for (int k = 0; k < KQ; ++k) pg[k] = 0.0;
    // ... placed in the window of the next round, whatever that round accumulates; when a holder owns it, this wave adds
    // its term (the holder takes it after (B)); in this wave's own pool (or the overflow pool) own_events covers it
    auto place_pending = [&](auto accc, int l, double (&zs)[RE], double (&rs)[KQ]) __attribute__((always_inline)) {
      constexpr bool ACC = decltype(accc)::value;
      if (!pend) {
        if (lane == 0) S.rq.ing_valid = 0;
        return;
      }
      const double s = rsqrt_full(parg);
      double zv[RE], gv[KQ];
#pragma unroll
      for (int q = 0; q < RE; ++q) zv[q] = pz[q] * s;
#pragma unroll
      for (int k = 0; k < KQ; ++k) gv[k] = (lane + 64 * k == pone) ? s : ((lane + 64 * k == pzero) ? 0.0 : pg[k] * s);
      const int owner = place_event(zv, gv, pdrop, pzero);
      pend = false;
      if (owner != 0) {  // (a holder's, or in the overflow pool: announced with the next request)
        double y = ACC ? __builtin_fma(pa2, lane_elem<RE>(zv, pj2), pa1 * lane_elem<RE>(zv, pj1)) : lane_elem<KQ>(gv, l);
        const double yz = ACC ? (pdrop ? y : -y) : y;
        const double yr = ACC ? y : (pdrop ? -y : y);
#pragma unroll
        for (int q = 0; q < RE; ++q) zs[q] = __builtin_fma(yz, zv[q], zs[q]);
#pragma unroll
        for (int k = 0; k < KQ; ++k) rs[k] = __builtin_fma(yr, gv[k], rs[k]);
      }
    };
    __builtin_amdgcn_s_setprio(2);  // the serial part of the workgroup
    // Remove working-set slot l: one drop event.  u = N*_l (variable lanes), sc = S^-1[:, l] (slot lanes) over ALL events (a
    // round with the holders), gamma = S^-1[l][l].  With `repair` (after a block start: a row whose multiplier came out
    // negative does not belong to the working set) the iterate also moves to the minimiser WITHOUT that row:
    // x -= (lam_l / gamma) u, lam -= (lam_l / gamma) sc.  false: the projected inverse lost definiteness numerically.
    auto drop_slot = [&](int l, bool repair) __attribute__((always_inline)) {
      double u[RE], sc[KQ];
#pragma unroll
      for (int q = 0; q < RE; ++q) u[q] = 0.0;
#pragma unroll
      for (int k = 0; k < KQ; ++k) sc[k] = 0.0;
      if (pend && !has_room()) {
        retry = true;
        return false;
      }
      post(CMD_DROPACC, l);
      place_pending(std::false_type{}, l, u, sc);
      own_events(std::false_type{}, l, u, sc);
      finish(u, sc);
      const double gamma = lane_elem<KQ>(sc, l);
      if (uni(!(gamma > 0.0))) {
        retry = true;
        return false;
      }
      if (repair) {
        const double coef = lane_elem<KQ>(lam, l) * fast_rcp(gamma);
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          xv[q] = __builtin_fma(-coef, u[q], xv[q]);
          S.xl[lane + 64 * q] = xv[q];
        }
#pragma unroll
        for (int k = 0; k < KQ; ++k) lam[k] = __builtin_fma(-coef, sc[k], lam[k]);
      }
      // the drop event (placed in the next round's window): N*_l / sqrt(gamma), -S^-1[:, l] / sqrt(gamma) on the slots in
      // use, zero in slot l
      const int de = lane_elem<KQ>(wcid, l);
#pragma unroll
      for (int q = 0; q < RE; ++q) pz[q] = u[q];
#pragma unroll
      for (int k = 0; k < KQ; ++k) pg[k] = (wcid[k] < 0) ? 0.0 : -sc[k];
      parg = gamma;
      pone = -1;
      pzero = l;
      pdrop = true;
      pend = true;
#pragma unroll
      for (int k = 0; k < KQ; ++k)
        if (lane + 64 * k == l) {
          wcid[k] = -1;
          lam[k] = 0.0;
        }
      const int dsl = de / 5, dty = de - 5 * dsl;
#pragma unroll
      for (int s2 = 0; s2 < SQ; ++s2)
        if (lane + 64 * s2 == dsl) amask[s2] &= ~(1u << dty);
      return true;
    };
    double c1[RE], c2[RE];  // the two columns of H^-1 of the constraint being added (rows of the symmetric work item)
    // ---- the most violated constraint outside the working set (normalised by its row norm) becomes p, and the loads of
    // its two columns are issued; false: none is violated (optimal) or the iteration limit is reached.  The stance-slot
    // lanes read x from its variable-indexed copy in LDS (kept current by every step), not through cross-lane gathers
    auto select_next = [&]() __attribute__((always_inline)) {
      unsigned key = 0;
#pragma unroll
      for (int s = 0; s < SQ; ++s) {
        const int sl = lane + 64 * s;
        const int sc = sl < nst ? sl : 0;
        const double x0 = S.xl[3 * sc], x1 = S.xl[3 * sc + 1], x2 = S.xl[3 * sc + 2];
        if (sl < nst) {
          const double fx = mi * x0, fy = mi * x1;
          double vmin = 0.0;
          int tmin = -1;
          const double sv[5] = {(fx + x2) * inv_fr, (x2 - fx) * inv_fr, (fy + x2) * inv_fr, (x2 - fy) * inv_fr, fmx[s] - x2};
#pragma unroll
          for (int ty = 0; ty < 5; ++ty) {
            const bool cand = !((amask[s] >> ty) & 1u) && sv[ty] < vmin;
            vmin = cand ? sv[ty] : vmin;
            tmin = cand ? ty : tmin;
          }
          if (vmin < -tol) {
            const unsigned kk = (__float_as_uint((float)(-vmin)) & ~0x3FFu) | (unsigned)(5 * sl + tmin);
            key = kk > key ? kk : key;
          }
        }
      }
      const unsigned best = wave_max_u32(key);
      if (best == 0u) return false;
      if (iters >= max_iter) {
        status |= QMPC_DEV_ST_MAXITER;
        return false;
      }
      p_e = (int)(best & 0x3FFu);
      psl = p_e / 5;
      pty = p_e - 5 * psl;
      con_coefs(p_e, mi, pj1, pj2, pa1, pa2);
      p_rhs = (pty == 4) ? -lane_elem<SQ>(fmx, psl) : 0.0;  // (f_max of the slot: from the slot lane's register, not LDS)
      lp = 0.0;
      // in flight across barrier (A) (which only orders LDS) and the accumulation over the events, and -- from the second
      // iteration on -- across the placement of the previous event; consumed right before (B)
      // (the work item holds the lower block triangle of H^-1: the 64-row blocks up to the column's own come from ROW
      //  pj -- symmetry, coalesced --, the blocks below it from the column itself, one row stride per lane)
      const int qb = pj1 >> 6;  // (pj1 and pj2 belong to the same stance slot; a slot may straddle two blocks)
      const int qb2 = pj2 >> 6;
#pragma unroll
      for (int q = 0; q < RE; ++q) {
        const int row = lane + 64 * q;
        c1[q] = (row < n) ? ((q <= qb) ? Hi[(size_t)pj1 * LD + row] : Hi[(size_t)row * LD + pj1]) : 0.0;
        c2[q] = (row < n) ? ((q <= qb2) ? Hi[(size_t)pj2 * LD + row] : Hi[(size_t)row * LD + pj2]) : 0.0;
      }
      return true;
    };
    // ---- after a block start: rows whose multiplier came out negative leave, most negative first, one drop event
    // each -- what remains is a Goldfarb-Idnani state (x optimal on W, multipliers >= 0)
    while (bkev > 0 && !retry) {  // (normally nothing left to do: the block start removes them itself while it has room)
      double worst = 0.0;
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        const double v = (wcid[k] >= 0 && lam[k] < 0.0) ? -lam[k] : 0.0;
        worst = v > worst ? v : worst;
      }
      const double wmax = wave_max_pos_f64(worst);
      if (!(wmax > 0.0)) break;
      if (iters >= max_iter) {
        retry = true;
        break;
      }
      int l = -1;
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        const unsigned long long hit = __ballot(wcid[k] >= 0 && lam[k] < 0.0 && -lam[k] == wmax);
        if (l < 0 && hit != 0ull) l = 64 * k + __ffsll((long long)hit) - 1;
      }
      if (!drop_slot(l, true)) break;
      iters += 1;
    }
    bool have_p = !retry && select_next();
    while (have_p) {
      iters = __builtin_amdgcn_readfirstlane(iters);
      khw = __builtin_amdgcn_readfirstlane(khw);
      nev = __builtin_amdgcn_readfirstlane(nev);
      nle = __builtin_amdgcn_readfirstlane(nle);
      ngl = __builtin_amdgcn_readfirstlane(ngl);
      pone = __builtin_amdgcn_readfirstlane(pone);
      pzero = __builtin_amdgcn_readfirstlane(pzero);
#pragma unroll
      for (int o = 0; o <= NH; ++o) cnt[o] = __builtin_amdgcn_readfirstlane(cnt[o]);
      status = __builtin_amdgcn_readfirstlane(status);
      p_e = __builtin_amdgcn_readfirstlane(p_e);
      psl = __builtin_amdgcn_readfirstlane(psl);
      pty = __builtin_amdgcn_readfirstlane(pty);
      pj1 = __builtin_amdgcn_readfirstlane(pj1);
      pj2 = __builtin_amdgcn_readfirstlane(pj2);
      QMPC_ESTAMP(0);
      // ---- room for the event the previous pass left (it is placed in this round's window)?
      if (pend && !has_room()) {
        retry = true;
        break;
      }
      // ---- z = P c_p (variable lanes), r = N*^T c_p (working-slot lanes)
      QMPC_ESTAMP(1);
      double z[RE], rw[KQ];
#pragma unroll
      for (int k = 0; k < KQ; ++k) rw[k] = 0.0;
#pragma unroll
      for (int q = 0; q < RE; ++q) z[q] = 0.0;
      double d1 = 0.0, d2 = 0.0, xp1 = 0.0, xp2 = 0.0;
      post(CMD_ACC, 0);
      {
        QMPC_ESTAMP(2);
        // (operands of the step that do not depend on the holders: read while they work)
        d1 = S.D[pj1];
        d2 = S.D[pj2];
        xp1 = S.xl[pj1];
        xp2 = S.xl[pj2];
        place_pending(std::true_type{}, 0, z, rw);
        own_events(std::true_type{}, 0, z, rw);
#pragma unroll
        for (int q = 0; q < RE; ++q)
          if (lane + 64 * q < n) z[q] += __builtin_fma(pa2, c2[q], pa1 * c1[q]);
        if (dbg_clk && lane == 0 && iters == QMPC_EDBG_ITER) {
          double zs = 0.0;  // (the stamp waits for the loads)
#pragma unroll
          for (int q = 0; q < RE; ++q) zs += z[q];
          asm volatile("" ::"v"(zs));
          dbg_clk[3] = clock64();
        }
      }
      finish(z, rw);
      QMPC_ESTAMP(4);
      const double delta = __builtin_fma(pa2, lane_elem<RE>(z, pj2), pa1 * lane_elem<RE>(z, pj1));
      const double cn = __builtin_fma(pa2 * pa2, d2, pa1 * pa1 * d1);  // scale of c_p^T H^-1 c_p
      const double sp = __builtin_fma(pa2, xp2, pa1 * xp1) - p_rhs;
      const bool dep = uni(!(delta > 1e-11 * cn));
      const double t2 = dep ? __builtin_inf() : -sp * fast_rcp(dep ? 1.0 : delta);
      double ratio[KQ], rmin = __builtin_inf();
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        ratio[k] = __builtin_inf();
        if (wcid[k] >= 0 && rw[k] > 0.0) {
          const double qv = lam[k] * fast_rcp(rw[k]);
          ratio[k] = qv > 0.0 ? qv : 0.0;
        }
        rmin = (k == 0 || ratio[k] < rmin) ? ratio[k] : rmin;
      }
      double t1 = __builtin_inf();
      int l = -1;
      if (khw > 0 && uni(rmin < __builtin_inf())) {
        // the smallest ratio: ONE 32-bit wave reduction on the ratios rounded to float (non-negative: bit pattern order ==
        // value order); the usual case -- one lane at the minimum -- is settled by a ballot, ties in float by the exact
        // two-pass reduction (the answer is the exact minimum either way)
        const unsigned rf = __float_as_uint((float)rmin);
        const unsigned mf = wave_min_u32(rf);
        const unsigned long long cand = __ballot(rf == mf);
        if ((cand & (cand - 1ull)) == 0ull) {
          t1 = readlane_f64(rmin, __ffsll((long long)cand) - 1);
        } else {
          t1 = wave_min_pos_f64(rmin);
        }
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          const unsigned long long hit = __ballot(ratio[k] == t1);
          if (l < 0 && hit != 0ull) l = 64 * k + __ffsll((long long)hit) - 1;
        }
      }
      const double t = (t2 <= t1) ? t2 : t1;
      if (dbg_clk && lane == 0 && iters == QMPC_EDBG_ITER) {
        asm volatile("" ::"v"(t));
        dbg_clk[5] = clock64();
      }
      if (uni(!(t < __builtin_inf()))) {
        status |= QMPC_DEV_ST_INFEASIBLE;
        break;
      }
      if (!dep) {
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          xv[q] = __builtin_fma(t, z[q], xv[q]);
          S.xl[lane + 64 * q] = xv[q];
        }
      }
#pragma unroll
      for (int k = 0; k < KQ; ++k) lam[k] -= t * rw[k];
      lp += t;
      iters += 1;
      if (uni(t2 <= t1)) {
        // ---- full step: p joins the working set in the first free slot (an add event)
        int qslot = -1;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          const unsigned long long fm = __ballot(wcid[k] < 0);
          if (qslot < 0 && fm != 0ull) qslot = 64 * k + __ffsll((long long)fm) - 1;
        }
        if (qslot < 0) {
          retry = true;  // out of working-set slots
          break;
        }
        // the add event (placed in the next round's window); its entries need the working set as it was: g~_w = -r_w /
        // sqrt(delta) on the slots in use, 1 / sqrt(delta) on the new one
#pragma unroll
        for (int q = 0; q < RE; ++q) pz[q] = z[q];
#pragma unroll
        for (int k = 0; k < KQ; ++k) pg[k] = (wcid[k] >= 0) ? -rw[k] : 0.0;
        parg = delta;
        pone = qslot;
        pzero = -1;
        pdrop = false;
        pend = true;
#pragma unroll
        for (int k = 0; k < KQ; ++k)
          if (lane + 64 * k == qslot) {
            wcid[k] = p_e;
            lam[k] = lp;
          }
#pragma unroll
        for (int s2 = 0; s2 < SQ; ++s2)
          if (lane + 64 * s2 == psl) amask[s2] |= (1u << pty);
        khw = (qslot + 1 > khw) ? qslot + 1 : khw;
        // the NEXT constraint is chosen and the loads of its columns issued; its request goes up at once (next pass), and
        // this event is placed while the holders accumulate
        have_p = select_next();
      } else {
        // ---- partial step: the multiplier of slot l reached zero -> drop it.  p stays (its columns are still in c1, c2)
        if (!drop_slot(l, false)) break;
      }
      __builtin_amdgcn_wave_barrier();
      if (dbg_clk && lane == 0 && iters == QMPC_EDBG_ITER + 1) dbg_clk[7] = clock64();
    }
    __builtin_amdgcn_s_setprio(0);
    if (lane == 0) S.rq.cmd = CMD_DONE;
    lds_barrier();  // (A) of the last round: the holders leave
    if (dbg_clk && lane == 0) dbg_clk[6] = clock64();
    if (!retry) {
      // outputs: get_solution(0..11) = forces of the four feet at horizon step 0 (convexMPC_interface.cpp:175-180,
      // ConvexMPCLocomotion.cpp:672-685); an iterate the method abandoned is not a solution: zeros and a status
      const bool dead = (status & (QMPC_DEV_ST_INFEASIBLE | QMPC_DEV_ST_WS_FULL)) != 0;
      if (lane < 12) P.grf[(size_t)rid * 12 + lane] = 0.f;
      __builtin_amdgcn_wave_barrier();
      bool nf = false;
#pragma unroll
      for (int q = 0; q < RE; ++q) {
        const int j = lane + 64 * q;
        if (j < n && !dead) {
          const int k = S.sidx[j / 3], ax = j % 3;  // foot-step of this variable
          if (k < 4) P.grf[(size_t)rid * 12 + 3 * k + ax] = (float)xv[q];
          if (P.soln) P.soln[(size_t)rid * 12 * h + 3 * k + ax] = xv[q];
        }
        nf |= (j < n) && !(__builtin_fabs(xv[q]) < __builtin_inf());
      }
      if (__ballot(nf)) status |= QMPC_DEV_ST_NONFINITE;
      const bool cmdm = P.c_position != nullptr;
      if (lane == 0) {
        P.status[rid] = S.status0 | status;
        if (P.iters) P.iters[rid] = iters;
        if (cmdm) {
          // the controller state owned by the packer, advanced by the run that produces the result
          // (ConvexMPCLocomotion.cpp:534-545, :632-640)
          const float* pos = P.c_position + (size_t)rid * 3;
          if (!(P.c_gait_type && P.c_gait_type[rid] == 4)) {
            P.c_wpd[(size_t)rid * 2 + 0] = qmpc_cmd_clamp(P.c_wpd[(size_t)rid * 2 + 0], pos[0]);
            P.c_wpd[(size_t)rid * 2 + 1] = qmpc_cmd_clamp(P.c_wpd[(size_t)rid * 2 + 1], pos[1]);
          }
          P.c_xci[rid] = qmpc_cmd_xci_next(P.c_xci[rid], pos[2], P.c_body_height, (float)P.dt, P.c_v_world[(size_t)rid * 3 + 0]);
        }
      }
      if (cmdm && P.f_ff) {
        if (lane < 12) S.fb[lane] = 0.f;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < RE; ++q) {
          const int j = lane + 64 * q;
          if (j < n && !dead && S.sidx[j / 3] < 4) S.fb[3 * S.sidx[j / 3] + j % 3] = (float)xv[q];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 12) {
          const int leg = lane / 3, ii = lane - 3 * leg;
          P.f_ff[(size_t)rid * 12 + lane] =
              qmpc_cmd_f2b(P.c_r_body + (size_t)rid * 9 + 3 * ii, S.fb[3 * leg], S.fb[3 * leg + 1], S.fb[3 * leg + 2]);
        }
      }
    } else if (lane == 0) {
      // handed back: the monolithic kernel of the class solves this robot from scratch (its pool is larger)
      const int slot = atomicAdd(P.fb_count, 1);
      P.fb_list[slot] = rid;
    }
  } else {
    // =============================================================== an event holder
    double zt[MAXL][RE], gt[MAXL][KQ];
#pragma unroll
    for (int li = 0; li < MAXL; ++li) {
#pragma unroll
      for (int q = 0; q < RE; ++q) zt[li][q] = 0.0;
#pragma unroll
      for (int k = 0; k < KQ; ++k) gt[li][k] = 0.0;
    }
    int nloc = 0, hround = 0;
    const GlobalF64* const hgov = P.wk_ovf ? (const GlobalF64*)P.wk_ovf + (size_t)blockIdx.x * ((size_t)C::MAXG * EV) : nullptr;
    unsigned long long dropm = 0ull;  // local events that are drop events
    if (bkev > 0) {
      // the block start's records (transposed in LDS: Rt[entry][record]): record e goes to holder 1 + e mod NH,
      // registers [e / NH]
      const double(*const Rt)[C::MAXB] = reinterpret_cast<const double(*)[C::MAXB]>(&S.epool[0][0]);
      nloc = (bkev > wv - 1) ? (bkev - (wv - 1) + NH - 1) / NH : 0;
      Upto<0, MAXL>::run(nloc, [&](auto lic) __attribute__((always_inline)) {
        constexpr int li = decltype(lic)::value;
        const int e = li * NH + (wv - 1);
#pragma unroll
        for (int q = 0; q < RE; ++q) zt[li][q] = Rt[lane + 64 * q][e];
#pragma unroll
        for (int k = 0; k < KQ; ++k) gt[li][k] = Rt[NP + lane + 64 * k][e];
        if (S.bk.sign[e] < 0) dropm |= (1ull << li);
      });
      lds_barrier();  // (the phase's LDS becomes the engine wave's event pool after this)
    }
    while (true) {
      lds_barrier();  // (A)
      const bool hst = dbg_clk && wv == 1 && lane == 0 && hround == QMPC_EDBG_ITER;  // (rounds ~ iterations while nothing is dropped)
      hround += 1;
      if (hst) dbg_clk[8] = clock64();
      // the request: two broadcast loads in flight together (one LDS round trip, not one per field)
      int4 r0 = *reinterpret_cast<const int4*>(&S.rq.cmd);
      F64x2 r1 = ld2(&S.rq.pa1);
      int hng = S.rq.nglv;
      asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(hng));
      const int cmd = __builtin_amdgcn_readfirstlane(r0.x);
      if (cmd == CMD_DONE) break;
      nloc = __builtin_amdgcn_readfirstlane(nloc);
      hng = __builtin_amdgcn_readfirstlane(hng);  // overflow events this round covers: this wave takes every NW-th
      if (hst) dbg_clk[9] = clock64();
      double zp[RE], rp[KQ];
#pragma unroll
      for (int q = 0; q < RE; ++q) zp[q] = 0.0;
#pragma unroll
      for (int k = 0; k < KQ; ++k) rp[k] = 0.0;
      if (cmd == CMD_ACC) {
        // z -= +-y z~ , r += y g~ with y = z~^T c_p = a1 z~[j1] + a2 z~[j2] over this holder's events; the two entries of
        // z~ come by readlane from registers whose 64-row block is a compile-time constant of the specialised loop
        const int hj1 = __builtin_amdgcn_readfirstlane(r0.y), hj2 = __builtin_amdgcn_readfirstlane(r0.z);
        // (the coefficients stay in VECTOR registers -- the entries of z~ arrive in scalar ones, and an instruction takes one
        //  scalar operand: two would cost a copy each -- and an event's sign is a multiplication by +-1.0 held in a scalar
        //  pair, not a negate-and-select: 10 vector instructions per event in the 128-row class where there were 17, in the
        //  loop that bounds the iteration)
        double ha1 = r1.x, ha2 = r1.y;
        asm volatile("" : "+v"(ha1), "+v"(ha2));
        const int l1 = hj1 & 63, l2 = hj2 & 63;
        dispatch_blocks<RE>(hj1 >> 6, hj2 >> 6, [&](auto q1c, auto q2c) __attribute__((always_inline)) {
          constexpr int Q1 = decltype(q1c)::value, Q2 = decltype(q2c)::value;
          Upto<0, MAXL>::run(nloc, [&](auto lic) __attribute__((always_inline)) {
            constexpr int li = decltype(lic)::value;
            const double y = __builtin_fma(ha2, readlane_f64(zt[li][Q2], l2), ha1 * readlane_f64(zt[li][Q1], l1));
            const double ys = y * event_sign((dropm >> li) & 1ull, true);
#pragma unroll
            for (int q = 0; q < RE; ++q) zp[q] = __builtin_fma(ys, zt[li][q], zp[q]);
#pragma unroll
            for (int k = 0; k < KQ; ++k) rp[k] = __builtin_fma(y, gt[li][k], rp[k]);
          });
        });
        if (hng > 0) ovf_accumulate<C, true>(hgov, S.gsign, wv, C::NW, hng, hj1, hj2, 0, ha1, ha2, lane, zp, rp);
      } else {
        // u = N*_l = sum z~ g~[l] ,  sc = S^-1[:, l] = sum_add g~ g~[l] - sum_drop g~ g~[l]
        const int hl = __builtin_amdgcn_readfirstlane(r0.w);
        Upto<0, MAXL>::run(nloc, [&](auto lic) __attribute__((always_inline)) {
          constexpr int li = decltype(lic)::value;
          const double y = lane_elem<KQ>(gt[li], hl);
          const double ys = y * event_sign((dropm >> li) & 1ull, false);
#pragma unroll
          for (int q = 0; q < RE; ++q) zp[q] = __builtin_fma(y, zt[li][q], zp[q]);
#pragma unroll
          for (int k = 0; k < KQ; ++k) rp[k] = __builtin_fma(ys, gt[li][k], rp[k]);
        });
        if (hng > 0) ovf_accumulate<C, false>(hgov, S.gsign, wv, C::NW, hng, 0, 0, hl, 0.0, 0.0, lane, zp, rp);
      }
      if (hst) {
        double zs = 0.0;
#pragma unroll
        for (int q = 0; q < RE; ++q) zs += zp[q];
        asm volatile("" ::"v"(zs));
        dbg_clk[10] = clock64();
      }
      double* const mine = S.part[wv - 1];
#pragma unroll
      for (int q = 0; q < RE; ++q) mine[lane + 64 * q] = zp[q];
#pragma unroll
      for (int k = 0; k < KQ; ++k) mine[NP + lane + 64 * k] = rp[k];
      if (hst) dbg_clk[11] = clock64();
      lds_barrier();  // (B)
      // ---- the event the engine placed in this round's window (if any), and the dropped working-set slot whose column
      // of every g~ is cleared: after the sums are out, off everybody's critical path.  One LDS round trip for the
      // word and the staged record, whoever it is for
      {
        int4 r2 = *reinterpret_cast<const int4*>(&S.rq.ing_valid);
        double zin[RE], gin[KQ];
#pragma unroll
        for (int q = 0; q < RE; ++q) zin[q] = S.stage[lane + 64 * q];
#pragma unroll
        for (int k = 0; k < KQ; ++k) gin[k] = S.stage[NP + lane + 64 * k];
        asm volatile("" : "+v"(r2.x), "+v"(r2.y), "+v"(r2.z), "+v"(r2.w), "+v"(zin[0]), "+v"(gin[0]));
        if (__builtin_amdgcn_readfirstlane(r2.x) != 0) {
          const int flags = __builtin_amdgcn_readfirstlane(r2.w);
          const int clr = (flags >> 8) - 1;
          if (clr >= 0) {
            const int ck = clr >> 6, cl = clr & 63;
            Upto<0, MAXL>::run(nloc, [&](auto lic) __attribute__((always_inline)) {
              constexpr int li = decltype(lic)::value;
#pragma unroll
              for (int k = 0; k < KQ; ++k) gt[li][k] = (k == ck && lane == cl) ? 0.0 : gt[li][k];
            });
          }
          if (__builtin_amdgcn_readfirstlane(r2.y) == wv) {
            const int myli = __builtin_amdgcn_readfirstlane(r2.z);
            StaticFor<0, MAXL>::run([&](auto lic) __attribute__((always_inline)) {
              constexpr int li = decltype(lic)::value;
              if (li == myli) {
#pragma unroll
                for (int q = 0; q < RE; ++q) zt[li][q] = zin[q];
#pragma unroll
                for (int k = 0; k < KQ; ++k) gt[li][k] = gin[k];
                // (a different instruction at the END of every case: the compiler otherwise sinks the stores of all cases
                //  into one store through a pointer chosen at run time, and the whole register array becomes scratch)
                asm volatile("; event registers %0" ::"n"(li));
              }
            });
            if (flags & 1) dropm |= (1ull << myli);
            nloc = myli + 1;
          }
        }
      }
    }
  }
  __syncthreads();  // every wave is done with this item's LDS state
}

}  // namespace

// One robot per workgroup, the items of the class consumed as a queue (entry blockIdx.x first, then whatever the head
// counter hands out).  WARM is reserved (the warm start across MPC cycles runs in the monolithic kernels).
template <int RE, int KQ, int SQ, int NW, int MAXL, int MAXE>
__global__ __launch_bounds__(64 * NW, 2) void qmpc_engine_kernel(const QmpcParams P) {
  using C = ECfg<RE, KQ, SQ, NW, MAXL, MAXE>;
  extern __shared__ __attribute__((aligned(16))) unsigned char qmpc_esmem[];
  ESmem<C>& S = *reinterpret_cast<ESmem<C>*>(qmpc_esmem);
  // the counter group of the NEXT chunk of this class (its previous user, the chunk before this one, has finished)
  if (blockIdx.x == 0 && P.wk_zero && threadIdx.x < QMPC_GRP_INTS) P.wk_zero[threadIdx.x] = 0;
  const int nitems = *P.wk_count;
  if ((int)blockIdx.x >= nitems) return;  // uniform
  static_assert(sizeof(QmpcParams) % 4 == 0 && sizeof(QmpcParams) / 4 <= 64 * NW, "parameter block copy");
  if (threadIdx.x < sizeof(QmpcParams) / 4)
    reinterpret_cast<uint32_t*>(&S.par)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&P)[threadIdx.x];
  // queue position -> work item: the sweep kernel filed the items in QMPC_ORDER_BUCKETS lists, hardest robots first
  // (complete: this kernel starts after the sweep kernel of its chunk has finished)
  auto item_at = [&](int pos) __attribute__((always_inline)) {
    int b = 0, cb = P.wk_bucket[0];
    while (b + 1 < QMPC_ORDER_BUCKETS && pos >= cb) {
      pos -= cb;
      cb = P.wk_bucket[++b];
    }
    return P.wk_order[(size_t)b * P.wk_cap + P.wk_base + pos];
  };
  if (threadIdx.x == 0) S.qnext = item_at((int)blockIdx.x);
  __syncthreads();
  for (;;) {
    const int item = S.qnext;
    int tid1 = (int)threadIdx.x;
    asm volatile("" : "+v"(tid1));
    __builtin_assume(tid1 >= 0 && tid1 < 64 * NW);
    engine_item<C, false>(item, tid1, S, P);  // (ends with a workgroup barrier)
    if (threadIdx.x == 0) {
      const int pos = (int)gridDim.x + atomicAdd(P.wk_qhead, 1);
      S.qnext = pos < nitems ? item_at(pos) : -1;
    }
    __syncthreads();
    if (S.qnext < 0) break;  // uniform
  }
}

namespace {
template <int RE, int KQ, int SQ, int NW, int MAXL, int MAXE>
struct EngineEntry {
  using C = ECfg<RE, KQ, SQ, NW, MAXL, MAXE>;
  static hipError_t prepare() {
    return hipFuncSetAttribute((const void*)qmpc_engine_kernel<RE, KQ, SQ, NW, MAXL, MAXE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)sizeof(ESmem<C>));
  }
  static int resident() {
    static int cached = 0;
    if (cached) return cached;
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, qmpc_engine_kernel<RE, KQ, SQ, NW, MAXL, MAXE>, 64 * NW,
                                                                      sizeof(ESmem<C>));
    if (e != hipSuccess || per < 1 || cus < 1) return 0;
    return cached = per * cus;
  }
  static hipError_t launch(const QmpcParams* P, int grid, hipStream_t stream) {
    hipLaunchKernelGGL((qmpc_engine_kernel<RE, KQ, SQ, NW, MAXL, MAXE>), dim3(grid), dim3(64 * NW), sizeof(ESmem<C>), stream, *P);
    return hipGetLastError();
  }
};
// engine of the 128-row class: 2 row blocks, 64 working slots, 64 stance slots; 3 holders x 26 events (6 VGPRs per event;
// 256 VGPRs per lane at two waves per SIMD, ~100 of them for everything else)
typedef EngineEntry<2, 1, 1, 4, 23, 42> Engine2;
// engine of the 192-row class: 3 row blocks, 128 working slots (all four feet down at horizon 14 / 16 ends with 70-90 rows at
// a bound when braking): 10 VGPRs per event.  Four waves and 79 KB of LDS -- TWO workgroups per CU (the class's engine
// phase is bound by the number of robots in flight, not by the slowest robot): 3 holders x 13 events + 25 in LDS, the
// rest of an unusually long history in the overflow pool
typedef EngineEntry<3, 2, 1, 4, 13, 25> Engine3;
// engine of the large problems (192 < n_r <= 432: seven row blocks, 192 working slots, up to 144 stance foot-steps): 20 VGPRs
// per event -- 3 holders x 6 in registers, 24 in LDS, the rest of the (long) histories in the overflow pool; one workgroup per CU
typedef EngineEntry<7, 3, 3, 4, 6, 24> Engine7;
static_assert(sizeof(ESmem<Engine3::C>) <= 80 * 1024 && sizeof(ESmem<Engine2::C>) <= 80 * 1024, "two engine workgroups per CU");

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// JCQP alternate on the LARGE problems (use_jcqp = 1 / 2 at horizons above 16: 192 < n <= 432; SURVEY row a10,
// src/JCQP/QpProblem.cpp:178-269 as SolverMPC.cpp:400-420, :558-631 drive it).  The producer (qmpc_big_kernel) has left
// M^-1 = (P + sigma I + A^T R A)^-1 -- the KKT matrix reduced to the x block, its friction part diagonal -- and the gradient
// in the robot's work item; this kernel runs the same ADMM as the ADMM instantiations of qmpc_kernels.hip (stage 5 there),
// one workgroup per item: thread = variable (x, M x), thread = foot-step (z, y, A x of its five rows), x~ = M^-1 rhs as a
// mat-vec over the ROWS of the symmetric item (row j is column j: coalesced across the threads), 1.5 MB per iteration --
// coverage of the interface, not speed.  Same updates, same stopping rule, same outputs (the ADMM iterate, not a minimiser).
constexpr int ADMM_BIG_NT = QMPC_BIG_LD;
__global__ __launch_bounds__(ADMM_BIG_NT) void qmpc_admm_big_kernel(const QmpcParams P) {
  constexpr int LD = QMPC_BIG_LD, NT = ADMM_BIG_NT, NWV = NT / 64;
  __shared__ double rv[LD], xs[LD], cw[LD], red[2 * NWV];
  __shared__ int s_next;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // the counter group of the NEXT chunk of this class (see qmpc_engine_kernel)
  if (blockIdx.x == 0 && P.wk_zero && threadIdx.x < QMPC_GRP_INTS) P.wk_zero[threadIdx.x] = 0;
  const int nitems = *P.wk_count;
  if ((int)blockIdx.x >= nitems) return;  // uniform
  const int h = P.horizon;
  for (int item = (int)blockIdx.x;;) {
    const QmpcWorkHdr* const hd = P.wk_hdr + item;
    const GlobalF64* const Mi = (const GlobalF64*)P.wk_hinv + (size_t)item * ((size_t)LD * LD);
    const int rid = hd->rid, n = hd->n, nst = hd->nst, status0 = hd->status0;
    const double mi = P.mu_inv, sig = P.admm_sigma, al = P.admm_alpha, rho = P.admm_rho, rinf = 1e-6;
    const double big = (double)5e10f;  // BIG_NUMBER through float (SolverMPC.cpp:15, :356)
    const int max_it = P.admm_max_iter;
    // ---- this thread's variable j = tid ...
    const bool isv = tid < n;
    const double gq = isv ? ((const GlobalF64*)P.wk_xu)[(size_t)item * LD + tid] : 0.0;
    const double f3 = isv ? (double)hd->fmaxk[tid / 3] : 1.0;
    const double rr3 = (__builtin_fabs(f3) < 1e-10) ? rho * 1e3 : (f3 > 1e10 ? rinf : rho);
    const double dj = sig + ((tid % 3 < 2) ? 2.0 * rinf * mi * mi : 4.0 * rinf + rr3);  // diag(M - P)
    double x = 0.0, mx = 0.0;  // cold start
    // ---- ... and foot-step sl = tid (computeConstraintInfos :276-291 for its fz <= f_max row)
    const bool iss = tid < nst;
    const double fmx = iss ? (double)hd->fmaxk[tid] : 0.0;
    double r4 = rho;
    if (__builtin_fabs(fmx) < 1e-10) r4 = rho * 1e3;
    else if (fmx > 1e10) r4 = rinf;
    const double ir4 = 1.0 / r4, irinf = 1.0 / rinf;
    double z[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0}, ax[5] = {0, 0, 0, 0, 0};
    double resid = __builtin_inf();
    int iters = 0;
    for (int it = 1; it <= max_it; ++it) {
      // rhs = sigma x - q + A^T (R z - y)                                       (solveLinearSystem :315-323)
      if (iss) {
        const double w0 = rinf * z[0] - y[0], w1 = rinf * z[1] - y[1], w2 = rinf * z[2] - y[2], w3 = rinf * z[3] - y[3],
                     w4 = r4 * z[4] - y[4];
        cw[3 * tid] = mi * (w0 - w1);
        cw[3 * tid + 1] = mi * (w2 - w3);
        cw[3 * tid + 2] = (w0 + w1) + (w2 + w3) + w4;
      }
      __syncthreads();
      const double rhs = isv ? sig * x - gq + cw[tid] : 0.0;
      rv[tid] = rhs;
      __syncthreads();
      // x~ = M^-1 rhs: sum over the rows j of the item, ascending, eight loads in flight
      double xt = 0.0;
      for (int j0 = 0; j0 < n; j0 += 8) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = (isv && j0 + u < n) ? Mi[(size_t)(j0 + u) * LD + tid] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) xt = __builtin_fma(mv[u], rv[(j0 + u < n) ? j0 + u : 0], xt);
      }
      xs[tid] = xt;
      // x, M x (stepX :340-347)
      x = al * xt + (1.0 - al) * x;
      mx = al * rhs + (1.0 - al) * mx;
      __syncthreads();
      // z~ = A x~ on the foot-step threads (f_block rows, SolverMPC.cpp:366-370); z, y, A x (stepZ :349-358, stepY :360-367)
      double pmax = 0.0;
      if (iss) {
        const double t0 = xs[3 * tid], t1 = xs[3 * tid + 1], t2 = xs[3 * tid + 2];
        const double zt[5] = {mi * t0 + t2, -mi * t0 + t2, mi * t1 + t2, -mi * t1 + t2, t2};
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const double rr = (r < 4) ? rinf : r4, ir = (r < 4) ? irinf : ir4, ub = (r < 4) ? big : fmx;
          const double zr = al * zt[r] + (1.0 - al) * z[r];
          double zn = zr + ir * y[r];
          zn = zn < 0.0 ? 0.0 : zn;
          zn = zn > ub ? ub : zn;
          y[r] = y[r] + rr * (zr - zn);
          ax[r] = al * zt[r] + (1.0 - al) * ax[r];  // A x of the relaxed iterate
          const double pr = __builtin_fabs(ax[r] - z[r]);  // ... against the PREVIOUS z (:388)
          pmax = pr > pmax ? pr : pmax;
          z[r] = zn;
        }
      }
      iters = it;
      if (it % 10 == 0) {  // residual check (:238-247): (|A x - z_prev|_inf + |P x + q + A^T y|_inf) / 4
        if (iss) {
          cw[3 * tid] = mi * (y[0] - y[1]);
          cw[3 * tid + 1] = mi * (y[2] - y[3]);
          cw[3 * tid + 2] = (y[0] + y[1]) + (y[2] + y[3]) + y[4];
        }
        __syncthreads();
        const double dv = isv ? __builtin_fabs((mx - dj * x) + gq + cw[tid]) : 0.0;
        const double pm = wave_max_pos_f64(pmax), dm = wave_max_pos_f64(dv);
        if (lane == 0) {
          red[wv] = pm;
          red[NWV + wv] = dm;
        }
        __syncthreads();
        double pmx = 0.0, dmx = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
          pmx = red[w] > pmx ? red[w] : pmx;
          dmx = red[NWV + w] > dmx ? red[NWV + w] : dmx;
        }
        resid = (dmx + pmx) * 0.25;
        if (resid < P.admm_term || it >= max_it) break;  // uniform
      }
    }
    // outputs: q_soln = the ADMM iterate (SolverMPC.cpp:598-602 / :613-617), not an exact minimiser
    if (tid < 12) P.grf[(size_t)rid * 12 + tid] = 0.f;
    __syncthreads();
    bool nf = false;
    if (isv) {
      const int k = hd->sidx[tid / 3], ax3 = tid % 3;
      if (k < 4) P.grf[(size_t)rid * 12 + 3 * k + ax3] = (float)x;
      if (P.soln) P.soln[(size_t)rid * 12 * h + 3 * k + ax3] = x;
      nf = !(__builtin_fabs(x) < __builtin_inf());
    }
    const int anynf = __syncthreads_or(nf ? 1 : 0);
    if (tid == 0) {
      int status = (resid < P.admm_term) ? 0 : QMPC_DEV_ST_MAXITER;
      if (anynf) status |= QMPC_DEV_ST_NONFINITE;
      P.status[rid] = status0 | status;
      if (P.iters) P.iters[rid] = iters;
      s_next = (int)gridDim.x + atomicAdd(P.wk_qhead, 1);
    }
    __syncthreads();
    item = s_next;
    if (item >= nitems) break;  // uniform
  }
}

extern "C" hipError_t qmpc_admm_big_launch(const QmpcParams* P, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(qmpc_admm_big_kernel, dim3(grid), dim3(ADMM_BIG_NT), 0, stream, *P);
  return hipGetLastError();
}

extern "C" hipError_t qmpc_engine_prepare(void) {
  hipError_t e = Engine2::prepare();
  if (e != hipSuccess) return e;
  if ((e = Engine7::prepare()) != hipSuccess) return e;
  return Engine3::prepare();
}
extern "C" int qmpc_engine_resident(int rb) {  // (rb 5: the large problems)
  return rb == 2 ? Engine2::resident() : (rb == 3 ? Engine3::resident() : (rb == 5 ? Engine7::resident() : 0));
}
extern "C" int qmpc_engine_capacity(int rb) {
  return rb == 2 ? Engine2::C::KEV : (rb == 3 ? Engine3::C::KEV : (rb == 5 ? Engine7::C::KEV : 0));
}
extern "C" hipError_t qmpc_engine_launch(int rb, const QmpcParams* P, int grid, hipStream_t stream) {
  if (rb == 2) return Engine2::launch(P, grid, stream);
  if (rb == 3) return Engine3::launch(P, grid, stream);
  if (rb == 5) return Engine7::launch(P, grid, stream);
  return hipErrorInvalidValue;
}
