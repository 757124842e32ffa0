// kernels.hip — hand-written gfx950 (CDNA4) kernels of the LP_MP dual block-coordinate-ascent sweep.
//
// What one launch does: every wavefront (or sub-wave lane group) takes ONE updated factor of the
// current level and executes the reference's FactorContainer::UpdateFactor on it
// (reference include/factors_messages.hxx:2256-2261): pull a weight-1 message through every active
// receiving message, then push omega-weighted messages through every active sending message, all
// computed from the factor state after the receives (the reference's tmp_factor copy, :2799-2805).
// Factors of one level touch disjoint memory (plan.cpp, level scheduling), so the result equals the
// sequential sweep of LP::ComputePass (reference include/LP_MP.h:981-1005).
//
// The inner op is min-plus over doubles: no MFMA.  The kernels are HBM-bound streams of pairwise
// tables (dense) or short vectors (Potts); they are built around coalesced 16-B-per-lane loads of the
// tables, LDS staging of the label vectors, and DPP / permute min-reductions inside 64-wide waves.
// All arithmetic is IEEE double without contraction (-ffp-contract=off) so that duals are
// bit-identical to the sequential CPU semantics (min and + are exact; evaluation order is copied).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "plan.hpp"

// The LPMP_ABLATE_* switches below REMOVE work from the kernel bodies (loads, reductions, bound tracking) to price it
// (tools/build_variant.sh, EXPERIMENTS.md): such a build computes WRONG duals.  They are only accepted together with
// LPMP_EXPERIMENT_BUILD, which tools/build_variant.sh sets and lp_mp_amd/build.py never does; the product library
// additionally exports lpmp_experiment_build() == 0 (tests/test_product_isolation.py).
#if (defined(LPMP_ABLATE_TABLE) || defined(LPMP_ABLATE_RECV_VEC) || defined(LPMP_ABLATE_SEND_VEC) || defined(LPMP_ABLATE_REDUCE) || \
     defined(LPMP_ABLATE_LB_TRACK)) && !defined(LPMP_EXPERIMENT_BUILD)
#error "LPMP_ABLATE_* builds compute wrong results: they need -DLPMP_EXPERIMENT_BUILD (tools/build_variant.sh) and must never be the product library"
#endif
#if defined(LPMP_ABLATE_TABLE) || defined(LPMP_ABLATE_RECV_VEC) || defined(LPMP_ABLATE_SEND_VEC) || defined(LPMP_ABLATE_REDUCE) || defined(LPMP_ABLATE_LB_TRACK)
#define LPMP_ANY_ABLATION 1
#else
#define LPMP_ANY_ABLATION 0
#endif
extern "C" int lpmp_experiment_build(void) {
#ifdef LPMP_EXPERIMENT_BUILD
  return 1 + LPMP_ANY_ABLATION;     // 1: an experimental build (tuning knobs only), 2: with ablations (wrong results)
#else
  static_assert(LPMP_ANY_ABLATION == 0, "ablations in a product build");
  return 0;
#endif
}

namespace lpmp {

#define LPMP_INF (__builtin_inf())
constexpr int GEN_MAXD = 512;       // generic kernel: max dual size / message length held in LDS per wave
constexpr int GEN_WAVES = 4;
constexpr int GEN_ADAPTIVE_SENDS = 64;   // adaptive send rule: sends per updated factor whose improvements a wave keeps

// Tracked lower bounds.  lb[f] holds FactorContainer::LowerBound of factor f, or NaN when it has to be
// recomputed.  A sweep kernel knows the bound of every factor it touches for free:
//   own vector factor            min(theta) after the update
//   pairwise peer after a receive on side s:  min_x (m_s_new[x] + q[x]) with q the min-marginal part it just
//                                computed  ( = min_a (m1[a] + min_b (T[a][b] + m2[b])), reference LP_MP.h:1507-1518 )
//   pairwise peer after a send   unknown without a table scan -> NaN
// so LP::LowerBound after a pass is a sum over an array instead of another read of all tables.
#define LPMP_NAN (__builtin_nan(""))

// Loads / stores of data that is touched once per launch and is far larger than L2 + Infinity Cache (tables and
// message vectors of an HBM-sized model): non-temporal, so that they do not displace anything and the memory
// pipeline does not try to keep them (measured on C3: -5 % time with nt table loads, -8 % with the vectors too).
// NT = false for cache-resident models (C2: the whole state lives in the Infinity Cache between launches).
template <bool NT, class T> __device__ __forceinline__ T ld_stream(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT, class T> __device__ __forceinline__ void st_stream(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
// Access policy of the DUALS (theta, message vectors).  ACC_PLAIN / ACC_NT: one launch per level, the launch boundary
// makes other workgroups' stores visible.  ACC_COH: the chain executor below — updates of different levels run inside
// ONE launch and hand their results over through flags, so every dual load / store is an agent-scope access
// (global_load / global_store ... sc1: write-through stores, loads that do not hit a stale line of this CU's L1; the
// XCDs' L2s are not coherent with each other, MI355X_MICROARCH.md "inter-workgroup visibility").  The pairwise tables
// are constants and keep the streaming policy in every mode.
// ACC_WG: the level loop below — one WAVE hands values from level to level: loads that do not stop at this CU's L1 (sc0).
enum Access : int { ACC_PLAIN = 0, ACC_NT = 1, ACC_COH = 2, ACC_WG = 3 };
template <int A> __device__ __forceinline__ double ld_dual(const double* p) {
  if constexpr (A == ACC_COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if constexpr (A == ACC_WG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if constexpr (A == ACC_NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int A> __device__ __forceinline__ void st_dual(double* p, double v) {
  if constexpr (A == ACC_COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if constexpr (A == ACC_NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// tracked lower bounds: several updates of ONE launch may write the same slot in the chain executor (a factor is
// touched at several levels); plain stores from different XCDs would reach memory in no particular order
template <int A> __device__ __forceinline__ void st_lb(double* p, double v) {
  if constexpr (A == ACC_COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}

// ---- chain executor: dependent levels inside one persistent launch -------------------------------------------------
// A deep schedule (row-major grids: one level per anti-diagonal; chains of tiny factors) is latency-bound when every
// level is its own launch.  Here the workgroups of ONE launch take "tickets" — blocks of records, numbered in level
// order — from a global counter.  A ticket names the tickets that hold its records' predecessors (the last earlier
// update of every factor it touches, from the host's level analysis); the workgroup requests everything constant
// (packet, pairwise tables) first, THEN waits for those tickets' completion flags, then loads the duals.  A ticket
// only ever waits for lower tickets, and those are held by workgroups that are already running, so the scheme needs
// no assumption about residency or dispatch order.  Every wait is bounded: on a timeout the launch sets an abort word
// and drains, and the host reports an error instead of hanging the device.
// What it DOES assume: the workgroups that are resident keep running — the device is this process's.  When several
// processes oversubscribe one device, its scheduler time-slices their queues XCD by XCD: workgroups of this launch that
// hold tickets can be switched out on one XCD for as long as another process's persistent launch runs there, while the
// workgroups on the other XCDs poll for those tickets — and that other launch may be waiting the same way for an XCD this
// one occupies.  Measured (round 4, 8 rank processes sharing the one GPU of the test box, profiles/r04_8rank_stall_*):
// about every third run stalled until the bound below ended it, every load and barrier of the stalled workgroups frozen
// for exactly as long, neither slower polls nor a read-modify-write publish changed it.  One process per device (what the
// engine is built for) never has this; drivers that put several ranks on one device for smoke runs switch the
// persistent launches off (bench.py; LPMP_NO_CHAIN=1 LPMP_NO_BLOCKED_PASSES=1).
struct ChainArgs {
  const int32_t* dep_off;    // [n_tickets + 1]
  const int32_t* dep;        // predecessor tickets
  int32_t* done;             // [n_tickets]: epoch of the run that completed the ticket
  int32_t* next;             // ticket counter (zeroed before the launch)
  int32_t* abort_flag;
  const int32_t* tk_launch;  // [n_tickets]: launch (level x class range) the ticket belongs to
  const int32_t* tk_block;   // [n_tickets]: block of records inside that launch
  int32_t n_tickets;
  int32_t epoch;
  long long* trace;          // debugging (LPMP_CHAIN_TRACE): 8 slots of time stamps per ticket, 100 MHz; nullptr otherwise
  // joined passes (engine.cpp rotation_chain) with per-pass lower bounds: row r = the tracked bounds of all factors as they
  // are at the END OF PASS r + 1 of the call; a launch writes into the row its ChainLaunch::hist names (nullptr: no rows)
  double* lb_hist; int64_t hist_stride;
  // launches with CHAIN_LAUNCH_MAILBOX (plan.cpp): rows of L granule pairs, see mailbox_put / mailbox_take
  unsigned long long* mailbox;
  // bound of every wait in ticks of s_memrealtime (100 MHz; engine.cpp: 20 s, LPMP_CHAIN_TIMEOUT_S).  Time, not a number of
  // polls: a device shared by several processes (N ranks of a smoke run on one GPU) serves a poll an order of magnitude
  // slower, and a count of polls that means seconds on an idle device was reached there by waits that were merely slow
  long long timeout_ticks;
  // PERIODIC ticket lists (the joined passes of lpmp_compute_pass(n), engine.cpp rotation_chain): the arrays above describe
  // a TEMPLATE — prologue tickets [0, per_begin), ONE period of per_len tickets, epilogue — and the launch executes the
  // period per_count times: ticket t of the launch is template ticket t - q * per_len of copy q = min((t - per_begin) /
  // per_len, per_count - 1) (0 in the prologue); its launch is the template's + q * per_launch_shift (every copy is a group
  // of as many steps later), its dependencies the template's + q * per_len, its bound row the template's + q * per_row_shift.
  // per_len == 0: plain lists.  Host work and device memory of an n-pass launch are then independent of n.
  int32_t per_begin, per_len, per_count, per_launch_shift, per_row_shift;
  // rows of lb_hist the launch may write (an n-pass call has n - 1 seams): a W step of a periodic template carries a row
  // even when, in a call that ends right behind it, it is the LAST step before T and has no seam behind it
  int32_t hist_rows;
  // ring > 0: done[] has `ring` slots, ticket t publishes {epoch, t / ring} into slot t % ring AFTER ticket t - ring has
  // published there (one more dependency of t), and a waiter accepts any generation >= the one it needs
  int32_t ring;
};
constexpr int CHAIN_GEN_BITS = 8;            // low bits of a ring slot: generation t / ring (< 256); the rest: the epoch
// debugging (LPMP_LEVEL_TRACE, engine.cpp): time stamps of the first levels of a level-loop launch, 8 slots per level
__device__ long long* g_level_trace = nullptr;
constexpr int LEVEL_TRACE_MAX = 4000;
__device__ __forceinline__ void level_stamp(int slot) {
  long long* p = g_level_trace;
  if (p && threadIdx.x == 0) { const long long l = p[0]; if (l < LEVEL_TRACE_MAX) p[8 + 8 * l + slot] = (long long)__builtin_amdgcn_s_memrealtime(); }
}
__device__ __forceinline__ void level_stamp_of(int slot, int tid) {   // the same from another thread (the wave that runs ahead: slots 6, 7)
  long long* p = g_level_trace;
  if (p && (int)threadIdx.x == tid) { const long long l = p[0]; if (l < LEVEL_TRACE_MAX) p[8 + 8 * l + slot] = (long long)__builtin_amdgcn_s_memrealtime(); }
}
__device__ __forceinline__ void chain_stamp(const ChainArgs& ca, int ticket, int k) {
  if (ca.trace && threadIdx.x == 0) ca.trace[8 * (int64_t)ticket + k] = (long long)__builtin_amdgcn_s_memrealtime();
}
// one launch (a level x class range of records) as the chain kernels see it: absolute device pointers, so that tickets of
// one persistent launch may come from several schedules (the joined passes of lpmp_compute_pass(n), engine.cpp)
// pad: flags of the level loop (CHAIN_LAUNCH_LABEL_*), or for the joined passes of the dense chain kernel HIST_* | row << 2
struct ChainLaunch { const Op* packets; const UpdRec* recs; const Op* ops; int64_t count; int32_t stride, pad; };
// Which tracked bounds of a launch also go to a row of ChainArgs::lb_hist.  n joined passes are H, W, (K, W)^(n-1), T
// (DESIGN.md 4): the state "after pass i" is never in memory as a whole — K_i holds the last receives of pass i AND the
// first sends of pass i + 1 — but every factor's bound at that moment is known to exactly one record:
//   HIST_END  (a W step)  the updated factor's bound at the end of the record (it is not touched again in this pass)
//   HIST_MID  (a K step)  the updated factor's bound after its receives, before its sends, and the bound of every
//                         pairwise factor it receives from, right after that receive
constexpr int HIST_END = 1, HIST_MID = 2;
// a wait gives up when it has lasted ChainArgs::timeout_ticks (the clock is first read after 1024 polls: short waits never
// read it) or when another wait of the launch has given up
__device__ __forceinline__ bool chain_wait_expired(const ChainArgs& ca, int spins, long long& t0, bool& own) {
  own = false;
  if ((spins & 1023) != 0) return false;
  if (__hip_atomic_load(ca.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
  const long long now = (long long)__builtin_amdgcn_s_memrealtime();
  if (spins == 1024) { t0 = now; return false; }
  own = now - t0 > ca.timeout_ticks;
  return own;
}
// Pause between two polls of a flag or granule.  Constant and short, whatever the age of the wait: in a deep chain the ticket
// that has waited longest is the one next in line, so a pause that grows with the wait lands on the critical path (round 4:
// a back-off to ~4 us after 256 polls made the 512 x 512 8-label Potts grid in row-major order 7.6 instead of 5.8 ms per
// pass, and did nothing for the stalls of a device shared by several processes it was tried against).
__device__ __forceinline__ void chain_poll_pause(int) { __builtin_amdgcn_s_sleep(1); }
// the wait that gives up FIRST says what it was waiting for (abort_flag[1 ...]: engine.cpp puts it into the error message)
__device__ __forceinline__ void chain_abort(const ChainArgs& ca, bool own, int ticket, int dep, int seen, long long t0) {
  if (own && atomicCAS(ca.abort_flag + 1, 0, 1) == 0) {
    const long long now = (long long)__builtin_amdgcn_s_memrealtime();
    ca.abort_flag[2] = ticket; ca.abort_flag[3] = dep; ca.abort_flag[4] = seen; ca.abort_flag[5] = ca.epoch;
    ca.abort_flag[6] = (int)(unsigned)now; ca.abort_flag[7] = (int)(now >> 32); ca.abort_flag[8] = (int)(unsigned)t0; ca.abort_flag[9] = (int)(t0 >> 32);
    ca.abort_flag[10] = __hip_atomic_load(ca.next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __hip_atomic_store(ca.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// template ticket and copy of a ticket of a periodic launch (ChainArgs::per_*)
struct TicketRef { int idx, copy; };
__device__ __forceinline__ TicketRef chain_ticket_ref(const ChainArgs& ca, int ticket) {
  if (ca.per_len == 0 || ticket < ca.per_begin) return {ticket, 0};
  const int q = min((ticket - ca.per_begin) / ca.per_len, ca.per_count - 1);
  return {ticket - q * ca.per_len, q};
}
__device__ __forceinline__ bool chain_flag_set(const ChainArgs& ca, int dep_ticket) {
  if (ca.ring == 0) return __hip_atomic_load(ca.done + dep_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ca.epoch;
  const int v = __hip_atomic_load(ca.done + dep_ticket % ca.ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (v >> CHAIN_GEN_BITS) == ca.epoch && (v & ((1 << CHAIN_GEN_BITS) - 1)) >= dep_ticket / ca.ring;
}
// all threads of the workgroup; returns false when the run was aborted
__device__ __forceinline__ bool chain_wait(const ChainArgs& ca, int ticket) {
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  const TicketRef tr = chain_ticket_ref(ca, ticket);
  const int b = ca.dep_off[tr.idx], e = ca.dep_off[tr.idx + 1];
  const int shift = tr.copy * ca.per_len;
  // ring: the slot this ticket will publish into must hold its previous occupant's completion (ticket - ring)
  const int extra = (ca.ring > 0 && ticket >= ca.ring) ? 1 : 0;
  for (int i = b + (int)threadIdx.x; i < e + extra; i += (int)blockDim.x) {
    const int d = i < e ? ca.dep[i] + shift : ticket - ca.ring;
    int spins = 0; long long t0 = 0; bool own;
    while (!chain_flag_set(ca, d)) {
      chain_poll_pause(spins);
      if (chain_wait_expired(ca, ++spins, t0, own)) {
        chain_abort(ca, own, ticket, d, __hip_atomic_load(ca.done + (ca.ring ? d % ca.ring : d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), t0);
        s_bad = 1;
        break;
      }
    }
  }
  __syncthreads();
  chain_stamp(ca, ticket, 1);                      // predecessors seen
  return s_bad == 0;
}
// all threads: every wave drains its stores, then one lane publishes the ticket.
// MEMORY-MODEL INVARIANT (why relaxed flags suffice, and what must stay true): the hand-over is not a release / acquire
// pair — at agent scope that would write back and invalidate the whole L2 of the XCD per ticket — but rests on three
// properties of the accesses themselves: (1) every dual load / store and every tracked-bound store of a chain body is an
// agent-scope atomic (sc1: stores write through to memory, loads do not hit a possibly stale L2 / L1 line) — enforced by
// the static_asserts `!CHAIN || A == ACC_COH` in the bodies; (2) a wave's stores have left the CU when its vmcnt reaches
// 0 (`s_waitcnt vmcnt(0)` below, then the workgroup barrier, then the flag store); (3) the waiting side reads duals only
// after it has seen the flag (chain_wait returns, then load_own_and_targets).  Data that is NOT accessed this way must
// not cross tickets: rounded labels (store_label) are written and read by the record of the SAME factor only, pairwise
// tables and packets are constants.
__device__ __forceinline__ void chain_publish(const ChainArgs& ca, int ticket) {
  chain_stamp(ca, ticket, 2);                      // body done, stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ca.ring == 0) __hip_atomic_store(ca.done + ticket, ca.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(ca.done + ticket % ca.ring, (ca.epoch << CHAIN_GEN_BITS) | (ticket / ca.ring), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  chain_stamp(ca, ticket, 3);                      // published
}


// Mailbox of a dense chain (plan.cpp decides which vectors travel this way): the value a send has just computed, as two
// self-validating 8-byte granules {half of the double, epoch of the running launch}.  An aligned 8-byte access is atomic,
// so a granule whose tag is this launch's epoch IS the producer's value — no ordering with any other store is needed, and
// the consumer has the value after ONE trip instead of two (completion flag seen, then the vector fetched).  What the
// consumer may conclude from a granule is only this value: the producer's other stores are not yet visible, which is why
// plan.cpp keeps a flag dependency wherever anything else of the producer is read or overwritten.
__device__ __forceinline__ void mailbox_put(unsigned long long* q, double v, int epoch) {
  const unsigned long long tag = (unsigned long long)(unsigned)epoch << 32;
  __hip_atomic_store(q, tag | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, tag | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// experiments (tools/build_variant.sh): polls of one granule pair in flight, spaced LPMP_MBOX_GAP sleeps apart
#ifndef LPMP_MBOX_PIPE
#define LPMP_MBOX_PIPE 1
#endif
#ifndef LPMP_MBOX_GAP
#define LPMP_MBOX_GAP 5
#endif
#ifndef LPMP_MBOX_SLEEP
#define LPMP_MBOX_SLEEP 1
#endif
__device__ __forceinline__ double mailbox_take(const ChainArgs& ca, const unsigned long long* q, bool& bad) {
  const unsigned tag = (unsigned)ca.epoch;
#if LPMP_MBOX_PIPE > 1
  // a poll is a round trip to the far side of the fabric; with several under way, a fraction of a trip apart, the granule is
  // seen that much sooner after it lands (the loads of a wave return in order)
  unsigned long long a[LPMP_MBOX_PIPE], b[LPMP_MBOX_PIPE];
#pragma unroll
  for (int i = 0; i < LPMP_MBOX_PIPE; ++i) {
    a[i] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b[i] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i + 1 < LPMP_MBOX_PIPE) __builtin_amdgcn_s_sleep(LPMP_MBOX_GAP);
  }
  long long t0 = 0;
  for (int spins = 0;;) {
    if ((unsigned)(a[0] >> 32) == tag && (unsigned)(b[0] >> 32) == tag) return __hiloint2double((int)(unsigned)b[0], (int)(unsigned)a[0]);
#pragma unroll
    for (int i = 0; i + 1 < LPMP_MBOX_PIPE; ++i) { a[i] = a[i + 1]; b[i] = b[i + 1]; }
    a[LPMP_MBOX_PIPE - 1] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b[LPMP_MBOX_PIPE - 1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  long long t0 = 0;
  for (int spins = 0;;) {
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag) return __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
#if LPMP_MBOX_SLEEP == 1
    chain_poll_pause(spins);
#else
    __builtin_amdgcn_s_sleep(LPMP_MBOX_SLEEP);
#endif
#endif
    bool own;
    if (chain_wait_expired(ca, ++spins, t0, own)) {
      chain_abort(ca, own, -1, -2, (int)(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32), t0);   // a mailbox granule
      bad = true;
      return 0.0;
    }
  }
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmin(v, shfl_xor_f64(v, m));
  return v;
}


// min all-reduce over aligned groups of CL lanes (CL <= 16) with DPP moves: no LDS-crossbar round trip.
// quad_perm [1,0,3,2] and [2,3,0,1] pair lanes inside a quad, row_half_mirror / row_mirror pair quads and
// half rows; for a min it only matters that every step pairs the two halves that are still missing.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CL>
__device__ __forceinline__ double row_allreduce_min(double v) {
  static_assert(CL == 2 || CL == 4 || CL == 8 || CL == 16, "row width");
  v = fmin(v, dpp_mov_f64<0xB1>(v));                       // lane ^ 1
  if constexpr (CL >= 4) v = fmin(v, dpp_mov_f64<0x4E>(v));   // lane ^ 2
  if constexpr (CL >= 8) v = fmin(v, dpp_mov_f64<0x141>(v));  // row_half_mirror: other quad of the 8
  if constexpr (CL >= 16) v = fmin(v, dpp_mov_f64<0x140>(v)); // row_mirror: other half of the 16
  return v;
}
template <int G, int L>
__device__ __forceinline__ double vec_min(double v) {   // min over the first L lanes of a G-lane group (others pass +inf)
  constexpr int W = L < G ? L : G;
  v = row_allreduce_min<(W < 16 ? W : 16)>(v);
  if constexpr (W > 16) v = fmin(v, shfl_xor_f64(v, 16));
  if constexpr (W > 32) v = fmin(v, shfl_xor_f64(v, 32));
  return v;
}

// MaximizePotentialAndComputePrimal of a vector factor held one element per lane (first `valid` lanes of a G-lane
// group, padded width W): index of the FIRST minimum, like std::min_element (reference test/test_model.hxx:27-33)
template <int G, int W>
__device__ __forceinline__ int group_argmin(double v, bool valid, int g) {
  const double x = valid ? v : LPMP_INF;
  const double mn = vec_min<G, W>(x);
  const unsigned long long hit = __ballot(valid && x == mn);
  const int base = (int)(threadIdx.x & 63) - g;
  const unsigned long long gm = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << base;
  return __ffsll((long long)(hit & gm)) - 1 - base;
}
// the label is only taken when the factor has none yet (primal_ unset = label count)
__device__ __forceinline__ void store_label(int32_t* __restrict__ primal, int factor, int n_labels, int label) {
  int32_t* pr = primal + 2 * (int64_t)factor;
  if (*pr >= n_labels) *pr = label;
}

// -------------------------------------------------------------------------------------------------
// Generic kernel: any factor kind, any message kind, either role.  Two shapes of the same code:
//   G = 64: one wave per updated factor, state (own duals, snapshot, delta) in LDS, up to GEN_MAXD doubles;
//   G = 1:  one LANE per updated factor for tiny factors (every dim <= SMALL_MAXD: labeling-list edge / triplet
//           factors, two-label toy factors, small pairwise factors that are themselves updated) — 64 factors per
//           wave, no cross-lane step at all, the per-lane state interleaved in LDS ([element][lane]: conflict free).
// -------------------------------------------------------------------------------------------------
constexpr int SMALL_WAVES = 1;
template <int G> struct GenCtx;
template <> struct GenCtx<64> {
  static constexpr int STRIDE = 64, FPB = GEN_WAVES, THREADS = 64 * GEN_WAVES;
  static constexpr int MAX_ADAPTIVE_SENDS = GEN_ADAPTIVE_SENDS;
  struct Lds { double own[GEN_MAXD]; double snap[GEN_MAXD]; double dl[GEN_MAXD]; double imp[GEN_ADAPTIVE_SENDS]; };
  Lds& s; const int lane;
  __device__ __forceinline__ double& imp(int i) const { return s.imp[i]; }
  __device__ __forceinline__ double& own(int i) const { return s.own[i]; }
  __device__ __forceinline__ double& snap(int i) const { return s.snap[i]; }
  __device__ __forceinline__ double& dl(int i) const { return s.dl[i]; }
  __device__ __forceinline__ int first() const { return lane; }
  __device__ __forceinline__ bool leader() const { return lane == 0; }
  __device__ __forceinline__ static double gmin(double v) { return wave_min(v); }
  __device__ __forceinline__ static int gmin(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
  }
  __device__ __forceinline__ static void sync() { wave_sync(); }
};
template <> struct GenCtx<1> {
  static constexpr int STRIDE = 1, FPB = 64 * SMALL_WAVES, THREADS = 64 * SMALL_WAVES;
  static constexpr int MAX_ADAPTIVE_SENDS = SMALL_MAXD;
  struct Lds { double own[SMALL_MAXD][64]; double snap[SMALL_MAXD][64]; double dl[SMALL_MAXD][64]; double imp[SMALL_MAXD][64]; };
  Lds& s; const int lane;
  __device__ __forceinline__ double& imp(int i) const { return s.imp[i][lane]; }
  __device__ __forceinline__ double& own(int i) const { return s.own[i][lane]; }
  __device__ __forceinline__ double& snap(int i) const { return s.snap[i][lane]; }
  __device__ __forceinline__ double& dl(int i) const { return s.dl[i][lane]; }
  __device__ __forceinline__ int first() const { return 0; }
  __device__ __forceinline__ bool leader() const { return true; }
  __device__ __forceinline__ static double gmin(double v) { return v; }
  __device__ __forceinline__ static int gmin(int v) { return v; }
  __device__ __forceinline__ static void sync() {}
};

// pairwise cost T(a,b) of a pairwise factor (dense table or Potts scalar)
__device__ __forceinline__ double pw_cost(const double* __restrict__ cdata, int64_t coff, int kind, int d1, int a, int b) {
  if (kind == LPMP_F_PAIRWISE_DENSE) return cdata[coff + (int64_t)a * d1 + b];
  return a == b ? 0.0 : cdata[coff];
}

// dl[x] = omega * (m_s[x] + min_y (T + m_o[y])) for side s of a pairwise factor whose message vectors are
// read through m (global: live peer, or LDS: own snapshot / live own state)
template <class C, class Src>
__device__ __forceinline__ void pw_min_marginal(const C& c, const double* __restrict__ cdata, int64_t coff, int kind, int d0, int d1,
                                                Src m, int side, double omega) {
  if (side == 0) {
    for (int a = 0; a < d0; ++a) {
      double v = LPMP_INF;
      for (int b = c.first(); b < d1; b += C::STRIDE) v = fmin(v, pw_cost(cdata, coff, kind, d1, a, b) + m(d0 + b));
      v = C::gmin(v);
      if (c.leader()) c.dl(a) = omega * (m(a) + v);
    }
  } else {
    for (int b = c.first(); b < d1; b += C::STRIDE) {
      double v = LPMP_INF;
      for (int a = 0; a < d0; ++a) v = fmin(v, pw_cost(cdata, coff, kind, d1, a, b) + m(a));
      c.dl(b) = omega * (m(d0 + b) + v);
    }
  }
  C::sync();
}

// labeling_message::compute_msg (reference labeling_list_factor.hxx:411-443): dl[l] = omega*(min_{r:tab[r]==l} R[r] - not_taken)
template <class C, class Src>
__device__ __forceinline__ void labeling_to_left(const C& c, Src R, int nr, const int32_t* __restrict__ tab, int nl, int implicit_origin,
                                                 double omega) {
  double nt = implicit_origin ? 0.0 : LPMP_INF;
  for (int r = 0; r < nr; ++r) if (tab[r] >= nl) nt = fmin(nt, R(r));
  for (int l = c.first(); l < nl; l += C::STRIDE) {
    double v = LPMP_INF;
    for (int r = 0; r < nr; ++r) if (tab[r] == l) v = fmin(v, R(r));
    c.dl(l) = omega * (v - nt);
  }
  C::sync();
}

template <class C, class Src>
__device__ __forceinline__ void minnorm_delta(const C& c, Src src, int n, double omega) {
  double mn = LPMP_INF;
  for (int i = c.first(); i < n; i += C::STRIDE) mn = fmin(mn, src(i));
  mn = C::gmin(mn);
  for (int i = c.first(); i < n; i += C::STRIDE) c.dl(i) = omega * (src(i) - mn);
  C::sync();
}

// FactorContainer::LowerBound of a factor seen through an accessor (its duals as they would be after a send):
// vector factor min_i d(i), clamped at 0 with an implicit origin; pairwise min_a ( m(a) + min_b (T[a][b] + m(d0 + b)) )
template <class C, class Acc>
__device__ __forceinline__ double vec_lb_through(const C& c, int n, Acc d, int implicit_origin) {
  double v = LPMP_INF;
  for (int i = c.first(); i < n; i += C::STRIDE) v = fmin(v, d(i));
  v = C::gmin(v);
  if (implicit_origin && 0.0 < v) v = 0.0;
  return v;
}
template <class C, class Acc>
__device__ __forceinline__ double pw_lb_through(const C& c, const double* __restrict__ cdata, int64_t coff, int kind, int d0, int d1, Acc m) {
  double best = LPMP_INF;
  for (int a = 0; a < d0; ++a) {
    double v = LPMP_INF;
    for (int b = c.first(); b < d1; b += C::STRIDE) v = fmin(v, pw_cost(cdata, coff, kind, d1, a, b) + m(d0 + b));
    v = C::gmin(v);
    best = fmin(best, m(a) + v);
  }
  return best;
}

// A: access policy of the duals (ACC_COH inside the chain executor)
template <int G, int A>
__device__ __forceinline__ void generic_body(const UpdRec* __restrict__ recs, const Op* __restrict__ ops, double* __restrict__ dual,
                                             const double* __restrict__ cdata, const int32_t* __restrict__ tabs, double* __restrict__ lb,
                                             int32_t* __restrict__ primal, const int32_t* __restrict__ pw_unary, int64_t first, int64_t count,
                                             int flags, int64_t block) {
  using C = GenCtx<G>;
  __shared__ typename C::Lds lds[G == 64 ? GEN_WAVES : SMALL_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t idx = block * C::FPB + (G == 64 ? wave : (int)threadIdx.x);
  if (idx >= count) return;
  const C c{lds[wave], lane};
  const UpdRec rec = recs[first + idx];
  const int okind = rec.kind_flags & 15;
  const int on = okind == LPMP_F_VECTOR ? rec.d0 : rec.d0 + rec.d1;   // own dual size
  double* own_g = dual + rec.dual_off;
  for (int i = c.first(); i < on; i += C::STRIDE) c.own(i) = ld_dual<A>(own_g + i);
  C::sync();
  if constexpr (G == 1 && A == ACC_WG) { asm volatile("" :: "v"(c.own(0))); level_stamp(1); }

  const int n_ops = rec.n_recv + rec.n_send;
  // delta of one message into c.dl: computed from the peer (receive) or from the snapshot / live own state (send)
  auto compute_delta = [&](const Op& op, const bool recv, const bool live_src, const double omega) {
    const int code = op.info & 15, role = (op.info >> 4) & 1, side = (op.info >> 5) & 1, imp = (op.info >> 6) & 1;
    const int pkind = (op.info >> 8) & 15;
    const double* peer = dual + op.peer_dual;
    const int len = op.len;
    auto from_peer = [&](int i) { return ld_dual<A>(peer + i); };
    auto from_own = [&](int i) { return live_src ? c.own(i) : c.snap(i); };
    const bool by_right = recv ? (role == 0) : (role == 1);
    if (code == OP_UP) {
      if (by_right) {   // min-marginal of the pairwise (right) factor
        if (recv) pw_min_marginal(c, cdata, op.peer_const, pkind, op.pd0, op.pd1, from_peer, side, omega);
        else pw_min_marginal(c, cdata, rec.const_off, okind, rec.d0, rec.d1, from_own, side, omega);
      } else {          // omega * theta of the unary (left) factor
        for (int i = c.first(); i < len; i += C::STRIDE) c.dl(i) = omega * (recv ? ld_dual<A>(peer + i) : from_own(i));
        C::sync();
      }
    } else if (code == OP_LABELING) {
      const int32_t* tab = tabs + op.peer_const;
      if (by_right) {
        if (recv) labeling_to_left(c, from_peer, op.pd0, tab, op.pd1, imp, omega);
        else labeling_to_left(c, from_own, rec.d0, tab, op.pd1, imp, omega);
      } else {
        for (int i = c.first(); i < len; i += C::STRIDE) c.dl(i) = omega * (recv ? ld_dual<A>(peer + i) : from_own(i));
        C::sync();
      }
    } else {            // OP_MINNORM
      if (recv) minnorm_delta(c, from_peer, len, omega);
      else minnorm_delta(c, from_own, len, omega);
    }
  };
  // one receive or send: compute delta, then +delta to the side that did not compute it and -delta to the side that did
  auto run_op = [&](const Op& op, const bool recv, const bool live_src, const double omega) {
    if (c.leader()) st_lb<A>(lb + op.peer, LPMP_NAN);
    const int code = op.info & 15, role = (op.info >> 4) & 1, side = (op.info >> 5) & 1, imp = (op.info >> 6) & 1;
    double* peer = dual + op.peer_dual;
    const int len = op.len;
    [[maybe_unused]] auto from_own = [&](int i) { return live_src ? c.own(i) : c.snap(i); };
    // the message is computed by the peer for a receive and by the updated factor for a send
    const bool by_right = recv ? (role == 0) : (role == 1);
    if constexpr (G == 1) {
      // labeling message with the updated factor on the left (multicut edge variable <-> triplet / quadruple factor),
      // lane-per-factor form: deep schedules of such factors are bound by the chain of dependent loads of ONE
      // update, so the match table and the peer's costs are requested together, unconditionally, as soon as the op
      // is known (2 round trips per op instead of 4) and the reparametrisation reuses the loaded costs.  Same
      // arithmetic as the general path below.
      if (code == OP_LABELING && role == 0) {
        const int32_t* tab = tabs + op.peer_const;
        const int nr = op.pd0, nl = op.pd1;
        int tv[SMALL_MAXD]; double Rv[SMALL_MAXD];
#pragma unroll
        for (int r = 0; r < SMALL_MAXD; ++r) { const bool in = r < nr; tv[r] = in ? tab[r] : nl; Rv[r] = in ? ld_dual<A>(peer + r) : LPMP_INF; }
        if (recv) {
          double nt = imp ? 0.0 : LPMP_INF;
#pragma unroll
          for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] >= nl) nt = fmin(nt, Rv[r]);   // entries beyond nr hold +inf
          for (int l = 0; l < nl; ++l) {
            double v = LPMP_INF;
#pragma unroll
            for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] == l) v = fmin(v, Rv[r]);
            c.dl(l) = omega * (v - nt);
          }
          for (int i = 0; i < len; ++i) c.own(i) += +1.0 * c.dl(i);
#pragma unroll
          for (int r = 0; r < SMALL_MAXD; ++r) if (r < nr && tv[r] < nl) st_dual<A>(peer + r, Rv[r] + -1.0 * c.dl(tv[r]));
        } else {
          for (int i = 0; i < len; ++i) c.dl(i) = omega * from_own(i);
          for (int i = 0; i < len; ++i) c.own(i) += -1.0 * c.dl(i);
#pragma unroll
          for (int r = 0; r < SMALL_MAXD; ++r) if (r < nr && tv[r] < nl) st_dual<A>(peer + r, Rv[r] + +1.0 * c.dl(tv[r]));
        }
        return;
      }
    }
    compute_delta(op, recv, live_src, omega);
    // (reference MessageContainerView::operator-=, factors_messages.hxx:495-508)
    const bool own_is_left = role == 0;
    const double s_left = by_right ? +1.0 : -1.0, s_right = -s_left;
    if (own_is_left) { for (int i = c.first(); i < len; i += C::STRIDE) c.own(i) += s_left * c.dl(i); }
    else { for (int i = c.first(); i < len; i += C::STRIDE) st_dual<A>(peer + i, ld_dual<A>(peer + i) + s_left * c.dl(i)); }
    if (code == OP_UP) {
      if (own_is_left) { double* m = peer + (side == 0 ? 0 : op.pd0); for (int i = c.first(); i < len; i += C::STRIDE) st_dual<A>(m + i, ld_dual<A>(m + i) + s_right * c.dl(i)); }
      else { const int o = side == 0 ? 0 : rec.d0; for (int i = c.first(); i < len; i += C::STRIDE) c.own(o + i) += s_right * c.dl(i); }
    } else if (code == OP_LABELING) {
      const int32_t* tab = tabs + op.peer_const;
      const int nl = op.pd1;
      if (own_is_left) { for (int r = c.first(); r < op.pd0; r += C::STRIDE) if (tab[r] < nl) st_dual<A>(peer + r, ld_dual<A>(peer + r) + s_right * c.dl(tab[r])); }
      else { for (int r = c.first(); r < rec.d0; r += C::STRIDE) if (tab[r] < nl) c.own(r) += s_right * c.dl(tab[r]); }
    } else {
      if (own_is_left) { for (int i = c.first(); i < len; i += C::STRIDE) st_dual<A>(peer + i, ld_dual<A>(peer + i) + s_right * c.dl(i)); }
      else { for (int i = c.first(); i < len; i += C::STRIDE) c.own(i) += s_right * c.dl(i); }
    }
    C::sync();
  };
  // The device op behind LPMP_MF_IMPROVEMENT (send_message_to_{left,right}_improvement of the message op, reference
  // factors_messages.hxx:734-747, :795-808): the exact change of LowerBound(left) + LowerBound(right) a weight-1 send
  // of this message from the factor's state after its receives would cause.  Nothing is written.
  auto improvement = [&](const Op& op) -> double {
    const int code = op.info & 15, role = (op.info >> 4) & 1, side = (op.info >> 5) & 1, pkind = (op.info >> 8) & 15;
    const double* peer = dual + op.peer_dual;
    const int len = op.len;
    const bool by_right = role == 1, own_is_left = role == 0;
    const double s_left = by_right ? +1.0 : -1.0, s_right = -s_left;
    compute_delta(op, false, true, 1.0);
    const int32_t* tab = code == OP_LABELING ? tabs + op.peer_const : nullptr;
    const int nl = op.pd1;
    const int own_io = (rec.kind_flags >> 4) & LPMP_FF_IMPLICIT_ORIGIN, peer_io = (op.info >> 7) & 1;
    double before, after;
    if (own_is_left) {          // left = this factor (vector), right = peer
      const double lb_l = vec_lb_through(c, on, [&](int i) { return c.own(i); }, own_io);
      const double la_l = vec_lb_through(c, on, [&](int i) { return i < len ? c.own(i) + s_left * c.dl(i) : c.own(i); }, own_io);
      double lb_r, la_r;
      if (code == OP_UP) {
        const int o = side == 0 ? 0 : op.pd0;
        lb_r = pw_lb_through(c, cdata, op.peer_const, pkind, op.pd0, op.pd1, [&](int j) { return ld_dual<A>(peer + j); });
        la_r = pw_lb_through(c, cdata, op.peer_const, pkind, op.pd0, op.pd1, [&](int j) { return (j >= o && j < o + len) ? ld_dual<A>(peer + j) + s_right * c.dl(j - o) : ld_dual<A>(peer + j); });
      } else if (code == OP_LABELING) {
        lb_r = vec_lb_through(c, op.pd0, [&](int j) { return ld_dual<A>(peer + j); }, peer_io);
        la_r = vec_lb_through(c, op.pd0, [&](int j) { return tab[j] < nl ? ld_dual<A>(peer + j) + s_right * c.dl(tab[j]) : ld_dual<A>(peer + j); }, peer_io);
      } else {
        lb_r = vec_lb_through(c, op.pd0, [&](int j) { return ld_dual<A>(peer + j); }, peer_io);
        la_r = vec_lb_through(c, op.pd0, [&](int j) { return ld_dual<A>(peer + j) + s_right * c.dl(j); }, peer_io);
      }
      before = lb_l + lb_r; after = la_l + la_r;
    } else {                    // left = peer (vector), right = this factor
      const double lb_l = vec_lb_through(c, op.pd0, [&](int i) { return ld_dual<A>(peer + i); }, peer_io);
      const double la_l = vec_lb_through(c, op.pd0, [&](int i) { return i < len ? ld_dual<A>(peer + i) + s_left * c.dl(i) : ld_dual<A>(peer + i); }, peer_io);
      double lb_r, la_r;
      if (code == OP_UP) {
        const int o = side == 0 ? 0 : rec.d0;
        lb_r = pw_lb_through(c, cdata, rec.const_off, okind, rec.d0, rec.d1, [&](int j) { return c.own(j); });
        la_r = pw_lb_through(c, cdata, rec.const_off, okind, rec.d0, rec.d1, [&](int j) { return (j >= o && j < o + len) ? c.own(j) + s_right * c.dl(j - o) : c.own(j); });
      } else if (code == OP_LABELING) {
        lb_r = vec_lb_through(c, on, [&](int j) { return c.own(j); }, own_io);
        la_r = vec_lb_through(c, on, [&](int j) { return tab[j] < nl ? c.own(j) + s_right * c.dl(tab[j]) : c.own(j); }, own_io);
      } else {
        lb_r = vec_lb_through(c, on, [&](int j) { return c.own(j); }, own_io);
        la_r = vec_lb_through(c, on, [&](int j) { return c.own(j) + s_right * c.dl(j); }, own_io);
      }
      before = lb_l + lb_r; after = la_l + la_r;
    }
    C::sync();
    return fabs(after - before);
  };
  // MaximizePotentialAndComputePrimal between the receives and the sends (vector factors of a COMPUTE_PRIMAL type)
  auto round_label = [&]() {
    if (!((flags & SWEEP_PRIMAL) && (rec.kind_flags & UPD_PRIMAL))) return;
    if (okind != LPMP_F_VECTOR) {
      // a pairwise factor that rounds itself (engine.cpp, ensure_primal): a side is given when its unary holds a
      // label (else when the factor's own slot does: a side without a unary), the free sides take the first minimiser
      // of T[a][b] + m1[a] + m2[b] in row-major order; the unaries of the filled sides are labelled
      if (!pw_unary) return;
      const int d0 = rec.d0, d1 = rec.d1;
      int32_t* pr = primal + 2 * (int64_t)rec.factor;
      const int u0 = pw_unary[2 * (int64_t)rec.factor], u1 = pw_unary[2 * (int64_t)rec.factor + 1];
      int x0 = u0 >= 0 ? primal[2 * (int64_t)u0] : pr[0], x1 = u1 >= 0 ? primal[2 * (int64_t)u1] : pr[1];
      const bool free0 = x0 >= d0, free1 = x1 >= d1;
      if (free0 || free1) {
        const int a0 = free0 ? 0 : x0, na = free0 ? d0 : 1, b0 = free1 ? 0 : x1, nb = free1 ? d1 : 1;
        double bv = LPMP_INF; int bi = 0x7fffffff;
        for (int i = c.first(); i < na * nb; i += C::STRIDE) {
          const int a = a0 + i / nb, b = b0 + i % nb;
          const double v = pw_cost(cdata, rec.const_off, okind, d1, a, b) + c.own(a) + c.own(d0 + b);
          if (bi == 0x7fffffff || v < bv) { bv = v; bi = i; }
        }
        const double mn = C::gmin(bv);
        const int cand = C::gmin((bi != 0x7fffffff && bv == mn) ? bi : 0x7fffffff);
        x0 = a0 + cand / nb; x1 = b0 + cand % nb;
      }
      if (c.leader()) {
        pr[0] = x0; pr[1] = x1;
        if (free0 && u0 >= 0) primal[2 * (int64_t)u0] = x0;
        if (free1 && u1 >= 0) primal[2 * (int64_t)u1] = x1;
      }
      return;
    }
    double bv = LPMP_INF; int bi = 0x7fffffff;
    for (int i = c.first(); i < on; i += C::STRIDE) { const double v = c.own(i); if (bi == 0x7fffffff || v < bv) { bv = v; bi = i; } }
    const double mn = C::gmin(bv);
    const int cand = C::gmin((bi != 0x7fffffff && bv == mn) ? bi : 0x7fffffff);
    if (c.leader()) store_label(primal, rec.factor, on, cand);
  };
  Op nxt{};
  if (n_ops > 0) nxt = ops[rec.op_begin];
  for (int k = 0; k < n_ops; ++k) {
    if (k == rec.n_recv) {   // state after the receives: what every shared send is computed from
      if constexpr (G == 1 && A == ACC_WG) level_stamp(2);
      round_label();
      for (int i = c.first(); i < on; i += C::STRIDE) c.snap(i) = c.own(i);
      C::sync();
    }
    if (k == rec.n_recv && (flags & SWEEP_ADAPTIVE)) break;   // the sends of the adaptive rule follow below
    const Op op = nxt;
    if (k + 1 < n_ops) nxt = ops[rec.op_begin + k + 1];   // requested before this op's dependent chain starts
    run_op(op, k < rec.n_recv, false, op.omega);
  }
  if (rec.n_recv == n_ops) round_label();
  if ((flags & SWEEP_ADAPTIVE) && rec.n_send > 0) {
    // send_messages_with_adaptive_weights (reference factors_messages.hxx:2860-2926, non-batch branch): the dual
    // improvement of every active message for weight 1, then adaptive_weight_rescaling (:2846-2857):
    // w = omega / 2 + omega_sum / 2 * improvement / improvement_sum if that sum is positive, else the improvements
    // themselves (all 0: nothing is sent); then SendMessages(w) from the state after the receives.
    // (inactive entries have weight and improvement 0: they change neither sum)
    for (int k = 0; k < rec.n_send; ++k) {
      const Op op = ops[rec.op_begin + rec.n_recv + k];
      const double v = (op.info & OP_HAS_IMPROVEMENT) ? improvement(op) : 0.0;
      if (c.leader()) c.imp(k) = v;
    }
    C::sync();
    double isum = 0.0, osum = 0.0;
    for (int k = 0; k < rec.n_send; ++k) { isum += c.imp(k); osum += ops[rec.op_begin + rec.n_recv + k].omega; }
    if (isum > 0) {
      for (int k = 0; k < rec.n_send; ++k) {
        const Op op = ops[rec.op_begin + rec.n_recv + k];
        const double w = 0.5 * op.omega + 0.5 * osum * c.imp(k) / isum;
        if (w != 0.0) run_op(op, false, false, w);
      }
    }
  }
  if (flags & SWEEP_RESIDUAL) {   // reference send_messages_residual, factors_messages.hxx:2960-3007
    double residual = 0.0;
    for (int k = rec.n_recv; k < n_ops; ++k) {
      const Op op = ops[rec.op_begin + k];
      residual += op.omega;
      run_op(op, false, true, residual);
    }
  }
  if constexpr (G == 1 && A == ACC_WG) level_stamp(3);
  if (c.leader()) st_lb<A>(lb + rec.factor, LPMP_NAN);
  for (int i = c.first(); i < on; i += C::STRIDE) st_dual<A>(own_g + i, c.own(i));
}

template <int G>
__global__ void __launch_bounds__(GenCtx<G>::THREADS)
sweep_generic_kernel(const UpdRec* __restrict__ recs, const Op* __restrict__ ops, double* __restrict__ dual,
                     const double* __restrict__ cdata, const int32_t* __restrict__ tabs, double* __restrict__ lb,
                     int32_t* __restrict__ primal, const int32_t* __restrict__ pw_unary, int64_t first, int64_t count, int flags) {
  generic_body<G, ACC_PLAIN>(recs, ops, dual, cdata, tabs, lb, primal, pw_unary, first, count, flags, (int64_t)blockIdx.x);
}

// -------------------------------------------------------------------------------------------------
// Dense fast path: unary simplex factor with L labels whose active messages all go to dense L x L
// pairwise factors.  G lanes per factor; each lane keeps NL double2 of the table in registers, loaded
// as one coalesced 16-B-per-lane stream (2G doubles per load step = RPL rows).
//   lane g: column pair c2 = g % (L/2) (columns 2*c2, 2*c2+1), row-in-step rl = g / (L/2)
//   element of load step i: row = i*RPL + rl
// side 0 (own label = row a):    q[a] = min_b T[a][b] + m2[b]  -> reduce over the L/2 lanes of a row
// side 1 (own label = column b): q[b] = min_a T[a][b] + m1[a]  -> local over steps, reduce over the RPL row-lanes
// Vectors (theta, m1, m2, delta) live one element per lane (lane g < L) and are transposed through LDS.
// -------------------------------------------------------------------------------------------------
template <int L> struct DenseCfg;
template <> struct DenseCfg<32> { static constexpr int G = 64; };
template <> struct DenseCfg<16> { static constexpr int G = 16; };
template <> struct DenseCfg<8>  { static constexpr int G = 8; };
template <> struct DenseCfg<4>  { static constexpr int G = 4; };

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int L>
__global__ void __launch_bounds__(256)
sweep_dense_kernel(const UpdRec* __restrict__ recs, const Op* __restrict__ ops, double* __restrict__ dual,
                   const double* __restrict__ cdata, double* __restrict__ lb, int32_t* __restrict__ primal,
                   int64_t first, int64_t count, int flags) {
  constexpr int G = DenseCfg<L>::G;
  constexpr int CL = L / 2;            // lanes per table row
  constexpr int RPL = 2 * G / L;       // rows per load step
  constexpr int NL = L / RPL;          // load steps
  constexpr int GPB = 256 / G;         // groups (factors) per block
  static_assert(G >= L / 2 && (2 * G) % L == 0 && L % RPL == 0, "bad dense config");
  // LDS: per group  mo[L] (other-side vector) + q[L] (min result), one group-private slab
  __shared__ double lds_mo[GPB][L];
  __shared__ double lds_q[GPB][L];
  const int grp = threadIdx.x / G, g = threadIdx.x % G;
  const int64_t idx = (int64_t)blockIdx.x * GPB + grp;
  const bool live = idx < count;
  const int c2 = g % CL, rl = g / CL;
  UpdRec rec;
  if (live) rec = recs[first + idx]; else { rec.n_recv = 0; rec.n_send = 0; rec.dual_off = 0; rec.op_begin = 0; }
  double* own_g = dual + rec.dual_off;
  const bool vl = live && g < L;        // this lane holds vector element g
  double theta = vl ? own_g[g] : 0.0;
  // every lane of a wave must run the same number of iterations (cross-lane ops inside)
  int n_recv = rec.n_recv;
  int max_recv = n_recv;
  if (G < 64) {
#pragma unroll
    for (int m = 32; m >= G; m >>= 1) max_recv = max(max_recv, __shfl_xor(max_recv, m, 64));
  }
  for (int k = 0; k < max_recv; ++k) {
    const bool act = k < n_recv;
    Op op;
    if (act) op = ops[rec.op_begin + k]; else { op.peer_dual = rec.dual_off; op.peer_const = 0; op.info = 0; }
    const int side = (op.info >> 5) & 1;
    const double* T = cdata + op.peer_const;
    double* ms = dual + op.peer_dual + (side == 0 ? 0 : L);
    const double* mo = dual + op.peer_dual + (side == 0 ? L : 0);
    double2_t t[NL];
    if (act) {
#pragma unroll
      for (int i = 0; i < NL; ++i) t[i] = *reinterpret_cast<const double2_t*>(T + (int64_t)i * 2 * G + 2 * g);
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) t[i] = double2_t{0.0, 0.0};
    }
    const double ms_v = (act && g < L) ? ms[g] : 0.0;
    const double mo_v = (act && g < L) ? mo[g] : 0.0;
    if (g < L) lds_mo[grp][g] = mo_v;
    wave_sync();
    if (side == 0) {
      // lane needs m2[2*c2], m2[2*c2+1]
      const double2_t mv = *reinterpret_cast<const double2_t*>(&lds_mo[grp][2 * c2]);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        double v = fmin(t[i].x + mv.x, t[i].y + mv.y);
        v = row_allreduce_min<CL>(v);
        if (c2 == 0) lds_q[grp][i * RPL + rl] = v;
      }
    } else {
      double vx = LPMP_INF, vy = LPMP_INF;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const double m1v = lds_mo[grp][i * RPL + rl];
        vx = fmin(vx, t[i].x + m1v);
        vy = fmin(vy, t[i].y + m1v);
      }
#pragma unroll
      for (int m = G / 2; m >= CL; m >>= 1) { vx = fmin(vx, shfl_xor_f64(vx, m)); vy = fmin(vy, shfl_xor_f64(vy, m)); }
      if (rl == 0) { lds_q[grp][2 * c2] = vx; lds_q[grp][2 * c2 + 1] = vy; }
    }
    wave_sync();
    double pb = LPMP_INF;                            // peer's bound after this receive
    if (act && g < L) {
      const double qv = lds_q[grp][g];
      const double delta = ms_v + qv;                // omega = 1: delta = min-marginal
      theta += delta;                                // RepamLeft(+delta)
      const double mn = ms_v - delta;
      ms[g] = mn;                                    // RepamRight(-delta)
      pb = mn + qv;
    }
    pb = vec_min<G, L>(pb);
    if (act && g == 0) lb[op.peer] = pb;
    wave_sync();
  }
  if (flags & SWEEP_PRIMAL) {
    const int lab = group_argmin<G, L>(theta, vl, g);
    if (live && g == 0 && (rec.kind_flags & UPD_PRIMAL)) store_label(primal, rec.factor, L, lab);
  }
  // sends: delta = omega * theta_snapshot; peer += delta; theta -= delta
  if (vl) {
    const double snap = theta;
    for (int k = 0; k < rec.n_send; ++k) {
      const Op op = ops[rec.op_begin + rec.n_recv + k];
      const int side = (op.info >> 5) & 1;
      double* ms = dual + op.peer_dual + (side == 0 ? 0 : L);
      const double delta = op.omega * snap;
      ms[g] += delta;
      theta -= delta;
      if (g == 0) lb[op.peer] = LPMP_NAN;
    }
    if (flags & SWEEP_RESIDUAL) {   // second round from the live factor with the running weight sum
      double residual = 0.0;
      for (int k = 0; k < rec.n_send; ++k) {
        const Op op = ops[rec.op_begin + rec.n_recv + k];
        const int side = (op.info >> 5) & 1;
        double* ms = dual + op.peer_dual + (side == 0 ? 0 : L);
        residual += op.omega;
        const double delta = residual * theta;
        ms[g] += delta;
        theta -= delta;
      }
    }
    own_g[g] = theta;
  }
  { const double ob = vec_min<G, L>(vl ? theta : LPMP_INF); if (live && g == 0) lb[rec.factor] = ob; }
}


// -------------------------------------------------------------------------------------------------
// Dense fast path, packed form (v2).  Same arithmetic as sweep_dense_kernel, restructured for latency:
//   * the factor's record and all its ops arrive as ONE coalesced packet (no rec -> ops dependent hop);
//   * the tables of up to KMAX receives and the target vectors of up to 4 sends are requested before
//     anything is reduced, so a factor's HBM round trips overlap instead of chaining
//     (what bounds the row-major order, where a level holds only <= min(H,W) factors).
// -------------------------------------------------------------------------------------------------
template <int G, class T> __device__ __forceinline__ T uni(T v) {
  if constexpr (G == 64 && sizeof(T) == 4) return (T)__builtin_amdgcn_readfirstlane((int)v);
  else return v;
}
template <int G> __device__ __forceinline__ int64_t uni64(int64_t v) {
  if constexpr (G == 64) {
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((int64_t)hi << 32) | (unsigned int)lo;
  } else return v;
}

// Stage one factor's record + ops in the lane group's LDS slab.
//   stride > 0: packet mode, record and ops contiguous at packets[idx * stride]   (one hop)
//   stride < 0: indirect mode, record at recs[idx], then its ops at ops[op_begin]   (two hops, any op count <= cap)
template <int G>
__device__ __forceinline__ void load_packet(double2_t* slab, const Op* __restrict__ packets, const UpdRec* __restrict__ recs,
                                            const Op* __restrict__ ops, int64_t idx, int stride, bool live, int g) {
  if (stride > 0) {
    if (live) {
      const double2_t* src = reinterpret_cast<const double2_t*>(packets + idx * stride);
      for (int p = g; p < 3 * stride; p += G) slab[p] = src[p];
    }
    wave_sync();
  } else {
    if (live) {
      const double2_t* src = reinterpret_cast<const double2_t*>(recs + idx);
      if (g < 3) slab[g] = src[g];
    }
    wave_sync();
    if (live) {
      const UpdRec* hdr = reinterpret_cast<const UpdRec*>(slab);
      const int n_ops = hdr->n_recv + hdr->n_send;
      const double2_t* src = reinterpret_cast<const double2_t*>(ops + hdr->op_begin);
      for (int p = g; p < 3 * n_ops; p += G) slab[3 + p] = src[p];
    }
    wave_sync();
  }
}

// VAR: L is the padded width; the label count of the factor (<= L) and the dims of each peer table (d0 x d1, both
// <= L, the own side's equal to the label count) are read at run time, lanes beyond them carry +inf / 0
// NT: streaming policy of the table loads.  A: access policy of the duals.  CHAIN: called from the chain executor with a
// ticket: everything constant is requested first, then the ticket's predecessors are awaited, then the duals are read.
template <int L, int KMAX, bool VAR, bool NT, int A, bool CHAIN, bool MBOX = false>
__device__ __forceinline__ void dense_pk_body(const Op* __restrict__ packets, const UpdRec* __restrict__ recs, const Op* __restrict__ ops,
                                              double* __restrict__ dual, const double* __restrict__ cdata, double* __restrict__ lb,
                                              int32_t* __restrict__ primal, int64_t count, int stride, int flags, int64_t block,
                                              const ChainArgs* ca, int ticket, double* __restrict__ lbh = nullptr, int hmode = 0,
                                              unsigned long long* __restrict__ mbox = nullptr, int n_deps = -1) {
  static_assert(!CHAIN || A == ACC_COH, "chain bodies hand results over through relaxed agent-scope flags: every dual access must be an agent-scope (sc1) access");
  static_assert(MAILBOX_SENDS >= 1 && MAILBOX_SENDS <= 4, "plan.hpp: the sends whose fields are held in registers (KS)");
  static_assert(!MBOX || CHAIN, "the mailbox belongs to the chain executor");   // (an instantiation of its own: the joined passes of the headline grid lost 8 % with the mailbox fields in their registers)
  constexpr int G = DenseCfg<L>::G;
  constexpr int CL = L / 2, RPL = 2 * G / L, NL = L / RPL, GPB = 256 / G;
  constexpr int KS = MBOX ? MAILBOX_SENDS : 4;   // sends whose target vectors are prefetched / forwarded
  constexpr int NFW = MBOX ? MAILBOX_SENDS : 4;  // receives whose result can be forwarded in registers (plan.cpp: hints of mailbox chains stay below)
  constexpr int PIECES = 3 * (1 + pk_dense_cap(L));      // 16-B pieces of the largest packet / op list
  __shared__ double2_t lds_pk[GPB][PIECES];
  __shared__ double lds_mo[GPB][L];
  __shared__ double lds_q[GPB][L];
  const int grp = threadIdx.x / G, g = threadIdx.x % G;
  const int64_t idx = block * GPB + grp;
  const bool live = idx < count;
  const int c2 = g % CL, rl = g / CL;
  load_packet<G>(lds_pk[grp], packets, recs, ops, idx, stride, live, g);
  const UpdRec* hdr = reinterpret_cast<const UpdRec*>(&lds_pk[grp][0]);
  const Op* lop = reinterpret_cast<const Op*>(&lds_pk[grp][3]);
  const int n_recv = live ? uni<G>((int)hdr->n_recv) : 0;
  const int n_send = live ? uni<G>((int)hdr->n_send) : 0;
  const bool preload_ok = live && (uni<G>(hdr->kind_flags) & UPD_PRELOAD_OK) != 0;
  double* own_g = dual + (live ? uni64<G>(hdr->dual_off) : 0);
  const int Lr = VAR ? (live ? uni<G>(hdr->d0) : 0) : L;      // label count of this factor
  const bool vl = live && g < Lr;
  double theta = 0.0;
  double sm[KS];                                 // target vectors of the first KS sends
#pragma unroll
  for (int k = 0; k < KS; ++k) sm[k] = 0.0;
  // chain executor: the fields of the first KS send ops, read from the LDS packet ONCE, before the receives write to
  // LDS — otherwise every field is re-read after those writes, one dependent LDS round trip each on the critical path
  // of a dependent level (0.64 us for two sends, tools/chain_trace.py)
  [[maybe_unused]] double* s_ms[KS]; [[maybe_unused]] double s_om[KS]; [[maybe_unused]] int s_fw[KS], s_peer[KS];
  [[maybe_unused]] unsigned long long* s_box[KS];   // mailbox row the send's vector also goes to (plan.hpp, OP_MAILBOX), or nullptr
  if constexpr (CHAIN) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      s_ms[k] = dual; s_om[k] = 0.0; s_fw[k] = 0; s_peer[k] = 0; s_box[k] = nullptr;
      if (k < n_send) {
        const Op& o = lop[n_recv + k];
        s_ms[k] = dual + uni64<G>(o.peer_dual) + (((uni<G>(o.info) >> 5) & 1) ? (VAR ? uni<G>(o.pd0) : L) : 0);
        s_om[k] = o.omega; s_fw[k] = uni<G>(o.pad); s_peer[k] = uni<G>(o.peer);
        if constexpr (MBOX) { if (uni<G>(o.info) & OP_MAILBOX) s_box[k] = mbox + uni64<G>(o.peer_const) * (2 * L); }
      }
    }
  }
  auto load_own_and_targets = [&]() {
    theta = vl ? ld_dual<A>(own_g + g) : 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (preload_ok && k < n_send && g < Lr) {
        if constexpr (CHAIN) sm[k] = ld_dual<A>(s_ms[k] + g);
        else {
          const Op& o = lop[n_recv + k];
#ifdef LPMP_ABLATE_SEND_VEC
          sm[k] = (double)o.peer_dual * 1e-300;
#else
          sm[k] = ld_dual<A>(dual + uni64<G>(o.peer_dual) + (((uni<G>(o.info) >> 5) & 1) ? (VAR ? uni<G>(o.pd0) : L) : 0) + g);
#endif
        }
      }
    }
  };
  if constexpr (!CHAIN) load_own_and_targets();
  double mnew[NFW];                              // results of receives whose store is deferred to a send
#pragma unroll
  for (int k = 0; k < NFW; ++k) mnew[k] = 0.0;
  int max_recv = n_recv;
  if (G < 64) {
#pragma unroll
    for (int m = 32; m >= G; m >>= 1) max_recv = max(max_recv, __shfl_xor(max_recv, m, 64));
  }
  bool aborted = false;
  // mailbox chains: the tracked bounds of the first chunk's peers are reduced after the sends (the next level waits for those)
  [[maybe_unused]] double late_pb[KMAX]; [[maybe_unused]] int late_peer[KMAX], late_mode[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { late_pb[j] = LPMP_INF; late_peer[j] = -1; late_mode[j] = 0; }

  // one chunk of up to KMAX receives starting at c; FW: c is a compile-time constant and results may be forwarded;
  // FIRST (chain executor): the tables (constants) are requested, then the predecessors awaited, then the duals read
  auto chunk = [&](const int c, auto fw_tag, auto first_tag) {
    constexpr bool FW = decltype(fw_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    double2_t t[KMAX][NL];
    double msv[KMAX], mov[KMAX];
    int64_t pdual[KMAX];
    int side[KMAX], defer[KMAX], roff[KMAX];     // roff: offset of the own-side message vector in the peer's dual
    int dR[KMAX], dC[KMAX];
    [[maybe_unused]] const unsigned long long* box[KMAX];   // mailbox row the other side's vector is polled from, or nullptr
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {             // request everything constant first
      const bool act = c + j < n_recv;
      pdual[j] = 0; side[j] = 0; defer[j] = 0; roff[j] = 0; msv[j] = 0.0; mov[j] = 0.0; dR[j] = L; dC[j] = L; box[j] = nullptr;
      if (act) {
        const Op& o = lop[c + j];
        pdual[j] = uni64<G>(o.peer_dual);
        if constexpr (MBOX) {
          if (uni<G>(o.info) & OP_MAILBOX) { long long row; __builtin_memcpy(&row, &o.omega, 8); box[j] = mbox + uni64<G>(row) * (2 * L); }
        }
        side[j] = (uni<G>(o.info) >> 5) & 1;
        defer[j] = FW ? uni<G>(o.pad) : 0;
        const double* T = cdata + uni64<G>(o.peer_const);
        if constexpr (VAR) {
          const int R = uni<G>(o.pd0), C = uni<G>(o.pd1);
          dR[j] = R; dC[j] = C;
          roff[j] = side[j] == 0 ? 0 : R;
          if (((uni<G>(o.info) >> 8) & 15) == LPMP_F_PAIRWISE_POTTS) {
            // a Potts neighbour among dense ones: its table diff * [a != b] is made up in registers
            const double diff = T[0];
#pragma unroll
            for (int i = 0; i < NL; ++i) {
              const int row = i * RPL + rl;
              t[j][i].x = (row < R && 2 * c2 < C) ? (row == 2 * c2 ? 0.0 : diff) : LPMP_INF;
              t[j][i].y = (row < R && 2 * c2 + 1 < C) ? (row == 2 * c2 + 1 ? 0.0 : diff) : LPMP_INF;
            }
          } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
              const int row = i * RPL + rl;
              const double* Tr = T + (int64_t)row * C + 2 * c2;
              t[j][i].x = (row < R && 2 * c2 < C) ? ld_stream<NT>(Tr) : LPMP_INF;
              t[j][i].y = (row < R && 2 * c2 + 1 < C) ? ld_stream<NT>(Tr + 1) : LPMP_INF;
            }
          }
        } else {
          roff[j] = side[j] == 0 ? 0 : L;
#pragma unroll
#ifdef LPMP_ABLATE_TABLE      // timing experiments only (tools/build_variant.sh): results are wrong
          for (int i = 0; i < NL; ++i) t[j][i] = double2_t{(double)(uintptr_t)T * 1e-300, 0.0};
#else
          for (int i = 0; i < NL; ++i) t[j][i] = ld_stream<NT>(reinterpret_cast<const double2_t*>(T + (int64_t)i * 2 * G + 2 * g));
#endif
        }
      } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) t[j][i] = double2_t{0.0, 0.0};
      }
    }
    if constexpr (CHAIN && FIRST) {              // the tables are in flight while the predecessors finish
      if (MBOX && n_deps == 0) chain_stamp(*ca, ticket, 1);   // (chain_loop_ahead knows: nothing to wait for, no barrier)
      else aborted = !chain_wait(*ca, ticket);
      load_own_and_targets();
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {             // then the message vectors of these receives
      if (c + j < n_recv) {
        if constexpr (VAR) {
          if (g < Lr) msv[j] = ld_dual<A>(dual + pdual[j] + roff[j] + g);
          if (!(MBOX && box[j]) && g < (side[j] == 0 ? dC[j] : dR[j])) mov[j] = ld_dual<A>(dual + pdual[j] + (side[j] == 0 ? dR[j] : 0) + g);
        } else if (g < L) {
#ifdef LPMP_ABLATE_RECV_VEC
          msv[j] = (double)pdual[j] * 1e-300; mov[j] = 0.0;
#else
          msv[j] = ld_dual<A>(dual + pdual[j] + (side[j] == 0 ? 0 : L) + g);
          if (!(MBOX && box[j])) mov[j] = ld_dual<A>(dual + pdual[j] + (side[j] == 0 ? L : 0) + g);
#endif
        }
      }
    }
    if constexpr (MBOX) {                        // ... and, with everything else in flight, the vectors that come by mailbox
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (c + j < n_recv && box[j] && g < (VAR ? (side[j] == 0 ? dC[j] : dR[j]) : L)) mov[j] = mailbox_take(*ca, box[j] + 2 * g, aborted);
    }
    if constexpr (CHAIN && FIRST) {
      // Loads and stores share one counter on this ISA and may complete out of order with respect to each other, so a
      // wait for an older load that the compiler places after a store drains that store too — a round trip to memory
      // for a write-through store, three or four times per record (measured: 4.3 us of body per dependent level,
      // tools/chain_trace.py).  Every dual this record will read is therefore awaited HERE, before its first store.
      asm volatile("" :: "v"(theta));
#pragma unroll
      for (int k = 0; k < KS; ++k) asm volatile("" :: "v"(sm[k]));
#pragma unroll
      for (int j = 0; j < KMAX; ++j) { asm volatile("" :: "v"(msv[j])); asm volatile("" :: "v"(mov[j])); }
      chain_stamp(*ca, ticket, 4);               // duals landed
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {             // then reduce, in message order
      if (c + j >= max_recv) break;
      const bool act = c + j < n_recv;
      if (g < L) lds_mo[grp][g] = mov[j];
      wave_sync();
      if (side[j] == 0) {
        const double2_t mv = *reinterpret_cast<const double2_t*>(&lds_mo[grp][2 * c2]);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          double v = fmin(t[j][i].x + mv.x, t[j][i].y + mv.y);
#ifndef LPMP_ABLATE_REDUCE
          v = row_allreduce_min<CL>(v);
#endif
          if (c2 == 0) lds_q[grp][i * RPL + rl] = v;
        }
      } else {
        double vx = LPMP_INF, vy = LPMP_INF;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const double m1v = lds_mo[grp][i * RPL + rl];
          vx = fmin(vx, t[j][i].x + m1v);
          vy = fmin(vy, t[j][i].y + m1v);
        }
#ifndef LPMP_ABLATE_REDUCE
#pragma unroll
        for (int m = G / 2; m >= CL; m >>= 1) { vx = fmin(vx, shfl_xor_f64(vx, m)); vy = fmin(vy, shfl_xor_f64(vy, m)); }
#endif
        if (rl == 0) { lds_q[grp][2 * c2] = vx; lds_q[grp][2 * c2 + 1] = vy; }
      }
      wave_sync();
      double pb = LPMP_INF;                       // peer's bound after this receive
      if (act && g < Lr) {
        const double qv = lds_q[grp][g];
        const double delta = msv[j] + qv;
        theta += delta;
        const double mn = msv[j] - delta;
        pb = mn + qv;
        bool stored = false;
        if constexpr (FW) {
          if (defer[j]) {                        // c + j < NFW by construction on the host
#pragma unroll
            for (int q = 0; q < NFW; ++q) if (q == c + j) mnew[q] = mn;
            stored = true;
          }
        }
#ifndef LPMP_ABLATE_RECV_VEC
        if (!stored) st_dual<A>(dual + pdual[j] + roff[j] + g, mn);
#else
        if (!stored && mn == 1.2345e-280) st_dual<A>(dual + pdual[j] + roff[j] + g, mn);
#endif
      }
#ifndef LPMP_ABLATE_LB_TRACK
      {
        const bool track = !(FW && defer[j]);     // a deferred receive is followed by a send that dirties the peer
        const bool hist = CHAIN && hmode == HIST_MID;   // ... but its bound at the seam between two passes is this one
        if constexpr (MBOX && FIRST) {
          late_pb[j] = pb; late_peer[j] = act ? uni<G>(lop[c + j].peer) : -1; late_mode[j] = track ? 1 : 0;
        } else
        if (track || hist) {
          pb = vec_min<G, L>(pb);
          if (act && g == 0) {
            const int peer = uni<G>(lop[c + j].peer);
            if (track) st_lb<A>(lb + peer, pb);
            if (hist) st_lb<A>(lbh + peer, pb);
          }
        }
      }
#endif
      wave_sync();
    }
  };
  // the first chunks are unrolled with constant indices so that forwarded results stay in registers
  if constexpr (CHAIN) {
    chunk(0, std::true_type{}, std::true_type{});   // also waits when the workgroup's factors receive nothing
    chain_stamp(*ca, ticket, 5);                    // first receives done
  } else {
    if (max_recv > 0) chunk(0, std::true_type{}, std::false_type{});
  }
  if constexpr (KMAX < NFW) { if (max_recv > KMAX) chunk(KMAX, std::true_type{}, std::false_type{}); }
  if constexpr (2 * KMAX < NFW) { if (max_recv > 2 * KMAX) chunk(2 * KMAX, std::true_type{}, std::false_type{}); if (max_recv > 3 * KMAX) chunk(3 * KMAX, std::true_type{}, std::false_type{}); }
  for (int c = (KMAX >= NFW ? KMAX : NFW); c < max_recv; c += KMAX) chunk(c, std::false_type{}, std::false_type{});

  if (flags & SWEEP_PRIMAL) {
    const int lab = group_argmin<G, L>(theta, vl, g);
    if (live && g == 0 && (uni<G>(hdr->kind_flags) & UPD_PRIMAL)) store_label(primal, uni<G>(hdr->factor), Lr, lab);
  }
#ifndef LPMP_ABLATE_LB_TRACK
  // Bound of a pairwise peer after a SEND that follows this record's receive through the same vector (plan.cpp marks the pair:
  // Op::pad of the send = index of that receive + 1).  The receive left m_s = -q (q[a] = min_b T[a][b] + m_o[b], up to the
  // rounding of m_s - (m_s + q)), nothing else of the peer moved since, and the send adds omega * theta_snap: the peer's
  // bound min_a (m_s[a] + q[a]) is omega * min_a theta_snap[a] up to ~1e-16 of |q| per factor (sums of millions of them
  // stay below 1e-12 of the bound).  In the weight modes in which every message is received and then sent (uniform /
  // damped_uniform: the rounding iterations of MpRoundingSolver) this keeps EVERY pairwise bound tracked, and
  // LP::LowerBound after such a pass is a sum instead of a scan of all tables (C3: 0.04 instead of 3.2 ms).
  // (parked in LDS — lds_q is free once the receives are done, and only the lane that wrote it reads it back: a register
  // held across the send loops takes the 32-label chain body from 167 to 170 VGPRs, i.e. from 3 to 2 waves per SIMD)
  // Not in the mailbox bodies: there a record is a link of a latency-bound chain, and the reduction (+ 2-3 % per level,
  // measured) buys nothing — deep schedules are not what a rounding iteration's bound waits for.
  if constexpr (!MBOX) {
    const double snap_min_v = vec_min<G, L>(vl ? theta : LPMP_INF);
    if (g == 0) lds_q[grp][0] = snap_min_v;
    if constexpr (CHAIN) {   // joined passes: the factor's own bound at the seam between two passes is this minimum too
      if (hmode == HIST_MID) { if (live && g == 0) st_lb<A>(lbh + uni<G>(hdr->factor), snap_min_v); }
    }
  }
  const bool send_bounds = !MBOX && !(flags & SWEEP_RESIDUAL);   // (the residual rule adds to the sent vectors once more)
#define LPMP_SNAP_MIN (lds_q[grp][0])
#else
  const bool send_bounds = false;
#define LPMP_SNAP_MIN 0.0
#endif
  if (vl && !aborted) {
    const double snap = theta;
    if constexpr (CHAIN) {
      // targets that could not be requested up front (a receive of this record rewrites them): all of them now, one
      // wait inside this branch — a load inside the send loop would put a wait at the loop's join that drains the
      // write-through stores of the receives on EVERY path (measured: 1 us per record, tools/chain_trace.py)
      if (!preload_ok) {
#pragma unroll
        for (int k = 0; k < KS; ++k) if (k < n_send && s_fw[k] == 0) sm[k] = ld_dual<A>(s_ms[k] + g);
#pragma unroll
        for (int k = 0; k < KS; ++k) asm volatile("" :: "v"(sm[k]));
      }
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if (k < n_send) {
          const int fw = s_fw[k];
          const double cur = fw > 0 ? (fw == 1 ? mnew[0] : fw == 2 ? mnew[NFW > 1 ? 1 : 0] : fw == 3 ? mnew[NFW > 2 ? 2 : 0] : mnew[NFW > 3 ? 3 : 0]) : sm[k];
          const double delta = s_om[k] * snap;
          // (the residual rule below adds to the vector once more: the mailbox gets the final value there)
          if constexpr (MBOX) { if (s_box[k] && !(flags & SWEEP_RESIDUAL)) mailbox_put(s_box[k] + 2 * g, cur + delta, ca->epoch); }
          st_dual<A>(s_ms[k] + g, cur + delta);
          theta -= delta;
#ifndef LPMP_ABLATE_LB_TRACK
          // a vector that goes to the mailbox has a reader later in this launch, which sets the peer's tracked bound itself
          // — and is not ordered after THIS store, so it is left out
          if (g == 0 && !(MBOX && s_box[k])) st_lb<A>(lb + s_peer[k], fw > 0 && send_bounds ? s_om[k] * LPMP_SNAP_MIN : LPMP_NAN);
#endif
        }
      }
      if constexpr (MBOX) chain_stamp(*ca, ticket, 6);   // first sends issued (the mailbox has them)
    } else {
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if (k < n_send) {
          const Op& o = lop[n_recv + k];
          double* ms = dual + uni64<G>(o.peer_dual) + (((uni<G>(o.info) >> 5) & 1) ? (VAR ? uni<G>(o.pd0) : L) : 0);
          const int fw = uni<G>(o.pad);
          double cur;
          if (fw > 0) cur = fw == 1 ? mnew[0] : fw == 2 ? mnew[NFW > 1 ? 1 : 0] : fw == 3 ? mnew[NFW > 2 ? 2 : 0] : mnew[NFW > 3 ? 3 : 0];
          else cur = preload_ok ? sm[k] : ld_dual<A>(ms + g);
          const double delta = o.omega * snap;
#ifdef LPMP_ABLATE_SEND_VEC
          if (cur + delta == 1.2345e-280)
#endif
          st_dual<A>(ms + g, cur + delta);
          theta -= delta;
#ifndef LPMP_ABLATE_LB_TRACK
          if (g == 0) st_lb<A>(lb + uni<G>(o.peer), fw > 0 && send_bounds ? o.omega * LPMP_SNAP_MIN : LPMP_NAN);
#endif
        }
      }
    }
    // the remaining sends SC at a time: the target vectors are requested together, then updated in message order (a record
    // never sends twice into one vector: plan.cpp gives such records an op-by-op class).  One at a time, every send was a
    // dependent load -> store round trip: a variable of a random graph (C4: ten neighbours on average, up to 30) spent
    // most of its time in this loop; the first level of C4's sweep (200 000 records that only send, nine messages each) is
    // bound by exactly this chain (profiles/r03_c4c_launch_rates.txt)
    constexpr int SC = 8;
    for (int k0 = KS; k0 < n_send; k0 += SC) {
      double* msk[SC]; double cur[SC], om[SC]; int pr[SC];
#pragma unroll
      for (int q = 0; q < SC; ++q) {
        msk[q] = own_g; cur[q] = 0.0; om[q] = 0.0; pr[q] = 0;
        if (k0 + q < n_send) {
          const Op& o = lop[n_recv + k0 + q];
          msk[q] = dual + o.peer_dual + (((o.info >> 5) & 1) ? (VAR ? o.pd0 : L) : 0);
          om[q] = o.omega; pr[q] = o.peer;
#ifdef LPMP_ABLATE_SEND_VEC
          cur[q] = (double)o.peer_dual * 1e-300;
#else
          cur[q] = ld_dual<A>(msk[q] + g);
#endif
        }
      }
#pragma unroll
      for (int q = 0; q < SC; ++q) {
        if (k0 + q < n_send) {
          const double delta = om[q] * snap;
#ifdef LPMP_ABLATE_SEND_VEC
          if (cur[q] + delta == 1.2345e-280)
#endif
          st_dual<A>(msk[q] + g, cur[q] + delta);
          theta -= delta;
          if (g == 0) st_lb<A>(lb + pr[q], LPMP_NAN);
        }
      }
    }
    if (flags & SWEEP_RESIDUAL) {
      double residual = 0.0;
      for (int k = 0; k < n_send; ++k) {
        const Op& o = lop[n_recv + k];
        double* ms = dual + o.peer_dual + (((o.info >> 5) & 1) ? (VAR ? o.pd0 : L) : 0);
        residual += o.omega;
        const double delta = residual * theta;
        const double v = ld_dual<A>(ms + g) + delta;
        if constexpr (MBOX) { if (k < KS && (o.info & OP_MAILBOX)) mailbox_put(mbox + o.peer_const * (2 * L) + 2 * g, v, ca->epoch); }
        st_dual<A>(ms + g, v);
        theta -= delta;
      }
    }
    st_dual<A>(own_g + g, theta);
  }
#ifndef LPMP_ABLATE_LB_TRACK
  if constexpr (MBOX) {
    // (a peer this record also SENDS to ends up stale: the send's mark must be the last word, so its bound is not stored)
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j >= max_recv) break;
      const double pb = vec_min<G, L>(late_pb[j]);
      if (late_peer[j] >= 0 && (late_mode[j] & 1) && g == 0) {
        bool sent = false;
        for (int k = 0; k < n_send; ++k) if (uni<G>(lop[n_recv + k].peer) == late_peer[j]) sent = true;
        if (!sent) st_lb<A>(lb + late_peer[j], pb);
      }
    }
  }
  {
    const double ob = vec_min<G, L>(vl ? theta : LPMP_INF);
    if (live && g == 0) {
      st_lb<A>(lb + uni<G>(hdr->factor), ob);
      if constexpr (CHAIN) { if (hmode == HIST_END) st_lb<A>(lbh + uni<G>(hdr->factor), ob); }
    }
  }
#endif
}

// experiments (tools/build_variant.sh): receives in flight per lane group of the 16-label class, minimum waves per SIMD
#ifndef LPMP_KMAX16
#define LPMP_KMAX16 2
#endif
#ifdef LPMP_PK_WPE
#define LPMP_PK_BOUNDS __launch_bounds__(256, LPMP_PK_WPE)
#else
#define LPMP_PK_BOUNDS __launch_bounds__(256)
#endif
template <int L, int KMAX, bool VAR, bool NT>
__global__ void LPMP_PK_BOUNDS
sweep_dense_pk_kernel(const Op* __restrict__ packets, const UpdRec* __restrict__ recs, const Op* __restrict__ ops,
                      double* __restrict__ dual, const double* __restrict__ cdata, double* __restrict__ lb,
                      int32_t* __restrict__ primal, int64_t count, int stride, int flags) {
  dense_pk_body<L, KMAX, VAR, NT, NT ? ACC_NT : ACC_PLAIN, false>(packets, recs, ops, dual, cdata, lb, primal, count, stride, flags,
                                                                   (int64_t)blockIdx.x, nullptr, 0);
}

// The chain executor's launch: workgroups take tickets until none is left (kernel classes of one BODY per launch).
// The next ticket is drawn while the current one is processed, so the counter's latency is off the critical path.
template <class Body>
__device__ __forceinline__ void chain_loop(const ChainArgs& ca, const ChainLaunch* __restrict__ launches, Body body) {
  __shared__ int s_ticket[2];
  if (threadIdx.x == 0) s_ticket[0] = atomicAdd(ca.next, 1);
  __syncthreads();
  for (int it = 0;; ++it) {
    const int ticket = s_ticket[it & 1];
    if (ticket >= ca.n_tickets) break;
    if (threadIdx.x == 0) s_ticket[(it + 1) & 1] = atomicAdd(ca.next, 1);
    chain_stamp(ca, ticket, 0);                    // ticket in hand
    const TicketRef tr = chain_ticket_ref(ca, ticket);
    ChainLaunch ln = launches[ca.tk_launch[tr.idx] + tr.copy * ca.per_launch_shift];
    if (ln.pad & 3) {                                                  // joined passes: bound row of this copy's pass (HIST_* | row << 2)
      ln.pad += (tr.copy * ca.per_row_shift) << 2;
      if ((ln.pad >> 2) >= ca.hist_rows) ln.pad = 0;
    }
    body(ln, (int64_t)ca.tk_block[tr.idx], ticket);
    chain_publish(ca, ticket);
    __syncthreads();                               // s_ticket[(it + 1) & 1] is written, the LDS of the body is free again
  }
}
// MBOX: a chain whose launches all carry CHAIN_LAUNCH_MAILBOX (plan.cpp marks every launch of such a chain)
#ifndef LPMP_MBOX_WPE
#define LPMP_MBOX_WPE 1
#endif
// The same loop for mailbox chains, where a level is a couple of microseconds and what a workgroup needs before it can
// even request its records — ticket number (an atomic on the far side of the fabric), then the ticket's launch, block and
// dependency count (three arrays streamed from HBM once per launch: a miss each) — was 3.5 us of round trips in a row per
// ticket (tools/chain_trace.py: t1 - t0), more than a level.  Here the first thread draws ticket numbers TWO iterations
// ahead and fetches the fields of the next ticket while the current one is processed: all of it returns during the body and
// is put down in LDS after it.  (A workgroup runs its tickets in increasing order, so the lowest unfinished ticket of the
// launch is always one that is running: drawing ahead cannot deadlock.)  Costs registers: not for the joined passes.
template <class Body>
__device__ __forceinline__ void chain_loop_ahead(const ChainArgs& ca, const ChainLaunch* __restrict__ launches, Body body) {
  __shared__ int s_tk[3][4];                       // ticket, launch, block, number of dependencies
  if (threadIdx.x == 0) {
    const int t = atomicAdd(ca.next, 1);
    s_tk[0][0] = t;
    if (t < ca.n_tickets) { s_tk[0][1] = ca.tk_launch[t]; s_tk[0][2] = ca.tk_block[t]; s_tk[0][3] = ca.dep_off[t + 1] - ca.dep_off[t]; }
    s_tk[1][0] = atomicAdd(ca.next, 1);
  }
  __syncthreads();
  for (int it = 0;; ++it) {
    const int cur = it % 3, nx1 = (it + 1) % 3, nx2 = (it + 2) % 3;
    const int ticket = s_tk[cur][0];
    if (ticket >= ca.n_tickets) break;
    int t2 = 0, f_launch = 0, f_block = 0, f_deps = 0;
    if (threadIdx.x == 0) {
      t2 = atomicAdd(ca.next, 1);
      const int t1 = s_tk[nx1][0];
      if (t1 < ca.n_tickets) { f_launch = ca.tk_launch[t1]; f_block = ca.tk_block[t1]; f_deps = ca.dep_off[t1 + 1] - ca.dep_off[t1]; }
    }
    chain_stamp(ca, ticket, 0);                    // ticket in hand
    if (ca.trace && threadIdx.x == 0) ca.trace[8 * (int64_t)ticket + 7] = (long long)blockIdx.x;   // ... by this workgroup
    const ChainLaunch ln = launches[s_tk[cur][1]];
    body(ln, (int64_t)s_tk[cur][2], ticket, s_tk[cur][3]);
    if (threadIdx.x == 0) { s_tk[nx1][1] = f_launch; s_tk[nx1][2] = f_block; s_tk[nx1][3] = f_deps; s_tk[nx2][0] = t2; }
    chain_publish(ca, ticket);
    __syncthreads();
  }
}
// (the exact 32-label body needs 167-171 VGPRs depending on small things: three waves per SIMD are asked for, so that it stays
// at the 170 that allows them — the joined passes of the headline grid lose 8 % at two)
template <int L, int KMAX, bool VAR, bool NT, bool MBOX>
__global__ void __launch_bounds__(256, MBOX ? LPMP_MBOX_WPE : (L == 32 && !VAR ? 3 : 1))
chain_dense_pk_kernel(ChainArgs ca, const ChainLaunch* __restrict__ launches, double* __restrict__ dual,
                      const double* __restrict__ cdata, double* __restrict__ lb, int32_t* __restrict__ primal, int flags) {
  if constexpr (MBOX) {
    chain_loop_ahead(ca, launches, [&](const ChainLaunch& ln, int64_t block, int ticket, int n_deps) {
      dense_pk_body<L, KMAX, VAR, NT, ACC_COH, true, true>(ln.packets, ln.recs, ln.ops, dual, cdata, lb, primal, ln.count, ln.stride, flags, block, &ca, ticket,
                                                           nullptr, 0, ca.mailbox, n_deps);
    });
  } else {
    chain_loop(ca, launches, [&](const ChainLaunch& ln, int64_t block, int ticket) {
      const int hmode = ca.lb_hist ? (ln.pad & 3) : 0;
      dense_pk_body<L, KMAX, VAR, NT, ACC_COH, true>(ln.packets, ln.recs, ln.ops, dual, cdata, lb, primal, ln.count, ln.stride, flags, block, &ca, ticket,
                                                     hmode ? ca.lb_hist + (int64_t)(ln.pad >> 2) * ca.hist_stride : nullptr, hmode);
    });
  }
}

// the generic kernels inside the chain executor (chains of tiny factors: multicut / C5 labeling lists): nothing
// constant worth requesting ahead, so the wait comes first
static_assert(GEN_WAVES == GENERIC_BLOCK_RECORDS && 64 * SMALL_WAVES == SMALL_BLOCK_RECORDS, "plan.hpp: records per workgroup of the generic kernels");
template <int G>
__global__ void __launch_bounds__(GenCtx<G>::THREADS)
chain_generic_kernel(ChainArgs ca, const ChainLaunch* __restrict__ launches, double* __restrict__ dual, const double* __restrict__ cdata,
                     const int32_t* __restrict__ tabs, double* __restrict__ lb, int flags) {
  chain_loop(ca, launches, [&](const ChainLaunch& ln, int64_t block, int ticket) {
    if (chain_wait(ca, ticket)) generic_body<G, ACC_COH>(ln.recs, ln.ops, dual, cdata, tabs, lb, nullptr, nullptr, 0, ln.count, flags, block);
  });
}

// Level loop: a deep schedule of TINY levels of a generic class as one launch of ONE workgroup that walks the launches in
// order with a workgroup barrier in between (plan.cpp decides; C5 with local triples: 11 887 levels of a dozen one-lane
// updates).  No launch per level, no flags through memory; what a level hands to the next stays in this compute unit's L2
// (stores drained before the barrier, dual loads that bypass the L1: ACC_WG).
//   G = 64 (wave per factor): the plain body.
//   G = 1 (lane per factor): LL_WAVES computing waves + one wave that touches, LEVEL_LOOP_AHEAD levels ahead, every
//   line the computing waves will need (records, op lists, match tables, own and peer duals).  Launches of
//   labeling-list records run with one lane per op (label_ops_body), the others on the generic body (one wave).
constexpr int LEVEL_LOOP_AHEAD = 8;
constexpr int LL_WAVES = 2;                        // computing waves of level_loop_kernel<1> (+ one that runs ahead)
constexpr int CHAIN_LAUNCH_LABEL_OPS_DEV = 1, CHAIN_LAUNCH_LABEL_PAIRED_DEV = 2;   // ChainLaunch::pad (plan.hpp CHAIN_LAUNCH_LABEL_*)

// Labeling-list records with one LANE PER OP (plan.cpp marks the launches: vector factors whose ops are all labeling
// messages with the factor on the left, at most 8 receives with distinct peers and 8 sends with distinct peers, message
// length = the factor's size).
// One record per lane runs as many op rounds as its longest receive and send lists (generic_body<1>; measured
// 9.8 + 4.3 us per level on C5 with local triples); but the receives of one record do not depend on each other —
// each delta comes from its peer alone — and neither do its sends, which all start from the snapshot.  So 8 lanes
// take one record: lane j computes op j, the deltas meet in LDS and are added to the factor's vector in op order
// (same additions, same order as the sequential form: bit-identical), two rounds — the receives, then the sends —
// instead of n_recv + n_send.
// A level of the level loop staged in LDS by the wave that runs ahead (level_loop_kernel<1>): the records, their ops and
// the match tables of their labeling messages — everything CONSTANT a level needs.  Read from the L2 they are a chain of
// three dependent round trips (launch -> record -> op -> table) in front of the one that matters (the peers' costs); of the
// ~5 round trips a level of C5's local triples took (5.4 us, 64 ms per pass) they were more than half.
constexpr int LL_STAGE_RECS = 16, LL_STAGE_OPS = 16, LL_RING = 4;
struct alignas(16) StagedLevel {
  UpdRec recs[LL_STAGE_RECS];
  Op ops[LL_STAGE_RECS][LL_STAGE_OPS];
  int32_t tab[LL_STAGE_RECS][LL_STAGE_OPS][SMALL_MAXD];
  const Op* ops_base;   // the launch's op array (for the stage that copies the ops)
  int32_t count;        // records staged, or -1: this level is read from memory (not a labeling-list launch, or too large)
  int32_t pad[1];
};
template <int A, bool PAIRED, bool STAGED = false>
__device__ __forceinline__ void label_ops_body(const ChainLaunch& ln, int64_t first, double* __restrict__ dual, const int32_t* __restrict__ tabs,
                                               double* __restrict__ lb, double (*D)[8][8], double (*S)[8], const StagedLevel* st = nullptr) {
  const int lane = threadIdx.x & 63, q = lane >> 3, j = lane & 7;
  const int64_t idx = first + q;
  const bool live = idx < ln.count;
  UpdRec rec;
  if (live) { if constexpr (STAGED) rec = st->recs[idx]; else rec = ln.recs[idx]; }
  else { rec.n_recv = 0; rec.n_send = 0; rec.d0 = 0; rec.dual_off = 0; rec.op_begin = 0; rec.factor = 0; }
  const int n_recv = rec.n_recv, n_send = rec.n_send, on = rec.d0;
  double* own_g = dual + rec.dual_off;
  double theta = (live && j < on) ? ld_dual<A>(own_g + j) : 0.0;      // lane j of the group holds element j
  // one op of this lane: its record, match table and the peer's costs; entries beyond the peer's size count as no match
  auto load_op = [&](bool has, int k, Op& o, int (&tv)[SMALL_MAXD], double (&R)[SMALL_MAXD]) {
    if (has) { if constexpr (STAGED) o = st->ops[idx][k]; else o = ln.ops[rec.op_begin + k]; }
    else { o.peer_dual = 0; o.peer_const = 0; o.omega = 0.0; o.info = 0; o.pd0 = 0; o.pd1 = 0; o.peer = 0; o.len = 0; }
    const double* peer = dual + o.peer_dual;
    const int32_t* tab = tabs + o.peer_const;
#pragma unroll
    for (int r = 0; r < SMALL_MAXD; ++r) {
      const bool in = has && r < o.pd0;
      if constexpr (STAGED) tv[r] = in ? st->tab[idx][k][r] : o.pd1; else tv[r] = in ? tab[r] : o.pd1;
      R[r] = in ? ld_dual<A>(peer + r) : LPMP_INF;
    }
  };
  if constexpr (PAIRED) {
    // every message of the record is received and then sent (send j goes where receive j came from): the peer's costs
    // stay in this lane's registers between the two rounds — one load and one store per peer, no drain in between
    const bool act = live && j < n_recv;
    Op o; int tv[SMALL_MAXD]; double R[SMALL_MAXD];
    load_op(act, j, o, tv, R);
    double omega_send = 0.0;
    if (act) { if constexpr (STAGED) omega_send = st->ops[idx][n_recv + j].omega; else omega_send = ln.ops[rec.op_begin + n_recv + j].omega; }
    const int nl = o.pd1;
    if (act) {
      st_lb<A>(lb + o.peer, LPMP_NAN);
      double nt = ((o.info >> 6) & 1) ? 0.0 : LPMP_INF;
#pragma unroll
      for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] >= nl) nt = fmin(nt, R[r]);
      for (int l = 0; l < nl; ++l) {
        double v = LPMP_INF;
#pragma unroll
        for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] == l) v = fmin(v, R[r]);
        D[q][j][l] = o.omega * (v - nt);
      }
#pragma unroll
      for (int r = 0; r < SMALL_MAXD; ++r) if (r < o.pd0 && tv[r] < nl) R[r] = R[r] + -1.0 * D[q][j][tv[r]];
    }
    wave_sync();
    for (int k = 0; k < n_recv; ++k) if (j < on) theta += +1.0 * D[q][k][j];
    if (live && j < on) S[q][j] = theta;
    wave_sync();
    if (act) {
      double* peer = dual + o.peer_dual;
      for (int l = 0; l < o.len; ++l) D[q][j][l] = omega_send * S[q][l];
#pragma unroll
      for (int r = 0; r < SMALL_MAXD; ++r) if (r < o.pd0 && tv[r] < nl) st_dual<A>(peer + r, R[r] + +1.0 * D[q][j][tv[r]]);
    }
  } else {
  // the host marks the records none of whose sends goes to a peer one of its receives rewrites (UPD_PRELOAD_OK, plan.cpp);
  // where every record of this wave's eight is one, lane j requests the costs of ITS send's peer together with those of its receive's
  // peer — one round trip instead of two — and the stores of the receives need not be drained before the sends
  bool pre = false;
  if constexpr (STAGED) pre = __all(!live || (rec.kind_flags & UPD_PRELOAD_OK) != 0) != 0;
  Op o2; int tv2[SMALL_MAXD]; double R2[SMALL_MAXD];
  {   // round 1: the receives, lane j = receive j
    const bool recv = live && j < n_recv;
    Op o; int tv[SMALL_MAXD]; double R[SMALL_MAXD];
    load_op(recv, j, o, tv, R);
    if (pre) load_op(live && j < n_send, n_recv + j, o2, tv2, R2);
    if (recv) {
      const int nl = o.pd1;
      double* peer = dual + o.peer_dual;
      st_lb<A>(lb + o.peer, LPMP_NAN);
      double nt = ((o.info >> 6) & 1) ? 0.0 : LPMP_INF;
#pragma unroll
      for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] >= nl) nt = fmin(nt, R[r]);
      for (int l = 0; l < nl; ++l) {
        double v = LPMP_INF;
#pragma unroll
        for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] == l) v = fmin(v, R[r]);
        D[q][j][l] = o.omega * (v - nt);
      }
#pragma unroll
      for (int r = 0; r < SMALL_MAXD; ++r) if (r < o.pd0 && tv[r] < nl) st_dual<A>(peer + r, R[r] + -1.0 * D[q][j][tv[r]]);
    }
  }
  if constexpr (STAGED) { if (first == 0) level_stamp(pre ? 1 : 2); }      // (tools/level_trace.py: receives issued; slot 1 when the sends' peers came along)
  if (!pre) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a send of this record may go to the peer a receive has just rewritten
  wave_sync();
  for (int k = 0; k < n_recv; ++k) if (j < on) theta += +1.0 * D[q][k][j];   // own(i) += +1.0 * dl(i), receive by receive
  if (live && j < on) S[q][j] = theta;                                        // the state every send starts from
  wave_sync();
  {   // round 2: the sends, lane j = send j
    const bool send = live && j < n_send;
    if (!pre) load_op(send, n_recv + j, o2, tv2, R2);
    if (send) {
      const int nl = o2.pd1;
      double* peer = dual + o2.peer_dual;
      st_lb<A>(lb + o2.peer, LPMP_NAN);
      for (int l = 0; l < o2.len; ++l) D[q][j][l] = o2.omega * S[q][l];
#pragma unroll
      for (int r = 0; r < SMALL_MAXD; ++r) if (r < o2.pd0 && tv2[r] < nl) st_dual<A>(peer + r, R2[r] + +1.0 * D[q][j][tv2[r]]);
    }
  }
  }
  wave_sync();
  for (int k = 0; k < n_send; ++k) if (j < on) theta += -1.0 * D[q][k][j];   // own(i) += -1.0 * dl(i), send by send
  if (live && j == 0) st_lb<A>(lb + rec.factor, LPMP_NAN);
  if (live && j < on) st_dual<A>(own_g + j, theta);
}

// The staged form of the general (not PAIRED) case.  Records, ops and match tables come from LDS (StagedLevel), so the only
// round trip in front of the arithmetic is the peers' costs — and the sends need none of their own: a send either goes to
// a peer NO receive of the record touches (its costs are requested together with the receives': nothing in the level writes
// them), or to the peer of one of the record's receives (the host marks the pair: Op::pad of the send = receive index + 1,
// of the receive = 1), and then the receive's lane hands the rewritten costs over in LDS instead of storing them — the send
// stores the final values.  One round trip per level instead of two, no drain between the rounds.  Same additions in the
// same order as label_ops_body (bit-identical).
template <int A>
__device__ __forceinline__ void label_ops_body_staged(const StagedLevel& st, int64_t first, double* __restrict__ dual, double* __restrict__ lb,
                                                      double (*D)[8][8], double (*S)[8], double (*RS)[8][8]) {
  const int lane = threadIdx.x & 63, q = lane >> 3, j = lane & 7;
  const int64_t idx = first + q;
  const bool live = idx < st.count;
  const UpdRec& rec = st.recs[live ? idx : 0];
  const int n_recv = live ? rec.n_recv : 0, n_send = live ? rec.n_send : 0, on = live ? rec.d0 : 0;
  double* own_g = dual + (live ? rec.dual_off : 0);
  double theta = (live && j < on) ? ld_dual<A>(own_g + j) : 0.0;
  const bool recv = live && j < n_recv, send = live && j < n_send;
  Op o, o2;
  o.peer_dual = 0; o.peer_const = 0; o.omega = 0.0; o.info = 0; o.pd0 = 0; o.pd1 = 0; o.peer = 0; o.len = 0; o.pad = 0; o2 = o;
  if (recv) o = st.ops[idx][j];
  if (send) o2 = st.ops[idx][n_recv + j];
  int tv[SMALL_MAXD], tv2[SMALL_MAXD]; double R[SMALL_MAXD], R2[SMALL_MAXD];
  const int fw = send ? o2.pad : 0;
  double* peer = dual + o.peer_dual; double* peer2 = dual + o2.peer_dual;
#pragma unroll
  for (int r = 0; r < SMALL_MAXD; ++r) {
    const bool in = recv && r < o.pd0, in2 = send && r < o2.pd0;
    tv[r] = in ? st.tab[idx][j][r] : o.pd1;
    tv2[r] = in2 ? st.tab[idx][n_recv + j][r] : o2.pd1;
    R[r] = in ? ld_dual<A>(peer + r) : LPMP_INF;
    R2[r] = (in2 && fw == 0) ? ld_dual<A>(peer2 + r) : LPMP_INF;
  }
  if (g_level_trace && first == 0) { asm volatile("" :: "v"(R[0]), "v"(R2[0]), "v"(theta)); level_stamp(1); }   // costs landed
  if (recv) {
    const int nl = o.pd1;
    st_lb<A>(lb + o.peer, LPMP_NAN);
    double nt = ((o.info >> 6) & 1) ? 0.0 : LPMP_INF;
#pragma unroll
    for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] >= nl) nt = fmin(nt, R[r]);
    for (int l = 0; l < nl; ++l) {
      double v = LPMP_INF;
#pragma unroll
      for (int r = 0; r < SMALL_MAXD; ++r) if (tv[r] == l) v = fmin(v, R[r]);
      D[q][j][l] = o.omega * (v - nt);
    }
#pragma unroll
    for (int r = 0; r < SMALL_MAXD; ++r) {
      if (r < o.pd0) {
        const double nv = tv[r] < nl ? R[r] + -1.0 * D[q][j][tv[r]] : R[r];
        if (o.pad) RS[q][j][r] = nv;                       // a send of this record takes it from here
        else if (tv[r] < nl) st_dual<A>(peer + r, nv);
      }
    }
  }
  wave_sync();
  if (first == 0) level_stamp(2);                                             // receives done
  for (int k = 0; k < n_recv; ++k) if (j < on) theta += +1.0 * D[q][k][j];   // own(i) += +1.0 * dl(i), receive by receive
  if (live && j < on) S[q][j] = theta;                                        // the state every send starts from
  wave_sync();
  if (send) {
    const int nl = o2.pd1;
    st_lb<A>(lb + o2.peer, LPMP_NAN);
    for (int l = 0; l < o2.len; ++l) D[q][j][l] = o2.omega * S[q][l];
#pragma unroll
    for (int r = 0; r < SMALL_MAXD; ++r) {
      if (r < o2.pd0 && tv2[r] < nl) {
        const double cur = fw > 0 ? RS[q][fw - 1][r] : R2[r];
        st_dual<A>(peer2 + r, cur + +1.0 * D[q][j][tv2[r]]);
      }
    }
  }
  wave_sync();
  for (int k = 0; k < n_send; ++k) if (j < on) theta += -1.0 * D[q][k][j];   // own(i) += -1.0 * dl(i), send by send
  if (live && j == 0) st_lb<A>(lb + rec.factor, LPMP_NAN);
  if (live && j < on) st_dual<A>(own_g + j, theta);
}

template <int G>
__global__ void __launch_bounds__(G == 1 ? 64 * (LL_WAVES + 1) : GenCtx<G>::THREADS)
level_loop_kernel(const ChainLaunch* __restrict__ launches, int n_launches, double* __restrict__ dual, const double* __restrict__ cdata,
                  const int32_t* __restrict__ tabs, double* __restrict__ lb, int flags) {
  using C = GenCtx<G>;
  if constexpr (G == 1) {
    __shared__ double D[LL_WAVES][8][8][8];
    __shared__ double S[LL_WAVES][8][8];
    __shared__ StagedLevel ring[LL_RING];
    __shared__ double RS[LL_WAVES][8][8][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool plain_rule = !(flags & (SWEEP_RESIDUAL | SWEEP_ADAPTIVE | SWEEP_PRIMAL));
    // The staging pipeline of the wave that runs ahead, one stage per level and iteration (each stage's loads depend on what
    // the previous stage left in LDS an iteration earlier, so an iteration issues all of them at once: one round trip):
    //   iteration l:  records of level l + 3  ->  ops of level l + 2  ->  match tables of level l + 1   (slot = level % LL_RING)
    // and the computing waves read level l from its slot.  stage(a, what) is also run for the first levels before the loop.
    // One iteration of the pipeline for the levels (ar, ao, at) = (l + 3, l + 2, l + 1): ALL loads are issued before anything
    // is written to LDS — three batches of independent loads, one round trip (written stage after stage, each stage's LDS
    // stores waited for its own loads and the three round trips came back in series: no gain over reading the L2 directly).
    // lnr: the ChainLaunch of level ar, loaded an iteration earlier; returns the one of level ar + 1.
    auto stage = [&](int ar, int ao, int at, const ChainLaunch& lnr, bool have_lnr) -> ChainLaunch {
      ChainLaunch next; next.packets = nullptr; next.recs = nullptr; next.ops = nullptr; next.count = 0; next.stride = 0; next.pad = 0;
      if (ar + 1 < n_launches) next = launches[ar + 1];
      // --- issue: records of level ar
      const bool r_ok = have_lnr && ar < n_launches && plain_rule && (lnr.pad & CHAIN_LAUNCH_LABEL_OPS_DEV) && lnr.count <= LL_STAGE_RECS;
      double2_t rr[3] = {double2_t{0.0, 0.0}, double2_t{0.0, 0.0}, double2_t{0.0, 0.0}};
      if (r_ok && lane < lnr.count) { const double2_t* src = reinterpret_cast<const double2_t*>(lnr.recs + lane); rr[0] = src[0]; rr[1] = src[1]; rr[2] = src[2]; }
      // --- issue: ops of level ao (its records are in LDS since the last iteration)
      StagedLevel& so = ring[((ao % LL_RING) + LL_RING) % LL_RING];
      const int no = ao < n_launches ? so.count : -1;
      double2_t oo[4][3]; bool o_has[4];           // an Op = three 16-byte pieces (plain vector registers, constant indices)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t, r = i / LL_STAGE_OPS, k = i % LL_STAGE_OPS;
        o_has[t] = no >= 0 && r < no && k < so.recs[r].n_recv + so.recs[r].n_send;
        oo[t][0] = oo[t][1] = oo[t][2] = double2_t{0.0, 0.0};
        if (o_has[t]) {
          const double2_t* src = reinterpret_cast<const double2_t*>(so.ops_base + so.recs[r].op_begin + k);
          oo[t][0] = src[0]; oo[t][1] = src[1]; oo[t][2] = src[2];
        }
      }
      // --- issue: match tables (+ the lines of the costs) of level at (its ops are in LDS since the last iteration)
      StagedLevel& st = ring[((at % LL_RING) + LL_RING) % LL_RING];
      const int nt = at < n_launches ? st.count : -1;
      int32_t tv[4][SMALL_MAXD]; bool t_has[4];
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t, r = i / LL_STAGE_OPS, k = i % LL_STAGE_OPS;
        t_has[t] = nt >= 0 && r < nt && k < st.recs[r].n_recv + st.recs[r].n_send;
        if (t_has[t]) {
          const Op& o = st.ops[r][k];
          const int32_t* tab = tabs + o.peer_const;
#pragma unroll
          for (int x = 0; x < SMALL_MAXD; ++x) tv[t][x] = x < o.pd0 ? tab[x] : o.pd1;
          const double* pd = dual + o.peer_dual;
          acc += __hip_atomic_load(pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          acc += __hip_atomic_load(pd + max(o.pd0 + o.pd1 - 1, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (k == 0) acc += __hip_atomic_load(dual + st.recs[r].dual_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      // --- everything has landed: into LDS
      if (ar < n_launches) {
        StagedLevel& sr = ring[ar % LL_RING];
        if (lane == 0) { sr.count = r_ok ? (int32_t)lnr.count : -1; sr.pad[0] = lnr.pad; sr.ops_base = lnr.ops; }
        if (r_ok && lane < lnr.count) { double2_t* dst = reinterpret_cast<double2_t*>(&sr.recs[lane]); dst[0] = rr[0]; dst[1] = rr[1]; dst[2] = rr[2]; }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t;
        if (o_has[t]) { double2_t* dst = reinterpret_cast<double2_t*>(&so.ops[i / LL_STAGE_OPS][i % LL_STAGE_OPS]); dst[0] = oo[t][0]; dst[1] = oo[t][1]; dst[2] = oo[t][2]; }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t;
        if (t_has[t]) {
#pragma unroll
          for (int x = 0; x < SMALL_MAXD; ++x) st.tab[i / LL_STAGE_OPS][i % LL_STAGE_OPS][x] = tv[t][x];
        }
      }
      asm volatile("" :: "v"(acc));
      return next;
    };
    ChainLaunch ln_ahead; ln_ahead.packets = nullptr; ln_ahead.recs = nullptr; ln_ahead.ops = nullptr; ln_ahead.count = 0; ln_ahead.stride = 0; ln_ahead.pad = 0;
    if (threadIdx.x == 0) for (int a = 0; a < LL_RING; ++a) ring[a].count = -1;
    __syncthreads();
    if (wave == LL_WAVES) {                          // fill the pipeline: three iterations ahead of the loop (levels 0, 1, 2 end up staged as far as the loop expects)
      if (n_launches > 0) ln_ahead = launches[0];
      for (int l = -3; l < 0; ++l) { ln_ahead = stage(l + 3, l + 2, l + 1, ln_ahead, true); wave_sync(); }
    }
    __syncthreads();
    for (int l = 0; l < n_launches; ++l) {
      if (g_level_trace && threadIdx.x == 0) g_level_trace[0] = l;
      level_stamp(0);
      if (wave == LL_WAVES) {                        // the wave that runs ahead
        level_stamp_of(6, 64 * LL_WAVES);
        ln_ahead = stage(l + 3, l + 2, l + 1, ln_ahead, true);
        level_stamp_of(7, 64 * LL_WAVES);
        const int la = l + LEVEL_LOOP_AHEAD;         // ... and, further ahead, the touch of everything a level that will NOT be staged reads
        ChainLaunch ln; ln.count = 0; ln.pad = 0;
        if (la < n_launches) ln = launches[la];
        if (la < n_launches && !(plain_rule && (ln.pad & CHAIN_LAUNCH_LABEL_OPS_DEV) && ln.count <= LL_STAGE_RECS)) {
          for (int64_t i = lane; i < ln.count; i += 64) {
            const UpdRec r = ln.recs[i];
            double acc = __hip_atomic_load(dual + r.dual_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int n_ops = r.n_recv + r.n_send;
            for (int k = 0; k < n_ops; ++k) {
              const Op o = ln.ops[r.op_begin + k];
              const double* pd = dual + o.peer_dual;
              acc += __hip_atomic_load(pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              acc += __hip_atomic_load(pd + max(o.pd0 + o.pd1 - 1, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // a peer of this class spans at most two lines
              if ((o.info & 15) == OP_LABELING) acc += (double)tabs[o.peer_const];
            }
            asm volatile("" :: "v"(acc));
          }
        }
      } else {
        const StagedLevel& sl = ring[l % LL_RING];
        ChainLaunch ln;
        if (sl.count >= 0) { ln.packets = nullptr; ln.recs = nullptr; ln.ops = nullptr; ln.count = sl.count; ln.stride = 0; ln.pad = sl.pad[0]; }
        else ln = launches[l];
        if (sl.count >= 0) {                         // records, ops and match tables from LDS
          if (ln.pad & CHAIN_LAUNCH_LABEL_PAIRED_DEV) { for (int64_t first = 8 * wave; first < ln.count; first += 8 * LL_WAVES) label_ops_body<ACC_WG, true, true>(ln, first, dual, tabs, lb, D[wave], S[wave], &sl); }
#ifdef LPMP_LL_OLD_BODY
          else { for (int64_t first = 8 * wave; first < ln.count; first += 8 * LL_WAVES) label_ops_body<ACC_WG, false, true>(ln, first, dual, tabs, lb, D[wave], S[wave], &sl); }
#else
          else { for (int64_t first = 8 * wave; first < ln.count; first += 8 * LL_WAVES) label_ops_body_staged<ACC_WG>(sl, first, dual, lb, D[wave], S[wave], RS[wave]); }
#endif
        } else if ((ln.pad & CHAIN_LAUNCH_LABEL_OPS_DEV) && plain_rule) {
          if (ln.pad & CHAIN_LAUNCH_LABEL_PAIRED_DEV) { for (int64_t first = 8 * wave; first < ln.count; first += 8 * LL_WAVES) label_ops_body<ACC_WG, true>(ln, first, dual, tabs, lb, D[wave], S[wave]); }
          else { for (int64_t first = 8 * wave; first < ln.count; first += 8 * LL_WAVES) label_ops_body<ACC_WG, false>(ln, first, dual, tabs, lb, D[wave], S[wave]); }
        } else if (wave == 0) {
          const int64_t nblk = (ln.count + C::FPB - 1) / C::FPB;
          for (int64_t b = 0; b < nblk; ++b)
            [&] { generic_body<1, ACC_WG>(ln.recs, ln.ops, dual, cdata, tabs, lb, nullptr, nullptr, 0, ln.count, flags, b); }();
        }
      }
      level_stamp(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this level's stores are in the L2 before the next level loads
      __syncthreads();
      level_stamp(5);
    }
  } else {
    for (int l = 0; l < n_launches; ++l) {
      const ChainLaunch ln = launches[l];
      const int64_t nblk = (ln.count + C::FPB - 1) / C::FPB;
      for (int64_t b = 0; b < nblk; ++b)
        [&] { generic_body<G, ACC_PLAIN>(ln.recs, ln.ops, dual, cdata, tabs, lb, nullptr, nullptr, 0, ln.count, flags, b); }();
      __syncthreads();
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Potts fast path: L lanes per unary factor; peers are pairwise_potts_factor(L, diff).
// min_b (diff*[a!=b] + m_o[b]) = min(m_o[a], diff + min_{b != a} m_o[b]), with min_{b != a} from the two
// smallest entries of m_o (reference vector::two_min, vector.hxx:348-443).
// -------------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256)
sweep_potts_kernel(const UpdRec* __restrict__ recs, const Op* __restrict__ ops, double* __restrict__ dual,
                   const double* __restrict__ cdata, double* __restrict__ lb, int32_t* __restrict__ primal,
                   int64_t first, int64_t count, int flags) {
  constexpr int GPB = 256 / L;
  const int grp = threadIdx.x / L, g = threadIdx.x % L;
  const int64_t idx = (int64_t)blockIdx.x * GPB + grp;
  const bool live = idx < count;
  UpdRec rec;
  if (live) rec = recs[first + idx]; else { rec.n_recv = 0; rec.n_send = 0; rec.dual_off = 0; rec.op_begin = 0; }
  double* own_g = dual + rec.dual_off;
  double theta = live ? own_g[g] : 0.0;
  int n_recv = rec.n_recv, max_recv = rec.n_recv;
#pragma unroll
  for (int m = 32; m >= L; m >>= 1) max_recv = max(max_recv, __shfl_xor(max_recv, m, 64));
  for (int k = 0; k < max_recv; ++k) {
    const bool act = k < n_recv;
    Op op;
    if (act) op = ops[rec.op_begin + k]; else { op.peer_dual = rec.dual_off; op.peer_const = 0; op.info = 0; }
    const int side = (op.info >> 5) & 1;
    double* ms = dual + op.peer_dual + (side == 0 ? 0 : L);
    const double* mo = dual + op.peer_dual + (side == 0 ? L : 0);
    const double diff = act ? cdata[op.peer_const] : 0.0;
    const double ms_v = act ? ms[g] : 0.0;
    const double mo_v = act ? mo[g] : 0.0;
    // two smallest of mo over the group (multiset semantics)
    double a1 = mo_v, a2 = LPMP_INF;
#pragma unroll
    for (int m = L / 2; m >= 1; m >>= 1) {
      const double b1 = shfl_xor_f64(a1, m), b2 = shfl_xor_f64(a2, m);
      const double n1 = fmin(a1, b1);
      const double n2 = fmin(fmax(a1, b1), fmin(a2, b2));
      a1 = n1; a2 = n2;
    }
    // exactly one lane may take the role of "the" minimum: the lowest lane holding a1
    const unsigned long long holders = __ballot(mo_v == a1);
    const int grp_shift = (threadIdx.x & 63) - g;
    const unsigned long long gmask = (L == 64 ? ~0ull : ((1ull << L) - 1ull)) << grp_shift;
    const int first_holder = __ffsll((long long)(holders & gmask)) - 1;
    const double min_except = ((int)(threadIdx.x & 63) == first_holder) ? a2 : a1;
    double pb = LPMP_INF;
    if (act) {
      const double q = fmin(0.0 + mo_v, diff + min_except);
      const double delta = ms_v + q;
      theta += delta;
      const double mn = ms_v - delta;
      ms[g] = mn;
      pb = mn + q;
    }
    pb = vec_min<L, L>(pb);
    if (act && g == 0) lb[op.peer] = pb;
  }
  if (flags & SWEEP_PRIMAL) {
    const int lab = group_argmin<L, L>(theta, live, g);
    if (live && g == 0 && (rec.kind_flags & UPD_PRIMAL)) store_label(primal, rec.factor, L, lab);
  }
  if (live) {
    const double snap = theta;
    for (int k = 0; k < rec.n_send; ++k) {
      const Op op = ops[rec.op_begin + rec.n_recv + k];
      const int side = (op.info >> 5) & 1;
      double* ms = dual + op.peer_dual + (side == 0 ? 0 : L);
      const double delta = op.omega * snap;
      ms[g] += delta;
      theta -= delta;
      if (g == 0) lb[op.peer] = LPMP_NAN;
    }
    if (flags & SWEEP_RESIDUAL) {   // second round from the live factor with the running weight sum
      double residual = 0.0;
      for (int k = 0; k < rec.n_send; ++k) {
        const Op op = ops[rec.op_begin + rec.n_recv + k];
        const int side = (op.info >> 5) & 1;
        double* ms = dual + op.peer_dual + (side == 0 ? 0 : L);
        residual += op.omega;
        const double delta = residual * theta;
        ms[g] += delta;
        theta -= delta;
      }
    }
    own_g[g] = theta;
  }
  { const double ob = vec_min<L, L>(live ? theta : LPMP_INF); if (live && g == 0) lb[rec.factor] = ob; }
}


// -------------------------------------------------------------------------------------------------
// Potts fast path, packed form: as sweep_potts_kernel, with the factor's record + ops in one packet and the
// vectors / coupling of up to 4 receives and 4 sends requested before anything is reduced (a 512 x 512 grid is
// launch-latency bound: what counts is the length of one factor's dependent chain).
// -------------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void two_min_merge(double& a1, double& a2) {   // two smallest over the L-lane group (multiset)
  auto step = [&](double b1, double b2) {
    const double n1 = fmin(a1, b1);
    const double n2 = fmin(fmax(a1, b1), fmin(a2, b2));
    a1 = n1; a2 = n2;
  };
  step(dpp_mov_f64<0xB1>(a1), dpp_mov_f64<0xB1>(a2));
  if constexpr (L >= 4) step(dpp_mov_f64<0x4E>(a1), dpp_mov_f64<0x4E>(a2));
  if constexpr (L >= 8) step(dpp_mov_f64<0x141>(a1), dpp_mov_f64<0x141>(a2));
  if constexpr (L >= 16) step(dpp_mov_f64<0x140>(a1), dpp_mov_f64<0x140>(a2));
  if constexpr (L >= 32) step(shfl_xor_f64(a1, 16), shfl_xor_f64(a2, 16));
}

// VAR: L is the padded width, the label count (<= L) is read at run time; lanes beyond it carry +inf
// A: access policy of the duals; CHAIN: called from the chain executor (see dense_pk_body)
// MBOX: the mailbox form of the chain body (see dense_pk_body and plan.cpp): message vectors between dependent records as
// tagged granules
template <int L, bool VAR, int A, bool CHAIN, bool MBOX = false>
__device__ __forceinline__ void potts_pk_body(const Op* __restrict__ packets, const UpdRec* __restrict__ recs, const Op* __restrict__ ops,
                                              double* __restrict__ dual, const double* __restrict__ cdata, double* __restrict__ lb,
                                              int32_t* __restrict__ primal, int64_t count, int stride, int flags, int64_t block,
                                              const ChainArgs* ca, int ticket, unsigned long long* __restrict__ mbox = nullptr, int n_deps = -1) {
  static_assert(!CHAIN || A == ACC_COH, "chain bodies hand results over through relaxed agent-scope flags: every dual access must be an agent-scope (sc1) access");
  static_assert(!MBOX || (CHAIN && MAILBOX_SENDS <= 4), "the mailbox belongs to the chain executor; this body forwards 4 receives and holds 4 sends");
  constexpr int GPB = 256 / L;
  constexpr int KR = 4, KS = 4;
  constexpr int PIECES = 3 * (1 + pk_indirect_cap(L));
  __shared__ double2_t lds_pk[GPB][PIECES];
  const int grp = threadIdx.x / L, g = threadIdx.x % L;
  const int64_t idx = block * GPB + grp;
  const bool live = idx < count;
  load_packet<L>(lds_pk[grp], packets, recs, ops, idx, stride, live, g);
  const UpdRec* hdr = reinterpret_cast<const UpdRec*>(&lds_pk[grp][0]);
  const Op* lop = reinterpret_cast<const Op*>(&lds_pk[grp][3]);
  const int n_recv = live ? (int)hdr->n_recv : 0;
  const int n_send = live ? (int)hdr->n_send : 0;
  const bool preload_ok = live && (hdr->kind_flags & UPD_PRELOAD_OK) != 0;
  double* own_g = dual + (live ? hdr->dual_off : 0);
  const int Lr = VAR ? (live ? (int)hdr->d0 : 0) : L;
  const bool vl = live && g < Lr;
  bool aborted = false;
  if constexpr (CHAIN) {                        // a Potts neighbour has no table to request ahead: only the coupling
    if (MBOX && n_deps == 0) chain_stamp(*ca, ticket, 1);
    else aborted = !chain_wait(*ca, ticket);
  }
  double theta = vl ? ld_dual<A>(own_g + g) : 0.0;
  double sm[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    sm[k] = 0.0;
    if (preload_ok && k < n_send && vl) {
      const Op& o = lop[n_recv + k];
      sm[k] = ld_dual<A>(dual + o.peer_dual + (((o.info >> 5) & 1) ? Lr : 0) + g);
    }
  }
  double mnew[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) mnew[k] = 0.0;
  int max_recv = n_recv;
#pragma unroll
  for (int m = 32; m >= L; m >>= 1) max_recv = max(max_recv, __shfl_xor(max_recv, m, 64));
  const int lane = threadIdx.x & 63;
  const unsigned long long gmask = (L == 64 ? ~0ull : ((1ull << L) - 1ull)) << (lane - g);

  auto chunk = [&](const int c, auto fw_tag) {
    constexpr bool FW = decltype(fw_tag)::value;
    double msv[KR], mov[KR], diff[KR];
    int64_t msoff[KR];
    int defer[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) {               // request everything first
      msv[j] = 0.0; mov[j] = VAR ? LPMP_INF : 0.0; diff[j] = 0.0; msoff[j] = 0; defer[j] = 0;
      if (c + j < n_recv) {
        const Op& o = lop[c + j];
        const int side = (o.info >> 5) & 1;
        msoff[j] = o.peer_dual + (side == 0 ? 0 : Lr) + g;
        bool boxed = false;
        if constexpr (MBOX) boxed = (o.info & OP_MAILBOX) != 0;
        if (vl) {
          msv[j] = ld_dual<A>(dual + msoff[j]);
          if (!boxed) mov[j] = ld_dual<A>(dual + o.peer_dual + (side == 0 ? Lr : 0) + g);
        }
        diff[j] = cdata[o.peer_const];
        defer[j] = FW ? o.pad : 0;
      }
    }
    if constexpr (MBOX) {                        // with everything else in flight: the vectors that come by mailbox
#pragma unroll
      for (int j = 0; j < KR; ++j) {
        if (c + j < n_recv) {
          const Op& o = lop[c + j];
          if ((o.info & OP_MAILBOX) && vl) { long long row; __builtin_memcpy(&row, &o.omega, 8); mov[j] = mailbox_take(*ca, mbox + row * (2 * L) + 2 * g, aborted); }
        }
      }
    }
    if constexpr (CHAIN) {                       // every dual awaited before the first store (see dense_pk_body)
      asm volatile("" :: "v"(theta));
#pragma unroll
      for (int k = 0; k < KS; ++k) asm volatile("" :: "v"(sm[k]));
#pragma unroll
      for (int j = 0; j < KR; ++j) { asm volatile("" :: "v"(msv[j])); asm volatile("" :: "v"(mov[j])); asm volatile("" :: "v"(diff[j])); }
    }
#pragma unroll
    for (int j = 0; j < KR; ++j) {
      if (c + j >= max_recv) break;
      const bool act = c + j < n_recv && vl;
      double a1 = mov[j], a2 = LPMP_INF;
      two_min_merge<L>(a1, a2);
      // exactly one lane may take the role of "the" minimum: the lowest lane holding a1
      const unsigned long long holders = __ballot(mov[j] == a1);
      const int first_holder = __ffsll((long long)(holders & gmask)) - 1;
      const double min_except = (lane == first_holder) ? a2 : a1;
      double pb = LPMP_INF;
      if (act) {
        const double q = fmin(0.0 + mov[j], diff[j] + min_except);
        const double delta = msv[j] + q;
        theta += delta;
        const double mn = msv[j] - delta;
        pb = mn + q;
        bool stored = false;
        if constexpr (FW) {
          if (defer[j]) {
#pragma unroll
            for (int q2 = 0; q2 < KR; ++q2) if (q2 == c + j) mnew[q2] = mn;
            stored = true;
          }
        }
        if (!stored) st_dual<A>(dual + msoff[j], mn);
      }
      if (!(FW && defer[j])) {
        pb = vec_min<L, L>(pb);
        if (c + j < n_recv && g == 0) st_lb<A>(lb + lop[c + j].peer, pb);
      }
    }
  };
  if (max_recv > 0) chunk(0, std::true_type{});
  for (int c = KR; c < max_recv; c += KR) chunk(c, std::false_type{});

  if (flags & SWEEP_PRIMAL) {
    const int lab = group_argmin<L, L>(theta, vl, g);
    if (live && g == 0 && (hdr->kind_flags & UPD_PRIMAL)) store_label(primal, hdr->factor, Lr, lab);
  }
  // (see dense_pk_body: the bound of a peer after a send that follows this record's receive through the same vector)
  double snap_min = 0.0;
  if constexpr (!MBOX) snap_min = vec_min<L, L>(vl ? theta : LPMP_INF);
  const bool send_bounds = !MBOX && !(flags & SWEEP_RESIDUAL);
  if (vl && !aborted) {
    const double snap = theta;
    if constexpr (CHAIN) {                          // no load inside the send loop (see dense_pk_body)
      if (!preload_ok) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          if (k < n_send) {
            const Op& o = lop[n_recv + k];
            if (o.pad == 0) sm[k] = ld_dual<A>(dual + o.peer_dual + (((o.info >> 5) & 1) ? Lr : 0) + g);
          }
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) asm volatile("" :: "v"(sm[k]));
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k < n_send) {
        const Op& o = lop[n_recv + k];
        double* ms = dual + o.peer_dual + (((o.info >> 5) & 1) ? Lr : 0);
        const int fw = o.pad;
        double cur;
        if (fw > 0) cur = fw == 1 ? mnew[0] : fw == 2 ? mnew[1] : fw == 3 ? mnew[2] : mnew[3];
        else if constexpr (CHAIN) cur = sm[k];
        else cur = preload_ok ? sm[k] : ld_dual<A>(ms + g);
        const double delta = o.omega * snap;
        bool boxed = false;                          // (see dense_pk_body: the reader sets the peer's tracked bound)
        if constexpr (MBOX) {
          boxed = (o.info & OP_MAILBOX) != 0;
          if (boxed && !(flags & SWEEP_RESIDUAL)) mailbox_put(mbox + o.peer_const * (2 * L) + 2 * g, cur + delta, ca->epoch);
        }
        st_dual<A>(ms + g, cur + delta);
        theta -= delta;
        if (g == 0 && !boxed) st_lb<A>(lb + o.peer, fw > 0 && send_bounds ? o.omega * snap_min : LPMP_NAN);
      }
    }
    for (int k = KS; k < n_send; ++k) {
      const Op& o = lop[n_recv + k];
      double* ms = dual + o.peer_dual + (((o.info >> 5) & 1) ? Lr : 0);
      const double delta = o.omega * snap;
      st_dual<A>(ms + g, ld_dual<A>(ms + g) + delta);
      theta -= delta;
      if (g == 0) st_lb<A>(lb + o.peer, LPMP_NAN);
    }
    if (flags & SWEEP_RESIDUAL) {
      double residual = 0.0;
      for (int k = 0; k < n_send; ++k) {
        const Op& o = lop[n_recv + k];
        double* ms = dual + o.peer_dual + (((o.info >> 5) & 1) ? Lr : 0);
        residual += o.omega;
        const double delta = residual * theta;
        const double v = ld_dual<A>(ms + g) + delta;
        if constexpr (MBOX) { if (k < KS && (o.info & OP_MAILBOX)) mailbox_put(mbox + o.peer_const * (2 * L) + 2 * g, v, ca->epoch); }
        st_dual<A>(ms + g, v);
        theta -= delta;
      }
    }
    st_dual<A>(own_g + g, theta);
  }
  { const double ob = vec_min<L, L>(vl ? theta : LPMP_INF); if (live && g == 0) st_lb<A>(lb + hdr->factor, ob); }
}

template <int L, bool VAR, bool NT>
__global__ void __launch_bounds__(256)
sweep_potts_pk_kernel(const Op* __restrict__ packets, const UpdRec* __restrict__ recs, const Op* __restrict__ ops,
                      double* __restrict__ dual, const double* __restrict__ cdata, double* __restrict__ lb,
                      int32_t* __restrict__ primal, int64_t count, int stride, int flags) {
  potts_pk_body<L, VAR, NT ? ACC_NT : ACC_PLAIN, false>(packets, recs, ops, dual, cdata, lb, primal, count, stride, flags, (int64_t)blockIdx.x, nullptr, 0);
}
template <int L, bool VAR, bool MBOX>
__global__ void __launch_bounds__(256)
chain_potts_pk_kernel(ChainArgs ca, const ChainLaunch* __restrict__ launches, double* __restrict__ dual,
                      const double* __restrict__ cdata, double* __restrict__ lb, int32_t* __restrict__ primal, int flags) {
  if constexpr (MBOX) {
    chain_loop_ahead(ca, launches, [&](const ChainLaunch& ln, int64_t block, int ticket, int n_deps) {
      potts_pk_body<L, VAR, ACC_COH, true, true>(ln.packets, ln.recs, ln.ops, dual, cdata, lb, primal, ln.count, ln.stride, flags, block, &ca, ticket, ca.mailbox, n_deps);
    });
  } else {
    chain_loop(ca, launches, [&](const ChainLaunch& ln, int64_t block, int ticket) {
      potts_pk_body<L, VAR, ACC_COH, true>(ln.packets, ln.recs, ln.ops, dual, cdata, lb, primal, ln.count, ln.stride, flags, block, &ca, ticket);
    });
  }
}

// -------------------------------------------------------------------------------------------------
// Streaming path: one wave per unary, pairwise peers of any dims up to BIG_MAX_LABELS, dense or Potts, mixed
// (class KC_DENSE_BIG: more than 32 labels, unaries with both kinds of edges; also what the run-time-dims classes
// fall back to).  Works op by op like the generic kernel, but a
// receive streams the table in blocks of 16 rows x 64 columns (16 coalesced 512-B row segments in flight per
// wave) and reduces without a round trip per row:
//   side 0 (own label = row):    16 per-lane partial minima, one per row, are transposed-and-reduced across the
//                                wave in 4 halving exchanges + 2 all-reduce steps (17 shuffles per 16 rows
//                                instead of 6 per row)
//   side 1 (own label = column): lane-local minimum over the rows, m1[row] broadcast from LDS
// -------------------------------------------------------------------------------------------------
constexpr int BIG_WAVES = 4;
static_assert(BIG_WAVES == BIG_BLOCK_RECORDS, "plan.hpp: records per workgroup of the streaming dense kernel");
// LDS of one wave: theta, m_o, q — `ldim` doubles each, ldim = the launch's largest label count rounded up to 64 (round 6: was
// BIG_MAX_LABELS for every launch, 48 KiB per workgroup, which alone held the kernel at 3 waves per SIMD whatever its registers)
struct BigLds { double* theta; double* mo; double* q; };
__device__ __forceinline__ int big_ldim(int flags) { const int k = (flags & SWEEP_BIGDIM_MASK) >> SWEEP_BIGDIM_SHIFT; return k ? 64 * k : BIG_MAX_LABELS; }
static size_t big_lds_bytes(int flags) { const int k = (flags & SWEEP_BIGDIM_MASK) >> SWEEP_BIGDIM_SHIFT; return (size_t)BIG_WAVES * 3 * (k ? 64 * k : BIG_MAX_LABELS) * sizeof(double); }
// record and op fields are the same in all lanes of the wave: as scalars, so that row addresses are scalar-base + lane offset
// (one VGPR of offsets for the 16 loads of a block instead of 16 64-bit addresses) and the loop bounds are uniform
__device__ __forceinline__ double uni_f64(double v) { return __longlong_as_double(uni64<64>(__double_as_longlong(v))); }

// v[r] = this lane's partial minimum of row r (16 rows).  Returns the minimum over all 64 lanes of ONE row:
// row 8*bit0 + 4*bit1 + 2*bit2 + bit3 of the lane index.
__device__ __forceinline__ double transpose_min16(double (&v)[16], int lane) {
  {
    const bool up = lane & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double keep = up ? v[i + 8] : v[i], send = up ? v[i] : v[i + 8];
      v[i] = fmin(keep, dpp_mov_f64<0xB1>(send));
    }
  }
  {
    const bool up = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4];
      v[i] = fmin(keep, dpp_mov_f64<0x4E>(send));
    }
  }
  {
    const bool up = lane & 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
      v[i] = fmin(keep, shfl_xor_f64(send, 4));
    }
  }
  {
    const bool up = lane & 8;
    const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    v[0] = fmin(keep, shfl_xor_f64(send, 8));
  }
  double r = v[0];
  r = fmin(r, shfl_xor_f64(r, 16));
  r = fmin(r, shfl_xor_f64(r, 32));
  return r;
}

// NT: the tables with non-temporal loads (one launch per step of an HBM-sized model); A: access policy of the duals (ACC_PLAIN in
// the launches of the product.  Round 6 also ran this body inside the chain executor with ACC_COH — joined passes of a 2-colour
// grid with 33 ... 48 labels as one persistent launch in Infinity-Cache order: bit-identical and SLOWER than one launch per step,
// 12.6 against 9.9 ms per pass at 33 labels, 15.4 against 12.1 at 40, 16.8 against 16.6 at 48, and from about 56 labels on the
// window cannot sit in the cache at all — EXPERIMENTS.md K; the kernel went, the policy parameter stayed)
template <bool NT, int A>
__device__ __forceinline__ void dense_big_body(const UpdRec* __restrict__ recs, const Op* __restrict__ ops, double* __restrict__ dual,
                                               const double* __restrict__ cdata, double* __restrict__ lb, int32_t* __restrict__ primal,
                                               int64_t first, int64_t count, int flags, int64_t block) {
  extern __shared__ double big_lds_pool[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t idx = block * BIG_WAVES + wave;
  if (idx >= count) return;
  const int ldim = big_ldim(flags);
  BigLds S{big_lds_pool + (size_t)wave * 3 * ldim, big_lds_pool + (size_t)wave * 3 * ldim + ldim, big_lds_pool + (size_t)wave * 3 * ldim + 2 * ldim};
  UpdRec rec = recs[first + idx];
  rec.dual_off = uni64<64>(rec.dual_off); rec.d0 = uni<64>(rec.d0); rec.op_begin = uni<64>(rec.op_begin);
  rec.n_recv = (int16_t)uni<64>((int)rec.n_recv); rec.n_send = (int16_t)uni<64>((int)rec.n_send);
  rec.factor = uni<64>(rec.factor); rec.kind_flags = uni<64>(rec.kind_flags);
  auto uni_op = [](Op o) {
    o.peer_dual = uni64<64>(o.peer_dual); o.peer_const = uni64<64>(o.peer_const); o.omega = uni_f64(o.omega);
    o.info = uni<64>(o.info); o.pd0 = uni<64>(o.pd0); o.pd1 = uni<64>(o.pd1); o.peer = uni<64>(o.peer);
    return o;
  };
  const int Lr = rec.d0;
  double* own_g = dual + rec.dual_off;
  for (int i = lane; i < Lr; i += 64) S.theta[i] = ld_dual<A>(own_g + i);
  const int my_row = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
  Op nxt{};
  if (rec.n_recv > 0) nxt = ops[rec.op_begin];
  for (int k = 0; k < rec.n_recv; ++k) {
    const Op op = uni_op(nxt);
    if (k + 1 < rec.n_recv) nxt = ops[rec.op_begin + k + 1];   // requested before this receive's table stream starts
    const int side = (op.info >> 5) & 1;
    const int R = op.pd0, C = op.pd1;
    const double* T = cdata + op.peer_const;
    double* ms = dual + op.peer_dual + (side == 0 ? 0 : R);
    const double* mo = dual + op.peer_dual + (side == 0 ? R : 0);
    const int Lo = side == 0 ? C : R;
    // the own side m_s is requested together with m_o (round 6: it used to be loaded after the table had been reduced — one more
    // exposed round trip per receive; one or two values per lane up to 128 labels, later ones are loaded where they are needed)
    double ms_pre[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) ms_pre[j] = lane + 64 * j < Lr ? ld_dual<A>(ms + lane + 64 * j) : 0.0;
    for (int i = lane; i < Lo; i += 64) S.mo[i] = ld_dual<A>(mo + i);
    wave_sync();
    if (((op.info >> 8) & 15) == LPMP_F_PAIRWISE_POTTS) {
      // q[x] = min(m_o[x], diff + min_{y != x} m_o[y]) from the two smallest entries of m_o (multiset) and the
      // first index that holds the smallest (reference vector::two_min, vector.hxx:348-443)
      const double diff = T[0];
      double a1 = LPMP_INF, a2 = LPMP_INF; int i1 = 0x7fffffff;
      for (int i = lane; i < Lo; i += 64) {
        const double x = S.mo[i];
        if (x < a1 || i1 == 0x7fffffff) { a2 = a1; a1 = x; i1 = i; } else if (x < a2) a2 = x;
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        const double b1 = shfl_xor_f64(a1, m), b2 = shfl_xor_f64(a2, m);
        const int j1 = __shfl_xor(i1, m, 64);
        const double n2 = fmin(fmax(a1, b1), fmin(a2, b2));
        if (b1 < a1 || (b1 == a1 && j1 < i1)) i1 = j1;
        a1 = fmin(a1, b1); a2 = n2;
      }
      for (int i = lane; i < Lo; i += 64) S.q[i] = fmin(0.0 + S.mo[i], diff + (i == i1 ? a2 : a1));
    } else if (side == 0) {
      for (int a0 = 0; a0 < R; a0 += 16) {
        double v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = LPMP_INF;
        for (int b0 = 0; b0 < C; b0 += 64) {
          const int b = b0 + lane;
          const bool cb = b < C;
          const double m = cb ? S.mo[b] : 0.0;
          double t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = (cb && a0 + r < R) ? ld_stream<NT>(T + (int64_t)(a0 + r) * C + b) : LPMP_INF;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmin(v[r], t[r] + m);
        }
        const double full = transpose_min16(v, lane);
        if (lane < 16 && a0 + my_row < R) S.q[a0 + my_row] = full;
      }
    } else {
      for (int b0 = 0; b0 < C; b0 += 64) {
        const int b = b0 + lane;
        const bool cb = b < C;
        double v = LPMP_INF;
        for (int a0 = 0; a0 < R; a0 += 16) {
          double t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = (cb && a0 + r < R) ? ld_stream<NT>(T + (int64_t)(a0 + r) * C + b) : LPMP_INF;
#pragma unroll
          for (int r = 0; r < 16; ++r) { const double m = a0 + r < R ? S.mo[a0 + r] : 0.0; v = fmin(v, t[r] + m); }
        }
        if (cb) S.q[b] = v;
      }
    }
    wave_sync();
    double pb = LPMP_INF;                              // peer's bound after this receive
    for (int i = lane, j = 0; i < Lr; i += 64, ++j) {
      const double msv = j == 0 ? ms_pre[0] : j == 1 ? ms_pre[1] : ld_dual<A>(ms + i), qv = S.q[i];
      const double delta = msv + qv;                   // omega = 1: delta = min-marginal
      S.theta[i] += delta;
      const double mn = msv - delta;
      st_dual<A>(ms + i, mn);
      pb = fmin(pb, mn + qv);
    }
    pb = wave_min(pb);
    if (lane == 0) st_lb<A>(lb + op.peer, pb);
    wave_sync();
  }
  if ((flags & SWEEP_PRIMAL) && (rec.kind_flags & UPD_PRIMAL)) {   // first minimiser of theta after the receives
    double bv = LPMP_INF; int bi = 0x7fffffff;
    for (int i = lane; i < Lr; i += 64) { const double x = S.theta[i]; if (bi == 0x7fffffff || x < bv) { bv = x; bi = i; } }
    const double mn = wave_min(bv);
    int cand = (bi != 0x7fffffff && bv == mn) ? bi : 0x7fffffff;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cand = min(cand, __shfl_xor(cand, m, 64));
    if (lane == 0) store_label(primal, rec.factor, Lr, cand);
  }
  // sends from the state after the receives (kept in q); a lane always owns the same elements: no barrier needed
  for (int i = lane; i < Lr; i += 64) S.q[i] = S.theta[i];
  for (int k = 0; k < rec.n_send; ++k) {
    const Op op = uni_op(ops[rec.op_begin + rec.n_recv + k]);
    double* ms = dual + op.peer_dual + (((op.info >> 5) & 1) ? op.pd0 : 0);
    for (int i = lane; i < Lr; i += 64) {
      const double delta = op.omega * S.q[i];
      st_dual<A>(ms + i, ld_dual<A>(ms + i) + delta);
      S.theta[i] -= delta;
    }
    if (lane == 0) st_lb<A>(lb + op.peer, LPMP_NAN);
  }
  if (flags & SWEEP_RESIDUAL) {
    double residual = 0.0;
    for (int k = 0; k < rec.n_send; ++k) {
      const Op op = uni_op(ops[rec.op_begin + rec.n_recv + k]);
      double* ms = dual + op.peer_dual + (((op.info >> 5) & 1) ? op.pd0 : 0);
      residual += op.omega;
      for (int i = lane; i < Lr; i += 64) {
        const double delta = residual * S.theta[i];
        st_dual<A>(ms + i, ld_dual<A>(ms + i) + delta);
        S.theta[i] -= delta;
      }
    }
  }
  double ob = LPMP_INF;
  for (int i = lane; i < Lr; i += 64) { const double x = S.theta[i]; st_dual<A>(own_g + i, x); ob = fmin(ob, x); }
  ob = wave_min(ob);
  if (lane == 0) st_lb<A>(lb + rec.factor, ob);
}
// (five waves per SIMD: the body needs 96-101 VGPRs depending on small things, and the kernel is bound by round trips per wave)
template <bool NT>
__global__ void __launch_bounds__(64 * BIG_WAVES, 5)
sweep_dense_big_kernel(const UpdRec* __restrict__ recs, const Op* __restrict__ ops, double* __restrict__ dual,
                       const double* __restrict__ cdata, double* __restrict__ lb, int32_t* __restrict__ primal,
                       int64_t first, int64_t count, int flags) {
  dense_big_body<NT, ACC_PLAIN>(recs, ops, dual, cdata, lb, primal, first, count, flags, (int64_t)blockIdx.x);
}
// -------------------------------------------------------------------------------------------------
// Updated pairwise factors (dense or Potts), packed form (classes KC_PW_4..32; `right` / `full` schedules, e.g. MPLP-style
// FMCs): the factor is on the right of all its (unary-pairwise) messages.  A receive pulls the whole unary in
// (delta = 1 * theta_u), the sends push omega * min-marginal back.  Same lane layout as sweep_dense_pk_kernel with
// run-time dims (d0 x d1 <= L x L): record + ops in one packet, the factor's OWN table requested right away and
// read ONCE for both sides' min-marginals of the snapshot.
// -------------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256)
sweep_pairwise_pk_kernel(const Op* __restrict__ packets, double* __restrict__ dual, const double* __restrict__ cdata,
                         double* __restrict__ lb, int64_t count, int stride) {
  constexpr int G = DenseCfg<L>::G;
  constexpr int CL = L / 2, RPL = 2 * G / L, NL = L / RPL, GPB = 256 / G;
  constexpr int PIECES = 3 * (1 + PW_MAX_OPS);
  __shared__ double2_t lds_pk[GPB][PIECES];
  __shared__ double lds_m1[GPB][L];
  __shared__ double lds_m2[GPB][L];
  __shared__ double lds_q0[GPB][L];
  __shared__ double lds_q1[GPB][L];
  const int grp = threadIdx.x / G, g = threadIdx.x % G;
  const int64_t idx = (int64_t)blockIdx.x * GPB + grp;
  const bool live = idx < count;
  const int c2 = g % CL, rl = g / CL;
  load_packet<G>(lds_pk[grp], packets, nullptr, nullptr, idx, stride, live, g);
  const UpdRec* hdr = reinterpret_cast<const UpdRec*>(&lds_pk[grp][0]);
  const Op* lop = reinterpret_cast<const Op*>(&lds_pk[grp][3]);
  const int n_recv = live ? (int)hdr->n_recv : 0;
  const int n_send = live ? (int)hdr->n_send : 0;
  const int R = live ? hdr->d0 : 0, C = live ? hdr->d1 : 0;
  double* own_g = dual + (live ? hdr->dual_off : 0);
  const double* T = cdata + (live ? hdr->const_off : 0);
  double2_t t[NL];
  if (live && (hdr->kind_flags & 15) == LPMP_F_PAIRWISE_POTTS) {   // diff * [a != b] (reference test/potts_factor.cpp:34-36)
    const double diff = T[0];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int row = i * RPL + rl;
      t[i].x = (row < R && 2 * c2 < C) ? (row == 2 * c2 ? 0.0 : diff) : LPMP_INF;
      t[i].y = (row < R && 2 * c2 + 1 < C) ? (row == 2 * c2 + 1 ? 0.0 : diff) : LPMP_INF;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int row = i * RPL + rl;
      const double* Tr = T + (int64_t)row * C + 2 * c2;
      t[i].x = (row < R && 2 * c2 < C) ? Tr[0] : LPMP_INF;
      t[i].y = (row < R && 2 * c2 + 1 < C) ? Tr[1] : LPMP_INF;
    }
  }
  double m1 = g < R ? own_g[g] : 0.0;            // message vector of side 0, element g
  double m2 = g < C ? own_g[R + g] : 0.0;        // message vector of side 1, element g
  // receives: delta = 1 * theta_u; the unary gives it up, the factor's vector of that side takes it
#pragma unroll
  for (int k = 0; k < PW_MAX_OPS; ++k) {
    if (k < n_recv) {
      const Op& o = lop[k];
      const int side = (o.info >> 5) & 1;
      if (g < o.len) {
        double* th = dual + o.peer_dual + g;
        const double v = *th;
        const double dl = 1.0 * v;
        *th = v + -1.0 * dl;
        if (side == 0) m1 += +1.0 * dl; else m2 += +1.0 * dl;
      }
      if (g == 0) lb[o.peer] = LPMP_NAN;
    }
  }
  // both min-marginal parts of the state after the receives: q0[a] = min_b T[a][b] + m2[b], q1[b] = min_a T[a][b] + m1[a]
  const double m1s = m1, m2s = m2;
  if (g < L) { lds_m1[grp][g] = m1s; lds_m2[grp][g] = m2s; }
  wave_sync();
  {
    const double2_t mv = *reinterpret_cast<const double2_t*>(&lds_m2[grp][2 * c2]);
    double vx = LPMP_INF, vy = LPMP_INF;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      double v = fmin(t[i].x + mv.x, t[i].y + mv.y);
      v = row_allreduce_min<CL>(v);
      if (c2 == 0) lds_q0[grp][i * RPL + rl] = v;
      const double m1v = lds_m1[grp][i * RPL + rl];
      vx = fmin(vx, t[i].x + m1v);
      vy = fmin(vy, t[i].y + m1v);
    }
#pragma unroll
    for (int m = G / 2; m >= CL; m >>= 1) { vx = fmin(vx, shfl_xor_f64(vx, m)); vy = fmin(vy, shfl_xor_f64(vy, m)); }
    if (rl == 0) { lds_q1[grp][2 * c2] = vx; lds_q1[grp][2 * c2 + 1] = vy; }
  }
  wave_sync();
  const double q0 = g < L ? lds_q0[grp][g] : 0.0, q1 = g < L ? lds_q1[grp][g] : 0.0;
  // sends: delta = omega * min-marginal of the snapshot
#pragma unroll
  for (int k = 0; k < PW_MAX_OPS; ++k) {
    if (k < n_send) {
      const Op& o = lop[n_recv + k];
      const int side = (o.info >> 5) & 1;
      if (g < o.len) {
        const double dl = o.omega * (side == 0 ? m1s + q0 : m2s + q1);
        double* th = dual + o.peer_dual + g;
        *th += +1.0 * dl;
        if (side == 0) m1 += -1.0 * dl; else m2 += -1.0 * dl;
      }
      if (g == 0) lb[o.peer] = LPMP_NAN;
    }
  }
  if (g < R) own_g[g] = m1;
  if (g < C) own_g[R + g] = m2;
  if (live && g == 0) lb[hdr->factor] = LPMP_NAN;
}

// -------------------------------------------------------------------------------------------------
// Lower bound (reference LP::LowerBound, LP_MP.h:1507-1518): per-factor bound, then a fixed-order sum.
// -------------------------------------------------------------------------------------------------
struct LbRec { int64_t dual_off; int64_t const_off; int32_t d0, d1; int32_t kind_flags; int32_t pad; };

// one wave per factor, any kind
__global__ void __launch_bounds__(256)
factor_lb_kernel(const LbRec* __restrict__ recs, const double* __restrict__ dual, const double* __restrict__ cdata,
                 double* __restrict__ out, int64_t count) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t f = (int64_t)blockIdx.x * 4 + wave;
  if (f >= count) return;
  const LbRec r = recs[f];
  const int kind = r.kind_flags & 15, flags = r.kind_flags >> 4;
  const double* d = dual + r.dual_off;
  double lb;
  if (kind == LPMP_F_VECTOR) {
    double v = LPMP_INF;
    for (int i = lane; i < r.d0; i += 64) v = fmin(v, d[i]);
    lb = wave_min(v);
    if ((flags & LPMP_FF_IMPLICIT_ORIGIN) && 0.0 < lb) lb = 0.0;
  } else {
    // min_a ( m1[a] + min_b (T[a][b] + m2[b]) ): lanes sweep the table row-major (coalesced)
    const int d0 = r.d0, d1 = r.d1;
    double best = LPMP_INF;
    if (kind == LPMP_F_PAIRWISE_DENSE) {
      const double* T = cdata + r.const_off;
      for (int a = 0; a < d0; ++a) {
        double v = LPMP_INF;
        for (int b = lane; b < d1; b += 64) v = fmin(v, T[(int64_t)a * d1 + b] + d[d0 + b]);
        v = wave_min(v);
        best = fmin(best, d[a] + v);
      }
    } else {
      const double diff = cdata[r.const_off];
      for (int a = 0; a < d0; ++a) {
        double v = LPMP_INF;
        for (int b = lane; b < d1; b += 64) v = fmin(v, (a == b ? 0.0 : diff) + d[d0 + b]);
        v = wave_min(v);
        best = fmin(best, d[a] + v);
      }
    }
    lb = best;
  }
  if (lane == 0) out[f] = lb;
}

// the same for an explicit list of factors (the ones whose tracked bound is stale)
__global__ void __launch_bounds__(256)
factor_lb_list_kernel(const LbRec* __restrict__ recs, const double* __restrict__ dual, const double* __restrict__ cdata,
                      double* __restrict__ out, const int32_t* __restrict__ list, int64_t count) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= count) return;
  const int32_t f = list[i];
  const LbRec r = recs[f];
  const int kind = r.kind_flags & 15, flags = r.kind_flags >> 4;
  const double* d = dual + r.dual_off;
  double lb;
  if (kind == LPMP_F_VECTOR) {
    double v = LPMP_INF;
    for (int k = lane; k < r.d0; k += 64) v = fmin(v, d[k]);
    lb = wave_min(v);
    if ((flags & LPMP_FF_IMPLICIT_ORIGIN) && 0.0 < lb) lb = 0.0;
  } else {
    const int d0 = r.d0, d1 = r.d1;
    double best = LPMP_INF;
    for (int a = 0; a < d0; ++a) {
      double v = LPMP_INF;
      for (int b = lane; b < d1; b += 64) v = fmin(v, pw_cost(cdata, r.const_off, kind, d1, a, b) + d[d0 + b]);
      v = wave_min(v);
      best = fmin(best, d[a] + v);
    }
    lb = best;
  }
  if (lane == 0) out[f] = lb;
}

// indices of the factors whose tracked bound is NaN
__global__ void __launch_bounds__(256)
lb_collect_stale_kernel(const double* __restrict__ lb, int64_t n, int32_t* __restrict__ list, unsigned long long* __restrict__ counter) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (lb[i] != lb[i]) list[atomicAdd(counter, 1ull)] = (int32_t)i;
}

// dense L x L pairwise bound with the streaming layout of sweep_dense_kernel (G lanes per factor)
template <int L>
__global__ void __launch_bounds__(256)
dense_lb_kernel(const LbRec* __restrict__ recs, const double* __restrict__ dual, const double* __restrict__ cdata,
                double* __restrict__ out, int64_t first, int64_t count) {
  constexpr int G = DenseCfg<L>::G;
  constexpr int CL = L / 2, RPL = 2 * G / L, NL = L / RPL, GPB = 256 / G;
  __shared__ double lds_m[GPB][2 * L];
  const int grp = threadIdx.x / G, g = threadIdx.x % G;
  const int64_t idx = (int64_t)blockIdx.x * GPB + grp;
  const bool live = idx < count;
  const int c2 = g % CL, rl = g / CL;
  LbRec r;
  if (live) r = recs[first + idx]; else { r.dual_off = 0; r.const_off = 0; }
  const double* T = cdata + r.const_off;
  const double* d = dual + r.dual_off;
  double2_t t[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) t[i] = live ? *reinterpret_cast<const double2_t*>(T + (int64_t)i * 2 * G + 2 * g) : double2_t{0.0, 0.0};
  for (int i = g; i < 2 * L; i += G) lds_m[grp][i] = live ? d[i] : 0.0;
  wave_sync();
  const double2_t m2 = *reinterpret_cast<const double2_t*>(&lds_m[grp][L + 2 * c2]);
  double best = LPMP_INF;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    double v = fmin(t[i].x + m2.x, t[i].y + m2.y);
    v = row_allreduce_min<CL>(v);
    best = fmin(best, lds_m[grp][i * RPL + rl] + v);
  }
#pragma unroll
  for (int m = G / 2; m >= CL; m >>= 1) best = fmin(best, shfl_xor_f64(best, m));
  if (live && g == 0) out[first + idx] = best;
}

// ---- primal rounding: bookkeeping around the sweep (engine.cpp) ------------------------------------
// conditionally_init_primal of every factor a primal pass touches (reference factors_messages.hxx:3302-3309; all of
// them carry the same time stamp, so the host decides whether this runs)
__global__ void __launch_bounds__(256)
primal_init_kernel(const PrimalInit* __restrict__ list, int64_t n, int32_t* __restrict__ primal) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const PrimalInit r = list[i];
  primal[2 * (int64_t)r.f] = r.a;
  primal[2 * (int64_t)r.f + 1] = r.b;
}
// propagate_primal_through_messages of the rounded unaries: right.primal_[side] = left.primal when that is set
// (MessageContainer::ComputeRightFromLeftPrimal, reference factors_messages.hxx:1313-1328)
__global__ void __launch_bounds__(256)
primal_propagate_kernel(const PrimalLink* __restrict__ links, int64_t n, int32_t* __restrict__ primal) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const PrimalLink l = links[i];
  const int32_t x = primal[2 * (int64_t)l.u];
  if (x < l.dim) primal[2 * (int64_t)l.p + l.side] = x;
}
// LP::CheckPrimalConsistency (reference LP_MP.h:1067-1082): every message's two sides agree
__global__ void __launch_bounds__(256)
primal_check_kernel(const PrimalLink* __restrict__ links, int64_t n, const int32_t* __restrict__ primal, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const PrimalLink l = links[i];
  if (primal[2 * (int64_t)l.u] != primal[2 * (int64_t)l.p + l.side]) *bad = 1;
}
// FactorContainer::EvaluatePrimal per factor: reparametrised cost at the factor's primal, +inf when a side is unset
__global__ void __launch_bounds__(256)
primal_cost_kernel(const LbRec* __restrict__ recs, const double* __restrict__ dual, const double* __restrict__ cdata,
                   const int32_t* __restrict__ primal, double* __restrict__ out, int64_t count) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= count) return;
  const LbRec r = recs[f];
  const int kind = r.kind_flags & 15;
  const double* d = dual + r.dual_off;
  const int a = primal[2 * f], b = primal[2 * f + 1];
  double c;
  if (kind == LPMP_F_VECTOR) c = a < r.d0 ? d[a] : LPMP_INF;
  else if (a >= r.d0 || b >= r.d1) c = LPMP_INF;
  else c = pw_cost(cdata, r.const_off, kind, r.d1, a, b) + d[a] + d[r.d0 + b];
  out[f] = c;
}

// deterministic two-stage sum: block b sums a fixed contiguous slice in a fixed tree order
__global__ void __launch_bounds__(256)
sum_stage_kernel(const double* __restrict__ in, double* __restrict__ out, int64_t n, int64_t per_block) {
  __shared__ double sh[256];
  const int64_t b0 = (int64_t)blockIdx.x * per_block;
  const int64_t b1 = b0 + per_block < n ? b0 + per_block : n;
  double s = 0.0;
  for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) s += in[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// counter-based generator of the synthetic workloads: bit-identical to lp_mp_amd.synthetic.u01
__global__ void synth_fill_kernel(double* __restrict__ out, int64_t n, uint64_t seed, uint64_t first) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t z = seed + (first + (uint64_t)i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    out[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
  }
}

// ---- launch wrappers (called from engine.cpp) -----------------------------------------------------
void launch_sweep(int kclass, const UpdRec* recs, const Op* ops, double* dual, const double* cdata, const int32_t* tabs,
                  double* lb, int32_t* primal, const int32_t* pw_unary, int64_t first, int64_t count, int flags, hipStream_t s) {
  if (count <= 0) return;
  auto blocks = [&](int per_block) { return dim3((unsigned)((count + per_block - 1) / per_block)); };
  switch (kclass) {
    case KC_DENSE_32: hipLaunchKernelGGL(sweep_dense_kernel<32>, blocks(256 / DenseCfg<32>::G), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_DENSE_16: hipLaunchKernelGGL(sweep_dense_kernel<16>, blocks(256 / DenseCfg<16>::G), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_DENSE_8: hipLaunchKernelGGL(sweep_dense_kernel<8>, blocks(256 / DenseCfg<8>::G), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_DENSE_4: hipLaunchKernelGGL(sweep_dense_kernel<4>, blocks(256 / DenseCfg<4>::G), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_POTTS_32: hipLaunchKernelGGL(sweep_potts_kernel<32>, blocks(256 / 32), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_POTTS_16: hipLaunchKernelGGL(sweep_potts_kernel<16>, blocks(256 / 16), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_POTTS_8: hipLaunchKernelGGL(sweep_potts_kernel<8>, blocks(256 / 8), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_POTTS_4: hipLaunchKernelGGL(sweep_potts_kernel<4>, blocks(256 / 4), dim3(256), 0, s, recs, ops, dual, cdata, lb, primal, first, count, flags); break;
    case KC_DENSE_BIG:
      if (flags & SWEEP_NT) hipLaunchKernelGGL(sweep_dense_big_kernel<true>, blocks(BIG_WAVES), dim3(64 * BIG_WAVES), big_lds_bytes(flags), s, recs, ops, dual, cdata, lb, primal, first, count, flags);
      else hipLaunchKernelGGL(sweep_dense_big_kernel<false>, blocks(BIG_WAVES), dim3(64 * BIG_WAVES), big_lds_bytes(flags), s, recs, ops, dual, cdata, lb, primal, first, count, flags);
      break;
    case KC_SMALL: hipLaunchKernelGGL(sweep_generic_kernel<1>, blocks(GenCtx<1>::FPB), dim3(GenCtx<1>::THREADS), 0, s, recs, ops, dual, cdata, tabs, lb, primal, pw_unary, first, count, flags); break;
    default: hipLaunchKernelGGL(sweep_generic_kernel<64>, blocks(GEN_WAVES), dim3(64 * GEN_WAVES), 0, s, recs, ops, dual, cdata, tabs, lb, primal, pw_unary, first, count, flags); break;
  }
}

bool launch_sweep_packed(int kclass, const Op* packets, const UpdRec* recs, const Op* ops, int stride, double* dual, const double* cdata,
                         double* lb, int32_t* primal, int64_t count, int flags, hipStream_t s) {
  if (count <= 0) return true;
  if (stride > 1 + PK_MAX_OPS) return false;
  auto blocks = [&](int per_block) { return dim3((unsigned)((count + per_block - 1) / per_block)); };
  const bool nt = (flags & SWEEP_NT) != 0;
  if (kc_is_pw(kclass)) {
    // (the residual rule recomputes the min-marginals after every send: the op-by-op generic kernel does that)
    if ((flags & SWEEP_RESIDUAL) || stride <= 0) return false;
#define PWK_LAUNCH(LL) hipLaunchKernelGGL((sweep_pairwise_pk_kernel<LL>), blocks(256 / DenseCfg<LL>::G), dim3(256), 0, s, packets, dual, cdata, lb, count, stride)
    switch (kclass) {
      case KC_PW_32: PWK_LAUNCH(32); return true;
      case KC_PW_16: PWK_LAUNCH(16); return true;
      case KC_PW_8: PWK_LAUNCH(8); return true;
      default: PWK_LAUNCH(4); return true;
    }
#undef PWK_LAUNCH
  }
#define PK_LAUNCH1(LL, KK, NTT) hipLaunchKernelGGL((sweep_dense_pk_kernel<LL, KK, false, NTT>), blocks(256 / DenseCfg<LL>::G), dim3(256), 0, s, packets, recs, ops, dual, cdata, lb, primal, count, stride, flags)
#define PK_LAUNCH(LL, KK) do { if (nt) PK_LAUNCH1(LL, KK, true); else PK_LAUNCH1(LL, KK, false); } while (0)
  switch (kclass) {
    // receives in flight per lane group: 2 at 16 / 32 labels (1 and 4 measured slower on C3), 4 below
    case KC_DENSE_32: PK_LAUNCH(32, 2); return true;
    case KC_DENSE_16: PK_LAUNCH(16, LPMP_KMAX16); return true;
    case KC_DENSE_8: PK_LAUNCH(8, 4); return true;
    case KC_DENSE_4: PK_LAUNCH(4, 4); return true;
#define PPK_LAUNCH1(LL, NTT) hipLaunchKernelGGL((sweep_potts_pk_kernel<LL, false, NTT>), blocks(256 / LL), dim3(256), 0, s, packets, recs, ops, dual, cdata, lb, primal, count, stride, flags)
#define PPK_LAUNCH(LL) do { if (nt) PPK_LAUNCH1(LL, true); else PPK_LAUNCH1(LL, false); } while (0)
    case KC_POTTS_32: PPK_LAUNCH(32); return true;
    case KC_POTTS_16: PPK_LAUNCH(16); return true;
    case KC_POTTS_8: PPK_LAUNCH(8); return true;
    case KC_POTTS_4: PPK_LAUNCH(4); return true;
#undef PPK_LAUNCH
#undef PPK_LAUNCH1
// (run-time dims: rows are not line-aligned, consecutive 8-B loads share lines — non-temporal loads cost 9 % there)
#define VPK_LAUNCH(LL, KK) hipLaunchKernelGGL((sweep_dense_pk_kernel<LL, KK, true, false>), blocks(256 / DenseCfg<LL>::G), dim3(256), 0, s, packets, recs, ops, dual, cdata, lb, primal, count, stride, flags)
    case KC_DENSE_V32: VPK_LAUNCH(32, 2); return true;
    case KC_DENSE_V16: VPK_LAUNCH(16, 2); return true;
    case KC_DENSE_V8: VPK_LAUNCH(8, 4); return true;
    case KC_DENSE_V4: VPK_LAUNCH(4, 4); return true;
#undef VPK_LAUNCH
#define VPPK_LAUNCH(LL) hipLaunchKernelGGL((sweep_potts_pk_kernel<LL, true, false>), blocks(256 / LL), dim3(256), 0, s, packets, recs, ops, dual, cdata, lb, primal, count, stride, flags)
    case KC_POTTS_V32: VPPK_LAUNCH(32); return true;
    case KC_POTTS_V16: VPPK_LAUNCH(16); return true;
    case KC_POTTS_V8: VPPK_LAUNCH(8); return true;
    case KC_POTTS_V4: VPPK_LAUNCH(4); return true;
#undef VPPK_LAUNCH
    default: return false;
  }
#undef PK_LAUNCH
#undef PK_LAUNCH1
}

// chain executor: one persistent launch for a deep single-class schedule; grid = what is resident at once (more
// workgroups would only queue behind the running ones).  Returns false for a class without a chain kernel.
template <class K>
static unsigned chain_grid(K kernel, int n_tickets, int threads = 256) {
  // compute units of the CURRENT device (a process may hold engines on several devices): cached per ordinal
  static int n_cu_of[64] = {0};
  int dev = 0; (void)hipGetDevice(&dev);
  int& n_cu = n_cu_of[dev >= 0 && dev < 64 ? dev : 0];
  if (n_cu == 0) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; n_cu = v; }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  const long cap = (long)n_cu * per_cu;
  return (unsigned)(n_tickets < cap ? n_tickets : cap);
}
void debug_set_level_trace(long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_level_trace), &p, sizeof(p)); }
bool launch_level_loop(int kclass, int flags, const void* launches, int n_launches, double* dual, const double* cdata,
                       const int32_t* tabs, double* lb, hipStream_t s) {
  const ChainLaunch* ln = static_cast<const ChainLaunch*>(launches);
  if (kclass == KC_SMALL) hipLaunchKernelGGL(level_loop_kernel<1>, dim3(1), dim3(64 * (LL_WAVES + 1)), 0, s, ln, n_launches, dual, cdata, tabs, lb, flags);
  else if (kclass == KC_GENERIC) hipLaunchKernelGGL(level_loop_kernel<64>, dim3(1), dim3(GenCtx<64>::THREADS), 0, s, ln, n_launches, dual, cdata, tabs, lb, flags);
  else return false;
  return true;
}
bool launch_chain(int kclass, int flags, const void* chain_args, const void* launches, double* dual, const double* cdata,
                  const int32_t* tabs, double* lb, int32_t* primal, hipStream_t s) {
  const ChainArgs ca = *static_cast<const ChainArgs*>(chain_args);
  const ChainLaunch* ln = static_cast<const ChainLaunch*>(launches);
  const bool nt = (flags & SWEEP_NT) != 0;
#define CHAIN_LAUNCH2(LL, KK, VV, NTT, MM) do { auto k = chain_dense_pk_kernel<LL, KK, VV, NTT, MM>; \
    hipLaunchKernelGGL(k, dim3(chain_grid(k, ca.n_tickets)), dim3(256), 0, s, ca, ln, dual, cdata, lb, primal, flags); } while (0)
#define CHAIN_LAUNCH1(LL, KK, VV, NTT) CHAIN_LAUNCH2(LL, KK, VV, NTT, false)
  // (a mailbox chain is a deep schedule: latency-bound, no streaming variant)
#define CHAIN_LAUNCH(LL, KK) do { if (ca.mailbox) CHAIN_LAUNCH2(LL, KK, false, false, true); else if (nt) CHAIN_LAUNCH1(LL, KK, false, true); else CHAIN_LAUNCH1(LL, KK, false, false); } while (0)
#define CHAIN_LAUNCH_V(LL, KK) do { if (ca.mailbox) CHAIN_LAUNCH2(LL, KK, true, false, true); else CHAIN_LAUNCH1(LL, KK, true, false); } while (0)
  switch (kclass) {
    case KC_GENERIC: { auto k = chain_generic_kernel<64>; hipLaunchKernelGGL(k, dim3(chain_grid(k, ca.n_tickets, GenCtx<64>::THREADS)), dim3(GenCtx<64>::THREADS), 0, s, ca, ln, dual, cdata, tabs, lb, flags); return true; }
    case KC_SMALL: { auto k = chain_generic_kernel<1>; hipLaunchKernelGGL(k, dim3(chain_grid(k, ca.n_tickets, GenCtx<1>::THREADS)), dim3(GenCtx<1>::THREADS), 0, s, ca, ln, dual, cdata, tabs, lb, flags); return true; }
#ifndef LPMP_MBOX_KMAX32           // experiments: receives in flight per record of the 32-label mailbox chain
#define LPMP_MBOX_KMAX32 2
#endif
    case KC_DENSE_32: if (ca.mailbox) CHAIN_LAUNCH2(32, LPMP_MBOX_KMAX32, false, false, true); else CHAIN_LAUNCH(32, 2); return true;
    case KC_DENSE_16: CHAIN_LAUNCH(16, 2); return true;
    case KC_DENSE_8: CHAIN_LAUNCH(8, 4); return true;
    case KC_DENSE_4: CHAIN_LAUNCH(4, 4); return true;
    case KC_DENSE_V32: CHAIN_LAUNCH_V(32, 2); return true;
    case KC_DENSE_V16: CHAIN_LAUNCH_V(16, 2); return true;
    case KC_DENSE_V8: CHAIN_LAUNCH_V(8, 4); return true;
    case KC_DENSE_V4: CHAIN_LAUNCH_V(4, 4); return true;
#define CHAIN_POTTS2(LL, VV, MM) do { auto k = chain_potts_pk_kernel<LL, VV, MM>; \
    hipLaunchKernelGGL(k, dim3(chain_grid(k, ca.n_tickets)), dim3(256), 0, s, ca, ln, dual, cdata, lb, primal, flags); } while (0)
#define CHAIN_POTTS(LL, VV) CHAIN_POTTS2(LL, VV, false)
#define CHAIN_POTTS_X(LL) do { if (ca.mailbox) CHAIN_POTTS2(LL, false, true); else CHAIN_POTTS2(LL, false, false); } while (0)
#define CHAIN_POTTS_XV(LL) do { if (ca.mailbox) CHAIN_POTTS2(LL, true, true); else CHAIN_POTTS2(LL, true, false); } while (0)
    case KC_POTTS_32: CHAIN_POTTS_X(32); return true;
    case KC_POTTS_16: CHAIN_POTTS_X(16); return true;
    case KC_POTTS_8: CHAIN_POTTS_X(8); return true;
    case KC_POTTS_4: CHAIN_POTTS_X(4); return true;
    case KC_POTTS_V32: CHAIN_POTTS_XV(32); return true;
    case KC_POTTS_V16: CHAIN_POTTS_XV(16); return true;
    case KC_POTTS_V8: CHAIN_POTTS_XV(8); return true;
    case KC_POTTS_V4: CHAIN_POTTS_XV(4); return true;
#undef CHAIN_POTTS_X
#undef CHAIN_POTTS_XV
#undef CHAIN_POTTS
#undef CHAIN_POTTS2
    default: return false;
  }
#undef CHAIN_LAUNCH
#undef CHAIN_LAUNCH_V
#undef CHAIN_LAUNCH1
#undef CHAIN_LAUNCH2
}

void launch_factor_lb(const void* recs, const double* dual, const double* cdata, double* out, int64_t count, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(factor_lb_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, (const LbRec*)recs, dual, cdata, out, count);
}

bool launch_dense_lb(int L, const void* recs, const double* dual, const double* cdata, double* out, int64_t first, int64_t count, hipStream_t s) {
  if (count <= 0) return true;
  auto blocks = [&](int per_block) { return dim3((unsigned)((count + per_block - 1) / per_block)); };
  switch (L) {
    case 32: hipLaunchKernelGGL(dense_lb_kernel<32>, blocks(256 / DenseCfg<32>::G), dim3(256), 0, s, (const LbRec*)recs, dual, cdata, out, first, count); return true;
    case 16: hipLaunchKernelGGL(dense_lb_kernel<16>, blocks(256 / DenseCfg<16>::G), dim3(256), 0, s, (const LbRec*)recs, dual, cdata, out, first, count); return true;
    case 8: hipLaunchKernelGGL(dense_lb_kernel<8>, blocks(256 / DenseCfg<8>::G), dim3(256), 0, s, (const LbRec*)recs, dual, cdata, out, first, count); return true;
    default: return false;
  }
}

void launch_lb_collect_stale(const double* lb, int64_t n, int32_t* list, unsigned long long* counter, hipStream_t s) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lb_collect_stale_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lb, n, list, counter);
}
void launch_factor_lb_list(const void* recs, const double* dual, const double* cdata, double* out, const int32_t* list, int64_t count, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(factor_lb_list_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, (const LbRec*)recs, dual, cdata, out, list, count);
}

static dim3 blocks256(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }
void launch_primal_init(const PrimalInit* list, int64_t n, int32_t* primal, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(primal_init_kernel, blocks256(n), dim3(256), 0, s, list, n, primal);
}
void launch_primal_propagate(const PrimalLink* links, int64_t n, int32_t* primal, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(primal_propagate_kernel, blocks256(n), dim3(256), 0, s, links, n, primal);
}
void launch_primal_check(const PrimalLink* links, int64_t n, const int32_t* primal, int* bad, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(primal_check_kernel, blocks256(n), dim3(256), 0, s, links, n, primal, bad);
}
void launch_primal_cost(const void* recs, const double* dual, const double* cdata, const int32_t* primal, double* out, int64_t count, hipStream_t s) {
  if (count > 0) hipLaunchKernelGGL(primal_cost_kernel, blocks256(count), dim3(256), 0, s, (const LbRec*)recs, dual, cdata, primal, out, count);
}

// ---- rows layout (engine.cpp): a dense pairwise factor's table and its two message vectors in ONE contiguous row ----------
// [T (d0 x d1) | m1 (d0) | m2 (d1)] of an engine-private buffer, so that a receive's three reads are one burst (a random
// graph's receives otherwise touch a 2-KiB table and two single 128-byte lines somewhere else: C4).  The packed
// arrays stay the boundary's format (serialize_dual order); these copies move between the two.
//   what 0: build a row (table from the packed constants, vectors from the packed duals)
//   what 1: packed duals -> rows (vectors only)      what 2: rows -> packed duals (vectors only)
struct RowRec { int64_t dual_off, const_off, row_off; int32_t d0, d1; };
__global__ void __launch_bounds__(256)
rows_copy_kernel(const RowRec* __restrict__ recs, int64_t n, const double* __restrict__ cdata, double* __restrict__ dual, double* __restrict__ rows, int what) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const RowRec r = recs[i];
  const int nt = r.d0 * r.d1, nm = r.d0 + r.d1;
  double* row = rows + r.row_off;
  if (what == 0) for (int x = lane; x < nt; x += 64) row[x] = cdata[r.const_off + x];
  if (what == 2) { for (int x = lane; x < nm; x += 64) dual[r.dual_off + x] = row[nt + x]; }
  else { for (int x = lane; x < nm; x += 64) row[nt + x] = dual[r.dual_off + x]; }
}
void launch_rows_copy(const void* recs, int64_t n, const double* cdata, double* dual, double* rows, int what, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(rows_copy_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (const RowRec*)recs, n, cdata, dual, rows, what);
}

void launch_sum_stage(const double* in, double* out, int64_t n, int64_t per_block, int64_t n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(sum_stage_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, in, out, n, per_block);
}

void launch_synth_fill(double* out, int64_t n, uint64_t seed, uint64_t first, hipStream_t s) {
  if (n <= 0) return;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, n, seed, first);
}

int generic_max_dual() { return GEN_MAXD; }
int generic_max_adaptive_sends() { return GEN_ADAPTIVE_SENDS; }

}  // namespace lpmp
