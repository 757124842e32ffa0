// plan.hpp — host-side analysis of a flat LP_MP model: per-factor message lists, update ordering,
// send weights / receive masks for every reparametrisation mode, and the level schedule the HIP
// sweep kernels execute.  Pure C++17 (no HIP): the same code runs in the CPU-only test container.
//
// Reference behaviour restated here (paths relative to /root/reference):
//   message lists  include/factors_messages.hxx:3339-3365, :3402-3419, storage order :2081-2119, :2030-2041
//   FactorUpdated  include/factors_messages.hxx:3125-3140
//   ordering       include/LP_MP.h:730-797, include/topological_sort.hxx:100-144
//   weights        include/LP_MP.h:1232-1415 (anisotropic), :1086-1154 (anisotropic2), :1422-1449 (uniform),
//                  :1489-1505 (full receive mask)
#pragma once
#include <algorithm>
#include <cstdint>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/lpmp_model.h"

namespace lpmp {

struct MsgEntry {          // one element of FactorContainer::get_messages()
  int32_t msg;
  int32_t adjacent;
  uint8_t role;            // 0: this factor is the message's left factor, 1: right
  uint8_t sends, receives, adj_sends, adj_receives;
};

template <class T>
struct Csr {
  std::vector<int64_t> off{0};
  std::vector<T> data;
  int64_t rows() const { return (int64_t)off.size() - 1; }
};

// Device-side records (plain structs shared with kernels.hip) -----------------------------------
enum OpCode : int32_t { OP_UP = 0, OP_LABELING = 1, OP_MINNORM = 2 };
constexpr int32_t OP_HAS_IMPROVEMENT = 1 << 12;   // Op::info: the message op defines send_message_to_*_improvement
// Op::info, only in the PACKET copy of an op of a chain launch with CHAIN_LAUNCH_MAILBOX (plan.cpp, chain plans): the
// message vector travels through the chain's mailbox.  A send: peer_const = mailbox row the new vector is also written to;
// a receive: the bits of omega = mailbox row (int64) the OTHER side's vector is polled from instead of the dual array.
constexpr int32_t OP_MAILBOX = 1 << 13;

struct alignas(16) UpdRec {   // one updated factor
  int64_t dual_off;   // own dual start
  int64_t const_off;  // own const start (pairwise) or -1
  int32_t d0, d1;     // own dims (vector: d0 = n, d1 = 0)
  int32_t op_begin;   // first op
  int16_t n_recv, n_send;
  int32_t factor;
  int32_t kind_flags; // own kind (bits 0-3) | flags << 4
};
static_assert(sizeof(UpdRec) == 48, "UpdRec layout");

struct alignas(16) Op {       // one active receive or send
  int64_t peer_dual;  // peer dual start
  int64_t peer_const; // peer const start (pairwise peer), or offset of the match table in tab_data (labeling)
  double omega;       // send weight (receives: 1.0)
  int32_t info;       // opcode | role<<4 | side<<5 | right_implicit_origin<<6 | peer_implicit_origin<<7 | peer_kind<<8 | has_improvement<<12
  int32_t pd0;        // peer dim0
  int32_t pd1;        // peer dim1 (dense pairwise peer) / n_left of the table (labeling)
  int32_t peer;       // peer factor index (slot of its tracked lower bound)
  int32_t len;        // message length (= dim of the left factor's variable)
  int32_t pad;
};
static_assert(sizeof(Op) == 48, "Op layout");

// kernel classes: which kernel runs an updated factor
enum KClass : int32_t {
  KC_GENERIC = 0,
  KC_DENSE_4, KC_DENSE_8, KC_DENSE_16, KC_DENSE_32,
  KC_POTTS_4, KC_POTTS_8, KC_POTTS_16, KC_POTTS_32,
  // any label count <= the padded width (runtime dims, also rectangular d0 x d1 tables for the dense classes)
  KC_DENSE_V4, KC_DENSE_V8, KC_DENSE_V16, KC_DENSE_V32,
  KC_POTTS_V4, KC_POTTS_V8, KC_POTTS_V16, KC_POTTS_V32,
  // one wave per unary, dense tables of any dims up to BIG_MAX_LABELS streamed in 16-row blocks
  KC_DENSE_BIG,
  // one LANE per updated factor: tiny factors of any kind (every dual size / message length <= SMALL_MAXD)
  KC_SMALL,
  // UPDATED dense pairwise factors (`right` / `full` schedules: the factor pulls its unaries in and sends both
  // min-marginals back), dims <= the padded width, at most PW_MAX_OPS ops: packet form, one read of the table
  KC_PW_4, KC_PW_8, KC_PW_16, KC_PW_32,
  KC_COUNT
};
constexpr int BIG_MAX_LABELS = 512;
constexpr int SMALL_MAXD = 8;
constexpr int PW_MAX_OPS = 6;
constexpr bool kc_is_pw(int kclass) { return kclass >= KC_PW_4 && kclass <= KC_PW_32; }
// lanes-per-vector width of a packed fast class (0: generic / streaming class)
constexpr int kc_width(int kclass) {
  if (kclass >= KC_PW_4 && kclass <= KC_PW_32) return 4 << (kclass - KC_PW_4);
  return (kclass == KC_GENERIC || kclass >= KC_DENSE_BIG) ? 0 : 4 << ((kclass - 1) % 4);
}
constexpr bool kc_is_dense(int kclass) { return (kclass >= KC_DENSE_4 && kclass <= KC_DENSE_32) || (kclass >= KC_DENSE_V4 && kclass <= KC_DENSE_V32); }
constexpr bool kc_is_var(int kclass) { return kclass >= KC_DENSE_V4 && kclass <= KC_POTTS_V32; }

struct LevelRange {            // one kernel launch: a range of UpdRec indices of one level and class
  int32_t kclass; int64_t begin, end;
  int32_t level = 0;                            // 1-based dependent step this launch belongs to
  int64_t n_recv = 0, n_send = 0, bytes = 0;   // active receives / sends / algorithmic bytes of the range
  // packed form (fast classes with few ops per factor): factor i of the range has its UpdRec in slot
  // pk_begin + i*stride of Schedule::packets and its ops in the following slots -> one coalesced load, no
  // dependent rec -> ops hop.  stride 0: not packed.
  int32_t stride = 0; int64_t pk_begin = 0;
  int32_t max_dim = 0;                          // largest label count of any vector or table side the launch's records touch
};
constexpr int PK_MAX_OPS = 8;                 // packets hold at most this many ops per factor
// launches whose factors have more ops than that (but at most this many: the LDS slab of a lane group) run the
// same kernels in INDIRECT mode (LevelRange::stride < 0): record from recs[], then all its ops from ops[] in one
// coalesced load — two dependent hops instead of one per op
constexpr int pk_indirect_cap(int labels) { return labels >= 16 ? 32 : labels >= 8 ? 16 : 8; }
// the dense classes at 16 labels hold 64 ops: 50 KB of LDS per workgroup of 16 records = three workgroups per CU, the
// occupancy the dense kernel's registers allow anyway — so the hubs of a random graph of mean degree 10 stay on the packed
// kernel (as launches of their own on the streaming kernel they cost C4 1.5 of 13.4 ms per pass, profiles/r03_c4b_*).
// The Potts kernels (more waves per SIMD) keep the smaller slab.
#ifndef LPMP_PK_DENSE_CAP16          // experiments (tools/build_variant.sh)
#define LPMP_PK_DENSE_CAP16 64
#endif
constexpr int pk_dense_cap(int labels) { return labels == 16 ? LPMP_PK_DENSE_CAP16 : pk_indirect_cap(labels); }
constexpr int pk_class_cap(int kclass) { return kc_is_dense(kclass) ? pk_dense_cap(kc_width(kclass)) : pk_indirect_cap(kc_width(kclass)); }
constexpr int32_t UPD_PRELOAD_OK = 1 << 16;   // UpdRec::kind_flags: no send targets a vector a receive writes
constexpr int32_t UPD_PRIMAL = 1 << 17;       // UpdRec::kind_flags: the factor type has COMPUTE_PRIMAL_SOLUTION

// kernel flags of the sweep kernels
constexpr int SWEEP_RESIDUAL = 1;   // --reparametrizationType residual
constexpr int SWEEP_NT = 4;         // host-side selector: the model is far larger than the caches -> non-temporal variants
constexpr int SWEEP_ADAPTIVE = 8;   // --reparametrizationType adaptive (generic kernels only)
// bits 8-11: streaming dense class — LDS per wave sized for ceil(max label count of the launch / 64) * 64 labels (0: BIG_MAX_LABELS)
constexpr int SWEEP_BIGDIM_SHIFT = 8, SWEEP_BIGDIM_MASK = 15 << SWEEP_BIGDIM_SHIFT;
constexpr int sweep_bigdim_flags(int max_dim) { return max_dim <= 0 ? 0 : (((max_dim + 63) / 64) << SWEEP_BIGDIM_SHIFT) & SWEEP_BIGDIM_MASK; }
constexpr int SWEEP_PRIMAL = 2;     // UpdateFactorPrimal (reference factors_messages.hxx:2332-2373): factors of a
                                    // COMPUTE_PRIMAL type round their label from the state after the receives

// primal rounding (engine.cpp / kernels.hip): one unary-pairwise message, and one lazily initialised factor
struct PrimalLink { int32_t u, p, side, dim; };   // left (vector) factor, right (pairwise) factor, side, label count of u
struct PrimalInit { int32_t f, a, b, pad; };      // primal_ of factor f when unset: (a, b)

// Chain executor (kernels.hip): the launches of a deep schedule as ONE persistent launch.  A ticket = one workgroup's
// block of records of one launch; dep = the tickets holding the predecessors of its records (the last earlier update
// of every factor a record touches).  Tickets are numbered in level order, so a dependency always has a lower number.
struct ChainLaunchHost { int64_t rec_begin, count, pk_begin; int32_t stride, ticket0; int32_t flags = 0; };   // flags: CHAIN_LAUNCH_LABEL_OPS
constexpr int32_t CHAIN_LAUNCH_LABEL_PAIRED = 2;  // ... and in every record send j goes to the peer receive j came from (each message received, then sent)
// dense chain launches: message vectors between dependent records travel as tagged granules (kernels.hip, mailbox); the
// dependencies they cover are not in dep[] any more.  Bit 30: the low bits of ChainLaunch::pad belong to the joined passes.
constexpr int32_t CHAIN_LAUNCH_MAILBOX = 1 << 30;
// the sends of a record that may go to the mailbox = the sends (and forwarded receives) the mailbox form of the dense body
// holds in registers (kernels.hip: KS, NFW)
#ifndef LPMP_MBOX_KS
#define LPMP_MBOX_KS 4
#endif
constexpr int MAILBOX_SENDS = LPMP_MBOX_KS;
constexpr int32_t CHAIN_LAUNCH_LABEL_OPS = 1;   // level loop: every record a vector factor whose ops are labeling messages with it on the left, <= 8 receives and <= 8 sends, no two of a kind on one peer
struct ChainPlan {
  bool valid = false;
  int32_t kclass = 0;                       // the one kernel class of the schedule
  std::vector<ChainLaunchHost> launches;    // parallel to Schedule::launches
  std::vector<int32_t> tk_launch;           // [n_tickets]
  std::vector<int32_t> tk_block;            // [n_tickets]: block of records inside that launch
  std::vector<int32_t> dep_off, dep;        // CSR over tickets
  bool banded = false;                      // tickets in Infinity-Cache order: the tables are read with plain loads
  bool level_loop = false;                  // many tiny levels of a generic class: ONE workgroup walks the launches (kernels.hip)
  int64_t mailbox_rows = 0;                 // message vectors that travel through the mailbox (rows of mailbox_width granule pairs)
  int64_t mailbox_receives = 0;             // receives that poll a row
  int32_t mailbox_width = 0;
};
// records one workgroup of the packed kernels takes (256 threads / lanes per record)
constexpr int GENERIC_BLOCK_RECORDS = 4, SMALL_BLOCK_RECORDS = 64;   // sweep_generic_kernel<64> / <1> (kernels.hip asserts them)
constexpr int BIG_BLOCK_RECORDS = 4;          // sweep_dense_big_kernel: one wave per record (kernels.hip asserts it)
constexpr int kc_block_records(int kclass) {
  if (kclass == KC_GENERIC) return GENERIC_BLOCK_RECORDS;
  if (kclass == KC_SMALL) return SMALL_BLOCK_RECORDS;
  const int w = kc_width(kclass);
  if (w == 0) return 0;
  const bool dense = (kclass >= KC_DENSE_4 && kclass <= KC_DENSE_32) || (kclass >= KC_DENSE_V4 && kclass <= KC_DENSE_V32);
  return dense ? (w == 32 ? 4 : 256 / w) : 256 / w;   // dense: G = 64 lanes at 32 labels, else one lane per label
}
constexpr bool kc_chain_capable(int kclass) { return (kclass >= KC_DENSE_4 && kclass <= KC_POTTS_V32) || kclass == KC_GENERIC || kclass == KC_SMALL; }
constexpr int64_t CHAIN_MIN_LAUNCHES = 9;   // shorter schedules run as plain launches

// vectors of op records are hundreds of megabytes at the headline size and every element is written right after the
// allocation: default-initialise (= leave alone) instead of zero-filling them first
// contiguous chunks of [0, n) on a few threads (host analysis of models with millions of factors; plan.cpp has the same for
// its own loops): f(begin, end); exceptions are rethrown on the caller's thread
template <class F>
inline void parallel_blocks(int64_t n, int64_t min_per_thread, F&& f) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, hw ? hw : 1), n / std::max<int64_t>(1, min_per_thread)));
  if (nt <= 1) { f((int64_t)0, n); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  th.reserve((size_t)nt);
  auto work = [&](int64_t t) { try { f(n * t / nt, n * (t + 1) / nt); } catch (...) { err[(size_t)t] = std::current_exception(); } };
  int64_t started = 0;
  try {
    for (; started < nt - 1; ++started) th.emplace_back(work, started);
  } catch (...) {}                                   // (a thread could not be started: the caller's thread takes the remaining chunks)
  for (int64_t t = started; t < nt; ++t) work(t);
  for (auto& x : th) x.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}

template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_allocator<U>; };
  default_init_allocator() = default;
  template <class U> default_init_allocator(const default_init_allocator<U>&) noexcept {}
  template <class U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using OpVec = std::vector<Op, default_init_allocator<Op>>;

struct Schedule {             // executable form of one (factor list, omega, mask) sweep
  std::vector<UpdRec> recs;   // sorted by (level, kclass)
  OpVec ops;
  OpVec packets;                      // packed launches: [UpdRec | Op x (stride-1)] per factor
  std::vector<LevelRange> launches;   // in execution order
  int64_t n_levels = 0;
  int64_t n_recv = 0, n_send = 0;     // active receives / sends = message updates per sweep
  int64_t alg_bytes = 0;              // algorithmic HBM bytes per sweep (DESIGN.md accounting)
  // deep schedules: one chain plan per kernel class (classes between which no dependency runs are independent
  // sequences), the launches of the remaining classes stay plain launches (indices into `launches`)
  std::vector<ChainPlan> chains;
  std::vector<int32_t> plain_launches;
};

struct Plan {
  // copied structure (no cost data)
  int32_t n_ftypes = 0, n_mtypes = 0, n_tables = 0;
  std::vector<uint8_t> ftype_primal;
  std::vector<lpmp_msg_type> mtypes;
  std::vector<int64_t> tab_off;
  std::vector<int32_t> tab_data, tab_nleft;
  int64_t nf = 0, nm = 0;
  std::vector<int32_t> f_type, f_dim0, f_dim1;
  std::vector<uint8_t> f_kind, f_flags;
  std::vector<int64_t> f_coff, f_doff;   // [nf+1]
  std::vector<int32_t> m_type, m_left, m_right;
  double constant = 0;
  // derived
  std::vector<int64_t> fm_off;
  std::vector<MsgEntry> fm;
  std::vector<uint8_t> updated;
  std::vector<int32_t> order[2], upd[2];
  Csr<double> omega[2][LPMP_REPAM_COUNT];
  Csr<uint8_t> mask[2][LPMP_REPAM_COUNT];
  bool have[LPMP_REPAM_COUNT] = {false, false, false, false};
  int max_dual = 1;
  // LP::put_in_same_partition pairs (call order) and what construct_factor_partition derives from them
  std::vector<int32_t> part_pairs;
  struct SegList { std::vector<int32_t> f; Csr<double> om; Csr<uint8_t> mk; };   // a factor list with the weights of a pass over it
  struct Partition {
    bool valid = false;
    std::vector<int64_t> off; std::vector<int32_t> f;          // partitions (updated factors only), CSR
    std::vector<SegList> fwd, bwd, push_fwd, push_bwd, ov_fwd, ov_bwd;
  } part;
  bool any_batch = false;        // some message op has a static batch send (lpmp_msg_flags)
  bool force_generic = false;    // schedule every update on the generic kernels (adaptive sends)
  // device memory a schedule's mailbox may take (16 bytes per label and mailbox send; C3 row-major: 2 GB): the engine sets it
  // from the free memory of its device, and a class whose mailbox would not fit is planned with completion flags only
  // (-1: no limit)
  int64_t mailbox_budget_bytes = -1;
  // Engine-private placement of factors on the device (engine.cpp, rows layout): where a factor's constants start relative
  // to the const base pointer and its duals relative to the dual base pointer, in doubles — possibly in ANOTHER allocation
  // (the kernels only ever form base + offset).  Empty: the packed offsets f_coff / f_doff.  Sizes always come from f_*.
  std::vector<int64_t> dev_coff, dev_doff;
  int64_t coff(int64_t f) const { return dev_coff.empty() ? f_coff[f] : dev_coff[f]; }
  int64_t doff(int64_t f) const { return dev_doff.empty() ? f_doff[f] : dev_doff[f]; }

  // throws std::runtime_error on invalid input (the reference throws too, LP_MP.h:458)
  void build(const lpmp_model& m);
  void ensure_weights(int mode);
  void anisotropic_weights(const int32_t* list, int64_t n, Csr<double>& om, Csr<uint8_t>& mk) const;
  // one sweep: a factor list with one omega row and one receive-mask row per listed factor
  struct Segment { const int32_t* factors; int64_t n; const int64_t* om_off; const double* om; const int64_t* mk_off; const uint8_t* mk; };
  // turn a sequence of sweeps into levels/records/ops; fuse: fold back-to-back updates of one factor (plan.cpp)
  // chains = false: a schedule of exactly three levels (the shape whose consecutive passes the engine joins into its own
  // persistent launch, engine.cpp rotation_chain) gets no chain plan of its own — a third of the planning time at the
  // headline size for lists the joined launch never reads
  // levels_only (optional): only the dependent level of every update is wanted — filled ([sum of the segments' n]; 0 for an update
  // that becomes no record: no active message and no primal to round) and nothing else is built (a third of the work)
  void make_schedule(const std::vector<Segment>& segs, bool fuse, Schedule& out, bool chains = true, std::vector<int32_t>* levels_only = nullptr) const;
  void make_schedule(const int32_t* factors, int64_t n, const int64_t* om_off, const double* om,
                     const int64_t* mk_off, const uint8_t* mk, Schedule& out) const;
  int64_t row_sends(int32_t f) const { return n_row_sends[(size_t)f]; }       // entries of f's message list that send / receive
  int64_t row_receives(int32_t f) const { return n_row_receives[(size_t)f]; }
  std::vector<int32_t> n_row_sends, n_row_receives;
  // LP::construct_factor_partition / construct_overlapping_factor_partition (reference LP_MP.h:1717-1843)
  void ensure_partition();
  // the iterator-range passes of compute_partition_pass (rtype 2, LP_MP.h:1932-1963) or
  // compute_overlapping_partition_pass (rtype 3, :1966-2051; without the plain sweeps that follow it), in order
  void partition_pass_segments(int rtype, int inner_iterations, std::vector<Segment>& out);
  // CallSendMessages' batch rule (reference factors_messages.hxx:2709-2726) as the weights the individual sends get
  void effective_send_weights(int32_t f, const double* omega, double* w) const;
  // why the adaptive send rule cannot run this model ("" if it can)
  std::string adaptive_obstacle() const;
};

// graph.cpp: an order of all factors with the updated ones colour by colour (rank[f] = position; returns the number of colours)
int32_t suggest_order(const Plan& p, uint64_t seed, int32_t* rank);

}  // namespace lpmp
