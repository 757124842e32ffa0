// engine.cpp — the C ABI of include/lpmp_engine.h: device memory, schedules, launches.
// Host code only; the kernels are in kernels.hip.  There is NO CPU execution path: every compute
// entry point needs a HIP device and fails with LPMP_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lpmp_engine.h"
#include "plan.hpp"

namespace lpmp {
void launch_sweep(int kclass, const UpdRec* recs, const Op* ops, double* dual, const double* cdata, const int32_t* tabs,
                  double* lb, int32_t* primal, const int32_t* pw_unary, int64_t first, int64_t count, int flags, hipStream_t s);
bool launch_sweep_packed(int kclass, const Op* packets, const UpdRec* recs, const Op* ops, int stride, double* dual, const double* cdata,
                         double* lb, int32_t* primal, int64_t count, int flags, hipStream_t s);
bool launch_chain(int kclass, int flags, const void* chain_args, const void* launches, double* dual, const double* cdata,
                  const int32_t* tabs, double* lb, int32_t* primal, hipStream_t s);
bool launch_level_loop(int kclass, int flags, const void* launches, int n_launches, double* dual, const double* cdata,
                       const int32_t* tabs, double* lb, hipStream_t s);
void debug_set_level_trace(long long* p);
void launch_primal_init(const PrimalInit* list, int64_t n, int32_t* primal, hipStream_t s);
void launch_primal_propagate(const PrimalLink* links, int64_t n, int32_t* primal, hipStream_t s);
void launch_primal_check(const PrimalLink* links, int64_t n, const int32_t* primal, int* bad, hipStream_t s);
void launch_primal_cost(const void* recs, const double* dual, const double* cdata, const int32_t* primal, double* out, int64_t count, hipStream_t s);
void launch_lb_collect_stale(const double* lb, int64_t n, int32_t* list, unsigned long long* counter, hipStream_t s);
void launch_factor_lb_list(const void* recs, const double* dual, const double* cdata, double* out, const int32_t* list, int64_t count, hipStream_t s);
void launch_factor_lb(const void* recs, const double* dual, const double* cdata, double* out, int64_t count, hipStream_t s);
bool launch_dense_lb(int L, const void* recs, const double* dual, const double* cdata, double* out, int64_t first, int64_t count, hipStream_t s);
void launch_sum_stage(const double* in, double* out, int64_t n, int64_t per_block, int64_t n_blocks, hipStream_t s);
void launch_synth_fill(double* out, int64_t n, uint64_t seed, uint64_t first, hipStream_t s);
void launch_rows_copy(const void* recs, int64_t n, const double* cdata, double* dual, double* rows, int what, hipStream_t s);
struct RowRecHost { int64_t dual_off, const_off, row_off; int32_t d0, d1; };
int generic_max_dual();
int generic_max_adaptive_sends();
struct LbRecHost { int64_t dual_off; int64_t const_off; int32_t d0, d1; int32_t kind_flags; int32_t pad; };
}  // namespace lpmp

using namespace lpmp;

static thread_local std::string g_error;
const char* lpmp_last_error(void) { return g_error.c_str(); }
extern "C" int lpmp_set_last_error(const char* msg) { g_error = msg ? msg : ""; return 0; }   // for the other translation units of the library
const char* lpmp_version(void) { return "lp_mp_amd 0.1 (gfx950)"; }

namespace {

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct StateError : std::runtime_error { using std::runtime_error::runtime_error; };
struct UnsupportedError : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw DeviceError(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

template <class F>
int guarded(F&& f) {
  try { f(); return LPMP_OK; }
  catch (const DeviceError& e) { g_error = e.what(); return LPMP_ERR_DEVICE; }
  catch (const StateError& e) { g_error = e.what(); return LPMP_ERR_STATE; }
  catch (const UnsupportedError& e) { g_error = e.what(); return LPMP_ERR_UNSUPPORTED; }
  catch (const std::bad_alloc&) { g_error = "out of host memory"; return LPMP_ERR_INVALID; }
  catch (const std::exception& e) { g_error = e.what(); return LPMP_ERR_INVALID; }
}

// one launch as the chain kernels see it (layout shared with kernels.hip: ChainLaunch)
struct ChainLaunchDev { const Op* packets; const UpdRec* recs; const Op* ops; int64_t count; int32_t stride, pad; };
static_assert(sizeof(ChainLaunchDev) == 40, "ChainLaunch layout");

struct DevSchedule {
  UpdRec* recs = nullptr;
  Op* ops = nullptr;
  Op* packets = nullptr;
  std::vector<LevelRange> launches;
  int64_t n_levels = 0, n_recv = 0, n_send = 0, alg_bytes = 0;
  hipGraphExec_t graph = nullptr;
  hipGraphExec_t graph_primal = nullptr;   // the same launches with the SWEEP_PRIMAL flag
  bool adaptive_built = false;             // built with every update on the generic kernels (adaptive send rule)
  size_t recs_cap = 0, ops_cap = 0, packets_cap = 0;   // allocated elements (a scratch schedule is refilled in place)
  // chain executor (deep schedules): device copies of the ChainPlans, one per kernel class; the other launches stay plain
  struct DevChain {
    int32_t kclass = 0, tickets = 0, epoch = 0;
    bool banded = false;                     // Infinity-Cache ticket order: plain table loads, not the streaming policy
    bool level_loop = false; int32_t n_launches = 0;   // one workgroup walks the launches (tiny levels of a generic class)
    ChainLaunchDev* launches = nullptr; int32_t *tk_launch = nullptr, *tk_block = nullptr, *dep_off = nullptr, *dep = nullptr, *done = nullptr, *next = nullptr;
    unsigned long long* mailbox = nullptr;   // tagged granules of the message vectors that travel between dependent records (plan.cpp)
  };
  std::vector<DevChain> chains;
  std::vector<LevelRange> plain;           // launches that do not belong to a chain
  bool chain = false;
  void release_chain() {
    for (auto& c : chains)
      for (void* p : {(void*)c.launches, (void*)c.tk_launch, (void*)c.tk_block, (void*)c.dep_off, (void*)c.dep, (void*)c.done, (void*)c.next, (void*)c.mailbox}) if (p) (void)hipFree(p);
    chains.clear(); plain.clear(); chain = false;
  }
  void release() {
    if (graph) { (void)hipGraphExecDestroy(graph); graph = nullptr; }
    if (graph_primal) { (void)hipGraphExecDestroy(graph_primal); graph_primal = nullptr; }
    if (recs) { (void)hipFree(recs); recs = nullptr; }
    if (ops) { (void)hipFree(ops); ops = nullptr; }
    if (packets) { (void)hipFree(packets); packets = nullptr; }
    release_chain();
    recs_cap = ops_cap = packets_cap = 0;
    launches.clear();
  }
};

// device-side description of a chain plan (layouts shared with kernels.hip: ChainArgs, ChainLaunch)
struct ChainArgsHost {
  const int32_t* dep_off; const int32_t* dep; int32_t* done; int32_t* next; int32_t* abort_flag; const int32_t* tk_launch;
  const int32_t* tk_block; int32_t n_tickets; int32_t epoch; long long* trace;
  double* lb_hist; int64_t hist_stride;     // per-pass bound rows of a joined-pass launch (kernels.hip, ChainArgs)
  unsigned long long* mailbox;              // or nullptr
  long long timeout_ticks;                  // bound of every wait, ticks of the 100 MHz s_memrealtime clock
  int32_t per_begin, per_len, per_count, per_launch_shift, per_row_shift;   // periodic ticket lists (kernels.hip, ChainArgs)
  int32_t hist_rows;                        // rows of lb_hist the launch may write
  int32_t ring;                             // slots of done[] when it is a ring (0: one flag per ticket)
};
constexpr int CHAIN_GEN_BITS = 8;           // kernels.hip
constexpr int CHAIN_ABORT_WORDS = 16;       // abort word + what the first wait that gave up was waiting for (kernels.hip, chain_abort)
static long long chain_timeout_ticks() {    // LPMP_CHAIN_TIMEOUT_S (default 20 s)
  static const long long v = [] { const char* e = std::getenv("LPMP_CHAIN_TIMEOUT_S"); const double s = e ? std::atof(e) : 20.0; return (long long)(std::max(0.001, s) * 1e8); }();
  return v;
}
constexpr int HIST_END = 1, HIST_MID = 2;   // kernels.hip

struct ClassTiming { double ms = 0; int64_t launches = 0, factors = 0, receives = 0, bytes = 0, chain_launches = 0; };

}  // namespace

// Joined passes as ONE persistent launch (chain executor with a skewed ticket order): what the expansion
// for n passes needs of the two fused schedules.  Templates: H, W, T = the three steps of forward+backward, K = the middle
// step of backward+forward; n passes = H, W, (K, W)^(n-1), T.
struct RotationInfo {
  bool valid = false;
  int kclass = 0, gpb = 1;
  struct Tmpl { int sched; LevelRange lr; int32_t nb; int64_t factors, recv, bytes; };   // sched: 0 forward+backward, 1 backward+forward
  Tmpl t[4];                                                   // H, W, K, T
  // predecessors of a step's blocks: kind 0 W after [H]; 1 K after [W, H]; 2 W after [K, W]; 3 K after [W, K];
  // 4 T after [W, K]; 5 T after [W, H].  (delta, block): the block of the step delta steps earlier
  std::vector<int64_t> off[6]; std::vector<int8_t> delta[6]; std::vector<int32_t> block[6];
  // every factor's bound at a pass seam is known to a W record (its own, at the end) or a K record (its own after the
  // receives, or as the pairwise peer of one of its receives): then a joined launch can emit one bound row per pass
  bool hist_ok = false;
  // how far AHEAD in a step's block list a steady-state block's predecessors of the step before lie, as a fraction of the
  // list (a W x H grid in a 2-colour order: one grid row, 1 / H): what the lag of the skewed ticket order has to cover before any slack
  double reach = 0;
};

struct lpmp_plan {
  Plan p;
  RotationInfo rot[LPMP_REPAM_COUNT];
  Schedule sched_cache[2][LPMP_REPAM_COUNT]; bool have_sched[2][LPMP_REPAM_COUNT] = {{false}};
  Schedule pass_cache[LPMP_REPAM_COUNT]; bool have_pass[LPMP_REPAM_COUNT] = {false};   // forward+backward as one fused sequence
  Schedule bf_cache[LPMP_REPAM_COUNT]; bool have_bf[LPMP_REPAM_COUNT] = {false};       // backward+forward (the seam between two passes)
  bool rotation_ok[LPMP_REPAM_COUNT] = {false};
  void drop_caches() {   // the kernel classes of every schedule change with Plan::force_generic
    for (int m = 0; m < LPMP_REPAM_COUNT; ++m) {
      for (int d = 0; d < 2; ++d) { sched_cache[d][m] = Schedule(); have_sched[d][m] = false; }
      pass_cache[m] = Schedule(); have_pass[m] = false; bf_cache[m] = Schedule(); have_bf[m] = false; rotation_ok[m] = false;
    }
  }
};

static void plan_pass_schedule(lpmp_plan* pl, int mode, bool chains = true) {
  if (pl->have_pass[mode]) return;
  pl->p.ensure_weights(mode);
  std::vector<Plan::Segment> segs;
  for (int d = 0; d < 2; ++d) {
    const auto& om = pl->p.omega[d][mode];
    const auto& mk = pl->p.mask[d][mode];
    segs.push_back({pl->p.upd[d].data(), (int64_t)pl->p.upd[d].size(), om.off.data(), om.data.data(), mk.off.data(), mk.data.data()});
  }
  pl->p.make_schedule(segs, true, pl->pass_cache[mode], chains);
  pl->have_pass[mode] = true;
}

static void plan_schedule(lpmp_plan* pl, int d, int mode) {
  if (pl->have_sched[d][mode]) return;
  pl->p.ensure_weights(mode);
  const auto& om = pl->p.omega[d][mode];
  const auto& mk = pl->p.mask[d][mode];
  pl->p.make_schedule(pl->p.upd[d].data(), (int64_t)pl->p.upd[d].size(), om.off.data(), om.data.data(), mk.off.data(),
                      mk.data.data(), pl->sched_cache[d][mode]);
  pl->have_sched[d][mode] = true;
}

// Seam between two consecutive passes.  If forward+backward fuses into exactly three steps
//   [H: head of the forward sweep] [W] [T: tail of the backward sweep]
// and backward+forward fuses into [H'] [K] [T'] where K's records are exactly T's receives followed by H's sends
// (same peers, sides, weights, order — checked op by op) and W's are T's' receives followed by H's' sends, then n
// passes are
//   H, W, (K, W) x (n-1), T
// — every record is the same sequence of receives and sends the unfused sweeps execute (plan.cpp, fusion).
// 2-colour orders of bipartite graphs (checkerboard grids) have this shape.
// records of one level keyed by factor
static std::vector<std::pair<int32_t, const UpdRec*>> level_records(const Schedule& s, int level) {
  std::vector<std::pair<int32_t, const UpdRec*>> f;
  for (const auto& lr : s.launches) if (lr.level == level) for (int64_t i = lr.begin; i < lr.end; ++i) f.emplace_back(s.recs[i].factor, &s.recs[i]);
  std::sort(f.begin(), f.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  return f;
}
static bool same_op(const Op& a, const Op& b) {   // everything but the forwarding hint (pad), which is per schedule
  return a.peer_dual == b.peer_dual && a.peer_const == b.peer_const && a.omega == b.omega && a.info == b.info &&
         a.pd0 == b.pd0 && a.pd1 == b.pd1 && a.peer == b.peer && a.len == b.len;
}
// does every record of `joined` (level lj of schedule sj) hold exactly the receives of the same factor's record in
// (sr, lr) followed by the sends of its record in (ss, ls) — same peers, sides, weights, order?
static bool level_is_join(const Schedule& sj, int lj, const Schedule& sr, int lr, const Schedule& ss, int ls) {
  const auto j = level_records(sj, lj), r = level_records(sr, lr), t = level_records(ss, ls);
  if (j.size() != r.size() || j.size() != t.size()) return false;
  for (size_t i = 0; i < j.size(); ++i) {
    if (j[i].first != r[i].first || j[i].first != t[i].first) return false;
    if (i > 0 && j[i].first == j[i - 1].first) return false;   // one record per factor and level
    const UpdRec& a = *j[i].second; const UpdRec& b = *r[i].second; const UpdRec& c = *t[i].second;
    if (b.n_send != 0 || c.n_recv != 0 || a.n_recv != b.n_recv || a.n_send != c.n_send) return false;
    for (int k = 0; k < a.n_recv; ++k) if (!same_op(sj.ops[a.op_begin + k], sr.ops[b.op_begin + k])) return false;
    for (int k = 0; k < a.n_send; ++k) if (!same_op(sj.ops[a.op_begin + a.n_recv + k], ss.ops[c.op_begin + k])) return false;
  }
  return true;
}
static void plan_rotation(lpmp_plan* pl, int mode) {
  if (pl->have_bf[mode]) return;
  pl->have_bf[mode] = true;
  pl->rotation_ok[mode] = false;
  const Schedule& fb = pl->pass_cache[mode];
  if (fb.n_levels != 3 || fb.recs.empty()) return;
  std::vector<Plan::Segment> segs;
  for (int d = 1; d >= 0; --d) {
    const auto& om = pl->p.omega[d][mode];
    const auto& mk = pl->p.mask[d][mode];
    segs.push_back({pl->p.upd[d].data(), (int64_t)pl->p.upd[d].size(), om.off.data(), om.data.data(), mk.off.data(), mk.data.data()});
  }
  Schedule& bf = pl->bf_cache[mode];
  pl->p.make_schedule(segs, true, bf, false);
  // K (middle step of backward+forward) must be exactly "receives of T, then sends of H" factor by factor, and W
  // (middle step of forward+backward) exactly "receives of T', then sends of H'": then H, W, (K, W)^(n-1), T executes
  // the same receives and sends as n unfused passes, in an order that differs only between independent updates
  const bool ok = bf.n_levels == 3 && level_is_join(bf, 2, fb, 3, fb, 1) && level_is_join(fb, 2, bf, 3, bf, 1);
  pl->rotation_ok[mode] = ok;
  if (!ok) bf = Schedule();
}


// who touches what in one launch: the block (of gpb records) whose record updates factor g or reaches it through an op
static std::vector<int32_t> launch_touchers(const Schedule& s, const LevelRange& lr, int gpb, int64_t nf) {
  std::vector<int32_t> t((size_t)nf, -1);
  for (int64_t i = lr.begin; i < lr.end; ++i) {
    const UpdRec& r = s.recs[i];
    const int32_t b = (int32_t)((i - lr.begin) / gpb);
    t[r.factor] = b;
    for (int k = 0; k < r.n_recv + r.n_send; ++k) t[s.ops[r.op_begin + k].peer] = b;
  }
  return t;
}
static void plan_rotation_chain(lpmp_plan* pl, int mode) {
  RotationInfo& ri = pl->rot[mode];
  ri = RotationInfo();
  if (!pl->rotation_ok[mode]) return;
  const Schedule& fb = pl->pass_cache[mode];
  const Schedule& bf = pl->bf_cache[mode];
  auto only_launch = [](const Schedule& s, int level, LevelRange& out) {
    int n = 0;
    for (const auto& lr : s.launches) if (lr.level == level) { out = lr; ++n; }
    return n == 1;
  };
  LevelRange h, w, k, t;
  if (!only_launch(fb, 1, h) || !only_launch(fb, 2, w) || !only_launch(fb, 3, t) || !only_launch(bf, 2, k)) return;
  const int kc = w.kclass;
  if (h.kclass != kc || k.kclass != kc || t.kclass != kc || !kc_chain_capable(kc) || kc_width(kc) == 0) return;

  if (h.stride == 0 || w.stride == 0 || k.stride == 0 || t.stride == 0) return;
  ri.kclass = kc; ri.gpb = kc_block_records(kc);
  const LevelRange* lrs[4] = {&h, &w, &k, &t};
  const Schedule* sch[4] = {&fb, &fb, &bf, &fb};
  std::vector<int32_t> touch[4];
  for (int i = 0; i < 4; ++i) {
    const int64_t cnt = lrs[i]->end - lrs[i]->begin;
    ri.t[i] = {i == 2 ? 1 : 0, *lrs[i], (int32_t)((cnt + ri.gpb - 1) / ri.gpb), cnt, lrs[i]->n_recv, lrs[i]->bytes};
    touch[i] = launch_touchers(*sch[i], *lrs[i], ri.gpb, pl->p.nf);
  }
  // kind -> (X, Y1, Y2)
  const int X[6] = {1, 2, 1, 2, 3, 3}, Y1[6] = {0, 1, 2, 1, 1, 1}, Y2[6] = {-1, 0, 1, 2, 2, 0};
  for (int kind = 0; kind < 6; ++kind) {
    const Schedule& s = *sch[X[kind]];
    const LevelRange& lr = *lrs[X[kind]];
    // (blocks are independent: chunks of them on several threads, see plan.hpp parallel_blocks)
    const int64_t nbk = ri.t[X[kind]].nb;
    std::vector<std::vector<std::pair<int8_t, int32_t>>> per((size_t)nbk);
    parallel_blocks(nbk, 4096, [&](int64_t b0, int64_t b1) {
      for (int64_t b = b0; b < b1; ++b) {
        auto& dst = per[(size_t)b];
        for (int64_t i = lr.begin + b * ri.gpb; i < std::min<int64_t>(lr.end, lr.begin + (b + 1) * ri.gpb); ++i) {
          const UpdRec& r = s.recs[i];
          auto visit = [&](int32_t g) {
            if (touch[Y1[kind]][g] >= 0) dst.emplace_back((int8_t)1, touch[Y1[kind]][g]);
            else if (Y2[kind] >= 0 && touch[Y2[kind]][g] >= 0) dst.emplace_back((int8_t)2, touch[Y2[kind]][g]);
          };
          visit(r.factor);
          for (int q = 0; q < r.n_recv + r.n_send; ++q) visit(s.ops[r.op_begin + q].peer);
        }
        std::sort(dst.begin(), dst.end());
        dst.erase(std::unique(dst.begin(), dst.end()), dst.end());
      }
    });
    ri.off[kind].assign(1, 0);
    for (auto& v : per) {
      for (const auto& d : v) { ri.delta[kind].push_back(d.first); ri.block[kind].push_back(d.second); }
      ri.off[kind].push_back((int64_t)ri.block[kind].size());
    }
    if (kind == 2 || kind == 3) {
      const double nbs = (double)std::max<int64_t>(1, nbk), nbp = (double)std::max<int32_t>(1, ri.t[Y1[kind]].nb);
      for (int64_t b = 0; b < nbk; ++b)
        for (const auto& d : per[(size_t)b])
          if (d.first == 1) ri.reach = std::max(ri.reach, (d.second + 0.5) / nbp - (b + 0.5) / nbs);
    }
  }
  {   // coverage of the per-pass bound rows (kernels.hip HIST_END / HIST_MID)
    std::vector<uint8_t> cov((size_t)pl->p.nf, 0);
    for (int64_t i = w.begin; i < w.end; ++i) cov[fb.recs[i].factor] = 1;
    for (int64_t i = k.begin; i < k.end; ++i) {
      const UpdRec& r = bf.recs[i];
      cov[r.factor] = 1;
      for (int q = 0; q < r.n_recv; ++q) cov[bf.ops[r.op_begin + q].peer] = 1;
    }
    ri.hist_ok = true;
    for (int64_t f = 0; f < pl->p.nf; ++f) if (!cov[f]) { ri.hist_ok = false; break; }
  }
  ri.valid = true;
}

struct lpmp_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t capture_stream = nullptr;   // graphs are captured here (the caller's stream may be the legacy default stream, which cannot capture) and replayed on `stream`
  std::unique_ptr<lpmp_plan> plan;
  double* d_dual = nullptr; bool own_dual = false;
  double* d_const = nullptr; bool own_const = false;
  int32_t* d_tabs = nullptr;
  LbRecHost* d_lbrecs = nullptr;
  double* d_lb = nullptr; double* d_part = nullptr; double* h_part = nullptr;
  // tracked per-factor lower bounds (kernels.hip): d_lb[f] is valid or NaN; lb_all_stale: recompute everything
  int32_t* d_stale = nullptr; unsigned long long* d_stale_n = nullptr; unsigned long long* h_stale_n = nullptr;
  bool lb_all_stale = true;
  bool use_lb_tracking = true;
  int64_t last_lb_recomputed = -1;   // factor bounds the last evaluation had to recompute (-1: none evaluated yet)
  // primal rounding: the factors' primal_ members, the lazily initialised set, the message links
  int32_t* d_primal = nullptr;
  PrimalInit* d_pinit = nullptr; int64_t n_pinit = 0;
  PrimalLink* d_plinks = nullptr; int64_t n_plinks = 0, n_pprop = 0;   // all messages; the first n_pprop propagate labels
  int32_t* d_pw_unary = nullptr;  // [2 nf] the unary on each side of a pairwise factor; only with pairwise types that round themselves
  double* d_pcost = nullptr; int* d_pbad = nullptr; int* h_pbad = nullptr;
  char* pinned = nullptr;         // this engine's block of device-written host words (from the pool)
  uint64_t primal_t = 0;          // primal_access_ of every factor a primal pass touches (they move together)
  bool have_primal = false;
  bool primal_pass = false;       // the launches being issued belong to an ...AndPrimal pass
  // rows layout (kernels.hip, rows_copy_kernel): dense pairwise factors live as [table | m1 | m2] rows of d_rows; the packed
  // dual array keeps the vector factors and is the format of every call that hands duals over.  packed_stale: the rows hold
  // newer message vectors than the packed array; rows_stale: the packed array was (or may have been) written by the caller
  bool want_rows = false, rows = false, packed_stale = false, rows_stale = false;
  double* d_rows = nullptr; RowRecHost* d_rowrecs = nullptr; int64_t n_rowrecs = 0;
  int nt_flag = 0;                // SWEEP_NT when tables + duals are far larger than L2 + Infinity Cache
  bool model_big = false;         // tables + duals > 1 GiB: only then is an Infinity-Cache ticket order worth a chain launch
  struct LbRun { int cls; int64_t first, count; };
  std::vector<LbRun> lb_runs;
  DevSchedule sched[2][LPMP_REPAM_COUNT];
  bool have_sched[LPMP_REPAM_COUNT] = {false, false, false, false};
  DevSchedule sched_pass[LPMP_REPAM_COUNT];            // fused forward+backward (ComputePass)
  DevSchedule sched_bf[LPMP_REPAM_COUNT];              // fused backward+forward: its middle step joins two passes
  DevSchedule sched_part[2]; bool have_part[2] = {false, false};   // compute_partition_pass / compute_overlapping_partition_pass as ONE sequence
  int inner_iterations = 5;                            // --innerIteration (reference LP_MP.h:590)
  bool rotation_ok[LPMP_REPAM_COUNT] = {false, false, false, false};
  bool have_pass[LPMP_REPAM_COUNT] = {false, false, false, false};
  bool use_fused = true;
  bool use_rotation = true;
  std::vector<std::unique_ptr<DevSchedule>> custom;   // prepared iterator-range passes
  DevSchedule scratch;                                 // the one-off schedule of lpmp_compute_pass_custom
  int mode = -1;
  int rtype = 0;   // enum lpmp_reparametrization_type
  bool use_graph = true;
  bool use_packed = true;
  bool use_chain = true;          // deep single-class schedules as one persistent launch (LPMP_NO_CHAIN=1: graph replay)
  int32_t* d_chain_abort = nullptr; bool chain_ran = false;
  // joined passes as one persistent launch: expansions of RotationInfo, by mode and pass count
  struct RotChain {
    DevSchedule::DevChain dc; int n_steps = 0; int64_t factors = 0, recv = 0, bytes = 0; uint64_t last_use = 0; size_t dev_bytes = 0;
    // periodic form (rotation_chain): dc holds the TEMPLATE of n_tmpl passes whose tickets [per_begin, per_begin + per_len) are
    // one group of `depth` steps that an n-pass launch executes 1 + (n - n_tmpl) / (depth / 2) times; per_* sums: one period
    bool periodic = false; int n_tmpl = 0, depth = 0; int32_t per_begin = 0, per_len = 0, ring = 0;
    int64_t per_factors = 0, per_recv = 0, per_bytes = 0;
  };
  // the ticket lists of a pass count are device memory (C3, 32 passes: ~350 MB): the cache of built chains is bounded in
  // BYTES over all modes (default 2 GiB, LPMP_CHAIN_CACHE_MB; least recently used first), lpmp_chain_cache_bytes reports it
  uint64_t rot_clock = 0;
  size_t rot_cache_bytes = 0, rot_cache_limit = (size_t)2 << 30;
  std::map<int, RotChain> rot_chain[LPMP_REPAM_COUNT];
  // ---- speculative passes (lpmp_set_speculation, include/lpmp_engine.h) ----
  struct Spec {
    int max_depth = 0;                 // 0: off
    int n = 0, pos = 0, mode = -1;     // the open batch: n passes were launched as one chain, the caller has asked for pos of them
    int run_len = 0, learned = 0, last_batch = 0;   // plain single passes since the last other call; length of the previous such run
    bool lb_ready = false; std::vector<double> lb;   // bounds after passes 1 ... n - 1 of the batch
    double* d_snap = nullptr; size_t snap_cap = 0;   // duals (+ tracked bounds) at the start of the batch
    double* d_hist = nullptr; size_t hist_cap = 0;   // (n - 1) rows of per-factor bounds
    double* d_hpart = nullptr; size_t hpart_cap = 0; // partial sums of those rows
    bool snap_lb_stale = false;
    int64_t batches = 0, passes_launched = 0, passes_used = 0, rollbacks = 0, alloc_failures = 0;
    void release() {
      for (double** p : {&d_snap, &d_hist, &d_hpart}) if (*p) { (void)hipFree(*p); *p = nullptr; }
      snap_cap = hist_cap = hpart_cap = 0; n = pos = 0; mode = -1; run_len = learned = last_batch = 0; lb_ready = false;
    }
  } spec;
  bool use_blocked_passes = true;     // LPMP_NO_BLOCKED_PASSES=1: the joined passes as one launch per step
  bool pass_chain_tried[LPMP_REPAM_COUNT] = {};   // ensure_pass_chain_plan ran for that mode
  bool deep_note_given = false;                   // the one-line note about a schedule of many levels was printed
  int rot_bands = 0, rot_lag = 3, rot_depth = 4;   // skewed ticket order (0 bands: from the table bytes per step); DESIGN.md 6 has the sweep
  bool rot_lag_set = false, rot_depth_set = false; // LPMP_ROT_LAG / LPMP_ROT_DEPTH given: used as they are; else from the model (rot_geometry)
  // tiled ticket order of the joined passes (rotation_chain): LPMP_ROT_TILES=T forces tiles of T blocks per step, =0 forbids them;
  // unset: the engine's own choice (tiles of 1024 blocks where the band order is down to depth 2 or does not fit at all)
  int rot_tiles = 0; bool rot_tiles_set = false;
  // delayed[sd]: share of a steady-state step's blocks that run later than their own tile's phase, sd steps into a group
  struct TileSet { bool built = false; int T = 0; std::vector<int32_t> w, k; int32_t n = 0; double radius = 0; double delayed[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
  TileSet rot_tile_set[LPMP_REPAM_COUNT];
  void release_rot_chains() {
    for (auto& m : rot_chain) {
      for (auto& kv : m) {
        auto& c = kv.second.dc;
        for (void* p : {(void*)c.launches, (void*)c.tk_launch, (void*)c.tk_block, (void*)c.dep_off, (void*)c.dep, (void*)c.done, (void*)c.next, (void*)c.mailbox}) if (p) (void)hipFree(p);
      }
      m.clear();
    }
    rot_cache_bytes = 0;
    for (auto& t : rot_tile_set) t = TileSet();
  }
  bool timing = false;
  ClassTiming ct[KC_COUNT];
  struct Pending { hipEvent_t a, b; int cls; int64_t factors, receives, bytes; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;

  void release_primal() {
    if (d_primal) { (void)hipFree(d_primal); d_primal = nullptr; }
    if (d_pinit) { (void)hipFree(d_pinit); d_pinit = nullptr; }
    if (d_plinks) { (void)hipFree(d_plinks); d_plinks = nullptr; }
    if (d_pw_unary) { (void)hipFree(d_pw_unary); d_pw_unary = nullptr; }
    if (d_pcost) { (void)hipFree(d_pcost); d_pcost = nullptr; }
    if (d_pbad) { (void)hipFree(d_pbad); d_pbad = nullptr; }
    h_pbad = nullptr;
    have_primal = false; primal_t = 0; n_pinit = n_plinks = n_pprop = 0;
  }
  // the built-in schedules (everything but the caller's prepared iterator-range passes)
  void release_schedules() {
    for (int d = 0; d < 2; ++d) for (int m = 0; m < LPMP_REPAM_COUNT; ++m) sched[d][m].release();
    for (int m = 0; m < LPMP_REPAM_COUNT; ++m) { have_sched[m] = false; sched_pass[m].release(); sched_bf[m].release(); have_pass[m] = false; rotation_ok[m] = false; pass_chain_tried[m] = false; }
    for (int k = 0; k < 2; ++k) { sched_part[k].release(); have_part[k] = false; }
    release_rot_chains();
  }
  void release_model() {
    release_schedules();
    deep_note_given = false;
    for (auto& c : custom) if (c) c->release();
    custom.clear();
    scratch.release();
    if (own_dual && d_dual) (void)hipFree(d_dual);
    if (own_const && d_const) (void)hipFree(d_const);
    d_dual = nullptr; d_const = nullptr; own_dual = own_const = false;
    if (d_tabs) { (void)hipFree(d_tabs); d_tabs = nullptr; }
    if (d_rows) { (void)hipFree(d_rows); d_rows = nullptr; }
    if (d_rowrecs) { (void)hipFree(d_rowrecs); d_rowrecs = nullptr; }
    rows = packed_stale = rows_stale = false; n_rowrecs = 0;
    if (d_lbrecs) { (void)hipFree(d_lbrecs); d_lbrecs = nullptr; }
    if (d_lb) { (void)hipFree(d_lb); d_lb = nullptr; }
    if (d_part) { (void)hipFree(d_part); d_part = nullptr; }
    h_part = nullptr;
    release_primal();
    if (d_stale) { (void)hipFree(d_stale); d_stale = nullptr; }
    if (d_stale_n) { (void)hipFree(d_stale_n); d_stale_n = nullptr; }
    h_stale_n = nullptr;
    lb_all_stale = true;
    lb_runs.clear();
    plan.reset();
    mode = -1;
    spec.release();
  }
  hipEvent_t get_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
  }
  void drain_timing() {
    for (auto& p : pending) {
      HIP_CHECK(hipEventSynchronize(p.b));
      float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
      ClassTiming& c = ct[p.cls];
      c.ms += ms; c.launches++; c.factors += p.factors; c.receives += p.receives; c.bytes += p.bytes;
      event_pool.push_back(p.a); event_pool.push_back(p.b);
    }
    pending.clear();
  }
};

namespace {

// Copies between caller memory (pageable) and the device go through a pinned staging buffer that this code owns:
// plain memcpy on the calling thread on the host side, hipMemcpyAsync between the pinned buffer and the device on the
// engine's stream, waited for.  Nothing in the HIP runtime then reads or writes caller memory — an asynchronous copy
// straight to / from pageable memory is staged by the runtime on its own, and a handful of runs in ~300 000 randomised
// test runs showed host heap corruption next to freshly freed download buffers.  The engine's stream is
// non-blocking, so the null stream (plain hipMemcpy) would not be ordered with its kernels either.
// Guard regions.  Every host buffer the device or the HIP runtime writes into on the engine's behalf (the staging
// buffer of h2d / d2h, the block of device-written words) sits between two GUARD_BYTES regions holding a fixed
// pattern; the pattern is verified after every staged copy, in lpmp_synchronize and in lpmp_destroy.  A DMA or a
// runtime-side write that runs past its buffer is reported instead of silently landing in a neighbour allocation
// (the host-heap corruption hunt).
constexpr size_t GUARD_BYTES = 4096;
constexpr uint64_t GUARD_WORD = 0xA5C3E1F00F1E3C5AULL;
void guard_fill(void* p) { uint64_t* w = (uint64_t*)p; for (size_t i = 0; i < GUARD_BYTES / 8; ++i) w[i] = GUARD_WORD ^ (uint64_t)i; }
bool guard_ok(const void* p) {
  const uint64_t* w = (const uint64_t*)p;
  for (size_t i = 0; i < GUARD_BYTES / 8; ++i) if (w[i] != (GUARD_WORD ^ (uint64_t)i)) return false;
  return true;
}
// pinned allocation of `bytes` usable bytes with a guard region on either side; returns the usable pointer
char* guarded_host_alloc(size_t bytes) {
  char* raw = nullptr;
  HIP_CHECK(hipHostMalloc((void**)&raw, bytes + 2 * GUARD_BYTES, hipHostMallocDefault));
  guard_fill(raw); guard_fill(raw + GUARD_BYTES + bytes);
  return raw + GUARD_BYTES;
}
void guarded_host_free(char* p) { if (p) (void)hipHostFree(p - GUARD_BYTES); }
bool guarded_ok(const char* p, size_t bytes) { return !p || (guard_ok(p - GUARD_BYTES) && guard_ok(p + bytes)); }

struct Staging {
  char* p = nullptr; size_t bytes = 0;
  void* get(size_t want) {
    if (want > bytes) {
      if (p) { check(); guarded_host_free(p); p = nullptr; bytes = 0; }
      p = guarded_host_alloc(want);
      bytes = want;
    }
    return p;
  }
  void check() const { if (!guarded_ok(p, bytes)) throw DeviceError("guard region of the pinned staging buffer was overwritten"); }
  // (freed only when it grows: it lives as long as the thread, and at thread exit the HIP runtime may already be gone)
};
// small device-written host words (partial sums, counters, flags): one pinned block per engine, taken from a pool
// and given back at destroy — no pinned allocation / free per uploaded model or per engine.  The pools are per thread
// and bounded: beyond POOL_MAX idle entries a returned block is freed / a returned stream destroyed.
constexpr size_t PINNED_WORDS_BYTES = 8 * 1024 + 256;   // 1024 partial sums, then words at +0 (stale counter), +64 (primal flag), +128 ... (chain abort report)
constexpr size_t POOL_MAX = 8;
struct PinnedPool {
  std::vector<char*> free_blocks;
  char* take() {
    if (!free_blocks.empty()) { char* p = free_blocks.back(); free_blocks.pop_back(); return p; }
    return guarded_host_alloc(PINNED_WORDS_BYTES);
  }
  void give(char* p) {
    if (!p) return;
    if (free_blocks.size() < POOL_MAX) free_blocks.push_back(p); else guarded_host_free(p);
  }
};
PinnedPool& pinned_pool() { static thread_local PinnedPool p; return p; }
// streams are pooled per thread and device: a long-lived process that creates and destroys thousands of engines does
// not churn HIP streams (each is a hardware queue with its own signals)
struct StreamPool {
  std::vector<std::pair<int, hipStream_t>> free_streams;
  hipStream_t take(int device) {
    for (size_t i = 0; i < free_streams.size(); ++i)
      if (free_streams[i].first == device) { hipStream_t s = free_streams[i].second; free_streams.erase(free_streams.begin() + i); return s; }
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
  }
  void give(int device, hipStream_t s) {
    if (!s) return;
    // LPMP_STREAM_POOL=0: destroy instead of pooling — the round-1 behaviour, kept as the A/B switch of the host-heap
    // corruption hunt (tests/fuzz_split.py)
    static const bool pooling = [] { const char* v = std::getenv("LPMP_STREAM_POOL"); return !(v && v[0] == '0'); }();
    if (pooling && free_streams.size() < POOL_MAX) free_streams.emplace_back(device, s); else (void)hipStreamDestroy(s);
  }
};
StreamPool& stream_pool() { static thread_local StreamPool p; return p; }
constexpr size_t STAGE_CHUNK = (size_t)32 << 20;
Staging& staging() { static thread_local Staging s; return s; }

// LPMP_DIRECT_COPIES=1: asynchronous copies straight to / from the caller's (pageable) memory followed by a stream
// synchronise — the round-1 original, kept as an A/B switch of the host-heap corruption hunt
bool direct_copies() { static const bool v = [] { const char* e = std::getenv("LPMP_DIRECT_COPIES"); return e && e[0] == '1'; }(); return v; }
void h2d(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (direct_copies()) { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream)); HIP_CHECK(hipStreamSynchronize(stream)); return; }
  for (size_t off = 0; off < bytes; off += STAGE_CHUNK) {
    const size_t n = std::min(STAGE_CHUNK, bytes - off);
    void* st = staging().get(std::min(bytes, STAGE_CHUNK));
    std::memcpy(st, (const char*)src + off, n);
    HIP_CHECK(hipMemcpyAsync((char*)dst + off, st, n, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  staging().check();
}
void d2h(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (direct_copies()) { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream)); HIP_CHECK(hipStreamSynchronize(stream)); return; }
  for (size_t off = 0; off < bytes; off += STAGE_CHUNK) {
    const size_t n = std::min(STAGE_CHUNK, bytes - off);
    void* st = staging().get(std::min(bytes, STAGE_CHUNK));
    HIP_CHECK(hipMemcpyAsync(st, (const char*)src + off, n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    staging().check();
    std::memcpy((char*)dst + off, st, n);
  }
}

template <class T, class V>
void fill_device(T*& dst, size_t& cap, const V& src, hipStream_t stream) {
  if (src.size() > cap) {
    if (dst) { HIP_CHECK(hipFree(dst)); dst = nullptr; cap = 0; }
    const size_t n = src.size() + src.size() / 4 + 16;
    HIP_CHECK(hipMalloc((void**)&dst, n * sizeof(T)));
    cap = n;
  }
  h2d(dst, src.data(), src.size() * sizeof(T), stream);
}

// keep: refill d's buffers in place where they are large enough (the scratch schedule of lpmp_compute_pass_custom)
void upload_schedule(const Schedule& s, DevSchedule& d, hipStream_t stream, bool keep = false, bool adaptive_built = false) {
  if (keep) { if (d.graph) { (void)hipGraphExecDestroy(d.graph); d.graph = nullptr; } if (d.graph_primal) { (void)hipGraphExecDestroy(d.graph_primal); d.graph_primal = nullptr; } }
  else d.release();
  d.launches = s.launches; d.n_levels = s.n_levels; d.n_recv = s.n_recv; d.n_send = s.n_send; d.alg_bytes = s.alg_bytes;
  d.adaptive_built = adaptive_built;
  fill_device(d.recs, d.recs_cap, s.recs, stream);
  fill_device(d.ops, d.ops_cap, s.ops, stream);
  fill_device(d.packets, d.packets_cap, s.packets, stream);
  d.release_chain();
  if (!s.chains.empty() && !adaptive_built) {
    auto up = [&](auto*& dst, const auto& v) {
      using T = std::remove_reference_t<decltype(*dst)>;
      HIP_CHECK(hipMalloc((void**)&dst, std::max<size_t>(1, v.size()) * sizeof(T)));
      if (!v.empty()) h2d(dst, v.data(), v.size() * sizeof(T), stream);
    };
    for (const ChainPlan& c : s.chains) {
      DevSchedule::DevChain dc;
      std::vector<ChainLaunchDev> lds;
      for (const auto& l : c.launches) lds.push_back({l.stride > 0 ? d.packets + l.pk_begin : nullptr, d.recs + l.rec_begin, d.ops, l.count, l.stride, l.flags});
      up(dc.launches, lds); up(dc.tk_launch, c.tk_launch); up(dc.tk_block, c.tk_block); up(dc.dep_off, c.dep_off); up(dc.dep, c.dep);
      dc.tickets = (int32_t)c.tk_launch.size();
      HIP_CHECK(hipMalloc((void**)&dc.done, std::max<size_t>(1, (size_t)dc.tickets) * sizeof(int32_t)));
      HIP_CHECK(hipMalloc((void**)&dc.next, sizeof(int32_t)));
      HIP_CHECK(hipMemsetAsync(dc.done, 0, std::max<size_t>(1, (size_t)dc.tickets) * sizeof(int32_t), stream));
      dc.kclass = c.kclass; dc.banded = c.banded; dc.level_loop = c.level_loop; dc.n_launches = (int32_t)c.launches.size();
      if (c.mailbox_rows > 0) {
        // a granule is valid when its tag is the epoch of the running launch: zeroed once, epochs start at 1
        const size_t bytes = (size_t)c.mailbox_rows * c.mailbox_width * 16;
        if (hipMalloc((void**)&dc.mailbox, bytes) != hipSuccess) {
          (void)hipGetLastError();
          throw DeviceError("no device memory for the mailbox of a deep schedule (" + std::to_string(bytes >> 20) +
                            " MiB: 16 bytes per label and mailbox send); LPMP_NO_MAILBOX=1 plans the same schedule with completion flags only");
        }
        HIP_CHECK(hipMemsetAsync(dc.mailbox, 0, bytes, stream));
      }
      d.chains.push_back(dc);
    }
    for (int32_t li : s.plain_launches) d.plain.push_back(s.launches[li]);
    HIP_CHECK(hipStreamSynchronize(stream));
    d.chain = true;
  }
}

void check_generic_limits(const Plan& p, const Schedule& s) {
  const int lim = generic_max_dual();
  for (const auto& lr : s.launches) {
    if (lr.kclass != KC_GENERIC) continue;
    for (int64_t i = lr.begin; i < lr.end; ++i) {
      const UpdRec& r = s.recs[i];
      const int own = (r.kind_flags & 15) == LPMP_F_VECTOR ? r.d0 : r.d0 + r.d1;
      if (p.force_generic && r.n_send > generic_max_adaptive_sends())
        throw UnsupportedError("adaptive sends: factor " + std::to_string(r.factor) + " has more active sends than the device kernel keeps improvements for");
      if (own > lim) throw UnsupportedError("factor " + std::to_string(r.factor) + ": dual size " + std::to_string(own) + " exceeds the device limit " + std::to_string(lim));
      for (int k = 0; k < r.n_recv + r.n_send; ++k)
        if (s.ops[r.op_begin + k].len > lim) throw UnsupportedError("message too long for the device kernels");
    }
  }
}

// one line, once per engine: a sweep of many dependent levels is latency-bound whatever executes it, and the order is the caller's
static void deep_schedule_note(lpmp_engine* e, int64_t n_levels, const char* what) {
  if (n_levels <= 64 || e->deep_note_given) return;
  e->deep_note_given = true;
  const char* q = std::getenv("LPMP_QUIET");
  if (q && q[0] == '1') return;
  std::fprintf(stderr, "lpmp: %s has %lld dependent levels (one launch step each): the factor order is the caller's input — "
               "lpmp_plan_suggest_order gives one with a level per colour (INTEGRATION.md 2a)\n", what, (long long)n_levels);
}

constexpr int64_t LAZY_SCHEDULES_MIN_FACTORS = (int64_t)1 << 20;
void ensure_device_schedules(lpmp_engine* e, int mode) {
  if (e->have_sched[mode]) return;
  for (int d = 0; d < 2; ++d) {
    plan_schedule(e->plan.get(), d, mode);
    deep_schedule_note(e, e->plan->sched_cache[d][mode].n_levels, d == 0 ? "the forward sweep" : "the backward sweep");
    check_generic_limits(e->plan->p, e->plan->sched_cache[d][mode]);
    upload_schedule(e->plan->sched_cache[d][mode], e->sched[d][mode], e->stream);
  }
  e->have_sched[mode] = true;
}

void ensure_pass_schedule(lpmp_engine* e, int mode) {
  if (e->have_pass[mode]) return;
  const bool timed_ = std::getenv("LPMP_PLAN_TIMES") != nullptr;
  auto t_last_ = std::chrono::steady_clock::now();
  auto lap_ = [&](const char* what) { if (!timed_) return; const auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "lpmp: pass schedule %-28s %.0f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last_).count()); t_last_ = now; };
  // (a three-level pass is the shape whose passes join: its own chain plan is only built if the joins do not check out)
  plan_pass_schedule(e->plan.get(), mode, false);
  deep_schedule_note(e, e->plan->pass_cache[mode].n_levels / 2, "a directional sweep");
  lap_("forward+backward planned");
  plan_rotation(e->plan.get(), mode);
  lap_("backward+forward planned, joins checked");
  if (!e->plan->rotation_ok[mode] && e->plan->pass_cache[mode].n_levels == 3) {
    e->plan->have_pass[mode] = false;
    plan_pass_schedule(e->plan.get(), mode, true);
    lap_("forward+backward planned again, with its chain plan");
  }
  check_generic_limits(e->plan->p, e->plan->pass_cache[mode]);
  upload_schedule(e->plan->pass_cache[mode], e->sched_pass[mode], e->stream);
  lap_("... uploaded");
  // LPMP_LAUNCH_LOG=<file> (profiling aid): one line per launch of the fused pass in execution order — level, class, records,
  // receives, sends, algorithmic bytes, packet stride — to lay beside the durations of a kernel trace (tools/launch_rates.py)
  if (const char* path = std::getenv("LPMP_LAUNCH_LOG")) {
    if (FILE* f = std::fopen(path, "w")) {
      std::fprintf(f, "level,kclass,records,receives,sends,bytes,stride\n");
      for (const auto& lr : e->plan->pass_cache[mode].launches)
        std::fprintf(f, "%d,%d,%lld,%lld,%lld,%lld,%d\n", lr.level, lr.kclass, (long long)(lr.end - lr.begin), (long long)lr.n_recv, (long long)lr.n_send, (long long)lr.bytes, lr.stride);
      std::fclose(f);
    }
  }
  e->rotation_ok[mode] = e->plan->rotation_ok[mode];
  if (e->rotation_ok[mode]) {
    upload_schedule(e->plan->bf_cache[mode], e->sched_bf[mode], e->stream);
    lap_("... uploaded");
    plan_rotation_chain(e->plan.get(), mode);
    lap_("block relations of the steps");
    e->plan->bf_cache[mode] = Schedule();
  }
  // the host copy is only needed for its summary
  Schedule& h = e->plan->pass_cache[mode];
  h.recs.clear(); h.recs.shrink_to_fit(); h.ops.clear(); h.ops.shrink_to_fit(); h.packets.clear(); h.packets.shrink_to_fit();
  e->have_pass[mode] = true;
}

// ensure_pass_schedule plans a three-level pass whose consecutive passes join WITHOUT a chain plan of its own (the joined
// launch never reads it).  A pass that is then run on its own after all — another send rule, one pass at a time while the joined
// launch is unavailable — gets the plan the first time that happens: on an HBM-sized model the banded Infinity-Cache launch of
// the single pass.  The joined-launch templates of that mode point into the schedule's device arrays and are dropped (rebuilt on demand).
void ensure_pass_chain_plan(lpmp_engine* e, int mode) {
  if (e->pass_chain_tried[mode] || !e->use_chain || !e->use_blocked_passes) return;
  e->pass_chain_tried[mode] = true;
  const DevSchedule& d = e->sched_pass[mode];
  if (!e->have_pass[mode] || !e->rotation_ok[mode] || d.chain || d.n_levels != 3 || !e->model_big) return;
  e->plan->have_pass[mode] = false;
  plan_pass_schedule(e->plan.get(), mode, true);
  check_generic_limits(e->plan->p, e->plan->pass_cache[mode]);
  HIP_CHECK(hipStreamSynchronize(e->stream));
  for (auto& kv : e->rot_chain[mode]) {
    auto& c = kv.second.dc;
    for (void* p : {(void*)c.launches, (void*)c.tk_launch, (void*)c.tk_block, (void*)c.dep_off, (void*)c.dep, (void*)c.done, (void*)c.next, (void*)c.mailbox}) if (p) (void)hipFree(p);
    e->rot_cache_bytes -= std::min(e->rot_cache_bytes, kv.second.dev_bytes);
  }
  e->rot_chain[mode].clear();
  e->rot_tile_set[mode] = lpmp_engine::TileSet();
  upload_schedule(e->plan->pass_cache[mode], e->sched_pass[mode], e->stream);
  Schedule& h = e->plan->pass_cache[mode];
  h.recs.clear(); h.recs.shrink_to_fit(); h.ops.clear(); h.ops.shrink_to_fit(); h.packets.clear(); h.packets.shrink_to_fit();
}

void issue_launches(lpmp_engine* e, const DevSchedule& s, bool timed, hipStream_t stream, int only_level = 0) {
  for (const auto& lr : s.launches) {
    if (only_level > 0 && lr.level != only_level) continue;
    hipEvent_t a = nullptr, b = nullptr;
    if (timed) { a = e->get_event(); b = e->get_event(); HIP_CHECK(hipEventRecord(a, stream)); }
    // UpdateFactorPrimal always sends 'shared' (reference factors_messages.hxx:2357-2359), whatever the send rule
    const int rule = e->rtype == LPMP_RTYPE_RESIDUAL ? SWEEP_RESIDUAL : e->rtype == LPMP_RTYPE_ADAPTIVE ? SWEEP_ADAPTIVE : 0;
    const int flags = (e->primal_pass ? SWEEP_PRIMAL : rule) | e->nt_flag;
    // primal pass over pairwise factors that round themselves: those records take the generic kernels (ensure_primal)
    const bool pw_rounds = e->primal_pass && e->d_pw_unary && kc_is_pw(lr.kclass);
    if (pw_rounds)
      launch_sweep(KC_GENERIC, s.recs, s.ops, e->d_dual, e->d_const, e->d_tabs, e->d_lb, e->d_primal, e->d_pw_unary, lr.begin, lr.end - lr.begin, flags, stream);
    else if (!(e->use_packed && lr.stride != 0 &&
          launch_sweep_packed(lr.kclass, lr.stride > 0 ? s.packets + lr.pk_begin : nullptr, s.recs + lr.begin, s.ops, lr.stride, e->d_dual,
                              e->d_const, e->d_lb, e->d_primal, lr.end - lr.begin, flags, stream)))
      launch_sweep(lr.kclass, s.recs, s.ops, e->d_dual, e->d_const, e->d_tabs, e->d_lb, e->d_primal, e->d_pw_unary, lr.begin, lr.end - lr.begin,
                   flags | (lr.kclass == KC_DENSE_BIG ? sweep_bigdim_flags(lr.max_dim) : 0), stream);   // (LDS of the streaming class: by the launch's label counts)
    if (timed) {
      HIP_CHECK(hipEventRecord(b, stream));
      e->pending.push_back({a, b, lr.kclass, lr.end - lr.begin, lr.n_recv, lr.bytes});
    }
  }
  HIP_CHECK(hipGetLastError());
}

// LPMP_CHAIN_TRACE=<file> (debugging): every chain run is followed by a synchronisation and its per-ticket time stamps
// (ticket in hand, predecessors seen, body done, published; 100 MHz) are written to the file together with the ticket
// -> launch map and the dependency lists: tools/chain_trace.py turns them into the latency budget of DESIGN.md 6
struct ChainTrace {
  long long* d = nullptr; int32_t n = 0;
  // ("%p" in the path: this process's id — several ranks on one box)
  static const char* path() {
    static const std::string p = [] {
      const char* e = std::getenv("LPMP_CHAIN_TRACE");
      std::string s = e ? e : "";
      const size_t k = s.find("%p");
      if (k != std::string::npos) s.replace(k, 2, std::to_string((long long)getpid()));
      return s;
    }();
    return p.empty() ? nullptr : p.c_str();
  }
  long long* begin(int32_t n_tickets, hipStream_t s) {
    if (!path()) return nullptr;
    n = n_tickets;
    HIP_CHECK(hipMalloc((void**)&d, (size_t)8 * n * sizeof(long long)));
    HIP_CHECK(hipMemsetAsync(d, 0, (size_t)8 * n * sizeof(long long), s));
    return d;
  }
  template <class DC> void end(const DC& c, hipStream_t s, const int32_t* d_abort = nullptr) {
    if (!d) return;
    HIP_CHECK(hipStreamSynchronize(s));
    // the dump of a run in which a wait gave up is kept under its own name (later runs do not overwrite it)
    std::string out = path();
    if (d_abort) { int32_t a = 0; HIP_CHECK(hipMemcpy(&a, d_abort, sizeof(a), hipMemcpyDeviceToHost)); if (a) out += ".aborted"; }
    std::vector<long long> st((size_t)8 * n);
    std::vector<int32_t> tl((size_t)n), off((size_t)n + 1);
    HIP_CHECK(hipMemcpy(st.data(), d, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(tl.data(), c.tk_launch, tl.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(off.data(), c.dep_off, off.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> dep((size_t)off[n]);
    if (!dep.empty()) HIP_CHECK(hipMemcpy(dep.data(), c.dep, dep.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d); d = nullptr;
    if (FILE* f = std::fopen(out.c_str(), "wb")) {
      const int64_t hdr[2] = {n, off[n]};
      std::fwrite(hdr, sizeof(hdr), 1, f);
      std::fwrite(st.data(), sizeof(long long), st.size(), f);
      std::fwrite(tl.data(), sizeof(int32_t), tl.size(), f);
      std::fwrite(off.data(), sizeof(int32_t), off.size(), f);
      std::fwrite(dep.data(), sizeof(int32_t), dep.size(), f);
      std::fclose(f);
    }
  }
};

// one sweep over a device schedule; long launch chains (row-major grids: one launch per anti-diagonal)
// are captured once into a hipGraph and replayed
void run_schedule(lpmp_engine* e, DevSchedule& s) {
  if (s.launches.empty()) return;
  if (e->timing) { issue_launches(e, s, true, e->stream); if (e->pending.size() > 4096) e->drain_timing(); return; }
  // (a primal pass rounds inside the packed kernels' chain form too; the generic chain kernels carry no labels, and
  // pairwise factors that round themselves need the generic kernels: those passes stay launch by launch)
  bool chain_ok = s.chain && e->use_chain;
  if (chain_ok && e->primal_pass) {
    if (e->d_pw_unary) chain_ok = false;
    for (const auto& c : s.chains) if (kc_width(c.kclass) == 0) chain_ok = false;
  }
  if (chain_ok) {
    if (!e->d_chain_abort) { HIP_CHECK(hipMalloc((void**)&e->d_chain_abort, CHAIN_ABORT_WORDS * sizeof(int32_t))); HIP_CHECK(hipMemsetAsync(e->d_chain_abort, 0, CHAIN_ABORT_WORDS * sizeof(int32_t), e->stream)); }
    // UpdateFactorPrimal always sends 'shared' (issue_launches)
    const int rule = e->primal_pass ? SWEEP_PRIMAL : e->rtype == LPMP_RTYPE_RESIDUAL ? SWEEP_RESIDUAL : e->rtype == LPMP_RTYPE_ADAPTIVE ? SWEEP_ADAPTIVE : 0;
    // classes are independent of each other (plan.cpp): the plain launches first, then one persistent launch per class
    if (!s.plain.empty()) {
      DevSchedule tmp;                       // a view: issue_launches only reads recs / ops / packets / launches
      tmp.recs = s.recs; tmp.ops = s.ops; tmp.packets = s.packets; tmp.launches = s.plain;
      try { issue_launches(e, tmp, false, e->stream); } catch (...) { tmp.recs = nullptr; tmp.ops = nullptr; tmp.packets = nullptr; throw; }
      tmp.recs = nullptr; tmp.ops = nullptr; tmp.packets = nullptr;
    }
    for (auto& c : s.chains) {
      if (c.level_loop) {
        // LPMP_LEVEL_TRACE=<file> (debugging): time stamps of the first 4000 levels, written after a synchronisation
        static const char* lt_path = std::getenv("LPMP_LEVEL_TRACE");
        long long* d_lt = nullptr;
        const size_t lt_n = 8 + 8 * 4000;
        if (lt_path) { HIP_CHECK(hipMalloc((void**)&d_lt, lt_n * sizeof(long long))); HIP_CHECK(hipMemset(d_lt, 0, lt_n * sizeof(long long))); debug_set_level_trace(d_lt); }
        if (!launch_level_loop(c.kclass, rule, c.launches, c.n_launches, e->d_dual, e->d_const, e->d_tabs, e->d_lb, e->stream))
          throw DeviceError("level loop: no kernel for class " + std::to_string(c.kclass));
        if (lt_path) {
          HIP_CHECK(hipStreamSynchronize(e->stream));
          std::vector<long long> h(lt_n);
          HIP_CHECK(hipMemcpy(h.data(), d_lt, lt_n * sizeof(long long), hipMemcpyDeviceToHost));
          debug_set_level_trace(nullptr); (void)hipFree(d_lt);
          if (FILE* f = std::fopen(lt_path, "wb")) { std::fwrite(h.data(), sizeof(long long), h.size(), f); std::fclose(f); }
        }
        continue;
      }
      HIP_CHECK(hipMemsetAsync(c.next, 0, sizeof(int32_t), e->stream));
      ChainTrace tr;
      const ChainArgsHost ca{c.dep_off, c.dep, c.done, c.next, e->d_chain_abort, c.tk_launch, c.tk_block, c.tickets, ++c.epoch, tr.begin(c.tickets, e->stream), nullptr, 0, c.mailbox, chain_timeout_ticks(), 0, 0, 0, 0, 0, 0, 0};
      if (!launch_chain(c.kclass, rule | (c.banded ? 0 : e->nt_flag), &ca, c.launches, e->d_dual, e->d_const, e->d_tabs, e->d_lb, e->d_primal, e->stream))
        throw DeviceError("chain executor: no kernel for class " + std::to_string(c.kclass));
      tr.end(c, e->stream, e->d_chain_abort);
    }
    HIP_CHECK(hipGetLastError());
    e->chain_ran = true;
    return;
  }
  // (graphs of up to ~20 k kernel nodes were exercised — C5; beyond 200 k the nodes are issued one by
  // one instead of instantiating a graph of that size)
  if (e->use_graph && s.launches.size() > 8 && s.launches.size() <= 200000) {
    hipGraphExec_t& exec = e->primal_pass ? s.graph_primal : s.graph;
    if (!exec) {
      hipGraph_t g = nullptr;
      if (!e->capture_stream) e->capture_stream = stream_pool().take(e->device);
      HIP_CHECK(hipStreamBeginCapture(e->capture_stream, hipStreamCaptureModeThreadLocal));
      try { issue_launches(e, s, false, e->capture_stream); }
      catch (...) { (void)hipStreamEndCapture(e->capture_stream, &g); if (g) (void)hipGraphDestroy(g); throw; }
      HIP_CHECK(hipStreamEndCapture(e->capture_stream, &g));
      HIP_CHECK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
      HIP_CHECK(hipGraphDestroy(g));
    }
    HIP_CHECK(hipGraphLaunch(exec, e->stream));
    return;
  }
  issue_launches(e, s, false, e->stream);
}

// weight / receive-mask rows of an iterator-range pass as they come over the C ABI: offsets start at 0 and do not
// decrease, and a row with entries needs the array it indexes
void check_rows(int64_t n, const int64_t* om_off, const double* om, const int64_t* mk_off, const uint8_t* mk) {
  if (n <= 0) return;
  if (om_off[0] != 0 || mk_off[0] != 0) throw std::runtime_error("omega / receive-mask offsets must start at 0");
  for (int64_t i = 0; i < n; ++i)
    if (om_off[i + 1] < om_off[i] || mk_off[i + 1] < mk_off[i]) throw std::runtime_error("omega / receive-mask offsets must not decrease");
  if (!om && om_off[n] > 0) throw std::runtime_error("omega array missing");
  if (!mk && mk_off[n] > 0) throw std::runtime_error("receive-mask array missing");
}

// n joined passes (H, W, (K, W)^(n-1), T) as ONE persistent launch of the chain executor.  Tickets are not taken step by
// step but in a skewed order: inside a group of `depth` consecutive steps, band j of the group's d-th step comes at time
// j + lag * d, so that the pairwise tables a step reads are read again by the next step while they are still in the
// 256 MiB Infinity Cache (every table is needed by both of its endpoints, i.e. by consecutive steps).  Only the order
// changes: the dependency flags keep every result identical to the sequential sweeps.  Returns nullptr when the model
// does not qualify (then the steps run as one launch each).
// Calls of n > ROT_EXPLICIT_MAX passes use a PERIODIC template instead of explicit lists: with `depth` even the groups of
// `depth` steps from the second one on are all alike — K, W, K, W with the same bands in the same order, their dependencies
// the same offsets into themselves and into the group before — so the lists of a template call (prologue = the first two
// groups, ONE more group = the period, then the tail: T, or K W T for an odd pass count) describe every pass count of that
// parity: the kernel maps ticket t to (template ticket, copy of the period) (kernels.hip, chain_ticket_ref), and the
// completion flags are a ring of a few groups.  Host work, upload and device memory of an n-pass launch no longer depend on n.
constexpr int ROT_EXPLICIT_MAX = 7;
// The window of the skewed order — (bands, lag, depth) — from the model (round 6).  A table is read by two consecutive steps; with
// bands of a step issued at time b + lag * d (d = the step's place in its group of `depth` steps) the second read comes
// lag * depth bands after the first, and it finds the table in the 256 MiB Infinity Cache while that window stays below it.
// The lag has to cover the REACH of the dependencies — how far ahead in the block list a block's predecessors lie: one grid row —
// plus slack: a ticket whose predecessors were issued fewer tickets ago than there are resident workgroups (256 CUs x 3) is drawn
// while they are still running, and its workgroup waits.  In ticket order the steps of a group are interleaved, so a slack of
// S tickets is S / depth blocks of one step.  Measured (profiles/r06_blocked_pass_probe_*.txt, 32 labels, bands of 16 MiB, a
// block = 156 KB): 1024^2 (row 20 MB) lag 3, depth 4: 5.12 ms per pass (lag 2: 5.24, lag 4: 5.18); 1536^2 (row 30 MB) lag 3 / 4 / 5:
// 12.17 / 11.94 / 12.85 (11.5 = 2.25 times the 1024^2 time; 14.5 launch by launch); 2048^2 (row 40 MB) 25.3 with lag 3 (slack
// 260 tickets: no better than one launch per step, 26.0), 21.3 with lag 4 (700 tickets; 20.4 = four times), 22.6 with lag 5
// (window 336 MB: the reuse goes); 3072^2 (row 60 MB) 65.4 with lag 3 / depth 4 (57.2 launch by launch), 50.3 with lag 6 / depth 2
// (46 = nine times; depth 3 and 4 with lags 5-6: 50.2-52.1).  So: a slack of 700 tickets behind the reach; depth 4 while
// (reach + slack) * 4 stays under 275 MiB, else 2 (half of the second reads instead of three quarters, but they hit); the lag
// stretched to a window of 200 MiB where the reach leaves room; `fits` = false when even depth 2 cannot hold the window (then the
// tiled order below, or one launch per step).
struct RotGeometry { int bands = 1, lag = 3, depth = 4; bool fits = true; double reach_bytes = 0; };
static RotGeometry rot_geometry(const lpmp_engine* e, const RotationInfo& ri) {
  RotGeometry g;
  g.lag = std::max(1, e->rot_lag); g.depth = std::max(1, e->rot_depth);
  if (!ri.valid) return g;
  g.bands = e->rot_bands > 0 ? e->rot_bands : (int)std::max<int64_t>(1, std::min<int64_t>(ri.t[1].nb, ri.t[1].bytes / ((int64_t)16 << 20)));
  if (e->rot_bands > 0) return g;                   // bands forced (tests, probes): lag and depth as given or their defaults
  constexpr double MiB = 1048576.0, SLACK_TICKETS = 700, WINDOW_TARGET = 200 * MiB, WINDOW_MAX = 275 * MiB;
  const double step_bytes = (double)ri.t[1].bytes, band_bytes = step_bytes / g.bands;
  const double slack1 = SLACK_TICKETS * step_bytes / (double)std::max<int32_t>(1, ri.t[1].nb);   // the slack as bytes of ONE step's block list, depth 1
  g.reach_bytes = ri.reach * step_bytes;
  if (!e->rot_depth_set) g.depth = (g.reach_bytes + slack1 / 4) * 4 <= WINDOW_MAX ? 4 : 2;
  const double need = g.reach_bytes + slack1 / g.depth;
  // (rounded up from .3: a band short on slack costs more than a band of window — 1536^2: 3.4 bands -> 4)
  if (!e->rot_lag_set) g.lag = std::max(2, (int)std::floor(std::max(need, WINDOW_TARGET / g.depth) / band_bytes + 0.7));
  g.fits = e->rot_depth_set || e->rot_lag_set || need * g.depth <= 1.25 * WINDOW_MAX;
  return g;
}
// Tiled ticket order (round 6; chosen in rotation_chain below, LPMP_ROT_TILES overrides).  The band order walks a step's block list in memory order, so its lag has
// to cover how far ahead a block's predecessors lie IN THAT LIST — a grid row, a z-slice of a 3-D grid — whatever the distance in
// the graph is.  Tiles are compact in the GRAPH instead: sets of about T blocks of either alternating step template (W and K; H and
// T update K's factors), grown breadth-first over the block dependencies.  Inside a group of `depth` steps a block runs in the phase
// of its own tile or of the latest tile one of its predecessors ran in, whichever is later — the skew of a time-tiled stencil
// without any geometry: valid by construction, a table is read again by the next step T blocks later, and nothing grows with the
// width of the grid.
static int32_t grow_tiles(const RotationInfo& ri, int64_t T, std::vector<int32_t>& tile_w, std::vector<int32_t>& tile_k, double& radius) {
  const int64_t nw = ri.t[1].nb, nk = ri.t[2].nb, nn = nw + nk;
  std::vector<int64_t> deg((size_t)nn + 1, 0);
  auto each_edge = [&](auto f) {      // W block j <-> K block p (kind 2: W after K), K block j <-> W block p (kind 3: K after W)
    for (int kind = 2; kind <= 3; ++kind) {
      const int64_t nb = kind == 2 ? nw : nk;
      for (int64_t j = 0; j < nb; ++j)
        for (int64_t q = ri.off[kind][j]; q < ri.off[kind][j + 1]; ++q)
          if (ri.delta[kind][q] == 1) { const int64_t a = kind == 2 ? j : nw + j, b = kind == 2 ? nw + ri.block[kind][q] : ri.block[kind][q]; f(a, b); }
    }
  };
  each_edge([&](int64_t a, int64_t b) { ++deg[a + 1]; ++deg[b + 1]; });
  for (int64_t i = 0; i < nn; ++i) deg[i + 1] += deg[i];
  std::vector<int32_t> adj((size_t)deg[nn]);
  { std::vector<int64_t> cur(deg.begin(), deg.end() - 1); each_edge([&](int64_t a, int64_t b) { adj[cur[a]++] = (int32_t)b; adj[cur[b]++] = (int32_t)a; }); }
  std::vector<int32_t> tile((size_t)nn, -1), queue, hops;
  int32_t n_tiles = 0;
  double radius_sum = 0; int64_t full_tiles = 0;    // hops from the seed to the last block of a tile that reached its size
  for (int64_t seed0 = 0; seed0 < std::max(nw, nk); ++seed0)
    for (int64_t seed : {seed0 < nw ? seed0 : (int64_t)-1, seed0 < nk ? nw + seed0 : (int64_t)-1}) {
      if (seed < 0 || tile[seed] >= 0) continue;
      queue.assign(1, (int32_t)seed); hops.assign(1, 0);
      int64_t taken = 0; int32_t last_hops = 0;
      for (size_t head = 0; head < queue.size() && taken < 2 * T; ++head) {
        const int32_t v = queue[head];
        if (tile[v] >= 0) continue;
        tile[v] = n_tiles; ++taken; last_hops = hops[head];
        for (int64_t q = deg[v]; q < deg[v + 1]; ++q) if (tile[adj[q]] < 0) { queue.push_back(adj[q]); hops.push_back(hops[head] + 1); }
      }
      if (taken >= 2 * T) { radius_sum += last_hops; ++full_tiles; }
      ++n_tiles;
    }
  radius = full_tiles ? radius_sum / (double)full_tiles : 0.0;
  tile_w.assign(tile.begin(), tile.begin() + nw);
  tile_k.assign(tile.begin() + nw, tile.end());
  return n_tiles;
}

// how much of a tile is left after sd steps: phases of a steady-state group (K, W, K, W, ...) of 8 steps, share of delayed blocks per step
static void tile_delays(const RotationInfo& ri, const std::vector<int32_t>& tile_w, const std::vector<int32_t>& tile_k, double (&delayed)[8]) {
  std::vector<int32_t> ph[3];
  for (int sd = 0; sd < 8; ++sd) {
    const int kd = sd % 2 == 0 ? 3 : 2;
    const std::vector<int32_t>& tl = sd % 2 == 0 ? tile_k : tile_w;
    const int64_t nb = (int64_t)tl.size();
    std::vector<int32_t>& cur = ph[sd % 3];
    cur.resize((size_t)nb);
    int64_t late = 0;
    for (int64_t j = 0; j < nb; ++j) {
      int32_t p = tl[j];
      if (sd > 0)
        for (int64_t q = ri.off[kd][j]; q < ri.off[kd][j + 1]; ++q) {
          const int dl = ri.delta[kd][q];
          if (dl <= sd) p = std::max(p, ph[(sd - dl) % 3][ri.block[kd][q]]);
        }
      cur[j] = p;
      late += p != tl[j];
    }
    delayed[sd] = nb ? (double)late / (double)nb : 0.0;
  }
}

lpmp_engine::RotChain* rotation_chain(lpmp_engine* e, int mode, int n_call) {
  RotGeometry geo = rot_geometry(e, e->plan->rot[mode]);
  // Band order or tiled order?  Measured (profiles/r06_tile_sweep.txt, r06_blocked_pass_probe_tiles_*.txt; tiles of 1024 blocks): where
  // the band order runs at depth 4 it is as good or better (1024^2: 5.09 against 5.31 ms per pass, 2048^2: 21.35 / 21.26); where
  // the reach of the dependencies has pushed it to depth 2 or out of the cache, tiles win — 3072^2: 49.5 -> 47.7 ms, 256 x 4096: 5.89
  // -> 5.44, 128 x 8192 (no band order fits: 6.95 launch by launch) -> 5.37, 3-D grids 96^3 x 32 labels: 8.88 -> 7.27, 128^3 x 16
  // labels: 6.23 -> 5.72.  Calls of fewer than 4 passes keep the band order (single passes: 9.0 against 6.8 ms at 1024^2).
  // Depth: a block whose predecessor ran in a later tile is delayed to that tile's phase — one more shell of every tile per step: the
  // deepest even depth (up to 8) whose LAST step still runs at least half of its blocks in their own tile's phase (tile_delays: the
  // flat tiles of a 2-D grid lose about 4 % per step: depth 8; the balls of a 3-D grid 16 %: depth 4 — measured there: depth 2 / 4 /
  // 6 / 8 = 7.9 / 7.27 / 7.28 / 7.6 ms per pass at 96^3 x 32 labels, 5.92 / 5.72 / - / 7.24 at 128^3 x 16).
  bool tiled = false;
  lpmp_engine::TileSet* ts = nullptr;
  {
    const RotationInfo& ri0 = e->plan->rot[mode];
    const bool worthwhile = ri0.valid && kc_is_dense(ri0.kclass) && !kc_is_var(ri0.kclass) &&
                            (e->rot_bands > 0 || (e->model_big && ri0.t[1].bytes >= ((int64_t)64 << 20)));
    const int want = !worthwhile || ri0.t[0].nb != ri0.t[2].nb || ri0.t[3].nb != ri0.t[2].nb ? 0
                   : e->rot_tiles_set ? e->rot_tiles
                   : (e->rot_bands <= 0 && (geo.depth != 4 || !geo.fits) && n_call >= 4 ? 1024 : 0);
    if (want > 0) {
      ts = &e->rot_tile_set[mode];
      if (!ts->built || ts->T != want) { ts->n = grow_tiles(ri0, want, ts->w, ts->k, ts->radius); tile_delays(ri0, ts->w, ts->k, ts->delayed); ts->T = want; ts->built = true; }
      tiled = true;
      if (!e->rot_depth_set) { geo.depth = 2; for (int d = 4; d <= 8; d += 2) if (ts->delayed[d - 1] <= 0.5) geo.depth = d; }
    }
  }
  const int depth = geo.depth;
  // template: groups 0, 1 (prologue), 2 (the period) and a tail as long as the call's: r = (2 n + 1) mod depth steps
  const int tail = depth % 2 == 0 ? (2 * n_call + 1) % depth : 0;
  const int n_template = (3 * depth + tail - 1) / 2;
  const bool periodic = depth % 2 == 0 && n_call > ROT_EXPLICIT_MAX && n_call >= n_template && !std::getenv("LPMP_ROT_EXPLICIT");
  const int n = periodic ? n_template : n_call;
  const int key = periodic ? -tail : n_call;
  auto it = e->rot_chain[mode].find(key);
  if (it != e->rot_chain[mode].end()) { it->second.last_use = ++e->rot_clock; return it->second.n_steps > 0 ? &it->second : nullptr; }
  // bound the cache in bytes: drop the least recently used built chains (of any mode) until the new one fits; the stream is drained first
  auto evict_for = [&](size_t need) {
    bool drained = false;
    while (e->rot_cache_bytes + need > e->rot_cache_limit) {
      std::map<int, lpmp_engine::RotChain>* vm = nullptr; std::map<int, lpmp_engine::RotChain>::iterator victim;
      for (auto& m : e->rot_chain)
        for (auto i2 = m.begin(); i2 != m.end(); ++i2)
          if (i2->second.n_steps > 0 && (!vm || i2->second.last_use < victim->second.last_use)) { vm = &m; victim = i2; }
      if (!vm) break;
      if (!drained) { HIP_CHECK(hipStreamSynchronize(e->stream)); drained = true; }
      auto& c = victim->second.dc;
      for (void* p : {(void*)c.launches, (void*)c.tk_launch, (void*)c.tk_block, (void*)c.dep_off, (void*)c.dep, (void*)c.done, (void*)c.next, (void*)c.mailbox}) if (p) (void)hipFree(p);
      e->rot_cache_bytes -= std::min(e->rot_cache_bytes, victim->second.dev_bytes);
      vm->erase(victim);
    }
  };
  lpmp_engine::RotChain& rc = e->rot_chain[mode][key];           // n_steps == 0: tried, not possible
  rc.last_use = ++e->rot_clock;
  const RotationInfo& ri = e->plan->rot[mode];
  const bool verbose = std::getenv("LPMP_ROT_VERBOSE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  auto no = [&](const char* why) -> lpmp_engine::RotChain* { if (verbose) std::fprintf(stderr, "lpmp: %d passes stay one launch per step: %s\n", n_call, why); return nullptr; };
  if (!ri.valid) return no("the pass does not have the H, W, K, T shape of one packed class");
  const int n_steps = 2 * n + 1;
  std::vector<int> tmpl(n_steps), kind(n_steps, -1);
  tmpl[0] = 0;
  for (int s = 1; s < n_steps - 1; ++s) tmpl[s] = (s & 1) ? 1 : 2;
  tmpl[n_steps - 1] = 3;
  for (int s = 1; s < n_steps; ++s) kind[s] = s == n_steps - 1 ? (n == 1 ? 5 : 4) : s == 1 ? 0 : s == 2 ? 1 : (s & 1) ? 2 : 3;
  std::vector<int64_t> base(n_steps + 1, 0);
  for (int s = 0; s < n_steps; ++s) base[s + 1] = base[s] + ri.t[tmpl[s]].nb;
  const int64_t N = base[n_steps];
  if (N > (int64_t)48 << 20) return no("too many tickets");                   // too many tickets for one launch: the caller splits the passes
  // a model whose tables fit the caches gains nothing from the order and is launch-bound: one launch per step then
  // (nor does one that fits the Infinity Cache as a whole: plain launches already re-read it on-die, and the chain's
  // agent-scope accesses only cost — C2, 512 x 512 8-label Potts: 0.065 ms per pass as launches, 0.10 as a chain)
  if (e->rot_bands <= 0 && (ri.t[1].bytes < ((int64_t)64 << 20) || !e->model_big)) return no("the model fits the caches");
  // (the run-time-dims classes read their tables with 8-byte loads of rows that are not line-aligned: as a chain in
  // Infinity-Cache order 1024 x 1024 x 21 labels takes 5.48 ms per pass against 4.38 launch by launch)
  if (e->rot_bands <= 0 && kc_is_var(ri.kclass)) return no("run-time-dims class");
  // (Potts steps stream message vectors only, at 7.6 TB/s with the non-temporal policy; as a chain their agent-scope
  // vector loads make 2048 x 2048 x 32 labels 6.4 ms per pass against 3.2)
  if (e->rot_bands <= 0 && !kc_is_dense(ri.kclass)) return no("no pairwise tables to re-read (Potts)");
  // bands: about 16 MiB of algorithmic bytes per band of a step.  What a group keeps alive between two reads of a table is
  // lag * depth bands (3 * 4 * 16 MiB = 192 MiB of the 256 MiB Infinity Cache); measured on C3: windows of 200-230 MB are
  // the fastest whatever the split (1024:3:4 5.09, 2048:4:6 5.03, 1536:3:6 5.09 ms per pass), 290 MB and more lose the
  // reuse (1024:3:5 5.67, 1024:3:6 6.37), lag 2 leaves the waiting workgroups less slack (1024:2:4 5.24)
  // (round 6: lag and depth follow the reach of the dependencies — rot_geometry above; C3 keeps 3 and 4)
  if (!geo.fits && !tiled) return no("a step's dependencies reach further than the Infinity Cache window can cover");
  const int bands = geo.bands;
  static const std::vector<int32_t> no_tiles;
  const std::vector<int32_t>& tile_w = tiled ? ts->w : no_tiles;
  const std::vector<int32_t>& tile_k = tiled ? ts->k : no_tiles;
  const int32_t n_tiles = tiled ? ts->n : 0;
  if (tiled && verbose) std::fprintf(stderr, "lpmp:   %d tiles of about %d blocks per step, radius %.1f hops, delayed after 1 / 3 / 5 / 7 steps: %.2f / %.2f / %.2f / %.2f\n", n_tiles, ts->T, ts->radius,
                                     ts->delayed[1], ts->delayed[3], ts->delayed[5], ts->delayed[7]);
  std::vector<int32_t> new_of((size_t)N), tk_launch((size_t)N), tk_block((size_t)N);
  std::vector<int64_t> group_begin;                                 // first ticket of every group of `depth` steps
  auto band_begin = [](int64_t b, int64_t nb, int64_t bands_) { return (b * nb + bands_ - 1) / bands_; };   // first block of band b
  for (int lag = geo.lag; lag <= std::max(16, 2 * geo.lag); ++lag) {
    int64_t at = 0;
    group_begin.clear();
    if (tiled) {
      // phase of (step of the group, block) = max(own tile, phases of its predecessors inside the group); tickets by (phase, step, block)
      std::vector<int32_t> ph[3];
      std::vector<int64_t> bucket;
      std::vector<std::vector<int32_t>> key_of((size_t)depth);
      for (int s0 = 0; s0 < n_steps; s0 += depth) {
        group_begin.push_back(at);
        const int d = std::min(depth, n_steps - s0);
        bucket.assign((size_t)n_tiles * d + 1, 0);
        for (int sd = 0; sd < d; ++sd) {
          const int s = s0 + sd;
          const int64_t nb = ri.t[tmpl[s]].nb;
          const std::vector<int32_t>& tl = tmpl[s] == 1 ? tile_w : tile_k;
          std::vector<int32_t>& cur = ph[sd % 3];
          cur.resize((size_t)nb);
          key_of[sd].resize((size_t)nb);
          const int kd = kind[s];
          for (int64_t j = 0; j < nb; ++j) {
            int32_t p = tl[j];
            if (kd >= 0)
              for (int64_t q = ri.off[kd][j]; q < ri.off[kd][j + 1]; ++q) {
                const int dl = ri.delta[kd][q];
                if (dl <= sd) p = std::max(p, ph[(sd - dl) % 3][ri.block[kd][q]]);
              }
            cur[j] = p;
            key_of[sd][j] = p * d + sd;
            ++bucket[(size_t)key_of[sd][j] + 1];
          }
        }
        for (size_t k = 0; k + 1 < bucket.size(); ++k) bucket[k + 1] += bucket[k];
        for (int sd = 0; sd < d; ++sd) {
          const int s = s0 + sd;
          const int64_t nb = ri.t[tmpl[s]].nb;
          for (int64_t j = 0; j < nb; ++j) {
            const int64_t t = at + bucket[key_of[sd][j]]++;
            new_of[base[s] + j] = (int32_t)t; tk_launch[t] = s; tk_block[t] = (int32_t)j;
          }
        }
        for (int sd = 0; sd < d; ++sd) at += ri.t[tmpl[s0 + sd]].nb;
      }
    } else
    for (int s0 = 0; s0 < n_steps; s0 += depth) {
      group_begin.push_back(at);
      const int d = std::min(depth, n_steps - s0);
      for (int64_t tau = 0; tau < bands + (int64_t)lag * (d - 1); ++tau)
        for (int sd = 0; sd < d; ++sd) {
          const int64_t b = tau - (int64_t)lag * sd;
          if (b < 0 || b >= bands) continue;
          const int s = s0 + sd;
          const int64_t nb = ri.t[tmpl[s]].nb;
          for (int64_t j = band_begin(b, nb, bands); j < band_begin(b + 1, nb, bands); ++j) {
            new_of[base[s] + j] = (int32_t)at; tk_launch[at] = s; tk_block[at] = (int32_t)j; ++at;
          }
        }
    }
    group_begin.push_back(at);
    if (at != N) throw std::runtime_error("rotation chain: ticket count");
    if (verbose) std::fprintf(stderr, "lpmp:   %d passes, lag %d: order after %.0f ms\n", n, lag, since());
    // every predecessor must come earlier
    bool ok = true;
    for (int s = 1; s < n_steps && ok; ++s) {
      const int kd = kind[s];
      const auto& off = ri.off[kd];
      for (int64_t j = 0; j < ri.t[tmpl[s]].nb && ok; ++j)
        for (int64_t q = off[j]; q < off[j + 1]; ++q)
          if (new_of[base[s - ri.delta[kd][q]] + ri.block[kd][q]] >= new_of[base[s] + j]) {
            if (verbose) std::fprintf(stderr, "lpmp:   lag %d: step %d (kind %d) block %lld of %lld needs block %d of step %d (%lld blocks), %d bands\n", lag, s, kd,
                                      (long long)j, (long long)ri.t[tmpl[s]].nb, ri.block[kd][q], s - ri.delta[kd][q], (long long)ri.t[tmpl[s - ri.delta[kd][q]]].nb, bands);
            ok = false; break;
          }
    }
    if (verbose) std::fprintf(stderr, "lpmp:   checked after %.0f ms (%s)\n", since(), ok ? "valid" : "a dependency points forward");
    if (!ok && tiled) { tiled = false; if (!geo.fits) return no("the tiled order broke a dependency and the band order does not fit"); --lag; continue; }
    if (!ok) continue;
    // dependencies in ticket order
    std::vector<int32_t> dep_off((size_t)N + 1, 0);
    for (int s = 1; s < n_steps; ++s) {
      const auto& off = ri.off[kind[s]];
      for (int64_t j = 0; j < ri.t[tmpl[s]].nb; ++j) dep_off[new_of[base[s] + j] + 1] = (int32_t)(off[j + 1] - off[j]);
    }
    for (int64_t i = 0; i < N; ++i) dep_off[i + 1] += dep_off[i];
    std::vector<int32_t> dep((size_t)dep_off[N]);
    for (int s = 1; s < n_steps; ++s) {
      const int kd = kind[s];
      const auto& off = ri.off[kd];
      for (int64_t j = 0; j < ri.t[tmpl[s]].nb; ++j) {
        int32_t* dst = dep.data() + dep_off[new_of[base[s] + j]];
        for (int64_t q = off[j]; q < off[j + 1]; ++q) *dst++ = new_of[base[s - ri.delta[kd][q]] + ri.block[kd][q]];
      }
    }
    if (verbose) std::fprintf(stderr, "lpmp:   dependency lists after %.0f ms (%zu)\n", since(), dep.size());
    int32_t ring = 0;
    if (periodic) {
      // the template is [group 0][group 1][group 2 = the period][tail]; what the kernel's map relies on, checked here:
      // groups 1 and 2 are the same tickets in the same order, the period's and the tail's dependencies all lie in the
      // group before the period or later (they move with the copy), the prologue's before the period
      if (group_begin.size() != 5) throw std::runtime_error("rotation chain: template groups");
      const int64_t g1 = group_begin[1], g2 = group_begin[2], g3 = group_begin[3], P = g3 - g2;
      bool fine = g2 - g1 == P;
      for (int64_t t = g2; t < g3 && fine; ++t) {
        fine = tk_launch[t] == tk_launch[t - P] + depth && tk_block[t] == tk_block[t - P];
        for (int64_t q = dep_off[t]; q < dep_off[t + 1] && fine; ++q) fine = dep[q] >= g1;
      }
      for (int64_t t = g3; t < N && fine; ++t) for (int64_t q = dep_off[t]; q < dep_off[t + 1] && fine; ++q) fine = dep[q] >= g2;
      for (int64_t t = 0; t < g2 && fine; ++t) for (int64_t q = dep_off[t]; q < dep_off[t + 1] && fine; ++q) fine = dep[q] < g2;
      // (group 2's dependencies into group 1 must be what a later copy's are into the copy before it: same relative offsets
      // as group 1's own... group 1 reaches into group 0, whose order differs, so that cannot be compared — the step kinds
      // of groups >= 2 are identical by construction: kind[s] depends on the parity of s only from s = 3 on)
      if (!fine) throw std::runtime_error("rotation chain: the template is not periodic");
      rc.per_begin = (int32_t)g2; rc.per_len = (int32_t)P;
      // flags: a ring of three groups — a dependency reaches at most into the group before, and a ticket may only publish
      // into a slot whose previous occupant (a ring earlier) has published (kernels.hip, chain_wait)
      ring = (int32_t)(3 * P);
      if ((int64_t)48 * (int64_t)(2 * P) / ring + 64 >= (1 << CHAIN_GEN_BITS)) throw std::runtime_error("rotation chain: ring too small for its generation counter");
    }
    std::vector<ChainLaunchDev> lds;
    for (int s = 0; s < n_steps; ++s) {
      const auto& t = ri.t[tmpl[s]];
      const DevSchedule& ds = t.sched == 0 ? e->sched_pass[mode] : e->sched_bf[mode];
      // per-pass bound rows (only written when the launch is given rows: speculative batches): W of pass i (step 2 i + 1)
      // and K after pass i (step 2 i + 2) write row i, for the passes i = 0 ... n - 2 that have a seam behind them
      // (periodic template: the tail's last W is the last W of ANY call of this parity)
      int32_t hist = 0;
      // (periodic template: every W carries its row — whether it has a seam behind it depends on the call, and the kernel drops
      // rows >= ChainArgs::hist_rows)
      if (tmpl[s] == 1 && ((s - 1) / 2 < n - 1 || periodic)) hist = HIST_END | (((s - 1) / 2) << 2);
      if (tmpl[s] == 2) hist = HIST_MID | (((s - 2) / 2) << 2);
      lds.push_back({t.lr.stride > 0 ? ds.packets + t.lr.pk_begin : nullptr, ds.recs + t.lr.begin, ds.ops, t.lr.end - t.lr.begin, t.lr.stride, hist});
      rc.factors += t.factors; rc.recv += t.recv; rc.bytes += t.bytes;
      if (periodic && s >= 2 * depth && s < 3 * depth) { rc.per_factors += t.factors; rc.per_recv += t.recv; rc.per_bytes += t.bytes; }
    }
    // (periodic: the kernel looks a ticket's launch up at its TEMPLATE step — the steps of a later copy of the period are the
    // same K / W launches, and the tail's the same K, W, T; only the bound row in `pad` moves with the copy, depth / 2 rows
    // per copy — so this table, too, is the template's whatever the call's pass count)
    const size_t n_done = periodic ? (size_t)ring : (size_t)N;
    rc.dev_bytes = lds.size() * sizeof(ChainLaunchDev) + (tk_launch.size() + tk_block.size() + dep_off.size() + dep.size() + n_done + 1) * sizeof(int32_t);
    evict_for(rc.dev_bytes);
    auto up = [&](auto*& dst, const auto& v) {
      using T = std::remove_reference_t<decltype(*dst)>;
      HIP_CHECK(hipMalloc((void**)&dst, std::max<size_t>(1, v.size()) * sizeof(T)));
      if (!v.empty()) h2d(dst, v.data(), v.size() * sizeof(T), e->stream);
    };
    auto& dc = rc.dc;
    try {
      up(dc.launches, lds); up(dc.tk_launch, tk_launch); up(dc.tk_block, tk_block); up(dc.dep_off, dep_off); up(dc.dep, dep);
      dc.tickets = (int32_t)N; dc.kclass = ri.kclass;
      HIP_CHECK(hipMalloc((void**)&dc.done, n_done * sizeof(int32_t)));
      HIP_CHECK(hipMalloc((void**)&dc.next, sizeof(int32_t)));
      HIP_CHECK(hipMemsetAsync(dc.done, 0, n_done * sizeof(int32_t), e->stream));
      HIP_CHECK(hipStreamSynchronize(e->stream));
    } catch (...) {
      // nothing half-built stays behind: the entry goes (a later call tries again), the bytes were never counted
      for (void* p : {(void*)dc.launches, (void*)dc.tk_launch, (void*)dc.tk_block, (void*)dc.dep_off, (void*)dc.dep, (void*)dc.done, (void*)dc.next}) if (p) (void)hipFree(p);
      e->rot_chain[mode].erase(key);
      throw;
    }
    e->rot_cache_bytes += rc.dev_bytes;
    rc.n_steps = n_steps; rc.periodic = periodic; rc.n_tmpl = n; rc.depth = depth; rc.ring = ring;
    if (verbose)
      std::fprintf(stderr, "lpmp: %d passes as one launch%s: %lld tickets, %d bands, lag %d, depth %d (reach %.1f MB of %.1f MB per band); built and uploaded in %.0f ms\n", n,
                   periodic ? (tiled ? " (periodic template, tiled order)" : " (periodic template)") : tiled ? " (tiled order)" : "", (long long)N, bands, lag, depth, geo.reach_bytes / 1e6, (double)ri.t[1].bytes / bands / 1e6, since());
    return &rc;
  }
  return no("no band order keeps the dependencies backwards");
}

bool run_rotation_chain(lpmp_engine* e, int mode, int n, double* lb_hist = nullptr) {
  if (!e->use_chain || !e->use_blocked_passes || e->primal_pass || e->rtype != LPMP_RTYPE_SHARED) return false;
  lpmp_engine::RotChain* rc = rotation_chain(e, mode, n);
  if (!rc) return false;
  if (!e->d_chain_abort) { HIP_CHECK(hipMalloc((void**)&e->d_chain_abort, CHAIN_ABORT_WORDS * sizeof(int32_t))); HIP_CHECK(hipMemsetAsync(e->d_chain_abort, 0, CHAIN_ABORT_WORDS * sizeof(int32_t), e->stream)); }
  auto& c = rc->dc;
  HIP_CHECK(hipMemsetAsync(c.next, 0, sizeof(int32_t), e->stream));
  // periodic template: the period runs once in the template and `extra` more times in this call
  const int extra = rc->periodic ? (n - rc->n_tmpl) / (rc->depth / 2) : 0;
  if (rc->periodic && (extra < 0 || rc->n_tmpl + extra * (rc->depth / 2) != n)) throw std::runtime_error("rotation chain: pass count does not fit the template");
  if (rc->periodic && c.epoch >= (1 << (31 - CHAIN_GEN_BITS)) - 2) {   // the epoch shares the flag word with the generation: start over
    HIP_CHECK(hipMemsetAsync(c.done, 0, (size_t)rc->ring * sizeof(int32_t), e->stream));
    c.epoch = 0;
  }
  ChainTrace tr;
  const int32_t n_tickets = c.tickets + extra * rc->per_len;
  const ChainArgsHost ca{c.dep_off, c.dep, c.done, c.next, e->d_chain_abort, c.tk_launch, c.tk_block, n_tickets, ++c.epoch, rc->periodic ? nullptr : tr.begin(c.tickets, e->stream),
                         lb_hist, lb_hist ? e->plan->p.nf : 0, nullptr, chain_timeout_ticks(),
                         rc->periodic ? rc->per_begin : 0, rc->periodic ? rc->per_len : 0, rc->periodic ? 1 + extra : 0, 0,
                         rc->periodic ? rc->depth / 2 : 0, lb_hist ? n - 1 : 0, rc->periodic ? rc->ring : 0};
  hipEvent_t a = nullptr, b = nullptr;
  if (e->timing) { a = e->get_event(); b = e->get_event(); HIP_CHECK(hipEventRecord(a, e->stream)); }
  // (plain table loads, not the streaming policy: the second reader of a table is meant to find it in the Infinity Cache)
  if (!launch_chain(c.kclass, 0, &ca, c.launches, e->d_dual, e->d_const, e->d_tabs, e->d_lb, nullptr, e->stream)) throw DeviceError("chain executor: no kernel for class " + std::to_string(c.kclass));
  if (!rc->periodic) tr.end(c, e->stream, e->d_chain_abort);
  if (e->timing) {
    HIP_CHECK(hipEventRecord(b, e->stream));
    e->pending.push_back({a, b, c.kclass, rc->factors + extra * rc->per_factors, rc->recv + extra * rc->per_recv, rc->bytes + extra * rc->per_bytes});
    e->ct[c.kclass].chain_launches++;
  }
  HIP_CHECK(hipGetLastError());
  e->chain_ran = true;
  return true;
}

void require_model(const lpmp_engine* e) { if (!e || !e->plan) throw StateError("no model uploaded"); }
void require_mode(const lpmp_engine* e) {
  require_model(e);
  if (e->mode < 0) throw StateError("no reparametrization mode set");   // reference LP_MP.h:414,458
}

}  // namespace

static void settle(lpmp_engine* e);          // speculative passes: make the device state the caller's state (below)
// rows layout: bring the side that is about to be read up to date (stream-ordered copies of the message vectors)
static void rows_refresh(lpmp_engine* e) {   // packed duals -> rows, before anything computes on the rows
  // (every hand-over first writes the rows out — rows_flush — and only then lets the caller write: both copies newer than the
  // other would mean one of the two gets lost)
  if (e->rows && e->rows_stale && e->packed_stale) throw StateError("rows layout: packed duals and rows both hold newer vectors");
  if (e->rows && e->rows_stale) { launch_rows_copy(e->d_rowrecs, e->n_rowrecs, e->d_const, e->d_dual, e->d_rows, 1, e->stream); HIP_CHECK(hipGetLastError()); e->rows_stale = false; }
}
static void rows_flush(lpmp_engine* e) {     // rows -> packed duals, before the packed array is handed to anybody
  if (e->rows && e->packed_stale) { launch_rows_copy(e->d_rowrecs, e->n_rowrecs, e->d_const, e->d_dual, e->d_rows, 2, e->stream); HIP_CHECK(hipGetLastError()); e->packed_stale = false; }
}
static void begin_compute(lpmp_engine* e) { rows_refresh(e); if (e->rows) e->packed_stale = true; }

extern "C" {

// ---- plan -----------------------------------------------------------------------------------------
int lpmp_plan_create(const lpmp_model* m, lpmp_plan** out) {
  return guarded([&] {
    if (!m || !out) throw std::runtime_error("null argument");
    auto p = std::make_unique<lpmp_plan>();
    p->p.build(*m);
    *out = p.release();
  });
}
void lpmp_plan_destroy(lpmp_plan* p) { delete p; }
int64_t lpmp_plan_n_factors(const lpmp_plan* p) { return p ? p->p.nf : 0; }
int64_t lpmp_plan_n_updated(const lpmp_plan* p, int d) { return p && (d == 0 || d == 1) ? (int64_t)p->p.upd[d].size() : 0; }
int lpmp_plan_get_order(const lpmp_plan* p, int d, int32_t* out) {
  return guarded([&] {
    if (!p || !out || d < 0 || d > 1) throw std::runtime_error("bad argument");
    std::memcpy(out, p->p.order[d].data(), p->p.order[d].size() * sizeof(int32_t));
  });
}
int lpmp_plan_get_update_order(const lpmp_plan* p, int d, int32_t* out) {
  return guarded([&] {
    if (!p || !out || d < 0 || d > 1) throw std::runtime_error("bad argument");
    std::memcpy(out, p->p.upd[d].data(), p->p.upd[d].size() * sizeof(int32_t));
  });
}
int64_t lpmp_plan_omega_nnz(lpmp_plan* p, int d) {
  int64_t s = 0;
  if (p && (d == 0 || d == 1)) for (int32_t f : p->p.upd[d]) s += p->p.row_sends(f);
  return s;
}
int64_t lpmp_plan_mask_nnz(lpmp_plan* p, int d) {
  int64_t s = 0;
  if (p && (d == 0 || d == 1)) for (int32_t f : p->p.upd[d]) s += p->p.row_receives(f);
  return s;
}
int lpmp_plan_get_omega(lpmp_plan* p, int d, int mode, int64_t* off, double* data) {
  return guarded([&] {
    if (!p || !off || d < 0 || d > 1) throw std::runtime_error("bad argument");
    p->p.ensure_weights(mode);
    const auto& c = p->p.omega[d][mode];
    std::memcpy(off, c.off.data(), c.off.size() * sizeof(int64_t));
    if (!c.data.empty()) std::memcpy(data, c.data.data(), c.data.size() * sizeof(double));
  });
}
int lpmp_plan_get_mask(lpmp_plan* p, int d, int mode, int64_t* off, uint8_t* data) {
  return guarded([&] {
    if (!p || !off || d < 0 || d > 1) throw std::runtime_error("bad argument");
    p->p.ensure_weights(mode);
    const auto& c = p->p.mask[d][mode];
    std::memcpy(off, c.off.data(), c.off.size() * sizeof(int64_t));
    if (!c.data.empty()) std::memcpy(data, c.data.data(), c.data.size());
  });
}
int lpmp_plan_get_msg_lists(const lpmp_plan* p, int64_t* off, int64_t* entries) {
  return guarded([&] {
    if (!p || !off || !entries) throw std::runtime_error("bad argument");
    std::memcpy(off, p->p.fm_off.data(), p->p.fm_off.size() * sizeof(int64_t));
    for (size_t i = 0; i < p->p.fm.size(); ++i) entries[i] = (int64_t)p->p.fm[i].msg * 2 + p->p.fm[i].role;
  });
}
int lpmp_plan_anisotropic_weights(const lpmp_plan* p, int64_t n, const int32_t* factors, int64_t* n_rows, int64_t* om_nnz,
                                  int64_t* mk_nnz, int64_t* om_off, double* om, int64_t* mk_off, uint8_t* mk) {
  return guarded([&] {
    if (!p || !factors || n < 0) throw std::runtime_error("bad argument");
    Csr<double> a; Csr<uint8_t> b;
    p->p.anisotropic_weights(factors, n, a, b);
    if (n_rows) *n_rows = a.rows();
    if (om_nnz) *om_nnz = (int64_t)a.data.size();
    if (mk_nnz) *mk_nnz = (int64_t)b.data.size();
    if (om_off && om && mk_off && mk) {
      std::memcpy(om_off, a.off.data(), a.off.size() * sizeof(int64_t));
      std::memcpy(mk_off, b.off.data(), b.off.size() * sizeof(int64_t));
      if (!a.data.empty()) std::memcpy(om, a.data.data(), a.data.size() * sizeof(double));
      if (!b.data.empty()) std::memcpy(mk, b.data.data(), b.data.size());
    }
  });
}
int lpmp_plan_schedule_info(lpmp_plan* p, int d, int mode, int64_t* n_levels, int64_t* n_launches, int64_t* n_recv,
                            int64_t* n_send, int64_t* alg_bytes) {
  return guarded([&] {
    if (!p || d < 0 || d > 1 || mode < 0 || mode >= LPMP_REPAM_COUNT) throw std::runtime_error("bad argument");
    plan_schedule(p, d, mode);
    const Schedule& s = p->sched_cache[d][mode];
    if (n_levels) *n_levels = s.n_levels;
    if (n_launches) *n_launches = (int64_t)s.launches.size();
    if (n_recv) *n_recv = s.n_recv;
    if (n_send) *n_send = s.n_send;
    if (alg_bytes) *alg_bytes = s.alg_bytes;
  });
}

int lpmp_plan_schedule_classes(lpmp_plan* p, int d, int mode, int64_t* factors) {
  static_assert(LPMP_KCLASS_COUNT == KC_COUNT, "public class count");
  return guarded([&] {
    if (!p || !factors || d < 0 || d > 1 || mode < 0 || mode >= LPMP_REPAM_COUNT) throw std::runtime_error("bad argument");
    plan_schedule(p, d, mode);
    for (int c = 0; c < KC_COUNT; ++c) factors[c] = 0;
    for (const auto& lr : p->sched_cache[d][mode].launches) factors[lr.kclass] += lr.end - lr.begin;
  });
}

int lpmp_plan_custom_schedule_info(lpmp_plan* p, int64_t n, const int32_t* factors, const int64_t* om_off, const double* om,
                                   const int64_t* mk_off, const uint8_t* mk, int fuse, int64_t* n_levels, int64_t* n_launches,
                                   int64_t* n_recv, int64_t* n_send, int64_t* alg_bytes) {
  return guarded([&] {
    if (!p || n < 0 || (n > 0 && (!factors || !om_off || !mk_off))) throw std::runtime_error("bad argument");
    check_rows(n, om_off, om, mk_off, mk);
    Schedule s;
    static const double dz = 0; static const uint8_t uz = 0;
    static const int64_t zero_off[1] = {0};
    p->p.make_schedule(std::vector<Plan::Segment>{Plan::Segment{factors, n, n > 0 ? om_off : zero_off, om ? om : &dz,
                                                                n > 0 ? mk_off : zero_off, mk ? mk : &uz}}, fuse != 0, s);
    if (n_levels) *n_levels = s.n_levels;
    if (n_launches) *n_launches = (int64_t)s.launches.size();
    if (n_recv) *n_recv = s.n_recv;
    if (n_send) *n_send = s.n_send;
    if (alg_bytes) *alg_bytes = s.alg_bytes;
  });
}

int lpmp_plan_get_update_levels(lpmp_plan* p, int d, int mode, int32_t* out) {
  return guarded([&] {
    if (!p || !out || d < 0 || d > 1 || mode < 0 || mode >= LPMP_REPAM_COUNT) throw std::runtime_error("bad argument");
    if (!p->have_sched[d][mode]) {
      // the sweep has not been planned (a multi-GPU host asking for the GLOBAL level structure never runs it): the levels alone
      p->p.ensure_weights(mode);
      const auto& om = p->p.omega[d][mode];
      const auto& mk = p->p.mask[d][mode];
      std::vector<int32_t> lv;
      Schedule scratch;
      p->p.make_schedule(std::vector<Plan::Segment>{{p->p.upd[d].data(), (int64_t)p->p.upd[d].size(), om.off.data(), om.data.data(), mk.off.data(), mk.data.data()}},
                         false, scratch, false, &lv);
      std::copy(lv.begin(), lv.end(), out);
      return;
    }
    const Schedule& s = p->sched_cache[d][mode];
    std::vector<int32_t> level_of(p->p.nf, 0);
    for (const auto& lr : s.launches) for (int64_t i = lr.begin; i < lr.end; ++i) level_of[s.recs[i].factor] = lr.level;
    const auto& upd = p->p.upd[d];
    for (size_t i = 0; i < upd.size(); ++i) out[i] = level_of[upd[i]];
  });
}

int lpmp_plan_suggest_order(lpmp_plan* p, uint64_t seed, int32_t* rank_of_factor, int32_t* n_colours) {
  return guarded([&] {
    if (!p || (!rank_of_factor && p->p.nf > 0)) throw std::runtime_error("null argument");
    const int32_t k = suggest_order(p->p, seed, rank_of_factor);
    if (n_colours) *n_colours = k;
  });
}
int lpmp_plan_pass_schedule_info(lpmp_plan* p, int mode, int64_t* n_levels, int64_t* n_launches, int64_t* n_recv,
                                 int64_t* n_send, int64_t* alg_bytes) {
  return guarded([&] {
    if (!p || mode < 0 || mode >= LPMP_REPAM_COUNT) throw std::runtime_error("bad argument");
    plan_pass_schedule(p, mode);
    const Schedule& s = p->pass_cache[mode];
    if (n_levels) *n_levels = s.n_levels;
    if (n_launches) *n_launches = (int64_t)s.launches.size();
    if (n_recv) *n_recv = s.n_recv;
    if (n_send) *n_send = s.n_send;
    if (alg_bytes) *alg_bytes = s.alg_bytes;
  });
}

int lpmp_plan_chain_info(lpmp_plan* p, int d, int mode, int64_t* n_chains, int64_t* n_tickets, int64_t* n_dependencies, int64_t* n_plain_launches) {
  return guarded([&] {
    if (!p || mode < 0 || mode >= LPMP_REPAM_COUNT || d < -1 || d > 1) throw std::runtime_error("bad argument");
    const Schedule* s;
    if (d < 0) { plan_pass_schedule(p, mode); s = &p->pass_cache[mode]; } else { plan_schedule(p, d, mode); s = &p->sched_cache[d][mode]; }
    int64_t t = 0, e = 0;
    for (const auto& c : s->chains) { t += (int64_t)c.tk_launch.size(); e += (int64_t)c.dep.size(); }
    if (n_chains) *n_chains = (int64_t)s->chains.size();
    if (n_tickets) *n_tickets = t;
    if (n_dependencies) *n_dependencies = e;
    if (n_plain_launches) *n_plain_launches = (int64_t)s->plain_launches.size();
  });
}

int lpmp_plan_mailbox_info(lpmp_plan* p, int d, int mode, int64_t* n_rows, int64_t* n_receives) {
  return guarded([&] {
    if (!p || mode < 0 || mode >= LPMP_REPAM_COUNT || d < -1 || d > 1) throw std::runtime_error("bad argument");
    const Schedule* s;
    if (d < 0) { plan_pass_schedule(p, mode); s = &p->pass_cache[mode]; } else { plan_schedule(p, d, mode); s = &p->sched_cache[d][mode]; }
    int64_t r = 0, v = 0;
    for (const auto& c : s->chains) { r += c.mailbox_rows; v += c.mailbox_receives; }
    if (n_rows) *n_rows = r;
    if (n_receives) *n_receives = v;
  });
}

int lpmp_plan_get_partitions(lpmp_plan* p, int64_t* n_partitions, int64_t* off, int32_t* factors) {
  return guarded([&] {
    if (!p || !n_partitions) throw std::runtime_error("bad argument");
    p->p.ensure_partition();
    const auto& pt = p->p.part;
    *n_partitions = (int64_t)pt.off.size() - 1;
    if (off && factors) {
      std::memcpy(off, pt.off.data(), pt.off.size() * sizeof(int64_t));
      if (!pt.f.empty()) std::memcpy(factors, pt.f.data(), pt.f.size() * sizeof(int32_t));
    }
  });
}

int lpmp_plan_pass_rotates(lpmp_plan* p, int mode) {
  int r = 0;
  const int rc = guarded([&] {
    if (!p || mode < 0 || mode >= LPMP_REPAM_COUNT) throw std::runtime_error("bad argument");
    if (!p->have_bf[mode]) { plan_pass_schedule(p, mode); if (p->pass_cache[mode].recs.empty() && !p->pass_cache[mode].launches.empty()) throw StateError("pass schedule already handed to the device"); plan_rotation(p, mode); }
    r = p->rotation_ok[mode] ? 1 : 0;
  });
  return rc == LPMP_OK ? r : rc;
}

// ---- engine ---------------------------------------------------------------------------------------
int lpmp_create(int device, lpmp_engine** out) {
  return guarded([&] {
    if (!out) throw std::runtime_error("null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw DeviceError("no HIP device available: the engine has no CPU path");
    if (device < 0 || device >= n) throw DeviceError("device ordinal out of range");
    HIP_CHECK(hipSetDevice(device));
    auto e = std::make_unique<lpmp_engine>();
    e->device = device;
    e->stream = stream_pool().take(device);
    e->own_stream = true;
    const char* ng = std::getenv("LPMP_NO_GRAPH");
    e->use_graph = !(ng && ng[0] == '1');
    const char* nf = std::getenv("LPMP_NO_FUSE");
    e->use_fused = !(nf && nf[0] == '1');
    const char* nr = std::getenv("LPMP_NO_ROTATION");
    e->use_rotation = !(nr && nr[0] == '1');
    const char* nt = std::getenv("LPMP_NO_LB_TRACKING");
    e->use_lb_tracking = !(nt && nt[0] == '1');
    const char* np = std::getenv("LPMP_NO_PACKED");
    e->use_packed = !(np && np[0] == '1');
    const char* nc = std::getenv("LPMP_NO_CHAIN");
    e->use_chain = !(nc && nc[0] == '1');
    const char* nb = std::getenv("LPMP_NO_BLOCKED_PASSES");
    e->use_blocked_passes = !(nb && nb[0] == '1');
    if (const char* v = std::getenv("LPMP_ROT_BANDS")) e->rot_bands = std::atoi(v);
    if (const char* v = std::getenv("LPMP_ROT_LAG")) { e->rot_lag = std::max(1, std::atoi(v)); e->rot_lag_set = true; }
    if (const char* v = std::getenv("LPMP_ROT_DEPTH")) { e->rot_depth = std::max(1, std::atoi(v)); e->rot_depth_set = true; }
    if (const char* v = std::getenv("LPMP_ROT_TILES")) { e->rot_tiles = std::max(0, std::atoi(v)); e->rot_tiles_set = true; }
    if (const char* v = std::getenv("LPMP_CHAIN_CACHE_MB")) e->rot_cache_limit = (size_t)std::max(1, std::atoi(v)) << 20;
    if (const char* v = std::getenv("LPMP_ROWS_LAYOUT")) e->want_rows = std::atoi(v) != 0;                             // as lpmp_set_rows_layout
    if (const char* v = std::getenv("LPMP_SPECULATION")) e->spec.max_depth = std::min(32, std::max(0, std::atoi(v)));   // as lpmp_set_speculation
    *out = e.release();
  });
}

void lpmp_destroy(lpmp_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  // a BORROWED dual buffer (LPMP_MEM_DEVICE) outlives the engine: an open batch of passes that ran ahead of the caller must
  // not stay in it
  if (e->plan && !e->own_dual && e->spec.n > 0) { try { settle(e); } catch (const std::exception& ex) { std::fprintf(stderr, "lpmp_destroy: could not roll back passes that ran ahead: %s\n", ex.what()); } }
  (void)hipStreamSynchronize(e->stream);
  for (auto& p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto ev : e->event_pool) (void)hipEventDestroy(ev);
  e->release_model();
  if (e->d_chain_abort) { (void)hipFree(e->d_chain_abort); e->d_chain_abort = nullptr; }
  if (e->own_stream && e->stream) stream_pool().give(e->device, e->stream);
  if (e->capture_stream) { (void)hipStreamSynchronize(e->capture_stream); stream_pool().give(e->device, e->capture_stream); }
  if (!guarded_ok(e->pinned, PINNED_WORDS_BYTES)) {   // a damaged block is reported and never reused
    std::fprintf(stderr, "lpmp_destroy: guard region of the engine's pinned words was overwritten\n");
    g_error = "guard region of the engine's pinned words was overwritten";
  } else pinned_pool().give(e->pinned);
  delete e;
}

int lpmp_set_stream(lpmp_engine* e, void* s) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    settle(e);
    HIP_CHECK(hipStreamSynchronize(e->stream));
    for (int d = 0; d < 2; ++d) for (int m = 0; m < LPMP_REPAM_COUNT; ++m)
      if (e->sched[d][m].graph) { (void)hipGraphExecDestroy(e->sched[d][m].graph); e->sched[d][m].graph = nullptr; }
    for (int m = 0; m < LPMP_REPAM_COUNT; ++m)
      if (e->sched_pass[m].graph) { (void)hipGraphExecDestroy(e->sched_pass[m].graph); e->sched_pass[m].graph = nullptr; }
    if (e->own_stream && e->stream) stream_pool().give(e->device, e->stream);
    e->stream = (hipStream_t)s; e->own_stream = false;
  });
}

static void check_rtype(const lpmp_engine* e, int rtype);
int lpmp_upload_model(lpmp_engine* e, const lpmp_model* m, int const_mem, int dual_mem) {
  return guarded([&] {
    if (!e || !m) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    if (e->plan && !e->own_dual && e->spec.n > 0) settle(e);   // the caller keeps the old model's (borrowed) dual buffer: leave it at the caller's pass
    HIP_CHECK(hipStreamSynchronize(e->stream));
    { const int d = e->spec.max_depth; e->release_model(); e->spec.max_depth = d; }   // (an open speculative batch in an engine-owned buffer dies with the old model)
    auto pl = std::make_unique<lpmp_plan>();
    pl->p.build(*m);
    const Plan& p = pl->p;
    const int64_t n_const = p.f_coff[p.nf], n_dual = p.f_doff[p.nf];
    if ((n_const > 0 && !m->const_data) || !m->dual_data) throw std::runtime_error("cost arrays missing");
    // streamed-once access policy (kernels.hip, ld_stream): only when the state cannot live in the 256 MiB Infinity Cache
    {
      const char* env = getenv("LPMP_NT");
      const bool big = (n_const + n_dual) * (int64_t)sizeof(double) > (int64_t)1 << 30;
      e->nt_flag = (env ? atoi(env) != 0 : big) ? SWEEP_NT : 0;
      e->model_big = big;
    }
    if (const_mem == LPMP_MEM_DEVICE) {
      e->d_const = const_cast<double*>(m->const_data);
      if (((uintptr_t)e->d_const & 15) != 0) throw std::runtime_error("device const buffer must be 16-byte aligned");
    } else if (n_const > 0) {
      HIP_CHECK(hipMalloc((void**)&e->d_const, (size_t)n_const * sizeof(double)));
      e->own_const = true;
      h2d(e->d_const, m->const_data, (size_t)n_const * sizeof(double), e->stream);
    }
    if (dual_mem == LPMP_MEM_DEVICE) {
      e->d_dual = const_cast<double*>(m->dual_data);
    } else {
      HIP_CHECK(hipMalloc((void**)&e->d_dual, (size_t)n_dual * sizeof(double)));
      e->own_dual = true;
      h2d(e->d_dual, m->dual_data, (size_t)n_dual * sizeof(double), e->stream);
    }
    if (e->want_rows && e->d_const) {
      // rows layout: every dense pairwise factor becomes one row [table | m1 | m2] of a private buffer; its device offsets
      // (relative to the const / dual base pointers, which the kernels add them to) point there from now on
      std::vector<RowRecHost> rr;
      int64_t at = 0;
      for (int64_t f = 0; f < p.nf; ++f)
        if (p.f_kind[f] == LPMP_F_PAIRWISE_DENSE) {
          rr.push_back({p.f_doff[f], p.f_coff[f], at, p.f_dim0[f], p.f_dim1[f]});
          at += ((int64_t)p.f_dim0[f] * p.f_dim1[f] + p.f_dim0[f] + p.f_dim1[f] + 1) / 2 * 2;      // rows start 16-byte aligned
        }
      if (!rr.empty()) {
        HIP_CHECK(hipMalloc((void**)&e->d_rows, (size_t)at * sizeof(double)));
        if ((((uintptr_t)e->d_rows - (uintptr_t)e->d_const) % 16) != 0 || (((uintptr_t)e->d_rows - (uintptr_t)e->d_dual) % 8) != 0)
          throw std::runtime_error("rows layout: buffers are not aligned to each other");
        HIP_CHECK(hipMalloc((void**)&e->d_rowrecs, rr.size() * sizeof(RowRecHost)));
        h2d(e->d_rowrecs, rr.data(), rr.size() * sizeof(RowRecHost), e->stream);
        e->n_rowrecs = (int64_t)rr.size();
        launch_rows_copy(e->d_rowrecs, e->n_rowrecs, e->d_const, e->d_dual, e->d_rows, 0, e->stream);
        HIP_CHECK(hipGetLastError());
        const int64_t c_shift = (int64_t)(((intptr_t)e->d_rows - (intptr_t)e->d_const) / 8), d_shift = (int64_t)(((intptr_t)e->d_rows - (intptr_t)e->d_dual) / 8);
        pl->p.dev_coff.assign(p.f_coff.begin(), p.f_coff.end()); pl->p.dev_doff.assign(p.f_doff.begin(), p.f_doff.end());
        size_t k = 0;
        for (int64_t f = 0; f < p.nf; ++f)
          if (p.f_kind[f] == LPMP_F_PAIRWISE_DENSE) {
            pl->p.dev_coff[f] = c_shift + rr[k].row_off;
            pl->p.dev_doff[f] = d_shift + rr[k].row_off + (int64_t)p.f_dim0[f] * p.f_dim1[f];
            ++k;
          }
        e->rows = true; e->packed_stale = false; e->rows_stale = false;
      }
    }
    if (!p.tab_data.empty()) {
      HIP_CHECK(hipMalloc((void**)&e->d_tabs, p.tab_data.size() * sizeof(int32_t)));
      h2d(e->d_tabs, p.tab_data.data(), p.tab_data.size() * sizeof(int32_t), e->stream);
    }
    // lower-bound records, in factor order, and runs of factors the streaming dense kernel can take
    std::vector<LbRecHost> lb(p.nf);
    auto lb_class = [&](int64_t f) {
      if (p.f_kind[f] == LPMP_F_PAIRWISE_DENSE && p.f_dim0[f] == p.f_dim1[f] && (p.coff(f) % 2) == 0 &&
          (p.f_dim0[f] == 8 || p.f_dim0[f] == 16 || p.f_dim0[f] == 32)) return p.f_dim0[f];
      return 0;
    };
    for (int64_t f = 0; f < p.nf; ++f) {
      lb[f] = {p.doff(f), p.f_kind[f] == LPMP_F_VECTOR ? -1 : p.coff(f), p.f_dim0[f], p.f_dim1[f], p.f_kind[f] | (p.f_flags[f] << 4), 0};
      const int c = lb_class(f);
      if (e->lb_runs.empty() || e->lb_runs.back().cls != c) e->lb_runs.push_back({c, f, 1}); else e->lb_runs.back().count++;
    }
    HIP_CHECK(hipMalloc((void**)&e->d_lbrecs, (size_t)p.nf * sizeof(LbRecHost)));
    h2d(e->d_lbrecs, lb.data(), (size_t)p.nf * sizeof(LbRecHost), e->stream);
    HIP_CHECK(hipMalloc((void**)&e->d_lb, (size_t)p.nf * sizeof(double)));
    HIP_CHECK(hipMalloc((void**)&e->d_part, 1024 * sizeof(double)));
    if (!e->pinned) e->pinned = pinned_pool().take();
    e->h_part = (double*)e->pinned;                            // [0, 1024) doubles: partial sums
    HIP_CHECK(hipMalloc((void**)&e->d_stale, (size_t)p.nf * sizeof(int32_t)));
    HIP_CHECK(hipMalloc((void**)&e->d_stale_n, sizeof(unsigned long long)));
    e->h_stale_n = (unsigned long long*)(e->pinned + 8 * 1024);   // one counter
    HIP_CHECK(hipMemsetAsync(e->d_lb, 0xFF, (size_t)p.nf * sizeof(double), e->stream));   // all NaN: nothing tracked yet
    HIP_CHECK(hipStreamSynchronize(e->stream));
    e->lb_all_stale = true;
    e->plan = std::move(pl);
    e->plan->p.force_generic = e->rtype == LPMP_RTYPE_ADAPTIVE;
    {   // mailbox budget of every schedule planned for this model: half of what the device has left now (LPMP_MAILBOX_MB overrides)
      size_t free_b = 0, total_b = 0;
      if (const char* v = std::getenv("LPMP_MAILBOX_MB")) e->plan->p.mailbox_budget_bytes = (int64_t)std::atoll(v) << 20;
      else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) e->plan->p.mailbox_budget_bytes = (int64_t)(free_b / 2);
      else (void)hipGetLastError();
    }
    try { check_rtype(e, e->rtype); } catch (...) { e->release_model(); throw; }
  });
}

int lpmp_set_reparametrization(lpmp_engine* e, int mode) {
  return guarded([&] {
    require_model(e);
    if (mode == LPMP_REPAM_MIXED) throw UnsupportedError("mixed reparametrization is assert(false) in the reference (LP_MP.h:1455)");
    if (mode < 0 || mode >= LPMP_REPAM_COUNT) throw std::runtime_error("unknown reparametrization mode");
    HIP_CHECK(hipSetDevice(e->device));
    if (mode != e->mode) settle(e);          // (Solver::PreIterate sets the same mode before every pass, solver.hxx:268-271)
    // The directional schedules (ComputeForwardPass / ComputeBackwardPass, the ...AndPrimal sweeps) are built here for models of
    // ordinary size — a model the device kernels cannot run is refused at this call — and on first use for models of millions of
    // factors: LP::ComputePass runs the fused pass schedule and the partitioned drivers their own iterator-range schedules, and
    // two directional schedules nobody runs were 2.6 s of the 10 s before the first pass of the 2 M / 10 M graph
    if (e->plan->p.nf <= LAZY_SCHEDULES_MIN_FACTORS) ensure_device_schedules(e, mode);
    else e->plan->p.ensure_weights(mode);
    e->mode = mode;
  });
}

// what keeps the uploaded model from running under a send rule (also checked at upload: the rule may be set first,
// as the reference parses --reparametrizationType in LP::Begin)
static void check_rtype(const lpmp_engine* e, int rtype) {
  if (!e->plan) return;
  const Plan& p = e->plan->p;
  if (rtype == LPMP_RTYPE_RESIDUAL && p.any_batch)
    throw UnsupportedError("residual sends with batch-capable message ops are not built (send_messages_residual's batch branch, factors_messages.hxx:2980-2991)");
  if (rtype == LPMP_RTYPE_ADAPTIVE) { const std::string why = p.adaptive_obstacle(); if (!why.empty()) throw UnsupportedError(why); }
}
static void apply_rtype(lpmp_engine* e, int rtype) {
  const bool generic = rtype == LPMP_RTYPE_ADAPTIVE;
  if (e->plan && e->plan->p.force_generic != generic) {   // other kernel classes: every built-in schedule is rebuilt
    e->release_schedules();
    e->plan->drop_caches();
    e->plan->p.force_generic = generic;
    e->mode = -1;
  }
  e->rtype = rtype;
}

int lpmp_set_reparametrization_type(lpmp_engine* e, int rtype) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    if (rtype < LPMP_RTYPE_SHARED || rtype > LPMP_RTYPE_ADAPTIVE) throw std::runtime_error("unknown reparametrization type");
    check_rtype(e, rtype);
    if (rtype != e->rtype) settle(e);
    if (rtype != e->rtype) {   // captured graphs bake the kernel flag in
      HIP_CHECK(hipStreamSynchronize(e->stream));
      for (int d = 0; d < 2; ++d) for (int m = 0; m < LPMP_REPAM_COUNT; ++m)
        if (e->sched[d][m].graph) { (void)hipGraphExecDestroy(e->sched[d][m].graph); e->sched[d][m].graph = nullptr; }
      for (int m = 0; m < LPMP_REPAM_COUNT; ++m)
        if (e->sched_pass[m].graph) { (void)hipGraphExecDestroy(e->sched_pass[m].graph); e->sched_pass[m].graph = nullptr; }
      for (int k = 0; k < 2; ++k) if (e->sched_part[k].graph) { (void)hipGraphExecDestroy(e->sched_part[k].graph); e->sched_part[k].graph = nullptr; }
      for (auto& c : e->custom) if (c && c->graph) { (void)hipGraphExecDestroy(c->graph); c->graph = nullptr; }
    }
    const int mode = e->mode;
    apply_rtype(e, rtype);
    if (e->mode < 0 && mode >= 0) { ensure_device_schedules(e, mode); e->mode = mode; }   // the mode survives the rebuild
  });
}

int lpmp_set_inner_iterations(lpmp_engine* e, int n) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    if (n < 1) throw std::runtime_error("innerIteration must be positive");
    if (n != e->inner_iterations) {
      settle(e);
      HIP_CHECK(hipStreamSynchronize(e->stream));
      for (int k = 0; k < 2; ++k) { e->sched_part[k].release(); e->have_part[k] = false; }
    }
    e->inner_iterations = n;
  });
}

// compute_partition_pass / compute_overlapping_partition_pass (reference LP_MP.h:1932-2051): a fixed sequence of
// iterator-range passes over the partitions' factor lists with their own anisotropic weights.  The whole sequence is
// level-scheduled as ONE schedule: inner iterations of partitions that touch no common factor run side by side.
static void ensure_partition_schedule(lpmp_engine* e, int rtype) {
  const int k = rtype - LPMP_RTYPE_PARTITION;
  if (e->have_part[k]) return;
  std::vector<Plan::Segment> segs;
  e->plan->p.partition_pass_segments(rtype, e->inner_iterations, segs);
  Schedule s;
  e->plan->p.make_schedule(segs, e->use_fused, s);
  check_generic_limits(e->plan->p, s);
  upload_schedule(s, e->sched_part[k], e->stream);
  e->have_part[k] = true;
}

int lpmp_compute_forward_pass(lpmp_engine* e) {
  return guarded([&] { require_mode(e); HIP_CHECK(hipSetDevice(e->device)); settle(e); ensure_device_schedules(e, e->mode); begin_compute(e); run_schedule(e, e->sched[0][e->mode]); });
}
int lpmp_compute_backward_pass(lpmp_engine* e) {
  return guarded([&] { require_mode(e); HIP_CHECK(hipSetDevice(e->device)); settle(e); ensure_device_schedules(e, e->mode); begin_compute(e); run_schedule(e, e->sched[1][e->mode]); });
}
static void compute_plain_passes(lpmp_engine* e, int n) {   // ComputeForwardPass(); ComputeBackwardPass(); n times
  if (e->use_fused) {
    ensure_pass_schedule(e, e->mode);
    if (e->rotation_ok[e->mode] && e->use_rotation) {
      // passes in slices of at most 32: each slice one persistent launch (memory of the ticket arrays stays bounded)
      int done = 0;
      while (done < n) {
        const int m = std::min(n - done, 32);
        if (!run_rotation_chain(e, e->mode, m)) break;
        done += m;
      }
      if (done == n) return;
      n -= done;
    }
    if (n >= 2 && e->rotation_ok[e->mode] && e->use_rotation) {
      const DevSchedule& fb = e->sched_pass[e->mode];
      const DevSchedule& bf = e->sched_bf[e->mode];
      const bool timed = e->timing;
      issue_launches(e, fb, timed, e->stream, 1);
      issue_launches(e, fb, timed, e->stream, 2);
      for (int i = 1; i < n; ++i) { issue_launches(e, bf, timed, e->stream, 2); issue_launches(e, fb, timed, e->stream, 2); }
      issue_launches(e, fb, timed, e->stream, 3);
      if (timed && e->pending.size() > 4096) e->drain_timing();
    } else {
      ensure_pass_chain_plan(e, e->mode);
      for (int i = 0; i < n; ++i) run_schedule(e, e->sched_pass[e->mode]);
    }
  } else {
    ensure_device_schedules(e, e->mode);
    for (int i = 0; i < n; ++i) { run_schedule(e, e->sched[0][e->mode]); run_schedule(e, e->sched[1][e->mode]); }
  }
}
// build whatever lpmp_compute_pass(e, n) needs that depends on n (the ticket order of n joined passes), outside of a
// timed region; optional
int lpmp_prepare_passes(lpmp_engine* e, int n) {
  return guarded([&] {
    require_mode(e);
    if (n < 1) throw std::runtime_error("bad argument");
    HIP_CHECK(hipSetDevice(e->device));
    if (e->rtype != LPMP_RTYPE_SHARED || !e->use_fused) return;
    ensure_pass_schedule(e, e->mode);
    if (!(e->rotation_ok[e->mode] && e->use_rotation && e->use_chain && e->use_blocked_passes)) return;
    for (int done = 0; done < n;) { const int m = std::min(n - done, 32); (void)rotation_chain(e, e->mode, m); done += m; }
  });
}
// ---- speculative passes ---------------------------------------------------------------------------------------------
// The reference's caller asks for ONE pass per iteration and, by default, for the bound after every pass
// (Solver::Iterate / PostIterate, solver.hxx:273-284), while the device is fastest when consecutive passes are ONE
// persistent launch in Infinity-Cache order (rotation_chain: 5.1 against 6.6 ms per pass on C3).  With speculation on,
// lpmp_compute_pass(e, 1) launches a BATCH of n passes ahead of the caller — after a snapshot of the duals — and the
// launch itself leaves the tracked bounds of all factors as they are at the end of every pass (one row per pass,
// kernels.hip HIST_END / HIST_MID).  The following n - 1 calls of lpmp_compute_pass(e, 1) only advance a cursor, and
// lpmp_lower_bound returns the bound of the pass the caller is at (the sum of that row, in the order lpmp_lower_bound sums).
// ANY other call settles first: if the caller stopped inside the batch the duals go back to the snapshot and exactly the
// passes it asked for are run again (bit-identical: n joined passes equal n single ones).  So the speculation
// is invisible except in time; how far ahead is learnt from the caller: batches double while single passes keep coming
// and restart at the length of the previous run (MpRoundingSolver: four plain passes between two rounding iterations).
static void spec_interrupt(lpmp_engine* e) {
  auto& sp = e->spec;
  if (sp.run_len > 0) sp.learned = sp.run_len;
  sp.run_len = 0; sp.last_batch = 0;
}
static void settle(lpmp_engine* e) {
  if (!e) return;
  auto& sp = e->spec;
  if (sp.n > 0) {
    const int n = sp.n, pos = sp.pos;
    sp.n = 0; sp.pos = 0; sp.lb_ready = false;
    if (pos < n && e->plan) {   // the caller stopped inside the batch: back to its start, then exactly the passes it asked for
      HIP_CHECK(hipSetDevice(e->device));
      const size_t nd = (size_t)e->plan->p.f_doff[e->plan->p.nf], nf = (size_t)e->plan->p.nf;
      HIP_CHECK(hipMemcpyAsync(e->d_dual, sp.d_snap, nd * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      HIP_CHECK(hipMemcpyAsync(e->d_lb, sp.d_snap + nd, nf * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      e->lb_all_stale = sp.snap_lb_stale;
      ++sp.rollbacks;
      const int mode = e->mode;
      e->mode = sp.mode;          // (set_reparametrization settles before it changes the mode)
      try { compute_plain_passes(e, pos); } catch (...) { e->mode = mode; throw; }
      e->mode = mode;
    }
  }
  spec_interrupt(e);
}
static void lb_sum_blocks(int64_t nf, int64_t& nb, int64_t& per) {   // the partition lpmp_lower_bound sums in
  nb = std::min<int64_t>(1024, (nf + 255) / 256);
  per = (nf + nb - 1) / nb;
  nb = (nf + per - 1) / per;
}
static bool spec_start_batch(lpmp_engine* e, int depth) {
  auto& sp = e->spec;
  ensure_pass_schedule(e, e->mode);
  // (the bound rows are written by the dense chain body only: kernels.hip, dense_pk_body)
  if (!(e->rotation_ok[e->mode] && e->plan->rot[e->mode].valid && e->plan->rot[e->mode].hist_ok && kc_is_dense(e->plan->rot[e->mode].kclass))) return false;
  depth = std::min(depth, 32);
  if (!rotation_chain(e, e->mode, depth)) return false;
  const size_t nd = (size_t)e->plan->p.f_doff[e->plan->p.nf], nf = (size_t)e->plan->p.nf;
  // (an allocation that fails switches speculation OFF for this engine — the plain pass needs none of these buffers — instead
  // of making every lpmp_compute_pass(e, 1) fail on a model that solves fine without: snapshot = all duals + tracked bounds)
  auto grow = [&](double*& p, size_t& cap, size_t want) -> bool {
    if (want <= cap) return true;
    HIP_CHECK(hipStreamSynchronize(e->stream));
    if (p) { HIP_CHECK(hipFree(p)); p = nullptr; cap = 0; }
    if (hipMalloc((void**)&p, want * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    cap = want;
    return true;
  };
  if (!grow(sp.d_snap, sp.snap_cap, nd + nf) || !grow(sp.d_hist, sp.hist_cap, (size_t)(std::min(sp.max_depth, 32) - 1) * nf) ||
      !grow(sp.d_hpart, sp.hpart_cap, (size_t)(std::min(sp.max_depth, 32) - 1) * 1024)) {
    sp.release();
    sp.max_depth = 0;
    ++sp.alloc_failures;
    return false;
  }
  HIP_CHECK(hipMemcpyAsync(sp.d_snap, e->d_dual, nd * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIP_CHECK(hipMemcpyAsync(sp.d_snap + nd, e->d_lb, nf * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  sp.snap_lb_stale = e->lb_all_stale;
  if (!run_rotation_chain(e, e->mode, depth, sp.d_hist)) return false;
  // the device is `depth` passes ahead from here on: the batch is open BEFORE anything else can fail, so that settle() rolls back
  sp.n = depth; sp.pos = 0; sp.mode = e->mode; sp.lb_ready = false; sp.last_batch = depth;
  ++sp.batches; sp.passes_launched += depth;
  int64_t nb, per; lb_sum_blocks((int64_t)nf, nb, per);
  for (int i = 0; i < depth - 1; ++i) launch_sum_stage(sp.d_hist + (size_t)i * nf, sp.d_hpart + (size_t)i * 1024, (int64_t)nf, per, nb, e->stream);
  HIP_CHECK(hipGetLastError());
  return true;
}
static void check_chain(lpmp_engine* e);
// the bound of the pass the caller is at, while that pass lies inside an open batch
static bool spec_lower_bound(lpmp_engine* e, double* out) {
  auto& sp = e->spec;
  if (!(sp.n > 0 && sp.pos < sp.n)) return false;
  if (!sp.lb_ready) {
    check_chain(e);
    const int64_t nf = e->plan->p.nf;
    int64_t nb, per; lb_sum_blocks(nf, nb, per);
    std::vector<double> h((size_t)(sp.n - 1) * 1024);
    d2h(h.data(), sp.d_hpart, h.size() * sizeof(double), e->stream);
    sp.lb.assign((size_t)sp.n - 1, 0.0);
    for (int i = 0; i < sp.n - 1; ++i) {
      double lb = e->plan->p.constant;
      for (int64_t j = 0; j < nb; ++j) lb += h[(size_t)i * 1024 + j];
      sp.lb[i] = lb;
    }
    sp.lb_ready = true;
  }
  *out = sp.lb[sp.pos - 1];
  return true;
}
static bool spec_usable(const lpmp_engine* e) {
  return e->spec.max_depth >= 2 && e->rtype == LPMP_RTYPE_SHARED && e->use_fused && e->use_rotation && e->use_chain &&
         e->use_blocked_passes && e->use_lb_tracking && !e->timing && !e->rows;
}

int lpmp_set_speculation(lpmp_engine* e, int max_passes_ahead) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    if (max_passes_ahead < 0) throw std::runtime_error("bad argument");
    settle(e);
    e->spec.max_depth = std::min(max_passes_ahead, 32);
    if (e->spec.max_depth < 2) { HIP_CHECK(hipStreamSynchronize(e->stream)); const int d = e->spec.max_depth; e->spec.release(); e->spec.max_depth = d; }
  });
}
int lpmp_speculation_stats(lpmp_engine* e, int64_t* batches, int64_t* passes_launched, int64_t* passes_used, int64_t* rollbacks) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    if (batches) *batches = e->spec.batches;
    if (passes_launched) *passes_launched = e->spec.passes_launched;
    if (passes_used) *passes_used = e->spec.passes_used;
    if (rollbacks) *rollbacks = e->spec.rollbacks;
  });
}
int64_t lpmp_chain_cache_bytes(const lpmp_engine* e) { return e ? (int64_t)e->rot_cache_bytes : 0; }

int lpmp_compute_pass(lpmp_engine* e, int n) {   // LP::ComputePass, LP_MP.h:869-887
  return guarded([&] {
    require_mode(e);
    HIP_CHECK(hipSetDevice(e->device));
    begin_compute(e);
    if (n == 1 && e->spec.max_depth >= 2) {
      auto& sp = e->spec;
      if (sp.n > 0) {
        if (sp.pos < sp.n) { ++sp.pos; ++sp.run_len; ++sp.passes_used; return; }   // already on its way
        sp.n = 0; sp.pos = 0; sp.lb_ready = false;                                  // used up: the device state is the caller's
      }
      if (spec_usable(e)) {
        int depth = sp.run_len == 0 ? (sp.learned > 0 ? sp.learned : 2) : std::max(2, 2 * sp.last_batch);
        depth = std::min(depth, sp.max_depth);
        if (depth >= 2 && spec_start_batch(e, depth)) { sp.pos = 1; ++sp.run_len; ++sp.passes_used; return; }
      }
      ++sp.run_len; sp.last_batch = 0;
      if (e->rtype == LPMP_RTYPE_SHARED) { compute_plain_passes(e, 1); return; }
    } else settle(e);
    if (e->rtype == LPMP_RTYPE_PARTITION) {
      ensure_partition_schedule(e, e->rtype);
      for (int i = 0; i < n; ++i) run_schedule(e, e->sched_part[0]);
    } else if (e->rtype == LPMP_RTYPE_OVERLAPPING_PARTITION) {
      ensure_partition_schedule(e, e->rtype);
      for (int i = 0; i < n; ++i) { run_schedule(e, e->sched_part[1]); compute_plain_passes(e, 1); }
    } else {
      compute_plain_passes(e, n);
    }
  });
}

// ---- primal rounding inside the sweep ------------------------------------------------------------------------------
// What the reference does per updated factor (UpdateFactorPrimal) is split in three: the lazy init of everything the
// pass touches (one kernel, only when the time stamp grows — all touched factors carry the same stamp), the label of
// every COMPUTE_PRIMAL unary inside the sweep kernels (state after its receives), and the copy of the labels into the
// pairwise factors afterwards.  The copy can wait because a pairwise factor's primal_[side] has a single writer and
// nothing reads it before EvaluatePrimal: with a `left` schedule the recursion of propagate_primal_through_messages
// stops at the pairwise factor (its other side is unset or already equal).
static void ensure_primal(lpmp_engine* e) {
  if (e->have_primal) return;
  e->release_primal();   // a previous attempt may have stopped half way
  const Plan& p = e->plan->p;
  for (const auto& mt : p.mtypes)
    if (mt.kind != LPMP_M_UNARY_PAIRWISE)
      throw UnsupportedError("primal rounding is built for unary / pairwise models (DESIGN.md 8)");
  // Pairwise factor types with COMPUTE_PRIMAL_SOLUTION (MPLP-style `right` / `full` schedules): the recursion of
  // propagate_primal_through_messages then runs pairwise -> its unaries -> their other pairwise factors and stops
  // (a slot that is set always equals its unary's label, so nothing changes further out).  The device keeps the
  // UNARY labels as the one source of truth inside a sweep: an updated pairwise factor of such a type reads its
  // sides from its unaries' labels, fills the free ones (first minimiser in row-major order given the others) and
  // labels those unaries; the pairwise slots are copies, made after the sweep as before.  Such records run on the
  // generic kernels in primal passes (issue_launches) and depend on all their unaries (plan.cpp, make_schedule).
  bool pw_computes = false;
  for (int64_t f = 0; f < p.nf; ++f) if (p.f_kind[f] != LPMP_F_VECTOR && p.ftype_primal[p.f_type[f]] && p.updated[f]) pw_computes = true;
  std::vector<PrimalLink> prop, rest;
  std::vector<int32_t> writer(2 * (size_t)p.nf, -1);
  std::vector<uint8_t> touched((size_t)p.nf, 0), labelable((size_t)p.nf, 0);
  for (int64_t m = 0; m < p.nm; ++m) {
    const int32_t l = p.m_left[m], r = p.m_right[m];
    if (p.f_kind[l] != LPMP_F_VECTOR || p.f_kind[r] == LPMP_F_VECTOR) throw UnsupportedError("primal rounding: unary-pairwise message between unexpected factor kinds");
    if (p.ftype_primal[p.f_type[l]] || (p.ftype_primal[p.f_type[r]] && p.updated[r])) labelable[l] = 1;
  }
  for (int64_t m = 0; m < p.nm; ++m) {
    const int32_t l = p.m_left[m], r = p.m_right[m];
    const int side = p.mtypes[p.m_type[m]].param;
    const PrimalLink k{l, r, side, p.f_dim0[l]};
    if (labelable[l]) {
      int32_t& w = writer[2 * (size_t)r + side];   // the copy into the pairwise factor is deferred: one writer per slot
      if (w >= 0 && w != l) throw UnsupportedError("primal rounding: two unaries on one side of a pairwise factor");
      w = l;
      prop.push_back(k);
      touched[r] = 1;
    } else rest.push_back(k);
  }
  if (pw_computes) {
    // conditionally_init_primal reaches one step further: a pairwise factor whose slot changes initialises all its unaries
    for (int64_t m = 0; m < p.nm; ++m) if (touched[p.m_right[m]]) touched[p.m_left[m]] = 2;
    std::vector<int32_t> h(2 * (size_t)p.nf);
    for (size_t i = 0; i < h.size(); ++i) h[i] = writer[i];
    HIP_CHECK(hipMalloc((void**)&e->d_pw_unary, std::max<size_t>(1, h.size()) * sizeof(int32_t)));
    if (!h.empty()) h2d(e->d_pw_unary, h.data(), h.size() * sizeof(int32_t), e->stream);
  }
  for (int64_t f = 0; f < p.nf; ++f) if (p.updated[f]) touched[f] = 1;
  auto unset = [&](int64_t f) { return PrimalInit{(int32_t)f, p.f_dim0[f], p.f_kind[f] == LPMP_F_VECTOR ? 0 : p.f_dim1[f], 0}; };
  std::vector<PrimalInit> init, all((size_t)p.nf);
  for (int64_t f = 0; f < p.nf; ++f) { all[f] = unset(f); if (touched[f]) init.push_back(unset(f)); }
  e->n_pprop = (int64_t)prop.size();
  prop.insert(prop.end(), rest.begin(), rest.end());
  e->n_plinks = (int64_t)prop.size();
  e->n_pinit = (int64_t)init.size();
  HIP_CHECK(hipMalloc((void**)&e->d_primal, std::max<size_t>(1, 2 * (size_t)p.nf) * sizeof(int32_t)));
  HIP_CHECK(hipMalloc((void**)&e->d_pcost, std::max<size_t>(1, (size_t)p.nf) * sizeof(double)));
  HIP_CHECK(hipMalloc((void**)&e->d_pbad, sizeof(int)));
  e->h_pbad = (int*)(e->pinned + 8 * 1024 + 64);               // one flag (the block exists: a model is uploaded)
  if (!prop.empty()) {
    HIP_CHECK(hipMalloc((void**)&e->d_plinks, prop.size() * sizeof(PrimalLink)));
    h2d(e->d_plinks, prop.data(), prop.size() * sizeof(PrimalLink), e->stream);
  }
  // every factor starts unset (init_primal), then only the touched ones are ever re-initialised
  if (p.nf > 0) {
    std::vector<int32_t> h(2 * (size_t)p.nf);
    for (int64_t f = 0; f < p.nf; ++f) { h[2 * f] = all[f].a; h[2 * f + 1] = all[f].b; }
    h2d(e->d_primal, h.data(), h.size() * sizeof(int32_t), e->stream);
  }
  if (!init.empty()) {
    HIP_CHECK(hipMalloc((void**)&e->d_pinit, init.size() * sizeof(PrimalInit)));
    h2d(e->d_pinit, init.data(), init.size() * sizeof(PrimalInit), e->stream);
  }
  e->primal_t = 0;
  e->have_primal = true;
}

static void run_primal_sweep(lpmp_engine* e, int d, uint64_t t) {
  require_mode(e);
  HIP_CHECK(hipSetDevice(e->device));
  settle(e);
  begin_compute(e);
  ensure_primal(e);
  // the reference asserts primal_access_ <= timestamp (factors_messages.hxx:3304); in a release build a smaller
  // stamp lowers primal_access_ of the rounded factors only and later passes depend on the update order
  if (t < e->primal_t) throw std::runtime_error("primal pass: time stamps (2*iteration+1 / +2) must not decrease");
  if (t > e->primal_t) {   // conditionally_init_primal: primal_access_ < timestamp
    launch_primal_init(e->d_pinit, e->n_pinit, e->d_primal, e->stream);
    e->primal_t = t;
  }
  ensure_device_schedules(e, e->mode);
  e->primal_pass = true;
  try { run_schedule(e, e->sched[d][e->mode]); } catch (...) { e->primal_pass = false; throw; }
  e->primal_pass = false;
  launch_primal_propagate(e->d_plinks, e->n_pprop, e->d_primal, e->stream);
  HIP_CHECK(hipGetLastError());
}

int lpmp_compute_forward_pass_and_primal(lpmp_engine* e, uint64_t iteration) {
  return guarded([&] { run_primal_sweep(e, 0, 2 * iteration + 1); });
}
int lpmp_compute_backward_pass_and_primal(lpmp_engine* e, uint64_t iteration) {
  return guarded([&] { run_primal_sweep(e, 1, 2 * iteration + 2); });
}
int lpmp_compute_pass_and_primal(lpmp_engine* e, uint64_t iteration) {
  return guarded([&] { run_primal_sweep(e, 0, 2 * iteration + 1); run_primal_sweep(e, 1, 2 * iteration + 2); });
}

static bool primal_consistent(lpmp_engine* e) {
  HIP_CHECK(hipMemsetAsync(e->d_pbad, 0, sizeof(int), e->stream));
  launch_primal_check(e->d_plinks, e->n_plinks, e->d_primal, e->d_pbad, e->stream);
  HIP_CHECK(hipMemcpyAsync(e->h_pbad, e->d_pbad, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_CHECK(hipStreamSynchronize(e->stream));
  return *e->h_pbad == 0;
}
int lpmp_check_primal_consistency(lpmp_engine* e, int* consistent) {
  return guarded([&] {
    require_model(e);
    if (!consistent) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    ensure_primal(e);
    *consistent = primal_consistent(e) ? 1 : 0;
  });
}
int lpmp_evaluate_primal(lpmp_engine* e, double* cost) {
  return guarded([&] {
    require_model(e);
    if (!cost) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    ensure_primal(e);
    if (!primal_consistent(e)) { *cost = std::numeric_limits<double>::infinity(); return; }
    const int64_t nf = e->plan->p.nf;
    rows_refresh(e);
    launch_primal_cost(e->d_lbrecs, e->d_dual, e->d_const, e->d_primal, e->d_pcost, nf, e->stream);
    int64_t nb = std::min<int64_t>(1024, (nf + 255) / 256);
    const int64_t per = (nf + nb - 1) / nb;
    nb = (nf + per - 1) / per;
    launch_sum_stage(e->d_pcost, e->d_part, nf, per, nb, e->stream);
    HIP_CHECK(hipMemcpyAsync(e->h_part, e->d_part, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK(hipStreamSynchronize(e->stream));
    double c = e->plan->p.constant;
    for (int64_t i = 0; i < nb; ++i) c += e->h_part[i];
    *cost = c;
  });
}
int lpmp_download_primal(lpmp_engine* e, int32_t* out) {
  return guarded([&] {
    require_model(e);
    if (!out) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    ensure_primal(e);
    d2h(out, e->d_primal, 2 * (size_t)e->plan->p.nf * sizeof(int32_t), e->stream);
  });
}
int lpmp_upload_primal(lpmp_engine* e, const int32_t* in) {
  return guarded([&] {
    require_model(e);
    if (!in) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    ensure_primal(e);
    h2d(e->d_primal, in, 2 * (size_t)e->plan->p.nf * sizeof(int32_t), e->stream);
  });
}

int lpmp_compute_pass_custom(lpmp_engine* e, int64_t n, const int32_t* factors, const int64_t* om_off, const double* om,
                             const int64_t* mk_off, const uint8_t* mk) {
  return guarded([&] {
    require_model(e);
    if (n < 0 || (n > 0 && (!factors || !om_off || !mk_off))) throw std::runtime_error("bad argument");
    check_rows(n, om_off, om, mk_off, mk);
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    begin_compute(e);
    Schedule s;
    static const double dz = 0; static const uint8_t uz = 0;
    e->plan->p.make_schedule(factors, n, om_off, om ? om : &dz, mk_off, mk ? mk : &uz, s);
    check_generic_limits(e->plan->p, s);
    // the schedule lives in a scratch buffer of the engine that is refilled in place: no allocation per call
    DevSchedule& d = e->scratch;
    upload_schedule(s, d, e->stream, true, e->plan->p.force_generic);
    const bool g = e->use_graph; e->use_graph = false;
    try { run_schedule(e, d); } catch (...) { e->use_graph = g; throw; }
    e->use_graph = g;
    HIP_CHECK(hipStreamSynchronize(e->stream));
  });
}

int lpmp_schedule_create(lpmp_engine* e, int64_t n, const int32_t* factors, const int64_t* om_off, const double* om,
                         const int64_t* mk_off, const uint8_t* mk, int* id_out) {
  return lpmp_schedule_create_fused(e, n, factors, om_off, om, mk_off, mk, 0, id_out);
}
int lpmp_schedule_create_fused(lpmp_engine* e, int64_t n, const int32_t* factors, const int64_t* om_off, const double* om,
                               const int64_t* mk_off, const uint8_t* mk, int fuse, int* id_out) {
  return guarded([&] {
    require_model(e);
    if (!id_out || n < 0 || (n > 0 && (!factors || !om_off || !mk_off))) throw std::runtime_error("bad argument");
    check_rows(n, om_off, om, mk_off, mk);
    HIP_CHECK(hipSetDevice(e->device));
    Schedule s;
    static const double dz = 0; static const uint8_t uz = 0;
    static const int64_t zero_off[1] = {0};
    e->plan->p.make_schedule(std::vector<Plan::Segment>{Plan::Segment{factors, n, n > 0 ? om_off : zero_off, om ? om : &dz,
                                                                      n > 0 ? mk_off : zero_off, mk ? mk : &uz}},
                             fuse != 0 && e->use_fused, s);
    check_generic_limits(e->plan->p, s);
    auto d = std::make_unique<DevSchedule>();
    try { upload_schedule(s, *d, e->stream, false, e->plan->p.force_generic); } catch (...) { d->release(); throw; }
    e->custom.push_back(std::move(d));
    *id_out = (int)e->custom.size() - 1;
  });
}
static DevSchedule& custom_schedule(lpmp_engine* e, int id) {
  require_model(e);
  if (id < 0 || id >= (int)e->custom.size() || !e->custom[id]) throw std::runtime_error("unknown schedule id");
  return *e->custom[id];
}
int lpmp_schedule_run(lpmp_engine* e, int id) {
  return guarded([&] {
    DevSchedule& d = custom_schedule(e, id);
    if (d.adaptive_built != (e->rtype == LPMP_RTYPE_ADAPTIVE))
      throw StateError("this schedule was prepared under another send rule (adaptive sends run on other kernels): create it again");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    begin_compute(e);
    run_schedule(e, d);
  });
}
int lpmp_schedule_info(lpmp_engine* e, int id, int64_t* n_levels, int64_t* n_launches, int64_t* n_recv, int64_t* n_send,
                       int64_t* alg_bytes) {
  return guarded([&] {
    DevSchedule& d = custom_schedule(e, id);
    if (n_levels) *n_levels = d.n_levels;
    if (n_launches) *n_launches = (int64_t)d.launches.size();
    if (n_recv) *n_recv = d.n_recv;
    if (n_send) *n_send = d.n_send;
    if (alg_bytes) *alg_bytes = d.alg_bytes;
  });
}
int lpmp_schedule_destroy(lpmp_engine* e, int id) {
  return guarded([&] {
    DevSchedule& d = custom_schedule(e, id);
    HIP_CHECK(hipStreamSynchronize(e->stream));
    d.release();
    e->custom[id].reset();
  });
}

// the chain executor bounds every wait; a run that gave up leaves the duals half updated and must not pass silently
static void check_chain(lpmp_engine* e) {
  if (!e->chain_ran || !e->d_chain_abort) return;
  int32_t* h = (int32_t*)(e->pinned + 8 * 1024 + 128);
  static_assert(8 * 1024 + 128 + CHAIN_ABORT_WORDS * 4 <= PINNED_WORDS_BYTES, "pinned block");
  HIP_CHECK(hipMemcpyAsync(h, e->d_chain_abort, CHAIN_ABORT_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_CHECK(hipStreamSynchronize(e->stream));
  e->chain_ran = false;
  if (h[0] != 0) {
    std::string what;
    if (h[1] != 0) {
      const long long now = (long long)(((unsigned long long)(unsigned)h[7] << 32) | (unsigned)h[6]), t0 = (long long)(((unsigned long long)(unsigned)h[9] << 32) | (unsigned)h[8]);
      what = h[3] == -2 ? " (a mailbox granule, tag seen " + std::to_string(h[4]) + ", epoch " + std::to_string(h[5])
                        : " (ticket " + std::to_string(h[2]) + " waited for ticket " + std::to_string(h[3]) + ": flag word " + std::to_string(h[4]) + ", epoch " + std::to_string(h[5]);
      what += ", " + std::to_string((double)(now - t0) * 1e-8) + " s, " + std::to_string(h[10]) + " tickets drawn)";
    }
    HIP_CHECK(hipMemsetAsync(e->d_chain_abort, 0, CHAIN_ABORT_WORDS * sizeof(int32_t), e->stream));
    throw DeviceError("chain executor: a dependency wait timed out" + what + "; the duals are in an undefined state (LPMP_NO_CHAIN=1 selects graph replay, LPMP_CHAIN_TIMEOUT_S the bound)");
  }
}

static void compute_factor_lbs(lpmp_engine* e) {
  check_chain(e);
  rows_refresh(e);
  if (e->use_lb_tracking && !e->lb_all_stale) {
    // the sweep kernels kept d_lb current except for the entries they marked NaN: recompute only those
    const int64_t nf = e->plan->p.nf;
    HIP_CHECK(hipMemsetAsync(e->d_stale_n, 0, sizeof(unsigned long long), e->stream));
    launch_lb_collect_stale(e->d_lb, nf, e->d_stale, e->d_stale_n, e->stream);
    HIP_CHECK(hipMemcpyAsync(e->h_stale_n, e->d_stale_n, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK(hipStreamSynchronize(e->stream));
    const int64_t n_stale = (int64_t)*e->h_stale_n;
    if (n_stale <= nf / 8) {
      launch_factor_lb_list(e->d_lbrecs, e->d_dual, e->d_const, e->d_lb, e->d_stale, n_stale, e->stream);
      HIP_CHECK(hipGetLastError());
      e->last_lb_recomputed = n_stale;
      return;
    }
  }
  e->last_lb_recomputed = e->plan->p.nf;
  for (const auto& r : e->lb_runs) {
    if (r.cls == 0 || !launch_dense_lb(r.cls, e->d_lbrecs, e->d_dual, e->d_const, e->d_lb, r.first, r.count, e->stream))
      launch_factor_lb(e->d_lbrecs + r.first, e->d_dual, e->d_const, e->d_lb + r.first, r.count, e->stream);
  }
  HIP_CHECK(hipGetLastError());
  e->lb_all_stale = false;
}

int lpmp_lower_bound(lpmp_engine* e, double* out) {
  return guarded([&] {
    require_model(e);
    if (!out) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    if (spec_lower_bound(e, out)) return;      // the caller is inside a batch of passes that ran ahead: that pass's own row
    compute_factor_lbs(e);
    const int64_t nf = e->plan->p.nf;
    int64_t nb = std::min<int64_t>(1024, (nf + 255) / 256);
    const int64_t per = (nf + nb - 1) / nb;
    nb = (nf + per - 1) / per;
    launch_sum_stage(e->d_lb, e->d_part, nf, per, nb, e->stream);
    HIP_CHECK(hipMemcpyAsync(e->h_part, e->d_part, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_CHECK(hipStreamSynchronize(e->stream));
    double lb = e->plan->p.constant;
    for (int64_t i = 0; i < nb; ++i) lb += e->h_part[i];
    *out = lb;
  });
}

int lpmp_factor_lower_bounds(lpmp_engine* e, double* out) {
  return guarded([&] {
    require_model(e);
    if (!out) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    compute_factor_lbs(e);
    d2h(out, e->d_lb, (size_t)e->plan->p.nf * sizeof(double), e->stream);
  });
}

int lpmp_synchronize(lpmp_engine* e) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);                                  // after this call the (possibly borrowed) dual buffer holds the caller's state
    rows_flush(e);
    if (e->rows && !e->own_dual) e->rows_stale = true;   // ... and the caller may write it: the rows are refreshed before the next pass
    HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->timing) e->drain_timing();
    check_chain(e);
    if (!guarded_ok(e->pinned, PINNED_WORDS_BYTES)) throw DeviceError("guard region of the engine's pinned words was overwritten");
    staging().check();
  });
}

// common entry of the lpmp_boundary_* calls (boundary.hip): the engine's device is current, a speculative batch is
// settled, and an aborted chain run is reported instead of its duals being consumed
int lpmp_boundary_enter(lpmp_engine* e) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    check_chain(e);
    // the boundary / halo kernels address dense pairwise vectors through device offsets, i.e. in the ROWS when that layout is on:
    // whatever the caller put into the packed array since the last hand-over (lpmp_upload_duals, a write after lpmp_synchronize /
    // lpmp_device_duals) must be in the rows before they are read, and before lpmp_boundary_leave marks the rows as the newer copy
    rows_refresh(e);
  });
}

int lpmp_streaming_access(const lpmp_engine* e) { return e && e->plan ? (e->nt_flag ? 1 : 0) : -1; }

int64_t lpmp_dual_size(const lpmp_engine* e) { return e && e->plan ? e->plan->p.f_doff[e->plan->p.nf] : 0; }

int lpmp_download_duals(lpmp_engine* e, double* out) {
  return guarded([&] {
    require_model(e);
    if (!out) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    check_chain(e);
    rows_flush(e);
    d2h(out, e->d_dual, (size_t)lpmp_dual_size(e) * sizeof(double), e->stream);
  });
}
int lpmp_upload_duals(lpmp_engine* e, const double* in) {
  return guarded([&] {
    require_model(e);
    if (!in) throw std::runtime_error("null argument");
    HIP_CHECK(hipSetDevice(e->device));
    settle(e);
    h2d(e->d_dual, in, (size_t)lpmp_dual_size(e) * sizeof(double), e->stream);
    if (e->rows) { e->rows_stale = true; e->packed_stale = false; }
    e->lb_all_stale = true;
  });
}
int lpmp_invalidate_lower_bounds(lpmp_engine* e) {
  // (the caller says it wrote the packed dual array behind the engine's back: with the rows layout, what it wrote there for
  // a dense pairwise factor replaces the row's vectors — after what the rows hold has been written out, so that everything
  // the caller did not touch survives)
  return guarded([&] { require_model(e); settle(e); if (e->rows) { rows_flush(e); e->rows_stale = true; } e->lb_all_stale = true; });
}
// The packed dual array (serialize_dual order) on the device.  With the rows layout the dense pairwise factors' vectors are
// written out to it first (stream-ordered on the engine's stream), and the caller is assumed to write it: the rows are
// refreshed from it before the next pass — callers that only read may say so by not calling this between passes.
void* lpmp_device_duals(lpmp_engine* e) {
  if (!e) return nullptr;
  if (e->rows) { try { (void)hipSetDevice(e->device); rows_flush(e); e->rows_stale = true; } catch (const std::exception& ex) { g_error = ex.what(); return nullptr; } }
  return e->d_dual;
}
// internal (boundary.hip): the base pointer itself, and where a packed dual offset lives on the device
void* lpmp_engine_dual_base(lpmp_engine* e) { return e ? e->d_dual : nullptr; }
int64_t lpmp_engine_device_dual_offset(lpmp_engine* e, int64_t packed_off) {
  if (!e || !e->plan || e->plan->p.dev_doff.empty()) return packed_off;
  const Plan& p = e->plan->p;
  const int64_t f = (int64_t)(std::upper_bound(p.f_doff.begin(), p.f_doff.end(), packed_off) - p.f_doff.begin()) - 1;
  if (f < 0 || f >= p.nf) return packed_off;
  return p.dev_doff[f] + (packed_off - p.f_doff[f]);
}
// internal (boundary.hip): is [packed_off, packed_off + len) a run of doubles inside ONE factor's dual?  (what the device
// offset mapping above and the kernels that follow it assume)
int lpmp_engine_dual_range_ok(lpmp_engine* e, int64_t packed_off, int64_t len) {
  if (!e || !e->plan || len < 0 || packed_off < 0) return 0;
  const Plan& p = e->plan->p;
  if (packed_off + len > p.f_doff[p.nf]) return 0;
  if (len == 0) return 1;
  const int64_t f = (int64_t)(std::upper_bound(p.f_doff.begin(), p.f_doff.end(), packed_off) - p.f_doff.begin()) - 1;
  return f >= 0 && f < p.nf && packed_off + len <= p.f_doff[f + 1] ? 1 : 0;
}
int lpmp_boundary_leave(lpmp_engine* e) {      // a boundary kernel wrote duals through device offsets: only the tracked bounds are stale
  return guarded([&] { require_model(e); e->lb_all_stale = true; if (e->rows) e->packed_stale = true; });
}
// Persistent launches (chain executor, joined passes in Infinity-Cache order) assume that resident workgroups keep running, i.e.
// that the device is this process's own (kernels.hip): a host that knows it shares the device switches them off per engine.
// The environment's LPMP_NO_CHAIN / LPMP_NO_BLOCKED_PASSES keep the last word (off stays off).
int lpmp_set_persistent_launches(lpmp_engine* e, int on) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    if (e->plan) { HIP_CHECK(hipSetDevice(e->device)); settle(e); }
    const char* nc = std::getenv("LPMP_NO_CHAIN"); const char* nb = std::getenv("LPMP_NO_BLOCKED_PASSES");
    e->use_chain = on != 0 && !(nc && nc[0] == '1');
    e->use_blocked_passes = on != 0 && !(nb && nb[0] == '1');
  });
}
int lpmp_persistent_launches(const lpmp_engine* e) { return e && e->use_chain && e->use_blocked_passes ? 1 : 0; }
// "pci=<domain:bus:device.function> uuid=<32 hex digits>" of a HIP device ordinal: what tells two ranks that they sit on the same
// physical GPU when every rank has a visibility mask of its own (then both see "device 0")
int lpmp_device_identity(int device, char* out, int64_t cap) {
  return guarded([&] {
    if (!out || cap < 64) throw std::runtime_error("lpmp_device_identity: buffer of at least 64 bytes needed");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw DeviceError("no HIP device available");
    if (device < 0 || device >= n) throw DeviceError("device ordinal out of range");
    char pci[32] = {0};
    HIP_CHECK(hipDeviceGetPCIBusId(pci, (int)sizeof(pci), device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    std::string s = std::string("pci=") + pci + " uuid=";
    static const char* hex = "0123456789abcdef";
    for (unsigned char c : prop.uuid.bytes) { s += hex[c >> 4]; s += hex[c & 15]; }
    if ((int64_t)s.size() + 1 > cap) s.resize((size_t)cap - 1);
    std::memcpy(out, s.c_str(), s.size() + 1);
  });
}
int lpmp_set_rows_layout(lpmp_engine* e, int on) {
  return guarded([&] { if (!e) throw std::runtime_error("null engine"); e->want_rows = on != 0; });
}
int lpmp_rows_layout(const lpmp_engine* e) { return e && e->rows ? 1 : 0; }
int64_t lpmp_lower_bound_recomputed(const lpmp_engine* e) { return e ? e->last_lb_recomputed : -1; }
void* lpmp_engine_stream(lpmp_engine* e) { return e ? (void*)e->stream : nullptr; }
const lpmp_plan* lpmp_engine_plan(const lpmp_engine* e) { return e ? e->plan.get() : nullptr; }
lpmp_plan* lpmp_engine_plan_mut(lpmp_engine* e) { return e ? e->plan.get() : nullptr; }

int lpmp_enable_kernel_timing(lpmp_engine* e, int on) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    settle(e);
    HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->timing) e->drain_timing();
    e->timing = on != 0;
  });
}
int lpmp_get_kernel_timing(lpmp_engine* e, int n, double* ms, int64_t* launches, int64_t* factors, int64_t* receives, int64_t* bytes) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    HIP_CHECK(hipStreamSynchronize(e->stream));
    e->drain_timing();
    for (int c = 0; c < n && c < KC_COUNT; ++c) {
      if (ms) ms[c] = e->ct[c].ms;
      if (launches) launches[c] = e->ct[c].launches;
      if (factors) factors[c] = e->ct[c].factors;
      if (receives) receives[c] = e->ct[c].receives;
      if (bytes) bytes[c] = e->ct[c].bytes;
    }
  });
}
// of the launches lpmp_get_kernel_timing reports per class: how many were persistent chain-executor launches
int lpmp_get_chain_launches(lpmp_engine* e, int n, int64_t* chain_launches) {
  return guarded([&] {
    if (!e || !chain_launches) throw std::runtime_error("null argument");
    HIP_CHECK(hipStreamSynchronize(e->stream));
    e->drain_timing();
    for (int c = 0; c < n && c < KC_COUNT; ++c) chain_launches[c] = e->ct[c].chain_launches;
  });
}
int lpmp_reset_kernel_timing(lpmp_engine* e) {
  return guarded([&] {
    if (!e) throw std::runtime_error("null engine");
    HIP_CHECK(hipStreamSynchronize(e->stream));
    e->drain_timing();
    for (auto& c : e->ct) c = ClassTiming();
  });
}

int lpmp_synth_fill(void* p, int64_t n, uint64_t seed, uint64_t first, void* stream) {
  return guarded([&] {
    if (!p && n > 0) throw std::runtime_error("null argument");
    launch_synth_fill((double*)p, n, seed, first, (hipStream_t)stream);
    HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"
