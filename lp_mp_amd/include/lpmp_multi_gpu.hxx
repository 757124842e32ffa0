// lpmp_multi_gpu.hxx — the partitioned (multi-GPU) sweep driven from C++ over the C ABI and RCCL.
//
// One process per GPU (or several parts per process: parts_per_rank), every part of the factor graph on its own
// lpmp_engine.  A pass = the part's main sweeps (iterator-range passes of the reference, include/LP_MP.h:981-1005: the
// part's own update lists and anisotropic rows with the ghost factors dropped) + the boundary step of DESIGN.md 7:
//
//     lpmp_schedule_run(ghost receive)  ->  lpmp_boundary_pack   -- ncclSend / ncclRecv, exchange #1 -->
//     lpmp_boundary_reply               <-- ncclSend / ncclRecv, exchange #2 --
//     lpmp_boundary_fold  ->  lpmp_schedule_run(ghost send)
//
// The exchange is ONE ncclGroupStart / ncclGroupEnd of point-to-point transfers per direction on the stream the engines
// work on (an all-to-all-v: xGMI is point-to-point, every pair of parts that shares cut edges talks directly; a part
// pair on one rank goes through ncclSend / ncclRecv to self inside the same group).  The bound of the whole model is
// the ncclAllReduce of the parts' bounds.  This is the loop lp_mp_amd/multi_gpu.py (PartitionedSweep) runs through
// torch.distributed; both produce the same duals bit for bit (tests/test_multi_gpu.py).
//
// Row strips of a grid MRF are built here in closed form (strip_part = multi_gpu.strip_local_part); any other partition
// can be handed in as a part_model filled by the caller.
//
// Needs: include/lpmp_engine.h, <rccl/rccl.h>, HIP runtime; link with -llpmp_engine -lrccl -lamdhip64.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/lpmp_engine.h"

namespace lpmp_mgpu {

inline void hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
inline void nccl_ok(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
inline void lpmp_ok(int rc) { if (rc != LPMP_OK) throw std::runtime_error(lpmp_last_error()); }   // the type the reference throws (LP_MP.h:458)

// ncclGroupStart ... ncclGroupEnd that is closed on every path: an exception between the two (a failed send, a count check)
// must not leave the group open — the next RCCL call of the process would be queued into it and never run
struct nccl_group {
  bool open = false;
  nccl_group() { nccl_ok(ncclGroupStart(), "ncclGroupStart"); open = true; }
  void end() { open = false; nccl_ok(ncclGroupEnd(), "ncclGroupEnd"); }
  ~nccl_group() { if (open) (void)ncclGroupEnd(); }
  nccl_group(const nccl_group&) = delete;
  nccl_group& operator=(const nccl_group&) = delete;
};

// Bounds a call that has no time-out of its own (ncclCommInitRank waits for ever for a rank that never comes): when the
// guarded scope is not left within timeout_s the process says which rank waited for what and EXITS non-zero (3) — never a retry
// in a process that has touched the GPU, never a re-exec.  timeout_s <= 0: no bound.
struct exit_watchdog {
  std::mutex m; std::condition_variable cv; bool done = false; std::thread th;
  exit_watchdog(double timeout_s, std::string what) {
    if (timeout_s <= 0) return;
    th = std::thread([this, timeout_s, what] {
      std::unique_lock<std::mutex> l(m);
      if (!cv.wait_for(l, std::chrono::duration<double>(timeout_s), [this] { return done; })) {
        std::fprintf(stderr, "lpmp_mgpu: %s did not return within %.0f s; exiting with code 3\n", what.c_str(), timeout_s);
        std::fflush(stderr);
        ::_exit(3);
      }
    });
  }
  ~exit_watchdog() {
    { std::lock_guard<std::mutex> l(m); done = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
  exit_watchdog(const exit_watchdog&) = delete;
  exit_watchdog& operator=(const exit_watchdog&) = delete;
};

constexpr double BOUNDARY_SHARE = 0.375;   // send weight of a boundary variable's cut messages, shared out

// one part of the partitioned model: an MRF in the layout of the reference's MRF constructor (unaries, ghosts, then the
// owned pairwise factors; per edge add_message<Left>(u_i, p), add_message<Right>(u_j, p), AddFactorRelation(u_i, p),
// AddFactorRelation(p, u_j)) + its cut lists in EXCHANGE ORDER (by peer part, then global edge id)
struct part_model {
  int32_t part = 0, n_parts = 1, L = 0;
  bool potts = false;
  int64_t n_local = 0, n_ghost = 0, n_edges = 0;
  std::vector<int32_t> f_type, f_dim0, f_dim1, m_type, m_left, m_right, rel;
  std::vector<uint8_t> f_kind, f_flags;
  lpmp_msg_type mtypes[2];
  // costs are generated in HBM from the counter stream (lpmp_synth_fill): unaries at un_first, pairwise data at pw_first
  uint64_t seed = 1, un_first = 0, pw_first = 0;
  std::vector<int32_t> out_peer, out_ghost; std::vector<int64_t> out_key;   // cut messages this part owns (ghost factor ids)
  std::vector<int32_t> in_peer, in_unary; std::vector<int64_t> in_key;     // cut messages owned elsewhere ending in a local unary
  int64_t n_factors() const { return (int64_t)f_type.size(); }
  int64_t const_doubles() const { return potts ? n_edges : n_edges * (int64_t)L * L; }
  int64_t dual_doubles() const { return (n_local + n_ghost) * (int64_t)L + n_edges * 2 * (int64_t)L; }
  lpmp_model view(const double* const_dev, const double* dual_dev) const {
    lpmp_model m{};
    m.n_ftypes = 2; m.ftype_computes_primal = nullptr; m.n_mtypes = 2; m.mtypes = mtypes;
    m.n_factors = n_factors(); m.f_type = f_type.data(); m.f_kind = f_kind.data(); m.f_flags = f_flags.data();
    m.f_dim0 = f_dim0.data(); m.f_dim1 = f_dim1.data(); m.const_data = const_dev; m.dual_data = dual_dev;
    m.n_messages = (int64_t)m_type.size(); m.m_type = m_type.data(); m.m_left = m_left.data(); m.m_right = m_right.data();
    m.n_rel_fwd = (int64_t)rel.size() / 2; m.rel_fwd = rel.data();
    // AddFactorRelation(f1, f2) = forward f1 -> f2 and backward f2 -> f1 (LP_MP.h:698-702)
    m.n_rel_bwd = (int64_t)rel_bwd.size() / 2; m.rel_bwd = rel_bwd.data();
    return m;
  }
  std::vector<int32_t> rel_bwd;
  void finish_relations() { rel_bwd.resize(rel.size()); for (size_t i = 0; i + 1 < rel.size(); i += 2) { rel_bwd[i] = rel[i + 1]; rel_bwd[i + 1] = rel[i]; } }
};

// var[r * W + c] = position of grid node (r, c) in the variable order (synthetic.grid_variable_order)
inline std::vector<int64_t> grid_variable_order(int H, int W, bool colour_major) {
  std::vector<int64_t> var((size_t)H * W);
  if (!colour_major) { std::iota(var.begin(), var.end(), 0); return var; }
  int64_t nb = 0;
  for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) if (((r + c) & 1) == 0) ++nb;
  int64_t b = 0, w = nb;
  for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) var[(size_t)r * W + c] = ((r + c) & 1) == 0 ? b++ : w++;
  return var;
}

// rows [part * H, (part + 1) * H) of a (n_parts * H) x W grid: multi_gpu.strip_local_part in closed form.  Edges in row-major
// node order, right edge then down edge per node (synthetic.grid_edges), then the W cut edges to the next strip in column
// order; a pairwise factor belongs to the strip of its earlier endpoint, the lower strip's endpoint is a ghost here.
inline part_model strip_part(int H, int W, int L, bool potts, bool colour_major, int part, int n_parts, uint64_t seed) {
  part_model p;
  p.part = part; p.n_parts = n_parts; p.L = L; p.potts = potts; p.seed = seed;
  const int64_t n_loc = (int64_t)H * W, e_int = (int64_t)H * (W - 1) + (int64_t)W * (H - 1);
  const bool has_down = part < n_parts - 1, has_up = part > 0;
  p.n_local = n_loc; p.n_ghost = has_down ? W : 0;
  const std::vector<int64_t> var = grid_variable_order(H, W, colour_major);
  std::vector<int64_t> li, lj;
  for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) {
    const int64_t a = var[(size_t)r * W + c];
    if (c < W - 1) { const int64_t b = var[(size_t)r * W + c + 1]; li.push_back(std::min(a, b)); lj.push_back(std::max(a, b)); }
    if (r < H - 1) { const int64_t b = var[(size_t)(r + 1) * W + c]; li.push_back(std::min(a, b)); lj.push_back(std::max(a, b)); }
  }
  if (has_down) for (int c = 0; c < W; ++c) { li.push_back(var[(size_t)(H - 1) * W + c]); lj.push_back(n_loc + c); }
  p.n_edges = (int64_t)li.size();
  const int64_t n_vec = n_loc + p.n_ghost, nf = n_vec + p.n_edges;
  p.f_type.assign((size_t)nf, 0); p.f_kind.assign((size_t)nf, LPMP_F_VECTOR); p.f_flags.assign((size_t)nf, 0);
  p.f_dim0.assign((size_t)nf, L); p.f_dim1.assign((size_t)nf, 0);
  for (int64_t e = 0; e < p.n_edges; ++e) {
    const int64_t f = n_vec + e;
    p.f_type[f] = 1; p.f_kind[f] = potts ? LPMP_F_PAIRWISE_POTTS : LPMP_F_PAIRWISE_DENSE; p.f_dim1[f] = L;
    p.m_type.push_back(0); p.m_left.push_back((int32_t)li[e]); p.m_right.push_back((int32_t)f);
    p.m_type.push_back(1); p.m_left.push_back((int32_t)lj[e]); p.m_right.push_back((int32_t)f);
    p.rel.push_back((int32_t)li[e]); p.rel.push_back((int32_t)f);
    p.rel.push_back((int32_t)f); p.rel.push_back((int32_t)lj[e]);
  }
  p.finish_relations();
  // unary = left factor, schedule `left`, unary side variable count (0), pairwise side exactly 1
  p.mtypes[0] = lpmp_msg_type{0, 1, LPMP_SCHED_LEFT, 0, 1, LPMP_M_UNARY_PAIRWISE, 0, 0};
  p.mtypes[1] = lpmp_msg_type{0, 1, LPMP_SCHED_LEFT, 0, 1, LPMP_M_UNARY_PAIRWISE, 1, 0};
  const int64_t n_vars = (int64_t)n_parts * n_loc, esz = potts ? 1 : (int64_t)L * L;
  const int64_t e_first = (int64_t)part * (e_int + W);
  p.un_first = (uint64_t)((int64_t)part * n_loc * L);
  p.pw_first = (uint64_t)(n_vars * L + e_first * esz);
  if (has_down) for (int c = 0; c < W; ++c) { p.out_peer.push_back(part + 1); p.out_ghost.push_back((int32_t)(n_loc + c)); p.out_key.push_back(e_first + e_int + c); }
  if (has_up) for (int c = 0; c < W; ++c) { p.in_peer.push_back(part - 1); p.in_unary.push_back((int32_t)var[(size_t)c]); p.in_key.push_back((int64_t)(part - 1) * (e_int + W) + e_int + c); }
  return p;
}

// the communicator: RANK / WORLD_SIZE processes, parts_per_rank parts each; part q lives on rank q / parts_per_rank
struct rccl_world {
  ncclComm_t comm = nullptr; int rank = 0, world = 1, parts_per_rank = 1;
  hipStream_t stream = nullptr;
  double* d_scalar = nullptr;
  int rank_of(int part) const { return part / parts_per_rank; }
  // The ncclUniqueId travels over a TCP connection to rank 0 on MASTER_ADDR : MASTER_PORT (the rendezvous every launcher
  // names), not through a file: nothing is left behind that a later launch on the same port could mistake for its own id
  // (a stale id makes ncclCommInitRank wait for ever), and every wait here is bounded (timeout_s).
  static void send_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n > 0) { const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) throw std::runtime_error("rccl_world: sending the ncclUniqueId failed"); c += k; n -= (size_t)k; }
  }
  static void recv_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n > 0) { const ssize_t k = ::recv(fd, c, n, 0); if (k <= 0) throw std::runtime_error("rccl_world: receiving the ncclUniqueId failed"); c += k; n -= (size_t)k; }
  }
  struct id_packet { uint64_t magic; int32_t world, rank; ncclUniqueId id; };
  static constexpr uint64_t ID_MAGIC = 0x4C504D5052434349ULL;   // "LPMPRCCI"
  static void hand_out_id(ncclUniqueId& id, int rank, int world, const std::string& addr, int port, double timeout_s) {
    if (world <= 1) return;
    sockaddr_in sa{}; sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, addr.c_str(), &sa.sin_addr) != 1) throw std::runtime_error("rccl_world: MASTER_ADDR must be an IPv4 address (e.g. 127.0.0.1), got " + addr);
    const auto t0 = std::chrono::steady_clock::now();
    auto left = [&] { return timeout_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (rank == 0) {
      const int ls = ::socket(AF_INET, SOCK_STREAM, 0); if (ls < 0) throw std::runtime_error("rccl_world: socket()");
      int one = 1; (void)setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
      sockaddr_in any = sa; any.sin_addr.s_addr = htonl(INADDR_ANY);
      if (::bind(ls, (sockaddr*)&any, sizeof(any)) != 0 || ::listen(ls, world) != 0) { ::close(ls); throw std::runtime_error("rccl_world: cannot listen on port " + std::to_string(port)); }
      std::vector<char> served((size_t)world, 0);
      for (int got = 0; got < world - 1;) {
        pollfd pf{ls, POLLIN, 0};
        const double l = left();
        if (l <= 0 || ::poll(&pf, 1, (int)std::min(l * 1000.0, 1000.0)) < 0) { if (l <= 0) { ::close(ls); throw std::runtime_error("rccl_world: timed out waiting for " + std::to_string(world - 1 - got) + " rank(s) to fetch the ncclUniqueId"); } continue; }
        if (!(pf.revents & POLLIN)) continue;
        const int fd = ::accept(ls, nullptr, nullptr); if (fd < 0) continue;
        timeval tv{5, 0}; (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        id_packet hello{};
        try {
          recv_all(fd, &hello, offsetof(id_packet, id));
          // a connection that is not one of this launch's ranks (wrong magic / world, a rank seen before) gets nothing
          if (hello.magic == ID_MAGIC && hello.world == world && hello.rank > 0 && hello.rank < world && !served[(size_t)hello.rank]) {
            id_packet out{ID_MAGIC, world, 0, id};
            send_all(fd, &out, sizeof(out));
            served[(size_t)hello.rank] = 1; ++got;
          }
        } catch (const std::exception&) {}
        ::close(fd);
      }
      ::close(ls);
    } else {
      for (;;) {
        const int fd = ::socket(AF_INET, SOCK_STREAM, 0); if (fd < 0) throw std::runtime_error("rccl_world: socket()");
        if (::connect(fd, (sockaddr*)&sa, sizeof(sa)) == 0) {
          timeval tv{(long)std::max(1.0, left()), 0}; (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
          id_packet hello{ID_MAGIC, world, rank, {}}, in{};
          try {
            send_all(fd, &hello, offsetof(id_packet, id));
            recv_all(fd, &in, sizeof(in));
            ::close(fd);
            if (in.magic != ID_MAGIC || in.world != world) throw std::runtime_error("rccl_world: " + addr + ":" + std::to_string(port) + " is not this launch's rank 0");
            id = in.id;
            return;
          } catch (const std::exception&) { ::close(fd); if (left() <= 0) throw; }
        } else ::close(fd);
        if (left() <= 0) throw std::runtime_error("rccl_world: no rank 0 on " + addr + ":" + std::to_string(port) + " within " + std::to_string((int)timeout_s) + " s");
        struct timespec ts{0, 20000000}; nanosleep(&ts, nullptr);
      }
    }
  }
  void init(int rank_, int world_, int parts_per_rank_, const std::string& master_addr, int master_port, hipStream_t s, double timeout_s = 120.0) {
    rank = rank_; world = world_; parts_per_rank = parts_per_rank_; stream = s;
    ncclUniqueId id;
    if (rank == 0) nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId");
    hand_out_id(id, rank, world, master_addr, master_port, timeout_s);
    {
      // (LPMP_RCCL_INIT_TIMEOUT_S overrides; the bound covers the rendezvous only, not later collectives)
      const char* ev = std::getenv("LPMP_RCCL_INIT_TIMEOUT_S");
      exit_watchdog wd(ev ? std::atof(ev) : timeout_s, "rank " + std::to_string(rank) + " of " + std::to_string(world) + ": ncclCommInitRank");
      nccl_ok(ncclCommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    }
    hip_ok(hipMalloc((void**)&d_scalar, 2 * sizeof(double)), "hipMalloc");
    self_test(timeout_s);
  }
  // one tiny all-reduce under the same bound before any real exchange: a communicator whose ranks cannot reach each other fails
  // HERE, named, instead of inside the first timed pass
  void self_test(double timeout_s) {
    exit_watchdog wd(timeout_s, "rank " + std::to_string(rank) + " of " + std::to_string(world) + ": the first ncclAllReduce");
    const double got = all_reduce_sum(1.0);
    if (got != (double)world) throw std::runtime_error("rccl_world: self test: all-reduce of 1 over " + std::to_string(world) + " ranks gave " + std::to_string(got));
  }
  // Where a pass spends its time (tools/mgpu_rccl_driver.cpp --time: an untimed repetition of the timed passes): with the probe on,
  // every exchange — pack -> ncclGroup of sends / receives -> unpack — is bracketed by a pair of events on the stream all parts of
  // this rank work on.  The span includes waiting for the slowest peer.
  struct exchange_probe {
    bool on = false; hipEvent_t open = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans; int64_t bytes_out = 0, exchanges = 0;
  } probe;
  void probe_begin() { if (!probe.on) return; hip_ok(hipEventCreate(&probe.open), "hipEventCreate"); hip_ok(hipEventRecord(probe.open, stream), "hipEventRecord"); }
  void probe_end(int64_t bytes_out) {
    if (!probe.on) return;
    hipEvent_t e = nullptr;
    hip_ok(hipEventCreate(&e), "hipEventCreate"); hip_ok(hipEventRecord(e, stream), "hipEventRecord");
    probe.spans.emplace_back(probe.open, e); probe.open = nullptr; probe.bytes_out += bytes_out; ++probe.exchanges;
  }
  // milliseconds inside exchanges since the probe was switched on (synchronises the stream; the events are released)
  double probe_exchange_ms() {
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    double ms = 0;
    for (auto& s : probe.spans) { float t = 0; hip_ok(hipEventElapsedTime(&t, s.first, s.second), "hipEventElapsedTime"); ms += t; (void)hipEventDestroy(s.first); (void)hipEventDestroy(s.second); }
    probe.spans.clear();
    return ms;
  }
  double all_reduce_max(double x) {
    hip_ok(hipMemcpyAsync(d_scalar, &x, sizeof(double), hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    nccl_ok(ncclAllReduce(d_scalar, d_scalar + 1, 1, ncclDouble, ncclMax, comm, stream), "ncclAllReduce");
    double out = 0;
    hip_ok(hipMemcpyAsync(&out, d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    return out;
  }
  // the probe's numbers of `run()` (n_passes passes), maximum over the ranks: {compute, exchange} ms per pass, exchanges and bytes
  // sent per pass on the busiest rank
  struct probe_result { double compute_ms = 0, exchange_ms = 0, exchanges = 0, bytes_out = 0; };
  template <class RUN> probe_result probe_run(RUN&& run, int n_passes) {
    (void)all_reduce_sum(0.0);
    probe = exchange_probe(); probe.on = true;
    hipEvent_t a = nullptr, b = nullptr;
    hip_ok(hipEventCreate(&a), "hipEventCreate"); hip_ok(hipEventCreate(&b), "hipEventCreate");
    hip_ok(hipEventRecord(a, stream), "hipEventRecord");
    run();
    hip_ok(hipEventRecord(b, stream), "hipEventRecord");
    const double ex = probe_exchange_ms();
    float total = 0;
    hip_ok(hipEventElapsedTime(&total, a, b), "hipEventElapsedTime");
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    probe.on = false;
    probe_result r;
    r.exchange_ms = all_reduce_max(ex / n_passes); r.compute_ms = all_reduce_max(((double)total - ex) / n_passes);
    r.exchanges = all_reduce_max((double)probe.exchanges / n_passes); r.bytes_out = all_reduce_max((double)probe.bytes_out / n_passes);
    return r;
  }
  double all_reduce_sum(double x) {
    hip_ok(hipMemcpyAsync(d_scalar, &x, sizeof(double), hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    nccl_ok(ncclAllReduce(d_scalar, d_scalar + 1, 1, ncclDouble, ncclSum, comm, stream), "ncclAllReduce");
    double out = 0;
    hip_ok(hipMemcpyAsync(&out, d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    return out;
  }
  void destroy() {
    if (d_scalar) { (void)hipFree(d_scalar); d_scalar = nullptr; }
    if (comm) { (void)ncclCommDestroy(comm); comm = nullptr; }
  }
};

// one part on its engine: schedules, boundary, exchange buffers
class part_sweep {
 public:
  part_model pm;
  lpmp_engine* e = nullptr;
  lpmp_boundary* bd = nullptr;
  double *d_const = nullptr, *d_dual = nullptr, *d_send = nullptr, *d_recv = nullptr, *d_reply = nullptr, *d_back = nullptr;
  int sid_f = -1, sid_b = -1, sid_fb = -1, sid_ghost_recv = -1, sid_ghost_send = -1;
  std::vector<int64_t> out_count, in_count;   // doubles per peer PART, exchange order
  int64_t n_out = 0, n_in = 0;
  int64_t updates_per_pass = 0;

  part_sweep() = default;
  part_sweep(const part_sweep&) = delete;
  part_sweep& operator=(const part_sweep&) = delete;
  ~part_sweep() {
    if (bd) lpmp_boundary_destroy(bd);
    if (e) lpmp_destroy(e);
    for (double* p : {d_const, d_dual, d_send, d_recv, d_reply, d_back}) if (p) (void)hipFree(p);
  }

  void build(part_model&& model, int device, hipStream_t stream, int mode, bool boundary_every_pass) {
    pm = std::move(model);
    hip_ok(hipSetDevice(device), "hipSetDevice");
    const int64_t nc = std::max<int64_t>(pm.const_doubles(), 2), nd = pm.dual_doubles();
    hip_ok(hipMalloc((void**)&d_const, (size_t)nc * sizeof(double)), "hipMalloc const");
    hip_ok(hipMalloc((void**)&d_dual, (size_t)nd * sizeof(double)), "hipMalloc dual");
    hip_ok(hipMemsetAsync(d_dual, 0, (size_t)nd * sizeof(double), stream), "hipMemsetAsync");
    lpmp_ok(lpmp_synth_fill(d_const, pm.const_doubles(), pm.seed, pm.pw_first, stream));
    lpmp_ok(lpmp_synth_fill(d_dual, pm.n_local * pm.L, pm.seed, pm.un_first, stream));   // ghosts stay 0
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    lpmp_ok(lpmp_create(device, &e));
    lpmp_ok(lpmp_set_stream(e, stream));
    const lpmp_model m = pm.view(d_const, d_dual);
    lpmp_ok(lpmp_upload_model(e, &m, LPMP_MEM_DEVICE, LPMP_MEM_DEVICE));
    lpmp_ok(lpmp_set_reparametrization(e, mode));
    // 1. main sweeps: the part's own update lists and weight rows with the ghost factors dropped
    lpmp_plan* plan = lpmp_engine_plan_mut(e);
    const int64_t n_vec = pm.n_local + pm.n_ghost;
    rows fwd = main_rows(plan, LPMP_FORWARD, mode, n_vec), bwd = main_rows(plan, LPMP_BACKWARD, mode, n_vec);
    sid_f = create(fwd, false); sid_b = create(bwd, false);
    if (boundary_every_pass) { rows fb = cat(fwd, bwd); sid_fb = create(fb, true); }
    for (int sid : {sid_f, sid_b}) {
      int64_t nr = 0, ns = 0;
      lpmp_ok(lpmp_schedule_info(e, sid, nullptr, nullptr, &nr, &ns, nullptr));
      updates_per_pass += nr + ns;
    }
    // 2. boundary passes on the ghosts: every ghost has exactly one message (side 1 of its cut edge)
    rows gr, gs;
    for (int64_t g = pm.n_local; g < n_vec; ++g) {
      gr.f.push_back((int32_t)g); gr.om.push_back(0.0); gr.mk.push_back(1); gr.om_off.push_back((int64_t)gr.om.size()); gr.mk_off.push_back((int64_t)gr.mk.size());
      gs.f.push_back((int32_t)g); gs.om.push_back(1.0); gs.mk.push_back(0); gs.om_off.push_back((int64_t)gs.om.size()); gs.mk_off.push_back((int64_t)gs.mk.size());
    }
    sid_ghost_recv = create(gr, false); sid_ghost_send = create(gs, false);
    // 3. boundary arithmetic + exchange plan (doubles per peer part)
    const int64_t nf = pm.n_factors();
    std::vector<int64_t> doff((size_t)nf + 1, 0);
    for (int64_t f = 0; f < nf; ++f) doff[f + 1] = doff[f] + lpmp_factor_dual_size(pm.f_kind[f], pm.f_dim0[f], pm.f_dim1[f]);
    std::vector<int64_t> out_off, in_off, in_order(pm.in_unary.size());
    std::vector<int32_t> out_len, in_len;
    out_count.assign((size_t)pm.n_parts, 0); in_count.assign((size_t)pm.n_parts, 0);
    for (size_t i = 0; i < pm.out_ghost.size(); ++i) { out_off.push_back(doff[pm.out_ghost[i]]); out_len.push_back(pm.f_dim0[pm.out_ghost[i]]); out_count[pm.out_peer[i]] += out_len.back(); }
    std::vector<int64_t> k_cut((size_t)nf, 0);
    for (int32_t u : pm.in_unary) ++k_cut[u];
    std::vector<double> in_omega;
    for (size_t i = 0; i < pm.in_unary.size(); ++i) {
      in_off.push_back(doff[pm.in_unary[i]]); in_len.push_back(pm.f_dim0[pm.in_unary[i]]); in_count[pm.in_peer[i]] += in_len.back();
      in_omega.push_back(BOUNDARY_SHARE / (double)std::max<int64_t>(k_cut[pm.in_unary[i]], 1));
    }
    // a variable with several cut messages receives / sends them in its message-list order: side-1 messages sit in a
    // LIFO list (reference factors_messages.hxx:2030-2041) => descending global edge id
    std::iota(in_order.begin(), in_order.end(), 0);
    std::stable_sort(in_order.begin(), in_order.end(), [&](int64_t a, int64_t b) {
      return pm.in_unary[a] != pm.in_unary[b] ? pm.in_unary[a] < pm.in_unary[b] : pm.in_key[a] > pm.in_key[b];
    });
    lpmp_ok(lpmp_boundary_create(e, (int64_t)out_off.size(), out_off.data(), out_len.data(), (int64_t)in_off.size(), in_off.data(), in_len.data(),
                                 in_omega.data(), in_order.data(), &bd));
    n_out = lpmp_boundary_out_doubles(bd); n_in = lpmp_boundary_in_doubles(bd);
    hip_ok(hipMalloc((void**)&d_send, (size_t)std::max<int64_t>(n_out, 1) * sizeof(double)), "hipMalloc");
    hip_ok(hipMalloc((void**)&d_back, (size_t)std::max<int64_t>(n_out, 1) * sizeof(double)), "hipMalloc");
    hip_ok(hipMalloc((void**)&d_recv, (size_t)std::max<int64_t>(n_in, 1) * sizeof(double)), "hipMalloc");
    hip_ok(hipMalloc((void**)&d_reply, (size_t)std::max<int64_t>(n_in, 1) * sizeof(double)), "hipMalloc");
    updates_per_pass += 2 * (int64_t)pm.in_unary.size() * (boundary_every_pass ? 1 : 2);
  }
  double local_lower_bound() {
    lpmp_ok(lpmp_invalidate_lower_bounds(e));   // the boundary kernels edit the duals
    double lb = 0; lpmp_ok(lpmp_lower_bound(e, &lb)); return lb;
  }
  std::vector<double> download_duals() {
    std::vector<double> d((size_t)lpmp_dual_size(e)); lpmp_ok(lpmp_download_duals(e, d.data())); return d;
  }

 private:
  struct rows { std::vector<int32_t> f; std::vector<int64_t> om_off{0}, mk_off{0}; std::vector<double> om; std::vector<uint8_t> mk; };
  rows main_rows(lpmp_plan* plan, int d, int mode, int64_t n_vec) const {
    const int64_t nu = lpmp_plan_n_updated(plan, d);
    std::vector<int32_t> upd((size_t)nu);
    lpmp_ok(lpmp_plan_get_update_order(plan, d, upd.data()));
    std::vector<int64_t> oo((size_t)nu + 1), mo((size_t)nu + 1);
    std::vector<double> om((size_t)std::max<int64_t>(lpmp_plan_omega_nnz(plan, d), 1));
    std::vector<uint8_t> mk((size_t)std::max<int64_t>(lpmp_plan_mask_nnz(plan, d), 1));
    lpmp_ok(lpmp_plan_get_omega(plan, d, mode, oo.data(), om.data()));
    lpmp_ok(lpmp_plan_get_mask(plan, d, mode, mo.data(), mk.data()));
    rows r;
    for (int64_t i = 0; i < nu; ++i) {
      if (upd[i] >= pm.n_local && upd[i] < n_vec) continue;   // a ghost: not updated in the main sweeps
      r.f.push_back(upd[i]);
      r.om.insert(r.om.end(), om.begin() + oo[i], om.begin() + oo[i + 1]);
      r.mk.insert(r.mk.end(), mk.begin() + mo[i], mk.begin() + mo[i + 1]);
      r.om_off.push_back((int64_t)r.om.size()); r.mk_off.push_back((int64_t)r.mk.size());
    }
    return r;
  }
  static rows cat(const rows& a, const rows& b) {
    rows r = a;
    r.f.insert(r.f.end(), b.f.begin(), b.f.end());
    for (size_t i = 1; i < b.om_off.size(); ++i) r.om_off.push_back((int64_t)a.om.size() + b.om_off[i]);
    for (size_t i = 1; i < b.mk_off.size(); ++i) r.mk_off.push_back((int64_t)a.mk.size() + b.mk_off[i]);
    r.om.insert(r.om.end(), b.om.begin(), b.om.end());
    r.mk.insert(r.mk.end(), b.mk.begin(), b.mk.end());
    return r;
  }
  int create(const rows& r, bool fuse) {
    int sid = -1;
    lpmp_ok(lpmp_schedule_create_fused(e, (int64_t)r.f.size(), r.f.data(), r.om_off.data(), r.om.data(), r.mk_off.data(), r.mk.data(), fuse ? 1 : 0, &sid));
    return sid;
  }
};

// The order in which a rank issues the point-to-point transfers of one exchange between parts: every (source part, destination
// part) pair that touches this rank, by source, then destination — the same global order on every rank, so the k-th send of rank a
// to rank b meets the k-th receive of b from a whatever the number of parts per rank.  f(src, dst, src_here, dst_here)
template <class F>
inline void for_each_transfer(int n_parts, int rank, int parts_per_rank, F&& f) {
  for (int src = 0; src < n_parts; ++src)
    for (int dst = 0; dst < n_parts; ++dst) {
      const bool src_here = src / parts_per_rank == rank, dst_here = dst / parts_per_rank == rank;
      if (src_here || dst_here) f(src, dst, src_here, dst_here);
    }
}

// One boundary step of all parts of this rank: pack, exchange #1, reply, exchange #2, fold.
// Transfers are issued in a fixed global order — by (source part, destination part) — so that the k-th send of rank a to
// rank b meets the k-th receive of b from a.
inline void exchange(std::vector<part_sweep*>& parts, rccl_world& w, bool first_leg) {
  const int n_parts = parts.empty() ? 0 : parts[0]->pm.n_parts;
  nccl_group grp;      // (closed on every path, also when a transfer below throws)
  for_each_transfer(n_parts, w.rank, w.parts_per_rank, [&](int src, int dst, bool src_here, bool dst_here) {
      // leg 1: what src OWNS toward dst (out lists) travels src -> dst; leg 2: dst's replies travel back dst -> src
      if (first_leg) {
        if (src_here) { part_sweep& p = *parts[src - w.rank * w.parts_per_rank]; const int64_t c = p.out_count[dst];
          if (c > 0) { int64_t off = 0; for (int q = 0; q < dst; ++q) off += p.out_count[q]; nccl_ok(ncclSend(p.d_send + off, (size_t)c, ncclDouble, w.rank_of(dst), w.comm, w.stream), "ncclSend"); } }
        if (dst_here) { part_sweep& p = *parts[dst - w.rank * w.parts_per_rank]; const int64_t c = p.in_count[src];
          if (c > 0) { int64_t off = 0; for (int q = 0; q < src; ++q) off += p.in_count[q]; nccl_ok(ncclRecv(p.d_recv + off, (size_t)c, ncclDouble, w.rank_of(src), w.comm, w.stream), "ncclRecv"); } }
      } else {
        if (dst_here) { part_sweep& p = *parts[dst - w.rank * w.parts_per_rank]; const int64_t c = p.in_count[src];
          if (c > 0) { int64_t off = 0; for (int q = 0; q < src; ++q) off += p.in_count[q]; nccl_ok(ncclSend(p.d_reply + off, (size_t)c, ncclDouble, w.rank_of(src), w.comm, w.stream), "ncclSend"); } }
        if (src_here) { part_sweep& p = *parts[src - w.rank * w.parts_per_rank]; const int64_t c = p.out_count[dst];
          if (c > 0) { int64_t off = 0; for (int q = 0; q < dst; ++q) off += p.out_count[q]; nccl_ok(ncclRecv(p.d_back + off, (size_t)c, ncclDouble, w.rank_of(dst), w.comm, w.stream), "ncclRecv"); } }
      }
    });
  grp.end();
}

inline void boundary_step(std::vector<part_sweep*>& parts, rccl_world& w) {
  w.probe_begin();
  for (part_sweep* p : parts) if (p->pm.n_ghost > 0) { lpmp_ok(lpmp_schedule_run(p->e, p->sid_ghost_recv)); lpmp_ok(lpmp_boundary_pack(p->e, p->bd, p->d_send)); }
  exchange(parts, w, true);
  for (part_sweep* p : parts) if (!p->pm.in_unary.empty()) lpmp_ok(lpmp_boundary_reply(p->e, p->bd, p->d_recv, p->d_reply));
  exchange(parts, w, false);
  for (part_sweep* p : parts) if (p->pm.n_ghost > 0) { lpmp_ok(lpmp_boundary_fold(p->e, p->bd, p->d_back)); lpmp_ok(lpmp_schedule_run(p->e, p->sid_ghost_send)); }
  if (w.probe.on) { int64_t by = 0; for (part_sweep* p : parts) by += 8 * (p->n_out + p->n_in); w.probe_end(by); }
}

// n passes.  boundary_every_pass: [forward + backward main sweeps as one fused schedule, boundary step] x n (what bench.py
// runs on row strips); otherwise [boundary, forward, boundary, backward] x n (PartitionedSweep.program, "sweep")
inline void compute_pass(std::vector<part_sweep*>& parts, rccl_world& w, int n, bool boundary_every_pass) {
  for (int i = 0; i < n; ++i) {
    if (boundary_every_pass) {
      for (part_sweep* p : parts) lpmp_ok(lpmp_schedule_run(p->e, p->sid_fb));
      boundary_step(parts, w);
    } else {
      boundary_step(parts, w);
      for (part_sweep* p : parts) lpmp_ok(lpmp_schedule_run(p->e, p->sid_f));
      boundary_step(parts, w);
      for (part_sweep* p : parts) lpmp_ok(lpmp_schedule_run(p->e, p->sid_b));
    }
  }
}

inline double lower_bound(std::vector<part_sweep*>& parts, rccl_world& w) {
  double lb = 0;
  for (part_sweep* p : parts) lb += p->local_lower_bound();
  return w.all_reduce_sum(lb);
}

}  // namespace lpmp_mgpu
