// lpmp_overlap.hxx — grids on several GPUs as plain joined passes over windows with ghost rows, from C++ over the C ABI
// and RCCL.  The exact multi-GPU schedule for 2-colour grids (lp_mp_amd/overlap.py is the same thing over
// torch.distributed, and tests/test_multi_gpu.py holds the two against each other bit for bit):
//
//   * the global model is the (n_parts * H) x W grid in ONE global colour-major order; part r holds rows
//     [r H - g, (r + 1) H + g) (clipped) as an ordinary model of its own in colour-major order — g even, so the window keeps the
//     global colouring and the relative order of neighbours, hence the global anisotropic weights (LP::ComputeAnisotropicWeights,
//     reference include/LP_MP.h:1232-1415) wherever all neighbours are inside the window;
//   * information moves one grid row per directional sweep: n = g / 2 - 1 calls of lpmp_compute_pass (the single-GPU joined-pass
//     launch) need no exchange, then every part overwrites its ghost rows with the owners' values;
//   * in a colour-major window the rows to ship are CONTIGUOUS ranges of the packed dual array — the black variables of the rows,
//     the white ones, the pairwise factors of the rows (grid edges are numbered row by row) — except the right-hand edges of the
//     one row whose downward edges the receiver does not hold (every other edge: a strided copy).  Pack = 3 + 1 device copies,
//     one ncclSend / ncclRecv per neighbour inside one group, unpack = 4 copies; no arithmetic.
//   Owned rows equal the unpartitioned sweep's bit for bit; the bound is the sum of lpmp_factor_lower_bounds over owned factors.
//
// Needs lpmp_multi_gpu.hxx (rccl_world, error helpers), include/lpmp_engine.h, <rccl/rccl.h>.
#pragma once

#include "lpmp_multi_gpu.hxx"

namespace lpmp_mgpu {

// blacks ((row + col) even) in rows [0, r) of a W-column grid; r may be a global or a window-local row count (windows start
// on even global rows)
inline int64_t blacks_in_rows(int64_t r, int64_t W) { return (r * W + ((W & 1) ? (r & 1) : 0)) / 2; }

struct window_model {
  int32_t part = 0, n_parts = 1, L = 0, H = 0, W = 0, g = 0;
  bool potts = false;
  int64_t r0 = 0, r1 = 0;                  // global rows of the window
  int64_t n_vars = 0, n_edges = 0;
  std::vector<int32_t> f_type, f_dim0, f_dim1, m_type, m_left, m_right, rel, rel_bwd;
  std::vector<uint8_t> f_kind, f_flags;
  lpmp_msg_type mtypes[2];
  uint64_t seed = 1;
  int64_t h() const { return r1 - r0; }
  int64_t g_up() const { return (int64_t)part * H - r0; }                  // ghost rows above the owned ones
  int64_t n_factors() const { return n_vars + n_edges; }
  int64_t const_doubles() const { return potts ? n_edges : n_edges * (int64_t)L * L; }
  int64_t dual_doubles() const { return n_vars * (int64_t)L + n_edges * 2 * (int64_t)L; }
  int64_t nb(int64_t rl) const { return blacks_in_rows(rl, W); }           // local black variables before local row rl
  int64_t nw(int64_t rl) const { return rl * W - nb(rl); }
  int64_t eoff(int64_t rl) const { return rl < h() ? rl * (2 * (int64_t)W - 1) : n_edges; }   // first edge of local row rl (rows before the last are full)
  lpmp_model view(const double* const_dev, const double* dual_dev) const {
    lpmp_model m{};
    m.n_ftypes = 2; m.ftype_computes_primal = nullptr; m.n_mtypes = 2; m.mtypes = mtypes;
    m.n_factors = n_factors(); m.f_type = f_type.data(); m.f_kind = f_kind.data(); m.f_flags = f_flags.data();
    m.f_dim0 = f_dim0.data(); m.f_dim1 = f_dim1.data(); m.const_data = const_dev; m.dual_data = dual_dev;
    m.n_messages = (int64_t)m_type.size(); m.m_type = m_type.data(); m.m_left = m_left.data(); m.m_right = m_right.data();
    m.n_rel_fwd = (int64_t)rel.size() / 2; m.rel_fwd = rel.data();
    m.n_rel_bwd = (int64_t)rel_bwd.size() / 2; m.rel_bwd = rel_bwd.data();
    return m;
  }
};

// overlap.strip_window_part in closed form: the window as a grid MRF in the layout of the reference's MRF constructor
inline window_model strip_window(int H, int W, int L, bool potts, int part, int n_parts, int g, uint64_t seed) {
  if (g % 2 || g < 4) throw std::runtime_error("overlap: the ghost depth must be even and at least 4");
  if (H % 2) throw std::runtime_error("overlap: strips need an even number of rows");
  if (n_parts > 1 && g > H) throw std::runtime_error("overlap: ghost rows reach beyond the neighbouring strip (g <= H)");
  window_model w;
  w.part = part; w.n_parts = n_parts; w.L = L; w.H = H; w.W = W; w.g = g; w.potts = potts; w.seed = seed;
  const int64_t GH = (int64_t)n_parts * H;
  w.r0 = std::max<int64_t>(0, (int64_t)part * H - g); w.r1 = std::min<int64_t>(GH, (int64_t)(part + 1) * H + g);
  const int64_t h = w.h();
  const std::vector<int64_t> var = grid_variable_order((int)h, W, true);
  w.n_vars = h * W;
  std::vector<int64_t> li, lj;
  for (int64_t r = 0; r < h; ++r) for (int c = 0; c < W; ++c) {
    const int64_t a = var[(size_t)(r * W + c)];
    if (c < W - 1) { const int64_t b = var[(size_t)(r * W + c + 1)]; li.push_back(std::min(a, b)); lj.push_back(std::max(a, b)); }
    if (r < h - 1) { const int64_t b = var[(size_t)((r + 1) * W + c)]; li.push_back(std::min(a, b)); lj.push_back(std::max(a, b)); }
  }
  w.n_edges = (int64_t)li.size();
  const int64_t nf = w.n_factors();
  w.f_type.assign((size_t)nf, 0); w.f_kind.assign((size_t)nf, LPMP_F_VECTOR); w.f_flags.assign((size_t)nf, 0);
  w.f_dim0.assign((size_t)nf, L); w.f_dim1.assign((size_t)nf, 0);
  for (int64_t e = 0; e < w.n_edges; ++e) {
    const int64_t f = w.n_vars + e;
    w.f_type[f] = 1; w.f_kind[f] = potts ? LPMP_F_PAIRWISE_POTTS : LPMP_F_PAIRWISE_DENSE; w.f_dim1[f] = L;
    w.m_type.push_back(0); w.m_left.push_back((int32_t)li[e]); w.m_right.push_back((int32_t)f);
    w.m_type.push_back(1); w.m_left.push_back((int32_t)lj[e]); w.m_right.push_back((int32_t)f);
    w.rel.push_back((int32_t)li[e]); w.rel.push_back((int32_t)f);
    w.rel.push_back((int32_t)f); w.rel.push_back((int32_t)lj[e]);
  }
  w.rel_bwd.resize(w.rel.size());
  for (size_t i = 0; i + 1 < w.rel.size(); i += 2) { w.rel_bwd[i] = w.rel[i + 1]; w.rel_bwd[i + 1] = w.rel[i]; }
  w.mtypes[0] = lpmp_msg_type{0, 1, LPMP_SCHED_LEFT, 0, 1, LPMP_M_UNARY_PAIRWISE, 0, 0};
  w.mtypes[1] = lpmp_msg_type{0, 1, LPMP_SCHED_LEFT, 0, 1, LPMP_M_UNARY_PAIRWISE, 1, 0};
  return w;
}

// one window on its engine
class window_sweep {
 public:
  window_model wm;
  lpmp_engine* e = nullptr;
  hipStream_t stream = nullptr;
  double *d_const = nullptr, *d_dual = nullptr, *d_send[2] = {nullptr, nullptr}, *d_recv[2] = {nullptr, nullptr};   // 0: the part above, 1: the part below
  // doubles sent to / received from the neighbour above [0] and below [1].  Not the same both ways: what goes DOWN ends with a
  // full row of edges (the receiver holds the rows below it), what goes UP with the right-hand edges of its last row only
  int64_t n_send[2] = {0, 0}, n_recv[2] = {0, 0};

  window_sweep() = default;
  window_sweep(const window_sweep&) = delete;
  window_sweep& operator=(const window_sweep&) = delete;
  ~window_sweep() {
    if (e) lpmp_destroy(e);
    for (double* p : {d_const, d_dual, d_send[0], d_send[1], d_recv[0], d_recv[1]}) if (p) (void)hipFree(p);
  }

  void build(window_model&& model, int device, hipStream_t s, int mode) {
    wm = std::move(model); stream = s;
    hip_ok(hipSetDevice(device), "hipSetDevice");
    const int64_t L = wm.L, W = wm.W, h = wm.h(), esz = wm.potts ? 1 : L * L;
    const int64_t GH = (int64_t)wm.n_parts * wm.H, n_g = GH * W;
    hip_ok(hipMalloc((void**)&d_const, (size_t)std::max<int64_t>(wm.const_doubles(), 2) * sizeof(double)), "hipMalloc const");
    hip_ok(hipMalloc((void**)&d_dual, (size_t)wm.dual_doubles() * sizeof(double)), "hipMalloc dual");
    hip_ok(hipMemsetAsync(d_dual, 0, (size_t)wm.dual_doubles() * sizeof(double), stream), "hipMemsetAsync");
    // costs from the GLOBAL counter stream (unaries [0, n L) in global variable order, then the pairwise data edge by edge):
    // a window's black variables, its white ones and its full rows of edges are contiguous runs of that stream
    const int64_t NBg = blacks_in_rows(GH, W), nbw = wm.nb(h);
    lpmp_ok(lpmp_synth_fill(d_dual, nbw * L, wm.seed, (uint64_t)(blacks_in_rows(wm.r0, W) * L), stream));
    lpmp_ok(lpmp_synth_fill(d_dual + nbw * L, (wm.n_vars - nbw) * L, wm.seed, (uint64_t)((NBg + wm.r0 * W - blacks_in_rows(wm.r0, W)) * L), stream));
    const int64_t full = (h - 1) * (2 * W - 1);               // edges of the rows before the window's last
    lpmp_ok(lpmp_synth_fill(d_const, full * esz, wm.seed, (uint64_t)(n_g * L + wm.r0 * (2 * W - 1) * esz), stream));
    int64_t* d_first = nullptr;
    if (W > 1) {
      // the last row holds its right-hand edges only: the last row of the grid numbers them consecutively, any other row
      // every other edge (its downward edges lie between them)
      std::vector<int64_t> first((size_t)(W - 1));
      for (int64_t c = 0; c < W - 1; ++c) first[(size_t)c] = n_g * L + ((wm.r1 - 1) * (2 * W - 1) + (wm.r1 == GH ? c : 2 * c)) * esz;
      hip_ok(hipMalloc((void**)&d_first, first.size() * sizeof(int64_t)), "hipMalloc");
      hip_ok(hipMemcpyAsync(d_first, first.data(), first.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
      lpmp_ok(lpmp_synth_fill_blocks(d_const + full * esz, W - 1, esz, wm.seed, d_first, stream));
    }
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (d_first) (void)hipFree(d_first);
    lpmp_ok(lpmp_create(device, &e));
    lpmp_ok(lpmp_set_stream(e, stream));
    const lpmp_model m = wm.view(d_const, d_dual);
    lpmp_ok(lpmp_upload_model(e, &m, LPMP_MEM_DEVICE, LPMP_MEM_DEVICE));
    lpmp_ok(lpmp_set_reparametrization(e, mode));
    // exchange buffers
    for (int d = 0; d < 2; ++d) {
      const bool have = d == 0 ? wm.part > 0 : wm.part < wm.n_parts - 1;
      if (!have) continue;
      for (const piece& p : pieces(d, 0)) n_send[d] += p.stride == 0 ? p.n : p.n * (W - 1);
      for (const piece& p : pieces(d, 1)) n_recv[d] += p.stride == 0 ? p.n : p.n * (W - 1);
      hip_ok(hipMalloc((void**)&d_send[d], (size_t)std::max<int64_t>(n_send[d], 1) * sizeof(double)), "hipMalloc");
      hip_ok(hipMalloc((void**)&d_recv[d], (size_t)std::max<int64_t>(n_recv[d], 1) * sizeof(double)), "hipMalloc");
    }
  }

  // the ranges of the packed dual array one exchange moves.  what 0: the rows I own that the neighbour on side d holds
  // (source of pack); what 1: the ghost rows on side d (target of unpack).  Every piece is {offset, doubles, stride}: stride 0 =
  // contiguous, else `doubles` is one block of 2 L doubles repeated W - 1 times every `stride` doubles
  struct piece { int64_t off, n, stride; };
  std::vector<piece> pieces(int d, int what) const {
    const int64_t L = wm.L, W = wm.W, gu = wm.g_up(), H = wm.H, g = wm.g, h = wm.h();
    const int64_t NB = wm.nb(h), pw = wm.n_vars * L;
    // variable rows [a, b) and edge rows: full rows [a, b - 1) + the right-hand edges of row b - 1 when the RECEIVER's window ends
    // there (its last row: no downward edges), all rows [a, b) otherwise
    int64_t a, b; bool last_row_right_only;
    if (what == 0) { a = d == 0 ? gu : gu + H - g; b = a + g; last_row_right_only = d == 0; }   // sent up: the upper part's window ends at my row b - 1
    else { a = d == 0 ? 0 : gu + H; b = a + g; last_row_right_only = d == 1; }                  // ghost rows below: my own window ends there
    std::vector<piece> p;
    p.push_back({wm.nb(a) * L, (wm.nb(b) - wm.nb(a)) * L, 0});
    p.push_back({(NB + wm.nw(a)) * L, (wm.nw(b) - wm.nw(a)) * L, 0});
    if (!last_row_right_only) { p.push_back({pw + wm.eoff(a) * 2 * L, (wm.eoff(b) - wm.eoff(a)) * 2 * L, 0}); return p; }
    p.push_back({pw + wm.eoff(a) * 2 * L, (wm.eoff(b - 1) - wm.eoff(a)) * 2 * L, 0});
    // row b - 1: in MY numbering its right-hand edges are consecutive if it is my last row, else every other edge
    if (W > 1) p.push_back({pw + wm.eoff(b - 1) * 2 * L, 2 * L, b - 1 == h - 1 ? 2 * L : 4 * L});
    return p;
  }
  void pack(int d) {
    int64_t at = 0;
    for (const piece& p : pieces(d, 0)) at += copy(d_send[d] + at, d_dual + p.off, p, false);
    if (at != n_send[d]) throw std::runtime_error("overlap: pack size");
  }
  void unpack(int d) {
    int64_t at = 0;
    for (const piece& p : pieces(d, 1)) at += copy(d_dual + p.off, d_recv[d] + at, p, true);
    if (at != n_recv[d]) throw std::runtime_error("overlap: unpack size");
  }
  double local_lower_bound() {
    lpmp_ok(lpmp_invalidate_lower_bounds(e));   // the exchange edits the duals behind the engine's back
    std::vector<double> flb((size_t)wm.n_factors());
    lpmp_ok(lpmp_factor_lower_bounds(e, flb.data()));
    const int64_t a = wm.g_up(), b = a + wm.H, NB = wm.nb(wm.h());
    double lb = 0;
    for (int64_t v = wm.nb(a); v < wm.nb(b); ++v) lb += flb[(size_t)v];
    for (int64_t v = NB + wm.nw(a); v < NB + wm.nw(b); ++v) lb += flb[(size_t)v];
    for (int64_t ed = wm.eoff(a); ed < wm.eoff(b); ++ed) lb += flb[(size_t)(wm.n_vars + ed)];   // a pairwise factor lives with its upper / left endpoint
    return lb;
  }
  std::vector<double> download_duals() {
    std::vector<double> d((size_t)lpmp_dual_size(e)); lpmp_ok(lpmp_download_duals(e, d.data())); return d;
  }

 private:
  // dense side <- / -> strided side; returns the doubles moved.  to_strided: dst is the packed dual array
  int64_t copy(double* dst, const double* src, const piece& p, bool to_strided) {
    if (p.n <= 0) return 0;
    if (p.stride == 0) { hip_ok(hipMemcpyAsync(dst, src, (size_t)p.n * sizeof(double), hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync"); return p.n; }
    const size_t width = (size_t)p.n * sizeof(double), rows = (size_t)(wm.W - 1);
    hip_ok(hipMemcpy2DAsync(dst, to_strided ? (size_t)p.stride * sizeof(double) : width, src, to_strided ? width : (size_t)p.stride * sizeof(double), width, rows,
                            hipMemcpyDeviceToDevice, stream), "hipMemcpy2DAsync");
    return p.n * (int64_t)rows;
  }
};

// owners' values into every part's ghost rows: one ncclSend / ncclRecv per pair of neighbouring parts, all in one group
inline void overlap_exchange(std::vector<window_sweep*>& parts, rccl_world& w) {
  w.probe_begin();
  for (window_sweep* p : parts) for (int d = 0; d < 2; ++d) if (p->n_send[d] > 0) p->pack(d);
  const int n_parts = parts.empty() ? 0 : parts[0]->wm.n_parts;
  for (int q = 0; q + 1 < n_parts; ++q)              // neighbouring windows of this rank must agree: checked before the group is opened
    if (w.rank_of(q) == w.rank && w.rank_of(q + 1) == w.rank) {
      window_sweep &up = *parts[(size_t)(q - w.rank * w.parts_per_rank)], &down = *parts[(size_t)(q + 1 - w.rank * w.parts_per_rank)];
      if (up.n_send[1] != down.n_recv[0] || down.n_send[0] != up.n_recv[1]) throw std::runtime_error("overlap: neighbouring windows disagree on the exchange");
    }
  nccl_group grp;      // (closed on every path, also when a transfer below throws)
  for (int q = 0; q + 1 < n_parts; ++q) {          // the pair (q, q + 1), both directions; fixed order: sends and receives of two ranks match
    const bool up_here = w.rank_of(q) == w.rank, down_here = w.rank_of(q + 1) == w.rank;
    if (up_here && down_here) {
      // both parts on this rank: buffer to buffer on the stream (sends of a rank to itself pair up in issue order, which is not
      // the order of this loop's receives)
      window_sweep &up = *parts[(size_t)(q - w.rank * w.parts_per_rank)], &down = *parts[(size_t)(q + 1 - w.rank * w.parts_per_rank)];
      hip_ok(hipMemcpyAsync(down.d_recv[0], up.d_send[1], (size_t)up.n_send[1] * sizeof(double), hipMemcpyDeviceToDevice, w.stream), "hipMemcpyAsync");
      hip_ok(hipMemcpyAsync(up.d_recv[1], down.d_send[0], (size_t)down.n_send[0] * sizeof(double), hipMemcpyDeviceToDevice, w.stream), "hipMemcpyAsync");
      continue;
    }
    if (up_here) {
      window_sweep& p = *parts[(size_t)(q - w.rank * w.parts_per_rank)];
      nccl_ok(ncclSend(p.d_send[1], (size_t)p.n_send[1], ncclDouble, w.rank_of(q + 1), w.comm, w.stream), "ncclSend");
      nccl_ok(ncclRecv(p.d_recv[1], (size_t)p.n_recv[1], ncclDouble, w.rank_of(q + 1), w.comm, w.stream), "ncclRecv");
    }
    if (down_here) {
      window_sweep& p = *parts[(size_t)(q + 1 - w.rank * w.parts_per_rank)];
      nccl_ok(ncclSend(p.d_send[0], (size_t)p.n_send[0], ncclDouble, w.rank_of(q), w.comm, w.stream), "ncclSend");
      nccl_ok(ncclRecv(p.d_recv[0], (size_t)p.n_recv[0], ncclDouble, w.rank_of(q), w.comm, w.stream), "ncclRecv");
    }
  }
  grp.end();
  for (window_sweep* p : parts) for (int d = 0; d < 2; ++d) if (p->n_recv[d] > 0) p->unpack(d);
  if (w.probe.on) { int64_t by = 0; for (window_sweep* p : parts) by += 8 * (p->n_send[0] + p->n_send[1]); w.probe_end(by); }
}

// n passes in chunks of at most g / 2 - 1 (chunk <= 0: that maximum), an exchange behind every chunk
inline void overlap_compute_pass(std::vector<window_sweep*>& parts, rccl_world& w, int n, int chunk = 0) {
  if (parts.empty()) return;
  const int max_chunk = (parts[0]->wm.g - 2) / 2;
  if (chunk <= 0) chunk = max_chunk;
  if (parts[0]->wm.n_parts > 1 && chunk > max_chunk) throw std::runtime_error("overlap: more passes between exchanges than the ghost rows allow (2 n + 2 rows for n passes)");
  if (parts[0]->wm.n_parts == 1) { lpmp_ok(lpmp_compute_pass(parts[0]->e, n)); return; }
  while (n > 0) {
    const int k = std::min(n, chunk);
    for (window_sweep* p : parts) lpmp_ok(lpmp_compute_pass(p->e, k));
    overlap_exchange(parts, w);
    n -= k;
  }
}

inline double overlap_lower_bound(std::vector<window_sweep*>& parts, rccl_world& w) {
  double lb = 0;
  for (window_sweep* p : parts) lb += p->local_lower_bound();
  return w.all_reduce_sum(lb);
}

}  // namespace lpmp_mgpu
