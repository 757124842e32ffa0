// lpmp_lockstep.hxx — the lock-step partitioned sweep driven from C++ over the C ABI and RCCL (the same schedule
// lp_mp_amd/lockstep.py runs through torch.distributed, bit for bit).
//
// Several parts execute THE unpartitioned sweep of a model (LP::ComputePass, reference include/LP_MP.h:981-1005): updates of one
// dependency level commute, so every part runs the updates of ITS variables with the GLOBAL weights (LP::get_omega, :412-460,
// from lpmp_plan_create on the structure of the whole model) level by level, and between two runs of levels the parts ship the
// message vectors the next run reads across the cut.  Side s of a pairwise factor's dual is written only by the updates of
// endpoint s; both parts of a cut edge hold the pairwise factor and a never-updated ghost of the remote variable.  A labeling-list
// factor's dual is rewritten as a whole by each of its messages: the whole vector is one exchange unit and goes to every other part
// holding the factor (lockstep_plan::build(lpmp_model, ...), lockstep_part::build(lpmp_model, ...): any `left`-schedule model).
//
//   lockstep_structure   an MRF as an edge list + the partition (strips_structure, graph_structure); model_file: any model with its costs
//   lockstep_plan        what every rank derives identically from it: levels, who writes / reads which vector when, the steps of n
//                        passes (runs of sub-levels and exchanges; an exchange ships only what is read before the next one)
//   lockstep_part        one part: its variables + ghosts + every edge touching a variable, costs generated in HBM from the counter
//                        stream, its rows of the global weights per sub-level; runs as lpmp_schedule_create[_fused] / _run,
//                        ships through lpmp_halo_pack / _unpack and ncclSend / ncclRecv (parts of one rank: device copies)
//
// Needs lpmp_multi_gpu.hxx (rccl_world, helpers); link with -llpmp_engine -lrccl -lamdhip64.
#pragma once

#include <map>

#include "lpmp_multi_gpu.hxx"

namespace lpmp_mgpu {

struct lockstep_structure {
  int64_t n_vars = 0; int32_t L = 0; bool potts = false; int n_parts = 1;
  std::vector<int64_t> ei, ej;          // edge e joins ei[e] < ej[e]; message 2 e + s: endpoint s
  std::vector<int32_t> part;            // part of every variable
  uint64_t seed = 1;                    // costs: unaries u01 stream [0, n L), pairwise data of edge e at n L + e * (L^2 or 1)
  int64_t n_edges() const { return (int64_t)ei.size(); }
};

// the (n_parts * H) x W strip grid of multi_gpu.strip_global_edges: strip-major variables, per strip its internal edges (node by
// node: right, then down) and then the W edges to the next strip
inline lockstep_structure strips_structure(int H, int W, int L, bool potts, bool colour_major, int n_parts, uint64_t seed) {
  lockstep_structure s; s.L = L; s.potts = potts; s.n_parts = n_parts; s.seed = seed;
  const int64_t n_loc = (int64_t)H * W;
  s.n_vars = n_loc * n_parts;
  const std::vector<int64_t> var = grid_variable_order(H, W, colour_major);
  for (int k = 0; k < n_parts; ++k) {
    const int64_t base = (int64_t)k * n_loc;
    for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) {
      const int64_t a = base + var[(size_t)r * W + c];
      if (c < W - 1) { const int64_t b = base + var[(size_t)r * W + c + 1]; s.ei.push_back(std::min(a, b)); s.ej.push_back(std::max(a, b)); }
      if (r < H - 1) { const int64_t b = base + var[(size_t)(r + 1) * W + c]; s.ei.push_back(std::min(a, b)); s.ej.push_back(std::max(a, b)); }
    }
    if (k < n_parts - 1) for (int c = 0; c < W; ++c) { s.ei.push_back(base + var[(size_t)(H - 1) * W + c]); s.ej.push_back(base + n_loc + var[(size_t)c]); }
  }
  s.part.resize((size_t)s.n_vars);
  for (int64_t v = 0; v < s.n_vars; ++v) s.part[(size_t)v] = (int32_t)(v / n_loc);
  return s;
}

// the C4-style random graph of synthetic.counter_graph_edges: edge e joins a = h(2 e) mod n and b = a + 1 + h(2 e + 1) mod (n - 1),
// h = the 64-bit words of the counter stream seeded with seed ^ 0x5DEECE66D; variables split into n_parts contiguous index ranges
// (a caller with a partitioner of its own overwrites `part`)
inline uint64_t counter_u64(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// var_rank (optional, [n]): the variables renamed — variable v becomes var_rank[v] (synthetic.counter_graph_model(..., rank=...);
// e.g. a colour-major order: 9 dependent levels per sweep on the C4 shape instead of 33); part_of (optional, [n], by NEW index)
inline lockstep_structure graph_structure(int64_t n, int64_t m, int L, int n_parts, uint64_t seed, const std::vector<int64_t>* var_rank = nullptr,
                                          const std::vector<int64_t>* part_of = nullptr) {
  lockstep_structure s; s.n_vars = n; s.L = L; s.potts = false; s.n_parts = n_parts; s.seed = seed;
  if ((var_rank && (int64_t)var_rank->size() != n) || (part_of && (int64_t)part_of->size() != n)) throw std::runtime_error("lockstep: one entry per variable in the order / partition");
  s.ei.resize((size_t)m); s.ej.resize((size_t)m);
  for (int64_t e = 0; e < m; ++e) {
    int64_t a = (int64_t)(counter_u64(seed ^ 0x5DEECE66DULL, (uint64_t)(2 * e)) % (uint64_t)n);
    int64_t b = (a + 1 + (int64_t)(counter_u64(seed ^ 0x5DEECE66DULL, (uint64_t)(2 * e + 1)) % (uint64_t)(n - 1))) % n;
    if (var_rank) { a = (*var_rank)[(size_t)a]; b = (*var_rank)[(size_t)b]; if (a < 0 || a >= n || b < 0 || b >= n) throw std::runtime_error("lockstep: variable order out of range"); }
    s.ei[(size_t)e] = std::min(a, b); s.ej[(size_t)e] = std::max(a, b);
  }
  s.part.resize((size_t)n);
  for (int64_t v = 0; v < n; ++v) s.part[(size_t)v] = part_of ? (int32_t)(*part_of)[(size_t)v] : (int32_t)(v * n_parts / n);
  return s;
}

// an MRF as flat model arrays (the layout of the reference's MRF constructor, as part_model)
struct mrf_arrays {
  std::vector<int32_t> f_type, f_dim0, f_dim1, m_type, m_left, m_right, rel, rel_bwd;
  std::vector<uint8_t> f_kind, f_flags;
  lpmp_msg_type mtypes[2];
  void build(int64_t n_vars, int32_t L, bool potts, const std::vector<int64_t>& li, const std::vector<int64_t>& lj) {
    const int64_t ne = (int64_t)li.size(), nf = n_vars + ne;
    f_type.assign((size_t)nf, 0); f_kind.assign((size_t)nf, LPMP_F_VECTOR); f_flags.assign((size_t)nf, 0);
    f_dim0.assign((size_t)nf, L); f_dim1.assign((size_t)nf, 0);
    m_type.resize((size_t)2 * ne); m_left.resize((size_t)2 * ne); m_right.resize((size_t)2 * ne); rel.resize((size_t)4 * ne);
    for (int64_t e = 0; e < ne; ++e) {
      const int64_t f = n_vars + e;
      f_type[f] = 1; f_kind[f] = potts ? LPMP_F_PAIRWISE_POTTS : LPMP_F_PAIRWISE_DENSE; f_dim1[f] = L;
      m_type[2 * e] = 0; m_left[2 * e] = (int32_t)li[e]; m_right[2 * e] = (int32_t)f;
      m_type[2 * e + 1] = 1; m_left[2 * e + 1] = (int32_t)lj[e]; m_right[2 * e + 1] = (int32_t)f;
      rel[4 * e] = (int32_t)li[e]; rel[4 * e + 1] = (int32_t)f; rel[4 * e + 2] = (int32_t)f; rel[4 * e + 3] = (int32_t)lj[e];
    }
    rel_bwd.resize(rel.size());
    for (size_t i = 0; i + 1 < rel.size(); i += 2) { rel_bwd[i] = rel[i + 1]; rel_bwd[i + 1] = rel[i]; }
    mtypes[0] = lpmp_msg_type{0, 1, LPMP_SCHED_LEFT, 0, 1, LPMP_M_UNARY_PAIRWISE, 0, 0};
    mtypes[1] = lpmp_msg_type{0, 1, LPMP_SCHED_LEFT, 0, 1, LPMP_M_UNARY_PAIRWISE, 1, 0};
  }
  lpmp_model view(const double* const_dev, const double* dual_dev) const {
    lpmp_model m{};
    m.n_ftypes = 2; m.n_mtypes = 2; m.mtypes = mtypes;
    m.n_factors = (int64_t)f_type.size(); m.f_type = f_type.data(); m.f_kind = f_kind.data(); m.f_flags = f_flags.data();
    m.f_dim0 = f_dim0.data(); m.f_dim1 = f_dim1.data(); m.const_data = const_dev; m.dual_data = dual_dev;
    m.n_messages = (int64_t)m_type.size(); m.m_type = m_type.data(); m.m_left = m_left.data(); m.m_right = m_right.data();
    m.n_rel_fwd = (int64_t)rel.size() / 2; m.rel_fwd = rel.data(); m.n_rel_bwd = (int64_t)rel_bwd.size() / 2; m.rel_bwd = rel_bwd.data();
    return m;
  }
};

// a whole model read from a flat binary file (the arrays of lpmp_model.h in their order; written by lp_mp_amd.model.FlatModel.dump):
//   int64 magic, n_ftypes, n_mtypes, n_tables, tab_total, n_factors, n_messages, n_rel_fwd, n_rel_bwd, n_const, n_dual; double constant;
//   u8 ftype_computes_primal[n_ftypes]; int32 mtypes[n_mtypes][8]; int64 tab_off[n_tables + 1]; int32 tab_data[tab_total]; int32 tab_nleft[n_tables];
//   int32 f_type[nf]; u8 f_kind[nf]; u8 f_flags[nf]; int32 f_dim0[nf]; int32 f_dim1[nf]; double const[n_const]; double dual[n_dual];
//   int32 m_type[nm], m_left[nm], m_right[nm]; int32 rel_fwd[n_rel_fwd][2]; int32 rel_bwd[n_rel_bwd][2]
struct model_file {
  static constexpr int64_t MAGIC = 0x4C504D504D4F444CLL;   // "LPMPMODL"
  std::vector<uint8_t> primal, f_kind, f_flags;
  std::vector<lpmp_msg_type> mtypes;
  std::vector<int64_t> tab_off;
  std::vector<int32_t> tab_data, tab_nleft, f_type, f_dim0, f_dim1, m_type, m_left, m_right, rel_fwd, rel_bwd;
  std::vector<double> cdata, ddata;
  double constant = 0;
  void load(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    struct closer { FILE* f; ~closer() { std::fclose(f); } } cl{f};
    auto rd = [&](void* p, size_t bytes) { if (bytes && std::fread(p, 1, bytes, f) != bytes) throw std::runtime_error("model file " + path + " is truncated"); };
    int64_t h[11];
    rd(h, sizeof(h)); rd(&constant, sizeof(double));
    if (h[0] != MAGIC) throw std::runtime_error(path + " is not a model file");
    for (int i = 1; i < 11; ++i) if (h[i] < 0) throw std::runtime_error("model file: negative count");
    auto get = [&](auto& v, int64_t n) { v.resize((size_t)n); rd(v.data(), (size_t)n * sizeof(v[0])); };
    get(primal, h[1]); get(mtypes, h[2]); get(tab_off, h[3] + 1); get(tab_data, h[4]); get(tab_nleft, h[3]);
    get(f_type, h[5]); get(f_kind, h[5]); get(f_flags, h[5]); get(f_dim0, h[5]); get(f_dim1, h[5]); get(cdata, h[9]); get(ddata, h[10]);
    get(m_type, h[6]); get(m_left, h[6]); get(m_right, h[6]); get(rel_fwd, 2 * h[7]); get(rel_bwd, 2 * h[8]);
  }
  lpmp_model view() const {
    lpmp_model m{};
    m.n_ftypes = (int32_t)primal.size(); m.ftype_computes_primal = primal.data(); m.n_mtypes = (int32_t)mtypes.size(); m.mtypes = mtypes.data();
    m.n_tables = (int32_t)tab_nleft.size(); m.tab_off = tab_off.data(); m.tab_data = tab_data.data(); m.tab_nleft = tab_nleft.data();
    m.n_factors = (int64_t)f_type.size(); m.f_type = f_type.data(); m.f_kind = f_kind.data(); m.f_flags = f_flags.data(); m.f_dim0 = f_dim0.data(); m.f_dim1 = f_dim1.data();
    m.const_data = cdata.empty() ? nullptr : cdata.data(); m.dual_data = ddata.data();
    m.n_messages = (int64_t)m_type.size(); m.m_type = m_type.data(); m.m_left = m_left.data(); m.m_right = m_right.data();
    m.n_rel_fwd = (int64_t)rel_fwd.size() / 2; m.rel_fwd = rel_fwd.data(); m.n_rel_bwd = (int64_t)rel_bwd.size() / 2; m.rel_bwd = rel_bwd.data();
    m.constant = constant;
    return m;
  }
};

struct lockstep_step { bool halo = false; std::vector<std::pair<int, int>> run; std::vector<int64_t> vecs; int id = -1; };

// what every rank computes identically from the global structure
struct lockstep_plan {
  int n_levels[2] = {0, 0};
  int64_t n_vecs = 0;                                                // one exchange unit per message (an MRF: 2 e + s)
  std::vector<int32_t> writer;                                       // [n_vecs] part of the message's left factor: the only writer of that slice
  std::vector<int64_t> dest_off; std::vector<int32_t> dest;         // CSR: the other parts holding the vector's factor
  std::vector<int64_t> v_off, v_len;                                 // the slice of its higher factor's dual a message writes
  std::vector<int32_t> owner;                                        // [n_factors] where a factor counts in the bound
  std::vector<std::vector<int64_t>> written[2], read[2];           // [direction][sub-level]: sorted cut vectors
  // the rows of the global sweep: update order, weights, masks, level, "touches a cut vector"
  std::vector<int32_t> upd[2], lev[2]; std::vector<int64_t> om_off[2], mk_off[2]; std::vector<double> om[2]; std::vector<uint8_t> mk[2], touches_cut[2];
  std::map<int, std::vector<lockstep_step>> programs;
  std::map<std::vector<int64_t>, int> halo_ids;
  bool is_cut(int64_t v) const { return dest_off[(size_t)v + 1] > dest_off[(size_t)v]; }

  void build(const lockstep_structure& s, int mode) {
    mrf_arrays g; g.build(s.n_vars, s.L, s.potts, s.ei, s.ej);
    std::vector<int32_t> part(s.part);
    part.resize((size_t)(s.n_vars + s.n_edges()), 0);
    build(g.view(nullptr, nullptr), part, s.n_parts, mode);
  }

  // any model whose messages all have the `left` schedule and whose factors are either variables (left factor of their messages,
  // or no message) or higher factors (right factor), with unary-pairwise or labeling messages (lockstep.lockstep_model).
  // part[f]: part of variable f (ignored for higher factors).  Costs are not read.
  void build(const lpmp_model& gm, const std::vector<int32_t>& part, int n_parts, int mode) {
    const int64_t nf = gm.n_factors, nm = gm.n_messages;
    if ((int64_t)part.size() != nf) throw std::runtime_error("lockstep: a part for every factor (used for the variables)");
    for (int t = 0; t < gm.n_mtypes; ++t)
      if (gm.mtypes[t].schedule != LPMP_SCHED_LEFT || (gm.mtypes[t].kind != LPMP_M_UNARY_PAIRWISE && gm.mtypes[t].kind != LPMP_M_LABELING))
        throw std::runtime_error("lockstep: only `left`-schedule unary-pairwise / labeling messages");
    std::vector<uint8_t> is_right((size_t)nf, 0), is_left((size_t)nf, 0);
    for (int64_t k = 0; k < nm; ++k) { is_right[(size_t)gm.m_right[k]] = 1; is_left[(size_t)gm.m_left[k]] = 1; }
    std::vector<int64_t> cnt((size_t)n_parts, 0);
    for (int64_t f = 0; f < nf; ++f) {
      if (is_left[(size_t)f] && is_right[(size_t)f]) throw std::runtime_error("lockstep: a factor is both left and right of messages");
      if (is_right[(size_t)f]) continue;
      if (part[(size_t)f] < 0 || part[(size_t)f] >= n_parts) throw std::runtime_error("lockstep: part out of range");
      ++cnt[(size_t)part[(size_t)f]];
    }
    for (int64_t c : cnt) if (c == 0) throw std::runtime_error("lockstep: a part without variables");
    lpmp_plan* pl = nullptr;
    lpmp_ok(lpmp_plan_create(&gm, &pl));
    struct guard { lpmp_plan* p; ~guard() { lpmp_plan_destroy(p); } } gd{pl};
    std::vector<int64_t> g_off((size_t)nf + 1), g_ent((size_t)2 * nm);
    lpmp_ok(lpmp_plan_get_msg_lists(pl, g_off.data(), g_ent.data()));
    n_vecs = nm;
    writer.resize((size_t)nm); v_off.resize((size_t)nm); v_len.resize((size_t)nm);
    std::vector<uint8_t> whole((size_t)nm);
    auto dual_size = [&](int64_t f) -> int64_t {
      const int64_t d0 = gm.f_dim0[f], d1 = gm.f_dim1 ? gm.f_dim1[f] : 0;
      return gm.f_kind[f] == LPMP_F_PAIRWISE_DENSE ? d0 + d1 : gm.f_kind[f] == LPMP_F_PAIRWISE_POTTS ? 2 * d0 : d0;
    };
    for (int64_t k = 0; k < nm; ++k) {
      const lpmp_msg_type& mt = gm.mtypes[gm.m_type[k]];
      const int64_t r = gm.m_right[k], d0 = gm.f_dim0[r], d1 = gm.f_kind[r] == LPMP_F_PAIRWISE_POTTS ? d0 : (gm.f_dim1 ? gm.f_dim1[r] : 0);
      writer[(size_t)k] = part[(size_t)gm.m_left[k]];
      if (mt.kind == LPMP_M_UNARY_PAIRWISE) { v_off[(size_t)k] = mt.param == 1 ? d0 : 0; v_len[(size_t)k] = mt.param == 1 ? d1 : d0; whole[(size_t)k] = 0; }
      else { v_off[(size_t)k] = 0; v_len[(size_t)k] = dual_size(r); whole[(size_t)k] = 1; }   // a labeling message rewrites all of its factor's dual
    }
    // the messages of every higher factor (siblings), the parts holding it, where it counts in the bound
    std::vector<int64_t> sib_off((size_t)nf + 1, 0), sib((size_t)nm);
    for (int64_t k = 0; k < nm; ++k) ++sib_off[(size_t)gm.m_right[k] + 1];
    for (int64_t f = 0; f < nf; ++f) sib_off[(size_t)f + 1] += sib_off[(size_t)f];
    { std::vector<int64_t> cur(sib_off.begin(), sib_off.end() - 1); for (int64_t k = 0; k < nm; ++k) sib[(size_t)cur[(size_t)gm.m_right[k]]++] = k; }
    owner.assign(part.begin(), part.end());
    dest_off.assign((size_t)nm + 1, 0);
    std::vector<std::vector<int32_t>> dests((size_t)nm);
    for (int64_t f = 0; f < nf; ++f) {
      if (!is_right[(size_t)f]) continue;
      std::vector<int32_t> hold; int64_t first_var = nf;
      for (int64_t j = sib_off[(size_t)f]; j < sib_off[(size_t)f + 1]; ++j) { hold.push_back(writer[(size_t)sib[(size_t)j]]); first_var = std::min<int64_t>(first_var, gm.m_left[sib[(size_t)j]]); }
      std::sort(hold.begin(), hold.end()); hold.erase(std::unique(hold.begin(), hold.end()), hold.end());
      owner[(size_t)f] = part[(size_t)first_var];
      for (int64_t j = sib_off[(size_t)f]; j < sib_off[(size_t)f + 1]; ++j) {
        const int64_t k = sib[(size_t)j];
        for (int32_t q : hold) if (q != writer[(size_t)k]) dests[(size_t)k].push_back(q);
      }
    }
    for (int64_t k = 0; k < nm; ++k) dest_off[(size_t)k + 1] = dest_off[(size_t)k] + (int64_t)dests[(size_t)k].size();
    dest.resize((size_t)dest_off[(size_t)nm]);
    for (int64_t k = 0; k < nm; ++k) std::copy(dests[(size_t)k].begin(), dests[(size_t)k].end(), dest.begin() + dest_off[(size_t)k]);
    for (int d = 0; d < 2; ++d) {
      const int64_t nu = lpmp_plan_n_updated(pl, d);
      upd[d].resize((size_t)nu); lev[d].resize((size_t)nu);
      lpmp_ok(lpmp_plan_get_update_order(pl, d, upd[d].data()));
      om_off[d].resize((size_t)nu + 1); mk_off[d].resize((size_t)nu + 1);
      om[d].resize((size_t)std::max<int64_t>(lpmp_plan_omega_nnz(pl, d), 1)); mk[d].resize((size_t)std::max<int64_t>(lpmp_plan_mask_nnz(pl, d), 1));
      lpmp_ok(lpmp_plan_get_omega(pl, d, mode, om_off[d].data(), om[d].data()));
      lpmp_ok(lpmp_plan_get_mask(pl, d, mode, mk_off[d].data(), mk[d].data()));
      lpmp_ok(lpmp_plan_get_update_levels(pl, d, mode, lev[d].data()));
      int nl = 0;
      for (int64_t u = 0; u < nu; ++u) {
        if (is_right[(size_t)upd[d][(size_t)u]]) throw std::runtime_error("lockstep: only the variables are updated (schedule `left`)");
        lev[d][(size_t)u] = std::max(lev[d][(size_t)u], 1);       // (0: no active message; runs with the first level)
        nl = std::max(nl, (int)lev[d][(size_t)u]);
        const int64_t f = upd[d][(size_t)u], len = g_off[(size_t)f + 1] - g_off[(size_t)f];
        if (len != om_off[d][(size_t)u + 1] - om_off[d][(size_t)u] || len != mk_off[d][(size_t)u + 1] - mk_off[d][(size_t)u])
          throw std::runtime_error("lockstep: a variable's message list, weights and mask differ in length");
      }
      n_levels[d] = nl;
      written[d].assign((size_t)2 * nl, {}); read[d].assign((size_t)2 * nl, {});
      touches_cut[d].assign((size_t)nu, 0);
      for (int64_t u = 0; u < nu; ++u) {
        const int64_t f = upd[d][(size_t)u];
        const int sl = 2 * (lev[d][(size_t)u] - 1);
        for (int64_t j = g_off[(size_t)f], x = 0; j < g_off[(size_t)f + 1]; ++j, ++x) {
          const int64_t k = g_ent[(size_t)j] / 2;                  // the entry's message = its own vector
          const double w = om[d][(size_t)(om_off[d][(size_t)u] + x)]; const bool r = mk[d][(size_t)(mk_off[d][(size_t)u] + x)] != 0;
          const bool writes = w != 0.0 || r;
          if (writes && is_cut(k)) { written[d][(size_t)sl].push_back(k); touches_cut[d][(size_t)u] = 1; }
          if (r || (writes && whole[(size_t)k])) {                 // what it reads across the cut: its factor's other messages written elsewhere
            const int64_t rf = gm.m_right[k];
            for (int64_t jj = sib_off[(size_t)rf]; jj < sib_off[(size_t)rf + 1]; ++jj) { const int64_t k2 = sib[(size_t)jj]; if (writer[(size_t)k2] != writer[(size_t)k]) read[d][(size_t)sl].push_back(k2); }
          }
        }
      }
      for (auto* lists : {&written[d], &read[d]}) for (auto& l : *lists) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
    }
  }

  // steps of n passes (LockstepSchedule.program): sub-level 2 l = the records of level l + 1 that touch a cut edge, 2 l + 1 = the
  // others.  An exchange comes before the first sub-level that reads a cut vector written since the last one, moved back over the
  // sub-levels that wrote nothing; it ships what is read before the next exchange
  const std::vector<lockstep_step>& program(int n_passes) {
    auto it = programs.find(n_passes);
    if (it != programs.end()) return it->second;
    std::vector<std::pair<int, int>> seq;
    for (int p = 0; p < n_passes; ++p) for (int d = 0; d < 2; ++d) for (int sl = 0; sl < 2 * n_levels[d]; ++sl) seq.push_back({d, sl});
    std::vector<uint8_t> dirty((size_t)n_vecs, 0);
    std::vector<int64_t> dirty_list;
    auto mark = [&](const std::vector<int64_t>& w) { for (int64_t v : w) if (!dirty[(size_t)v]) { dirty[(size_t)v] = 1; dirty_list.push_back(v); } };
    auto clear = [&] { for (int64_t v : dirty_list) dirty[(size_t)v] = 0; dirty_list.clear(); };
    std::vector<int64_t> where; int64_t last = 0;
    for (int64_t i = 0; i < (int64_t)seq.size(); ++i) {
      bool hit = false;
      for (int64_t v : read[seq[(size_t)i].first][(size_t)seq[(size_t)i].second]) if (dirty[(size_t)v]) { hit = true; break; }
      if (hit) {
        int64_t pos = i;
        while (pos - 1 >= last && written[seq[(size_t)pos - 1].first][(size_t)seq[(size_t)pos - 1].second].empty()) --pos;
        where.push_back(pos); clear(); last = pos;
      }
      mark(written[seq[(size_t)i].first][(size_t)seq[(size_t)i].second]);
    }
    clear();
    std::vector<std::pair<int64_t, std::vector<int64_t>>> halos;
    int64_t start = 0;
    for (size_t k = 0; k < where.size(); ++k) {
      const int64_t pos = where[k], nxt = k + 1 < where.size() ? where[k + 1] : (int64_t)seq.size();
      for (int64_t i = start; i < pos; ++i) mark(written[seq[(size_t)i].first][(size_t)seq[(size_t)i].second]);
      std::vector<int64_t> ship;
      for (int64_t i = pos; i < nxt; ++i) for (int64_t v : read[seq[(size_t)i].first][(size_t)seq[(size_t)i].second]) if (dirty[(size_t)v] == 1) { dirty[(size_t)v] = 2; ship.push_back(v); }
      std::sort(ship.begin(), ship.end());
      for (int64_t v : ship) dirty[(size_t)v] = 0;
      dirty_list.erase(std::remove_if(dirty_list.begin(), dirty_list.end(), [&](int64_t v) { return !dirty[(size_t)v]; }), dirty_list.end());
      if (!ship.empty()) halos.push_back({pos, std::move(ship)});
      start = pos;
    }
    for (int64_t i = start; i < (int64_t)seq.size(); ++i) mark(written[seq[(size_t)i].first][(size_t)seq[(size_t)i].second]);
    auto halo_step = [&](std::vector<int64_t> vecs) {
      lockstep_step st; st.halo = true;
      auto ins = halo_ids.insert({vecs, (int)halo_ids.size()});
      st.id = ins.first->second; st.vecs = std::move(vecs);
      return st;
    };
    std::vector<lockstep_step> steps;
    start = 0;
    for (auto& h : halos) {
      if (h.first > start) { lockstep_step st; st.run.assign(seq.begin() + start, seq.begin() + h.first); steps.push_back(std::move(st)); }
      steps.push_back(halo_step(std::move(h.second))); start = h.first;
    }
    if (start < (int64_t)seq.size()) { lockstep_step st; st.run.assign(seq.begin() + start, seq.end()); steps.push_back(std::move(st)); }
    if (!dirty_list.empty()) { std::sort(dirty_list.begin(), dirty_list.end()); steps.push_back(halo_step(dirty_list)); }   // the copies agree again when the call returns
    return programs[n_passes] = std::move(steps);
  }
  double exchanges_per_pass(int n) { int64_t h = 0; for (const auto& s : program(n)) h += s.halo; return (double)h / n; }
};

// one part on its engine
class lockstep_part {
 public:
  int part = 0; int32_t L = 0; bool potts = false;
  mrf_arrays arrays;
  std::vector<int64_t> vars_global, edges_global;     // local variable / local edge -> global (both ascending); MRF parts
  std::vector<int64_t> factors_global;                // local factor -> global factor (parts of a general model)
  std::vector<uint8_t> is_ghost, owned;
  std::vector<int64_t> vec_ids, vec_start; std::vector<int32_t> vec_len;   // the exchange units held here: global id (ascending), first element in the local dual array, length
  // a general part keeps the model arrays its engine was given
  std::vector<int32_t> g_type, g_dim0, g_dim1, gm_type, gm_left, gm_right, g_rel_fwd, g_rel_bwd;
  std::vector<uint8_t> g_kind, g_flags;
  std::vector<double> g_const, g_dual;
  struct rows { std::vector<int32_t> factors; std::vector<int64_t> om_off, mk_off; std::vector<double> om; std::vector<uint8_t> mk; };
  std::vector<rows> sub[2];                          // this part's updates per (direction, sub-level), sequence order inside
  lpmp_engine* e = nullptr; hipStream_t stream = nullptr;
  double *d_const = nullptr, *d_dual = nullptr;
  std::map<std::vector<std::pair<int, int>>, int> sids;
  struct halo { lpmp_halo* h = nullptr; double *d_send = nullptr, *d_recv = nullptr; std::vector<int64_t> out_count, in_count; int64_t n_out = 0, n_in = 0; bool ready = false; };
  std::map<int, halo> halos;
  int64_t updates_per_pass = 0;

  lockstep_part() = default;
  lockstep_part(const lockstep_part&) = delete;
  lockstep_part& operator=(const lockstep_part&) = delete;
  ~lockstep_part() {
    for (auto& kv : halos) { if (kv.second.h) lpmp_halo_destroy(kv.second.h); for (double* p : {kv.second.d_send, kv.second.d_recv}) if (p) (void)hipFree(p); }
    if (e) lpmp_destroy(e);
    for (double* p : {d_const, d_dual}) if (p) (void)hipFree(p);
  }
  int64_t n_vec() const { return (int64_t)vars_global.size(); }

  void build(const lockstep_structure& s, const lockstep_plan& pl, int k, int device, hipStream_t st, int mode) {
    part = k; L = s.L; potts = s.potts; stream = st;
    const int64_t n = s.n_vars, ne = s.n_edges();
    std::vector<uint8_t> in_part((size_t)n, 0);
    for (int64_t e_ = 0; e_ < ne; ++e_)                       // every edge touching a local variable, global order
      if (s.part[(size_t)s.ei[e_]] == k || s.part[(size_t)s.ej[e_]] == k) { edges_global.push_back(e_); in_part[(size_t)s.ei[e_]] = in_part[(size_t)s.ej[e_]] = 1; }
    for (int64_t v = 0; v < n; ++v) if (s.part[(size_t)v] == k) in_part[(size_t)v] = 1;
    std::vector<int64_t> lmap((size_t)n, -1);
    for (int64_t v = 0; v < n; ++v) if (in_part[(size_t)v]) { lmap[(size_t)v] = (int64_t)vars_global.size(); vars_global.push_back(v); is_ghost.push_back(s.part[(size_t)v] != k); }
    std::vector<int64_t> li, lj;
    for (int64_t e_ : edges_global) { li.push_back(lmap[(size_t)s.ei[e_]]); lj.push_back(lmap[(size_t)s.ej[e_]]); }
    arrays.build(n_vec(), L, potts, li, lj);
    owned.assign((size_t)(n_vec() + (int64_t)edges_global.size()), 0);
    for (int64_t v = 0; v < n_vec(); ++v) owned[(size_t)v] = !is_ghost[(size_t)v];
    for (size_t x = 0; x < edges_global.size(); ++x) owned[(size_t)n_vec() + x] = s.part[(size_t)s.ei[(size_t)edges_global[x]]] == k;   // a pairwise factor counts where its earlier endpoint lives
    for (size_t x = 0; x < edges_global.size(); ++x) for (int sd = 0; sd < 2; ++sd) {
      vec_ids.push_back(2 * edges_global[x] + sd); vec_start.push_back(n_vec() * L + (int64_t)x * 2 * L + sd * L); vec_len.push_back(L);
    }
    rows_of(pl, [&](int64_t f) { return s.part[(size_t)f] == k; }, lmap);
    // costs from the counter stream, generated in HBM: unary of local vector x at vars_global[x] L, pairwise data of local edge x
    // at n L + edges_global[x] esz (ghost unaries are generated too: they are never read)
    hip_ok(hipSetDevice(device), "hipSetDevice");
    build_device(s, n, device, mode);
  }

  // a part of ANY model the plan accepts (lockstep.lockstep_model): its variables, every higher factor touching one of them with ALL
  // its messages, never-updated ghosts of the remote variables behind those; factors and messages keep the global relative order.
  // Costs come from the host arrays of ``gm`` (const_data / dual_data)
  void build(const lpmp_model& gm, const std::vector<int32_t>& part_of, const lockstep_plan& pl, int k, int device, hipStream_t st, int mode) {
    part = k; L = 0; stream = st;
    const int64_t nf = gm.n_factors, nm = gm.n_messages;
    std::vector<uint8_t> is_right((size_t)nf, 0), keep((size_t)nf, 0), in_r((size_t)nf, 0);
    for (int64_t m_ = 0; m_ < nm; ++m_) is_right[(size_t)gm.m_right[m_]] = 1;
    auto local = [&](int64_t f) { return !is_right[(size_t)f] && part_of[(size_t)f] == k; };
    for (int64_t m_ = 0; m_ < nm; ++m_) if (local(gm.m_left[m_])) in_r[(size_t)gm.m_right[m_]] = 1;
    std::vector<int64_t> mk_sel;
    for (int64_t m_ = 0; m_ < nm; ++m_) if (in_r[(size_t)gm.m_right[m_]]) { mk_sel.push_back(m_); keep[(size_t)gm.m_left[m_]] = 1; }
    for (int64_t f = 0; f < nf; ++f) if (local(f) || in_r[(size_t)f]) keep[(size_t)f] = 1;
    std::vector<int64_t> lmap((size_t)nf, -1), coff((size_t)nf + 1, 0), doff((size_t)nf + 1, 0);
    for (int64_t f = 0; f < nf; ++f) {
      const int64_t d0 = gm.f_dim0[f], d1 = gm.f_dim1 ? gm.f_dim1[f] : 0;
      coff[(size_t)f + 1] = coff[(size_t)f] + (gm.f_kind[f] == LPMP_F_PAIRWISE_DENSE ? d0 * d1 : gm.f_kind[f] == LPMP_F_PAIRWISE_POTTS ? 1 : 0);
      doff[(size_t)f + 1] = doff[(size_t)f] + (gm.f_kind[f] == LPMP_F_PAIRWISE_DENSE ? d0 + d1 : gm.f_kind[f] == LPMP_F_PAIRWISE_POTTS ? 2 * d0 : d0);
    }
    std::vector<int64_t> ldoff(1, 0);
    for (int64_t f = 0; f < nf; ++f) {
      if (!keep[(size_t)f]) continue;
      lmap[(size_t)f] = (int64_t)factors_global.size(); factors_global.push_back(f);
      g_type.push_back(gm.f_type[f]); g_kind.push_back(gm.f_kind[f]); g_flags.push_back(gm.f_flags ? gm.f_flags[f] : 0);
      g_dim0.push_back(gm.f_dim0[f]); g_dim1.push_back(gm.f_dim1 ? gm.f_dim1[f] : 0);
      if (gm.const_data) g_const.insert(g_const.end(), gm.const_data + coff[(size_t)f], gm.const_data + coff[(size_t)f + 1]);
      if (gm.dual_data) g_dual.insert(g_dual.end(), gm.dual_data + doff[(size_t)f], gm.dual_data + doff[(size_t)f + 1]); else g_dual.resize(g_dual.size() + (size_t)(doff[(size_t)f + 1] - doff[(size_t)f]), 0.0);
      ldoff.push_back(ldoff.back() + doff[(size_t)f + 1] - doff[(size_t)f]);
      is_ghost.push_back(!is_right[(size_t)f] && part_of[(size_t)f] != k);
      owned.push_back(pl.owner[(size_t)f] == k);
      if (!is_right[(size_t)f]) vars_global.push_back(f);
    }
    for (int64_t m_ : mk_sel) {
      gm_type.push_back(gm.m_type[m_]); gm_left.push_back((int32_t)lmap[(size_t)gm.m_left[m_]]); gm_right.push_back((int32_t)lmap[(size_t)gm.m_right[m_]]);
      vec_ids.push_back(m_); vec_start.push_back(ldoff[(size_t)lmap[(size_t)gm.m_right[m_]]] + pl.v_off[(size_t)m_]); vec_len.push_back((int32_t)pl.v_len[(size_t)m_]);
    }
    auto map_rel = [&](const int32_t* rel, int64_t n_rel, std::vector<int32_t>& out) {
      for (int64_t i = 0; i < n_rel; ++i) { const int64_t a = lmap[(size_t)rel[2 * i]], b = lmap[(size_t)rel[2 * i + 1]]; if (a >= 0 && b >= 0) { out.push_back((int32_t)a); out.push_back((int32_t)b); } }
    };
    map_rel(gm.rel_fwd, gm.n_rel_fwd, g_rel_fwd); map_rel(gm.rel_bwd, gm.n_rel_bwd, g_rel_bwd);
    rows_of(pl, local, lmap);
    hip_ok(hipSetDevice(device), "hipSetDevice");
    lpmp_model m = gm;                          // types, tables, constant: the global model's
    m.n_factors = (int64_t)factors_global.size(); m.f_type = g_type.data(); m.f_kind = g_kind.data(); m.f_flags = g_flags.data();
    m.f_dim0 = g_dim0.data(); m.f_dim1 = g_dim1.data();
    if (g_const.empty()) g_const.push_back(0.0);
    m.const_data = g_const.data(); m.dual_data = g_dual.data();
    m.n_messages = (int64_t)gm_type.size(); m.m_type = gm_type.data(); m.m_left = gm_left.data(); m.m_right = gm_right.data();
    m.n_rel_fwd = (int64_t)g_rel_fwd.size() / 2; m.rel_fwd = g_rel_fwd.data(); m.n_rel_bwd = (int64_t)g_rel_bwd.size() / 2; m.rel_bwd = g_rel_bwd.data();
    m.n_part_pairs = 0; m.part_pairs = nullptr;
    if (k != 0) m.constant = 0.0;
    lpmp_ok(lpmp_create(device, &e));
    lpmp_ok(lpmp_set_stream(e, stream));
    lpmp_ok(lpmp_upload_model(e, &m, LPMP_MEM_HOST, LPMP_MEM_HOST));
    lpmp_ok(lpmp_set_reparametrization(e, mode));
  }

  // this part's rows of the global sweep per (direction, sub-level)
  template <class Local>
  void rows_of(const lockstep_plan& pl, Local&& local, const std::vector<int64_t>& lmap) {
    for (int d = 0; d < 2; ++d) {
      const int nsl = 2 * pl.n_levels[d];
      sub[d].assign((size_t)nsl, {});
      for (auto& r : sub[d]) { r.om_off.assign(1, 0); r.mk_off.assign(1, 0); }
      for (int64_t u = 0; u < (int64_t)pl.upd[d].size(); ++u) {       // update order = sequence order inside a sub-level
        const int64_t f = pl.upd[d][(size_t)u];
        if (!local(f)) continue;
        rows& r = sub[d][(size_t)(2 * (pl.lev[d][(size_t)u] - 1) + (pl.touches_cut[d][(size_t)u] ? 0 : 1))];
        r.factors.push_back((int32_t)lmap[(size_t)f]);
        for (int64_t j = pl.om_off[d][(size_t)u]; j < pl.om_off[d][(size_t)u + 1]; ++j) { r.om.push_back(pl.om[d][(size_t)j]); updates_per_pass += pl.om[d][(size_t)j] != 0.0; }
        for (int64_t j = pl.mk_off[d][(size_t)u]; j < pl.mk_off[d][(size_t)u + 1]; ++j) { r.mk.push_back(pl.mk[d][(size_t)j]); updates_per_pass += pl.mk[d][(size_t)j] != 0; }
        r.om_off.push_back((int64_t)r.om.size()); r.mk_off.push_back((int64_t)r.mk.size());
      }
    }
  }

  void build_device(const lockstep_structure& s, int64_t n, int device, int mode) {
    const int64_t esz = potts ? 1 : (int64_t)L * L, n_e = (int64_t)edges_global.size();
    const int64_t nc = std::max<int64_t>(n_e * esz, 2), nd = n_vec() * L + n_e * 2 * L;
    hip_ok(hipMalloc((void**)&d_const, (size_t)nc * sizeof(double)), "hipMalloc const");
    hip_ok(hipMalloc((void**)&d_dual, (size_t)nd * sizeof(double)), "hipMalloc dual");
    hip_ok(hipMemsetAsync(d_dual, 0, (size_t)nd * sizeof(double), stream), "hipMemsetAsync");
    std::vector<int64_t> first;
    int64_t* d_first = nullptr;
    auto fill = [&](double* dst, int64_t n_blocks, int64_t block_len) {
      if (n_blocks == 0) return;
      if (d_first) { (void)hipFree(d_first); d_first = nullptr; }
      hip_ok(hipMalloc((void**)&d_first, first.size() * sizeof(int64_t)), "hipMalloc");
      hip_ok(hipMemcpyAsync(d_first, first.data(), first.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
      lpmp_ok(lpmp_synth_fill_blocks(dst, n_blocks, block_len, s.seed, d_first, stream));
      hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    };
    first.resize((size_t)n_vec());
    for (int64_t x = 0; x < n_vec(); ++x) first[(size_t)x] = vars_global[(size_t)x] * L;
    fill(d_dual, n_vec(), L);
    first.resize((size_t)n_e);
    for (int64_t x = 0; x < n_e; ++x) first[(size_t)x] = n * L + edges_global[(size_t)x] * esz;
    fill(d_const, n_e, esz);
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (d_first) (void)hipFree(d_first);
    lpmp_ok(lpmp_create(device, &e));
    lpmp_ok(lpmp_set_stream(e, stream));
    const lpmp_model m = arrays.view(d_const, d_dual);
    lpmp_ok(lpmp_upload_model(e, &m, LPMP_MEM_DEVICE, LPMP_MEM_DEVICE));
    lpmp_ok(lpmp_set_reparametrization(e, mode));
  }

  // a run of sub-levels as one schedule (rows concatenated; fused when the run spans more than one sweep)
  int schedule(const std::vector<std::pair<int, int>>& seg) {
    auto it = sids.find(seg);
    if (it != sids.end()) return it->second;
    std::vector<int32_t> f; std::vector<int64_t> oo(1, 0), mo(1, 0); std::vector<double> o; std::vector<uint8_t> m;
    for (const auto& ds : seg) {
      const rows& r = sub[ds.first][(size_t)ds.second];
      f.insert(f.end(), r.factors.begin(), r.factors.end());
      for (size_t i = 1; i < r.om_off.size(); ++i) oo.push_back((int64_t)o.size() + r.om_off[i]);
      for (size_t i = 1; i < r.mk_off.size(); ++i) mo.push_back((int64_t)m.size() + r.mk_off[i]);
      o.insert(o.end(), r.om.begin(), r.om.end()); m.insert(m.end(), r.mk.begin(), r.mk.end());
    }
    int n_sweeps = 1;
    for (size_t i = 1; i < seg.size(); ++i) if (seg[i].first != seg[i - 1].first || seg[i].second < seg[i - 1].second) ++n_sweeps;
    int sid = -1;
    if (!f.empty()) {
      if (o.empty()) o.push_back(0.0);
      if (m.empty()) m.push_back(0);
      if (n_sweeps > 1) lpmp_ok(lpmp_schedule_create_fused(e, (int64_t)f.size(), f.data(), oo.data(), o.data(), mo.data(), m.data(), 1, &sid));
      else lpmp_ok(lpmp_schedule_create(e, (int64_t)f.size(), f.data(), oo.data(), o.data(), mo.data(), m.data(), &sid));
    }
    return sids[seg] = sid;
  }
  void run(const std::vector<std::pair<int, int>>& seg) { const int sid = schedule(seg); if (sid >= 0) lpmp_ok(lpmp_schedule_run(e, sid)); }

  // the exchange plan of one halo step: what this part sends (its own cut vectors of the step, by destination part, then vector)
  // and receives (by source part, then vector)
  halo& halo_plan(const lockstep_plan& pl, const lockstep_step& st, int n_parts) {
    auto it = halos.find(st.id);
    if (it != halos.end()) { if (!it->second.ready) throw std::runtime_error("lockstep: an exchange plan that failed to build"); return it->second; }
    halo& h = halos[st.id];                  // (in the map from the start: whatever is allocated below is released with the part)
    h.out_count.assign((size_t)n_parts, 0); h.in_count.assign((size_t)n_parts, 0);
    std::vector<std::pair<int32_t, int64_t>> out, in;
    for (int64_t v : st.vecs) {
      for (int64_t j = pl.dest_off[(size_t)v]; j < pl.dest_off[(size_t)v + 1]; ++j) {
        if (pl.writer[(size_t)v] == part) out.push_back({pl.dest[(size_t)j], v});
        if (pl.dest[(size_t)j] == part) in.push_back({pl.writer[(size_t)v], v});
      }
    }
    std::sort(out.begin(), out.end()); std::sort(in.begin(), in.end());
    auto offsets = [&](const std::vector<std::pair<int32_t, int64_t>>& l, std::vector<int64_t>& off, std::vector<int32_t>& len, std::vector<int64_t>& count) {
      int64_t total = 0;
      for (const auto& pv : l) {
        const auto pos = std::lower_bound(vec_ids.begin(), vec_ids.end(), pv.second);
        if (pos == vec_ids.end() || *pos != pv.second) throw std::runtime_error("lockstep: a cut vector of a factor this part does not hold");
        const size_t x = (size_t)(pos - vec_ids.begin());
        off.push_back(vec_start[x]); len.push_back(vec_len[x]); count[(size_t)pv.first] += vec_len[x]; total += vec_len[x];
      }
      return total;
    };
    std::vector<int64_t> o_off, i_off; std::vector<int32_t> o_len, i_len;
    h.n_out = offsets(out, o_off, o_len, h.out_count); h.n_in = offsets(in, i_off, i_len, h.in_count);
    lpmp_ok(lpmp_halo_create(e, (int64_t)o_off.size(), o_off.data(), o_len.data(), (int64_t)i_off.size(), i_off.data(), i_len.data(), &h.h));
    hip_ok(hipMalloc((void**)&h.d_send, (size_t)std::max<int64_t>(h.n_out, 1) * sizeof(double)), "hipMalloc");
    hip_ok(hipMalloc((void**)&h.d_recv, (size_t)std::max<int64_t>(h.n_in, 1) * sizeof(double)), "hipMalloc");
    h.ready = true;
    return h;
  }

  double local_lower_bound() {
    std::vector<double> flb(owned.size());
    lpmp_ok(lpmp_factor_lower_bounds(e, flb.data()));
    double lb = 0;
    for (size_t f = 0; f < owned.size(); ++f) if (owned[f]) lb += flb[f];
    return lb;
  }
  std::vector<double> download_duals() { std::vector<double> d((size_t)lpmp_dual_size(e)); lpmp_ok(lpmp_download_duals(e, d.data())); return d; }
};

// one exchange: every part packs, one ncclGroup of sends / receives between the ranks (parts of one rank: device copies), every
// part unpacks.  parts = this rank's parts in part order.  Transfers are issued in a fixed global order — by (source part,
// destination part) — so that the k-th send of rank a to rank b meets the k-th receive of b from a (as lpmp_multi_gpu.hxx)
inline void lockstep_exchange(std::vector<lockstep_part*>& parts, lockstep_plan& pl, const lockstep_step& st, rccl_world& w, int n_parts) {
  std::vector<lockstep_part::halo*> hs;
  for (lockstep_part* p : parts) hs.push_back(&p->halo_plan(pl, st, n_parts));      // (plans are built outside the probed span)
  w.probe_begin();
  for (size_t x = 0; x < parts.size(); ++x) lpmp_ok(lpmp_halo_pack(parts[x]->e, hs[x]->h, hs[x]->d_send));
  const int first_part = w.rank * w.parts_per_rank;
  auto offset = [](const std::vector<int64_t>& count, int q) { int64_t off = 0; for (int r = 0; r < q; ++r) off += count[(size_t)r]; return off; };
  // (two parts of this rank must agree on what travels between them: checked before the group is opened)
  for_each_transfer(n_parts, w.rank, w.parts_per_rank, [&](int src, int dst, bool src_here, bool dst_here) {
      if (src_here && dst_here && hs[(size_t)(src - first_part)]->out_count[(size_t)dst] != hs[(size_t)(dst - first_part)]->in_count[(size_t)src])
        throw std::runtime_error("lockstep: two parts disagree on an exchange");
    });
  nccl_group grp;      // (closed on every path, also when a transfer below throws)
  for_each_transfer(n_parts, w.rank, w.parts_per_rank, [&](int src, int dst, bool src_here, bool dst_here) {
      if (src_here && dst_here) {
        lockstep_part::halo &hs_ = *hs[(size_t)(src - first_part)], &hd = *hs[(size_t)(dst - first_part)];
        const int64_t c = hs_.out_count[(size_t)dst];
        if (c > 0) hip_ok(hipMemcpyAsync(hd.d_recv + offset(hd.in_count, src), hs_.d_send + offset(hs_.out_count, dst), (size_t)c * sizeof(double), hipMemcpyDeviceToDevice, w.stream), "hipMemcpyAsync");
      } else if (src_here) {
        lockstep_part::halo& h = *hs[(size_t)(src - first_part)];
        const int64_t c = h.out_count[(size_t)dst];
        if (c > 0) nccl_ok(ncclSend(h.d_send + offset(h.out_count, dst), (size_t)c, ncclDouble, w.rank_of(dst), w.comm, w.stream), "ncclSend");
      } else {
        lockstep_part::halo& h = *hs[(size_t)(dst - first_part)];
        const int64_t c = h.in_count[(size_t)src];
        if (c > 0) nccl_ok(ncclRecv(h.d_recv + offset(h.in_count, src), (size_t)c, ncclDouble, w.rank_of(src), w.comm, w.stream), "ncclRecv");
      }
    });
  grp.end();
  for (size_t x = 0; x < parts.size(); ++x) lpmp_ok(lpmp_halo_unpack(parts[x]->e, hs[x]->h, hs[x]->d_recv));
  if (w.probe.on) { int64_t by = 0; for (auto* h : hs) for (int64_t c : h->out_count) by += 8 * c; w.probe_end(by); }
}

inline void lockstep_compute_pass(std::vector<lockstep_part*>& parts, lockstep_plan& pl, rccl_world& w, int n_parts, int n) {
  for (const lockstep_step& st : pl.program(n)) {
    if (!st.halo) { for (lockstep_part* p : parts) p->run(st.run); continue; }
    lockstep_exchange(parts, pl, st, w, n_parts);
  }
}
// schedules and exchange plans of an n-pass call, outside a timed region
inline void lockstep_prepare(std::vector<lockstep_part*>& parts, lockstep_plan& pl, int n_parts, int n) {
  for (const lockstep_step& st : pl.program(n)) for (lockstep_part* p : parts) { if (st.halo) (void)p->halo_plan(pl, st, n_parts); else (void)p->schedule(st.run); }
}
inline double lockstep_lower_bound(std::vector<lockstep_part*>& parts, rccl_world& w) {
  double lb = 0;
  for (lockstep_part* p : parts) lb += p->local_lower_bound();
  return w.all_reduce_sum(lb);
}

}  // namespace lpmp_mgpu
