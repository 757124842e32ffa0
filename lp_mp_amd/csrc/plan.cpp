// plan.cpp — see plan.hpp.  Host restatement of the reference's ordering / weight rules, plus the
// level scheduling that makes the Gauss-Seidel sweep executable as a few wide kernel launches.
#include "plan.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace lpmp {

namespace {

// Contiguous chunks of [0, n) on up to LPMP_PLAN_THREADS (default: the hardware's, at most 16) threads; small ranges run on the
// caller's thread.  The analysis below is a handful of linear passes over millions of updates: the two that build the
// op records and the packets are independent per update / per record.  An exception of any chunk is rethrown.
constexpr int PLAN_MAX_THREADS = 64;
template <class F>
void parallel_chunks(int64_t n, int64_t min_per_thread, F&& f) {
  static const int max_threads = [] {
    const char* e = std::getenv("LPMP_PLAN_THREADS");
    const int hw = (int)std::thread::hardware_concurrency();
    // (the per-thread scratch of the callers below has PLAN_MAX_THREADS slots: the thread index never exceeds it)
    return std::max(1, std::min(PLAN_MAX_THREADS, e ? std::atoi(e) : std::min(16, hw > 0 ? hw : 1)));
  }();
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(max_threads, n / std::max<int64_t>(1, min_per_thread)));
  if (nt <= 1) { f((int64_t)0, n, 0); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  th.reserve((size_t)nt);
  auto work = [&](int t) { try { f(n * t / nt, n * (t + 1) / nt, t); } catch (...) { err[(size_t)t] = std::current_exception(); } };
  int started = 0;
  try {
    for (; started < nt - 1; ++started) th.emplace_back(work, started);
  } catch (...) {}                                   // (no more threads to be had: the caller's thread takes the rest, chunk by chunk)
  for (int t = started; t < nt; ++t) work(t);
  for (auto& x : th) x.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}

// message_passing_schedule -> what each side does (reference factors_messages.hxx:1530-1545)
struct SchedCaps { bool to_left, to_right, from_left, from_right; };
SchedCaps caps(int s) {
  switch (s) {
    case LPMP_SCHED_LEFT: return {false, true, false, true};
    case LPMP_SCHED_RIGHT: return {true, false, true, false};
    case LPMP_SCHED_FULL: return {true, true, true, true};
    case LPMP_SCHED_ONLY_SEND: return {true, true, false, false};
    default: return {false, false, false, false};
  }
}

[[noreturn]] void fail(const std::string& s) { throw std::runtime_error(s); }

// reference topological_sort.hxx:100-144 — DFS, roots by index, successors by insertion, reversed post-order
std::vector<int32_t> reference_topological_order(int64_t nf, const int32_t* rel, int64_t n_rel) {
  std::vector<int64_t> head(nf + 1, 0);
  for (int64_t i = 0; i < n_rel; ++i) {
    const int32_t a = rel[2 * i], b = rel[2 * i + 1];
    if (a < 0 || a >= nf || b < 0 || b >= nf) fail("factor relation out of range");
    head[a + 1]++;
  }
  std::partial_sum(head.begin(), head.end(), head.begin());
  std::vector<int32_t> succ(n_rel);
  {
    std::vector<int64_t> cur(head.begin(), head.end() - 1);
    for (int64_t i = 0; i < n_rel; ++i) succ[cur[rel[2 * i]]++] = rel[2 * i + 1];
  }
  std::vector<uint8_t> seen(nf, 0);
  std::vector<int32_t> post;
  post.reserve(nf);
  struct Frame { int32_t node; int64_t next; };
  std::vector<Frame> st;
  for (int64_t root = 0; root < nf; ++root) {
    if (seen[root]) continue;
    seen[root] = 1;
    st.push_back({(int32_t)root, head[root]});
    while (!st.empty()) {
      Frame& fr = st.back();
      const int64_t end = head[fr.node + 1];
      while (fr.next != end && seen[succ[fr.next]]) ++fr.next;
      if (fr.next == end) {
        post.push_back(fr.node);
        st.pop_back();
      } else {
        const int32_t nx = succ[fr.next++];
        seen[nx] = 1;
        st.push_back({nx, head[nx]});
      }
    }
  }
  std::reverse(post.begin(), post.end());
  return post;
}

}  // namespace

void Plan::build(const lpmp_model& m) {
  const bool timed_ = std::getenv("LPMP_PLAN_TIMES") != nullptr;
  auto t_last_ = std::chrono::steady_clock::now();
  auto lap_ = [&](const char* what) { if (!timed_) return; const auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "lpmp: plan build %-10s %.0f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last_).count()); t_last_ = now; };
  if (m.n_ftypes <= 0 || m.n_factors <= 0) fail("model has no factors");
  if (m.n_factors > std::numeric_limits<int32_t>::max() || m.n_messages > std::numeric_limits<int32_t>::max() / 2)
    fail("model too large for 32-bit factor/message indices");
  n_ftypes = m.n_ftypes; n_mtypes = m.n_mtypes; n_tables = m.n_tables;
  ftype_primal.assign(n_ftypes, 0);
  if (m.ftype_computes_primal) ftype_primal.assign(m.ftype_computes_primal, m.ftype_computes_primal + n_ftypes);
  mtypes.assign(m.mtypes, m.mtypes + n_mtypes);
  if (n_tables < 0) fail("negative table count");
  if (n_tables > 0) {
    if (!m.tab_off || !m.tab_data || !m.tab_nleft) fail("labeling tables missing");
    tab_off.assign(m.tab_off, m.tab_off + n_tables + 1);
    tab_data.assign(m.tab_data, m.tab_data + tab_off[n_tables]);
    tab_nleft.assign(m.tab_nleft, m.tab_nleft + n_tables);
    // an entry indexes the left factor's labelings (tab_nleft = "no matching left labeling"): anything else would be
    // an out-of-bounds access in the labeling receive / send kernels
    if (tab_off[0] != 0) fail("labeling tables: tab_off must start at 0");
    for (int t = 0; t < n_tables; ++t) {
      if (tab_off[t + 1] < tab_off[t] || tab_nleft[t] <= 0) fail("labeling table " + std::to_string(t) + ": bad offsets / left labeling count");
      for (int64_t j = tab_off[t]; j < tab_off[t + 1]; ++j)
        if (tab_data[j] < 0 || tab_data[j] > tab_nleft[t]) fail("labeling table " + std::to_string(t) + ": entry out of range");
    }
  }
  nf = m.n_factors; nm = m.n_messages; constant = m.constant;
  f_type.assign(m.f_type, m.f_type + nf);
  f_kind.assign(m.f_kind, m.f_kind + nf);
  if (m.f_flags) f_flags.assign(m.f_flags, m.f_flags + nf); else f_flags.assign(nf, 0);
  f_dim0.assign(m.f_dim0, m.f_dim0 + nf);
  if (m.f_dim1) f_dim1.assign(m.f_dim1, m.f_dim1 + nf); else f_dim1.assign(nf, 0);
  f_coff.assign(nf + 1, 0); f_doff.assign(nf + 1, 0);
  max_dual = 1;
  for (int64_t f = 0; f < nf; ++f) {
    if (f_kind[f] > LPMP_F_PAIRWISE_POTTS) fail("factor " + std::to_string(f) + ": unknown kind");
    if (f_type[f] < 0 || f_type[f] >= n_ftypes) fail("factor " + std::to_string(f) + ": type out of range");
    if (f_dim0[f] <= 0 || (f_kind[f] == LPMP_F_PAIRWISE_DENSE && f_dim1[f] <= 0)) fail("factor " + std::to_string(f) + ": bad dimension");
    if (f_kind[f] == LPMP_F_PAIRWISE_POTTS) f_dim1[f] = f_dim0[f];
    if (f_kind[f] == LPMP_F_VECTOR) f_dim1[f] = 0;
    f_coff[f + 1] = f_coff[f] + lpmp_factor_const_size(f_kind[f], f_dim0[f], f_dim1[f]);
    const int64_t ds = lpmp_factor_dual_size(f_kind[f], f_dim0[f], f_dim1[f]);
    f_doff[f + 1] = f_doff[f] + ds;
    max_dual = std::max<int64_t>(max_dual, ds);
  }
  if (m.n_part_pairs < 0 || (m.n_part_pairs > 0 && !m.part_pairs)) fail("partition pairs missing");
  part_pairs.assign(m.part_pairs, m.part_pairs + 2 * m.n_part_pairs);
  for (int32_t x : part_pairs) if (x < 0 || x >= nf) fail("put_in_same_partition: factor out of range");
  part = Partition();
  m_type.assign(m.m_type, m.m_type + nm);
  m_left.assign(m.m_left, m.m_left + nm);
  m_right.assign(m.m_right, m.m_right + nm);
  any_batch = false;
  for (int t = 0; t < n_mtypes; ++t) {
    const auto& mt = mtypes[t];
    if (mt.flags & ~(LPMP_MF_IMPROVEMENT | LPMP_MF_BATCH_TO_RIGHT | LPMP_MF_BATCH_TO_LEFT)) fail("message type " + std::to_string(t) + ": unknown flags");
    any_batch = any_batch || (mt.flags & (LPMP_MF_BATCH_TO_RIGHT | LPMP_MF_BATCH_TO_LEFT)) != 0;
    if (mt.left_ftype < 0 || mt.left_ftype >= n_ftypes || mt.right_ftype < 0 || mt.right_ftype >= n_ftypes)
      fail("message type " + std::to_string(t) + ": factor type out of range");
    if (mt.schedule < 0 || mt.schedule > LPMP_SCHED_NONE) fail("message type " + std::to_string(t) + ": bad schedule");
    if (mt.kind < 0 || mt.kind > LPMP_M_MINNORM) fail("message type " + std::to_string(t) + ": unknown kind");
  }
  for (int64_t i = 0; i < nm; ++i) {
    const int t = m_type[i];
    const int32_t l = m_left[i], r = m_right[i];
    if (t < 0 || t >= n_mtypes || l < 0 || l >= nf || r < 0 || r >= nf || l == r) fail("message " + std::to_string(i) + ": index out of range");
    const auto& mt = mtypes[t];
    bool ok = f_type[l] == mt.left_ftype && f_type[r] == mt.right_ftype && f_kind[l] == LPMP_F_VECTOR;
    if (ok && mt.kind == LPMP_M_UNARY_PAIRWISE)
      ok = f_kind[r] != LPMP_F_VECTOR && (mt.param == 0 || mt.param == 1) && f_dim0[l] == (mt.param == 0 ? f_dim0[r] : f_dim1[r]);
    if (ok && mt.kind == LPMP_M_LABELING)
      ok = f_kind[r] == LPMP_F_VECTOR && mt.param >= 0 && mt.param < n_tables && tab_nleft[mt.param] == f_dim0[l] &&
           tab_off[mt.param + 1] - tab_off[mt.param] == f_dim0[r];
    if (ok && mt.kind == LPMP_M_MINNORM) ok = f_kind[r] == LPMP_F_VECTOR && f_dim0[r] == f_dim0[l];
    if (!ok) fail("message " + std::to_string(i) + ": factors do not fit the message type");
  }
  lap_("checks");

  // ---- per-factor message lists: dispatcher order = left-role types in MessageList order, then right-role
  std::vector<int32_t> rank_l(n_mtypes), rank_r(n_mtypes);
  for (int k = 0; k < n_ftypes; ++k) {
    int r = 0;
    for (int t = 0; t < n_mtypes; ++t) if (mtypes[t].left_ftype == k) rank_l[t] = r++;
    for (int t = 0; t < n_mtypes; ++t) if (mtypes[t].right_ftype == k) rank_r[t] = r++;
  }
  fm_off.assign(nf + 1, 0);
  for (int64_t i = 0; i < nm; ++i) { fm_off[m_left[i] + 1]++; fm_off[m_right[i] + 1]++; }
  std::partial_sum(fm_off.begin(), fm_off.end(), fm_off.begin());
  fm.resize(2 * nm);
  struct Tmp { int32_t rank; int32_t msg; uint8_t role; };
  std::vector<Tmp> tmp(2 * nm);
  {
    std::vector<int64_t> cur(fm_off.begin(), fm_off.end() - 1);
    for (int64_t i = 0; i < nm; ++i) {   // insertion order inside each factor
      tmp[cur[m_left[i]]++] = {rank_l[m_type[i]], (int32_t)i, 0};
      tmp[cur[m_right[i]]++] = {rank_r[m_type[i]], (int32_t)i, 1};
    }
  }
  updated.assign(nf, 0);
  n_row_sends.assign(nf, 0); n_row_receives.assign(nf, 0);
  lap_("msg lists");
  parallel_chunks(nf, 65536, [&](int64_t f_begin, int64_t f_end, int) {
  for (int64_t f = f_begin; f < f_end; ++f) {
    Tmp* b = tmp.data() + fm_off[f];
    Tmp* e = tmp.data() + fm_off[f + 1];
    std::stable_sort(b, e, [](const Tmp& x, const Tmp& y) { return x.rank < y.rank; });
    for (Tmp* p = b; p != e;) {          // runs of one dispatcher; LIFO storages iterate newest first
      Tmp* q = p;
      while (q != e && q->rank == p->rank) ++q;
      const auto& mt = mtypes[m_type[p->msg]];
      const bool lifo = p->role == 0 ? (mt.n_left == 0 && mt.n_right != 0) : (mt.n_right == 0 && mt.n_left != 0);
      if (lifo) std::reverse(p, q);
      p = q;
    }
    bool upd_f = ftype_primal[f_type[f]] != 0;
    int32_t n_s = 0, n_r = 0;
    for (Tmp* p = b; p != e; ++p) {
      const SchedCaps c = caps(mtypes[m_type[p->msg]].schedule);
      MsgEntry& en = fm[fm_off[f] + (p - b)];
      en.msg = p->msg; en.role = p->role;
      if (p->role == 0) {
        en.adjacent = m_right[p->msg];
        en.sends = c.to_right; en.receives = c.from_right; en.adj_sends = c.to_left; en.adj_receives = c.from_left;
      } else {
        en.adjacent = m_left[p->msg];
        en.sends = c.to_left; en.receives = c.from_left; en.adj_sends = c.to_right; en.adj_receives = c.from_right;
      }
      upd_f = upd_f || en.sends || en.receives;
      n_s += en.sends; n_r += en.receives;
    }
    updated[f] = upd_f;
    n_row_sends[f] = n_s; n_row_receives[f] = n_r;
  }
  });
  lap_("dispatch");

  // ---- orderings
  const int32_t* rels[2] = {m.rel_fwd, m.rel_bwd};
  const int64_t nrels[2] = {m.n_rel_fwd, m.n_rel_bwd};
  parallel_chunks(2, 1, [&](int64_t d_begin, int64_t d_end, int) {       // the two directions are independent
    for (int64_t d = d_begin; d < d_end; ++d) {
      order[d] = reference_topological_order(nf, rels[d], nrels[d]);
      upd[d].clear();
      for (int32_t f : order[d]) if (updated[f]) upd[d].push_back(f);
    }
  });
  lap_("orderings");
}

// rows for the updated members of a list (reference allocate_omega / allocate_receive_mask)
// row_of (optional): per list position its row, -1 for a member that is not updated
static void shape_rows(const Plan& p, const int32_t* list, int64_t n, Csr<double>& om, Csr<uint8_t>& mk, std::vector<int64_t>* row_of = nullptr) {
  om.off.assign(1, 0); mk.off.assign(1, 0);
  if (row_of) row_of->assign((size_t)n, -1);
  for (int64_t i = 0; i < n; ++i) {
    if (!p.updated[list[i]]) continue;
    if (row_of) (*row_of)[(size_t)i] = (int64_t)om.off.size() - 1;
    om.off.push_back(om.off.back() + p.row_sends(list[i]));
    mk.off.push_back(mk.off.back() + p.row_receives(list[i]));
  }
  om.data.assign(om.off.back(), 0.0);
  mk.data.assign(mk.off.back(), 0);
}

// reference LP_MP.h:1232-1415
void Plan::anisotropic_weights(const int32_t* list, int64_t n, Csr<double>& om, Csr<uint8_t>& mk) const {
  constexpr int64_t NONE = -1, INF = std::numeric_limits<int64_t>::max();
  std::vector<int64_t> pos(nf, NONE);
  for (int64_t i = 0; i < n; ++i) {
    if (list[i] < 0 || list[i] >= nf) fail("factor index out of range");
    pos[list[i]] = i;
  }
  std::vector<int64_t> n_later(n, 0), last(n, 0), first(n, INF);
  parallel_chunks(n, 65536, [&](int64_t i_begin, int64_t i_end, int) {       // every member on its own
  for (int64_t i = i_begin; i < i_end; ++i) {
    const int32_t f = list[i];
    for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
      const MsgEntry& e = fm[j];
      const int64_t a = pos[e.adjacent];
      if (a != NONE && e.adj_receives && a > i) {
        ++n_later[i];
        last[i] = std::max(last[i], a);
        first[i] = std::min(first[i], a);
      }
    }
  }
  });
  // factors outside the list: only those adjacent to >= 2 members get real values, all others read the
  // value-initialised 0 of the reference's unordered_map::operator[] (LP_MP.h:1283-1303, :1322, :1342)
  std::vector<int64_t> out_min_send, out_max_recv;
  if (n < nf) {
    out_min_send.assign(nf, 0); out_max_recv.assign(nf, 0);
    std::vector<int32_t> touch(nf, 0);
    for (int64_t i = 0; i < n; ++i)
      for (int64_t j = fm_off[list[i]]; j < fm_off[list[i] + 1]; ++j)
        if (pos[fm[j].adjacent] == NONE) ++touch[fm[j].adjacent];
    for (int64_t g = 0; g < nf; ++g) {
      if (touch[g] < 2) continue;
      int64_t mn = INF, mx = 0;
      for (int64_t j = fm_off[g]; j < fm_off[g + 1]; ++j) {
        const int64_t a = pos[fm[j].adjacent];
        if (a == NONE) continue;
        if (fm[j].adj_sends) mn = std::min(mn, a);
        if (fm[j].adj_receives) mx = std::max(mx, a);
      }
      out_min_send[g] = mn; out_max_recv[g] = mx;
    }
  }
  std::vector<int64_t> row_of;
  shape_rows(*this, list, n, om, mk, &row_of);
  parallel_chunks(n, 65536, [&](int64_t i_begin, int64_t i_end, int) {       // every row on its own
  for (int64_t i = i_begin; i < i_end; ++i) {
    const int32_t f = list[i];
    const int64_t row = row_of[(size_t)i];
    if (row < 0) continue;
    double* o = om.data.data() + om.off[row];
    uint8_t* r = mk.data.data() + mk.off[row];
    int64_t ns = 0, na = 0, nr = 0;
    for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
      const MsgEntry& e = fm[j];
      const int64_t a = pos[e.adjacent];
      if (e.sends) {
        const bool s = a != NONE ? ((i < a && updated[e.adjacent]) || last[a] > i) : (i < out_max_recv[e.adjacent]);
        o[ns++] = s ? 1.0 : 0.0;
        na += s;
      }
      if (e.receives) r[nr++] = a != NONE ? (a < i || first[a] < i) : (out_min_send[e.adjacent] < i);
    }
    if (na > 0) {
      const double w = 1.0 / double(n_later[i] + std::max(na, ns - na));   // srmp_weight, :1397
      for (int64_t k = 0; k < ns; ++k) if (o[k] > 0) o[k] *= w;
    }
  }
  });
}

void Plan::ensure_weights(int mode) {
  if (mode < 0 || mode >= LPMP_REPAM_COUNT) fail("no reparametrization mode set");
  if (have[mode]) return;
  parallel_chunks(2, 1, [&](int64_t d_begin, int64_t d_end, int) {              // the two directions are independent
  for (int64_t d = d_begin; d < d_end; ++d) {
    Csr<double>& om = omega[d][mode];
    Csr<uint8_t>& mk = mask[d][mode];
    const std::vector<int32_t>& ord = order[d];
    if (mode == LPMP_REPAM_ANISOTROPIC) {
      anisotropic_weights(ord.data(), nf, om, mk);
    } else if (mode == LPMP_REPAM_ANISOTROPIC2) {   // reference LP_MP.h:1086-1154
      std::vector<int64_t> inv(nf), later(nf, 0);
      for (int64_t i = 0; i < nf; ++i) inv[ord[i]] = i;
      for (int64_t i = 0; i < nm; ++i) {
        const SchedCaps c = caps(mtypes[m_type[i]].schedule);
        const int64_t il = inv[m_left[i]], ir = inv[m_right[i]];
        if (c.to_right && il < ir) ++later[il];
        if (c.to_left && ir < il) ++later[ir];
      }
      shape_rows(*this, ord.data(), nf, om, mk);
      int64_t row = 0;
      for (int64_t i = 0; i < nf; ++i) {
        const int32_t f = ord[i];
        if (!updated[f]) continue;
        int64_t ks = om.off[row], kr = mk.off[row];
        for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
          const int64_t a = inv[fm[j].adjacent];
          if (fm[j].sends) om.data[ks++] = i < a ? 1.0 / double(later[i]) : 0.0;
          if (fm[j].receives) mk.data[kr++] = a < i;
        }
        ++row;
      }
    } else {   // uniform / damped uniform, reference LP_MP.h:1422-1449 with leave_weight 0 / 1, full mask :1489
      const double leave = mode == LPMP_REPAM_DAMPED_UNIFORM ? 1.0 : 0.0;
      shape_rows(*this, ord.data(), nf, om, mk);
      for (int64_t r = 0; r < om.rows(); ++r) {
        const double w = 1.0 / (double(om.off[r + 1] - om.off[r]) + leave);
        std::fill(om.data.begin() + om.off[r], om.data.begin() + om.off[r + 1], w);
      }
      std::fill(mk.data.begin(), mk.data.end(), 1);
    }
  }
  });
  have[mode] = true;
}

// ------------------------------------------------------------------------------------------------
// Level scheduling.  The reference sweep is strictly sequential (LP_MP.h:989-992).  Two updates
// commute exactly when they touch disjoint factors, so update u gets
//   level(u) = 1 + max(level of the latest earlier update that touched u or one of the factors u touches)
// and all updates of one level run concurrently with a result identical to the sequential sweep.
//
// Several sweeps can be scheduled as ONE sequence (forward then backward of a pass).  With `fuse`, an
// update u2 that directly follows an update u1 of the SAME factor — nothing else touched anything u2
// touches in between — is folded into u1's record when u1 has no sends: "receive R1; receive R2; send S2
// from the state after the receives" is exactly what running u1 then u2 computes, and the factor's dual
// makes one round trip instead of two (2-colour grids: the receive level of the forward sweep and the
// send level of the backward sweep are the same factors).
void Plan::make_schedule(const std::vector<Segment>& segs, bool fuse, Schedule& out, bool chains, std::vector<int32_t>* levels_only) const {
  out = Schedule();
  const bool timed_ = std::getenv("LPMP_PLAN_TIMES") != nullptr;
  auto t_last_ = std::chrono::steady_clock::now();
  auto lap_ = [&](const char* what) { if (!timed_) return; const auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "lpmp: make_schedule %-8s %.0f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last_).count()); t_last_ = now; };
  int64_t N = 0;
  for (const auto& sg : segs) N += sg.n;
  struct Upd { int32_t f; int32_t owner; int64_t om, mk; };   // om / mk: absolute pointers are per segment
  std::vector<int32_t> uf(N), owner(N), level(N, 0);
  std::vector<const double*> uom(N);
  std::vector<const uint8_t*> umk(N);
  std::vector<int32_t> n_recv_of(N, 0), n_send_of(N, 0);       // active ops accumulated on the owner record
  {
    int64_t u = 0;
    for (const auto& sg : segs)
      for (int64_t i = 0; i < sg.n; ++i, ++u) {
        const int32_t f = sg.factors[i];
        if (f < 0 || f >= nf) fail("factor index out of range");
        if (sg.om_off[i + 1] - sg.om_off[i] != row_sends(f) || sg.mk_off[i + 1] - sg.mk_off[i] != row_receives(f))
          fail("row " + std::to_string(i) + ": omega / receive mask length does not match the factor's messages");
        uf[u] = f; uom[u] = sg.om + sg.om_off[i]; umk[u] = sg.mk + sg.mk_off[i];
      }
  }
  if (N > std::numeric_limits<int32_t>::max()) fail("too many updates for one schedule");
  // batch-capable message ops: the weights the individual sends end up with (CallSendMessages' batch rule)
  std::vector<double> eff_store;
  if (any_batch) {
    size_t total = 0;
    for (int64_t u = 0; u < N; ++u) total += (size_t)row_sends(uf[u]);
    eff_store.resize(total + 1);
    size_t at = 0;
    for (int64_t u = 0; u < N; ++u) {
      const int64_t ns = row_sends(uf[u]);
      for (int64_t k = 0; k < ns; ++k) if (uom[u][k] < 0) fail("negative send weight");
      effective_send_weights(uf[u], uom[u], eff_store.data() + at);
      uom[u] = eff_store.data() + at;
      at += (size_t)ns;
    }
  }
  lap_("rows");
  std::vector<int32_t> last_level(nf, 0), last_toucher(nf, -1), last_update_of(nf, -1);
  std::vector<int32_t> nr_of_u(N, 0), ns_of_u(N, 0);            // active receives / sends of every single update (its own, not the owner's sum)
  int32_t max_level = 0;
  // what every update touches (its own factor first, then the peers of its active messages) — independent per update, so the
  // walk over the message lists and the weight / mask rows runs on the planner's threads; the recurrence over the levels below
  // is sequential by nature and only chases these lists
  std::vector<int64_t> t_off((size_t)N + 1, 0);
  parallel_chunks(N, 65536, [&](int64_t u_begin, int64_t u_end, int) {
    for (int64_t u = u_begin; u < u_end; ++u) {
      const int32_t f = uf[u];
      int32_t nr = 0, ns = 0; int64_t nt = 1;
      int64_t ks = 0, kr = 0;
      const bool all = ftype_primal[f_type[f]] && f_kind[f] != LPMP_F_VECTOR;
      for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
        const MsgEntry& e = fm[j];
        bool active = false;
        if (e.receives && umk[u][kr++]) { active = true; ++nr; }
        if (e.sends) { const double w = uom[u][ks++]; if (w < 0) fail("negative send weight"); if (w != 0.0) { active = true; ++ns; } }
        if (active || all) ++nt;
      }
      nr_of_u[u] = nr; ns_of_u[u] = ns; t_off[(size_t)u + 1] = nt;
    }
  });
  for (int64_t u = 0; u < N; ++u) t_off[(size_t)u + 1] += t_off[(size_t)u];
  std::vector<int32_t, default_init_allocator<int32_t>> t_data((size_t)t_off[(size_t)N]);
  parallel_chunks(N, 65536, [&](int64_t u_begin, int64_t u_end, int) {
    for (int64_t u = u_begin; u < u_end; ++u) {
      const int32_t f = uf[u];
      int32_t* out_t = t_data.data() + t_off[(size_t)u];
      *out_t++ = f;
      int64_t ks = 0, kr = 0;
      // a pairwise factor that rounds itself reads and writes the labels of ALL its unaries in a primal pass
      // (engine.cpp, ensure_primal), whether or not the message is active in this sweep
      const bool all = ftype_primal[f_type[f]] && f_kind[f] != LPMP_F_VECTOR;
      for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
        const MsgEntry& e = fm[j];
        bool active = false;
        if (e.receives && umk[u][kr++]) active = true;
        if (e.sends && uom[u][ks++] != 0.0) active = true;
        if (active || all) *out_t++ = e.adjacent;
      }
    }
  });
  struct Touched { const int32_t* b; const int32_t* e; const int32_t* begin() const { return b; } const int32_t* end() const { return e; } };
  for (int64_t u = 0; u < N; ++u) {
    const int32_t f = uf[u];
    const Touched touched{t_data.data() + t_off[(size_t)u], t_data.data() + t_off[(size_t)u + 1]};
    const int32_t nr = nr_of_u[u], ns = ns_of_u[u];
    int32_t lv = 0;
    for (int32_t g : touched) lv = std::max(lv, last_level[g]);
    const int32_t prev = last_update_of[f];
    // u2 must not receive: the packed kernels request the vectors of several receives of a record at once, so a
    // record may not receive through the same message twice (forward and backward masks can both select it)
    bool merge = fuse && prev >= 0 && nr == 0 && ns > 0 && n_send_of[prev] == 0 && level[prev] == lv &&
                 n_recv_of[prev] + ns <= 32000;
    if (merge)
      for (int32_t g : touched)
        if (last_level[g] == lv && last_toucher[g] != prev) { merge = false; break; }
    if (merge) {
      owner[u] = prev; level[u] = lv;
      n_recv_of[prev] += nr; n_send_of[prev] += ns;
      for (int32_t g : touched) { last_level[g] = lv; last_toucher[g] = prev; }
    } else {
      owner[u] = (int32_t)u; level[u] = lv + 1;
      n_recv_of[u] = nr; n_send_of[u] = ns;
      max_level = std::max(max_level, level[u]);
      // an update without active ops touches nothing and is dropped, unless its factor type computes a primal:
      // that record stays (it rounds the label in primal passes) and reads / writes its own duals
      if (nr + ns > 0 || ftype_primal[f_type[f]]) {
        for (int32_t g : touched) { last_level[g] = level[u]; last_toucher[g] = (int32_t)u; }
        last_update_of[f] = (int32_t)u;
      }
    }
  }
  lap_("levels");
  out.n_levels = max_level;
  if (levels_only) {
    levels_only->resize((size_t)N);
    for (int64_t u = 0; u < N; ++u) {
      const int32_t o = owner[u];
      (*levels_only)[(size_t)u] = (n_recv_of[o] + n_send_of[o] > 0 || ftype_primal[f_type[uf[o]]]) ? level[o] : 0;
    }
    return;
  }

  // records of the owners, ops = all receives of the members (sequence order), then all sends
  std::vector<int64_t> op_start(N + 1, 0);
  for (int64_t u = 0; u < N; ++u) op_start[u + 1] = op_start[u] + (owner[u] == u ? n_recv_of[u] + n_send_of[u] : 0);
  if (op_start[N] > std::numeric_limits<int32_t>::max()) fail("too many active message operations for one schedule");
  OpVec ops((size_t)op_start[N]);                   // (every slot is written below: the counts are the same walk over the rows)
  // where every update writes inside its owner's op range: receives of the members in sequence order, then the sends
  std::vector<int32_t> r_at(N, 0), s_at(N, 0);
  {
    std::vector<int32_t> cur_r(N, 0), cur_s(N, 0);
    for (int64_t u = 0; u < N; ++u) { const int32_t o = owner[u]; r_at[u] = cur_r[o]; cur_r[o] += nr_of_u[u]; s_at[u] = cur_s[o]; cur_s[o] += ns_of_u[u]; }
  }
  std::vector<int64_t> rec_bytes(N, 0);
  std::vector<uint8_t> all_dense(N, 1), all_potts(N, 1);     // exact classes: every peer L x L, L the own label count
  std::vector<uint8_t> var_dense(N, 1), var_potts(N, 1);     // padded classes: runtime dims
  std::vector<uint8_t> up_any(N, 1);                         // streaming class: dense and Potts peers mixed
  std::vector<uint8_t> small_ok(N, 1);                       // lane-per-factor class: every size <= SMALL_MAXD
  std::vector<uint8_t> pw_right(N, 1);                       // updated dense pairwise factor, every op unary-pairwise with the factor on the right
  std::vector<int32_t> max_dim(N, 0);                        // largest peer table dim of the record
  std::vector<int64_t> alg_bytes_of_thread(PLAN_MAX_THREADS, 0);
  // (several updates may share an owner record — folded sweeps — and land on different threads: the per-owner flags only
  // ever go from 1 to 0, sums and maxima are atomic)
  auto clear_flag = [](uint8_t& x) { __atomic_store_n(&x, (uint8_t)0, __ATOMIC_RELAXED); };
  auto atomic_max = [](int32_t& x, int32_t v) { int32_t cur = __atomic_load_n(&x, __ATOMIC_RELAXED); while (cur < v && !__atomic_compare_exchange_n(&x, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} };
  parallel_chunks(N, 65536, [&](int64_t u_begin, int64_t u_end, int thread) {
  int64_t alg_local = 0;
  for (int64_t u = u_begin; u < u_end; ++u) {
    const int32_t f = uf[u];
    const int32_t o = owner[u];
    const int32_t own_d0 = f_dim0[f];
    if (f_doff[f + 1] - f_doff[f] > SMALL_MAXD) clear_flag(small_ok[o]);   // also for a record without any op (COMPUTE_PRIMAL types)
    Op* base = ops.data() + op_start[o];
    auto fill = [&](const MsgEntry& e, double w) {
      const auto& mt = mtypes[m_type[e.msg]];
      const int32_t peer = e.adjacent;
      Op op{};
      op.peer_dual = doff(peer);
      op.omega = w;
      op.peer = peer;
      op.len = f_dim0[m_left[e.msg]];
      op.pd0 = f_dim0[peer]; op.pd1 = f_dim1[peer];
      const int32_t right = m_right[e.msg];
      int side = 0, imp = 0;
      if (mt.kind == LPMP_M_UNARY_PAIRWISE) {
        side = mt.param;
        op.peer_const = e.role == 0 ? coff(peer) : -1;
        if (!(f_kind[f] == LPMP_F_VECTOR && e.role == 0 && f_kind[peer] == LPMP_F_PAIRWISE_DENSE && f_dim0[peer] == own_d0 && f_dim1[peer] == own_d0 && (coff(peer) % 2) == 0)) clear_flag(all_dense[o]);
        if (!(f_kind[f] == LPMP_F_VECTOR && e.role == 0 && f_kind[peer] == LPMP_F_PAIRWISE_POTTS && f_dim0[peer] == own_d0)) clear_flag(all_potts[o]);
        if (!(f_kind[f] == LPMP_F_VECTOR && e.role == 0 && f_kind[peer] == LPMP_F_PAIRWISE_DENSE && (side == 0 ? f_dim0[peer] : f_dim1[peer]) == own_d0)) clear_flag(var_dense[o]);
        if (!(f_kind[f] == LPMP_F_VECTOR && e.role == 0 && f_kind[peer] == LPMP_F_PAIRWISE_POTTS && f_dim0[peer] == own_d0 && f_dim1[peer] == own_d0)) clear_flag(var_potts[o]);
        atomic_max(max_dim[o], std::max(f_dim0[peer], f_dim1[peer]));
        if (!(f_kind[f] == LPMP_F_VECTOR && e.role == 0 && (f_kind[peer] == LPMP_F_PAIRWISE_DENSE || f_kind[peer] == LPMP_F_PAIRWISE_POTTS) &&
              (side == 0 ? f_dim0[peer] : f_dim1[peer]) == own_d0)) clear_flag(up_any[o]);
      } else {
        clear_flag(all_dense[o]); clear_flag(all_potts[o]); clear_flag(var_dense[o]); clear_flag(var_potts[o]); clear_flag(up_any[o]);
        if (mt.kind == LPMP_M_LABELING) {
          op.peer_const = tab_off[mt.param];
          op.pd1 = tab_nleft[mt.param];
          imp = (f_flags[right] & LPMP_FF_IMPLICIT_ORIGIN) ? 1 : 0;
        }
      }
      op.info = mt.kind | (e.role << 4) | (side << 5) | (imp << 6) | ((f_flags[peer] & LPMP_FF_IMPLICIT_ORIGIN) ? 1 << 7 : 0) | (f_kind[peer] << 8) |
                ((mt.flags & LPMP_MF_IMPROVEMENT) ? OP_HAS_IMPROVEMENT : 0);
      if (std::max(op.len, std::max(op.pd0, op.pd1)) > SMALL_MAXD || f_doff[f + 1] - f_doff[f] > SMALL_MAXD) clear_flag(small_ok[o]);
      if (!(mt.kind == LPMP_M_UNARY_PAIRWISE && e.role == 1 && f_kind[f] != LPMP_F_VECTOR && f_kind[peer] == LPMP_F_VECTOR &&
            f_dim1[f] > 0 && op.len == (side == 0 ? f_dim0[f] : f_dim1[f]))) clear_flag(pw_right[o]);
      return op;
    };
    // algorithmic bytes, counted per update as the reference executes it: own dual read + written
    // once, per receive the peer's table and both message vectors read and one written, per send one peer
    // vector read and written
    auto op_bytes = [&](const Op& op, bool recv) -> int64_t {
      const int code = op.info & 15, pk = (op.info >> 8) & 15;
      if (code == LPMP_M_UNARY_PAIRWISE) {
        const int64_t L = op.len;
        if (recv) return 24 * L + (pk == LPMP_F_PAIRWISE_DENSE ? 8 * (int64_t)op.pd0 * op.pd1 : (pk == LPMP_F_PAIRWISE_POTTS ? 8 : 0));
        return 16 * L;
      }
      return 16 * (int64_t)op.pd0;
    };
    int64_t ks = 0, kr = 0, bytes = 0, n_act = 0;
    int32_t at_r = r_at[u], at_s = n_recv_of[o] + s_at[u];
    for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
      const MsgEntry& e = fm[j];
      if (e.receives && umk[u][kr++]) { Op op = fill(e, 1.0); bytes += op_bytes(op, true); base[at_r++] = op; ++n_act; }
    }
    for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
      const MsgEntry& e = fm[j];
      if (e.sends) { const double w = uom[u][ks++]; if (w != 0.0) { Op op = fill(e, w); bytes += op_bytes(op, false); base[at_s++] = op; ++n_act; } }
    }
    if (n_act > 0) bytes += 16 * (f_doff[f + 1] - f_doff[f]);
    // an updated dense pairwise factor reads its own table once to compute the min-marginals it sends
    if (ks > 0 && f_kind[f] == LPMP_F_PAIRWISE_DENSE) {
      bool sends_any = false;
      for (int64_t j = 0; j < ks; ++j) if (uom[u][j] != 0.0) { sends_any = true; break; }
      if (sends_any) bytes += 8 * (int64_t)f_dim0[f] * f_dim1[f];
    }
    __atomic_fetch_add(&rec_bytes[o], bytes, __ATOMIC_RELAXED);
    alg_local += bytes;
  }
  alg_bytes_of_thread[(size_t)thread] = alg_local;
  });
  for (int64_t b : alg_bytes_of_thread) out.alg_bytes += b;
  lap_("ops");
  // Records with two receives, or two sends, into ONE vector (duplicate messages between the same two factors) need an
  // op-by-op kernel: the packed kernels request a record's vectors before reducing.  They get a class of their own
  // (the streaming / generic / lane-per-factor kernels work op by op), so that one such record does not take its whole
  // launch off the packed kernels.  (Two pairwise factors between the same two variables are NOT this case: their
  // messages go to different vectors.)
  std::vector<uint8_t> dup_vec(N, 0);
  for (int64_t u = 0; u < N; ++u) {
    if (owner[u] != u) continue;
    const Op* o = ops.data() + op_start[u];
    const int nr = n_recv_of[u], ns = n_send_of[u];
    auto same = [&](int a, int b) { return o[a].peer_dual == o[b].peer_dual && ((o[a].info >> 5) & 1) == ((o[b].info >> 5) & 1); };
    for (int a = 0; a < nr && !dup_vec[u]; ++a) for (int b = a + 1; b < nr; ++b) if (same(a, b)) { dup_vec[u] = 1; break; }
    for (int a = nr; a < nr + ns && !dup_vec[u]; ++a) for (int b = a + 1; b < nr + ns; ++b) if (same(a, b)) { dup_vec[u] = 1; break; }
  }
  lap_("dup");
  // bucket the owner records by (level, class); updates without any active op are dropped
  std::vector<int32_t> kclass(N, KC_GENERIC);
  auto cls_of = [&](int64_t u) -> int32_t {
    const int d0 = f_dim0[uf[u]];
    if (force_generic) return small_ok[u] && n_send_of[u] <= SMALL_MAXD ? KC_SMALL : KC_GENERIC;
    if (dup_vec[u] && f_kind[uf[u]] == LPMP_F_VECTOR) {
      if (small_ok[u]) return KC_SMALL;
      const int wd = std::max(d0, max_dim[u]);
      return up_any[u] && wd >= 1 && wd <= BIG_MAX_LABELS ? KC_DENSE_BIG : KC_GENERIC;
    }
    if (f_kind[uf[u]] != LPMP_F_VECTOR) {                  // updated pairwise factors
      if (small_ok[u]) return KC_SMALL;
      const int w = std::max(f_dim0[uf[u]], f_dim1[uf[u]]);
      if (pw_right[u] && w <= 32 && n_recv_of[u] + n_send_of[u] <= PW_MAX_OPS)
        return KC_PW_4 + (w <= 4 ? 0 : w <= 8 ? 1 : w <= 16 ? 2 : 3);
      return KC_GENERIC;
    }
    const bool pow = d0 == 4 || d0 == 8 || d0 == 16 || d0 == 32;
    // more ops than the LDS slab of a lane group holds (a hub of a random graph: C4 has a few 30-neighbour variables among
    // 2 M): such a RECORD goes to the op-by-op streaming kernel — left in its class it took its whole launch off the packed
    // kernels (21 of C4's 44 launches per pass ran on sweep_dense_kernel<16>: 5.5 of 12.6 ms, profiles/r03_c4a_*)
    if (pow && (all_dense[u] || all_potts[u]) && n_recv_of[u] + n_send_of[u] > (all_dense[u] ? pk_dense_cap(d0) : pk_indirect_cap(d0)) && up_any[u]) return KC_DENSE_BIG;
    if (pow && all_dense[u]) return d0 == 4 ? KC_DENSE_4 : d0 == 8 ? KC_DENSE_8 : d0 == 16 ? KC_DENSE_16 : KC_DENSE_32;
    if (pow && all_potts[u]) return d0 == 4 ? KC_POTTS_4 : d0 == 8 ? KC_POTTS_8 : d0 == 16 ? KC_POTTS_16 : KC_POTTS_32;
    const int w = std::max(d0, max_dim[u]);
    if (w < 1) return small_ok[u] ? KC_SMALL : KC_GENERIC;      // no unary-pairwise peer at all
    if (w > 32) return up_any[u] && w <= BIG_MAX_LABELS ? KC_DENSE_BIG : KC_GENERIC;
    const int slot = w <= 4 ? 0 : w <= 8 ? 1 : w <= 16 ? 2 : 3;
    if (var_dense[u]) return KC_DENSE_V4 + slot;
    if (var_potts[u]) return KC_POTTS_V4 + slot;
    if (up_any[u]) return KC_DENSE_V4 + slot;               // unaries with both dense and Potts edges: Potts tables made up in registers
    return small_ok[u] ? KC_SMALL : KC_GENERIC;
  };
  // a COMPUTE_PRIMAL factor is updated even without any active message (FactorUpdated, reference
  // factors_messages.hxx:3125-3130): the primal passes round its label
  auto is_rec = [&](int64_t u) { return owner[u] == u && (n_recv_of[u] + n_send_of[u] > 0 || ftype_primal[f_type[uf[u]]]); };
  // compact keys: only the (level, class) pairs that occur (deep schedules have millions of levels)
  static_assert(KC_COUNT <= 32, "class mask");
  std::vector<uint32_t> level_mask(max_level + 1, 0);
  for (int64_t u = 0; u < N; ++u) {
    if (!is_rec(u)) continue;
    if (n_recv_of[u] > 32767 || n_send_of[u] > 32767) fail("factor has too many messages");
    kclass[u] = cls_of(u);
    level_mask[level[u] - 1] |= 1u << kclass[u];
  }
  std::vector<int64_t> level_base(max_level + 1, 0);
  for (int64_t l = 0; l < max_level; ++l) level_base[l + 1] = level_base[l] + __builtin_popcount(level_mask[l]);
  const int64_t n_keys = level_base[max_level];
  auto key = [&](int64_t u) {
    const int64_t l = level[u] - 1;
    return level_base[l] + __builtin_popcount(level_mask[l] & ((1u << kclass[u]) - 1u));
  };
  std::vector<int64_t> key_count(n_keys + 1, 0), key_recv(n_keys, 0), key_send(n_keys, 0), key_bytes(n_keys, 0);
  std::vector<int32_t> key_level(n_keys, 0), key_class(n_keys, 0), key_maxdim(n_keys, 0);
  for (int64_t u = 0; u < N; ++u) {
    if (!is_rec(u)) continue;
    const int64_t k = key(u);
    ++key_count[k + 1];
    key_level[k] = level[u]; key_class[k] = kclass[u];
    key_recv[k] += n_recv_of[u]; key_send[k] += n_send_of[u]; key_bytes[k] += rec_bytes[u];
    key_maxdim[k] = std::max(key_maxdim[k], std::max(std::max(f_dim0[uf[u]], f_dim1[uf[u]]), max_dim[u]));
    out.n_recv += n_recv_of[u]; out.n_send += n_send_of[u];
  }
  std::partial_sum(key_count.begin(), key_count.end(), key_count.begin());
  out.recs.resize(key_count[n_keys]);
  std::vector<int32_t> rec_upd(key_count[n_keys]);   // update (position in the sequence) each record stands for
  {
    std::vector<int64_t> cur(key_count.begin(), key_count.end() - 1);
    for (int64_t u = 0; u < N; ++u) {
      if (!is_rec(u)) continue;
      const int32_t f = uf[u];
      UpdRec r{};
      r.dual_off = doff(f);
      r.const_off = f_kind[f] == LPMP_F_VECTOR ? -1 : coff(f);
      r.d0 = f_dim0[f]; r.d1 = f_dim1[f];
      r.op_begin = (int32_t)op_start[u];
      r.n_recv = (int16_t)n_recv_of[u]; r.n_send = (int16_t)n_send_of[u];
      r.factor = f;
      r.kind_flags = f_kind[f] | (f_flags[f] << 4) | (ftype_primal[f_type[f]] ? UPD_PRIMAL : 0);
      rec_upd[cur[key(u)]] = (int32_t)u;
      out.recs[cur[key(u)]++] = r;
    }
  }
  for (int64_t k = 0; k < n_keys; ++k) {
    LevelRange lr;
    lr.kclass = key_class[k]; lr.begin = key_count[k]; lr.end = key_count[k + 1];
    lr.level = key_level[k];
    lr.n_recv = key_recv[k]; lr.n_send = key_send[k]; lr.bytes = key_bytes[k];
    lr.max_dim = key_maxdim[k];
    out.launches.push_back(lr);
  }
  // Inside a launch the order of the records is free (they are independent).  They were placed in SEQUENCE order; what the
  // kernels and the Infinity-Cache ticket orders want is MEMORY order — duals and tables lie in factor insertion order — so that
  // blocks that are near in the list touch tables that are near in HBM, in every step alike.  The two agree for a sweep in
  // insertion order; a backward sweep whose order is the exact reverse of the forward one (a chain of relations through all
  // factors: lpmp_plan_suggest_order's answer) lists every level backwards, and a band order across its steps would need block 0
  // of one step to wait for the last block of the step before (measured: 6.75 instead of 5.25 ms per pass on the headline grid).
  // Records of a launch that are not ascending in the factor index are put in that order (reversed when exactly descending).
  for (const auto& lr : out.launches) {
    const int64_t nrec = lr.end - lr.begin;
    if (nrec < 2) continue;
    bool asc = true, desc = true;
    for (int64_t i = lr.begin + 1; i < lr.end && (asc || desc); ++i) {
      asc = asc && out.recs[i - 1].factor <= out.recs[i].factor;
      desc = desc && out.recs[i - 1].factor >= out.recs[i].factor;
    }
    if (asc) continue;
    if (desc) {
      std::reverse(out.recs.begin() + lr.begin, out.recs.begin() + lr.end);
      std::reverse(rec_upd.begin() + lr.begin, rec_upd.begin() + lr.end);
      continue;
    }
    std::vector<int64_t> perm((size_t)nrec);
    std::iota(perm.begin(), perm.end(), lr.begin);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) { return out.recs[x].factor < out.recs[y].factor; });
    std::vector<UpdRec> tr(perm.size()); std::vector<int32_t> tu(perm.size());
    for (size_t i = 0; i < perm.size(); ++i) { tr[i] = out.recs[perm[i]]; tu[i] = rec_upd[perm[i]]; }
    std::copy(tr.begin(), tr.end(), out.recs.begin() + lr.begin);
    std::copy(tu.begin(), tu.end(), rec_upd.begin() + lr.begin);
  }
  lap_("records");
  out.ops = std::move(ops);
  // inside a launch the order of the records is free (they are independent): sub-wave kernels run several
  // factors per wavefront, so neighbours in the list should have similar amounts of work.  Sorted inside windows of
  // 1024 records only: the order of the sequence carries the model's locality (rows of a grid), and the Infinity-Cache
  // ticket orders (make_schedule below, engine.cpp rotation_chain) need blocks that are near in the list to be near in
  // the model — sorted globally, the border rows of a grid ended up in the last blocks and no band order was valid
  constexpr int64_t SORT_WINDOW = 1024;
  for (const auto& lr : out.launches)
    if (lr.kclass != KC_GENERIC && lr.kclass != KC_DENSE_32 && lr.kclass != KC_DENSE_V32 && lr.kclass != KC_DENSE_BIG && lr.kclass != KC_PW_32) {   // incl. KC_SMALL
      std::vector<int64_t> perm(lr.end - lr.begin);
      std::iota(perm.begin(), perm.end(), lr.begin);
      // (a launch with many different amounts of work per record — a random graph, degrees 2 ... 25 — has no locality
      // worth keeping and balances better sorted as a whole: C4 12.1 against 13.6 ms per pass)
      std::vector<int32_t> shapes;
      for (int64_t i = lr.begin; i < lr.end && shapes.size() <= 6; ++i) {
        const int32_t sh = out.recs[i].n_recv * 65536 + out.recs[i].n_send;
        if (std::find(shapes.begin(), shapes.end(), sh) == shapes.end()) shapes.push_back(sh);
      }
      const int64_t window = shapes.size() <= 6 ? SORT_WINDOW : std::numeric_limits<int64_t>::max();
      std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) {
        const UpdRec& a = out.recs[x]; const UpdRec& b = out.recs[y];
        const int64_t wx = (x - lr.begin) / window, wy = (y - lr.begin) / window;
        if (wx != wy) return wx < wy;
        return a.n_recv != b.n_recv ? a.n_recv > b.n_recv : a.n_send > b.n_send;
      });
      std::vector<UpdRec> tr(perm.size()); std::vector<int32_t> tu(perm.size());
      for (size_t i = 0; i < perm.size(); ++i) { tr[i] = out.recs[perm[i]]; tu[i] = rec_upd[perm[i]]; }
      std::copy(tr.begin(), tr.end(), out.recs.begin() + lr.begin);
      std::copy(tu.begin(), tu.end(), rec_upd.begin() + lr.begin);
    }
  lap_("sorted");
  // flags of the fast-class records, kept in recs / ops themselves (packets are plain copies)
  static_assert(sizeof(UpdRec) == sizeof(Op), "a packet slot holds either record");
  int64_t pk_total = 0;                      // packet slots of all packed launches (the array is allocated once, below)
  for (auto& lr : out.launches) {
    if (kc_is_pw(lr.kclass)) {               // updated pairwise factors: plain packets (no preload / forwarding flags)
      int kmax = 0;
      for (int64_t i = lr.begin; i < lr.end; ++i) kmax = std::max<int>(kmax, out.recs[i].n_recv + out.recs[i].n_send);
      lr.stride = 1 + kmax;
      lr.pk_begin = pk_total;
      pk_total += (lr.end - lr.begin) * lr.stride;
      continue;
    }
    if (lr.kclass == KC_SMALL) {
      // lane-per-factor records: no send goes to a peer one of the record's receives rewrites -> the level loop's staged
      // labeling body may request the send peers' costs together with the receive peers' (kernels.hip, label_ops_body)
      for (int64_t i = lr.begin; i < lr.end; ++i) {
        UpdRec& r = out.recs[i];
        const Op* o = out.ops.data() + r.op_begin;
        bool ok = true;
        for (int a = 0; a < r.n_recv && ok; ++a)
          for (int b = r.n_recv; b < r.n_recv + r.n_send; ++b) if (o[a].peer_dual == o[b].peer_dual) { ok = false; break; }
        if (ok) r.kind_flags |= UPD_PRELOAD_OK;
        // a send into the peer ONE receive of the record has just rewritten (labeling lists: the middle variables of a
        // triplet): the staged body hands the rewritten costs over in LDS — receive: pad = 1 (no store), send: pad = index
        // of that receive + 1.  Only a hint: the op-by-op bodies store and reload.
        Op* ow = out.ops.data() + r.op_begin;
        for (int b = r.n_recv; b < r.n_recv + r.n_send; ++b) {
          int hit = -1, n_hit = 0, n_same = 0;
          for (int a = 0; a < r.n_recv; ++a) if (ow[a].peer_dual == ow[b].peer_dual) { hit = a; ++n_hit; }
          for (int b2 = r.n_recv; b2 < r.n_recv + r.n_send; ++b2) if (ow[b2].peer_dual == ow[b].peer_dual) ++n_same;
          if (n_hit == 1 && n_same == 1 && hit < 8 && ow[hit].peer_const == ow[b].peer_const && ow[hit].pd0 == ow[b].pd0 && ow[hit].pd1 == ow[b].pd1) { ow[hit].pad = 1; ow[b].pad = hit + 1; }
        }
      }
      continue;
    }
    if (lr.kclass == KC_GENERIC || lr.kclass >= KC_DENSE_BIG) continue;   // packed dense and Potts classes
    auto same_vec = [](const Op* o, int a, int b) { return o[a].peer_dual == o[b].peer_dual && ((o[a].info >> 5) & 1) == ((o[b].info >> 5) & 1); };
    if (kc_is_var(lr.kclass)) {
      // the padded classes only exist in packed / indirect form: what those cannot run goes to the streaming kernel
      bool ok = true;
      for (int64_t i = lr.begin; i < lr.end && ok; ++i) {
        const UpdRec& r = out.recs[i];
        const Op* o = out.ops.data() + r.op_begin;
        if (r.n_recv + r.n_send > pk_class_cap(lr.kclass)) ok = false;
        for (int a = 0; a < r.n_recv && ok; ++a)
          for (int a2 = a + 1; a2 < r.n_recv; ++a2) if (same_vec(o, a, a2)) { ok = false; break; }
        for (int b = r.n_recv; b < r.n_recv + r.n_send && ok; ++b)
          for (int b2 = b + 1; b2 < r.n_recv + r.n_send; ++b2) if (same_vec(o, b, b2)) { ok = false; break; }
      }
      // (the streaming dense kernel works op by op, so duplicates and any op count are fine for it)
      if (!ok) { lr.kclass = KC_DENSE_BIG; continue; }
    }
    // flags of the records (independent of each other: chunks of the launch on several threads), then the packet stride
    const int64_t n_lr = lr.end - lr.begin;
    std::vector<int> kmax_of(PLAN_MAX_THREADS, 0); std::vector<uint8_t> dup_of(PLAN_MAX_THREADS, 0);
    parallel_chunks(n_lr, 32768, [&](int64_t c0, int64_t c1, int thread) {
      int kmax_l = 0; bool dup_l = false;
      for (int64_t i = lr.begin + c0; i < lr.begin + c1; ++i) {
        UpdRec& r = out.recs[i];
        Op* o = out.ops.data() + r.op_begin;
        kmax_l = std::max<int>(kmax_l, r.n_recv + r.n_send);
        auto same = [&](int a, int b) { return o[a].peer_dual == o[b].peer_dual && ((o[a].info >> 5) & 1) == ((o[b].info >> 5) & 1); };
        bool preload_ok = true;   // a send may be requested early unless a receive of this update writes the same vector
        for (int a = 0; a < r.n_recv && preload_ok; ++a)
          for (int b = r.n_recv; b < r.n_recv + r.n_send; ++b)
            if (same(a, b)) { preload_ok = false; break; }
        if (preload_ok) r.kind_flags |= UPD_PRELOAD_OK;
        // two receives, or two sends, into one vector (duplicate messages between the same two factors)
        for (int a = 0; a < r.n_recv && !dup_l; ++a)
          for (int a2 = a + 1; a2 < r.n_recv; ++a2) if (same(a, a2)) { dup_l = true; break; }
        for (int b = r.n_recv; b < r.n_recv + r.n_send && !dup_l; ++b)
          for (int b2 = b + 1; b2 < r.n_recv + r.n_send; ++b2) if (same(b, b2)) { dup_l = true; break; }
        // register forwarding: send b targets the vector receive a (one of the first 4) has just rewritten ->
        // the receive keeps its result in a register (pad = 1: no store) and the send reads it from there
        // (pad = a + 1); at most one send per receive, and only if no other receive/send touches that vector
        for (int b = r.n_recv; b < r.n_recv + r.n_send && b - r.n_recv < 4; ++b) {
          int hit = -1, n_hit = 0, n_send_same = 0;
          for (int a = 0; a < r.n_recv; ++a) if (same(a, b)) { hit = a; ++n_hit; }
          for (int b2 = r.n_recv; b2 < r.n_recv + r.n_send; ++b2) if (same(b2, b)) ++n_send_same;
          if (n_hit == 1 && n_send_same == 1 && hit < 4) { o[hit].pad = 1; o[b].pad = hit + 1; }
        }
      }
      kmax_of[(size_t)thread] = kmax_l; dup_of[(size_t)thread] = dup_l;
    });
    int kmax = 0; bool dup_recv = false;
    for (int t = 0; t < PLAN_MAX_THREADS; ++t) { kmax = std::max(kmax, kmax_of[(size_t)t]); dup_recv = dup_recv || dup_of[(size_t)t]; }
    if (dup_recv) continue;                  // only the op-by-op kernels are safe for that: stride stays 0
    if (kmax > PK_MAX_OPS) {                 // too many ops for a packet: indirect mode if they fit the LDS slab
      if (kmax <= pk_class_cap(lr.kclass)) lr.stride = -1;
      continue;
    }
    lr.stride = 1 + kmax;
    lr.pk_begin = pk_total;                  // (filled below, when the size of the whole array is known)
    pk_total += n_lr * lr.stride;
  }
  lap_("pk-flags");
  // packets: ONE allocation, never zero-filled (every slot is written: the record, its ops, zeros behind them)
  out.packets.resize((size_t)pk_total);
  for (const auto& lr : out.launches) {
    if (lr.stride <= 0) continue;
    parallel_chunks(lr.end - lr.begin, 32768, [&](int64_t c0, int64_t c1, int) {
      for (int64_t i = lr.begin + c0; i < lr.begin + c1; ++i) {
        Op* slot = out.packets.data() + lr.pk_begin + (i - lr.begin) * lr.stride;
        const UpdRec& r = out.recs[i];
        std::memcpy(slot, &r, sizeof(Op));
        const int n = r.n_recv + r.n_send;
        for (int k = 0; k < n; ++k) slot[1 + k] = out.ops[r.op_begin + k];
        if (n + 1 < lr.stride) std::memset((void*)(slot + 1 + n), 0, (size_t)(lr.stride - 1 - n) * sizeof(Op));
      }
    });
  }
  lap_("packets");
  if (!chains && max_level == 3) return;
  // ---- chain plans: a deep schedule becomes persistent launches (kernels.hip, chain executor), one per kernel class.
  // Dependencies: update u must see the results of the last earlier update that touched u's factor or a factor u
  // touches — the same relation the levels were computed from.  Classes are separate launches and cannot wait for each
  // other, so a schedule qualifies only if no dependency runs between records of different classes (C5: the Potts grid
  // and the labeling-list factors are separate components); classes with few launches stay plain launches.
  // (LPMP_CHAIN_MIN: experiments — the smallest number of launches that makes a class a chain)
  const int64_t chain_min = [] { const char* v = std::getenv("LPMP_CHAIN_MIN"); return v ? (int64_t)std::atoll(v) : CHAIN_MIN_LAUNCHES; }();
  const int bands = [] { const char* v = std::getenv("LPMP_CHAIN_BANDS"); return v ? std::atoi(v) : 0; }();
  const int lag = [] { const char* v = std::getenv("LPMP_CHAIN_LAG"); return v ? std::atoi(v) : 2; }();
  const bool no_level_loop = std::getenv("LPMP_NO_LEVEL_LOOP") != nullptr;
  const bool no_auto_bands = std::getenv("LPMP_NO_BLOCKED_PASSES") != nullptr;
  // (LPMP_BAND_MIN_BYTES, LPMP_BAND_BYTES: tests force the banded order on small models)
  const char* bmin_env = std::getenv("LPMP_BAND_MIN_BYTES");
  const int64_t band_min_bytes = bmin_env ? std::atoll(bmin_env) : ((int64_t)64 << 20);
  // a model that fits the 256 MiB Infinity Cache as a whole is re-read on-die by plain launches already
  const bool model_big = bmin_env != nullptr || (f_coff[nf] + f_doff[nf]) * (int64_t)sizeof(double) > ((int64_t)1 << 30);
  const char* bb_env = std::getenv("LPMP_BAND_BYTES");
  const int64_t band_bytes = bb_env ? std::max<int64_t>(1, std::atoll(bb_env)) : ((int64_t)16 << 20);
  const char* chain_all_env = std::getenv("LPMP_CHAIN_ALL");
  const bool chain_all = chain_all_env && std::atoi(chain_all_env) != 0;
  bool any_big = false;
  for (const auto& lr : out.launches) any_big = any_big || (model_big && kc_is_dense(lr.kclass) && !kc_is_var(lr.kclass) && lr.n_recv > 0 && lr.bytes >= band_min_bytes);
  // A long schedule of HEAVY launches (C4 at full size: 66 levels of ~0.8 GB each, 150 - 300 us per launch) gains nothing from a
  // persistent launch — the gaps between its kernels are a per cent of their run time (measured: 12.15 ms per pass as a chain,
  // 11.96 ms of kernel time launch by launch) — while its ticket, dependency and mailbox tables are seconds of planning: it
  // stays a replayed graph of plain launches.  (Few big steps are the banded case below; LPMP_CHAIN_HEAVY_BYTES moves the bar.)
  if (!chain_all && bands <= 1 && out.launches.size() > 8) {
    const char* hv = std::getenv("LPMP_CHAIN_HEAVY_BYTES");
    const int64_t heavy = hv ? std::atoll(hv) : ((int64_t)256 << 20);
    int64_t total = 0;
    for (const auto& lr : out.launches) total += lr.bytes;
    if (heavy > 0 && total / (int64_t)out.launches.size() >= heavy) return;
  }
  if (((int64_t)out.launches.size() >= chain_min || (any_big && out.launches.size() >= 2 && !no_auto_bands)) && !out.launches.empty()) {
    std::vector<int64_t> n_launches_of(KC_COUNT, 0);
    bool ok = true;
    for (const auto& lr : out.launches) {
      n_launches_of[lr.kclass]++;
      ok = ok && kc_chain_capable(lr.kclass) && (kc_width(lr.kclass) == 0 || lr.stride != 0);
      // lane-per-factor and generic records: every dual access of a chain kernel is a device-scope access that goes past
      // the L2, and these bodies issue them one dependent access at a time — measured slower than replaying a hipGraph
      // of plain launches (C5: 202 ms against 188 ms per pass).  The kernels stay available: LPMP_CHAIN_ALL=1
      // Those classes get the level loop instead when their levels are tiny (below).
    }
    if (ok) {
      // tickets per class
      std::vector<ChainPlan> cps(KC_COUNT);
      std::vector<int32_t> ticket_of_update(N, -1), class_of_update(N, -1);
      std::vector<int32_t> t0(KC_COUNT, 0);
      for (size_t li = 0; li < out.launches.size(); ++li) {
        const auto& lr = out.launches[li];
        ChainPlan& cp = cps[lr.kclass];
        cp.kclass = lr.kclass;
        const int gpb = kc_block_records(lr.kclass);
        const int32_t nb = (int32_t)((lr.end - lr.begin + gpb - 1) / gpb);
        cp.launches.push_back({lr.begin, lr.end - lr.begin, lr.pk_begin, lr.stride, t0[lr.kclass]});
        for (int32_t b = 0; b < nb; ++b) { cp.tk_launch.push_back((int32_t)cp.launches.size() - 1); cp.tk_block.push_back(b); }
        for (int64_t i = lr.begin; i < lr.end; ++i) { ticket_of_update[rec_upd[i]] = t0[lr.kclass] + (int32_t)((i - lr.begin) / gpb); class_of_update[rec_upd[i]] = lr.kclass; }
        t0[lr.kclass] += nb;
        if ((int64_t)t0[lr.kclass] + nb > std::numeric_limits<int32_t>::max() / 2) ok = false;
      }
      // ---- mailbox (kernels.hip, dense_pk_body): in a deep chain of a dense class the vector a send writes is what the
      // neighbour's receive one level later waits for.  Through the completion flag that hand-over costs two trips (flag seen,
      // then the vector fetched); a send therefore ALSO writes its vector as tagged granules into a mailbox row, the receive
      // polls that row instead of the dual array, and the dependency between the two tickets needs no flag.
      // src_rec / src_k: per receive op, the record and send index that LAST wrote the vector the receive reads (the other
      // side of the pairwise factor) — by vector, not by factor: a record that synchronised on a granule has not seen the
      // producer's ticket complete, so no one may read that producer's vector from the dual array on its word.
      const bool no_mailbox = std::getenv("LPMP_NO_MAILBOX") != nullptr;
      std::vector<char> mbox_class(KC_COUNT, 0);
      std::vector<int32_t> src_rec, rec_of_upd, rec_launch;
      std::vector<int8_t> src_k;
      bool any_mbox = false;
      if (!no_mailbox && bands <= 1 && ok) {
        for (int c = 0; c < KC_COUNT; ++c) {
          if (!(c >= KC_DENSE_4 && c <= KC_POTTS_V32)) continue;                   // the packed dense and Potts classes, exact and run-time dims
          // (fewer launches: plain launches, or — a few HBM-sized steps — the banded order below; LPMP_CHAIN_MIN lowers the
          // bar for the randomised tests, which then run the mailbox on every small chain)
          bool el = n_launches_of[c] >= chain_min && !(model_big && n_launches_of[c] <= 8 && !no_auto_bands);
          for (const auto& lr : out.launches) if (lr.kclass == c && lr.stride <= 0) el = false;
          mbox_class[c] = el; any_mbox = any_mbox || el;
        }
      }
      if (any_mbox) {
        src_rec.assign(out.ops.size(), -1); src_k.assign(out.ops.size(), -1);
        rec_of_upd.assign(N, -1); rec_launch.assign(out.recs.size(), -1);
        std::vector<int32_t> lw_rec((size_t)2 * nf, -1);      // last writer of (factor, side): record ...
        std::vector<int8_t> lw_k((size_t)2 * nf, -1);         // ... and its send index (-1: written by a receive)
        for (size_t li = 0; li < out.launches.size(); ++li) {
          const auto& lr = out.launches[li];
          for (int64_t i = lr.begin; i < lr.end; ++i) { rec_of_upd[rec_upd[i]] = (int32_t)i; rec_launch[i] = (int32_t)li; }
          if (!mbox_class[lr.kclass]) continue;
          if (MAILBOX_SENDS < 4)                    // the mailbox body forwards fewer results in registers: take the other hints back
            for (int64_t i = lr.begin; i < lr.end; ++i) {
              const UpdRec& r = out.recs[i];
              Op* o = out.ops.data() + r.op_begin;
              Op* pk = out.packets.data() + lr.pk_begin + (i - lr.begin) * lr.stride + 1;
              for (int b = r.n_recv; b < r.n_recv + r.n_send; ++b) {
                const int hit = o[b].pad - 1;
                if (hit >= 0 && (hit >= MAILBOX_SENDS || b - r.n_recv >= MAILBOX_SENDS)) {
                  o[hit].pad = 0; o[b].pad = 0; pk[hit].pad = 0; pk[b].pad = 0;
                  out.recs[i].kind_flags &= ~UPD_PRELOAD_OK;      // (it was clear already: the send targets what a receive writes)
                  Op* hdr = pk - 1; UpdRec rr; std::memcpy(&rr, hdr, sizeof(rr)); rr.kind_flags &= ~UPD_PRELOAD_OK; std::memcpy(hdr, &rr, sizeof(rr));
                }
              }
            }
          for (int64_t i = lr.begin; i < lr.end; ++i) {
            const UpdRec& r = out.recs[i];
            const Op* o = out.ops.data() + r.op_begin;
            for (int j = 0; j < r.n_recv; ++j) {
              const int64_t v = (int64_t)2 * o[j].peer + (1 - ((o[j].info >> 5) & 1));
              const int32_t w = lw_rec[v];
              if (w >= 0 && lw_k[v] >= 0 && lw_k[v] < MAILBOX_SENDS && out.launches[rec_launch[w]].kclass == lr.kclass) { src_rec[r.op_begin + j] = w; src_k[r.op_begin + j] = lw_k[v]; }
            }
            for (int j = 0; j < r.n_recv + r.n_send; ++j) {
              const int64_t v = (int64_t)2 * o[j].peer + ((o[j].info >> 5) & 1);
              lw_rec[v] = (int32_t)i; lw_k[v] = j < r.n_recv ? (int8_t)-1 : (int8_t)std::min(j - r.n_recv, 127);
              // a send after a receive of the same record through the same factor whose result was NOT handed over in a
              // register: that receive stored the factor's tracked bound, and the reader's own store of that bound is not
              // ordered after it by a granule -> flag
              if (j >= r.n_recv && o[j].pad == 0)
                for (int a = 0; a < r.n_recv; ++a) if (o[a].peer == o[j].peer) lw_k[v] = -1;
            }
          }
        }
      }
      if (any_mbox && mailbox_budget_bytes >= 0) {
        // rows of a class <= its receives with a mailbox source (every row is polled by at least one of them): a class whose
        // mailbox would not fit the budget keeps its completion flags — decided HERE, before any dependency is dropped
        std::vector<int64_t> rows_upper(KC_COUNT, 0);
        for (size_t li = 0; li < out.launches.size(); ++li) {
          const auto& lr = out.launches[li];
          if (!mbox_class[lr.kclass]) continue;
          for (int64_t i = lr.begin; i < lr.end; ++i) {
            const UpdRec& r = out.recs[i];
            for (int j = 0; j < r.n_recv; ++j) if (src_rec[r.op_begin + j] >= 0) ++rows_upper[lr.kclass];
          }
        }
        int64_t left = mailbox_budget_bytes;
        for (int c = 0; c < KC_COUNT; ++c) {
          if (!mbox_class[c]) continue;
          const int64_t bytes = rows_upper[c] * (int64_t)kc_width(c) * 16;
          if (bytes > left) mbox_class[c] = 0; else left -= bytes;
        }
      }
      // replay the sequence: who touched each factor last
      std::vector<int32_t> toucher(nf, -1);
      std::vector<std::vector<std::pair<int32_t, int32_t>>> edges(KC_COUNT);     // per class: (ticket, predecessor ticket)
      for (int64_t u = 0; u < N && ok; ++u) {
        const int32_t o = owner[u];
        const int32_t tk = ticket_of_update[o];
        if (tk < 0) continue;                             // dropped update (no active message)
        const int32_t f = uf[u];
        auto visit = [&](int32_t g, bool via_message = false) {
          const int32_t w = toucher[g];
          if (w >= 0 && w != o) {
            if (class_of_update[w] != class_of_update[o]) ok = false;      // a dependency between classes
            else if (ticket_of_update[w] != tk) {
              // covered by the mailbox: o receives through g exactly the vector w's send wrote (and w's own reads of g
              // precede that send in w's program order, so what o writes into g cannot overtake them)
              bool covered = false;
              if (via_message && mbox_class[class_of_update[o]]) {
                const UpdRec& r = out.recs[rec_of_upd[o]];
                for (int j = 0; j < r.n_recv; ++j)
                  if (out.ops[r.op_begin + j].peer == g && src_rec[r.op_begin + j] == rec_of_upd[w]) covered = true;
              }
              if (!covered) edges[class_of_update[o]].emplace_back(tk, ticket_of_update[w]);
            }
          }
          toucher[g] = o;
        };
        visit(f);
        int64_t ks = 0, kr = 0;
        for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) {
          const MsgEntry& e = fm[j];
          bool active = false;
          if (e.receives && umk[u][kr++]) active = true;
          if (e.sends && uom[u][ks++] != 0.0) active = true;
          if (active) visit(e.adjacent, true);
        }
      }
      for (int c = 0; c < KC_COUNT && ok; ++c) {
        if (n_launches_of[c] == 0) continue;
        // A few HBM-sized launches of a dense class (the colour steps of a big grid: forward or backward sweep alone,
        // a fused pass in a weight mode that does not rotate, the per-pass schedule of a multi-GPU part) are worth a
        // chain as well: not for the launch gaps but for the ORDER — consecutive steps read the same pairwise tables,
        // and band j of step l issued at time j + lag * l finds them in the 256 MiB Infinity Cache.
        if (kc_width(c) == 0 && !chain_all) {
          // Many TINY levels of the lane-per-factor / generic class (C5 with local triples: 11 887 levels of a dozen
          // one-lane updates): one workgroup walks the levels with a workgroup barrier in between — no launch per level,
          // no flags through memory, and the duals it hands from level to level stay in its L2.  Wide levels stay plain.
          int64_t recs_c = 0;
          for (const auto& lr : out.launches) if (lr.kclass == c) recs_c += lr.end - lr.begin;
          if (n_launches_of[c] >= chain_min && recs_c <= (int64_t)kc_block_records(c) * n_launches_of[c] && !no_level_loop) {
            ChainPlan& lp = cps[c];
            lp.level_loop = true; lp.valid = true;
            int dbg_left = 5;
            if (c == KC_SMALL)
              for (auto& cl : lp.launches) {           // launches the op-parallel labeling body can run (kernels.hip, label_ops_body)
                bool fine = true, paired = true;
                for (int64_t i = cl.rec_begin; i < cl.rec_begin + cl.count && fine; ++i) {
                  const UpdRec& r = out.recs[i];
                  const Op* o = out.ops.data() + r.op_begin;
                  const int n = r.n_recv + r.n_send;
                  fine = r.n_recv <= 8 && r.n_send <= 8 && (r.kind_flags & 15) == LPMP_F_VECTOR && r.d0 <= SMALL_MAXD;
                  for (int a = 0; a < n && fine; ++a) {
                    if ((o[a].info & 15) != OP_LABELING || ((o[a].info >> 4) & 1) != 0 || o[a].pd0 > SMALL_MAXD || o[a].len != r.d0 || o[a].pd1 != r.d0) fine = false;   // message length = label count of the table = the factor's size
                    // the receives run side by side, and so do the sends: no two of a kind on one peer
                    for (int b = a + 1; b < n && fine; ++b) if (o[a].peer_dual == o[b].peer_dual && (a < r.n_recv) == (b < r.n_recv)) fine = false;
                  }
                  if (r.n_recv != r.n_send) paired = false;
                  for (int a = 0; a < r.n_recv && paired && fine; ++a)
                    if (o[a].peer_dual != o[r.n_recv + a].peer_dual || o[a].peer_const != o[r.n_recv + a].peer_const || o[a].pd0 != o[r.n_recv + a].pd0) paired = false;
                }
                if (fine) cl.flags |= CHAIN_LAUNCH_LABEL_OPS | (paired ? CHAIN_LAUNCH_LABEL_PAIRED : 0);
                else if (std::getenv("LPMP_ROT_VERBOSE") && dbg_left-- > 0) {
                  const UpdRec& r = out.recs[cl.rec_begin]; const Op* o = out.ops.data() + r.op_begin;
                  std::fprintf(stderr, "lpmp:   not eligible: first record kind %d d0 %d ops %d+%d; op0 code %d role %d pd0 %d pd1 %d len %d\n", r.kind_flags & 15, r.d0, r.n_recv, r.n_send,
                               (r.n_recv + r.n_send) ? (o[0].info & 15) : -1, (r.n_recv + r.n_send) ? ((o[0].info >> 4) & 1) : -1, (r.n_recv + r.n_send) ? o[0].pd0 : -1, (r.n_recv + r.n_send) ? o[0].pd1 : -1, (r.n_recv + r.n_send) ? o[0].len : -1);
                }
              }
            if (std::getenv("LPMP_ROT_VERBOSE")) {
              int64_t nf_ = 0, np_ = 0; for (const auto& cl : lp.launches) { nf_ += (cl.flags & CHAIN_LAUNCH_LABEL_OPS) != 0; np_ += (cl.flags & CHAIN_LAUNCH_LABEL_PAIRED) != 0; }
              std::fprintf(stderr, "lpmp: level loop over %zu launches of class %d, %lld of them with one lane per op (%lld paired)\n", lp.launches.size(), c, (long long)nf_, (long long)np_);
            }
            lp.tk_launch.clear(); lp.tk_block.clear(); lp.dep_off.assign(1, 0); lp.dep.clear();
            out.chains.push_back(std::move(lp));
          } else {
            for (size_t li = 0; li < out.launches.size(); ++li) if (out.launches[li].kclass == c) out.plain_launches.push_back((int32_t)li);
          }
          continue;
        }
        // (only receives read tables: a directional sweep of a 2-colour grid has ONE such step and gains nothing)
        int64_t max_bytes = 0; int n_table_steps = 0;
        for (const auto& lr : out.launches) if (lr.kclass == c) { max_bytes = std::max(max_bytes, lr.bytes); if (lr.n_recv > 0 && lr.bytes >= band_min_bytes) ++n_table_steps; }
        const bool dense_cls = kc_is_dense(c) && !kc_is_var(c);   // (run-time-dims classes: slower as a banded chain, engine.cpp plan_rotation_chain)
        const bool big_steps = model_big && dense_cls && n_table_steps >= 2 && n_launches_of[c] <= 8 && !no_auto_bands;
        if (n_launches_of[c] < chain_min && !big_steps && !(bands > 1)) {               // few launches: plain
          for (size_t li = 0; li < out.launches.size(); ++li) if (out.launches[li].kclass == c) out.plain_launches.push_back((int32_t)li);
          continue;
        }
        ChainPlan& cp = cps[c];
        auto& ed = edges[c];
        const int64_t n_tickets = (int64_t)cp.tk_launch.size();
        if (big_steps && !(bands > 1) && n_tickets > 0) {
          // about 16 MiB of algorithmic bytes per band; the smallest lag from 3 on that keeps every dependency backwards
          const int64_t nl = (int64_t)cp.launches.size();
          const int gpb = kc_block_records(c);
          int64_t max_nb = 1;                               // (a band narrower than a few blocks cannot keep the dependencies)
          for (const auto& l : cp.launches) max_nb = std::max<int64_t>(max_nb, (l.count + gpb - 1) / gpb);
          const int nbands = (int)std::max<int64_t>(2, std::min<int64_t>(max_bytes / band_bytes, max_nb / 4));
          std::vector<int32_t> order((size_t)n_tickets), new_of((size_t)n_tickets);
          std::vector<int64_t> key((size_t)n_tickets);
          for (int lg = 3; lg <= 16 && !cp.banded; ++lg) {
            for (int64_t t = 0; t < n_tickets; ++t) {
              const int64_t l = cp.tk_launch[t];
              const int64_t nb = (cp.launches[l].count + gpb - 1) / gpb;
              key[t] = ((int64_t)cp.tk_block[t] * nbands / nb + (int64_t)lg * l) * (nl + 1) + l;
            }
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
            for (int64_t i = 0; i < n_tickets; ++i) new_of[order[i]] = (int32_t)i;
            bool fine = true;
            for (const auto& e : ed) if (new_of[e.second] >= new_of[e.first]) { fine = false; break; }
            if (!fine) continue;
            for (auto& e : ed) { e.first = new_of[e.first]; e.second = new_of[e.second]; }
            std::vector<int32_t> tl((size_t)n_tickets), tb((size_t)n_tickets);
            for (int64_t i = 0; i < n_tickets; ++i) { tl[i] = cp.tk_launch[order[i]]; tb[i] = cp.tk_block[order[i]]; }
            cp.tk_launch.swap(tl); cp.tk_block.swap(tb);
            cp.banded = true;
          }
          if (!cp.banded && n_launches_of[c] < chain_min) {   // no valid order and nothing else to gain: plain launches
            for (size_t li = 0; li < out.launches.size(); ++li) if (out.launches[li].kclass == c) out.plain_launches.push_back((int32_t)li);
            continue;
          }
        }
        // Temporal blocking (LPMP_CHAIN_BANDS=NB, experiments): tickets are not taken level by level but in a skewed
        // order — band j of the l-th launch at time j + LAG * l — so that what a level reads (pairwise tables) is read
        // again by the next level while it is still in the Infinity Cache.  Only an order: the dependency flags keep
        // the result identical; an order that would put a dependency behind its dependent is refused.
        if (bands > 1 && n_tickets > 0) {
          std::vector<int64_t> key((size_t)n_tickets);
          const int64_t nl = (int64_t)cp.launches.size();
          for (int64_t t = 0; t < n_tickets; ++t) {
            const int64_t l = cp.tk_launch[t];
            const int gpb = kc_block_records(c);
            const int64_t nb = (cp.launches[l].count + gpb - 1) / gpb;
            key[t] = ((int64_t)cp.tk_block[t] * bands / nb + (int64_t)lag * l) * (nl + 1) + l;
          }
          std::vector<int32_t> order((size_t)n_tickets);
          std::iota(order.begin(), order.end(), 0);
          std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
          std::vector<int32_t> new_of((size_t)n_tickets);
          for (int64_t i = 0; i < n_tickets; ++i) new_of[order[i]] = (int32_t)i;
          for (auto& e : ed) { e.first = new_of[e.first]; e.second = new_of[e.second]; }
          std::vector<int32_t> tl((size_t)n_tickets), tb((size_t)n_tickets);
          for (int64_t i = 0; i < n_tickets; ++i) { tl[i] = cp.tk_launch[order[i]]; tb[i] = cp.tk_block[order[i]]; }
          cp.tk_launch.swap(tl); cp.tk_block.swap(tb);
          cp.banded = true;
        }
        std::sort(ed.begin(), ed.end());
        ed.erase(std::unique(ed.begin(), ed.end()), ed.end());
        cp.dep_off.assign((size_t)n_tickets + 1, 0);
        for (const auto& e : ed) { if (e.second >= e.first) fail("chain plan: dependency on a later ticket (LPMP_CHAIN_BANDS / LPMP_CHAIN_LAG?)"); cp.dep_off[e.first + 1]++; }
        std::partial_sum(cp.dep_off.begin(), cp.dep_off.end(), cp.dep_off.begin());
        cp.dep.resize(ed.size());
        for (size_t i = 0; i < ed.size(); ++i) cp.dep[i] = ed[i].second;   // sorted by ticket: already in CSR order
        cp.valid = true;
        if (mbox_class[c]) {
          if (cp.banded) fail("chain plan: mailbox in a banded order");
          // rows for the sends some receive polls; the packet copies of both ops carry the row (plan.hpp, OP_MAILBOX)
          std::vector<int64_t> row_of_op;                     // per op of out.ops (sends): mailbox row, assigned on first use
          row_of_op.assign(out.ops.size(), -1);
          auto slot_of = [&](int64_t i) { const auto& lr = out.launches[rec_launch[i]]; return out.packets.data() + lr.pk_begin + (i - lr.begin) * lr.stride; };
          int64_t rows = 0;
          for (const auto& cl : cp.launches)
            for (int64_t i = cl.rec_begin; i < cl.rec_begin + cl.count; ++i) {
              const UpdRec& r = out.recs[i];
              for (int j = 0; j < r.n_recv; ++j) {
                const int32_t w = src_rec[r.op_begin + j];
                if (w < 0) continue;
                const UpdRec& rw = out.recs[w];
                const int64_t sop = (int64_t)rw.op_begin + rw.n_recv + src_k[r.op_begin + j];
                if (row_of_op[sop] < 0) {
                  row_of_op[sop] = rows++;
                  Op& ps = slot_of(w)[1 + rw.n_recv + src_k[r.op_begin + j]];
                  ps.peer_const = row_of_op[sop]; ps.info |= OP_MAILBOX;
                }
                Op& pr = slot_of(i)[1 + j];
                std::memcpy(&pr.omega, &row_of_op[sop], sizeof(double)); pr.info |= OP_MAILBOX;
                ++cp.mailbox_receives;
              }
            }
          if (rows > 0) {
            cp.mailbox_rows = rows; cp.mailbox_width = kc_width(c);
            for (auto& cl : cp.launches) cl.flags |= CHAIN_LAUNCH_MAILBOX;
          }
        }
        out.chains.push_back(std::move(cp));
      }
      if (!ok) { out.chains.clear(); out.plain_launches.clear(); }
    }
  }
}


// reference factors_messages.hxx:2699-2744.  A dispatcher = the run of list entries with one message type and role; a
// batch-capable one (MessageDispatcher::CanCallSendMessages: SendMessagesToRight for left-role entries, ...ToLeft for
// right-role ones) with MORE THAN ONE active message (omega > 0) makes one call with the sum of the run's weights;
// the device batch op (lpmp_msg_flags) gives each active message sum / n_active times the plain message.
void Plan::effective_send_weights(int32_t f, const double* omega, double* w) const {
  int64_t k = 0;
  for (int64_t j = fm_off[f]; j < fm_off[f + 1];) {
    int64_t j2 = j;
    while (j2 < fm_off[f + 1] && m_type[fm[j2].msg] == m_type[fm[j].msg] && fm[j2].role == fm[j].role) ++j2;
    if (!fm[j].sends) { j = j2; continue; }
    const int64_t n = j2 - j;
    const int fl = mtypes[m_type[fm[j].msg]].flags;
    const bool batch = fm[j].role == 0 ? (fl & LPMP_MF_BATCH_TO_RIGHT) != 0 : (fl & LPMP_MF_BATCH_TO_LEFT) != 0;
    if (batch) {
      int64_t n_active = 0; double sum = 0.0;
      for (int64_t i = 0; i < n; ++i) { if (omega[k + i] > 0.0) ++n_active; sum += omega[k + i]; }
      const double each = n_active > 1 ? sum / double(n_active) : 0.0;
      for (int64_t i = 0; i < n; ++i) w[k + i] = omega[k + i] > 0.0 ? (n_active > 1 ? each : omega[k + i]) : 0.0;
    } else {
      for (int64_t i = 0; i < n; ++i) w[k + i] = omega[k + i];
    }
    k += n; j = j2;
  }
}

// send_messages_with_adaptive_weights (reference factors_messages.hxx:2860-2926) walks ALL dispatchers of the factor
// with the iterator over the SENDING weights: defined only when every message of an updated factor sends; its batch
// branch (and send_messages_residual's) calls op members no device op is defined for
std::string Plan::adaptive_obstacle() const {
  if (any_batch) return "adaptive sends with batch-capable message ops are not built";
  for (int64_t f = 0; f < nf; ++f) {
    if (!updated[f]) continue;
    bool any = false, all = true;
    for (int64_t j = fm_off[f]; j < fm_off[f + 1]; ++j) { any = any || fm[j].sends; all = all && fm[j].sends; }
    if (any && !all) return "adaptive sends: factor " + std::to_string(f) + " has messages it does not send through (undefined in the reference)";
  }
  return "";
}

// ---- partition sweeps (reference LP_MP.h:1717-1843).  union_find.hxx:5-93: union by size, the first argument's root
// wins ties, contiguous ids in increasing root index; partitions without updated factors are dropped (:1736-1745).
// Intra-partition order: the reference sorts by position in forwardOrdering_ with a comparator that is false for every
// pair (`std::get<0>(a) < std::get<0>(a)`, :1775), so the result depends on the standard library's sort; the engine
// keeps the order the partition was populated in (insertion order of the updated factors, :1755-1760) — what a stable
// sort returns for that comparator, and what libstdc++'s std::sort leaves for partitions of up to 16 factors.
void Plan::ensure_partition() {
  if (part.valid) return;
  part = Partition();
  std::vector<int64_t> id(nf), sz(nf, 1);
  std::iota(id.begin(), id.end(), 0);
  auto find = [&](int64_t p) {
    int64_t root = p;
    while (root != id[root]) root = id[root];
    while (p != root) { const int64_t nx = id[p]; id[p] = root; p = nx; }
    return root;
  };
  for (size_t k = 0; k + 1 < part_pairs.size(); k += 2) {
    const int64_t i = find(part_pairs[k]), j = find(part_pairs[k + 1]);
    if (i == j) continue;
    if (sz[i] < sz[j]) { id[i] = j; sz[j] += sz[i]; } else { id[j] = i; sz[i] += sz[j]; }
  }
  std::vector<int64_t> root_id(nf, -1), count;
  for (int64_t i = 0; i < nf; ++i) root_id[find(i)] = 1;
  int64_t next = 0;
  for (int64_t d = 0; d < nf; ++d) if (root_id[d] == 1) root_id[d] = next++;
  count.assign(next, 0);
  for (int64_t i = 0; i < nf; ++i) if (updated[i]) count[root_id[find(i)]]++;
  std::vector<int64_t> part_of(next, -1);
  int64_t P = 0;
  for (int64_t c = 0; c < next; ++c) if (count[c] > 0) part_of[c] = P++;
  if (P == 0) fail("partition sweeps: the model has no updated factor");
  part.off.assign(P + 1, 0);
  for (int64_t c = 0; c < next; ++c) if (part_of[c] >= 0) part.off[part_of[c] + 1] = count[c];
  std::partial_sum(part.off.begin(), part.off.end(), part.off.begin());
  part.f.resize(part.off[P]);
  {
    std::vector<int64_t> cur(part.off.begin(), part.off.end() - 1);
    for (int64_t i = 0; i < nf; ++i) if (updated[i]) part.f[cur[part_of[root_id[find(i)]]]++] = (int32_t)i;
  }
  auto seg = [&](int64_t a, bool rev_a, int64_t b, bool rev_b) {
    SegList s;
    auto push = [&](int64_t p, bool rev) {
      if (p < 0) return;
      const int32_t* x = part.f.data() + part.off[p]; const int64_t n = part.off[p + 1] - part.off[p];
      for (int64_t i = 0; i < n; ++i) s.f.push_back(rev ? x[n - 1 - i] : x[i]);
    };
    push(a, rev_a); push(b, rev_b);
    anisotropic_weights(s.f.data(), (int64_t)s.f.size(), s.om, s.mk);
    return s;
  };
  for (int64_t i = 0; i < P; ++i) { part.fwd.push_back(seg(i, false, -1, false)); part.bwd.push_back(seg(i, true, -1, false)); }
  for (int64_t i = 0; i + 1 < P; ++i) {
    part.push_fwd.push_back(seg(i, false, i + 1, true));      // :1806-1811
    part.ov_fwd.push_back(seg(i, false, i + 1, true));        // :1835-1836
    part.ov_bwd.push_back(seg(i + 1, false, i, true));        // :1838-1839
  }
  for (int64_t ri = 0; ri + 1 < P; ++ri) { const int64_t i = P - ri - 1; part.push_bwd.push_back(seg(i, false, i - 1, true)); }   // :1813-1820
  part.valid = true;
}

void Plan::partition_pass_segments(int rtype, int inner, std::vector<Segment>& out) {
  ensure_partition();
  const int64_t P = (int64_t)part.fwd.size();
  auto add = [&](const SegList& s) { out.push_back({s.f.data(), (int64_t)s.f.size(), s.om.off.data(), s.om.data.data(), s.mk.off.data(), s.mk.data.data()}); };
  if (rtype == 2) {            // compute_partition_pass, LP_MP.h:1932-1963
    for (int64_t i = 0; i < P; ++i) {
      for (int it = 0; it < inner; ++it) { add(part.fwd[i]); add(part.bwd[i]); }
      if (i < P - 1) add(part.push_fwd[i]);
    }
    for (int64_t ri = 0; ri < P; ++ri) {
      const int64_t i = P - ri - 1;
      for (int it = 0; it < inner; ++it) { add(part.fwd[i]); add(part.bwd[i]); }
      if (i != 0) add(part.push_bwd[ri]);
    }
  } else {                     // compute_overlapping_partition_pass, LP_MP.h:2024-2050
    for (int64_t i = 0; i + 1 < P; ++i) {
      for (int it = 0; it < inner; ++it) { add(part.ov_fwd[i]); add(part.ov_bwd[i]); }
      add(part.ov_fwd[i]);
    }
    for (int64_t ri = 1; ri < P; ++ri) {
      const int64_t i = P - ri - 1;
      for (int it = 0; it < inner; ++it) { add(part.ov_bwd[i]); add(part.ov_fwd[i]); }
      add(part.ov_bwd[i]);
    }
  }
}

void Plan::make_schedule(const int32_t* factors, int64_t n, const int64_t* om_off, const double* om,
                         const int64_t* mk_off, const uint8_t* mk, Schedule& out) const {
  make_schedule(std::vector<Segment>{Segment{factors, n, om_off, om, mk_off, mk}}, false, out);
}

}  // namespace lpmp
