// graph.cpp — host-side graph work of the planner (no GPU): the variable order that exposes parallelism and the refinement of a
// partition, on the planner's threads.
//
// The sweep is Gauss-Seidel over the factor ORDER, which is input data (AddFactorRelation, reference include/LP_MP.h:698-702;
// topological sort include/topological_sort.hxx:100-144): the engine runs the order it is given, one launch per dependent level.
// A grid inserted row by row has H + W - 1 levels per direction, the same grid in a 2-colour order 2.  lpmp_graph_colour_major_order
// computes such an order for any pairwise conflict graph, lpmp_plan_suggest_order applies it to a planned model and hands the
// result back as a position per factor, so that a C++ caller can turn it into AddFactorRelation calls.
// lp_mp_amd/ordering.py holds the same algorithms in numpy (the readable statement; tests/test_graph_host.py compares the two
// bit for bit), lp_mp_amd/multi_gpu.py the numpy form of the partition refinement.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lpmp_engine.h"
#include "plan.hpp"

extern "C" int lpmp_set_last_error(const char* msg);   // engine.cpp

namespace lpmp {

namespace {

constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ULL;

inline uint64_t splitmix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// undirected adjacency with multiplicities kept (two pairwise factors between the same variables are two entries)
struct Adj {
  std::vector<int64_t> off;
  std::vector<int32_t> adj;
  bool self_loop = false;   // an edge v - v was given (dropped from the lists unless asked for)
};

// keep_self_loops: v - v as two entries of v's list (what A + A^T makes of a diagonal entry)
Adj build_adj(int64_t n, int64_t m, const int64_t* ei, const int64_t* ej, bool keep_self_loops = false) {
  Adj g;
  g.off.assign((size_t)n + 1, 0);
  for (int64_t e = 0; e < m; ++e) {
    const int64_t a = ei[e], b = ej[e];
    if (a < 0 || a >= n || b < 0 || b >= n) throw std::runtime_error("graph: an edge names a variable outside [0, n)");
    if (a == b) { g.self_loop = true; if (!keep_self_loops) continue; }
    ++g.off[(size_t)a + 1]; ++g.off[(size_t)b + 1];
  }
  for (int64_t v = 0; v < n; ++v) g.off[(size_t)v + 1] += g.off[(size_t)v];
  g.adj.resize((size_t)g.off[(size_t)n]);
  std::vector<int64_t> at(g.off.begin(), g.off.end() - 1);
  for (int64_t e = 0; e < m; ++e) {
    const int64_t a = ei[e], b = ej[e];
    if (a == b && !keep_self_loops) continue;
    g.adj[(size_t)at[(size_t)a]++] = (int32_t)b; g.adj[(size_t)at[(size_t)b]++] = (int32_t)a;
  }
  return g;
}

// Component by component: colours in {0, 1} = BFS depth parity from the lowest-numbered vertex of a component that has no odd cycle
// (ordering.two_colouring: the component labelling of the bipartite double cover gives exactly this colouring); the vertices of
// every other component stay uncoloured (-1) for the greedy colouring.  A model of several parts (a 2-colourable grid beside
// higher-order factors: C5) keeps the 2 levels of its grid — and with them the joined passes — whatever the rest needs.
// Returns the number of vertices left uncoloured.
int64_t two_colouring_by_component(int64_t n, const Adj& g, const int64_t* ei, const int64_t* ej, int64_t m, std::vector<int8_t>& colour) {
  colour.assign((size_t)n, -1);
  std::vector<uint8_t> loop((size_t)n, 0);                 // (in the double cover a loop joins the two copies of its vertex)
  if (g.self_loop) for (int64_t e = 0; e < m; ++e) if (ei[e] == ej[e]) loop[(size_t)ei[e]] = 1;
  std::vector<int32_t> queue;
  queue.reserve(1024);
  int64_t left = 0;
  for (int64_t s = 0; s < n; ++s) {
    if (colour[(size_t)s] != -1) continue;
    colour[(size_t)s] = 0;
    queue.clear(); queue.push_back((int32_t)s);
    bool odd = false;
    for (size_t h = 0; h < queue.size(); ++h) {
      const int32_t v = queue[h];
      const int8_t c = colour[(size_t)v];
      odd = odd || loop[(size_t)v];
      for (int64_t k = g.off[(size_t)v]; k < g.off[(size_t)v + 1]; ++k) {
        const int32_t u = g.adj[(size_t)k];
        if (colour[(size_t)u] == -1) { colour[(size_t)u] = (int8_t)(1 - c); queue.push_back(u); }
        else if (colour[(size_t)u] == c) odd = true;
      }
    }
    if (odd) { for (int32_t v : queue) colour[(size_t)v] = -2; left += (int64_t)queue.size(); }   // (-2: seen, not 2-colourable)
  }
  if (left) for (int64_t v = 0; v < n; ++v) if (colour[(size_t)v] == -2) colour[(size_t)v] = -1;
  return left;
}

// Jones-Plassmann greedy colouring (ordering.greedy_colouring): in every round the uncoloured vertices whose priority beats all
// their uncoloured neighbours' take the smallest colour none of their coloured neighbours has.  Priorities: the counter hash of
// the vertex (ties by index), so that the numpy statement and this one agree without sharing a random generator.  Winners of a
// round are never adjacent, and a round reads only the colours of earlier rounds: the result does not depend on the threads.
// Vertices that already hold a colour (the 2-colourable components) keep it; they are adjacent to none of the others.
void greedy_colouring(int64_t n, const Adj& g, uint64_t seed, std::vector<int8_t>& colour) {
  std::vector<uint64_t> h((size_t)n);
  parallel_blocks(n, 1 << 16, [&](int64_t b, int64_t e) { for (int64_t v = b; v < e; ++v) h[(size_t)v] = splitmix(seed + (uint64_t)(v + 1) * GOLD); });
  auto beats = [&](int32_t a, int32_t b) { return h[(size_t)a] > h[(size_t)b] || (h[(size_t)a] == h[(size_t)b] && a > b); };
  std::vector<int32_t> live, next;
  for (int64_t v = 0; v < n; ++v) if (colour[(size_t)v] < 0) live.push_back((int32_t)v);
  std::vector<int8_t> chosen((size_t)n, -1);
  while (!live.empty()) {
    const int64_t nl = (int64_t)live.size();
    parallel_blocks(nl, 1 << 14, [&](int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) {
        const int32_t v = live[(size_t)i];
        uint64_t used = 0;
        bool wins = true;
        for (int64_t k = g.off[(size_t)v]; k < g.off[(size_t)v + 1]; ++k) {
          const int32_t u = g.adj[(size_t)k];
          const int8_t cu = colour[(size_t)u];
          if (cu >= 0) used |= (uint64_t)1 << cu;
          else if (beats(u, v)) { wins = false; break; }
        }
        if (!wins) { chosen[(size_t)v] = -1; continue; }
        const uint64_t free_ = ~used;
        const int c = __builtin_ctzll(free_);
        if (c >= 63) throw std::runtime_error("graph needs more than 63 colours");
        chosen[(size_t)v] = (int8_t)c;
      }
    });
    next.clear();
    for (int64_t i = 0; i < nl; ++i) {
      const int32_t v = live[(size_t)i];
      if (chosen[(size_t)v] >= 0) colour[(size_t)v] = chosen[(size_t)v]; else next.push_back(v);
    }
    if (next.size() == live.size()) throw std::logic_error("greedy colouring made no progress");
    live.swap(next);
  }
}

// rank[v] = position of v when the vertices are sorted by colour (stable in the index)
int32_t colour_major_rank(int64_t n, const std::vector<int8_t>& colour, int64_t* rank) {
  int64_t count[64] = {0};
  int32_t n_col = 0;
  for (int64_t v = 0; v < n; ++v) { ++count[colour[(size_t)v]]; n_col = std::max<int32_t>(n_col, colour[(size_t)v] + 1); }
  int64_t first[64], at = 0;
  for (int c = 0; c < 64; ++c) { first[c] = at; at += count[c]; }
  for (int64_t v = 0; v < n; ++v) rank[v] = first[colour[(size_t)v]]++;
  return n_col;
}

int32_t colour_major_order(int64_t n, int64_t m, const int64_t* ei, const int64_t* ej, uint64_t seed, int64_t* rank) {
  if (n >= ((int64_t)1 << 31)) throw std::runtime_error("graph: more than 2^31 variables");
  const Adj g = build_adj(n, m, ei, ej);
  std::vector<int8_t> colour;
  if (two_colouring_by_component(n, g, ei, ej, m, colour) > 0) greedy_colouring(n, g, seed, colour);
  return colour_major_rank(n, colour, rank);
}

// multi_gpu.refine_partition on the planner's threads, move for move: per round the gain of moving v to the part most of its
// neighbours live in; a pseudo-random half of the variables with positive gain, best gains first (stable), target by target in
// part order as long as the target has room.
void refine_partition(int64_t n, const Adj& g, int32_t world, int32_t rounds, double imbalance, uint64_t seed, int64_t* part) {
  const int64_t cap = (int64_t)std::ceil((double)n / world * (1.0 + imbalance));
  std::vector<int32_t> best((size_t)n);
  std::vector<double> gain((size_t)n);
  std::vector<int64_t> size((size_t)world);
  std::vector<int32_t> cand, order;
  for (int32_t r = 0; r < rounds; ++r) {
    parallel_blocks(n, 1 << 14, [&](int64_t b, int64_t e) {
      std::vector<double> cnt((size_t)world);
      for (int64_t v = b; v < e; ++v) {
        std::fill(cnt.begin(), cnt.end(), 0.0);
        for (int64_t k = g.off[(size_t)v]; k < g.off[(size_t)v + 1]; ++k) cnt[(size_t)part[g.adj[(size_t)k]]] += 1.0;
        int32_t bp = 0;
        for (int32_t p = 1; p < world; ++p) if (cnt[(size_t)p] > cnt[(size_t)bp]) bp = p;      // argmax: the first maximum
        best[(size_t)v] = bp; gain[(size_t)v] = cnt[(size_t)bp] - cnt[(size_t)part[v]];
      }
    });
    cand.clear();
    const int shift = r % 20;
    const uint64_t want = (uint64_t)(r & 1);
    for (int64_t v = 0; v < n; ++v) {
      const uint64_t coin = ((uint64_t)v * GOLD + seed) >> 40;
      if (gain[(size_t)v] > 0 && ((coin >> shift) & 1) == want) cand.push_back((int32_t)v);
    }
    if (cand.empty()) { if (r > 2) break; continue; }
    std::fill(size.begin(), size.end(), 0);
    for (int64_t v = 0; v < n; ++v) ++size[(size_t)part[v]];
    order = cand;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return gain[(size_t)a] > gain[(size_t)b]; });
    int64_t moved = 0;
    for (int32_t t = 0; t < world; ++t) {
      int64_t room = cap - size[(size_t)t];
      if (room <= 0) continue;
      int64_t took = 0;
      for (int32_t v : order) {
        if (best[(size_t)v] != t) continue;
        if (took == room) break;
        --size[(size_t)part[v]]; part[v] = t; ++took;
      }
      size[(size_t)t] += took; moved += took;
    }
    if (moved == 0 && r > 2) break;
  }
}

}  // namespace

// An order of ALL factors of a planned model in which the updated factors come colour by colour (two updated factors conflict when
// a message joins them or they touch a common factor — the relation the dependent levels are computed from, plan.cpp), and every
// other factor keeps its place RELATIVE to the updated factors around it: a factor that came after k of its updated neighbours in
// the current forward order comes after k of them again (an MRF's pairwise factor between its two unaries, u_i -> p_ij -> u_j; a
// multicut triplet after all its edges).  rank[f] = position of factor f; a caller turns it into relations as a chain through all
// factors: AddFactorRelation(by_rank[i], by_rank[i + 1]) — its only topological order is this one.
int32_t suggest_order(const Plan& p, uint64_t seed, int32_t* rank) {
  const int64_t nf = p.nf;
  if (nf == 0) return 0;
  std::vector<int32_t> pos_cur((size_t)nf);
  for (int64_t i = 0; i < nf; ++i) pos_cur[(size_t)p.order[0][(size_t)i]] = (int32_t)i;
  // updated factors, numbered in factor order
  std::vector<int32_t> uid((size_t)nf, -1), ufac;
  for (int64_t f = 0; f < nf; ++f) if (p.updated[(size_t)f]) { uid[(size_t)f] = (int32_t)ufac.size(); ufac.push_back((int32_t)f); }
  const int64_t nu = (int64_t)ufac.size();
  // neighbours of every factor through messages
  std::vector<int64_t> noff((size_t)nf + 1, 0);
  for (int64_t k = 0; k < p.nm; ++k) { ++noff[(size_t)p.m_left[(size_t)k] + 1]; ++noff[(size_t)p.m_right[(size_t)k] + 1]; }
  for (int64_t f = 0; f < nf; ++f) noff[(size_t)f + 1] += noff[(size_t)f];
  std::vector<int32_t> nb((size_t)noff[(size_t)nf]);
  { std::vector<int64_t> at(noff.begin(), noff.end() - 1);
    for (int64_t k = 0; k < p.nm; ++k) { const int32_t l = p.m_left[(size_t)k], r = p.m_right[(size_t)k]; nb[(size_t)at[(size_t)l]++] = r; nb[(size_t)at[(size_t)r]++] = l; } }
  // conflict edges between updated factors: two updates conflict when what they touch — their own factor and the peers of their
  // messages — overlaps (plan.cpp, level recurrence), i.e. all updated factors around ANY factor f (f itself included when it
  // is updated: `right` / `full` schedules update the higher factors too) are pairwise in conflict
  std::vector<int64_t> ei, ej;
  std::vector<int32_t> un;
  for (int64_t f = 0; f < nf; ++f) {
    un.clear();
    if (p.updated[(size_t)f]) un.push_back(uid[(size_t)f]);
    for (int64_t k = noff[(size_t)f]; k < noff[(size_t)f + 1]; ++k) if (p.updated[(size_t)nb[(size_t)k]]) un.push_back(uid[(size_t)nb[(size_t)k]]);
    std::sort(un.begin(), un.end()); un.erase(std::unique(un.begin(), un.end()), un.end());
    if (un.size() > 63) throw std::runtime_error("suggest_order: a factor with more than 62 updated neighbours (they would need as many colours)");
    for (size_t a = 0; a < un.size(); ++a) for (size_t b = a + 1; b < un.size(); ++b) { ei.push_back(un[a]); ej.push_back(un[b]); }
  }
  std::vector<int64_t> urank((size_t)std::max<int64_t>(nu, 1));
  const int32_t n_col = nu > 0 ? colour_major_order(nu, (int64_t)ei.size(), ei.data(), ej.data(), seed, urank.data()) : 0;
  // keys: an updated factor (its new position, 0); another factor behind the k-th of its updated neighbours, k = how many of them
  // precede it now — before the first of them (minor -1) when none does; factors without updated neighbours last, in factor order
  struct Key { int64_t major; int32_t minor, f; };
  std::vector<Key> keys((size_t)nf);
  std::vector<int64_t> q;
  for (int64_t f = 0; f < nf; ++f) {
    if (p.updated[(size_t)f]) { keys[(size_t)f] = {urank[(size_t)uid[(size_t)f]], 0, (int32_t)f}; continue; }
    q.clear();
    int64_t before = 0;
    for (int64_t k = noff[(size_t)f]; k < noff[(size_t)f + 1]; ++k) {
      const int32_t g = nb[(size_t)k];
      if (!p.updated[(size_t)g]) continue;
      q.push_back(urank[(size_t)uid[(size_t)g]]);
      if (pos_cur[(size_t)g] < pos_cur[(size_t)f]) ++before;
    }
    if (q.empty()) { keys[(size_t)f] = {nu, 1, (int32_t)f}; continue; }
    std::sort(q.begin(), q.end());
    // (duplicate messages between the same two factors count once per message on both sides of the comparison)
    keys[(size_t)f] = before > 0 ? Key{q[(size_t)before - 1], 1, (int32_t)f} : Key{q[0], -1, (int32_t)f};
  }
  std::vector<int32_t> idx((size_t)nf);
  for (int64_t f = 0; f < nf; ++f) idx[(size_t)f] = (int32_t)f;
  std::sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) {
    const Key &x = keys[(size_t)a], &y = keys[(size_t)b];
    return x.major != y.major ? x.major < y.major : x.minor != y.minor ? x.minor < y.minor : x.f < y.f;
  });
  for (int64_t i = 0; i < nf; ++i) rank[idx[(size_t)i]] = (int32_t)i;
  return n_col;
}

}  // namespace lpmp

using namespace lpmp;

namespace {
template <class F>
int guarded_host(F&& f) {
  try { f(); return LPMP_OK; }
  catch (const std::bad_alloc&) { lpmp_set_last_error("out of host memory"); return LPMP_ERR_INVALID; }
  catch (const std::exception& e) { lpmp_set_last_error(e.what()); return LPMP_ERR_INVALID; }
}
}  // namespace

extern "C" {

int lpmp_graph_colour_major_order(int64_t n, int64_t m, const int64_t* edge_i, const int64_t* edge_j, uint64_t seed, int64_t* rank_out,
                                  int32_t* n_colours_out) {
  return guarded_host([&] {
    if (n < 0 || m < 0 || (m > 0 && (!edge_i || !edge_j)) || (n > 0 && !rank_out)) throw std::runtime_error("bad argument");
    const int32_t k = n > 0 ? colour_major_order(n, m, edge_i, edge_j, seed, rank_out) : 0;
    if (n_colours_out) *n_colours_out = k;
  });
}

int lpmp_graph_refine_partition(int64_t n, int64_t m, const int64_t* edge_i, const int64_t* edge_j, int32_t world, int32_t rounds,
                                double imbalance, uint64_t seed, int64_t* part_inout) {
  return guarded_host([&] {
    if (n < 0 || m < 0 || world < 1 || rounds < 0 || (m > 0 && (!edge_i || !edge_j)) || (n > 0 && !part_inout)) throw std::runtime_error("bad argument");
    if (n >= ((int64_t)1 << 31)) throw std::runtime_error("graph: more than 2^31 variables");
    for (int64_t v = 0; v < n; ++v) if (part_inout[v] < 0 || part_inout[v] >= world) throw std::runtime_error("refine_partition: part out of range");
    if (world == 1 || rounds == 0 || n == 0) return;
    const Adj g = build_adj(n, m, edge_i, edge_j, true);
    refine_partition(n, g, world, rounds, imbalance, seed, part_inout);
  });
}

}  // extern "C"
