// tools/offload_solver_loop.cpp — what the reference's UNMODIFIED Solve loop costs per iteration through the off-load adapter.
//
// A reference-shaped LP<FMC> (tests/cpp/mock_reference_lp.hxx: the reference's member names and container surface) holding a
// grid MRF is wrapped in lpmp_offload::offloaded<> and driven exactly the way Solver::Solve drives an LP (reference
// include/solver.hxx:238-243, :268-284): per iteration  set_reparametrization(mode); ComputePass(iter); lb = LowerBound();
// The adapter lets the engine run passes ahead of the loop (lpmp_set_speculation); the loop is run with that on and off and
// the two bound histories are compared (they must be identical).
//
// --rounding 1: the DEFAULT cycle of MpRoundingSolver under StandardVisitor (include/solver.hxx:387-397,
// include/visitors/standard_visitor.hxx:37,172-185): every primalComputationInterval-th (5th) iteration runs
//   set_reparametrization(damped_uniform); ComputeForwardPassAndPrimal(iter); RegisterPrimal(); ComputeBackwardPassAndPrimal(iter);
//   RegisterPrimal();  [RegisterPrimal = EvaluatePrimal, + CheckPrimalConsistency when the cost improved: solver.hxx:320-337]
// the others the anisotropic ComputePass; LowerBound() after every iteration.  Reported: mean ms per iteration of the cycle and
// of each kind of iteration; the bound / primal-cost history with and without passes running ahead must be identical.
//
// --order row_major: the grid inserted row by row (u_i -> p_ij -> u_j in row-major variable order: H + W - 1 dependent levels per
// directional sweep); --order suggested: the same insertion, then the order the engine suggests for it — lpmp_plan_suggest_order on
// the planned row-major model — applied to a second LP as a chain of AddFactorRelation calls: a reference user
// who inserts a grid row by row gets the colour-major time without knowing about colours.
//
//   offload_solver_loop [--grid 512] [--labels 32] [--iterations 60] [--warm 24] [--rounding 0|1] [--order colour_major|row_major|suggested]
// Build: g++ -std=c++17 -O2 -I lp_mp_amd/include -I tests/cpp tools/offload_solver_loop.cpp -L lp_mp_amd/csrc -llpmp_engine -Wl,-rpath,$PWD/lp_mp_amd/csrc
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "mock_reference_lp.hxx"
#include "lpmp_offload.hxx"

namespace user {
using LP_MP::REAL; using LP_MP::INDEX;
struct Unary {
  explicit Unary(const std::vector<REAL>& c_) : c(c_) {}
  REAL LowerBound() const { return *std::min_element(c.begin(), c.end()); }
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(c); }
  std::vector<REAL> c;
};
struct Pairwise {
  Pairwise(INDEX a, INDEX b) : d1(a), d2(b), pw(a * b, 0.0), m1(a, 0.0), m2(b, 0.0) {}
  REAL LowerBound() const { return 0; }
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(m1, m2); }
  INDEX d1, d2; std::vector<REAL> pw, m1, m2;
};
template <LP_MP::Chirality C> struct UPMsg {};
}  // namespace user
namespace lpmp_offload {
template <> struct device_kind<user::Unary> : vector_kind<> {};
template <> struct device_kind<user::Pairwise> : pairwise_dense_kind<user::Pairwise> {
  static std::size_t dim1(const user::Pairwise& f) { return f.d1; }
  static std::size_t dim2(const user::Pairwise& f) { return f.d2; }
  static double table(const user::Pairwise& f, std::size_t a, std::size_t b) { return f.pw[a * f.d2 + b]; }
};
template <LP_MP::Chirality C> struct device_message<user::UPMsg<C>> : unary_pairwise_message<C == LP_MP::Chirality::left ? 0 : 1> {};
}  // namespace lpmp_offload
struct FMC_MRF {
  using U = LP_MP::FactorContainer<user::Unary, FMC_MRF, 0, true>;
  using P = LP_MP::FactorContainer<user::Pairwise, FMC_MRF, 1>;
  using ML = LP_MP::MessageContainer<user::UPMsg<LP_MP::Chirality::left>, 0, 1, LP_MP::message_passing_schedule::left, LP_MP::variableMessageNumber, 1, FMC_MRF, 0>;
  using MR = LP_MP::MessageContainer<user::UPMsg<LP_MP::Chirality::right>, 0, 1, LP_MP::message_passing_schedule::left, LP_MP::variableMessageNumber, 1, FMC_MRF, 1>;
  using FactorList = LP_MP::meta::list<U, P>;
  using MessageList = LP_MP::meta::list<ML, MR>;
};

static double u01(uint64_t& st) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(st >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
  int G = 512, L = 32, iters = 60, warm = 24, rounding = 0;
  std::string order = "colour_major";
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string a = argv[i];
    if (a == "--grid") G = std::atoi(argv[i + 1]); else if (a == "--labels") L = std::atoi(argv[i + 1]);
    else if (a == "--iterations") iters = std::atoi(argv[i + 1]); else if (a == "--warm") warm = std::atoi(argv[i + 1]);
    else if (a == "--rounding") rounding = std::atoi(argv[i + 1]); else if (a == "--order") order = argv[i + 1];
  }
  if (order != "colour_major" && order != "row_major" && order != "suggested") { std::fprintf(stderr, "offload_solver_loop: --order colour_major|row_major|suggested\n"); return 2; }
  try {
    using LP_device = lpmp_offload::offloaded<LP_MP::LP<FMC_MRF>>;
    LP_MP::mock_cmd_line cmd;
    // the grid as the caller inserts it: variable order `pos` (colour-major: black cells first — the order whose consecutive passes
    // join; else row by row), u_i -> p_ij -> u_j per edge — or, with `rank`, the same factors and messages with the
    // relations replaced by a chain through all factors in the suggested order
    auto build = [&](LP_device& lp, bool colour_major, const std::vector<int32_t>* rank) {
      std::vector<int> pos((size_t)G * G);
      if (colour_major) { int b = 0, w = (G * G + 1) / 2; for (int r = 0; r < G; ++r) for (int c = 0; c < G; ++c) pos[(size_t)r * G + c] = ((r + c) & 1) == 0 ? b++ : w++; }
      else for (int i = 0; i < G * G; ++i) pos[(size_t)i] = i;
      std::vector<int> cell_of((size_t)G * G);
      for (int i = 0; i < G * G; ++i) cell_of[pos[i]] = i;
      std::vector<FMC_MRF::U*> u((size_t)G * G);
      std::vector<LP_MP::FactorTypeAdapter*> all;            // every factor in insertion order = the engine's factor index
      uint64_t st = 42;
      std::vector<double> c((size_t)L);
      for (int k = 0; k < G * G; ++k) { for (auto& x : c) x = u01(st); u[(size_t)cell_of[k]] = lp.add_factor<FMC_MRF::U>(c); all.push_back(u[(size_t)cell_of[k]]); }
      auto edge = [&](int a, int b) {
        if (pos[a] > pos[b]) std::swap(a, b);
        auto* p = lp.add_factor<FMC_MRF::P>((LP_MP::INDEX)L, (LP_MP::INDEX)L);
        all.push_back(p);
        for (auto& x : p->GetFactor()->pw) x = u01(st);
        lp.add_message<FMC_MRF::ML>(u[a], p); lp.add_message<FMC_MRF::MR>(u[b], p);
        if (!rank) { lp.AddFactorRelation(u[a], p); lp.AddFactorRelation(p, u[b]); }
      };
      for (int r = 0; r < G; ++r) for (int cc = 0; cc < G; ++cc) {
        if (cc + 1 < G) edge(r * G + cc, r * G + cc + 1);
        if (r + 1 < G) edge(r * G + cc, (r + 1) * G + cc);
      }
      if (rank) {                                            // the suggestion as AddFactorRelation calls
        std::vector<int32_t> by_rank(rank->size());
        for (size_t f = 0; f < rank->size(); ++f) by_rank[(size_t)(*rank)[f]] = (int32_t)f;
        for (size_t i = 0; i + 1 < by_rank.size(); ++i) lp.AddFactorRelation(all[(size_t)by_rank[i]], all[(size_t)by_rank[i + 1]]);
      }
    };
    std::vector<int32_t> rank;
    int32_t n_colours = 0;
    double suggest_ms = 0;
    if (order == "suggested") {
      LP_device first(cmd);
      build(first, false, nullptr);
      first.Begin();
      (void)first.engine();                                  // (flatten + upload outside the timed call)
      const auto t0 = std::chrono::steady_clock::now();
      const std::vector<int32_t> by_rank = first.suggested_order(0, &n_colours);   // offloaded<>: lpmp_plan_suggest_order on the planned LP
      suggest_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      rank.resize(by_rank.size());
      for (size_t i = 0; i < by_rank.size(); ++i) rank[(size_t)by_rank[i]] = (int32_t)i;
      first.End();
    }
    LP_device lp(cmd);
    build(lp, order == "colour_major", order == "suggested" ? &rank : nullptr);
    lp.Begin();
    std::size_t iter = 0;                                    // Solver::iter: grows over the whole solve (the primal time stamps follow it)
    std::size_t iter_base = 0;                               // the second configuration continues the stamps of the first (they must not decrease)
    double best_primal = std::numeric_limits<double>::infinity();
    double ms_plain = 0, ms_round = 0; int n_plain = 0, n_round = 0;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto register_primal = [&](std::vector<double>* hist) {   // Solver::RegisterPrimal, solver.hxx:320-337
      const double cost = lp.EvaluatePrimal();
      if (hist) hist->push_back(cost);
      if (cost < best_primal && lp.CheckPrimalConsistency()) best_primal = cost;
    };
    auto run = [&](int n, std::vector<double>* hist) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int it = 0; it < n; ++it, ++iter) {               // Solver::Solve: PreIterate, Iterate, PostIterate (computeLowerBound)
        // StandardVisitor::visit: curIter_ >= primalComputationStart_ (1) && (curIter_ - 1) % primalComputationInterval_ (5) == 0
        const bool primal = rounding && iter >= 1 && (iter - 1) % 5 == 0;
        const double t_it = now_ms();
        if (primal) {
          lp.set_reparametrization(LP_MP::LPReparametrizationMode::DampedUniform);
          lp.ComputeForwardPassAndPrimal(iter_base + iter);
          register_primal(hist);
          lp.ComputeBackwardPassAndPrimal(iter_base + iter);
          register_primal(hist);
        } else {
          lp.set_reparametrization(LP_MP::LPReparametrizationMode::Anisotropic);
          lp.ComputePass(iter);
        }
        const double lb = lp.LowerBound();
        if (hist) hist->push_back(lb);
        (primal ? ms_round : ms_plain) += now_ms() - t_it; ++(primal ? n_round : n_plain);
      }
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
    };
    // the same duals for both runs: remember the start, restore it in between
    lp.set_reparametrization(LP_MP::LPReparametrizationMode::Anisotropic);
    const double lb0 = lp.LowerBound();
    lpmp_engine* e = lp.engine();
    std::vector<double> start((size_t)lpmp_dual_size(e));
    lpmp_offload::check(lpmp_download_duals(e, start.data()));
    double ms[2], kind_ms[2][2]; std::vector<double> hist[2]; int64_t stats[2][4];
    for (int k = 0; k < 2; ++k) {
      lp.set_speculation(k == 0 ? 0 : 16);
      lpmp_offload::check(lpmp_upload_duals(e, start.data()));
      (void)lp.LowerBound();
      iter_base += iter; iter = 0; best_primal = std::numeric_limits<double>::infinity();
      run(warm, &hist[k]);                                   // builds the ticket lists of the batch sizes, warms the clocks
      // (a batch of passes launched ahead during the warm-up must not be credited to the timed iterations: settle first — the
      // timed run then starts from the caller's own state, as a Solve loop does.  With the rounding cycle a rounding iteration
      // settles by itself: choose --warm so that the timed run begins with one, (warm - 1) % 5 == 0, and nothing here disturbs
      // what the engine has learnt about the caller's rhythm)
      if (!rounding) lpmp_offload::check(lpmp_synchronize(e));
      else if ((warm - 1) % 5 != 0) std::fprintf(stderr, "offload_solver_loop: --rounding 1 wants (warm - 1) %% 5 == 0 (the timed run then starts at a rounding iteration)\n");
      ms_plain = ms_round = 0; n_plain = n_round = 0;
      ms[k] = run(iters, &hist[k]);
      kind_ms[k][0] = n_plain ? ms_plain / n_plain : 0; kind_ms[k][1] = n_round ? ms_round / n_round : 0;
      lpmp_offload::check(lpmp_speculation_stats(e, &stats[k][0], &stats[k][1], &stats[k][2], &stats[k][3]));
    }
    const bool same = hist[0] == hist[1];
    int64_t lv[2] = {0, 0};
    for (int d = 0; d < 2; ++d) lpmp_offload::check(lpmp_plan_schedule_info(lpmp_engine_plan_mut(e), d, 0, &lv[d], nullptr, nullptr, nullptr, nullptr));
    std::printf("{\"tool\": \"offload_solver_loop\", \"order\": \"%s\", \"levels_per_direction\": [%lld, %lld], \"suggest_order_ms\": %.1f, \"colours\": %d, ",
                order.c_str(), (long long)lv[0], (long long)lv[1], suggest_ms, (int)n_colours);
    std::printf("\"grid\": %d, \"labels\": %d, \"iterations\": %d, \"lower_bound_start\": %.17g, \"lower_bound_end\": %.17g, "
                "\"ms_per_iteration_every_call_as_it_comes\": %.4f, \"ms_per_iteration_passes_running_ahead\": %.4f, "
                "\"rounding_cycle\": %s, \"ms_plain_iteration\": [%.4f, %.4f], \"ms_rounding_iteration\": [%.4f, %.4f], \"best_primal_cost\": %s, "
                "\"bound_history_identical\": %s, \"batches\": %lld, \"passes_launched\": %lld, \"passes_used\": %lld, \"rollbacks\": %lld}\n",
                G, L, iters, lb0, hist[1].back(), ms[0], ms[1], rounding ? "true" : "false", kind_ms[0][0], kind_ms[1][0], kind_ms[0][1], kind_ms[1][1],
                (best_primal < std::numeric_limits<double>::infinity() ? std::to_string(best_primal) : std::string("null")).c_str(), same ? "true" : "false",
                (long long)(stats[1][0] - stats[0][0]), (long long)(stats[1][1] - stats[0][1]), (long long)(stats[1][2] - stats[0][2]), (long long)(stats[1][3] - stats[0][3]));
    lp.End();
    return same ? 0 : 1;
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "offload_solver_loop: %s\n", ex.what());
    return 2;
  }
}
