// C++ drop-in test: reads like the reference's own tests
//   test/test_model.cpp:18-48   (3 toy factors, 2 messages, Solver default 1000 iterations, LB = 1.0)
//   test/graphical_model.cpp:90-137 (binary chain / frustrated cycle, LB = 0)
//   test/multicut.cpp:8-32      (labeling_factor lower bounds)
// but is built against lp_mp_amd/include/LP_gpu.hxx and runs the sweep on the MI355X.
// usage: test_model_gpu [--host-only]   (host-only: construction + counts, no device call)
#include <cmath>
#include <cstring>
#include <sstream>
#include <iostream>
#include <stdexcept>

#include "LP_gpu_solver.hxx"   // LP_gpu.hxx (containers, LP) + Solver / StandardVisitor

using namespace LP_MP;

static void test(const bool pred) { if (!pred) throw std::runtime_error("Test failed."); }   // reference test/test.h:7-11

struct test_FMC {   // reference test/test_model.hxx:130-137, verbatim surface
  constexpr static const char* name = "test model";
  using factor = FactorContainer<test_factor, test_FMC, 0>;
  using message = MessageContainer<test_message, 0, 0, message_passing_schedule::left, variableMessageNumber, variableMessageNumber, test_FMC, 0>;
  using FactorList = meta::list<factor>;
  using MessageList = meta::list<message>;
  using ProblemDecompositionList = meta::list<>;
};

struct FMC_SRMP {   // unary / pairwise MRF as LP_MP-MRF declares it
  constexpr static const char* name = "SRMP";
  using UnaryFactor = FactorContainer<UnarySimplexFactor, FMC_SRMP, 0>;
  using PairwiseFactor = FactorContainer<PairwiseSimplexFactor, FMC_SRMP, 1>;
  using UnaryPairwiseMessageLeftContainer = MessageContainer<UnaryPairwiseMessage<Chirality::left>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_SRMP, 0>;
  using UnaryPairwiseMessageRightContainer = MessageContainer<UnaryPairwiseMessage<Chirality::right>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_SRMP, 1>;
  using FactorList = meta::list<UnaryFactor, PairwiseFactor>;
  using MessageList = meta::list<UnaryPairwiseMessageLeftContainer, UnaryPairwiseMessageRightContainer>;
  using ProblemDecompositionList = meta::list<>;
};

struct FMC_SRMP_CONST {   // unary / pairwise MRF plus the reference's ConstantFactor (include/factors/constant_factor.hxx)
  constexpr static const char* name = "SRMP + constant";
  using UnaryFactor = FactorContainer<UnarySimplexFactor, FMC_SRMP_CONST, 0>;
  using PairwiseFactor = FactorContainer<PairwiseSimplexFactor, FMC_SRMP_CONST, 1>;
  using ConstantFactorContainer = FactorContainer<ConstantFactor, FMC_SRMP_CONST, 2>;
  using UnaryPairwiseMessageLeftContainer = MessageContainer<UnaryPairwiseMessage<Chirality::left>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_SRMP_CONST, 0>;
  using UnaryPairwiseMessageRightContainer = MessageContainer<UnaryPairwiseMessage<Chirality::right>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_SRMP_CONST, 1>;
  using FactorList = meta::list<UnaryFactor, PairwiseFactor, ConstantFactorContainer>;
  using MessageList = meta::list<UnaryPairwiseMessageLeftContainer, UnaryPairwiseMessageRightContainer>;
  using ProblemDecompositionList = meta::list<>;
};

struct FMC_SRMP_ROUNDING {   // the same with COMPUTE_PRIMAL_SOLUTION on the unaries, as LP_MP-MRF's FMC_SRMP has it
  constexpr static const char* name = "SRMP with rounding";
  using UnaryFactor = FactorContainer<UnarySimplexFactor, FMC_SRMP_ROUNDING, 0, true>;
  using PairwiseFactor = FactorContainer<PairwiseSimplexFactor, FMC_SRMP_ROUNDING, 1, false>;
  using UnaryPairwiseMessageLeftContainer = MessageContainer<UnaryPairwiseMessage<Chirality::left>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_SRMP_ROUNDING, 0>;
  using UnaryPairwiseMessageRightContainer = MessageContainer<UnaryPairwiseMessage<Chirality::right>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_SRMP_ROUNDING, 1>;
  using FactorList = meta::list<UnaryFactor, PairwiseFactor>;
  using MessageList = meta::list<UnaryPairwiseMessageLeftContainer, UnaryPairwiseMessageRightContainer>;
  using ProblemDecompositionList = meta::list<>;
};

using edge_labelings = labelings<labeling<1>>;
using triplet_labelings = labelings<labeling<0, 1, 1>, labeling<1, 0, 1>, labeling<1, 1, 0>, labeling<1, 1, 1>>;
using multicut_edge_factor = labeling_factor<edge_labelings, true>;
using multicut_triplet_factor = labeling_factor<triplet_labelings, true>;
struct FMC_MULTICUT {
  constexpr static const char* name = "multicut";
  using edge_factor_container = FactorContainer<multicut_edge_factor, FMC_MULTICUT, 0>;
  using triplet_factor_container = FactorContainer<multicut_triplet_factor, FMC_MULTICUT, 1>;
  using m0 = MessageContainer<labeling_message<edge_labelings, triplet_labelings, 0>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_MULTICUT, 0>;
  using m1 = MessageContainer<labeling_message<edge_labelings, triplet_labelings, 1>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_MULTICUT, 1>;
  using m2 = MessageContainer<labeling_message<edge_labelings, triplet_labelings, 2>, 0, 1, message_passing_schedule::left, variableMessageNumber, 1, FMC_MULTICUT, 2>;
  using FactorList = meta::list<edge_factor_container, triplet_factor_container>;
  using MessageList = meta::list<m0, m1, m2>;
  using ProblemDecompositionList = meta::list<>;
};

template <class MRF>
static void add_pairwise(MRF& lp, typename FMC_SRMP::UnaryFactor* u1, typename FMC_SRMP::UnaryFactor* u2, const double c[2][2]) {
  auto* p = lp.template add_factor<typename FMC_SRMP::PairwiseFactor>(2, 2);
  for (INDEX a = 0; a < 2; ++a) for (INDEX b = 0; b < 2; ++b) p->GetFactor()->cost(a, b) = c[a][b];
  lp.template add_message<typename FMC_SRMP::UnaryPairwiseMessageLeftContainer>(u1, p);
  lp.template add_message<typename FMC_SRMP::UnaryPairwiseMessageRightContainer>(u2, p);
  lp.AddFactorRelation(u1, p);
  lp.AddFactorRelation(p, u2);
}

int main(int argc, char** argv) {
  const bool host_only = argc > 1 && std::strcmp(argv[1], "--host-only") == 0;
  {   // ---- StandardVisitor: the iteration line and the --maxMemory stop (reference standard_visitor.hxx:116-128, 152-160) ----
    std::ostringstream line;
    std::streambuf* was = std::cout.rdbuf(line.rdbuf());
    StandardVisitor v({"--maxIter", "100", "-v", "1"});
    int dummy = 0;
    LpControl c = v.begin(dummy);
    c = v.visit(c, 1.0, 9.0);                               // a plain iteration: bound only
    LpControl p = c; p.computePrimal = true; p.computeLowerBound = true;
    (void)v.visit(p, 2.0, 7.5);                             // a rounding iteration: bound and primal
    std::cout.rdbuf(was);
    const std::string out = line.str();
    test(out.find("iteration = 0, lower bound = 1, time elapsed") != std::string::npos);
    test(out.find("iteration = 1, lower bound = 2, upper bound = 7.5, time elapsed") != std::string::npos);
    StandardVisitor m({"--maxIter", "100", "--maxMemory", "1"});   // 1 MB: less than any process here uses
    c = m.begin(dummy);
    c = m.visit(c, 1.0, 9.0);
    test(!c.end && c.computePrimal && c.computeLowerBound);   // one iteration remaining: the rounding one
    c = m.visit(c, 2.0, 9.0);
    test(c.end);
    StandardVisitor roomy({"--maxIter", "100", "--maxMemory", "100000000"});
    c = roomy.begin(dummy);
    c = roomy.visit(c, 1.0, 9.0);
    test(!c.end && c.computePrimal);                           // iteration 1 is a rounding iteration by the interval rule only
    c = roomy.visit(c, 2.0, 9.0);
    test(!c.end && !c.computePrimal);
  }
  {   // ---- test/test_model.cpp ----
    Solver<LP<test_FMC>, StandardVisitor> s;
    auto& lp = s.GetLP();
    auto* f1 = lp.template add_factor<typename test_FMC::factor>(0, 1);
    auto* f2 = lp.template add_factor<typename test_FMC::factor>(1, 0);
    auto* f3 = lp.template add_factor<typename test_FMC::factor>(0, 0);
    lp.template add_message<typename test_FMC::message>(f1, f2);
    lp.template add_message<typename test_FMC::message>(f1, f3);
    test(lp.GetNumberOfFactors() == 3);
    test(lp.GetNumberOfMessages() == 2);
    test(lp.GetFactor(0) == f1);
    test(lp.GetFactor(1) == f2);
    test(lp.GetFactor(2) == f3);
    test(f1->no_messages() == 2);
    test(f1->no_send_messages() == 2);
    test(f2->no_messages() == 1);
    test(f2->no_send_messages() == 0);
    test(f3->no_messages() == 1);
    test(f3->no_send_messages() == 0);
    if (!host_only) {
      std::cout << "lower bound before optimization = " << s.GetLP().LowerBound() << "\n";
      s.Solve();
      std::cout << "lower bound after optimization = " << s.GetLP().LowerBound() << "\n";
      test(std::abs(s.GetLP().LowerBound() - 1.0) <= eps);
      test(s.iter == 1000);
      // the factor ops hold the reparametrised costs after End(), as in the reference
      test(std::abs((*f1->GetFactor())[0] - 1.0) <= eps && std::abs((*f1->GetFactor())[1] - 1.0) <= eps);
      // call-order error of the reference (LP_MP.h:458)
      bool threw = false;
      lp.Begin();
      try { lp.ComputePass(0); } catch (const std::runtime_error&) { threw = true; }
      test(threw);
    }
  }
  {   // ---- test/simplex.cpp, test/potts_factor.cpp, test/multicut.cpp: factor op known answers (host) ----
    UnarySimplexFactor simplex(std::vector<double>{0.1, 0.2, 0.05, 1});
    test(simplex.LowerBound() == 0.05);
    PairwiseSimplexFactor pw(3, 3);
    for (INDEX x1 = 0; x1 < 3; ++x1) for (INDEX x2 = 0; x2 < 3; ++x2) pw.cost(x1, x2) = x1 != x2 ? 0 : -REAL(x1) - 1.0;
    test(pw.LowerBound() == -3.0);
    pairwise_potts_factor potts(3, 1.0);
    PairwiseSimplexFactor potts2(3, 3);
    for (INDEX a = 0; a < 3; ++a) for (INDEX b = 0; b < 3; ++b) potts2.cost(a, b) = a == b ? 0.0 : 1.0;
    potts.msg1(0) = -0.1; potts2.msg1(0) = -0.1; potts.msg1(1) = 0.5; potts2.msg1(1) = 0.5; potts.msg1(2) = 0.8; potts2.msg1(2) = 0.8;
    potts.msg2(0) = 1.5; potts2.msg2(0) = 1.5; potts.msg2(1) = 1.0; potts2.msg2(1) = 1.0;
    test(potts.LowerBound() == potts2.LowerBound());
    multicut_edge_factor unary;
    unary[0] = 1.0; test(unary.size() == 1); test(unary.LowerBound() == 0);
    unary[0] = -1.0; test(unary.LowerBound() == -1);
    multicut_triplet_factor triangle;
    test(triangle.size() == 4);
    triangle[0] = 1.0; triangle[1] = 2.0; triangle[2] = 3.3; triangle[3] = 1.5; test(triangle.LowerBound() == 0.0);
    triangle[1] = -0.5; triangle[2] = -0.3; test(triangle.LowerBound() == -0.5);
    using msg0 = labeling_message<edge_labelings, triplet_labelings, 0>;
    const auto tab = msg0::match_table();
    test(tab.size() == 4 && tab[0] == 1 && tab[1] == 0 && tab[2] == 0 && tab[3] == 0);
  }
  const double negPotts[2][2] = {{1.0, 0.0}, {0.0, 1.0}};
  const double posPotts[2][2] = {{0.0, 1.0}, {1.0, 0.0}};
  for (int cycle = 0; cycle < 2; ++cycle) {   // ---- test/graphical_model.cpp:90-137 ----
    Solver<LP<FMC_SRMP>, StandardVisitor> s(std::vector<std::string>{"--maxIter", "100", "--standardReparametrization", "anisotropic"});
    auto& lp = s.GetLP();
    std::vector<typename FMC_SRMP::UnaryFactor*> u;
    for (int i = 0; i < (cycle ? 4 : 5); ++i) u.push_back(lp.template add_factor<typename FMC_SRMP::UnaryFactor>(std::vector<REAL>(2, 0.0)));
    add_pairwise(lp, u[0], u[1], negPotts);
    add_pairwise(lp, u[1], u[2], posPotts);
    add_pairwise(lp, u[2], u[3], posPotts);
    if (cycle) add_pairwise(lp, u[0], u[3], posPotts); else add_pairwise(lp, u[3], u[4], posPotts);
    test(lp.GetNumberOfFactors() == (cycle ? 8u : 9u));
    if (!host_only) {
      s.Solve();
      test(std::abs(s.lower_bound() - 0.0) <= eps);
      auto om = lp.get_omega();
      test(om.forward.size() == u.size());
    }
  }
  {   // ---- test/factor_message_containers.cpp:8-88: message counts of the unaries of a 5-variable MRF ----
    LP<FMC_SRMP> lp;
    std::vector<typename FMC_SRMP::UnaryFactor*> u;
    for (int i = 0; i < 5; ++i) u.push_back(lp.template add_factor<typename FMC_SRMP::UnaryFactor>(std::vector<REAL>(2, 0.0)));
    const double zero[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    const int edges[8][2] = {{0, 1}, {1, 2}, {2, 3}, {0, 3}, {0, 4}, {1, 4}, {2, 4}, {3, 4}};
    for (auto& e : edges) add_pairwise(lp, u[e[0]], u[e[1]], zero);
    test(u[0]->no_messages() == 3 && u[0]->no_send_messages() == 3);      // (GetNoMessages / no_send_messages there)
    test(u[4]->no_messages() == 4 && u[4]->no_send_messages() == 4);
    test(lp.GetNumberOfFactors() == 13 && lp.GetNumberOfMessages() == 16);
  }
  {   // ---- ConstantFactor: an offset that takes part in the bound and is never touched by the sweep ----
    using FMC = FMC_SRMP_CONST;
    ConstantFactor cf(2.5);
    test(cf.size() == 0 && cf.LowerBound() == 2.5);
    cf.AddToOffset(-1.0);
    test(cf.LowerBound() == 1.5 && cf.EvaluatePrimal() == 1.5);
    LP<FMC> lp;
    auto* u1 = lp.template add_factor<typename FMC::UnaryFactor>(std::vector<REAL>{0.0, 1.0});
    auto* u2 = lp.template add_factor<typename FMC::UnaryFactor>(std::vector<REAL>{1.0, 0.0});
    auto* p = lp.template add_factor<typename FMC::PairwiseFactor>(2, 2);
    p->GetFactor()->cost(0, 1) = 1.0; p->GetFactor()->cost(1, 0) = 1.0;
    auto* c = lp.template add_factor<typename FMC::ConstantFactorContainer>(1.5);
    lp.template add_message<typename FMC::UnaryPairwiseMessageLeftContainer>(u1, p);
    lp.template add_message<typename FMC::UnaryPairwiseMessageRightContainer>(u2, p);
    lp.AddFactorRelation(u1, p); lp.AddFactorRelation(p, u2);
    test(c->no_messages() == 0);
    if (!host_only) {
      lp.Begin();
      lp.set_reparametrization(LPReparametrizationMode::Anisotropic);
      test(std::abs(lp.LowerBound() - 1.5) <= eps);           // 0 + 0 + 0 + offset
      for (int it = 0; it < 5; ++it) lp.ComputePass(it);
      test(std::abs(lp.LowerBound() - 2.5) <= eps);           // optimum of the pair: 1, plus the offset
      lp.End();
      test(c->GetFactor()->LowerBound() == 1.5);
    }
  }
  {   // ---- quiet iterations: what the solver may run as one device call is exactly what the visits would return ----
    const char* combos[][6] = {{"--maxIter", "40", "--lowerBoundComputationInterval", "7", "--primalComputationInterval", "11"},
                               {"--maxIter", "25", "--lowerBoundComputationInterval", "1", "--primalComputationInterval", "5"},
                               {"--maxIter", "33", "--lowerBoundComputationInterval", "100", "--primalComputationInterval", "3"},
                               {"--maxIter", "3", "--lowerBoundComputationInterval", "2", "--primalComputationInterval", "50"}};
    for (auto& o : combos) {
      StandardVisitor v(std::vector<std::string>(o, o + 6));
      int dummy = 0;
      LpControl c = v.begin(dummy);
      int visits = 0, batched = 0;
      while (!c.end) {
        const INDEX q = v.quiet_iterations(c);
        test(q >= 1);
        const LpControl first = c;
        for (INDEX j = 0; j < q; ++j) {
          // every iteration of the run sees a control that asks for nothing (unless the run is the single iteration)
          if (q > 1) { test(!c.computeLowerBound && !c.computePrimal && !c.end && c.repam == first.repam); ++batched; }
          c = v.visit(c, 0.0, std::numeric_limits<REAL>::infinity());
          ++visits;
        }
        // and the run is maximal: what follows is not another quiet iteration of the same kind
        if (q > 1) test(c.end || c.computeLowerBound || c.computePrimal);
      }
      test(visits == std::atoi(o[1]));
      if (std::atoi(o[3]) == 7) test(batched > 20);
    }
  }
  for (int interval : {1, 6}) {   // ---- a solve with batched quiet iterations ends where the unbatched one does ----
    static REAL lb_ref = 0.0; static std::size_t hist_ref = 0;
    Solver<LP<FMC_SRMP>, StandardVisitor> s(std::vector<std::string>{"--maxIter", "31", "--standardReparametrization", "anisotropic",
                                                                      "--lowerBoundComputationInterval", std::to_string(interval)});
    auto& lp = s.GetLP();
    std::vector<typename FMC_SRMP::UnaryFactor*> u;
    for (int i = 0; i < 6; ++i) u.push_back(lp.template add_factor<typename FMC_SRMP::UnaryFactor>(std::vector<REAL>{0.1 * i, 0.3 - 0.05 * i}));
    const double c1[2][2] = {{0.7, 0.1}, {0.2, 0.9}}, c2[2][2] = {{0.0, 0.6}, {0.8, 0.3}};
    for (int i = 0; i + 1 < 6; ++i) add_pairwise(lp, u[i], u[i + 1], i % 2 ? c1 : c2);
    add_pairwise(lp, u[0], u[5], c1);
    if (!host_only) {
      s.Solve();
      test(s.iter == 31);
      if (interval == 1) { lb_ref = s.lower_bound(); hist_ref = s.GetVisitor().lower_bound_history().size(); }
      else { test(s.lower_bound() == lb_ref); test(s.GetVisitor().lower_bound_history().size() == hist_ref); }
    }
  }
  {   // ---- MpRoundingSolver (reference solver.hxx:380-400) on a chain with a unique optimum ----
    using FMC = FMC_SRMP_ROUNDING;
    MpRoundingSolver<Solver<LP<FMC>, StandardVisitor>> s(std::vector<std::string>{
        "--maxIter", "40", "--primalComputationInterval", "5", "--standardReparametrization", "anisotropic",
        "--roundingReparametrization", "anisotropic"});
    auto& lp = s.GetLP();
    const int n = 6, L = 3;
    std::vector<typename FMC::UnaryFactor*> u;
    // unary i prefers label i % 3 by 0.1; every edge adds 1 unless x_{i+1} = (x_i + 1) % 3: optimum 0,1,2,0,1,2 at cost 0
    for (int i = 0; i < n; ++i) {
      std::vector<REAL> c(L, 0.1); c[i % L] = 0.0;
      u.push_back(lp.template add_factor<typename FMC::UnaryFactor>(c));
    }
    for (int i = 0; i + 1 < n; ++i) {
      auto* p = lp.template add_factor<typename FMC::PairwiseFactor>(L, L);
      for (int a = 0; a < L; ++a) for (int b = 0; b < L; ++b) p->GetFactor()->cost(a, b) = b == (a + 1) % L ? 0.0 : 1.0;
      lp.template add_message<typename FMC::UnaryPairwiseMessageLeftContainer>(u[i], p);
      lp.template add_message<typename FMC::UnaryPairwiseMessageRightContainer>(u[i + 1], p);
      lp.AddFactorRelation(u[i], p);
      lp.AddFactorRelation(p, u[i + 1]);
    }
    if (!host_only) {
      s.Solve();
      test(std::abs(s.lower_bound() - 0.0) <= eps);
      test(std::abs(s.primal_cost() - 0.0) <= eps);           // the visitor stops at primal <= lower bound + eps
      test(s.iter < 40);
      const auto& x = s.solution();
      test(x.size() == (std::size_t)(2 * n - 1));
      for (int i = 0; i < n; ++i) test(x[i][0] == i % L);
      for (int i = 0; i + 1 < n; ++i) test(x[n + i][0] == i % L && x[n + i][1] == (i + 1) % L);
      test(lp.CheckPrimalConsistency());
    }
  }
  if (!host_only) {   // ---- multicut-style triangle through the labeling-list family ----
    LP<FMC_MULTICUT> lp;
    const double ec[3] = {-1.0, 2.0, 2.0};
    typename FMC_MULTICUT::edge_factor_container* e[3];
    for (int k = 0; k < 3; ++k) { e[k] = lp.template add_factor<typename FMC_MULTICUT::edge_factor_container>(); (*e[k]->GetFactor())[0] = ec[k]; }
    auto* t = lp.template add_factor<typename FMC_MULTICUT::triplet_factor_container>();
    lp.template add_message<typename FMC_MULTICUT::m0>(e[0], t);
    lp.template add_message<typename FMC_MULTICUT::m1>(e[1], t);
    lp.template add_message<typename FMC_MULTICUT::m2>(e[2], t);
    for (int k = 0; k < 3; ++k) lp.AddFactorRelation(e[k], t);
    lp.Begin();
    lp.set_reparametrization(LPReparametrizationMode::Anisotropic);
    const double lb0 = lp.LowerBound();
    test(lb0 == -1.0);
    for (int it = 0; it < 50; ++it) lp.ComputePass(it);
    // cutting only edge 0 (relaxed optimum -1) is not a multicut of a triangle: the best labeling is the all-zero one
    test(lp.LowerBound() >= lb0 - eps && lp.LowerBound() <= 0.0 + eps);
    test(std::abs(lp.LowerBound() - 0.0) <= 1e-6);
  }
  if (!host_only) {   // ---- a grid inserted row by row, then run in the order the engine suggests (LP::apply_suggested_order) ----
    using FMC = FMC_SRMP_ROUNDING;
    const int H = 9, W = 8, L = 3;
    auto build = [&](LP<FMC>& lp) {
      std::vector<typename FMC::UnaryFactor*> u;
      for (int i = 0; i < H * W; ++i) { std::vector<REAL> c(L); for (int a = 0; a < L; ++a) c[a] = ((i * 7 + a * 3) % 11) / 11.0; u.push_back(lp.template add_factor<typename FMC::UnaryFactor>(c)); }
      auto edge = [&](int a, int b) {
        auto* p = lp.template add_factor<typename FMC::PairwiseFactor>(L, L);
        for (int x = 0; x < L; ++x) for (int y = 0; y < L; ++y) p->GetFactor()->cost(x, y) = ((a + 2 * b + 5 * x + 3 * y) % 7) / 7.0;
        lp.template add_message<typename FMC::UnaryPairwiseMessageLeftContainer>(u[a], p);
        lp.template add_message<typename FMC::UnaryPairwiseMessageRightContainer>(u[b], p);
        lp.AddFactorRelation(u[a], p);
        lp.AddFactorRelation(p, u[b]);
      };
      for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) { if (c + 1 < W) edge(r * W + c, r * W + c + 1); if (r + 1 < H) edge(r * W + c, (r + 1) * W + c); }
    };
    LP<FMC> row_major, suggested;
    build(row_major); build(suggested);
    for (LP<FMC>* lp : {&row_major, &suggested}) { lp->Begin(); lp->set_reparametrization(LPReparametrizationMode::Anisotropic); }
    const double lb0 = row_major.LowerBound();
    test(std::abs(suggested.LowerBound() - lb0) <= eps);                 // same problem
    test(row_major.forward_update_ordering().size() == (std::size_t)(H * W));
    test(suggested.apply_suggested_order() == 2);                        // a grid is 2-colourable: 2 dependent levels per sweep instead of H + W - 1
    suggested.set_reparametrization(LPReparametrizationMode::Anisotropic);
    // black cells first, then white ones: the updated factors colour by colour
    const auto order = suggested.forward_update_ordering();
    test(order.size() == (std::size_t)(H * W));
    bool colour_major = true;
    for (std::size_t i = 0; i < order.size(); ++i) { const int cell = (int)order[i]->index_; const bool black = ((cell / W + cell % W) & 1) == 0; if (black != (i < (std::size_t)((H * W + 1) / 2))) colour_major = false; }
    test(colour_major);
    for (int it = 0; it < 30; ++it) { row_major.ComputePass(it); suggested.ComputePass(it); }
    // two trajectories of the same dual ascent: both bounds ascend from lb0 and end close to each other
    test(row_major.LowerBound() > lb0 && suggested.LowerBound() > lb0);
    test(std::abs(row_major.LowerBound() - suggested.LowerBound()) <= 0.05 * std::abs(row_major.LowerBound()));
  }
  std::cout << "all tests passed\n";
  return 0;
}
