// test_offload_mock.cpp — lp_mp_amd/include/lpmp_offload.hxx against a reference-SHAPED LP.
//
// The LP, its containers and the factor / message ops below are NOT the mirror of LP_gpu.hxx: they come from
// tests/cpp/mock_reference_lp.hxx (the surface of the reference's LP<FMC> the offload header uses) and from op classes
// written here the way a user of the reference writes them — LowerBound(), template serialize_dual(ARCHIVE&), no
// device_kind member anywhere.  The kinds are registered outside the ops.  The same TU also includes LP_gpu.hxx and
// LP_gpu_solver.hxx: they define nothing in namespace LP_MP, so they coexist with reference-shaped headers.
//
//   usage: test_offload_mock [--host-only]   (host-only: construction + flattening, no device call)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <stdexcept>

#include "mock_reference_lp.hxx"    // namespace LP_MP: reference-shaped LP (test double)
#include "lpmp_offload.hxx"
#include "LP_gpu_solver.hxx"        // namespace LP_MP_gpu only: no clash with the above

static void test(const bool pred, const char* what = "") { if (!pred) throw std::runtime_error(std::string("Test failed: ") + what); }

// ---- ops as a user of the reference writes them ---------------------------------------------------------------------
namespace user {
using LP_MP::REAL; using LP_MP::INDEX;
struct my_vector : std::vector<REAL> {          // like the reference's vector<REAL>: begin() + size() ...
  using std::vector<REAL>::vector;
  // ... AND, inherited from its expression-template bases (reference vector.hxx:20-24, 45-47, 69-72), dim1() / dim2() and a
  // BY-VALUE two-index call operator: it must still be packed as a range, not as a matrix (the real headers showed this,
  // tools/check_offload_against_reference.sh)
  INDEX dim1() const { return size(); } INDEX dim2() const { return 1; }
  REAL operator()(INDEX i1, INDEX) const { return (*this)[i1]; }
};
struct my_matrix {                              // like the reference's matrix<REAL>: dim1() / dim2() / operator()(i, j)
  my_matrix(INDEX a, INDEX b) : d1_(a), d2_(b), v_(a * b, 0.0) {}
  INDEX dim1() const { return d1_; } INDEX dim2() const { return d2_; }
  REAL& operator()(INDEX i, INDEX j) { return v_[i * d2_ + j]; }
  REAL operator()(INDEX i, INDEX j) const { return v_[i * d2_ + j]; }
  INDEX d1_, d2_; std::vector<REAL> v_;
};
class test_factor {                              // reference test/test_model.hxx:10-64
 public:
  test_factor(REAL x, REAL y) : cost{{x, y}} {}
  REAL LowerBound() const { return std::min(cost[0], cost[1]); }
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(cost); }
  std::array<REAL, 2> cost;
};
struct test_message {};                          // reference test/test_model.hxx:66-98
class Unary {                                    // Unary{vector<REAL> c}
 public:
  explicit Unary(const std::vector<REAL>& c) : c(c.begin(), c.end()) {}
  REAL LowerBound() const { return *std::min_element(c.begin(), c.end()); }
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(c); }
  my_vector c;
};
class Pairwise {                                 // Pairwise{matrix pw; vector m1, m2}
 public:
  Pairwise(INDEX a, INDEX b) : pw(a, b), m1(a, 0.0), m2(b, 0.0) {}
  REAL LowerBound() const {
    REAL lb = std::numeric_limits<REAL>::infinity();
    for (INDEX a = 0; a < pw.dim1(); ++a) {
      REAL mn = std::numeric_limits<REAL>::infinity();
      for (INDEX b = 0; b < pw.dim2(); ++b) mn = std::min(mn, pw(a, b) + m2[b]);
      lb = std::min(lb, m1[a] + mn);
    }
    return lb;
  }
  // this factor lists its table among the duals too, as a reparametrised matrix would be: the registration below
  // tells the device which part is the (constant) table and which the message vectors
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(m1, m2, pw); }
  my_matrix pw; my_vector m1, m2;
};
template <LP_MP::Chirality C> struct UPMsg {};
}  // namespace user

// ---- kind registration, outside the ops ----------------------------------------------------------------------------
namespace lpmp_offload {
template <> struct device_kind<user::test_factor> : vector_kind<> {};
template <> struct device_kind<user::Unary> : vector_kind<> {};
template <> struct device_kind<user::Pairwise> : pairwise_dense_kind<user::Pairwise> {
  static std::size_t dim1(const user::Pairwise& f) { return f.pw.dim1(); }
  static std::size_t dim2(const user::Pairwise& f) { return f.pw.dim2(); }
  static double table(const user::Pairwise& f, std::size_t a, std::size_t b) { return f.pw(a, b); }
  // serialize_dual lists (m1, m2, pw): only the leading m1 | m2 are duals on the device
  static void export_dual(user::Pairwise& f, double* out) { out = std::copy(f.m1.begin(), f.m1.end(), out); std::copy(f.m2.begin(), f.m2.end(), out); }
  static void import_dual(user::Pairwise& f, const double* in) { std::copy(in, in + f.m1.size(), f.m1.begin()); std::copy(in + f.m1.size(), in + f.m1.size() + f.m2.size(), f.m2.begin()); }
};
template <> struct device_message<user::test_message> : min_normalised_message<> {};
template <LP_MP::Chirality C> struct device_message<user::UPMsg<C>> : unary_pairwise_message<C == LP_MP::Chirality::left ? 0 : 1> {};
}  // namespace lpmp_offload

// ---- FMCs, written against the (mock) reference's containers ------------------------------------------------------
struct test_FMC {   // reference test/test_model.hxx:130-137
  constexpr static const char* name = "test model";
  using factor = LP_MP::FactorContainer<user::test_factor, test_FMC, 0>;
  using message = LP_MP::MessageContainer<user::test_message, 0, 0, LP_MP::message_passing_schedule::left, LP_MP::variableMessageNumber, LP_MP::variableMessageNumber, test_FMC, 0>;
  using FactorList = LP_MP::meta::list<factor>;
  using MessageList = LP_MP::meta::list<message>;
};
struct FMC_MRF {
  using U = LP_MP::FactorContainer<user::Unary, FMC_MRF, 0, true>;
  using P = LP_MP::FactorContainer<user::Pairwise, FMC_MRF, 1>;
  using ML = LP_MP::MessageContainer<user::UPMsg<LP_MP::Chirality::left>, 0, 1, LP_MP::message_passing_schedule::left, LP_MP::variableMessageNumber, 1, FMC_MRF, 0>;
  using MR = LP_MP::MessageContainer<user::UPMsg<LP_MP::Chirality::right>, 0, 1, LP_MP::message_passing_schedule::left, LP_MP::variableMessageNumber, 1, FMC_MRF, 1>;
  using FactorList = LP_MP::meta::list<U, P>;
  using MessageList = LP_MP::meta::list<ML, MR>;
};
// the same problem on the standalone mirror (LP_gpu.hxx), to compare against
struct FMC_MIRROR {
  using U = LP_MP_gpu::FactorContainer<LP_MP_gpu::UnarySimplexFactor, FMC_MIRROR, 0, true>;
  using P = LP_MP_gpu::FactorContainer<LP_MP_gpu::PairwiseSimplexFactor, FMC_MIRROR, 1>;
  using ML = LP_MP_gpu::MessageContainer<LP_MP_gpu::UnaryPairwiseMessage<LP_MP_gpu::Chirality::left>, 0, 1, LP_MP_gpu::message_passing_schedule::left, LP_MP_gpu::variableMessageNumber, 1, FMC_MIRROR, 0>;
  using MR = LP_MP_gpu::MessageContainer<LP_MP_gpu::UnaryPairwiseMessage<LP_MP_gpu::Chirality::right>, 0, 1, LP_MP_gpu::message_passing_schedule::left, LP_MP_gpu::variableMessageNumber, 1, FMC_MIRROR, 1>;
  using FactorList = LP_MP_gpu::meta::list<U, P>;
  using MessageList = LP_MP_gpu::meta::list<ML, MR>;
};

static double u01(uint64_t& st) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(st >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
  const bool host_only = argc > 1 && !std::strcmp(argv[1], "--host-only");
  LP_MP::mock_cmd_line cmd;
  {   // reference test/serialization.cpp:16-124 against the offload's archives (allocate / save / load, in doubles): the
      // same member kinds, the same sizes (a 5 x 3 matrix is 15 entries: padding stripped), the same round trips
    using user::REAL; using user::INDEX;
    struct bin { REAL* pointer; INDEX no_elements; };             // serialization.hxx binary_data<T>
    REAL i_r = 10; std::array<REAL, 2> a_r{20, 30}; std::vector<REAL> v_r{30, 40, 50}; REAL p_r[4] = {60, 70, 80, 90};
    user::my_vector r_r{1.5, 2.5, 3.5};
    user::my_matrix m_r(5, 3);
    for (INDEX a = 0; a < 5; ++a) { m_r(a, 0) = 1.0 + a; m_r(a, 1) = 11.0 + a; m_r(a, 2) = 111.0 + a; }
    auto size_of = [](auto&& e) { lpmp_offload::dual_counter c; c(e); return c.count; };
    test(size_of(i_r) == 1 && size_of(a_r) == 2 && size_of(v_r) == 3 && size_of(bin{p_r, 4}) == 4 && size_of(r_r) == 3 && size_of(m_r) == 15, "archive sizes");
    lpmp_offload::dual_counter all; all(i_r, a_r, v_r, bin{p_r, 4}, r_r, m_r);
    test(all.count == 28, "collective size");
    for (int collective = 0; collective < 2; ++collective) {
      std::vector<double> ar(all.count, -1.0);
      lpmp_offload::dual_saver sv(ar.data());
      if (collective) sv(i_r, a_r, v_r, bin{p_r, 4}, r_r, m_r);
      else { sv(i_r); sv(a_r); sv(v_r); sv(bin{p_r, 4}); sv(r_r); sv(m_r); }
      test(sv.out == ar.data() + ar.size() && ar[0] == 10 && ar[3] == 30 && ar[6] == 60 && ar[13] == 1.0 && ar[14] == 11.0 && ar[15] == 111.0 && ar[27] == 115.0, "saved layout");
      REAL i_t = 0; std::array<REAL, 2> a_t{0, 0}; std::vector<REAL> v_t(3); REAL p_t[4] = {0, 0, 0, 0}; user::my_vector r_t(3); user::my_matrix m_t(5, 3);
      lpmp_offload::dual_loader ld(ar.data());
      if (collective) ld(i_t, a_t, v_t, bin{p_t, 4}, r_t, m_t);
      else { ld(i_t); ld(a_t); ld(v_t); ld(bin{p_t, 4}); ld(r_t); ld(m_t); }
      test(i_t == i_r && a_t == a_r && v_t == v_r && p_t[0] == 60 && p_t[3] == 90 && r_t == r_r && m_t.v_ == m_r.v_, "round trip");
    }
  }
  {   // reference test/test_model.cpp:18-48 on the offloaded LP
    using LP_device = lpmp_offload::offloaded<LP_MP::LP<test_FMC>>;
    LP_device lp(cmd);
    auto* f1 = lp.add_factor<test_FMC::factor>(0, 1);
    auto* f2 = lp.add_factor<test_FMC::factor>(1, 0);
    auto* f3 = lp.add_factor<test_FMC::factor>(0, 0);
    lp.add_message<test_FMC::message>(f1, f2);
    lp.add_message<test_FMC::message>(f2, f3);
    lp.AddFactorRelation(f1, f2); lp.AddFactorRelation(f2, f3);
    test(lp.GetNumberOfFactors() == 3 && lp.GetNumberOfMessages() == 2, "counts");
    if (!host_only) {
      lp.Begin();
      bool threw = false;
      try { lp.ComputePass(0); } catch (const std::runtime_error&) { threw = true; }   // no mode yet (LP_MP.h:458)
      test(threw, "ComputePass before set_reparametrization must throw");
      lp.set_reparametrization(LP_MP::LPReparametrizationMode::Anisotropic);
      for (int it = 0; it < 1000; ++it) lp.ComputePass(it);
      test(std::abs(lp.LowerBound() - 1.0) <= 1e-8, "toy model lower bound 1.0");
      lp.End();                                   // duals back in the factor ops through serialize_dual
      const double host_lb = f1->LowerBound() + f2->LowerBound() + f3->LowerBound();
      test(std::abs(host_lb - 1.0) <= 1e-8, "factor ops hold the device's duals after End()");
      const double sum = f1->GetFactor()->cost[0] + f1->GetFactor()->cost[1] + f2->GetFactor()->cost[0] + f2->GetFactor()->cost[1] +
                         f3->GetFactor()->cost[0] + f3->GetFactor()->cost[1];
      test(std::isfinite(sum), "finite duals");
    } else {
      test(lp.flat_model_host_only().f_type.size() == 3, "flattened without a device");
    }
  }
  {   // a grid MRF: offloaded reference-shaped LP against the standalone mirror, same costs
    const int H = 7, W = 6, L = 5;
    using LP_device = lpmp_offload::offloaded<LP_MP::LP<FMC_MRF>>;
    for (const char* rtype : {"shared", "residual", "partition"}) {
      LP_device lp(cmd, rtype, 2);
      LP_MP_gpu::LP_gpu<FMC_MIRROR> mirror(0);
      mirror.set_reparametrization_type(rtype); mirror.set_inner_iterations(2);
      std::vector<FMC_MRF::U*> u; std::vector<FMC_MIRROR::U*> um;
      std::vector<FMC_MRF::P*> ps;
      uint64_t st = 42;
      for (int i = 0; i < H * W; ++i) {
        std::vector<double> c(L);
        for (auto& x : c) x = u01(st);
        u.push_back(lp.add_factor<FMC_MRF::U>(c));
        um.push_back(mirror.add_factor<FMC_MIRROR::U>(c));
      }
      auto edge = [&](int a, int b) {
        auto* p = lp.add_factor<FMC_MRF::P>(L, L);
        auto* pm = mirror.add_factor<FMC_MIRROR::P>(L, L);
        for (int x = 0; x < L; ++x) for (int y = 0; y < L; ++y) { const double v = u01(st); p->GetFactor()->pw(x, y) = v; pm->GetFactor()->cost(x, y) = v; }
        lp.add_message<FMC_MRF::ML>(u[a], p); lp.add_message<FMC_MRF::MR>(u[b], p);
        lp.AddFactorRelation(u[a], p); lp.AddFactorRelation(p, u[b]);
        mirror.add_message<FMC_MIRROR::ML>(um[a], pm); mirror.add_message<FMC_MIRROR::MR>(um[b], pm);
        mirror.AddFactorRelation(um[a], pm); mirror.AddFactorRelation(pm, um[b]);
        if ((a % W) / 3 == (b % W) / 3) { lp.put_in_same_partition(u[a], u[b]); mirror.put_in_same_partition(um[a], um[b]); }
        ps.push_back(p);
      };
      for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) {
        if (c + 1 < W) edge(r * W + c, r * W + c + 1);
        if (r + 1 < H) edge(r * W + c, (r + 1) * W + c);
      }
      lp.add_to_constant(0.25); mirror.add_to_constant(0.25);
      if (host_only) {
        const auto& m = lp.flat_model_host_only();
        test((int)m.f_type.size() == H * W + (int)ps.size() && m.m_type.size() == 2 * ps.size(), "flat model sizes");
        test(m.cdata.size() == ps.size() * L * L && m.dual.size() == (size_t)H * W * L + ps.size() * 2 * L, "flat model arrays");
        test(m.mtypes.size() == 2 && m.mtypes[0].schedule == LPMP_SCHED_LEFT && m.mtypes[1].param == 1 && m.mtypes[0].n_left == 0 && m.mtypes[0].n_right == 1, "message types");
        test(m.ftype_primal[0] == 1 && m.ftype_primal[1] == 0 && m.constant == 0.25, "factor types");
        continue;
      }
      lp.Begin(); mirror.Begin();
      lp.set_reparametrization(LP_MP::LPReparametrizationMode::Anisotropic);
      mirror.set_reparametrization(LP_MP_gpu::LPReparametrizationMode::Anisotropic);
      test(lp.LowerBound() == mirror.LowerBound(), "initial bound");
      for (int it = 0; it < 5; ++it) { lp.ComputePass(it); mirror.ComputePass(it); }
      test(lp.LowerBound() == mirror.LowerBound(), "bound after 5 passes equals the mirror's");
      lp.set_reparametrization(LP_MP::LPReparametrizationMode::DampedUniform);      // what the visitor does on rounding iterations
      mirror.set_reparametrization(LP_MP_gpu::LPReparametrizationMode::DampedUniform);
      lp.ComputeForwardPassAndPrimal(5); mirror.ComputeForwardPassAndPrimal(5);
      lp.ComputeBackwardPassAndPrimal(5); mirror.ComputeBackwardPassAndPrimal(5);
      test(lp.CheckPrimalConsistency() && lp.EvaluatePrimal() == mirror.EvaluatePrimal(), "rounded cost equals the mirror's");
      test(lp.EvaluatePrimal() >= lp.LowerBound() - 1e-9, "primal above dual");
      // structural change on a live LP: one more edge; the duals reached so far must survive the re-upload
      const double lb_before = lp.LowerBound();
      edge(0, W + 1);
      test(std::abs(lp.LowerBound() - mirror.LowerBound()) <= 1e-12 && lp.LowerBound() >= lb_before - 1e-9, "bound after adding a factor");
      lp.ComputePass(6); mirror.ComputePass(6);
      test(lp.LowerBound() == mirror.LowerBound(), "bound after the structural change equals the mirror's");
      lp.End(); mirror.End();
      double host_lb = 0.25;
      for (auto* f : u) host_lb += f->LowerBound();
      for (auto* p : ps) host_lb += p->LowerBound();
      test(std::abs(host_lb - lp.LowerBound()) <= 1e-9 * std::max(1.0, std::abs(host_lb)), "factor ops hold the device's duals after End()");
    }
  }
  if (!host_only) {   // the reference's Solve loop (solver.hxx:238-243, 268-284) through the adapter, with and without passes running ahead
    // a 2-colour (checkerboard) variable order: consecutive passes join, so batches of passes run as one launch
    // (LPMP_ROT_BANDS forces that on this small model); the bound after every iteration must be the same either way
    setenv("LPMP_ROT_BANDS", "4", 1);
    const int H = 18, W = 14, L = 8;
    std::vector<double> hist[2];
    for (int k = 0; k < 2; ++k) {
      using LP_device = lpmp_offload::offloaded<LP_MP::LP<FMC_MRF>>;
      LP_device lp(cmd);
      lp.set_speculation(k == 0 ? 0 : 8);
      std::vector<int> pos((size_t)H * W);
      { int b = 0, w = (H * W + 1) / 2; for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) pos[(size_t)r * W + c] = ((r + c) & 1) == 0 ? b++ : w++; }
      std::vector<int> cell_of((size_t)H * W);
      for (int i = 0; i < H * W; ++i) cell_of[pos[i]] = i;
      std::vector<FMC_MRF::U*> u((size_t)H * W);
      uint64_t st = 7;
      for (int i = 0; i < H * W; ++i) { std::vector<double> c(L); for (auto& x : c) x = u01(st); u[(size_t)cell_of[i]] = lp.add_factor<FMC_MRF::U>(c); }
      auto edge = [&](int a, int b) {
        if (pos[a] > pos[b]) std::swap(a, b);
        auto* p = lp.add_factor<FMC_MRF::P>(L, L);
        for (int x = 0; x < L; ++x) for (int y = 0; y < L; ++y) p->GetFactor()->pw(x, y) = u01(st);
        lp.add_message<FMC_MRF::ML>(u[a], p); lp.add_message<FMC_MRF::MR>(u[b], p);
        lp.AddFactorRelation(u[a], p); lp.AddFactorRelation(p, u[b]);
      };
      for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) { if (c + 1 < W) edge(r * W + c, r * W + c + 1); if (r + 1 < H) edge(r * W + c, (r + 1) * W + c); }
      lp.Begin();
      lp.set_reparametrization(LP_MP::LPReparametrizationMode::Anisotropic);
      (void)lp.LowerBound();
      for (int it = 0; it < 23; ++it) {
        lp.set_reparametrization(LP_MP::LPReparametrizationMode::Anisotropic);   // Solver::PreIterate
        lp.ComputePass(it);                                                       // Solver::Iterate
        hist[k].push_back(lp.LowerBound());                                       // Solver::PostIterate
        if (it % 5 == 4) {                                                        // a rounding iteration of MpRoundingSolver
          lp.set_reparametrization(LP_MP::LPReparametrizationMode::DampedUniform);
          lp.ComputeForwardPassAndPrimal(it); lp.ComputeBackwardPassAndPrimal(it);
          hist[k].push_back(lp.EvaluatePrimal());
        }
      }
      int64_t batches = 0, launched = 0, used = 0, rollbacks = 0;
      lpmp_offload::check(lpmp_speculation_stats(lp.engine(), &batches, &launched, &used, &rollbacks));
      if (k == 1) test(batches > 0 && used == 23 && launched >= used, "passes ran ahead of the loop in batches");
      else test(batches == 0, "no batches without speculation");
      lp.End();
      double host_lb = 0;
      for (auto* f : u) host_lb += f->LowerBound();
      hist[k].push_back(host_lb);                                                 // the factor ops hold the duals the loop asked for, not more
    }
    test(hist[0] == hist[1], "bounds, rounded costs and final duals do not depend on passes running ahead");
    unsetenv("LPMP_ROT_BANDS");
  }
  std::cout << "all tests passed\n";
  return 0;
}
