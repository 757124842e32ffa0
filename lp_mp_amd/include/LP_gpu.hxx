// LP_gpu.hxx — C++ host-side mirror of the reference's plug-in surface for the sweep path.
//
// A user of pawelswoboda/LP_MP declares an FMC struct (FactorList / MessageList of FactorContainer /
// MessageContainer instantiations, reference test/test_model.hxx:130-137), builds the problem with
// add_factor / add_message / AddFactorRelation and lets Solver<LP,VISITOR>::Solve() drive
// LP::ComputePass / LP::LowerBound (reference include/solver.hxx:230-287).  This header keeps exactly
// that surface — same template parameter lists, method names, argument meaning, exceptions — and
// executes the sweep on the MI355X through the C ABI of include/lpmp_engine.h.
//
// What is different from the reference, by necessity: a factor / message OP must be registered as one of the
// device-capable kinds (an lpmp_offload::device_kind / device_message specialisation OUTSIDE the op, see
// lpmp_offload.hxx; the ops below are registered right after their definition); any other op type fails to compile
// (static_assert) instead of silently running on the CPU.  There is no CPU execution path here.
//
// NAMES.  Everything is defined in namespace LP_MP_gpu, so this header can be included next to the reference's own
// headers.  Where the reference is absent (none of its include guards is defined) LP_MP becomes an alias of
// LP_MP_gpu, and code written against the reference's names (tests/cpp/test_model_gpu.cpp is the reference's
// test/test_model.cpp) compiles unchanged.  Solver / StandardVisitor / MpRoundingSolver: LP_gpu_solver.hxx.  To run an
// EXISTING reference LP<FMC> on the device instead, use lpmp_offload::offloaded<LP<FMC>> (lpmp_offload.hxx).
//
// Reference lines mirrored (relative to /root/reference):
//   enums / LpControl            include/config.hxx:39-105
//   FactorContainer              include/factors_messages.hxx:2137-2261 (template list, GetFactor, no_messages...)
//   MessageContainer             include/factors_messages.hxx:571-578, :1530-1545
//   LP<FMC>                      include/LP_MP.h:239-285, :330, :412-460, :462, :698-728, :869-911, :981-1005, :1507-1518
//   Solver / StandardVisitor     include/solver.hxx:230-287, include/visitors/standard_visitor.hxx:28-199
//   factor ops                   test/simplex.cpp, test/potts_factor.cpp, test/test_model.hxx:10-98,
//                                include/factors/labeling_list_factor.hxx:19-218 (labeling / labelings), :220, :346
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/lpmp_engine.h"
#include "lpmp_offload.hxx"

namespace LP_MP_gpu {

using REAL = double;
using INDEX = std::size_t;
using SIGNED_INDEX = long int;
constexpr REAL eps = 1e-8;

enum class Chirality { left, right };
enum class message_passing_schedule { left, right, full, only_send, none };
enum class LPReparametrizationMode { Anisotropic, Anisotropic2, Uniform, DampedUniform, Mixed, Undefined };

constexpr SIGNED_INDEX variableMessageNumber = 0;
constexpr SIGNED_INDEX atMostOneMessage = -1;
constexpr SIGNED_INDEX atMostTwoMessages = -2;
constexpr SIGNED_INDEX atMostThreeMessages = -3;
constexpr SIGNED_INDEX atMostFourMessages = -4;

inline LPReparametrizationMode LPReparametrizationModeConvert(const std::string& s) {
  if (s == "anisotropic") return LPReparametrizationMode::Anisotropic;
  if (s == "anisotropic2") return LPReparametrizationMode::Anisotropic2;
  if (s == "uniform") return LPReparametrizationMode::Uniform;
  if (s == "damped_uniform") return LPReparametrizationMode::DampedUniform;
  if (s == "mixed") return LPReparametrizationMode::Mixed;
  throw std::runtime_error("reparametrization mode " + s + " unknown");
}

class LpControl {
 public:
  LPReparametrizationMode repam = LPReparametrizationMode::Undefined;
  bool computePrimal = false;
  bool computeLowerBound = false;
  bool tighten = false;
  bool end = false;
  bool error = false;
  INDEX tightenConstraints = 0;
  REAL tightenMinDualIncrease = 0.0;
};

namespace meta {
template <class... T> struct list { static constexpr std::size_t size() { return sizeof...(T); } };
}

// ---- device-capable factor ops (the op concept of the reference: LowerBound, serialize_dual) -------------

class UnarySimplexFactor {   // reference test/simplex.cpp:8-12
 public:
  explicit UnarySimplexFactor(const std::vector<REAL>& cost) : cost_(cost) {}
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(cost_); }
  explicit UnarySimplexFactor(INDEX n) : cost_(n, 0.0) {}
  INDEX size() const { return cost_.size(); }
  REAL& operator[](INDEX i) { return cost_[i]; }
  REAL operator[](INDEX i) const { return cost_[i]; }
  REAL LowerBound() const { return *std::min_element(cost_.begin(), cost_.end()); }
  std::vector<REAL>& dual() { return cost_; }
  const std::vector<REAL>& dual() const { return cost_; }
 private:
  std::vector<REAL> cost_;
};

// reference include/factors/constant_factor.hxx:10-30: a factor without variables that carries an offset; its dual is
// the offset itself (serialize_dual), its bound the offset.  On the device: a vector factor with one entry.
class ConstantFactor {
 public:
  explicit ConstantFactor(const REAL offset = 0) : offset_(1, offset) {}
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(offset_); }   // constant_factor.hxx:24
  constexpr static INDEX size() { return 0; }
  REAL LowerBound() const { return offset_[0]; }
  REAL EvaluatePrimal() const { return offset_[0]; }
  void AddToOffset(const REAL delta) { offset_[0] += delta; }
  std::vector<REAL>& dual() { return offset_; }
  const std::vector<REAL>& dual() const { return offset_; }
 private:
  std::vector<REAL> offset_;
};

struct test_factor : UnarySimplexFactor {   // reference test/test_model.hxx:10-16
  test_factor(REAL x, REAL y) : UnarySimplexFactor(std::vector<REAL>{x, y}) {}
};

class PairwiseSimplexFactor {   // reference test/simplex.cpp:52-65; cost(x1,x2) + msg1(x1) + msg2(x2)
 public:
  PairwiseSimplexFactor(INDEX d1, INDEX d2) : d1_(d1), d2_(d2), pairwise_(d1 * d2, 0.0), msg_(d1 + d2, 0.0) {}
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(msg_); }
  INDEX dim1() const { return d1_; }
  INDEX dim2() const { return d2_; }
  REAL& cost(INDEX x1, INDEX x2) { return pairwise_[x1 * d2_ + x2]; }
  REAL cost(INDEX x1, INDEX x2) const { return pairwise_[x1 * d2_ + x2]; }
  REAL& msg1(INDEX x1) { return msg_[x1]; }
  REAL& msg2(INDEX x2) { return msg_[d1_ + x2]; }
  REAL LowerBound() const {
    REAL lb = std::numeric_limits<REAL>::infinity();
    for (INDEX a = 0; a < d1_; ++a) {
      REAL mn = std::numeric_limits<REAL>::infinity();
      for (INDEX b = 0; b < d2_; ++b) mn = std::min(mn, pairwise_[a * d2_ + b] + msg_[d1_ + b]);
      lb = std::min(lb, msg_[a] + mn);
    }
    return lb;
  }
  const std::vector<REAL>& table() const { return pairwise_; }
  std::vector<REAL>& dual() { return msg_; }
  const std::vector<REAL>& dual() const { return msg_; }
 private:
  INDEX d1_, d2_;
  std::vector<REAL> pairwise_, msg_;
};

class pairwise_potts_factor {   // reference test/potts_factor.cpp:34-36
 public:
  pairwise_potts_factor(INDEX dim, REAL diff_cost) : dim_(dim), diff_(diff_cost), msg_(2 * dim, 0.0) {}
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(msg_); }
  INDEX dim() const { return dim_; }
  REAL& diff_cost() { return diff_; }
  REAL diff_cost() const { return diff_; }
  REAL& msg1(INDEX x) { return msg_[x]; }
  REAL& msg2(INDEX x) { return msg_[dim_ + x]; }
  REAL LowerBound() const {
    REAL lb = std::numeric_limits<REAL>::infinity();
    for (INDEX a = 0; a < dim_; ++a) {
      REAL mn = std::numeric_limits<REAL>::infinity();
      for (INDEX b = 0; b < dim_; ++b) mn = std::min(mn, (a == b ? 0.0 : diff_) + msg_[dim_ + b]);
      lb = std::min(lb, msg_[a] + mn);
    }
    return lb;
  }
  std::vector<REAL>& dual() { return msg_; }
  const std::vector<REAL>& dual() const { return msg_; }
 private:
  INDEX dim_;
  REAL diff_;
  std::vector<REAL> msg_;
};

// labeling<...> / labelings<...> (reference include/factors/labeling_list_factor.hxx:19-218)
template <INDEX... LABELS>
struct labeling {
  static constexpr INDEX no_labels() { return sizeof...(LABELS); }
  static constexpr std::array<INDEX, sizeof...(LABELS)> labels() { return {{LABELS...}}; }
};
template <class... LABELINGS>
struct labelings {
  static constexpr INDEX no_labelings() { return sizeof...(LABELINGS); }
  static std::vector<std::vector<INDEX>> as_vectors() {
    std::vector<std::vector<INDEX>> v;
    (v.push_back(std::vector<INDEX>(LABELINGS::labels().begin(), LABELINGS::labels().end())), ...);
    return v;
  }
};

template <class LABELINGS, bool IMPLICIT_ORIGIN>
class labeling_factor : public std::array<REAL, LABELINGS::no_labelings()> {   // reference :220-275
 public:
  using labelings_type = LABELINGS;
  labeling_factor() { this->fill(0.0); }
  template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(*static_cast<std::array<REAL, LABELINGS::no_labelings()>*>(this)); }   // :337-338
  static constexpr bool has_implicit_origin() { return IMPLICIT_ORIGIN; }
  static constexpr INDEX size() { return LABELINGS::no_labelings(); }
  REAL LowerBound() const {
    const REAL mn = *std::min_element(this->begin(), this->end());
    return IMPLICIT_ORIGIN ? std::min(0.0, mn) : mn;
  }
};

// ---- device-capable message ops ----------------------------------------------------------------------
template <Chirality C>
struct UnaryPairwiseMessage {};  // reference test/simplex_marginalization.cpp:19-20; Chirality = side of the pair
struct test_message {};         // reference test/test_model.hxx:66-98
template <class LEFT_LABELINGS, class RIGHT_LABELINGS, INDEX... INDICES>
struct labeling_message {       // reference include/factors/labeling_list_factor.hxx:346-402
  static std::vector<int32_t> match_table() {   // matching_left_labeling for every right labeling
    const auto L = LEFT_LABELINGS::as_vectors();
    const auto R = RIGHT_LABELINGS::as_vectors();
    const std::array<INDEX, sizeof...(INDICES)> idx{{INDICES...}};
    std::vector<int32_t> t;
    for (const auto& r : R) {
      int32_t m = (int32_t)L.size();
      for (std::size_t l = 0; l < L.size() && m == (int32_t)L.size(); ++l) {
        bool ok = true;
        for (std::size_t k = 0; k < idx.size(); ++k) ok = ok && L[l][k] == r[idx[k]];
        if (ok) m = (int32_t)l;
      }
      t.push_back(m);
    }
    return t;
  }
  static int32_t n_left() { return (int32_t)LEFT_LABELINGS::no_labelings(); }
};

}  // namespace LP_MP_gpu

// ---- kind registration of the ops above: non-intrusive, outside the ops (lpmp_offload.hxx) ------------------------------
namespace lpmp_offload {
template <> struct device_kind<LP_MP_gpu::UnarySimplexFactor> : vector_kind<> {};
template <> struct device_kind<LP_MP_gpu::ConstantFactor> : vector_kind<> {};
template <> struct device_kind<LP_MP_gpu::test_factor> : vector_kind<> {};
template <class LABELINGS, bool IMPLICIT_ORIGIN> struct device_kind<LP_MP_gpu::labeling_factor<LABELINGS, IMPLICIT_ORIGIN>> : vector_kind<IMPLICIT_ORIGIN> {};
template <> struct device_kind<LP_MP_gpu::PairwiseSimplexFactor> : pairwise_dense_kind<LP_MP_gpu::PairwiseSimplexFactor> {
  static std::size_t dim1(const LP_MP_gpu::PairwiseSimplexFactor& f) { return f.dim1(); }
  static std::size_t dim2(const LP_MP_gpu::PairwiseSimplexFactor& f) { return f.dim2(); }
  static double table(const LP_MP_gpu::PairwiseSimplexFactor& f, std::size_t a, std::size_t b) { return f.cost(a, b); }
};
template <> struct device_kind<LP_MP_gpu::pairwise_potts_factor> : pairwise_potts_kind<LP_MP_gpu::pairwise_potts_factor> {
  static std::size_t dim(const LP_MP_gpu::pairwise_potts_factor& f) { return f.dim(); }
  static double diff(const LP_MP_gpu::pairwise_potts_factor& f) { return f.diff_cost(); }
};
template <LP_MP_gpu::Chirality C> struct device_message<LP_MP_gpu::UnaryPairwiseMessage<C>> : unary_pairwise_message<C == LP_MP_gpu::Chirality::left ? 0 : 1> {};
template <> struct device_message<LP_MP_gpu::test_message> : min_normalised_message<> {};
template <class LL, class RL, LP_MP_gpu::INDEX... I> struct device_message<LP_MP_gpu::labeling_message<LL, RL, I...>> : labeling_list_message<> {
  static std::vector<int32_t> match_table() { return LP_MP_gpu::labeling_message<LL, RL, I...>::match_table(); }
  static int32_t n_left() { return LP_MP_gpu::labeling_message<LL, RL, I...>::n_left(); }
};
}  // namespace lpmp_offload

namespace LP_MP_gpu {

// ---- containers ----------------------------------------------------------------------------------------
class FactorTypeAdapter {   // reference include/LP_MP.h:46-145 (the part the sweep path needs)
 public:
  virtual ~FactorTypeAdapter() {}
  virtual REAL LowerBound() const = 0;
  virtual INDEX no_messages() const { return no_messages_; }
  virtual INDEX no_send_messages() const { return no_send_messages_; }
  virtual INDEX dual_size() const = 0;
  INDEX index_ = 0, no_messages_ = 0, no_send_messages_ = 0;
};

template <class FACTOR_TYPE, class FACTOR_MESSAGE_TRAIT, INDEX FACTOR_NO, bool COMPUTE_PRIMAL_SOLUTION = false>
class FactorContainer : public FactorTypeAdapter {
  static_assert(lpmp_offload::is_registered_kind<FACTOR_TYPE>::value,
                "this factor op has no lpmp_offload::device_kind registration: only device-capable factor ops can be plugged into LP_gpu (no CPU fallback)");
 public:
  using FactorType = FACTOR_TYPE;
  using FMC = FACTOR_MESSAGE_TRAIT;
  static constexpr INDEX factor_no = FACTOR_NO;
  static constexpr bool compute_primal = COMPUTE_PRIMAL_SOLUTION;
  template <class... ARGS> explicit FactorContainer(ARGS... args) : factor_(args...) {}
  FactorType* GetFactor() { return &factor_; }
  const FactorType* GetFactor() const { return &factor_; }
  REAL LowerBound() const final { return factor_.LowerBound(); }
  INDEX dual_size() const final { return lpmp_offload::serialized_dual_size(const_cast<FACTOR_TYPE&>(factor_)); }   // factors_messages.hxx:3214-3217
  static constexpr bool CanComputePrimal() { return COMPUTE_PRIMAL_SOLUTION; }
 private:
  FactorType factor_;
};

template <class MESSAGE_TYPE, INDEX LEFT_FACTOR_NO, INDEX RIGHT_FACTOR_NO, message_passing_schedule MPS,
          SIGNED_INDEX NO_OF_LEFT_FACTORS, SIGNED_INDEX NO_OF_RIGHT_FACTORS, class FACTOR_MESSAGE_TRAIT, INDEX MESSAGE_NO>
class MessageContainer {
 public:
  using MessageType = MESSAGE_TYPE;
  static constexpr INDEX leftFactorNumber = LEFT_FACTOR_NO;
  static constexpr INDEX rightFactorNumber = RIGHT_FACTOR_NO;
  static constexpr message_passing_schedule schedule = MPS;
  static constexpr SIGNED_INDEX no_left_factors() { return NO_OF_LEFT_FACTORS; }
  static constexpr SIGNED_INDEX no_right_factors() { return NO_OF_RIGHT_FACTORS; }
  static constexpr INDEX message_no = MESSAGE_NO;
  // reference factors_messages.hxx:1530-1545
  static constexpr bool sends_message_to_left_constexpr() { return MPS == message_passing_schedule::right || MPS == message_passing_schedule::full || MPS == message_passing_schedule::only_send; }
  static constexpr bool sends_message_to_right_constexpr() { return MPS == message_passing_schedule::left || MPS == message_passing_schedule::full || MPS == message_passing_schedule::only_send; }
  static constexpr bool receives_message_from_left_constexpr() { return MPS == message_passing_schedule::right || MPS == message_passing_schedule::full; }
  static constexpr bool receives_message_from_right_constexpr() { return MPS == message_passing_schedule::left || MPS == message_passing_schedule::full; }
  MessageContainer(FactorTypeAdapter* l, FactorTypeAdapter* r) : left_(l), right_(r) {}
  FactorTypeAdapter* GetLeftFactor() const { return left_; }
  FactorTypeAdapter* GetRightFactor() const { return right_; }
 private:
  FactorTypeAdapter *left_, *right_;
};

// ---- LP ------------------------------------------------------------------------------------------------
template <class FMC_T>
class LP_gpu {
 public:
  using FMC = FMC_T;
  using weight_slice = std::pair<const REAL*, const REAL*>;
  struct csr_real { std::vector<int64_t> off; std::vector<REAL> data; INDEX size() const { return off.size() - 1; } };
  struct csr_mask { std::vector<int64_t> off; std::vector<unsigned char> data; INDEX size() const { return off.size() - 1; } };
  struct omega_storage { csr_real forward, backward; csr_mask receive_mask_forward, receive_mask_backward; };

  explicit LP_gpu(int device = 0) : device_(device) {}
  template <class CMD> explicit LP_gpu(CMD&) : device_(0) {}   // LP(TCLAP::CmdLine&) call sites (solver.hxx:57)
  ~LP_gpu() { if (engine_) lpmp_destroy(engine_); }
  // how many passes the engine may run ahead of the Solve loop (0: every call as it comes; default 16)
  void set_speculation(int max_passes_ahead) { speculation_ = max_passes_ahead; if (engine_) check(lpmp_set_speculation(engine_, speculation_)); }
  LP_gpu(const LP_gpu&) = delete;
  LP_gpu& operator=(const LP_gpu&) = delete;

  template <class FACTOR_CONTAINER_TYPE, class... ARGS>
  FACTOR_CONTAINER_TYPE* add_factor(ARGS... args) {           // reference LP_MP.h:239-253
    static_assert(std::is_same_v<typename FACTOR_CONTAINER_TYPE::FMC, FMC>, "factor container of another FMC");
    pull_duals();
    auto* f = new FACTOR_CONTAINER_TYPE(args...);
    f->index_ = f_.size();
    f_.emplace_back(f);
    f_type_.push_back((int32_t)FACTOR_CONTAINER_TYPE::factor_no);
    flatteners_.push_back(&LP_gpu::template flatten_factor<FACTOR_CONTAINER_TYPE>);
    unflatteners_.push_back(&LP_gpu::template unflatten_factor<FACTOR_CONTAINER_TYPE>);
    set_flags_dirty();
    return f;
  }

  template <class MESSAGE_CONTAINER_TYPE, class LEFT_FACTOR, class RIGHT_FACTOR>
  MESSAGE_CONTAINER_TYPE* add_message(LEFT_FACTOR* l, RIGHT_FACTOR* r) {   // reference LP_MP.h:267-285
    static_assert(LEFT_FACTOR::factor_no == MESSAGE_CONTAINER_TYPE::leftFactorNumber, "left factor type mismatch");
    static_assert(RIGHT_FACTOR::factor_no == MESSAGE_CONTAINER_TYPE::rightFactorNumber, "right factor type mismatch");
    pull_duals();
    auto m = std::make_shared<MESSAGE_CONTAINER_TYPE>(l, r);
    msg_keep_.push_back(m);
    m_type_.push_back((int32_t)MESSAGE_CONTAINER_TYPE::message_no);
    m_left_.push_back((int32_t)l->index_);
    m_right_.push_back((int32_t)r->index_);
    l->no_messages_++; r->no_messages_++;
    if (MESSAGE_CONTAINER_TYPE::sends_message_to_right_constexpr()) l->no_send_messages_++;
    if (MESSAGE_CONTAINER_TYPE::sends_message_to_left_constexpr()) r->no_send_messages_++;
    set_flags_dirty();
    return m.get();
  }

  INDEX GetNumberOfFactors() const { return f_.size(); }
  FactorTypeAdapter* GetFactor(const INDEX i) const { return f_[i].get(); }
  INDEX GetNumberOfMessages() const { return m_type_.size(); }

  void AddFactorRelation(FactorTypeAdapter* f1, FactorTypeAdapter* f2) { ForwardPassFactorRelation(f1, f2); BackwardPassFactorRelation(f2, f1); }
  void ForwardPassFactorRelation(FactorTypeAdapter* f1, FactorTypeAdapter* f2) { rel_fwd_.push_back((int32_t)f1->index_); rel_fwd_.push_back((int32_t)f2->index_); set_flags_dirty(); }
  void BackwardPassFactorRelation(FactorTypeAdapter* f1, FactorTypeAdapter* f2) { rel_bwd_.push_back((int32_t)f1->index_); rel_bwd_.push_back((int32_t)f2->index_); set_flags_dirty(); }

  // Not in the reference: replace this LP's factor relations by the order the engine suggests for it (lpmp_plan_suggest_order: the
  // updated factors colour by colour, one dependent level — one launch step — per colour) as a chain of relations through all
  // factors, exactly what INTEGRATION.md 2a shows a caller of the reference's LP doing with AddFactorRelation.  Same factors,
  // messages and costs; another, equally valid sweep order.  Returns the number of colours.
  int apply_suggested_order(uint64_t seed = 0) {
    ready();
    pull_duals();                                          // (duals the device holds go back into the factor ops before the re-upload)
    std::vector<int32_t> rank(f_.size()), by_rank(f_.size());
    int32_t colours = 0;
    check(lpmp_plan_suggest_order(lpmp_engine_plan_mut(engine_), seed, rank.data(), &colours));
    for (std::size_t f = 0; f < rank.size(); ++f) by_rank[(std::size_t)rank[f]] = (int32_t)f;
    rel_fwd_.clear(); rel_bwd_.clear();
    for (std::size_t i = 0; i + 1 < by_rank.size(); ++i) {
      rel_fwd_.push_back(by_rank[i]); rel_fwd_.push_back(by_rank[i + 1]);
      rel_bwd_.push_back(by_rank[i + 1]); rel_bwd_.push_back(by_rank[i]);
    }
    set_flags_dirty();
    return (int)colours;
  }

  void put_in_same_partition(FactorTypeAdapter* f1, FactorTypeAdapter* f2) { part_.push_back((int32_t)f1->index_); part_.push_back((int32_t)f2->index_); set_flags_dirty(); }   // LP_MP.h:465
  void set_inner_iterations(const INDEX n) { inner_ = (int)n; }   // --innerIteration, LP_MP.h:590
  void Begin() { repamMode_ = LPReparametrizationMode::Undefined; }   // reference LP_MP.h:705-708
  // --reparametrizationType {shared|residual|partition|overlapping_partition|adaptive} (reference LP_MP.h:589-593,
  // :710-722): all five run on the device (include/lpmp_engine.h)
  void set_reparametrization_type(const std::string& t) {
    if (t == "shared") rtype_ = LPMP_RTYPE_SHARED;
    else if (t == "residual") rtype_ = LPMP_RTYPE_RESIDUAL;
    else if (t == "partition") rtype_ = LPMP_RTYPE_PARTITION;
    else if (t == "overlapping_partition") rtype_ = LPMP_RTYPE_OVERLAPPING_PARTITION;
    else if (t == "adaptive") rtype_ = LPMP_RTYPE_ADAPTIVE;
    else throw std::runtime_error("reparametrization type " + t + " unknown");
  }
  void End() { pull_duals(); }
  void set_reparametrization(const LPReparametrizationMode r) { repamMode_ = r; }
  LPReparametrizationMode GetRepamMode() const { return repamMode_; }
  void add_to_constant(const REAL x) { constant_ += x; set_flags_dirty(); }
  void set_flags_dirty() { dirty_ = true; }

  void ComputePass(const INDEX /*iteration*/) { ready_mode(); check(lpmp_compute_pass(engine_, 1)); duals_on_device_ = true; }
  // n consecutive passes in one call (no reference counterpart: a caller that needs nothing between passes lets the
  // engine join them — one persistent launch in Infinity-Cache order on HBM-sized models); same results
  void ComputePasses(const INDEX n) { if (n == 0) return; ready_mode(); check(lpmp_compute_pass(engine_, (int)n)); duals_on_device_ = true; }
  void ComputeForwardPass() { ready_mode(); check(lpmp_compute_forward_pass(engine_)); duals_on_device_ = true; }
  void ComputeBackwardPass() { ready_mode(); check(lpmp_compute_backward_pass(engine_)); duals_on_device_ = true; }

  // primal rounding inside the sweep (reference LP_MP.h:914-940, 1067-1082, 1521-1536): needs FactorContainers declared
  // with COMPUTE_PRIMAL_SOLUTION = true, like LP_MP-MRF's FMC_SRMP::UnaryFactor
  void ComputeForwardPassAndPrimal(const INDEX iteration) { ready_mode(); check(lpmp_compute_forward_pass_and_primal(engine_, iteration)); duals_on_device_ = true; }
  void ComputeBackwardPassAndPrimal(const INDEX iteration) { ready_mode(); check(lpmp_compute_backward_pass_and_primal(engine_, iteration)); duals_on_device_ = true; }
  void ComputePassAndPrimal(const INDEX iteration) { ComputeForwardPassAndPrimal(iteration); ComputeBackwardPassAndPrimal(iteration); }
  bool CheckPrimalConsistency() { ready(); int ok = 0; check(lpmp_check_primal_consistency(engine_, &ok)); return ok != 0; }
  REAL EvaluatePrimal() { ready(); REAL c = 0; check(lpmp_evaluate_primal(engine_, &c)); return c; }
  // the factors' primal_ members in factor order: vector factor (label, 0), pairwise factor (x0, x1); unset = dimension
  std::vector<std::array<int32_t, 2>> primal() {
    ready();
    std::vector<std::array<int32_t, 2>> out(f_.size());
    check(lpmp_download_primal(engine_, out.empty() ? nullptr : out[0].data()));
    return out;
  }

  // LP::ComputePass(factorIt, factorItEnd, omegaIt, receive_it), reference LP_MP.h:981-1005.
  // omegaIt / receive_it iterate over ranges (anything with begin()/end()), one per listed factor.
  template <class FACTOR_ITERATOR, class OMEGA_ITERATOR, class RECEIVE_MASK_ITERATOR>
  void ComputePass(FACTOR_ITERATOR factorIt, const FACTOR_ITERATOR factorItEnd, OMEGA_ITERATOR omegaIt, RECEIVE_MASK_ITERATOR receive_it) {
    ready();
    std::vector<int32_t> f;
    std::vector<int64_t> oo{0}, mo{0};
    std::vector<REAL> om;
    std::vector<uint8_t> mk;
    for (; factorIt != factorItEnd; ++factorIt, ++omegaIt, ++receive_it) {
      f.push_back((int32_t)(*factorIt)->index_);
      for (auto x : *omegaIt) om.push_back(x);
      for (auto x : *receive_it) mk.push_back((uint8_t)x);
      oo.push_back((int64_t)om.size()); mo.push_back((int64_t)mk.size());
    }
    check(lpmp_compute_pass_custom(engine_, (int64_t)f.size(), f.data(), oo.data(), om.data(), mo.data(), mk.data()));
    duals_on_device_ = true;
  }

  double LowerBound() {   // reference LP_MP.h:1507-1518
    ready();
    double lb = 0;
    check(lpmp_lower_bound(engine_, &lb));
    return lb;
  }

  omega_storage get_omega() {   // reference LP_MP.h:412-460
    ready();
    const int mode = mode_index();
    lpmp_plan* p = lpmp_engine_plan_mut(engine_);
    omega_storage s;
    csr_real* o[2] = {&s.forward, &s.backward};
    csr_mask* k[2] = {&s.receive_mask_forward, &s.receive_mask_backward};
    for (int d = 0; d < 2; ++d) {
      const int64_t n = lpmp_plan_n_updated(p, d);
      o[d]->off.resize(n + 1); o[d]->data.resize(lpmp_plan_omega_nnz(p, d));
      k[d]->off.resize(n + 1); k[d]->data.resize(lpmp_plan_mask_nnz(p, d));
      check(lpmp_plan_get_omega(p, d, mode, o[d]->off.data(), o[d]->data.data()));
      check(lpmp_plan_get_mask(p, d, mode, k[d]->off.data(), k[d]->data.data()));
    }
    return s;
  }

  std::vector<FactorTypeAdapter*> forward_update_ordering() { return update_ordering(0); }
  std::vector<FactorTypeAdapter*> backward_update_ordering() { return update_ordering(1); }

  // copy the device duals back into the factor ops (what the reference's factors hold after a pass)
  void pull_duals() {
    if (!engine_ || !duals_on_device_ || dirty_) return;
    std::vector<REAL> d((size_t)lpmp_dual_size(engine_));
    check(lpmp_download_duals(engine_, d.data()));
    const REAL* p = d.data();
    for (INDEX i = 0; i < f_.size(); ++i) p = (this->*unflatteners_[i])(f_[i].get(), p);
    duals_on_device_ = false;
  }

 private:
  template <class L> struct mtype_table;
  template <class... MC> struct mtype_table<meta::list<MC...>> {
    static void fill(std::vector<lpmp_msg_type>& t, std::vector<int64_t>& tab_off, std::vector<int32_t>& tab_data, std::vector<int32_t>& tab_nleft) {
      (add<MC>(t, tab_off, tab_data, tab_nleft), ...);
    }
    template <class C>
    static void add(std::vector<lpmp_msg_type>& t, std::vector<int64_t>& tab_off, std::vector<int32_t>& tab_data, std::vector<int32_t>& tab_nleft) {
      using Op = typename C::MessageType;
      static_assert(lpmp_offload::is_registered_message<Op>::value, "message op without an lpmp_offload::device_message registration");
      using R = lpmp_offload::device_message<Op>;
      lpmp_msg_type m{};
      m.left_ftype = (int32_t)C::leftFactorNumber; m.right_ftype = (int32_t)C::rightFactorNumber;
      m.schedule = (int32_t)C::schedule; m.n_left = (int32_t)C::no_left_factors(); m.n_right = (int32_t)C::no_right_factors();
      m.kind = R::kind; m.flags = R::flags;
      if constexpr (R::kind == LPMP_M_LABELING) {
        m.param = (int32_t)tab_nleft.size();
        const auto tab = R::match_table();
        tab_data.insert(tab_data.end(), tab.begin(), tab.end());
        tab_off.push_back((int64_t)tab_data.size());
        tab_nleft.push_back(R::n_left());
      } else m.param = R::param;
      if (t.size() != C::message_no) throw std::runtime_error("MessageList: message numbers must be consecutive");
      t.push_back(m);
    }
  };
  template <class L> struct ftype_table;
  template <class... FC> struct ftype_table<meta::list<FC...>> {
    static std::vector<uint8_t> primal() { return {(uint8_t)FC::compute_primal...}; }
  };

  struct Flat {
    std::vector<uint8_t> kind, flags;
    std::vector<int32_t> d0, d1;
    std::vector<REAL> cdata, dual;
  };
  template <class FC>
  void flatten_factor(FactorTypeAdapter* fa, Flat& fl) {   // what serialize_dual lists -> packed duals (factors_messages.hxx:3196-3223)
    auto& op = *static_cast<FC*>(fa)->GetFactor();
    using K = lpmp_offload::device_kind<typename FC::FactorType>;
    int32_t d0 = 0, d1 = 0;
    K::dims(op, d0, d1);
    fl.kind.push_back((uint8_t)K::kind); fl.flags.push_back((uint8_t)K::flags); fl.d0.push_back(d0); fl.d1.push_back(d1);
    const std::size_t nc = K::const_size(op), nd = (std::size_t)lpmp_factor_dual_size(K::kind, d0, d1);
    fl.cdata.resize(fl.cdata.size() + nc);
    if (nc) K::export_const(op, fl.cdata.data() + fl.cdata.size() - nc);
    fl.dual.resize(fl.dual.size() + nd);
    K::export_dual(op, fl.dual.data() + fl.dual.size() - nd);
  }
  template <class FC>
  const REAL* unflatten_factor(FactorTypeAdapter* fa, const REAL* p) {
    auto& op = *static_cast<FC*>(fa)->GetFactor();
    using K = lpmp_offload::device_kind<typename FC::FactorType>;
    int32_t d0 = 0, d1 = 0;
    K::dims(op, d0, d1);
    K::import_dual(op, p);
    return p + lpmp_factor_dual_size(K::kind, d0, d1);
  }

  static void check(int rc) { if (rc != LPMP_OK) throw std::runtime_error(lpmp_last_error()); }

  int mode_index() const {
    if (repamMode_ == LPReparametrizationMode::Undefined) throw std::runtime_error("no reparametrization mode set");   // LP_MP.h:458
    return (int)repamMode_;
  }

  void ready() {
    if (f_.size() <= 1) throw std::runtime_error("LP needs more than one factor");   // reference assert LP_MP.h:708
    if (!engine_) { check(lpmp_create(device_, &engine_)); check(lpmp_set_speculation(engine_, speculation_)); }   // passes may run ahead of the Solve loop (include/lpmp_engine.h)
    if (!dirty_) { check(lpmp_set_inner_iterations(engine_, inner_)); check(lpmp_set_reparametrization_type(engine_, rtype_)); return; }
    Flat fl;
    for (INDEX i = 0; i < f_.size(); ++i) (this->*flatteners_[i])(f_[i].get(), fl);
    std::vector<lpmp_msg_type> mt;
    std::vector<int64_t> tab_off{0};
    std::vector<int32_t> tab_data, tab_nleft;
    mtype_table<typename FMC::MessageList>::fill(mt, tab_off, tab_data, tab_nleft);
    const std::vector<uint8_t> primal = ftype_table<typename FMC::FactorList>::primal();
    lpmp_model m{};
    m.n_ftypes = (int32_t)FMC::FactorList::size(); m.ftype_computes_primal = primal.data();
    m.n_mtypes = (int32_t)mt.size(); m.mtypes = mt.data();
    m.n_tables = (int32_t)tab_nleft.size(); m.tab_off = tab_off.data(); m.tab_data = tab_data.data(); m.tab_nleft = tab_nleft.data();
    m.n_factors = (int64_t)f_.size(); m.f_type = f_type_.data(); m.f_kind = fl.kind.data(); m.f_flags = fl.flags.data();
    m.f_dim0 = fl.d0.data(); m.f_dim1 = fl.d1.data(); m.const_data = fl.cdata.data(); m.dual_data = fl.dual.data();
    m.n_messages = (int64_t)m_type_.size(); m.m_type = m_type_.data(); m.m_left = m_left_.data(); m.m_right = m_right_.data();
    m.n_rel_fwd = (int64_t)rel_fwd_.size() / 2; m.rel_fwd = rel_fwd_.data();
    m.n_rel_bwd = (int64_t)rel_bwd_.size() / 2; m.rel_bwd = rel_bwd_.data();
    m.constant = constant_;
    m.n_part_pairs = (int64_t)part_.size() / 2; m.part_pairs = part_.data();
    static const double zero = 0;
    if (!m.const_data) m.const_data = &zero;
    check(lpmp_upload_model(engine_, &m, LPMP_MEM_HOST, LPMP_MEM_HOST));
    check(lpmp_set_inner_iterations(engine_, inner_));
    check(lpmp_set_reparametrization_type(engine_, rtype_));
    dirty_ = false;
    duals_on_device_ = false;
  }
  void ready_mode() { ready(); check(lpmp_set_reparametrization(engine_, mode_index())); }

  std::vector<FactorTypeAdapter*> update_ordering(int d) {
    ready();
    const lpmp_plan* p = lpmp_engine_plan(engine_);
    std::vector<int32_t> idx((size_t)lpmp_plan_n_updated(p, d));
    check(lpmp_plan_get_update_order(p, d, idx.data()));
    std::vector<FactorTypeAdapter*> out;
    for (int32_t i : idx) out.push_back(f_[i].get());
    return out;
  }

  int device_;
  int speculation_ = 16;
  lpmp_engine* engine_ = nullptr;
  bool dirty_ = true, duals_on_device_ = false;
  int rtype_ = LPMP_RTYPE_SHARED, inner_ = 5;
  std::vector<int32_t> part_;
  LPReparametrizationMode repamMode_ = LPReparametrizationMode::Undefined;
  REAL constant_ = 0;
  std::vector<std::unique_ptr<FactorTypeAdapter>> f_;
  std::vector<std::shared_ptr<void>> msg_keep_;
  std::vector<int32_t> f_type_, m_type_, m_left_, m_right_, rel_fwd_, rel_bwd_;
  using flatten_fn = void (LP_gpu::*)(FactorTypeAdapter*, Flat&);
  using unflatten_fn = const REAL* (LP_gpu::*)(FactorTypeAdapter*, const REAL*);
  std::vector<flatten_fn> flatteners_;
  std::vector<unflatten_fn> unflatteners_;
};

template <class FMC> using LP = LP_gpu<FMC>;   // drop-in name

}  // namespace LP_MP_gpu

#if !defined(LP_MP_MAIN) && !defined(LP_MP_CONFIG_HXX) && !defined(LP_MP_FACTORS_MESSAGES_HXX) && !defined(LP_MP_SOLVER_HXX) && \
    !defined(LPMP_NO_LP_MP_ALIAS) && !defined(LPMP_LP_MP_ALIAS_DEFINED)
#define LPMP_LP_MP_ALIAS_DEFINED
namespace LP_MP = LP_MP_gpu;   // standalone use: the reference's names
#endif
