// lpmp_offload.hxx — run the dual block-coordinate-ascent sweep of an EXISTING pawelswoboda/LP_MP problem on the MI355X.
//
// This header defines NO name the reference defines (everything lives in namespace lpmp_offload), so it can be included
// next to the reference's own LP_MP.h / solver.hxx / standard_visitor.hxx.  What it does:
//
//   1. kind registration, non-intrusive.  A factor op or message op of the reference is told to the engine by
//      specialising a trait OUTSIDE the op:
//          template <> struct lpmp_offload::device_kind<LP_MP::UnarySimplexFactor> : lpmp_offload::vector_kind<> {};
//          template <> struct lpmp_offload::device_kind<LP_MP::PairwiseSimplexFactor>
//              : lpmp_offload::pairwise_dense_kind<LP_MP::PairwiseSimplexFactor> { ...dim1 / dim2 / table(a,b)... };
//          template <> struct lpmp_offload::device_message<LP_MP::UnaryPairwiseMessage<LP_MP::Chirality::left>>
//              : lpmp_offload::unary_pairwise_message<0> {};
//      An op without a registration is a compile error with a readable message: nothing ever falls back to the CPU.
//
//   2. flatten_through_serialize_dual: walks a reference-shaped LP — the per-type container vectors `factors_` /
//      `messages_` (reference include/LP_MP.h:480-496), the insertion-order list `f_` (:476), the ordering relations
//      (:519), `constant_`, `partition_graph` (:564) — and packs what every factor's serialize_dual enumerates
//      (reference include/factors_messages.hxx:3196-3223) into the flat arrays of include/lpmp_model.h, with visitors
//      that mirror allocate_archive / save_archive / load_archive (reference include/serialization.hxx:22-95, :228-327,
//      :330-424) for doubles.
//
//   3. offloaded<LP_BASE>: a subclass of the reference's LP<FMC> (the precedent for substituting the LP type is
//      tree_decomposition.hxx:714, :918-929, which subclasses and re-declares ComputePass) whose ComputePass /
//      ComputeForwardPass / ComputeBackwardPass / LowerBound / ...AndPrimal / EvaluatePrimal run on the device.  It is
//      handed to the reference's own Solver<LP_TYPE, VISITOR> unchanged:
//          using LP_device = lpmp_offload::offloaded<LP_MP::LP<FMC>>;
//          LP_MP::MpRoundingSolver<LP_MP::Solver<LP_device, LP_MP::StandardVisitor>> solver(argc, argv);
//      Duals live on the device between calls; End() (which Solver::Solve calls, solver.hxx:247) and every structural
//      change write them back into the factor ops through serialize_dual.
//
// Needs C++17 and include/lpmp_engine.h; links against liblpmp_engine.so.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/lpmp_engine.h"

namespace lpmp_offload {

// ---------------------------------------------------------------------------------------------------------------------
// serialize_dual visitors.  A factor op's `template <class ARCHIVE> void serialize_dual(ARCHIVE& ar) { ar(a, b, ...); }`
// is called with one of these instead of the reference's byte archives; members may be arithmetic values, std::array,
// std::vector, anything with begin() + size() (the reference's vector<T>), anything with dim1() / dim2() / operator()(i, j)
// (the reference's matrix<T>: row-major, unpadded, serialization.hxx:277-294), or a {pointer, no_elements} pair
// (binary_data<T>, serialization.hxx:14-20).
namespace detail {
template <class T, class = void> struct is_matrix_like : std::false_type {};
// the element access must yield an lvalue: the reference's vector<T> inherits a by-value `T operator()(i1, i2) const` and
// dim1() / dim2() from its expression-template bases (vector.hxx:20-24, 45-47, 69-72) and is NOT a matrix — it goes the
// range way (found by tools/check_offload_against_reference.sh against the real headers)
template <class T> struct is_matrix_like<T, std::void_t<decltype(std::declval<const T&>().dim1()), decltype(std::declval<const T&>().dim2()),
                                                         decltype(std::declval<T&>()(std::size_t(0), std::size_t(0)))>>
    : std::is_lvalue_reference<decltype(std::declval<T&>()(std::size_t(0), std::size_t(0)))> {};
template <class T, class = void> struct is_range_like : std::false_type {};
template <class T> struct is_range_like<T, std::void_t<decltype(std::declval<T&>().begin()), decltype(std::declval<const T&>().size())>> : std::true_type {};
template <class T, class = void> struct is_binary_data : std::false_type {};
template <class T> struct is_binary_data<T, std::void_t<decltype(std::declval<const T&>().pointer), decltype(std::declval<const T&>().no_elements)>> : std::true_type {};
template <class T> struct dependent_false : std::false_type {};
}  // namespace detail

template <class Derived>
struct dual_visitor {   // the variadic call operator of the reference's archives (serialization.hxx:76-85)
  template <class... T> void operator()(T&&... members) { (static_cast<Derived*>(this)->member(members), ...); }
};
struct dual_counter : dual_visitor<dual_counter> {   // allocate_archive, in doubles
  std::size_t count = 0;
  template <class T> void member(T& m) {
    using U = std::remove_cv_t<std::remove_reference_t<T>>;
    if constexpr (std::is_arithmetic_v<U>) count += 1;
    else if constexpr (detail::is_matrix_like<U>::value) count += (std::size_t)m.dim1() * (std::size_t)m.dim2();
    else if constexpr (detail::is_binary_data<U>::value) count += (std::size_t)m.no_elements;
    else if constexpr (detail::is_range_like<U>::value) count += (std::size_t)m.size();
    else static_assert(detail::dependent_false<U>::value, "serialize_dual member of a type the offload visitors do not know");
  }
};
struct dual_saver : dual_visitor<dual_saver> {       // save_archive: factor -> packed doubles
  double* out;
  explicit dual_saver(double* p) : out(p) {}
  template <class T> void member(T& m) {
    using U = std::remove_cv_t<std::remove_reference_t<T>>;
    if constexpr (std::is_arithmetic_v<U>) *out++ = (double)m;
    else if constexpr (detail::is_matrix_like<U>::value) { for (std::size_t i = 0; i < (std::size_t)m.dim1(); ++i) for (std::size_t j = 0; j < (std::size_t)m.dim2(); ++j) *out++ = (double)m(i, j); }
    else if constexpr (detail::is_binary_data<U>::value) { for (std::size_t i = 0; i < (std::size_t)m.no_elements; ++i) *out++ = (double)m.pointer[i]; }
    else if constexpr (detail::is_range_like<U>::value) { for (auto it = m.begin(); it != m.end(); ++it) *out++ = (double)*it; }
    else static_assert(detail::dependent_false<U>::value, "serialize_dual member of a type the offload visitors do not know");
  }
};
struct dual_loader : dual_visitor<dual_loader> {     // load_archive: packed doubles -> factor
  const double* in;
  explicit dual_loader(const double* p) : in(p) {}
  template <class T> void member(T& m) {
    using U = std::remove_cv_t<std::remove_reference_t<T>>;
    if constexpr (std::is_arithmetic_v<U>) m = (U)*in++;
    else if constexpr (detail::is_matrix_like<U>::value) { for (std::size_t i = 0; i < (std::size_t)m.dim1(); ++i) for (std::size_t j = 0; j < (std::size_t)m.dim2(); ++j) m(i, j) = *in++; }
    else if constexpr (detail::is_binary_data<U>::value) { for (std::size_t i = 0; i < (std::size_t)m.no_elements; ++i) m.pointer[i] = *in++; }
    else if constexpr (detail::is_range_like<U>::value) { for (auto it = m.begin(); it != m.end(); ++it) *it = *in++; }
    else static_assert(detail::dependent_false<U>::value, "serialize_dual member of a type the offload visitors do not know");
  }
};
template <class Op> std::size_t serialized_dual_size(Op& op) { dual_counter c; op.serialize_dual(c); return c.count; }

// ---------------------------------------------------------------------------------------------------------------------
// kind registration
template <class Op, class = void> struct device_kind;       // specialise for every factor op that goes to the device
template <class MsgOp, class = void> struct device_message; // specialise for every message op
template <class T, class = void> struct is_registered_kind : std::false_type {};
template <class T> struct is_registered_kind<T, std::void_t<decltype(device_kind<T>::kind)>> : std::true_type {};
template <class T, class = void> struct is_registered_message : std::false_type {};
template <class T> struct is_registered_message<T, std::void_t<decltype(device_message<T>::kind)>> : std::true_type {};

// vector factors (LPMP_F_VECTOR): the dual is everything serialize_dual lists, LowerBound its minimum (clamped at 0
// with an implicit origin: labeling_factor<..., true>, include/factors/labeling_list_factor.hxx:241-275)
template <bool IMPLICIT_ORIGIN = false>
struct vector_kind {
  static constexpr int kind = LPMP_F_VECTOR;
  static constexpr int flags = IMPLICIT_ORIGIN ? LPMP_FF_IMPLICIT_ORIGIN : 0;
  template <class Op> static void dims(Op& op, int32_t& d0, int32_t& d1) { d0 = (int32_t)serialized_dual_size(op); d1 = 0; }
  template <class Op> static std::size_t const_size(Op&) { return 0; }
  template <class Op> static void export_const(Op&, double*) {}
  template <class Op> static void export_dual(Op& op, double* out) { dual_saver s(out); op.serialize_dual(s); }
  template <class Op> static void import_dual(Op& op, const double* in) { dual_loader l(in); op.serialize_dual(l); }
};
// dense pairwise factors (LPMP_F_PAIRWISE_DENSE): cost(a, b) = table(a, b) + m1[a] + m2[b] with a constant table and
// the two message vectors as the dual.  Derive a registration from this and give it
//     static std::size_t dim1(const Op&), dim2(const Op&);   static double table(const Op&, std::size_t a, std::size_t b);
// By default the dual is what serialize_dual lists and must be exactly m1[dim1] | m2[dim2]; a factor that also lists its
// table there (or keeps the vectors elsewhere) overrides export_dual / import_dual.
template <class Op, class Registration = void>
struct pairwise_dense_kind {
  static constexpr int kind = LPMP_F_PAIRWISE_DENSE;
  static constexpr int flags = 0;
  using R = std::conditional_t<std::is_void_v<Registration>, device_kind<Op>, Registration>;
  static void dims(Op& op, int32_t& d0, int32_t& d1) { d0 = (int32_t)R::dim1(op); d1 = (int32_t)R::dim2(op); }
  static std::size_t const_size(Op& op) { return (std::size_t)R::dim1(op) * (std::size_t)R::dim2(op); }
  static void export_const(Op& op, double* out) {
    const std::size_t a = R::dim1(op), b = R::dim2(op);
    for (std::size_t i = 0; i < a; ++i) for (std::size_t j = 0; j < b; ++j) *out++ = R::table(op, i, j);
  }
  static void export_dual(Op& op, double* out) {
    if (serialized_dual_size(op) != (std::size_t)R::dim1(op) + (std::size_t)R::dim2(op))
      throw std::runtime_error("pairwise factor: serialize_dual must list the two message vectors (or the registration overrides export_dual)");
    dual_saver s(out); op.serialize_dual(s);
  }
  static void import_dual(Op& op, const double* in) { dual_loader l(in); op.serialize_dual(l); }
};
// Potts pairwise factors (LPMP_F_PAIRWISE_POTTS): cost(a, b) = diff * [a != b] + m1[a] + m2[b].  The registration gives
//     static std::size_t dim(const Op&);   static double diff(const Op&);
template <class Op, class Registration = void>
struct pairwise_potts_kind {
  static constexpr int kind = LPMP_F_PAIRWISE_POTTS;
  static constexpr int flags = 0;
  using R = std::conditional_t<std::is_void_v<Registration>, device_kind<Op>, Registration>;
  static void dims(Op& op, int32_t& d0, int32_t& d1) { d0 = d1 = (int32_t)R::dim(op); }
  static std::size_t const_size(Op&) { return 1; }
  static void export_const(Op& op, double* out) { *out = R::diff(op); }
  static void export_dual(Op& op, double* out) {
    if (serialized_dual_size(op) != 2 * (std::size_t)R::dim(op)) throw std::runtime_error("Potts factor: serialize_dual must list the two message vectors");
    dual_saver s(out); op.serialize_dual(s);
  }
  static void import_dual(Op& op, const double* in) { dual_loader l(in); op.serialize_dual(l); }
};

// message ops.  FLAGS: enum lpmp_msg_flags (optional op members: improvement for adaptive sends, static batch sends)
template <int SIDE, int FLAGS = 0> struct unary_pairwise_message {   // unary (left) <-> pairwise (right), SIDE 0 / 1 of the pair
  static constexpr int kind = LPMP_M_UNARY_PAIRWISE; static constexpr int param = SIDE; static constexpr int flags = FLAGS;
};
template <int FLAGS = 0> struct min_normalised_message {             // test_message, reference test/test_model.hxx:66-98
  static constexpr int kind = LPMP_M_MINNORM; static constexpr int param = 0; static constexpr int flags = FLAGS;
};
// labeling_message<LEFT, RIGHT, INDICES...> (include/factors/labeling_list_factor.hxx:346-402): the registration gives
//     static std::vector<int32_t> match_table();   // per right labeling: index of the matching left labeling, n_left if none
//     static int32_t n_left();
template <int FLAGS = 0> struct labeling_list_message { static constexpr int kind = LPMP_M_LABELING; static constexpr int flags = FLAGS; };

// ---------------------------------------------------------------------------------------------------------------------
// the flat model, owned
struct model_storage {
  std::vector<uint8_t> ftype_primal, f_kind, f_flags;
  std::vector<lpmp_msg_type> mtypes;
  std::vector<int64_t> tab_off{0};
  std::vector<int32_t> tab_data, tab_nleft, f_type, f_dim0, f_dim1, m_type, m_left, m_right, rel_fwd, rel_bwd, part_pairs;
  std::vector<double> cdata, dual;
  std::vector<int64_t> dual_off{0};
  double constant = 0;
  lpmp_model view() const {
    static const double zero = 0;
    lpmp_model m{};
    m.n_ftypes = (int32_t)ftype_primal.size(); m.ftype_computes_primal = ftype_primal.data();
    m.n_mtypes = (int32_t)mtypes.size(); m.mtypes = mtypes.data();
    m.n_tables = (int32_t)tab_nleft.size(); m.tab_off = tab_off.data(); m.tab_data = tab_data.data(); m.tab_nleft = tab_nleft.data();
    m.n_factors = (int64_t)f_type.size(); m.f_type = f_type.data(); m.f_kind = f_kind.data(); m.f_flags = f_flags.data();
    m.f_dim0 = f_dim0.data(); m.f_dim1 = f_dim1.data();
    m.const_data = cdata.empty() ? &zero : cdata.data(); m.dual_data = dual.empty() ? &zero : dual.data();
    m.n_messages = (int64_t)m_type.size(); m.m_type = m_type.data(); m.m_left = m_left.data(); m.m_right = m_right.data();
    m.n_rel_fwd = (int64_t)rel_fwd.size() / 2; m.rel_fwd = rel_fwd.data();
    m.n_rel_bwd = (int64_t)rel_bwd.size() / 2; m.rel_bwd = rel_bwd.data();
    m.constant = constant;
    m.n_part_pairs = (int64_t)part_pairs.size() / 2; m.part_pairs = part_pairs.data();
    return m;
  }
};

namespace detail {
// any variadic class template holding types (the reference's meta::list)
template <class L> struct type_list;
template <template <class...> class L, class... T> struct type_list<L<T...>> {
  static constexpr std::size_t size = sizeof...(T);
  template <class F> static void for_each(F&& f) { std::size_t i = 0; ((f(static_cast<T*>(nullptr), i++)), ...); }
};
template <class C, class = void> struct container_computes_primal { static constexpr bool value = false; };
template <class C> struct container_computes_primal<C, std::void_t<decltype(C::CanComputePrimal())>> { static constexpr bool value = C::CanComputePrimal(); };
template <class Tuple, class F, std::size_t... I> void tuple_for_each(Tuple& t, F&& f, std::index_sequence<I...>) { (f(std::get<I>(t), std::integral_constant<std::size_t, I>{}), ...); }
template <class Tuple, class F> void tuple_for_each(Tuple& t, F&& f) { tuple_for_each(t, f, std::make_index_sequence<std::tuple_size_v<Tuple>>{}); }
inline int schedule_of(bool to_left, bool to_right, bool from_left, bool from_right) {   // factors_messages.hxx:1530-1545, inverted
  if (!to_left && to_right && !from_left && from_right) return LPMP_SCHED_LEFT;
  if (to_left && !to_right && from_left && !from_right) return LPMP_SCHED_RIGHT;
  if (to_left && to_right && from_left && from_right) return LPMP_SCHED_FULL;
  if (to_left && to_right && !from_left && !from_right) return LPMP_SCHED_ONLY_SEND;
  if (!to_left && !to_right && !from_left && !from_right) return LPMP_SCHED_NONE;
  throw std::runtime_error("message container with an unknown message_passing_schedule");
}
}  // namespace detail

// What flatten_through_serialize_dual reads of the LP: references to the members the reference's LP<FMC> holds
// (include/LP_MP.h:476-496, :519, :554, :564).  offloaded<> fills it from its base class.
template <class FactorTuple, class MessageTuple, class FactorVector, class RelationVector, class PartitionVector>
struct lp_view {
  FactorTuple& factors;            // std::tuple<std::vector<FactorContainer_k*>...>, one vector per entry of FMC::FactorList
  MessageTuple& messages;          // std::tuple<std::vector<MessageContainer_k*>...>, one per entry of FMC::MessageList
  const FactorVector& f;           // every factor in add_factor order (as FactorTypeAdapter*)
  const RelationVector& rel_fwd;   // ForwardPassFactorRelation pairs
  const RelationVector& rel_bwd;
  const PartitionVector& partition_graph;
  double constant;
};

// one factor's dual: how many doubles, and the typed load / save through its container
struct factor_io {
  void* container; void (*save)(void*, double*); void (*load)(void*, const double*);
};

template <class FMC, class View>
void flatten_through_serialize_dual(View v, model_storage& s, std::vector<factor_io>* io = nullptr) {
  using FL = detail::type_list<typename FMC::FactorList>;
  using ML = detail::type_list<typename FMC::MessageList>;
  static_assert(std::tuple_size_v<std::remove_reference_t<decltype(v.factors)>> == FL::size, "factors_ does not match FMC::FactorList");
  static_assert(std::tuple_size_v<std::remove_reference_t<decltype(v.messages)>> == ML::size, "messages_ does not match FMC::MessageList");
  s = model_storage();
  const std::size_t nf = v.f.size();
  std::unordered_map<const void*, int32_t> index;    // factor_address_to_index_ (LP_MP.h:521)
  index.reserve(nf);
  for (std::size_t i = 0; i < nf; ++i) index.emplace(static_cast<const void*>(v.f[i]), (int32_t)i);
  auto idx = [&](const auto* factor_adapter) {
    auto it = index.find(static_cast<const void*>(factor_adapter));
    if (it == index.end()) throw std::runtime_error("factor is not part of this LP");
    return it->second;
  };
  // ---- factor types, then every factor in insertion order
  s.ftype_primal.assign(FL::size, 0);
  s.f_type.assign(nf, -1); s.f_kind.assign(nf, 0); s.f_flags.assign(nf, 0); s.f_dim0.assign(nf, 0); s.f_dim1.assign(nf, 0);
  std::vector<std::size_t> csize(nf, 0), dsize(nf, 0);
  std::vector<factor_io> ios(nf);
  detail::tuple_for_each(v.factors, [&](auto& vec, auto type_no) {
    using FC = std::remove_pointer_t<typename std::remove_reference_t<decltype(vec)>::value_type>;
    using Op = typename FC::FactorType;
    static_assert(is_registered_kind<Op>::value, "factor op without an lpmp_offload::device_kind registration: it cannot run on the device (there is no CPU fallback)");
    using K = device_kind<Op>;
    s.ftype_primal[type_no] = detail::container_computes_primal<FC>::value ? 1 : 0;
    for (FC* c : vec) {
      const int32_t i = idx(c);
      Op& op = *c->GetFactor();
      s.f_type[i] = (int32_t)type_no; s.f_kind[i] = (uint8_t)K::kind; s.f_flags[i] = (uint8_t)K::flags;
      K::dims(op, s.f_dim0[i], s.f_dim1[i]);
      csize[i] = K::const_size(op);
      dsize[i] = (std::size_t)lpmp_factor_dual_size(K::kind, s.f_dim0[i], s.f_dim1[i]);
      ios[i] = {c, [](void* p, double* out) { K::export_dual(*static_cast<FC*>(p)->GetFactor(), out); },
                [](void* p, const double* in) { K::import_dual(*static_cast<FC*>(p)->GetFactor(), in); }};
    }
  });
  std::vector<std::size_t> coff(nf + 1, 0);
  s.dual_off.assign(nf + 1, 0);
  for (std::size_t i = 0; i < nf; ++i) {
    if (s.f_type[i] < 0) throw std::runtime_error("factor " + std::to_string(i) + " is in f_ but in none of the per-type lists");
    coff[i + 1] = coff[i] + csize[i]; s.dual_off[i + 1] = s.dual_off[i] + (int64_t)dsize[i];
  }
  s.cdata.assign(coff[nf], 0.0); s.dual.assign((std::size_t)s.dual_off[nf], 0.0);
  detail::tuple_for_each(v.factors, [&](auto& vec, auto) {
    using FC = std::remove_pointer_t<typename std::remove_reference_t<decltype(vec)>::value_type>;
    using K = device_kind<typename FC::FactorType>;
    for (FC* c : vec) {
      const int32_t i = idx(c);
      if (csize[i]) K::export_const(*c->GetFactor(), s.cdata.data() + coff[i]);
      K::export_dual(*c->GetFactor(), s.dual.data() + s.dual_off[i]);
    }
  });
  // ---- message types, then the messages type by type (inside a type in insertion order: all that the per-factor
  // message lists depend on, factors_messages.hxx:3339-3365)
  detail::tuple_for_each(v.messages, [&](auto& vec, auto type_no) {
    using MC = std::remove_pointer_t<typename std::remove_reference_t<decltype(vec)>::value_type>;
    using MsgOp = typename MC::MessageType;
    static_assert(is_registered_message<MsgOp>::value, "message op without an lpmp_offload::device_message registration");
    using R = device_message<MsgOp>;
    lpmp_msg_type t{};
    t.left_ftype = (int32_t)MC::leftFactorNumber; t.right_ftype = (int32_t)MC::rightFactorNumber;
    t.schedule = detail::schedule_of(MC::sends_message_to_left_constexpr(), MC::sends_message_to_right_constexpr(),
                                     MC::receives_message_from_left_constexpr(), MC::receives_message_from_right_constexpr());
    t.n_left = (int32_t)(long)MC::no_left_factors(); t.n_right = (int32_t)(long)MC::no_right_factors();   // the reference returns them as INDEX
    t.kind = R::kind; t.flags = R::flags;
    if constexpr (R::kind == LPMP_M_LABELING) {
      t.param = (int32_t)s.tab_nleft.size();
      const std::vector<int32_t> tab = R::match_table();
      s.tab_data.insert(s.tab_data.end(), tab.begin(), tab.end());
      s.tab_off.push_back((int64_t)s.tab_data.size());
      s.tab_nleft.push_back(R::n_left());
    } else t.param = R::param;
    s.mtypes.push_back(t);
    for (MC* m : vec) {
      s.m_type.push_back((int32_t)type_no);
      s.m_left.push_back(idx(m->GetLeftFactor()));
      s.m_right.push_back(idx(m->GetRightFactor()));
    }
  });
  for (const auto& r : v.rel_fwd) { s.rel_fwd.push_back(idx(std::get<0>(r))); s.rel_fwd.push_back(idx(std::get<1>(r))); }
  for (const auto& r : v.rel_bwd) { s.rel_bwd.push_back(idx(std::get<0>(r))); s.rel_bwd.push_back(idx(std::get<1>(r))); }
  for (const auto& p : v.partition_graph) { s.part_pairs.push_back(idx(p[0])); s.part_pairs.push_back(idx(p[1])); }
  s.constant = v.constant;
  if (io) *io = std::move(ios);
}

// ---------------------------------------------------------------------------------------------------------------------
inline void check(int rc) { if (rc != LPMP_OK) throw std::runtime_error(lpmp_last_error()); }   // the type the reference throws (LP_MP.h:458)

// LP_BASE = the reference's LP<FMC> (or anything with its protected members).  Name hiding, not virtual dispatch: the
// reference's Solver<LP_TYPE, VISITOR> holds an LP_TYPE by value and calls these members on it directly
// (solver.hxx:185, :263-281, :306, :323, :390-392), exactly as it does with LP_tree / LP_with_trees.
template <class LP_BASE>
class offloaded : public LP_BASE {
 public:
  using FMC = typename LP_BASE::FMC;
  using LP_BASE::LP_BASE;
  ~offloaded() { if (engine_) lpmp_destroy(engine_); }
  offloaded(const offloaded&) = delete;
  offloaded& operator=(const offloaded&) = delete;

  void set_device(int device) { device_ = device; }
  // how many passes the engine may run ahead of the Solve loop (0: every call as it comes; default 16)
  void set_speculation(int max_passes_ahead) { speculation_ = max_passes_ahead; if (engine_) check(lpmp_set_speculation(engine_, speculation_)); }
  lpmp_engine* engine() { sync_to_device(); return engine_; }
  // The order the engine suggests for this LP as it stands (lpmp_plan_suggest_order): by_rank[i] = the factor
  // (by insertion index: the i-th add_factor) at position i.  A caller whose insertion order is deep (a grid row by row, chains of
  // local higher-order factors: one launch step per dependent level) builds its LP with
  //   for (i = 0; i + 1 < by_rank.size(); ++i) lp.AddFactorRelation(factor[by_rank[i]], factor[by_rank[i + 1]]);
  // INSTEAD of its own relations: same factors, messages and costs, the updated factors colour by colour.
  std::vector<int32_t> suggested_order(uint64_t seed = 0, int32_t* n_colours = nullptr) {
    sync_to_device();
    lpmp_plan* plan = lpmp_engine_plan_mut(engine_);
    std::vector<int32_t> rank((std::size_t)lpmp_plan_n_factors(plan)), by_rank(rank.size());
    check(lpmp_plan_suggest_order(plan, seed, rank.data(), n_colours));
    for (std::size_t f = 0; f < rank.size(); ++f) by_rank[(std::size_t)rank[f]] = (int32_t)f;
    return by_rank;
  }

  void ComputePass(const std::size_t /*iteration*/) { ready_mode(); check(lpmp_compute_pass(engine_, 1)); device_ahead_ = true; }
  // n consecutive passes in one call (the engine joins them); results equal n calls of ComputePass
  void ComputePasses(const std::size_t n) { if (n == 0) return; ready_mode(); check(lpmp_compute_pass(engine_, (int)n)); device_ahead_ = true; }
  void ComputeForwardPass() { ready_mode(); check(lpmp_compute_forward_pass(engine_)); device_ahead_ = true; }
  void ComputeBackwardPass() { ready_mode(); check(lpmp_compute_backward_pass(engine_)); device_ahead_ = true; }
  void ComputeForwardPassAndPrimal(const std::size_t iteration) { ready_mode(); check(lpmp_compute_forward_pass_and_primal(engine_, iteration)); device_ahead_ = true; }
  void ComputeBackwardPassAndPrimal(const std::size_t iteration) { ready_mode(); check(lpmp_compute_backward_pass_and_primal(engine_, iteration)); device_ahead_ = true; }
  void ComputePassAndPrimal(const std::size_t iteration) { ComputeForwardPassAndPrimal(iteration); ComputeBackwardPassAndPrimal(iteration); }
  double LowerBound() { sync_to_device(); double lb = 0; check(lpmp_lower_bound(engine_, &lb)); return lb; }
  double EvaluatePrimal() { sync_to_device(); double c = 0; check(lpmp_evaluate_primal(engine_, &c)); return c; }
  bool CheckPrimalConsistency() { sync_to_device(); int ok = 0; check(lpmp_check_primal_consistency(engine_, &ok)); return ok != 0; }
  // LP::ComputePass(factorIt, factorItEnd, omegaIt, receive_it), LP_MP.h:981-1005
  template <class FACTOR_ITERATOR, class OMEGA_ITERATOR, class RECEIVE_MASK_ITERATOR>
  void ComputePass(FACTOR_ITERATOR factorIt, const FACTOR_ITERATOR factorItEnd, OMEGA_ITERATOR omegaIt, RECEIVE_MASK_ITERATOR receive_it) {
    sync_to_device();
    std::vector<int32_t> f; std::vector<int64_t> oo{0}, mo{0}; std::vector<double> om; std::vector<uint8_t> mk;
    for (; factorIt != factorItEnd; ++factorIt, ++omegaIt, ++receive_it) {
      f.push_back(index_of(*factorIt));
      for (auto x : *omegaIt) om.push_back(x);
      for (auto x : *receive_it) mk.push_back((uint8_t)x);
      oo.push_back((int64_t)om.size()); mo.push_back((int64_t)mk.size());
    }
    check(lpmp_compute_pass_custom(engine_, (int64_t)f.size(), f.data(), oo.data(), om.data(), mo.data(), mk.data()));
    device_ahead_ = true;
  }
  // the factor ops get the device's duals back (what the reference's factors hold after its own passes)
  void End() { sync_to_host(); LP_BASE::End(); }
  void sync_to_host() {
    if (!engine_ || !device_ahead_) return;
    std::vector<double> d((std::size_t)lpmp_dual_size(engine_));
    check(lpmp_download_duals(engine_, d.data()));
    for (std::size_t i = 0; i < io_.size(); ++i) io_[i].load(io_[i].container, d.data() + model_.dual_off[i]);
    device_ahead_ = false;
  }
  const model_storage& flat_model() { sync_to_device(); return model_; }
  // the flat model without touching a device (inspection, host-only tests)
  const model_storage& flat_model_host_only() { flatten(); return model_; }
  // factor ops were edited on the host without a structural change: upload again before the next pass
  void invalidate() { sync_to_host(); uploaded_ = false; }

 private:
  int32_t index_of(const void* f) const {
    auto it = index_.find(f);
    if (it == index_.end()) throw std::runtime_error("factor is not part of this LP");
    return it->second;
  }
  // anything that calls set_flags_dirty in the reference (LP_MP.h:1623) changes one of these counts
  std::array<std::size_t, 5> signature() const {
    return {this->f_.size(), this->m_.size(), this->forward_pass_factor_rel_.size(), this->backward_pass_factor_rel_.size(), this->partition_graph.size()};
  }
  void sync_to_device() {
    if (this->f_.size() <= 1) throw std::runtime_error("LP needs more than one factor");   // reference assert, LP_MP.h:708
    if (!engine_) {
      check(lpmp_create(device_, &engine_));
      // the reference's Solver asks for ONE pass per iteration and the bound after each (solver.hxx:273-284): let the engine
      // run the coming passes as one launch and hand out each pass's own bound (include/lpmp_engine.h); results unchanged
      check(lpmp_set_speculation(engine_, speculation_));
    }
    if (uploaded_ && signature() == signature_ && this->constant_ == model_.constant) return;
    sync_to_host();                                   // duals of the factors that already existed
    flatten();
    const lpmp_model m = model_.view();
    check(lpmp_upload_model(engine_, &m, LPMP_MEM_HOST, LPMP_MEM_HOST));
    signature_ = signature(); uploaded_ = true; device_ahead_ = false;
  }
  void flatten() {
    using View = lp_view<decltype(this->factors_), decltype(this->messages_), decltype(this->f_),
                         decltype(this->forward_pass_factor_rel_), decltype(this->partition_graph)>;
    flatten_through_serialize_dual<FMC>(View{this->factors_, this->messages_, this->f_, this->forward_pass_factor_rel_,
                                             this->backward_pass_factor_rel_, this->partition_graph, (double)this->constant_}, model_, &io_);
    index_.clear();
    for (std::size_t i = 0; i < this->f_.size(); ++i) index_.emplace(static_cast<const void*>(this->f_[i]), (int32_t)i);
  }
  void ready_mode() {
    sync_to_device();
    // --reparametrizationType / --innerIteration as LP::Begin parsed them (LP_MP.h:710-722), the visitor's current mode
    check(lpmp_set_inner_iterations(engine_, (int)this->inner_iteration_number_arg_.getValue()));
    check(lpmp_set_reparametrization_type(engine_, (int)this->reparametrization_type_));
    const int mode = (int)this->repamMode_;   // LPReparametrizationMode and lpmp_repam_mode share config.hxx:71's numbering
    if (mode < 0 || mode > LPMP_REPAM_MIXED) throw std::runtime_error("no reparametrization mode set");   // LP_MP.h:458 (mixed: refused by the engine)
    check(lpmp_set_reparametrization(engine_, mode));
  }

  int device_ = 0, speculation_ = 16;
  lpmp_engine* engine_ = nullptr;
  bool uploaded_ = false, device_ahead_ = false;
  std::array<std::size_t, 5> signature_{};
  model_storage model_;
  std::vector<factor_io> io_;
  std::unordered_map<const void*, int32_t> index_;
};

}  // namespace lpmp_offload
