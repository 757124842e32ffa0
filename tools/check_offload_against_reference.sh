#!/bin/bash
# tools/check_offload_against_reference.sh — COMPILE CHECK ONLY (build container, never on the GPU box).
#
# Does lp_mp_amd/include/lpmp_offload.hxx fit the reference's REAL headers?  The adapter is otherwise compiled only against
# tests/cpp/mock_reference_lp.hxx, a test double written by the same author; this script instantiates
#     lpmp_offload::offloaded<LP_MP::LP<FMC>>   and   LP_MP::Solver<offloaded<...>, LP_MP::StandardVisitor>
# with the factor / message types of the reference's own test/test_model.hxx against /root/reference/include and runs
# `g++ -std=c++17 -fsyntax-only`.  The reference's headers include four third-party headers whose submodules are empty in this
# image (tclap, meta, libsimdpp, DD_ILP); for THIS syntax check throw-away declarations of just the names used are written into a
# temp dir that is deleted afterwards.  Nothing is built, linked or run; this is not an oracle, pins nothing for parity and
# nothing of it travels to the GPU box (the reference stays "unbuildable here" for oracle purposes).
#
#   tools/check_offload_against_reference.sh [output log]      exit 0 = the adapter compiles against the real headers
set -u
REF=${LPMP_REFERENCE:-/root/reference}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LOG=${1:-/dev/stdout}
if [ ! -d "$REF/include" ]; then echo "no reference tree at $REF: nothing to check" > "$LOG"; exit 0; fi
T=$(mktemp -d /tmp/lpmp_offload_check.XXXXXX)
trap 'rm -rf "$T"' EXIT
mkdir -p "$T/tclap" "$T/meta" "$T/simdpp"

# ---- declarations only, just enough for the parser ----
cat > "$T/tclap/CmdLine.h" <<'EOF'
#pragma once
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
namespace TCLAP {
struct ArgException { std::string error() const { return {}; } std::string argId() const { return {}; } };
template <class T> struct Constraint { virtual std::string description() const = 0; virtual std::string shortID() const = 0; virtual bool check(const T&) const = 0; virtual ~Constraint() {} };
template <class T> struct ValuesConstraint : Constraint<T> { ValuesConstraint(std::vector<T>&) {} std::string description() const override { return {}; } std::string shortID() const override { return {}; } bool check(const T&) const override { return true; } };
struct CmdLine { CmdLine(const std::string&, char = ' ', const std::string& = "", bool = true) {} void parse(int, char**) {} void parse(int, const char* const*) {} void parse(std::vector<std::string>&) {} template <class A> void add(A&) {} };
struct Arg { bool isSet() const { return false; } };
template <class T> struct ValueArg : Arg {
  T v{};
  ValueArg(const std::string&, const std::string&, const std::string&, bool, T d, const std::string&, CmdLine&) : v(d) {}
  ValueArg(const std::string&, const std::string&, const std::string&, bool, T d, Constraint<T>*, CmdLine&) : v(d) {}
  ValueArg(const std::string&, const std::string&, const std::string&, bool, T d, const std::string&) : v(d) {}
  ValueArg(const std::string&, const std::string&, const std::string&, bool, T d, Constraint<T>*) : v(d) {}
  T& getValue() { return v; } const T& getValue() const { return v; }
};
template <class T> struct MultiArg : Arg { std::vector<T> v; MultiArg(const std::string&, const std::string&, const std::string&, bool, const std::string&, CmdLine&) {} const std::vector<T>& getValue() const { return v; } };
struct SwitchArg : Arg { bool v; SwitchArg(const std::string&, const std::string&, const std::string&, CmdLine&, bool d = false) : v(d) {} SwitchArg(const std::string&, const std::string&, const std::string&, bool d = false) : v(d) {} bool getValue() const { return v; } };
template <class T> struct UnlabeledValueArg : Arg { T v{}; UnlabeledValueArg(const std::string&, const std::string&, bool, T d, const std::string&, CmdLine&) : v(d) {} T& getValue() { return v; } };
}
EOF
cat > "$T/meta/meta.hpp" <<'EOF'
#pragma once
#include <cstddef>
#include <type_traits>
#include <utility>
namespace meta {
template <class... T> struct list { using type = list; static constexpr std::size_t size() noexcept { return sizeof...(T); } };
template <std::size_t N> using size_t = std::integral_constant<std::size_t, N>;
template <class L> struct size;
template <class... T> struct size<list<T...>> : std::integral_constant<std::size_t, sizeof...(T)> {};
namespace detail {
template <class L, std::size_t N> struct at;
template <class H, class... T> struct at<list<H, T...>, 0> { using type = H; };
template <class H, class... T, std::size_t N> struct at<list<H, T...>, N> : at<list<T...>, N - 1> {};
template <class L, class X> struct find_index;
template <class X> struct find_index<list<>, X> : std::integral_constant<std::size_t, 0> {};
template <class H, class... T, class X> struct find_index<list<H, T...>, X> : std::integral_constant<std::size_t, std::is_same<H, X>::value ? 0 : 1 + find_index<list<T...>, X>::value> {};
template <class... L> struct concat;
template <> struct concat<> { using type = list<>; };
template <class... A> struct concat<list<A...>> { using type = list<A...>; };
template <class... A, class... B, class... R> struct concat<list<A...>, list<B...>, R...> : concat<list<A..., B...>, R...> {};
template <class L, class F> struct transform;
template <class... T, class F> struct transform<list<T...>, F> { using type = list<typename F::template invoke<T>...>; };
template <class L, class P> struct filter;
template <class P> struct filter<list<>, P> { using type = list<>; };
template <class H, class... T, class P> struct filter<list<H, T...>, P> {
  using rest = typename filter<list<T...>, P>::type;
  using type = typename std::conditional<P::template invoke<H>::value, typename concat<list<H>, rest>::type, rest>::type;
};
template <class L, class X> struct contains;
template <class X> struct contains<list<>, X> : std::false_type {};
template <class H, class... T, class X> struct contains<list<H, T...>, X> : std::integral_constant<bool, std::is_same<H, X>::value || contains<list<T...>, X>::value> {};
template <class In, class Out> struct unique;
template <class Out> struct unique<list<>, Out> { using type = Out; };
template <class H, class... T, class... O> struct unique<list<H, T...>, list<O...>> : unique<list<T...>, typename std::conditional<contains<list<O...>, H>::value, list<O...>, list<O..., H>>::type> {};
template <class L, class P> struct any_of;
template <class... T, class P> struct any_of<list<T...>, P> : std::integral_constant<bool, (false || ... || P::template invoke<T>::value)> {};
template <class F, class L> struct apply;
template <class F, class... T> struct apply<F, list<T...>> { using type = typename F::template invoke<T...>; };
}
template <class L, std::size_t N> using at_c = typename detail::at<L, N>::type;
template <class L, class X> struct find_index : detail::find_index<L, X> {};   // npos == size when absent (the reference only compares with size)
template <class... L> using concat = typename detail::concat<L...>::type;
template <class L, class F> using transform = typename detail::transform<L, F>::type;
template <class L, class P> using filter = typename detail::filter<L, P>::type;
template <class L> using unique = typename detail::unique<L, list<>>::type;
template <class L, class P> using any_of = detail::any_of<L, P>;
template <bool B, class T, class E> using if_c = typename std::conditional<B, T, E>::type;
template <template <class...> class C> struct quote { template <class... T> using invoke = C<T...>; };
template <class F, class L> using apply = typename detail::apply<F, L>::type;
template <class... T, class F> F for_each(list<T...>, F f) { (f(T{}), ...); return f; }
}
EOF
cat > "$T/simdpp/simd.h" <<'EOF'
#pragma once
#include <algorithm>
namespace simdpp {
template <unsigned N> struct float64 { double v[N]; float64() {} float64(const float64&) = default; template <class X> float64(const X&) {} template <class X> float64& operator=(const X&) { return *this; } float64<N>& vec(unsigned) { return *this; } };
template <unsigned N> struct float32 { float v[N]; float32() {} float32(const float32&) = default; template <class X> float32(const X&) {} template <class X> float32& operator=(const X&) { return *this; } float32<(N > 4 ? N / 2 : N)> vec(unsigned) const { return {}; } };
struct load_proxy { template <unsigned N> operator float64<N>() const { return {}; } template <unsigned N> operator float32<N>() const { return {}; } };
inline load_proxy load(const void*) { return {}; }
inline load_proxy load_u(const void*) { return {}; }
inline load_proxy load_splat(const void*) { return {}; }
template <class V> void store(void*, const V&) {}
template <class V> void store_u(void*, const V&) {}
template <class... A> load_proxy make_float(A...) { return {}; }
template <unsigned N> float64<N> min(const float64<N>& a, const float64<N>&) { return a; }
template <unsigned N> float64<N> max(const float64<N>& a, const float64<N>&) { return a; }
template <unsigned N> float32<N> min(const float32<N>& a, const float32<N>&) { return a; }
template <unsigned N> float32<N> max(const float32<N>& a, const float32<N>&) { return a; }
template <unsigned N> float64<N> operator+(const float64<N>& a, const float64<N>&) { return a; }
template <unsigned N> float64<N> operator-(const float64<N>& a, const float64<N>&) { return a; }
template <unsigned N> float64<N> operator*(const float64<N>& a, const float64<N>&) { return a; }
template <unsigned N> float32<N> operator+(const float32<N>& a, const float32<N>&) { return a; }
template <unsigned N> float32<N> operator-(const float32<N>& a, const float32<N>&) { return a; }
template <unsigned N> float32<N> operator*(const float32<N>& a, const float32<N>&) { return a; }
template <unsigned N> double reduce_min(const float64<N>&) { return 0; }
template <unsigned N> double reduce_max(const float64<N>&) { return 0; }
template <unsigned N> double reduce_add(const float64<N>&) { return 0; }
template <unsigned N> float reduce_min(const float32<N>&) { return 0; }
template <unsigned N> float reduce_max(const float32<N>&) { return 0; }
template <unsigned N> float reduce_add(const float32<N>&) { return 0; }
template <unsigned I, unsigned N> double extract(const float64<N>&) { return 0; }
template <unsigned I, unsigned N> float extract(const float32<N>&) { return 0; }
template <unsigned N, class A, class B> void split(const float32<N>&, A&, B&) {}
template <unsigned N, class A, class B> void split(const float64<N>&, A&, B&) {}
inline void prefetch_read(const void*) {}
inline void prefetch_write(const void*) {}
}
EOF
cat > "$T/DD_ILP.hxx" <<'EOF'
#pragma once
#include <cstddef>
#include <string>
#include <vector>
namespace DD_ILP {
struct sat_solver {}; struct problem_export {};
struct variable_counters {};
template <class BASE> struct external_solver_interface {
  struct variable { template <class... A> variable(A...) {} };
  struct vector { template <class... A> vector(A...) {} variable operator[](std::size_t) const { return {}; } std::size_t size() const { return 0; } variable* begin() const { return nullptr; } variable* end() const { return nullptr; } };
  struct matrix { template <class... A> matrix(A...) {} variable operator()(std::size_t, std::size_t) const { return {}; } std::size_t dim1() const { return 0; } std::size_t dim2() const { return 0; } };
  struct tensor { template <class... A> tensor(A...) {} variable operator()(std::size_t, std::size_t, std::size_t) const { return {}; } };
  template <class... A> variable add_variable(A...) { return {}; }
  template <class... A> vector add_vector(A...) { return {}; }
  template <class... A> matrix add_matrix(A...) { return {}; }
  template <class... A> tensor add_tensor(A...) { return {}; }
  template <class... A> variable load_variable(A...) { return {}; }
  template <class... A> vector load_vector(A...) { return {}; }
  template <class... A> matrix load_matrix(A...) { return {}; }
  template <class... A> tensor load_tensor(A...) { return {}; }
  template <class... A> void add_variable_objective(A...) {}
  template <class... A> void add_vector_objective(A...) {}
  template <class... A> void add_matrix_objective(A...) {}
  template <class... A> void add_tensor_objective(A...) {}
  template <class... A> void add_objective(A...) {}
  template <class... A> void add_simplex_constraint(A...) {}
  template <class... A> void add_at_most_one_constraint(A...) {}
  template <class... A> void add_implication(A...) {}
  template <class... A> void make_equal(A...) {}
  template <class... A> bool solution(A...) const { return false; }
  variable_counters get_variable_counters() const { return {}; }
  void set_variable_counters(const variable_counters&) {}
  void init_variable_loading() {}
  bool solve() { return false; }
  template <class... A> void write_to_file(A...) {}
};
}
EOF

# ---- the translation unit: the reference's LP / Solver / visitor + the adapter + the reference's own test model types ----
cat > "$T/check.cpp" <<EOF
#include "LP_MP.h"
#include "solver.hxx"
#include "visitors/standard_visitor.hxx"
#include "$REF/test/test_model.hxx"
#include "$ROOT/lp_mp_amd/include/lpmp_offload.hxx"

// registrations for the ops of test/test_model.hxx (outside the ops, as INTEGRATION.md 2a shows)
template <> struct lpmp_offload::device_kind<LP_MP::test_factor> : lpmp_offload::vector_kind<> {};
template <> struct lpmp_offload::device_message<LP_MP::test_message> : lpmp_offload::min_normalised_message<> {};

using LP_device = lpmp_offload::offloaded<LP_MP::LP<LP_MP::test_FMC>>;
using SolverT = LP_MP::Solver<LP_device, LP_MP::StandardVisitor>;

// every member the adapter re-declares or reads, instantiated
template class lpmp_offload::offloaded<LP_MP::LP<LP_MP::test_FMC>>;

double drive(SolverT& s) {
  auto& lp = s.GetLP();
  // the factors / messages of test/test_model.hxx:140-175 (build_test_model itself needs LP_with_trees::add_tree)
  auto* f1 = lp.template add_factor<typename LP_MP::test_FMC::factor>(0, 1);
  auto* f2 = lp.template add_factor<typename LP_MP::test_FMC::factor>(1, 0);
  auto* f3 = lp.template add_factor<typename LP_MP::test_FMC::factor>(0, 0);
  lp.template add_message<typename LP_MP::test_FMC::message>(f1, f2);
  lp.template add_message<typename LP_MP::test_FMC::message>(f2, f3);
  lp.AddFactorRelation(f1, f2); lp.ForwardPassFactorRelation(f2, f3); lp.BackwardPassFactorRelation(f3, f2);
  lp.put_in_same_partition(f1, f2);
  lp.Begin();
  lp.set_reparametrization(LP_MP::LPReparametrizationMode::Anisotropic);
  lp.ComputePass(1);
  lp.ComputePasses(3);
  lp.ComputeForwardPass(); lp.ComputeBackwardPass();
  {  // the iterator-range pass (LP_MP.h:981-1005) with the reference's own weight / mask arrays (get_omega, :412-460)
    auto omega = lp.get_omega();
    std::vector<LP_MP::FactorTypeAdapter*> list{f1, f2, f3};
    lp.ComputePass(list.begin(), list.end(), omega.forward.begin(), omega.receive_mask_forward.begin());
  }
  lp.ComputeForwardPassAndPrimal(1); lp.ComputeBackwardPassAndPrimal(1);
  const bool ok = lp.CheckPrimalConsistency();
  const double c = lp.EvaluatePrimal();
  const auto& flat = lp.flat_model_host_only();
  lp.End();
  return lp.LowerBound() + c + (ok ? 1 : 0) + (double)flat.f_type.size();
}
int run_the_reference_solver_unchanged(SolverT& s) { return s.Solve(); }
EOF
{
  echo "== tools/check_offload_against_reference.sh: g++ -std=c++17 -fsyntax-only of offloaded<LP_MP::LP<test_FMC>> + Solver<..., StandardVisitor>"
  echo "== against $REF/include (LP_MP.h, solver.hxx, visitors/standard_visitor.hxx, test/test_model.hxx); third-party headers: throw-away declarations"
  g++ --version | head -1
} > "$LOG"
g++ -std=c++17 -fsyntax-only -w -DNDEBUG -I "$T" -I "$REF/include" -I "$REF/test" "$T/check.cpp" >> "$LOG" 2>&1
rc=$?
if [ $rc -eq 0 ]; then echo "RESULT: OK — the adapter compiles against the reference's real headers" >> "$LOG"; else echo "RESULT: FAILED (rc $rc)" >> "$LOG"; fi
exit $rc
