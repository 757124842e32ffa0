// oracle/ref_two_dim.cpp — TEST INFRASTRUCTURE (never linked or called by the product).
// A driver around the REFERENCE's own CSR container, compiled from /root/reference/include/two_dimensional_variable_array.hxx where
// it lies (oracle/build_ref.py -> oracle/_ref/ref_two_dim): `two_dim_variable_array<REAL>` is the `weight_array` the reference's
// sweep indexes (omega.forward[i][j], include/LP_MP.h:989-992; allocate_omega :1008-1040 sizes row i by the number of sending
// messages of the i-th updated factor).  It prints, for the row sizes read from stdin, what the container itself says about its
// layout — row count, row sizes, the element offset of every row's first entry and the storage order of (row, column) — so that
// tests/test_oracle_ref.py can hold lpmp_plan_get_omega's CSR against the real thing instead of against a restatement.
// stdin: n, then n row sizes.   stdout: "rows n", "row i size s offset o" per row, "flat r c" per element in storage order.
#include <cstdio>
#include <vector>

#include "two_dimensional_variable_array.hxx"

int main() {
  std::size_t n = 0;
  if (std::scanf("%zu", &n) != 1) return 2;
  std::vector<std::size_t> size(n);
  for (auto& x : size) if (std::scanf("%zu", &x) != 1) return 2;
  LP_MP::two_dim_variable_array<double> a(size.begin(), size.end(), 0.0);
  for (std::size_t i = 0; i < a.size(); ++i) for (std::size_t j = 0; j < a[i].size(); ++j) a(i, j) = 1000.0 * (double)i + (double)j;
  std::printf("rows %zu\n", a.size());
  const double* base = nullptr;
  std::size_t total = 0;
  for (std::size_t i = 0; i < a.size(); ++i) { if (!base && a[i].size() > 0) base = &a(i, 0); total += a[i].size(); }
  for (std::size_t i = 0; i < a.size(); ++i)
    std::printf("row %zu size %zu offset %ld\n", i, a[i].size(), a[i].size() > 0 ? (long)(&a(i, 0) - base) : -1L);
  for (std::size_t k = 0; k < total; ++k) { const long v = (long)base[k]; std::printf("flat %ld %ld\n", v / 1000, v % 1000); }
  return 0;
}
