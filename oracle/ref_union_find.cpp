// oracle/ref_union_find.cpp — TEST INFRASTRUCTURE (never linked or called by the product).
// The REFERENCE's own union_find (compiled from /root/reference/include/union_find.hxx where it lies, oracle/build_ref.py ->
// oracle/_ref/ref_union_find) driven the way LP::construct_factor_partition drives it (include/LP_MP.h:1724-1745: merge every
// put_in_same_partition pair in call order, get_contiguous_ids, count the updated factors per set, drop the sets without any): the
// numbering of the partitions of a partition sweep depends on how that class picks roots and numbers them, so
// tests/test_oracle_ref.py holds lpmp_plan_get_partitions (and the C oracle) against this instead of against a restatement.
// stdin: n_factors, the n updated flags (0/1), n_pairs, the pairs.   stdout: "partitions P", then "factor i partition p" per UPDATED factor.
#include <cassert>
#include <cstdio>
#include <limits>
#include <vector>

#include "union_find.hxx"

int main() {
  std::size_t n = 0, np = 0;
  if (std::scanf("%zu", &n) != 1) return 2;
  std::vector<int> updated(n);
  for (auto& u : updated) if (std::scanf("%d", &u) != 1) return 2;
  if (std::scanf("%zu", &np) != 1) return 2;
  LP_MP::union_find uf(n);
  for (std::size_t k = 0; k < np; ++k) { std::size_t a, b; if (std::scanf("%zu %zu", &a, &b) != 2) return 2; uf.merge(a, b); }
  auto contiguous_ids = uf.get_contiguous_ids();
  std::vector<std::size_t> partition_size(uf.count(), 0);
  for (std::size_t i = 0; i < contiguous_ids.size(); ++i) if (updated[i]) partition_size[contiguous_ids[uf.find(i)]]++;
  std::vector<std::size_t> to_partition(contiguous_ids.size(), std::numeric_limits<std::size_t>::max());
  std::size_t P = 0;
  for (std::size_t i = 0; i < partition_size.size(); ++i) if (partition_size[i] > 0) to_partition[i] = P++;
  std::printf("partitions %zu\n", P);
  for (std::size_t i = 0; i < n; ++i) if (updated[i]) std::printf("factor %zu partition %zu\n", i, to_partition[contiguous_ids[uf.find(i)]]);
  return 0;
}
