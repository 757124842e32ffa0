// oracle/ref_two_smallest.cpp — TEST INFRASTRUCTURE (never linked or called by the product).
// The REFERENCE's own two_smallest_elements (compiled from /root/reference/include/help_functions.hxx:106-120 where it lies,
// oracle/build_ref.py -> oracle/_ref/ref_two_smallest): the scalar form of the two-minimum behind the O(L) Potts message
// (min over x2 of diff * [x1 != x2] + m[x2] = min(m[x1], diff + (m[x1] is the smallest ? second smallest : smallest))).
// tests/test_oracle_ref.py holds the oracle's Potts message and (-m gpu) the device's two_min butterfly against it.
// stdin: any number of vectors, each as "n v_1 ... v_n" (strtod syntax: hex floats, inf, -inf).   stdout per vector: "%a %a".
#include <array>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "help_functions.hxx"

int main() {
  long n = 0;
  while (std::scanf("%ld", &n) == 1) {
    std::vector<double> v((std::size_t)n);
    for (auto& x : v) { char tok[64]; if (std::scanf("%63s", tok) != 1) return 2; x = std::strtod(tok, nullptr); }
    const std::array<double, 2> s = LP_MP::two_smallest_elements<double>(v.begin(), v.end());
    std::printf("%a %a\n", s[0], s[1]);
  }
  return 0;
}
