// Re-creates the random inputs of the reference runs recorded in SURVEY.md 8(c):
// std::mt19937_64(seed) + std::uniform_real_distribution<double>(0,1), libstdc++.
// usage: gen_mt19937 <seed> <count> <out.bin>   (writes <count> little-endian doubles)
// TEST INFRASTRUCTURE ONLY.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
int main(int argc, char** argv) {
  if (argc != 4) { std::fprintf(stderr, "usage: %s seed count out.bin\n", argv[0]); return 2; }
  const unsigned long long seed = std::strtoull(argv[1], nullptr, 10);
  const size_t n = std::strtoull(argv[2], nullptr, 10);
  std::mt19937_64 gen(seed);
  std::uniform_real_distribution<double> dist(0.0, 1.0);
  std::vector<double> buf(n);
  for (size_t i = 0; i < n; ++i) buf[i] = dist(gen);
  FILE* f = std::fopen(argv[3], "wb");
  if (!f) return 1;
  std::fwrite(buf.data(), sizeof(double), n, f);
  std::fclose(f);
  return 0;
}
