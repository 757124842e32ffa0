// The ncclUniqueId hand-out of lp_mp_amd/include/lpmp_multi_gpu.hxx (rccl_world::hand_out_id) without any GPU or RCCL call:
// rank 0 serves a known 128-byte pattern on 127.0.0.1:<port>, the other ranks fetch it.  Usage: test_id_handout rank world port
// [timeout_s]; prints the id as hex.  tests/test_multi_gpu.py starts the ranks (and an intruder with a wrong greeting).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lpmp_multi_gpu.hxx"

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s rank world port [timeout_s]\n", argv[0]); return 2; }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]), port = std::atoi(argv[3]);
  const double timeout_s = argc > 4 ? std::atof(argv[4]) : 30.0;
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  if (rank == 0) for (size_t i = 0; i < sizeof(id); ++i) ((unsigned char*)&id)[i] = (unsigned char)(37 * i + port % 251);
  try {
    lpmp_mgpu::rccl_world::hand_out_id(id, rank, world, "127.0.0.1", port, timeout_s);
  } catch (const std::exception& e) { std::fprintf(stderr, "rank %d: %s\n", rank, e.what()); return 1; }
  for (size_t i = 0; i < sizeof(id); ++i) std::printf("%02x", ((unsigned char*)&id)[i]);
  std::printf("\n");
  return 0;
}
