// for_each_transfer (lp_mp_amd/include/lpmp_multi_gpu.hxx): the order in which the ranks of the C++ multi-GPU host issue the
// ncclSend / ncclRecv calls of one exchange.  RCCL pairs the k-th send of rank a to rank b with the k-th receive of b from a; the
// 1-GPU test box cannot run two RCCL ranks, so the pairing is checked here on the host: every rank's sequence is generated, and
// for every ordered pair of ranks the (source part, destination part) sequence of a's sends to b must equal that of b's receives
// from a.  Usage: test_transfer_order  (exit code 0 = every layout pairs up)
#include <cstdio>
#include <utility>
#include <vector>

#include "lpmp_multi_gpu.hxx"

int main() {
  int checked = 0;
  for (int world = 1; world <= 8; ++world)
    for (int ppr = 1; ppr <= 4; ++ppr) {
      const int n_parts = world * ppr;
      // sends[a][b] / recvs[b][a]: the transfers between ranks a -> b as each side issues them
      std::vector<std::vector<std::vector<std::pair<int, int>>>> sends(world, std::vector<std::vector<std::pair<int, int>>>(world)), recvs = sends;
      for (int rank = 0; rank < world; ++rank)
        lpmp_mgpu::for_each_transfer(n_parts, rank, ppr, [&](int src, int dst, bool src_here, bool dst_here) {
          if (src_here == (src / ppr == rank) && dst_here == (dst / ppr == rank)) ++checked;
          if (src_here && dst_here) return;                              // parts of one rank: a device copy, no RCCL call
          if (src_here) sends[rank][dst / ppr].push_back({src, dst});
          else recvs[rank][src / ppr].push_back({src, dst});
        });
      for (int a = 0; a < world; ++a)
        for (int b = 0; b < world; ++b) {
          if (a == b) { if (!sends[a][b].empty() || !recvs[a][b].empty()) { std::printf("world %d ppr %d: rank %d talks to itself\n", world, ppr, a); return 1; } continue; }
          if (sends[a][b] != recvs[b][a]) { std::printf("world %d ppr %d: sends of rank %d to %d do not pair with the receives\n", world, ppr, a, b); return 1; }
          if ((int)sends[a][b].size() != ppr * ppr) { std::printf("world %d ppr %d: %zu transfers between ranks %d and %d\n", world, ppr, sends[a][b].size(), a, b); return 1; }
        }
    }
  std::printf("for_each_transfer: sends and receives pair up for world 1..8 x 1..4 parts per rank (%d calls checked)\n", checked);
  return 0;
}
