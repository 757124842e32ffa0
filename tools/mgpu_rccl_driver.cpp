// tools/mgpu_rccl_driver.cpp — the partitioned sweep of DESIGN.md 7 driven from a C++ host over the C ABI and RCCL
// (lp_mp_amd/include/lpmp_multi_gpu.hxx): what INTEGRATION.md 2c describes, as a program.
//
// One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT from the environment, as torchrun sets them; without
// them a single process).  Workload: row strips of a grid MRF — rank r holds parts r*P ... r*P+P-1 (P = --parts-per-rank)
// of a (n_parts * H) x W grid, costs generated in HBM from the counter stream (the model bench.py --gpus N runs).
//
//   mgpu_rccl_driver [--H 64] [--W 64] [--L 8] [--pairwise dense|potts] [--order colour_major|row_major] [--passes 4]
//                    [--parts-per-rank 1] [--boundary pass|sweep] [--mode 0] [--out PREFIX] [--time K]
//                    [--schedule boundary|overlap|lockstep] [--ghost-rows 12] [--chunk 0] [--graph n m [--order-file f] [--part-file f]] [--model-file f --part-file f]
// --schedule overlap (lpmp_overlap.hxx): the EXACT schedule for colour-major grids — every part a window with ghost rows of the
// global (n_parts * H) x W grid, plain lpmp_compute_pass calls, one exchange per (ghost-rows / 2 - 1) passes (or --chunk).
//
// Prints one JSON line on rank 0 (lower bound before / after; with --time K: ms per pass, msg-updates/s, and — from an untimed
// repetition with every exchange bracketed by events, maximum over the ranks — compute_ms_per_pass, exchange_ms_per_pass,
// exchanges_per_pass, exchange_bytes_out_per_pass: where the time of a pass went); with --out every part's
// packed duals go to PREFIX.<part>.bin (tests/test_multi_gpu.py compares them with lp_mp_amd/multi_gpu.py's run).
// Build: hipcc -std=c++17 -O2 tools/mgpu_rccl_driver.cpp -o build/mgpu_rccl_driver -L lp_mp_amd/csrc -llpmp_engine -lrccl
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../lp_mp_amd/include/lpmp_overlap.hxx"
#include "../lp_mp_amd/include/lpmp_lockstep.hxx"

using namespace lpmp_mgpu;

static int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
static void print_probe(const rccl_world::probe_result& r) {
  std::printf(", \"compute_ms_per_pass\": %.6f, \"exchange_ms_per_pass\": %.6f, \"exchanges_per_pass\": %.3f, \"exchange_bytes_out_per_pass\": %.0f",
              r.compute_ms, r.exchange_ms, r.exchanges, r.bytes_out);
}

int main(int argc, char** argv) {
  int H = 64, W = 64, L = 8, passes = 4, ppr = 1, mode = LPMP_REPAM_ANISOTROPIC, timed = 0;
  bool potts = false, colour = true, every_pass = true, overlap = false, lockstep = false;
  int ghost = 12, chunk = 0;
  long long graph_n = 0, graph_m = 0;
  std::string order_file, part_file, model_path;
  std::string out;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value after %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "--H") H = std::atoi(next()); else if (a == "--W") W = std::atoi(next()); else if (a == "--L") L = std::atoi(next());
    else if (a == "--passes") passes = std::atoi(next()); else if (a == "--parts-per-rank") ppr = std::atoi(next());
    else if (a == "--mode") mode = std::atoi(next()); else if (a == "--time") timed = std::atoi(next());
    else if (a == "--pairwise") potts = std::string(next()) == "potts";
    else if (a == "--order") colour = std::string(next()) == "colour_major";
    else if (a == "--boundary") every_pass = std::string(next()) == "pass";
    else if (a == "--schedule") { const std::string v = next(); overlap = v == "overlap"; lockstep = v == "lockstep"; }
    else if (a == "--graph") { graph_n = std::atoll(next()); graph_m = std::atoll(next()); }
    else if (a == "--order-file") order_file = next(); else if (a == "--part-file") part_file = next();
    else if (a == "--model-file") model_path = next();
    else if (a == "--ghost-rows") ghost = std::atoi(next()); else if (a == "--chunk") chunk = std::atoi(next());
    else if (a == "--out") out = next();
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local_rank = env_int("LOCAL_RANK", rank);
  try {
    int n_dev = 0;
    hip_ok(hipGetDeviceCount(&n_dev), "hipGetDeviceCount");
    if (n_dev <= 0) throw std::runtime_error("no HIP device: the engine has no CPU path");
    const int device = local_rank % n_dev;
    hip_ok(hipSetDevice(device), "hipSetDevice");
    hipStream_t stream = nullptr;
    hip_ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
    rccl_world w;
    const char* addr = std::getenv("MASTER_ADDR");
    // (the id hand-out takes the port next to MASTER_PORT: a torch.distributed store of the same launch may hold that one)
    w.init(rank, world, ppr, addr && *addr ? addr : "127.0.0.1", env_int("LPMP_NCCL_ID_PORT", env_int("MASTER_PORT", 29500) + 1), stream);

    const int n_parts = world * ppr;
    if (overlap) {
      if (!colour) throw std::runtime_error("--schedule overlap is for colour-major grids");
      std::vector<std::unique_ptr<window_sweep>> own;
      std::vector<window_sweep*> parts;
      for (int k = 0; k < ppr; ++k) {
        own.emplace_back(new window_sweep());
        own.back()->build(strip_window(H, W, L, potts, rank * ppr + k, n_parts, ghost, 1), device, stream, mode);
        parts.push_back(own.back().get());
      }
      const double lb0 = overlap_lower_bound(parts, w);
      overlap_compute_pass(parts, w, passes, chunk);
      const double lb1 = overlap_lower_bound(parts, w);
      if (!out.empty())
        for (window_sweep* p : parts) {
          const std::vector<double> d = p->download_duals();
          const std::string path = out + "." + std::to_string(p->wm.part) + ".bin";
          FILE* f = std::fopen(path.c_str(), "wb");
          if (!f || std::fwrite(d.data(), sizeof(double), d.size(), f) != d.size()) throw std::runtime_error("cannot write " + path);
          std::fclose(f);
        }
      double ms_per_pass = 0;
      rccl_world::probe_result pr;
      if (timed > 0) {
        (void)w.all_reduce_sum(0.0);
        const auto t0 = std::chrono::steady_clock::now();
        overlap_compute_pass(parts, w, timed, chunk);
        (void)w.all_reduce_sum(0.0);
        ms_per_pass = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
        pr = w.probe_run([&] { overlap_compute_pass(parts, w, timed, chunk); }, timed);
      }
      if (rank == 0) {
        const double E = (double)n_parts * H * (W - 1) + ((double)n_parts * H - 1) * W;       // edges of the whole grid: 4 E message updates per anisotropic pass
        std::printf("{\"driver\": \"mgpu_rccl_driver (C++ host, C ABI + RCCL)\", \"schedule\": \"overlap\", \"world\": %d, \"parts\": %d, \"grid_per_part\": [%d, %d], "
                    "\"labels\": %d, \"pairwise\": \"%s\", \"ghost_rows\": %d, \"passes_between_exchanges\": %d, \"passes\": %d, \"lower_bound_before\": %.17g, "
                    "\"lower_bound_after\": %.17g", world, n_parts, H, W, L, potts ? "potts" : "dense", ghost, chunk > 0 ? chunk : (ghost - 2) / 2, passes, lb0, lb1);
        if (timed > 0) std::printf(", \"ms_per_pass\": %.6f, \"msg_updates_per_s\": %.6g", ms_per_pass, 4.0 * E / (ms_per_pass * 1e-3));
        if (timed > 0) print_probe(pr);
        std::printf("}\n");
      }
      own.clear();
      w.destroy();
      (void)hipStreamDestroy(stream);
      return 0;
    }
    if (lockstep) {
      // the exact schedule for any MRF and partition (lpmp_lockstep.hxx): strips of the grid, or --graph n m (the C4-style
      // random graph in the generator's index order, contiguous index ranges as parts)
      // (--order-file / --part-file: int64 arrays, one entry per variable — a variable order such as ordering.colour_major_order
      // and a partition such as multi_gpu.graph_partition, computed by the caller)
      auto load = [](const std::string& path, long long n) {
        std::vector<int64_t> v((size_t)n);
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f || std::fread(v.data(), sizeof(int64_t), v.size(), f) != v.size()) throw std::runtime_error("cannot read " + std::to_string(n) + " int64 values from " + path);
        std::fclose(f);
        return v;
      };
      if (!model_path.empty()) {
        // any model with `left`-schedule messages (labeling-list factors: C5, multicut), read from a file with its costs;
        // --part-file: the part of every FACTOR (used for the variables), int64
        model_file mf; mf.load(model_path);
        const lpmp_model gm = mf.view();
        if (part_file.empty()) throw std::runtime_error("--model-file needs --part-file");
        const std::vector<int64_t> p64 = load(part_file, gm.n_factors);
        const std::vector<int32_t> part_of(p64.begin(), p64.end());
        lockstep_plan pl;
        pl.build(gm, part_of, n_parts, mode);
        std::vector<std::unique_ptr<lockstep_part>> own;
        std::vector<lockstep_part*> parts;
        for (int k = 0; k < ppr; ++k) { own.emplace_back(new lockstep_part()); own.back()->build(gm, part_of, pl, rank * ppr + k, device, stream, mode); parts.push_back(own.back().get()); }
        const double lb0 = lockstep_lower_bound(parts, w);
        lockstep_compute_pass(parts, pl, w, n_parts, passes);
        const double lb1 = lockstep_lower_bound(parts, w);
        if (!out.empty())
          for (lockstep_part* p : parts) {
            const std::vector<double> d = p->download_duals();
            const std::string path = out + "." + std::to_string(p->part) + ".bin";
            FILE* f = std::fopen(path.c_str(), "wb");
            if (!f || std::fwrite(d.data(), sizeof(double), d.size(), f) != d.size()) throw std::runtime_error("cannot write " + path);
            std::fclose(f);
          }
        double ms_per_pass = 0;
        rccl_world::probe_result pr;
        if (timed > 0) {
          lockstep_prepare(parts, pl, n_parts, timed);
          (void)w.all_reduce_sum(0.0);
          const auto t0 = std::chrono::steady_clock::now();
          lockstep_compute_pass(parts, pl, w, n_parts, timed);
          (void)w.all_reduce_sum(0.0);
          ms_per_pass = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
          pr = w.probe_run([&] { lockstep_compute_pass(parts, pl, w, n_parts, timed); }, timed);
        }
        if (rank == 0) {
          std::printf("{\"driver\": \"mgpu_rccl_driver (C++ host, C ABI + RCCL)\", \"schedule\": \"lockstep\", \"model\": \"%s\", \"world\": %d, \"parts\": %d, \"factors\": %lld, \"messages\": %lld, "
                      "\"levels\": [%d, %d], \"exchanges_per_pass\": %.3f, \"passes\": %d, \"lower_bound_before\": %.17g, \"lower_bound_after\": %.17g",
                      model_path.c_str(), world, n_parts, (long long)gm.n_factors, (long long)gm.n_messages, pl.n_levels[0], pl.n_levels[1], pl.exchanges_per_pass(std::max(passes, 1)), passes, lb0, lb1);
          if (timed > 0) { std::printf(", \"ms_per_pass\": %.6f", ms_per_pass); print_probe(pr); }
          std::printf("}\n");
        }
        own.clear();
        w.destroy();
        (void)hipStreamDestroy(stream);
        return 0;
      }
      std::vector<int64_t> var_rank, part_of;
      if (graph_n > 0 && !order_file.empty()) var_rank = load(order_file, graph_n);
      if (graph_n > 0 && !part_file.empty()) part_of = load(part_file, graph_n);
      const lockstep_structure st = graph_n > 0 ? graph_structure(graph_n, graph_m, L, n_parts, 1, var_rank.empty() ? nullptr : &var_rank, part_of.empty() ? nullptr : &part_of)
                                                : strips_structure(H, W, L, potts, colour, n_parts, 1);
      lockstep_plan pl;
      pl.build(st, mode);
      std::vector<std::unique_ptr<lockstep_part>> own;
      std::vector<lockstep_part*> parts;
      for (int k = 0; k < ppr; ++k) {
        own.emplace_back(new lockstep_part());
        own.back()->build(st, pl, rank * ppr + k, device, stream, mode);
        parts.push_back(own.back().get());
      }
      const double lb0 = lockstep_lower_bound(parts, w);
      lockstep_compute_pass(parts, pl, w, n_parts, passes);
      const double lb1 = lockstep_lower_bound(parts, w);
      if (!out.empty())
        for (lockstep_part* p : parts) {
          const std::vector<double> d = p->download_duals();
          const std::string path = out + "." + std::to_string(p->part) + ".bin";
          FILE* f = std::fopen(path.c_str(), "wb");
          if (!f || std::fwrite(d.data(), sizeof(double), d.size(), f) != d.size()) throw std::runtime_error("cannot write " + path);
          std::fclose(f);
        }
      double ms_per_pass = 0;
      rccl_world::probe_result pr;
      if (timed > 0) {
        lockstep_prepare(parts, pl, n_parts, timed);
        (void)w.all_reduce_sum(0.0);
        const auto t0 = std::chrono::steady_clock::now();
        lockstep_compute_pass(parts, pl, w, n_parts, timed);
        (void)w.all_reduce_sum(0.0);
        ms_per_pass = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
        pr = w.probe_run([&] { lockstep_compute_pass(parts, pl, w, n_parts, timed); }, timed);
      }
      double upd = 0;
      for (lockstep_part* p : parts) upd += (double)p->updates_per_pass;
      upd = w.all_reduce_sum(upd);
      if (rank == 0) {
        std::printf("{\"driver\": \"mgpu_rccl_driver (C++ host, C ABI + RCCL)\", \"schedule\": \"lockstep\", \"world\": %d, \"parts\": %d, \"variables\": %lld, \"edges\": %lld, "
                    "\"labels\": %d, \"pairwise\": \"%s\", \"levels\": [%d, %d], \"exchanges_per_pass\": %.3f, \"passes\": %d, \"lower_bound_before\": %.17g, \"lower_bound_after\": %.17g",
                    world, n_parts, (long long)st.n_vars, (long long)st.n_edges(), L, potts ? "potts" : "dense", pl.n_levels[0], pl.n_levels[1], pl.exchanges_per_pass(std::max(passes, 1)), passes, lb0, lb1);
        if (timed > 0) { std::printf(", \"ms_per_pass\": %.6f, \"msg_updates_per_s\": %.6g", ms_per_pass, upd / (ms_per_pass * 1e-3)); print_probe(pr); }
        std::printf("}\n");
      }
      own.clear();
      w.destroy();
      (void)hipStreamDestroy(stream);
      return 0;
    }
    std::vector<std::unique_ptr<part_sweep>> own;
    std::vector<part_sweep*> parts;
    for (int k = 0; k < ppr; ++k) {
      own.emplace_back(new part_sweep());
      own.back()->build(strip_part(H, W, L, potts, colour, rank * ppr + k, n_parts, 1), device, stream, mode, every_pass);
      parts.push_back(own.back().get());
    }
    const double lb0 = lower_bound(parts, w);
    compute_pass(parts, w, passes, every_pass);
    const double lb1 = lower_bound(parts, w);
    if (!out.empty())
      for (part_sweep* p : parts) {
        const std::vector<double> d = p->download_duals();
        const std::string path = out + "." + std::to_string(p->pm.part) + ".bin";
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f || std::fwrite(d.data(), sizeof(double), d.size(), f) != d.size()) throw std::runtime_error("cannot write " + path);
        std::fclose(f);
      }
    double ms_per_pass = 0, updates = 0;
    rccl_world::probe_result pr;
    if (timed > 0) {
      double upd = 0;
      for (part_sweep* p : parts) upd += (double)p->updates_per_pass;
      updates = w.all_reduce_sum(upd);
      (void)w.all_reduce_sum(0.0);                                     // barrier
      const auto t0 = std::chrono::steady_clock::now();
      compute_pass(parts, w, timed, every_pass);
      (void)w.all_reduce_sum(0.0);                                     // every rank's stream drained + barrier
      ms_per_pass = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
      pr = w.probe_run([&] { compute_pass(parts, w, timed, every_pass); }, timed);
    }
    if (rank == 0) {
      std::printf("{\"driver\": \"mgpu_rccl_driver (C++ host, C ABI + RCCL)\", \"world\": %d, \"parts\": %d, \"grid_per_part\": [%d, %d], \"labels\": %d, "
                  "\"pairwise\": \"%s\", \"boundary_every\": \"%s\", \"passes\": %d, \"lower_bound_before\": %.17g, \"lower_bound_after\": %.17g",
                  world, n_parts, H, W, L, potts ? "potts" : "dense", every_pass ? "pass" : "sweep", passes, lb0, lb1);
      if (timed > 0) { std::printf(", \"ms_per_pass\": %.6f, \"msg_updates_per_s\": %.6g", ms_per_pass, updates / (ms_per_pass * 1e-3)); print_probe(pr); }
      std::printf("}\n");
    }
    own.clear();
    w.destroy();
    (void)hipStreamDestroy(stream);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "mgpu_rccl_driver (rank %d): %s\n", rank, ex.what());
    return 1;
  }
  return 0;
}
