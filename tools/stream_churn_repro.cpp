// stream_churn_repro.cpp — does HIP stream creation / destruction alone corrupt the host heap of a long-lived process?
//
// Context: test processes that created and destroyed tens of thousands of engines — each with two
// hipStreamCreateWithFlags / hipStreamDestroy pairs — showed host-heap corruption about once per 50 000 runs; pooling
// the streams made it disappear.  This program has NO engine code: per iteration it creates streams, runs the kind of
// work an engine issues on them (small kernels, a captured + replayed graph, pinned D2H copies of a few bytes, device
// allocations), destroys them (or, with --pool, keeps them), and verifies malloc'd canary blocks of many sizes.
//
//   hipcc --offload-arch=gfx950 -O2 -o build/stream_churn_repro tools/stream_churn_repro.cpp
//   build/stream_churn_repro SECONDS [--pool]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

__global__ void touch(double* p, int n, double a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0;
}
__global__ void reduce(const double* p, int n, double* out) {
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  atomicAdd(out, s);
}

struct Canary { unsigned char* p; size_t n; unsigned char tag; };

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 60.0;
  const bool pool = argc > 2 && !std::strcmp(argv[2], "--pool");
  static const size_t sizes[] = {48, 200, 1000, 4096, 20000, 70000, 150000, 600000};
  std::vector<Canary> can;
  for (int i = 0; i < 4000; ++i) {
    const size_t n = sizes[i % 8];
    Canary c{(unsigned char*)std::malloc(n), n, (unsigned char)((i * 2654435761u) & 0xFF)};
    std::memset(c.p, c.tag, n);
    can.push_back(c);
  }
  hipStream_t pooled[2] = {nullptr, nullptr};
  double* h_word = nullptr;
  CK(hipHostMalloc((void**)&h_word, 4096, hipHostMallocDefault));
  const auto t0 = std::chrono::steady_clock::now();
  long iters = 0, events = 0;
  unsigned rng = 12345;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipStream_t s[2];
    for (int k = 0; k < 2; ++k) {
      if (pool && pooled[k]) s[k] = pooled[k];
      else { CK(hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking)); if (pool) pooled[k] = s[k]; }
    }
    rng = rng * 1664525u + 1013904223u;
    const int n = 256 + (rng >> 20) % 4096;
    double *d = nullptr, *d_out = nullptr;
    CK(hipMalloc((void**)&d, n * sizeof(double)));
    CK(hipMalloc((void**)&d_out, sizeof(double)));
    CK(hipMemsetAsync(d, 0, n * sizeof(double), s[0]));
    CK(hipMemsetAsync(d_out, 0, sizeof(double), s[0]));
    // a captured chain of small launches, replayed (what run_schedule does for deep schedules)
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    CK(hipStreamBeginCapture(s[1], hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 12; ++k) hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, s[1], d, n, 0.5);
    CK(hipStreamEndCapture(s[1], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    CK(hipGraphLaunch(ge, s[0]));
    hipLaunchKernelGGL(reduce, dim3(1), dim3(256), 0, s[0], d, n, d_out);
    CK(hipMemcpyAsync(h_word, d_out, sizeof(double), hipMemcpyDeviceToHost, s[0]));
    CK(hipStreamSynchronize(s[0]));
    CK(hipGraphExecDestroy(ge));
    CK(hipFree(d)); CK(hipFree(d_out));
    if (!pool) for (int k = 0; k < 2; ++k) { CK(hipStreamSynchronize(s[k])); CK(hipStreamDestroy(s[k])); }
    // host-heap activity between iterations, like a test process: free / re-allocate a few canaries
    for (int k = 0; k < 50; ++k) {
      rng = rng * 1664525u + 1013904223u;
      Canary& c = can[(rng >> 8) % can.size()];
      for (size_t i = 0; i < c.n; ++i) if (c.p[i] != c.tag) { ++events; std::printf("CANARY size %zu offset %zu byte %02x (tag %02x) iteration %ld\n", c.n, i, c.p[i], c.tag, iters); break; }
      std::free(c.p); c.p = (unsigned char*)std::malloc(c.n); std::memset(c.p, c.tag, c.n);
    }
    if (++iters % 2000 == 0) {
      for (auto& c : can)
        for (size_t i = 0; i < c.n; ++i) if (c.p[i] != c.tag) { ++events; std::printf("CANARY size %zu offset %zu byte %02x (tag %02x) iteration %ld\n", c.n, i, c.p[i], c.tag, iters); std::memset(c.p, c.tag, c.n); break; }
      std::printf("  %ld iterations, %.0f s, %ld events\n", iters, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), events);
      std::fflush(stdout);
    }
  }
  for (auto& c : can)
    for (size_t i = 0; i < c.n; ++i) if (c.p[i] != c.tag) { ++events; std::printf("CANARY size %zu offset %zu at exit\n", c.n, i); break; }
  std::printf("done: %ld iterations (%s streams), %ld canary events\n", iters, pool ? "pooled" : "created and destroyed", events);
  return events ? 1 : 0;
}
