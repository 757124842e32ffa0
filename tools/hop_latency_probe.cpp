// tools/hop_latency_probe.cpp — what one dependent hop of the chain executor costs on this chip, piece by piece.
//   hipcc --offload-arch=gfx950 -O2 tools/hop_latency_probe.cpp -o /tmp/hop && /tmp/hop
// Two workgroups play ping-pong: producer writes a payload (256 threads x 8 B = 2 KiB, the message vectors of a few
// records), waits for its stores (s_waitcnt vmcnt(0)), barrier, one lane publishes a flag; the consumer polls the flag,
// barrier, loads the payload, checks it, and answers the same way.  One round trip = 2 hops.
// Variants: partner on another XCD (block 1) or on the same XCD (block 8: workgroups go round-robin over the 8 XCDs);
// payload loads / stores at agent scope (sc1: past the L2, what the chain executor does) or at workgroup scope (plain:
// through the XCD's L2 — only coherent when producer and consumer share the XCD; the check counts stale reads);
// flag only (no payload).  And the same exchange between two WAVES of one workgroup through global memory and LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum Scope { AGENT = 0, L2 = 1 };

template <int S> __device__ __forceinline__ double ld(const double* p) {
  if constexpr (S == AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int S> __device__ __forceinline__ void st(double* p, double v) {
  // stores always go through to memory (another XCD may read them later); S only changes the loads
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain() { __builtin_amdgcn_s_waitcnt(0); }   // vmcnt(0) expcnt(0) lgkmcnt(0)

// out[0] = cycles (s_memrealtime, 100 MHz) of `iters` round trips seen by block 0, out[1] = stale payload reads,
// out[2] = XCC_ID of block 0, out[3] = XCC_ID of the partner
template <int S, bool PAYLOAD>
__global__ void __launch_bounds__(256) pingpong(int* flags, double* data, int iters, int partner, long long* out) {
  const int b = blockIdx.x;
  if (b != 0 && b != partner) return;
  const int me = b == 0 ? 0 : 1;
  int* my_flag = flags + 64 * me;
  int* other_flag = flags + 64 * (1 - me);
  double* my_data = data + 4096 * me;
  double* other_data = data + 4096 * (1 - me);
  const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;   // XCC_ID[3:0]
  if (threadIdx.x == 0) out[2 + me] = xcc;
  long long stale = 0;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 1; it <= iters; ++it) {
    if (me == 0) {
      if (PAYLOAD) { st<S>(my_data + threadIdx.x, (double)(it * 1000 + threadIdx.x)); drain(); }
      __syncthreads();
      if (threadIdx.x == 0) st_flag(my_flag, it);
    }
    if (threadIdx.x == 0) while (ld_flag(other_flag) != it) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (PAYLOAD) { const double v = ld<S>(other_data + threadIdx.x); if (v != (double)(it * 1000 + threadIdx.x)) ++stale; }
    if (me == 1) {
      if (PAYLOAD) { st<S>(my_data + threadIdx.x, (double)(it * 1000 + threadIdx.x)); drain(); }
      __syncthreads();
      if (threadIdx.x == 0) st_flag(my_flag, it);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memrealtime();
  if (me == 0 && threadIdx.x == 0) out[0] = t1 - t0;
  if (stale) atomicAdd((unsigned long long*)&out[1], (unsigned long long)stale);
}

// two waves of ONE workgroup: wave 0 writes 64 doubles, wave 1 reads them and answers
// MODE 0: global memory, agent-scope loads; 1: global memory, workgroup-scope loads (L2 / L1 path); 2: LDS
template <int MODE>
__global__ void __launch_bounds__(128) intra_wg(double* data, int iters, long long* out) {
  __shared__ double lds[2][64];
  __shared__ int lflag[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 2) lflag[threadIdx.x] = 0;
  __syncthreads();
  long long stale = 0;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 1; it <= iters; ++it) {
    // step A: wave 0 produces, barrier, wave 1 consumes; step B: the other way round
    for (int step = 0; step < 2; ++step) {
      const int prod = step;
      if (wave == prod) {
        const double v = (double)(it * 1000 + lane + step);
        if (MODE == 2) lds[prod][lane] = v;
        else { st<AGENT>(data + 64 * prod + lane, v); drain(); }
      }
      __syncthreads();
      if (wave != prod) {
        double v;
        if (MODE == 2) v = lds[prod][lane];
        else if (MODE == 1) v = ld<L2>(data + 64 * prod + lane);
        else v = ld<AGENT>(data + 64 * prod + lane);
        if (v != (double)(it * 1000 + lane + step)) ++stale;
      }
      __syncthreads();
    }
  }
  const long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if (stale) atomicAdd((unsigned long long*)&out[1], (unsigned long long)stale);
}

int main() {
  int* flags; double* data; long long* out;
  CHECK(hipMalloc((void**)&flags, 4096));
  CHECK(hipMalloc((void**)&data, 2 * 4096 * sizeof(double)));
  CHECK(hipMalloc((void**)&out, 8 * sizeof(long long)));
  const int iters = 20000;
  auto run = [&](const char* name, auto kernel, int partner) {
    CHECK(hipMemset(flags, 0, 4096)); CHECK(hipMemset(out, 0, 64)); CHECK(hipMemset(data, 0, 2 * 4096 * sizeof(double)));
    hipLaunchKernelGGL(kernel, dim3(16), dim3(256), 0, 0, flags, data, iters, partner, out);
    CHECK(hipDeviceSynchronize());
    long long h[8]; CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("%-58s XCC %lld <-> %lld: %7.3f us per hop, stale reads %lld\n", name, h[2], h[3], h[0] * 0.01 / (2.0 * iters), h[1]);
  };
  for (int partner : {1, 8}) {
    run("flag only", pingpong<AGENT, false>, partner);
    run("2 KiB payload, agent-scope loads (chain executor)", pingpong<AGENT, true>, partner);
    run("2 KiB payload, workgroup-scope loads (through the L2)", pingpong<L2, true>, partner);
  }
  auto run2 = [&](const char* name, auto kernel) {
    CHECK(hipMemset(out, 0, 64)); CHECK(hipMemset(data, 0, 2 * 4096 * sizeof(double)));
    hipLaunchKernelGGL(kernel, dim3(1), dim3(128), 0, 0, data, iters, out);
    CHECK(hipDeviceSynchronize());
    long long h[8]; CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("%-58s %7.3f us per step, stale reads %lld\n", name, h[0] * 0.01 / (2.0 * iters), h[1]);
  };
  run2("one workgroup, wave to wave: global, agent-scope loads", intra_wg<0>);
  run2("one workgroup, wave to wave: global, workgroup-scope loads", intra_wg<1>);
  run2("one workgroup, wave to wave: LDS", intra_wg<2>);
  return 0;
}
