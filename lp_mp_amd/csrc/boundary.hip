// boundary.hip — the boundary step of the partitioned (multi-GPU) sweep as device kernels behind C entry points that
// take DEVICE pointers: a C++ host drives RCCL itself (ncclSend / ncclRecv or an all-to-all on the same HIP stream)
// and calls these in between; lp_mp_amd/multi_gpu.py does the same through torch.distributed.
//
// One boundary step (every part of it is the reference's UpdateFactor of the non-owner endpoint of a cut
// message, restricted to its cut messages, include/factors_messages.hxx:2256-2261):
//   owner      ghost <- min-marginal toward the remote variable      lpmp_schedule_run(ghost receive schedule)
//   owner      pack: send[...] = ghost vectors, ghosts zeroed         lpmp_boundary_pack        ---- exchange #1 ---->
//   non-owner  theta += received message (message-list order), reply = omega_b * theta (state after all receives),
//              theta -= reply (same order)                            lpmp_boundary_reply       <--- exchange #2 -----
//   owner      ghost <- reply; a weight-1 send folds it into the cut edge's pairwise factor
//                                                                     lpmp_boundary_fold + lpmp_schedule_run(ghost send schedule)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lpmp_engine.h"

namespace lpmp {

// one vector of the exchange: `len` doubles at dual[dual_off ...] <-> buffer[buf_off ...]
struct BVec { int64_t dual_off; int64_t buf_off; int32_t len; int32_t pad; };
// one boundary variable (non-owner endpoint): its cut messages in message-list order = entries [first, first + n) of seq
struct BVar { int64_t dual_off; int32_t len; int32_t first; int32_t n; int32_t pad; };
struct BMsg { int64_t buf_off; double omega; };

__global__ void __launch_bounds__(256)
boundary_pack_kernel(const BVec* __restrict__ v, int64_t n, double* __restrict__ dual, double* __restrict__ send) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const BVec b = v[i];
  for (int x = lane; x < b.len; x += 64) { send[b.buf_off + x] = dual[b.dual_off + x]; dual[b.dual_off + x] = 0.0; }
}
__global__ void __launch_bounds__(256)
boundary_fold_kernel(const BVec* __restrict__ v, int64_t n, double* __restrict__ dual, const double* __restrict__ back) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const BVec b = v[i];
  for (int x = lane; x < b.len; x += 64) dual[b.dual_off + x] = back[b.buf_off + x];
}
// lock-step halos (lp_mp_amd/lockstep.py): plain copies of message vectors, G lanes per vector (16 for the short vectors of a
// 16-label model: four vectors per wave instead of one)
template <int G>
__global__ void __launch_bounds__(256)
halo_copy_kernel(const BVec* __restrict__ v, int64_t n, double* __restrict__ dual, double* __restrict__ buf, int to_dual) {
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const int64_t i = (int64_t)blockIdx.x * (256 / G) + grp;
  if (i >= n) return;
  const BVec b = v[i];
  if (to_dual) { for (int x = lane; x < b.len; x += G) dual[b.dual_off + x] = buf[b.buf_off + x]; }
  else { for (int x = lane; x < b.len; x += G) buf[b.buf_off + x] = dual[b.dual_off + x]; }
}
// per boundary variable and label: theta += r_1; theta += r_2; ...; snapshot; reply_k = omega_k * snapshot; theta -= reply_1; ...
__global__ void __launch_bounds__(256)
boundary_reply_kernel(const BVar* __restrict__ vars, const BMsg* __restrict__ seq, int64_t n, double* __restrict__ dual,
                      const double* __restrict__ recv, double* __restrict__ reply) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const BVar u = vars[i];
  for (int x = lane; x < u.len; x += 64) {
    double th = dual[u.dual_off + x];
    for (int k = 0; k < u.n; ++k) th += recv[seq[u.first + k].buf_off + x];
    const double snap = th;
    for (int k = 0; k < u.n; ++k) {
      const BMsg m = seq[u.first + k];
      const double r = m.omega * snap;
      reply[m.buf_off + x] = r;
      th -= r;
    }
    dual[u.dual_off + x] = th;
  }
}

}  // namespace lpmp

using namespace lpmp;

struct lpmp_boundary {
  BVec* d_out = nullptr; int64_t n_out = 0, out_doubles = 0;
  BVar* d_vars = nullptr; BMsg* d_seq = nullptr; int64_t n_vars = 0, in_doubles = 0;
  ~lpmp_boundary() {
    if (d_out) (void)hipFree(d_out);
    if (d_vars) (void)hipFree(d_vars);
    if (d_seq) (void)hipFree(d_seq);
  }
};

struct lpmp_halo {
  BVec* d_vec[2] = {nullptr, nullptr};      // 0: what pack reads, 1: what unpack writes
  int64_t n[2] = {0, 0}, doubles[2] = {0, 0};
  int32_t max_len[2] = {0, 0};
  ~lpmp_halo() { for (BVec* p : d_vec) if (p) (void)hipFree(p); }
};

extern "C" {
void* lpmp_engine_stream(lpmp_engine* e);      // engine.cpp
int lpmp_set_last_error(const char* msg);      // engine.cpp
int lpmp_boundary_enter(lpmp_engine* e);       // engine.cpp: the engine's device current, speculative passes settled, no aborted chain run behind
void* lpmp_engine_dual_base(lpmp_engine* e);   // engine.cpp: the dual base pointer the device offsets are relative to
int64_t lpmp_engine_device_dual_offset(lpmp_engine* e, int64_t packed_off);   // engine.cpp: a packed dual offset as a device offset (rows layout)
int lpmp_boundary_leave(lpmp_engine* e);       // engine.cpp: duals were written through device offsets
int lpmp_engine_dual_range_ok(lpmp_engine* e, int64_t packed_off, int64_t len);   // engine.cpp: a run of doubles inside one factor's dual

#define B_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { lpmp_set_last_error((std::string(#x) + ": " + hipGetErrorString(e_)).c_str()); return LPMP_ERR_DEVICE; } } while (0)

// no exception crosses the C boundary (std::vector may throw), nothing leaks on an error path (the handle owns its arrays)
static int boundary_create(lpmp_engine* e, int64_t n_out, const int64_t* out_dual_off, const int32_t* out_len, int64_t n_in,
                           const int64_t* in_dual_off, const int32_t* in_len, const double* in_omega, const int64_t* in_order,
                           lpmp_boundary** out) {
  std::unique_ptr<lpmp_boundary> b(new lpmp_boundary());
  std::vector<BVec> ov((size_t)n_out);
  int64_t at = 0;
  for (int64_t i = 0; i < n_out; ++i) {
    if (!lpmp_engine_dual_range_ok(e, out_dual_off[i], out_len[i])) { lpmp_set_last_error("boundary: an outgoing vector is not inside one factor's dual"); return LPMP_ERR_INVALID; }
    ov[i] = {lpmp_engine_device_dual_offset(e, out_dual_off[i]), at, out_len[i], 0}; at += out_len[i];
  }
  for (int64_t i = 0; i < n_in; ++i)
    if (!lpmp_engine_dual_range_ok(e, in_dual_off[i], in_len[i])) { lpmp_set_last_error("boundary: an incoming vector is not inside one factor's dual"); return LPMP_ERR_INVALID; }
  b->n_out = n_out; b->out_doubles = at;
  // incoming messages arrive in exchange order (buffer offsets by prefix sum); in_order lists them grouped by variable,
  // inside a variable in the order its message list holds them
  std::vector<int64_t> in_buf((size_t)n_in + 1, 0);
  for (int64_t i = 0; i < n_in; ++i) in_buf[i + 1] = in_buf[i] + in_len[i];
  b->in_doubles = in_buf[n_in];
  std::vector<BVar> vars; std::vector<BMsg> seq((size_t)n_in);
  for (int64_t k = 0; k < n_in; ++k) {
    const int64_t m = in_order[k];
    if (m < 0 || m >= n_in) { lpmp_set_last_error("boundary: in_order out of range"); return LPMP_ERR_INVALID; }
    seq[k] = {in_buf[m], in_omega[m]};
    const int64_t dev_off = lpmp_engine_device_dual_offset(e, in_dual_off[m]);
    if (vars.empty() || vars.back().dual_off != dev_off) vars.push_back({dev_off, in_len[m], (int32_t)k, 0, 0});
    if (vars.back().len != in_len[m]) { lpmp_set_last_error("boundary: messages of one variable differ in length"); return LPMP_ERR_INVALID; }
    vars.back().n++;
  }
  b->n_vars = (int64_t)vars.size();
  hipStream_t s = (hipStream_t)lpmp_engine_stream(e);
  if (n_out > 0) { B_TRY(hipMalloc((void**)&b->d_out, ov.size() * sizeof(BVec))); B_TRY(hipMemcpyAsync(b->d_out, ov.data(), ov.size() * sizeof(BVec), hipMemcpyHostToDevice, s)); }
  if (n_in > 0) {
    B_TRY(hipMalloc((void**)&b->d_vars, vars.size() * sizeof(BVar))); B_TRY(hipMemcpyAsync(b->d_vars, vars.data(), vars.size() * sizeof(BVar), hipMemcpyHostToDevice, s));
    B_TRY(hipMalloc((void**)&b->d_seq, seq.size() * sizeof(BMsg))); B_TRY(hipMemcpyAsync(b->d_seq, seq.data(), seq.size() * sizeof(BMsg), hipMemcpyHostToDevice, s));
  }
  B_TRY(hipStreamSynchronize(s));
  *out = b.release();
  return LPMP_OK;
}
int lpmp_boundary_create(lpmp_engine* e, int64_t n_out, const int64_t* out_dual_off, const int32_t* out_len, int64_t n_in,
                         const int64_t* in_dual_off, const int32_t* in_len, const double* in_omega, const int64_t* in_order,
                         lpmp_boundary** out) {
  if (!e || !out || n_out < 0 || n_in < 0 || (n_out > 0 && (!out_dual_off || !out_len)) ||
      (n_in > 0 && (!in_dual_off || !in_len || !in_omega || !in_order))) { lpmp_set_last_error("bad argument"); return LPMP_ERR_INVALID; }
  if (const int rc = lpmp_boundary_enter(e)) return rc;
  try { return boundary_create(e, n_out, out_dual_off, out_len, n_in, in_dual_off, in_len, in_omega, in_order, out); }
  catch (const std::bad_alloc&) { lpmp_set_last_error("out of host memory"); return LPMP_ERR_INVALID; }
  catch (const std::exception& ex) { lpmp_set_last_error(ex.what()); return LPMP_ERR_INVALID; }
}
void lpmp_boundary_destroy(lpmp_boundary* b) { delete b; }
int64_t lpmp_boundary_out_doubles(const lpmp_boundary* b) { return b ? b->out_doubles : 0; }
int64_t lpmp_boundary_in_doubles(const lpmp_boundary* b) { return b ? b->in_doubles : 0; }

int lpmp_boundary_pack(lpmp_engine* e, lpmp_boundary* b, double* send_dev) {
  if (!e || !b || (b->n_out > 0 && !send_dev)) { lpmp_set_last_error("bad argument"); return LPMP_ERR_INVALID; }
  if (const int rc = lpmp_boundary_enter(e)) return rc;
  if (b->n_out > 0) hipLaunchKernelGGL(boundary_pack_kernel, dim3((unsigned)((b->n_out + 3) / 4)), dim3(256), 0, (hipStream_t)lpmp_engine_stream(e),
                                       b->d_out, b->n_out, (double*)lpmp_engine_dual_base(e), send_dev);
  B_TRY(hipGetLastError());
  return lpmp_boundary_leave(e);
}
int lpmp_boundary_reply(lpmp_engine* e, lpmp_boundary* b, const double* recv_dev, double* reply_dev) {
  if (!e || !b || (b->n_vars > 0 && (!recv_dev || !reply_dev))) { lpmp_set_last_error("bad argument"); return LPMP_ERR_INVALID; }
  if (const int rc = lpmp_boundary_enter(e)) return rc;
  if (b->n_vars > 0) hipLaunchKernelGGL(boundary_reply_kernel, dim3((unsigned)((b->n_vars + 3) / 4)), dim3(256), 0, (hipStream_t)lpmp_engine_stream(e),
                                        b->d_vars, b->d_seq, b->n_vars, (double*)lpmp_engine_dual_base(e), recv_dev, reply_dev);
  B_TRY(hipGetLastError());
  return lpmp_boundary_leave(e);
}
int lpmp_boundary_fold(lpmp_engine* e, lpmp_boundary* b, const double* back_dev) {
  if (!e || !b || (b->n_out > 0 && !back_dev)) { lpmp_set_last_error("bad argument"); return LPMP_ERR_INVALID; }
  if (const int rc = lpmp_boundary_enter(e)) return rc;
  if (b->n_out > 0) hipLaunchKernelGGL(boundary_fold_kernel, dim3((unsigned)((b->n_out + 3) / 4)), dim3(256), 0, (hipStream_t)lpmp_engine_stream(e),
                                       b->d_out, b->n_out, (double*)lpmp_engine_dual_base(e), back_dev);
  B_TRY(hipGetLastError());
  return lpmp_boundary_leave(e);
}

// ---- lock-step halos: the vectors one exchange ships, as (packed dual offset, length) lists in exchange order
static int halo_create(lpmp_engine* e, const int64_t n[2], const int64_t* const off[2], const int32_t* const len[2], lpmp_halo** out) {
  std::unique_ptr<lpmp_halo> h(new lpmp_halo());
  hipStream_t s = (hipStream_t)lpmp_engine_stream(e);
  for (int k = 0; k < 2; ++k) {
    std::vector<BVec> v((size_t)n[k]);
    int64_t at = 0;
    for (int64_t i = 0; i < n[k]; ++i) {
      if (!lpmp_engine_dual_range_ok(e, off[k][i], len[k][i])) {
        lpmp_set_last_error(k == 0 ? "halo: an outgoing vector is not a run of doubles inside one factor's dual" : "halo: an incoming vector is not a run of doubles inside one factor's dual");
        return LPMP_ERR_INVALID;
      }
      v[(size_t)i] = {lpmp_engine_device_dual_offset(e, off[k][i]), at, len[k][i], 0};
      at += len[k][i]; h->max_len[k] = std::max(h->max_len[k], len[k][i]);
    }
    h->n[k] = n[k]; h->doubles[k] = at;
    if (n[k] > 0) { B_TRY(hipMalloc((void**)&h->d_vec[k], v.size() * sizeof(BVec))); B_TRY(hipMemcpyAsync(h->d_vec[k], v.data(), v.size() * sizeof(BVec), hipMemcpyHostToDevice, s)); }
    B_TRY(hipStreamSynchronize(s));      // (v leaves scope)
  }
  *out = h.release();
  return LPMP_OK;
}
int lpmp_halo_create(lpmp_engine* e, int64_t n_out, const int64_t* out_dual_off, const int32_t* out_len, int64_t n_in,
                     const int64_t* in_dual_off, const int32_t* in_len, lpmp_halo** out) {
  if (!e || !out || n_out < 0 || n_in < 0 || (n_out > 0 && (!out_dual_off || !out_len)) || (n_in > 0 && (!in_dual_off || !in_len))) {
    lpmp_set_last_error("bad argument"); return LPMP_ERR_INVALID;
  }
  if (const int rc = lpmp_boundary_enter(e)) return rc;
  const int64_t n[2] = {n_out, n_in};
  const int64_t* const off[2] = {out_dual_off, in_dual_off};
  const int32_t* const len[2] = {out_len, in_len};
  try { return halo_create(e, n, off, len, out); }
  catch (const std::bad_alloc&) { lpmp_set_last_error("out of host memory"); return LPMP_ERR_INVALID; }
  catch (const std::exception& ex) { lpmp_set_last_error(ex.what()); return LPMP_ERR_INVALID; }
}
void lpmp_halo_destroy(lpmp_halo* h) { delete h; }
int64_t lpmp_halo_out_doubles(const lpmp_halo* h) { return h ? h->doubles[0] : 0; }
int64_t lpmp_halo_in_doubles(const lpmp_halo* h) { return h ? h->doubles[1] : 0; }
static int halo_copy(lpmp_engine* e, lpmp_halo* h, int k, double* buf) {
  if (!e || !h || (h->n[k] > 0 && !buf)) { lpmp_set_last_error("bad argument"); return LPMP_ERR_INVALID; }
  if (const int rc = lpmp_boundary_enter(e)) return rc;
  if (h->n[k] > 0) {
    hipStream_t s = (hipStream_t)lpmp_engine_stream(e);
    double* dual = (double*)lpmp_engine_dual_base(e);
    if (h->max_len[k] <= 16) hipLaunchKernelGGL(halo_copy_kernel<16>, dim3((unsigned)((h->n[k] + 15) / 16)), dim3(256), 0, s, h->d_vec[k], h->n[k], dual, buf, k);
    else hipLaunchKernelGGL(halo_copy_kernel<64>, dim3((unsigned)((h->n[k] + 3) / 4)), dim3(256), 0, s, h->d_vec[k], h->n[k], dual, buf, k);
  }
  B_TRY(hipGetLastError());
  return k == 1 ? lpmp_boundary_leave(e) : LPMP_OK;      // only unpack writes duals
}
int lpmp_halo_pack(lpmp_engine* e, lpmp_halo* h, double* send_dev) { return halo_copy(e, h, 0, send_dev); }
int lpmp_halo_unpack(lpmp_engine* e, lpmp_halo* h, const double* recv_dev) { return halo_copy(e, h, 1, const_cast<double*>(recv_dev)); }

// out[b * block_len + i] = u01(seed, first[b] + i): the cost blocks of a scattered subset of a global stream (a rank's
// own pairwise tables of a partitioned synthetic model), generated in HBM
}  // extern "C"

namespace lpmp {
__global__ void synth_fill_blocks_kernel(double* __restrict__ out, int64_t n_blocks, int64_t block_len, uint64_t seed, const int64_t* __restrict__ first) {
  const int64_t total = n_blocks * block_len;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
    const int64_t b = j / block_len, i = j - b * block_len;
    uint64_t z = seed + ((uint64_t)first[b] + (uint64_t)i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    out[j] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
  }
}
}  // namespace lpmp

extern "C" int lpmp_synth_fill_blocks(void* device_ptr, int64_t n_blocks, int64_t block_len, uint64_t seed, const int64_t* first_dev, void* hip_stream) {
  if ((!device_ptr || !first_dev) && n_blocks > 0) { lpmp_set_last_error("null argument"); return LPMP_ERR_INVALID; }
  if (n_blocks <= 0 || block_len <= 0) return LPMP_OK;
  int64_t blocks = (n_blocks * block_len + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(lpmp::synth_fill_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, (double*)device_ptr, n_blocks, block_len, seed, first_dev);
  B_TRY(hipGetLastError());
  return LPMP_OK;
}
