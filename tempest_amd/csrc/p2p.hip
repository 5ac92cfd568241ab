// Small-message collectives over peer-mapped device memory (xGMI between the GPUs of one node): tph_comm_p2p_*.
//
// The sharded sampler issues dozens of tiny collectives per iteration -- the (max, s1, s2) triples of every reweight
// evaluation, the acceptance sums of EVERY MCMC step, block totals, moments -- 16 B to a few KB each.  Through a framework's
// process group each costs a host call, two cross-stream event waits and a collective kernel (>= 20 us; the step itself is
// 100 us at config 4's shard size).  xGMI is point to point and every GPU of the node can map every other GPU's memory, so
// for these sizes the exchange is ONE single-block kernel on the ctx stream: each rank stores its values into slot [rank] of
// every peer's inbox, raises a sequence flag behind a system-scope release, waits for the world's flags in its own inbox and
// reduces the slots in rank order (every rank forms bit-identical sums).  No host call, no second stream, no library: the
// kernel takes its sequence number from device memory, so a launch has no per-call arguments and can be part of a captured
// step graph.  Larger messages keep going through the attached callbacks.
//
// Inbox of rank r (uncached device memory, mapped by every peer through a HIP IPC handle):
//   data  [ring 2][source world][TPH_P2P_SLOT bytes]        flag [ring 2][source world] x 128 B (one 8-byte sequence word each)
// Ring depth 2 is enough: a rank cannot start exchange k+2 before every peer has finished reading exchange k (it needs the
// peers' flags of k+1, which they raise only after completing k).
#include "common.h"
#include "p2p.h"

#include <stdlib.h>
#include <string.h>
#include <vector>

struct tph_p2p {
  p2p_args a{};
  char* inbox_local = nullptr;
  bool opened[TPH_P2P_MAX] = {};
  unsigned int* err_host = nullptr;
  bool ready = false;
  // row window of the one-sided resample shuffle (tph_resample_put_global): [n_local][2 d + 2] doubles on every rank
  char* win_local = nullptr;
  size_t win_bytes = 0;
  char* win[TPH_P2P_MAX] = {};
  bool win_opened[TPH_P2P_MAX] = {};
  std::vector<void*> retired;                   // outgrown windows: freed at release (a peer may still hold a mapping)
  bool win_failed = false;                      // the ranks agreed that the windows cannot be mapped: callers use their all-to-all
  unsigned long long put_seq = 0;
};

template <typename T>
__global__ void __launch_bounds__(256) k_p2p(p2p_args a, const T* src, T* dst, int count, int op) {
  (void)p2p_block_exchange(a, src, dst, count, op);
}

// self-test patterns (tph_comm_p2p_attach)
__device__ __forceinline__ double p2p_pattern(int rank, int round, int i) { return (double)((rank + 1) * 4096 + round * 7) + 0.25 * i; }
__global__ void k_p2p_fill(double* __restrict__ src, int n, int rank, int round) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) src[i] = p2p_pattern(rank, round, i);
}
__global__ void k_p2p_check(const double* __restrict__ sum, const double* __restrict__ all, int n, int G, int round, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double want = 0.0;
  bool ok = true;
  for (int r = 0; r < G; ++r) {
    const double v = p2p_pattern(r, round, i);
    ok = ok && all[(size_t)r * n + i] == v;
    want += v;                                   // rank order, as the exchange adds
  }
  if (!ok || sum[i] != want) atomicOr(bad, 1);
}

static size_t p2p_inbox_bytes(int world) { return 2 * (size_t)world * (TPH_P2P_SLOT + TPH_P2P_FLAG); }

static void p2p_release(tph_ctx* ctx) {
  tph_p2p* p = ctx->p2p;
  if (!p) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);      // after this rank's last exchange nobody writes into its inbox any more
  for (int r = 0; r < TPH_P2P_MAX; ++r) {
    if (p->opened[r]) (void)hipIpcCloseMemHandle(p->a.inbox[r]);
    if (p->win_opened[r]) (void)hipIpcCloseMemHandle(p->win[r]);
  }
  if (p->win_local) (void)hipFree(p->win_local);
  for (void* w : p->retired) (void)hipFree(w);
  if (p->inbox_local) (void)hipFree(p->inbox_local);
  if (p->a.seq) (void)hipFree(p->a.seq);
  if (p->err_host) (void)hipHostFree(p->err_host);
  delete p;
  ctx->p2p = nullptr;
}
void tph_p2p_release(tph_ctx* ctx) { p2p_release(ctx); }
// text of the sticky error word: 1 + rank that never raised its flag, or TPH_P2P_ERR_TAG (a shuffled row with a foreign tag)
constexpr unsigned int TPH_P2P_ERR_TAG = 1000u;
static int p2p_fail(unsigned int code) {
  if (code == TPH_P2P_ERR_TAG)
    tph_set_error("one-sided resample shuffle: a slot of this rank's window was not written for this exchange (window tag mismatch: "
                  "a peer selected different rows or ran a different sequence of shuffles)");
  else
    tph_set_error("a peer-to-peer exchange timed out: rank %u never raised its flag (a peer died, ran a different sequence of "
                  "collectives, or is more than TEMPEST_AMD_P2P_TIMEOUT seconds behind)", code - 1);
  return -2;
}


bool tph_p2p_fits(const tph_ctx* ctx, int64_t count, int dtype) {
  return ctx->p2p && ctx->p2p->ready && count > 0 && (size_t)count * (dtype == TPH_DT_I32 ? 4 : 8) <= TPH_P2P_SLOT;
}

const p2p_args* tph_p2p_ready(tph_ctx* ctx, int64_t count, int dtype) {
  if (!tph_p2p_fits(ctx, count, dtype)) return nullptr;
  if (*ctx->p2p->err_host != 0) {
    (void)p2p_fail(*ctx->p2p->err_host);
    return nullptr;
  }
  return &ctx->p2p->a;
}

// one exchange on the ctx stream (src/dst: device pointers; op < 0 = all-gather)
int tph_p2p_exchange(tph_ctx* ctx, const void* src, void* dst, int64_t count, int dtype, int op) {
  tph_p2p* p = ctx->p2p;
  TPH_REQUIRE(p && p->ready, "peer-to-peer collectives are not attached");
  if (*p->err_host != 0) return p2p_fail(*p->err_host);
  ctx->stat[0] += 1;
  switch (dtype) {
    case TPH_DT_F64: hipLaunchKernelGGL(k_p2p<double>, dim3(1), dim3(256), 0, ctx->stream, p->a, (const double*)src, (double*)dst, (int)count, op); break;
    case TPH_DT_I64: hipLaunchKernelGGL(k_p2p<long long>, dim3(1), dim3(256), 0, ctx->stream, p->a, (const long long*)src, (long long*)dst, (int)count, op); break;
    case TPH_DT_I32: hipLaunchKernelGGL(k_p2p<int>, dim3(1), dim3(256), 0, ctx->stream, p->a, (const int*)src, (int*)dst, (int)count, op); break;
    default: TPH_REQUIRE(false, "peer-to-peer exchange: dtype %d", dtype);
  }
  TPH_LAUNCH_CHECK();
  return 0;
}

extern "C" int tph_comm_p2p_export(tph_ctx* ctx, void* handle_out) {
  TPH_REQUIRE(ctx && handle_out, "tph_comm_p2p_export: NULL argument");
  TPH_REQUIRE(ctx->comm_active(), "tph_comm_p2p_export: attach the communicator first (tph_comm_attach)");
  TPH_REQUIRE(ctx->world <= TPH_P2P_MAX, "tph_comm_p2p_export: at most %d ranks (one node), world is %d", TPH_P2P_MAX, ctx->world);
  TPH_HIP(hipSetDevice(ctx->device));
  p2p_release(ctx);
  tph_p2p* p = new tph_p2p();
  ctx->p2p = p;
  const size_t bytes = p2p_inbox_bytes(ctx->world);
  // uncached (fine-grained) so that a peer's stores are seen by a kernel that is already running here
  if (hipExtMallocWithFlags((void**)&p->inbox_local, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    p2p_release(ctx);
    TPH_REQUIRE(false, "tph_comm_p2p_export: no uncached device memory for the inbox (%zu B)", bytes);
  }
  if (hipMemsetAsync(p->inbox_local, 0, bytes, ctx->stream) != hipSuccess || hipMalloc((void**)&p->a.seq, 64) != hipSuccess ||
      hipMemsetAsync(p->a.seq, 0, 64, ctx->stream) != hipSuccess ||
      hipHostMalloc((void**)&p->err_host, 64, hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    p2p_release(ctx);
    TPH_REQUIRE(false, "tph_comm_p2p_export: allocation failed");
  }
  *p->err_host = 0;
  if (hipHostGetDevicePointer((void**)&p->a.err, p->err_host, 0) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
    (void)hipGetLastError();
    p2p_release(ctx);
    TPH_REQUIRE(false, "tph_comm_p2p_export: pinned error word");
  }
  hipIpcMemHandle_t h;
  static_assert(sizeof(hipIpcMemHandle_t) == TPH_P2P_HANDLE_BYTES, "IPC handle size");
  if (ctx->world > 1) {
    const hipError_t e = hipIpcGetMemHandle(&h, p->inbox_local);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      p2p_release(ctx);
      TPH_REQUIRE(false, "tph_comm_p2p_export: hipIpcGetMemHandle -> %s", hipGetErrorString(e));
    }
  } else {
    memset(&h, 0, sizeof(h));
  }
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

extern "C" int tph_comm_p2p_attach(tph_ctx* ctx, const void* handles, int* ok_out) {
  TPH_REQUIRE(ctx && handles && ok_out, "tph_comm_p2p_attach: NULL argument");
  tph_p2p* p = ctx->p2p;
  TPH_REQUIRE(p && p->inbox_local, "tph_comm_p2p_attach: call tph_comm_p2p_export first");
  TPH_HIP(hipSetDevice(ctx->device));
  *ok_out = 0;
  const int G = ctx->world;
  p->a.world = G; p->a.rank = ctx->rank;
  // how far apart two ranks may arrive at an exchange before it is declared dead.  A step of a sharded run may legitimately take
  // long on one rank (first-use compilation of a callback, a shared GPU): TEMPEST_AMD_P2P_TIMEOUT sets the bound explicitly,
  // else a longer TEMPEST_AMD_STEP_TIMEOUT (the host's own patience with a step, mcmc.py: wait_record) extends the default
  double secs = 120.0;
  if (const char* env = getenv("TEMPEST_AMD_STEP_TIMEOUT")) secs = atof(env) > secs ? atof(env) : secs;
  if (const char* env = getenv("TEMPEST_AMD_P2P_TIMEOUT")) secs = atof(env) > 0 ? atof(env) : secs;
  p->a.timeout = (unsigned long long)(secs * 1e8);
  int ok = 1;
  for (int r = 0; r < G; ++r) {
    if (r == ctx->rank) { p->a.inbox[r] = p->inbox_local; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)r * sizeof(h), sizeof(h));
    void* ptr = nullptr;
    if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      ok = 0;
      p->a.inbox[r] = p->inbox_local;           // keeps the self-test's stores inside mapped memory
      continue;
    }
    p->a.inbox[r] = (char*)ptr;
    p->opened[r] = true;
  }
  // every rank now agrees (through the attached all-reduce) whether all mappings exist, then proves the exchange itself:
  // an all-gather of the rank numbers and an all-reduce, checked on the host
  TPH_REQUIRE(ctx->comm_bytes >= 4096, "tph_comm_p2p_attach: staging block too small");
  int* flag = (int*)ctx->comm_buf;
  TPH_HIP(hipMemcpyAsync(flag, &ok, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (tph_comm_allreduce(ctx, 0, 1, TPH_DT_I32, TPH_OP_MIN)) return -2;
  TPH_HIP(hipMemcpyAsync(&ok, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  if (ok) {
    p->ready = true;
    const unsigned long long keep = p->a.timeout;
    p->a.timeout = (unsigned long long)(20.0 * 1e8);
    // 24 back-to-back rounds without a host wait in between (ring reuse under whatever skew the ranks have): an all-gather of
    // 1024 patterned doubles and an all-reduce of the same, both verified on the device
    constexpr int PN = 1024, ROUNDS = 24;
    TPH_REQUIRE(ctx->comm_bytes >= 4096 + sizeof(double) * PN * (size_t)(G + 2), "tph_comm_p2p_attach: staging block too small");
    double* src = (double*)(ctx->comm_buf + 4096);
    double* dst = src + PN;                                       // [G][PN], then the all-reduce in place of src
    int* bad = (int*)(ctx->comm_buf + 1024);
    TPH_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    int rc = 0;
    for (int round = 0; round < ROUNDS && !rc; ++round) {
      hipLaunchKernelGGL(k_p2p_fill, dim3(PN / 256), dim3(256), 0, ctx->stream, src, PN, ctx->rank, round);
      rc = tph_p2p_exchange(ctx, src, dst, PN, TPH_DT_F64, -1);
      if (!rc) rc = tph_p2p_exchange(ctx, src, src, PN, TPH_DT_F64, TPH_OP_SUM);
      hipLaunchKernelGGL(k_p2p_check, dim3(PN / 256), dim3(256), 0, ctx->stream, src, dst, PN, G, round, bad);
    }
    if (rc) return rc;
    int bad_host = 1;
    TPH_HIP(hipMemcpyAsync(&bad_host, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    p->a.timeout = keep;
    int good = *p->err_host == 0 && bad_host == 0;
    p->ready = false;                          // the verdict itself goes through the callback
    TPH_HIP(hipMemcpyAsync(flag, &good, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (tph_comm_allreduce(ctx, 0, 1, TPH_DT_I32, TPH_OP_MIN)) return -2;
    TPH_HIP(hipMemcpyAsync(&ok, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TPH_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (!ok) {
    p2p_release(ctx);
    return 0;                                   // not an error: the callbacks carry everything
  }
  p->ready = true;
  *ok_out = 1;
  return 0;
}

extern "C" int tph_comm_p2p_active(const tph_ctx* ctx) { return ctx && ctx->p2p && ctx->p2p->ready ? 1 : 0; }

extern "C" int tph_comm_p2p_status(tph_ctx* ctx) {
  TPH_REQUIRE(ctx, "tph_comm_p2p_status: ctx is NULL");
  if (!ctx->p2p || !ctx->p2p->ready) return 0;
  if (*ctx->p2p->err_host != 0) return p2p_fail(*ctx->p2p->err_host);
  return 0;
}

// small all-reduce of a device array outside the staging block (the per-step acceptance sums, cluster counts): peer-to-peer
// when attached and small enough, else staged through the callback
extern "C" int tph_comm_allreduce_dev(tph_ctx* ctx, void* data_dev, int64_t count, int dtype, int op) {
  TPH_REQUIRE(ctx && data_dev && count > 0, "tph_comm_allreduce_dev: bad argument");
  TPH_REQUIRE(dtype >= 0 && dtype <= 2 && op >= 0 && op <= 2, "tph_comm_allreduce_dev: dtype %d / op %d", dtype, op);
  if (!ctx->comm_active()) return 0;
  if (tph_p2p_fits(ctx, count, dtype)) return tph_p2p_exchange(ctx, data_dev, data_dev, count, dtype, op);
  const size_t bytes = (size_t)count * (dtype == TPH_DT_I32 ? 4 : 8);
  if (tph_comm_require(ctx, bytes, "tph_comm_allreduce_dev")) return -2;
  TPH_HIP(hipMemcpyAsync(ctx->comm_buf, data_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  if (tph_comm_allreduce(ctx, 0, count, dtype, op)) return -2;
  TPH_HIP(hipMemcpyAsync(data_dev, ctx->comm_buf, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// One-sided resample shuffle.  After tph_resample_select_global every global slot k has exactly one rank that holds its
// history row (idx[k] >= 0 there, -1 elsewhere); slot k belongs to rank k / n_local.  Instead of counting, packing and an
// all-to-all-v through the host, the holder WRITES the row -- (u, x, logl) as one record of 2 d + 2 doubles, the last a
// sequence tag -- straight into record k % n_local of the owner's window over the peer mapping; one small exchange is the
// barrier; every owner then transposes its window into the SoA arrays of its active set and checks that each record carries
// this shuffle's tag.  No host synchronisation, no intermediate copies; records of neighbouring slots are contiguous.
struct put_args {
  char* win[TPH_P2P_MAX];
};

__global__ void __launch_bounds__(256) k_put_rows(put_args w, const double* __restrict__ hu, const double* __restrict__ hx,
                                                  const double* __restrict__ hl, int64_t cap, const double* __restrict__ mirror,
                                                  int d, const int64_t* __restrict__ idx, int64_t n_slots, int64_t n_local,
                                                  double tag, int self) {
  const int rec = 2 * d + 2, half = rec / 2;                    // rec is even: a lane carries TWO fields, one 16-byte store
  typedef double v2d __attribute__((ext_vector_type(2)));
  const int64_t k0 = (int64_t)blockIdx.x * 64;
  auto field = [&](int64_t s, int c) -> double {                  // from the row-major mirror when there is one (one record)
    if (c == 2 * d + 1) return tag;
    if (mirror) return mirror[(size_t)s * (rec - 1) + c];
    return c < d ? hu[(size_t)c * cap + s] : (c < 2 * d ? hx[(size_t)(c - d) * cap + s] : hl[s]);
  };
  for (int e = threadIdx.x; e < 64 * half; e += 256) {           // consecutive lanes: consecutive field pairs of one record
    const int r = e / half, c = 2 * (e - r * half);
    const int64_t k = k0 + r;
    if (k >= n_slots) break;
    const int64_t s = idx[k];
    if (s < 0) continue;
    const int owner = (int)(k / n_local);
    if (owner == self) continue;                                  // rows that stay on this rank never pass through a window
    // records are 16 d + 16 bytes, the window is page-aligned: even fields sit on 16-byte boundaries.  Plain stores: the window
    // is uncached (write-through) memory, and the fence below orders them before the kernel counts as complete
    v2d* dst = (v2d*)((double*)w.win[owner] + (size_t)(k - (int64_t)owner * n_local) * rec + c);
    *dst = v2d{field(s, c), field(s, c + 1)};
  }
  __threadfence_system();     // the stores have landed before this kernel counts as complete (the barrier follows in stream order)
}

// window [n_local][rec] -> u, x (d x ld, dimension-major), logl; a record without this shuffle's tag raises the error word
__global__ void __launch_bounds__(256) k_unpack_rows(const double* win, int d, int64_t n_local, double tag, double* __restrict__ u,
                                                     double* __restrict__ x, double* __restrict__ l, int64_t ld, unsigned int* err,
                                                     const int64_t* __restrict__ idx_mine, const double* __restrict__ mirror,
                                                     const double* __restrict__ hu, const double* __restrict__ hx,
                                                     const double* __restrict__ hl, int64_t cap) {
  // idx_mine: the selection's entries of THIS rank's slots; a slot whose row this rank holds itself (idx >= 0) is gathered
  // straight from the history (through the mirror when there is one), the others come out of the window
  extern __shared__ double tile[];                                // [64][rec + 1]
  const int rec = 2 * d + 2, pitch = rec + 1;
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * rec; e += 256) {
    const int r = e / rec, c = e - r * rec;
    if (i0 + r >= n_local) continue;
    const int64_t s = idx_mine[i0 + r];
    double v;
    if (s < 0) v = __hip_atomic_load(win + (size_t)(i0 + r) * rec + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (c == 2 * d + 1) v = tag;
    else if (mirror) v = mirror[(size_t)s * (rec - 1) + c];
    else v = c < d ? hu[(size_t)c * cap + s] : (c < 2 * d ? hx[(size_t)(c - d) * cap + s] : hl[s]);
    tile[r * pitch + c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * rec; e += 256) {            // consecutive lanes: consecutive particles of one column
    const int c = e / 64, r = e - c * 64;
    const int64_t i = i0 + r;
    if (i >= n_local) continue;
    const double v = tile[r * pitch + c];
    if (c < d) u[(size_t)c * ld + i] = v;
    else if (c < 2 * d) x[(size_t)(c - d) * ld + i] = v;
    else if (c == 2 * d) l[i] = v;
    else if (v != tag) *err = TPH_P2P_ERR_TAG;
  }
}

// Grows the row window and maps the peers' windows.  Allocation and mapping can fail on one rank only, so the ranks AGREE on the
// outcome (a MIN over the peer-to-peer exchange itself) before anybody uses a window: *usable = 1 on every rank or 0 on every
// rank -- then for good (win_failed), and the caller shuffles through its all-to-all instead.
static int p2p_window_reserve(tph_ctx* ctx, size_t need, int* usable) {
  tph_p2p* p = ctx->p2p;
  *usable = 0;
  if (p->win_failed) return 0;
  if (p->win_bytes >= need) { *usable = 1; return 0; }
  const int G = ctx->world;
  // every rank takes this branch in the same call (n_local and n_dim are global quantities)
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  for (int r = 0; r < G; ++r)
    if (p->win_opened[r]) { (void)hipIpcCloseMemHandle(p->win[r]); p->win_opened[r] = false; p->win[r] = nullptr; }
  if (p->win_local) p->retired.push_back(p->win_local);
  p->win_local = nullptr; p->win_bytes = 0;
  const size_t bytes = (need + (need >> 2) + 4095) / 4096 * 4096;
  int ok = 1;
  if (getenv("TEMPEST_AMD_P2P_NOWINDOW") && ctx->rank == 0) ok = 0;      // tests: one rank fails, all must fall back
  if (!ok || hipExtMallocWithFlags((void**)&p->win_local, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    p->win_local = nullptr;
    ok = 0;
  } else {
    TPH_HIP(hipMemsetAsync(p->win_local, 0, bytes, ctx->stream));
  }
  p->win[ctx->rank] = p->win_local;
  if (tph_comm_require(ctx, 8192 + sizeof(hipIpcMemHandle_t) * (size_t)G, "tph_resample_put_global (window handles)")) return -2;
  if (G > 1) {
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof(h));
    if (ok && hipIpcGetMemHandle(&h, p->win_local) != hipSuccess) { (void)hipGetLastError(); ok = 0; }
    TPH_HIP(hipMemcpyAsync(ctx->comm_buf, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    if (tph_p2p_exchange(ctx, ctx->comm_buf, ctx->comm_buf + 4096, sizeof(h) / 8, TPH_DT_I64, -1)) return -2;
    std::vector<hipIpcMemHandle_t> all(G);
    TPH_HIP(hipMemcpyAsync(all.data(), ctx->comm_buf + 4096, sizeof(h) * (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < G && ok; ++r) {
      if (r == ctx->rank) continue;
      void* ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, all[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = 0; break; }
      p->win[r] = (char*)ptr;
      p->win_opened[r] = true;
    }
  }
  // agreement
  double flag = (double)ok;
  double* tok = (double*)ctx->comm_buf;
  TPH_HIP(hipMemcpyAsync(tok, &flag, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (tph_p2p_exchange(ctx, tok, tok, 1, TPH_DT_F64, TPH_OP_MIN)) return -2;
  TPH_HIP(hipMemcpyAsync(&flag, tok, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  if (flag != 1.0) {
    for (int r = 0; r < G; ++r)
      if (p->win_opened[r]) { (void)hipIpcCloseMemHandle(p->win[r]); p->win_opened[r] = false; p->win[r] = nullptr; }
    if (p->win_local) p->retired.push_back(p->win_local);
    p->win_local = nullptr;
    p->win_failed = true;
    return 0;
  }
  p->win_bytes = bytes;
  *usable = 1;
  return 0;
}

extern "C" int tph_resample_put_global(tph_ctx* ctx, const int64_t* idx_dev, int64_t n_slots, int64_t n_local, double* u_out,
                                       double* x_out, double* logl_out, int64_t ld_out) {
  TPH_REQUIRE(ctx && idx_dev && u_out && x_out && logl_out, "tph_resample_put_global: NULL argument");
  TPH_REQUIRE(ctx->p2p && ctx->p2p->ready, "tph_resample_put_global: needs the peer-to-peer exchange (tph_comm_p2p_attach)");
  TPH_REQUIRE(n_local > 0 && n_slots == n_local * ctx->world && ld_out >= n_local && ctx->size > 0, "tph_resample_put_global: bad sizes");
  tph_p2p* p = ctx->p2p;
  if (*p->err_host != 0) return p2p_fail(*p->err_host);
  const int d = ctx->d, rec = 2 * d + 2;
  int usable = 0;
  if (p2p_window_reserve(ctx, sizeof(double) * (size_t)n_local * rec, &usable)) return -2;
  if (!usable) return 1;                          // not an error: every rank returns 1 and shuffles through its all-to-all
  const double tag = (double)(++p->put_seq);
  put_args w{};
  for (int r = 0; r < ctx->world; ++r) w.win[r] = p->win[r];
  const double* mirror = tph_rows_sync(ctx);
  ctx->stat[3] += n_local;                              // this rank's slots refilled by the shuffle ((world-1)/world of them from peers)
  ctx->stat[4] += n_local * (int64_t)rec * 8;
  if (ctx->world > 1) {                                 // rows for OTHER ranks' slots go into their windows
    hipLaunchKernelGGL(k_put_rows, dim3((unsigned)((n_slots + 63) / 64)), dim3(256), 0, ctx->stream, w, ctx->u, ctx->x, ctx->logl,
                       ctx->cap, mirror, d, idx_dev, n_slots, n_local, tag, ctx->rank);
    TPH_LAUNCH_CHECK();
  }
  // barrier: a rank raises its flag only after its put kernel has completed (stream order), i.e. after its stores have landed
  double* token = (double*)ctx->comm_buf;
  TPH_HIP(hipMemsetAsync(token, 0, sizeof(double), ctx->stream));
  if (tph_p2p_exchange(ctx, token, token, 1, TPH_DT_F64, TPH_OP_SUM)) return -2;
  const size_t lds = sizeof(double) * 64 * (size_t)(rec + 1);
  if (lds > 64 * 1024) TPH_HIP(hipFuncSetAttribute((const void*)k_unpack_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((n_local + 63) / 64)), dim3(256), lds, ctx->stream, (const double*)p->win_local, d, n_local,
                     tag, u_out, x_out, logl_out, ld_out, p->a.err, idx_dev + (int64_t)ctx->rank * n_local, mirror, ctx->u, ctx->x, ctx->logl,
                     ctx->cap);
  TPH_LAUNCH_CHECK();
  return 0;
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_p2p(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
