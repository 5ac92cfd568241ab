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

struct tph_p2p {
  p2p_args a{};
  char* inbox_local = nullptr;
  bool opened[TPH_P2P_MAX] = {};
  unsigned int* err_host = nullptr;
  bool ready = false;
};

template <typename T>
__global__ void __launch_bounds__(256) k_p2p(p2p_args a, const T* src, T* dst, int count, int op) {
  (void)p2p_block_exchange(a, src, dst, count, op);
}

static size_t p2p_inbox_bytes(int world) { return 2 * (size_t)world * (TPH_P2P_SLOT + TPH_P2P_FLAG); }

static void p2p_release(tph_ctx* ctx) {
  tph_p2p* p = ctx->p2p;
  if (!p) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);      // after this rank's last exchange nobody writes into its inbox any more
  for (int r = 0; r < TPH_P2P_MAX; ++r)
    if (p->opened[r]) (void)hipIpcCloseMemHandle(p->a.inbox[r]);
  if (p->inbox_local) (void)hipFree(p->inbox_local);
  if (p->a.seq) (void)hipFree(p->a.seq);
  if (p->err_host) (void)hipHostFree(p->err_host);
  delete p;
  ctx->p2p = nullptr;
}
void tph_p2p_release(tph_ctx* ctx) { p2p_release(ctx); }

bool tph_p2p_fits(const tph_ctx* ctx, int64_t count, int dtype) {
  return ctx->p2p && ctx->p2p->ready && count > 0 && (size_t)count * (dtype == TPH_DT_I32 ? 4 : 8) <= TPH_P2P_SLOT;
}

const p2p_args* tph_p2p_ready(tph_ctx* ctx, int64_t count, int dtype) {
  if (!tph_p2p_fits(ctx, count, dtype)) return nullptr;
  if (*ctx->p2p->err_host != 0) {
    tph_set_error("a peer-to-peer exchange timed out: rank %u never raised its flag (a peer died or ran a different sequence of "
                  "collectives)", *ctx->p2p->err_host - 1);
    return nullptr;
  }
  return &ctx->p2p->a;
}

// one exchange on the ctx stream (src/dst: device pointers; op < 0 = all-gather)
int tph_p2p_exchange(tph_ctx* ctx, const void* src, void* dst, int64_t count, int dtype, int op) {
  tph_p2p* p = ctx->p2p;
  TPH_REQUIRE(p && p->ready, "peer-to-peer collectives are not attached");
  TPH_REQUIRE(*p->err_host == 0, "a peer-to-peer exchange timed out: rank %u never raised its flag (a peer died or ran a different "
              "sequence of collectives)", *p->err_host - 1);
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
  double secs = 120.0;
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
    double* probe = (double*)(ctx->comm_buf + 256);              // [0] = rank + 1 -> gathered at [8 .. 8 + G), sum at [0]
    std::vector<double> host(8 + G, 0.0);
    host[0] = (double)(ctx->rank + 1);
    TPH_HIP(hipMemcpyAsync(probe, host.data(), sizeof(double) * (8 + G), hipMemcpyHostToDevice, ctx->stream));
    int rc = tph_p2p_exchange(ctx, probe, probe + 8, 1, TPH_DT_F64, -1);
    if (!rc) rc = tph_p2p_exchange(ctx, probe, probe, 1, TPH_DT_F64, TPH_OP_SUM);
    if (rc) return rc;
    TPH_HIP(hipMemcpyAsync(host.data(), probe, sizeof(double) * (8 + G), hipMemcpyDeviceToHost, ctx->stream));
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    p->a.timeout = keep;
    int good = *p->err_host == 0 && host[0] == 0.5 * G * (G + 1);
    for (int r = 0; r < G; ++r) good = good && host[8 + r] == (double)(r + 1);
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
  TPH_REQUIRE(*ctx->p2p->err_host == 0, "a peer-to-peer exchange timed out: rank %u never raised its flag (a peer died or ran a "
              "different sequence of collectives)", *ctx->p2p->err_host - 1);
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
