// Context + persistent particle history (reference: tempest/state_manager.py:171-176,267-320,356-416)
// and the cached log-mixture denominator of the MIS weights (state_manager.py:466-471).
#include "common.h"
#include <stdlib.h>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[1024] = "";

void tph_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tph_last_error(void) { return g_err; }
extern "C" int tph_version(void) { return TPH_VERSION; }

int tph_scratch_reserve(tph_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return 0;
  size_t nb = ctx->scratch_bytes ? ctx->scratch_bytes : (size_t)1 << 20;
  while (nb < bytes) nb *= 2;
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->scratch) TPH_HIP(hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  hipError_t e = hipMalloc(&ctx->scratch, nb);
  if (e != hipSuccess && nb > bytes) {              // not the doubled size: exactly what is asked for
    (void)hipGetLastError();
    nb = (bytes + 255) / 256 * 256;
    e = hipMalloc(&ctx->scratch, nb);
  }
  if (e != hipSuccess && ctx->rows) {               // the row-major mirror is a cache: its memory goes back before this fails
    (void)hipGetLastError();
    tph_rows_drop(ctx);
    e = hipMalloc(&ctx->scratch, nb);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ctx->scratch = nullptr;
    tph_set_error("tph_scratch_reserve: cannot allocate %zu bytes (%s)", nb, hipGetErrorString(e));
    return -1;
  }
  ctx->scratch_bytes = nb;
  return 0;
}

static int table_reserve(tph_ctx* ctx, int T) {
  if (T <= ctx->table_cap) return 0;
  int nc = ctx->table_cap ? ctx->table_cap : 256;
  while (nc < T) nc *= 2;
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->table_dev) TPH_HIP(hipFree(ctx->table_dev));
  if (ctx->table_host) TPH_HIP(hipHostFree(ctx->table_host));
  ctx->table_dev = nullptr; ctx->table_host = nullptr; ctx->table_uploaded = 0;
  TPH_HIP(hipMalloc((void**)&ctx->table_dev, sizeof(double) * 3 * (size_t)nc));
  TPH_HIP(hipHostMalloc((void**)&ctx->table_host, sizeof(double) * 3 * (size_t)nc));
  ctx->table_cap = nc;
  return 0;
}

// (beta_t, logZ_t, log n_t) for t < T on the device.  The host mirror is PINNED and persistent, and only the rows not yet
// on the device are copied (each row's three entries are written once and never again), so a commit enqueues three small
// asynchronous copies and does not wait for the stream (the first version re-uploaded the whole table from a temporary
// vector and synchronised on every commit).
static int table_upload(tph_ctx* ctx) {
  int T = (int)ctx->beta_t.size();
  if (T == 0) return 0;
  if (table_reserve(ctx, T)) return -1;
  const int cap = ctx->table_cap;
  if (ctx->table_uploaded > T) ctx->table_uploaded = 0;       // history cleared / reloaded: start over
  const int t0 = ctx->table_uploaded;
  for (int t = t0; t < T; ++t) {
    ctx->table_host[t] = ctx->beta_t[t];
    ctx->table_host[cap + t] = ctx->logz_t[t];
    ctx->table_host[2 * (size_t)cap + t] = log((double)ctx->n_global_t[t]);
  }
  const size_t nb = sizeof(double) * (size_t)(T - t0);
  for (int k = 0; k < 3; ++k)
    TPH_HIP(hipMemcpyAsync(ctx->table_dev + (size_t)k * cap + t0, ctx->table_host + (size_t)k * cap + t0, nb, hipMemcpyHostToDevice,
                           ctx->stream));
  ctx->table_uploaded = T;
  return 0;
}

// rows x n doubles between two pitched device arrays (dimension-major blocks of the history): a kernel, because hipMemcpy2D
// rejects addresses inside a mapped range (invalid argument on ROCm 7.2); 16 B per lane where the alignment allows
__global__ void __launch_bounds__(256) k_copy2d(double* __restrict__ dst, int64_t dst_ld, const double* __restrict__ src, int64_t src_ld,
                                                int64_t n) {
  const double* s = src + (size_t)blockIdx.y * src_ld;
  double* d = dst + (size_t)blockIdx.y * dst_ld;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
static int copy2d(tph_ctx* ctx, double* dst, int64_t dst_ld, const double* src, int64_t src_ld, int64_t n, int rows) {
  if (n <= 0 || rows <= 0) return 0;
  int gx = tph_grid_for(n, 256, 4, 2048);
  hipLaunchKernelGGL(k_copy2d, dim3(gx, rows), dim3(256), 0, ctx->stream, dst, dst_ld, src, src_ld, n);
  TPH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------- mapped, growing arrays (tph_vm_set)
size_t tph_vm_piece(int device, size_t first_bytes_per_slab) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t g = 0;
  if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) {
    (void)hipGetLastError();
    return 0;
  }
  // 2 MiB pieces for small sets, 32 MiB once a slab starts at 64 MiB or more (a 100 GB history is ~3 000 pieces of ~15 us)
  const size_t want = first_bytes_per_slab >= ((size_t)64 << 20) ? (size_t)32 << 20 : (size_t)2 << 20;
  return want % g == 0 ? want : 0;
}

int tph_vm_reserve(tph_vm_set* v, int device, int slabs, size_t stride_bytes, size_t piece_bytes) {
  TPH_REQUIRE(!v->on() && slabs > 0 && piece_bytes > 0, "tph_vm_reserve: bad argument");
  TPH_REQUIRE(stride_bytes > 0 && stride_bytes % piece_bytes == 0, "tph_vm_reserve: the stride must be a multiple of %zu bytes", piece_bytes);
  void* p = nullptr;
  hipError_t e = hipMemAddressReserve(&p, stride_bytes * (size_t)slabs, piece_bytes, nullptr, 0);
  if (e != hipSuccess || !p) {
    (void)hipGetLastError();
    tph_set_error("tph_vm_reserve: cannot reserve %zu bytes of address space (%s)", stride_bytes * (size_t)slabs, hipGetErrorString(e));
    return -1;
  }
  v->device = device; v->base = (char*)p; v->stride = stride_bytes; v->slabs = slabs; v->mapped = 0; v->piece = piece_bytes;
  v->handles.clear();
  return 0;
}

static hipError_t vm_map_piece(const tph_vm_set* v, char* at, hipMemGenericAllocationHandle_t h) {
  hipError_t e = hipMemMap(at, v->piece, 0, h, 0);
  if (e != hipSuccess) return e;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = v->device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(at, v->piece, &acc, 1);
  if (e != hipSuccess) (void)hipMemUnmap(at, v->piece);
  return e;
}

// every slab backed up to `want` bytes (rounded up to whole pieces): one physical allocation per slab and piece.  A call that
// cannot be completed is undone entirely -- what was mapped before stays as it is (return 1: out of memory).
int tph_vm_grow(tph_vm_set* v, size_t want) {
  TPH_REQUIRE(v->on(), "tph_vm_grow: nothing reserved");
  want = (want + v->piece - 1) / v->piece * v->piece;
  if (want <= v->mapped) return 0;
  TPH_REQUIRE(want <= v->stride, "tph_vm_grow: %zu bytes per array exceed the reserved %zu", want, v->stride);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = v->device;
  const size_t first = v->handles.size();
  hipError_t e = hipSuccess;
  for (size_t off = v->mapped; off < want && e == hipSuccess; off += v->piece)
    for (int sl = 0; sl < v->slabs; ++sl) {
      hipMemGenericAllocationHandle_t h;
      e = hipMemCreate(&h, v->piece, &prop, 0);
      if (e != hipSuccess) break;
      e = vm_map_piece(v, v->base + (size_t)sl * v->stride + off, h);
      if (e != hipSuccess) { (void)hipMemRelease(h); break; }
      v->handles.push_back(h);
    }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (size_t k = v->handles.size(); k > first; --k) {        // undo this call's pieces (in mapping order: piece-major, slab inner)
      const size_t idx = k - 1;
      const size_t off = (idx / (size_t)v->slabs) * v->piece;
      const int sl = (int)(idx % (size_t)v->slabs);
      (void)hipMemUnmap(v->base + (size_t)sl * v->stride + off, v->piece);
      (void)hipMemRelease(v->handles[idx]);
    }
    v->handles.resize(first);
    tph_set_error("tph_vm_grow: cannot back %d arrays up to %zu bytes each (%s)", v->slabs, want, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? 1 : -1;
  }
  v->mapped = want;
  return 0;
}

// the same physical memory behind a WIDER spacing of the slabs: a new address range, every piece unmapped from the old one and
// mapped at its slab's new place.  The stream is drained first (kernels in flight still use the old addresses).
int tph_vm_restride(tph_vm_set* v, size_t new_stride, hipStream_t stream) {
  TPH_REQUIRE(v->on() && new_stride % v->piece == 0 && new_stride >= v->mapped, "tph_vm_restride: bad stride");
  TPH_HIP(hipStreamSynchronize(stream));
  void* p = nullptr;
  hipError_t e = hipMemAddressReserve(&p, new_stride * (size_t)v->slabs, v->piece, nullptr, 0);
  if (e != hipSuccess || !p) {
    (void)hipGetLastError();
    tph_set_error("tph_vm_restride: cannot reserve %zu bytes of address space (%s)", new_stride * (size_t)v->slabs, hipGetErrorString(e));
    return -1;
  }
  char* nb = (char*)p;
  for (size_t idx = 0; idx < v->handles.size(); ++idx) {
    const size_t off = (idx / (size_t)v->slabs) * v->piece;
    const int sl = (int)(idx % (size_t)v->slabs);
    TPH_HIP(hipMemUnmap(v->base + (size_t)sl * v->stride + off, v->piece));
    hipError_t m = vm_map_piece(v, nb + (size_t)sl * new_stride + off, v->handles[idx]);
    TPH_REQUIRE(m == hipSuccess, "tph_vm_restride: re-mapping failed (%s): the history is lost", hipGetErrorString(m));
  }
  (void)hipMemAddressFree(v->base, v->stride * (size_t)v->slabs);
  v->base = nb;
  v->stride = new_stride;
  return 0;
}

void tph_vm_release(tph_vm_set* v) {
  if (!v->on()) return;
  for (size_t idx = 0; idx < v->handles.size(); ++idx) {
    const size_t off = (idx / (size_t)v->slabs) * v->piece;
    const int sl = (int)(idx % (size_t)v->slabs);
    (void)hipMemUnmap(v->base + (size_t)sl * v->stride + off, v->piece);
    (void)hipMemRelease(v->handles[idx]);
  }
  (void)hipMemAddressFree(v->base, v->stride * (size_t)v->slabs);
  (void)hipGetLastError();
  *v = tph_vm_set();
}

// ------------------------------------------------------------------------------------ history capacity
// u and x move into a mapped range when they are asked to hold 16 GB or more, or when a plain history has to grow and the
// doubling copy (old and new arrays alive together) would not leave 40 % of the free memory; below that a history is two plain
// allocations grown by doubling -- cheap at that size, and nothing the measured configurations of 10^6 particles ever leave
constexpr size_t HIST_VM_MIN_BYTES = (size_t)16 << 30;
static bool history_wants_vm(const tph_ctx* ctx, int64_t need) {
  if (ctx->hist_vm.on() || ctx->hist_vm_mode == 2) return true;
  if (ctx->hist_vm_mode == 0) return false;
  const size_t row = sizeof(double) * 2 * (size_t)ctx->d;
  if (row * (size_t)need >= HIST_VM_MIN_BYTES) return true;
  if (ctx->cap > 0 && ctx->size > 0) {
    int64_t nc = ctx->cap;
    while (nc < need) nc *= 2;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && row * (size_t)nc > free_b / 5 * 3) return true;
    (void)hipGetLastError();
  }
  return false;
}

// logl and cmix (8 bytes per row each): plain allocations, doubled and copied -- a spike of 16 bytes per row
static int history_reserve_scalars(tph_ctx* ctx, int64_t need) {
  if (need <= ctx->cap1) return 0;
  int64_t nc = ctx->cap1 ? ctx->cap1 : 1024;
  while (nc < need) nc *= 2;
  nc = (nc + 255) / 256 * 256;
  double* fresh[2] = {nullptr, nullptr};
  double** arrs[2] = {&ctx->logl, &ctx->cmix};
  for (int a = 0; a < 2; ++a) {
    hipError_t e = hipMalloc((void**)&fresh[a], sizeof(double) * (size_t)nc);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      for (int b = 0; b < a; ++b) (void)hipFree(fresh[b]);
      tph_set_error("history_reserve: cannot grow the history to %lld rows (%s)", (long long)nc, hipGetErrorString(e));
      return -1;
    }
    if (ctx->size > 0) {
      hipError_t c = hipMemcpyAsync(fresh[a], *arrs[a], sizeof(double) * (size_t)ctx->size, hipMemcpyDeviceToDevice, ctx->stream);
      if (c != hipSuccess) {
        for (int b = 0; b <= a; ++b) (void)hipFree(fresh[b]);
        tph_set_error("history_reserve: copy failed (%s)", hipGetErrorString(c));
        return -1;
      }
    }
  }
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  for (int a = 0; a < 2; ++a) {
    if (*arrs[a]) (void)hipFree(*arrs[a]);
    *arrs[a] = fresh[a];
  }
  ctx->cap1 = nc;
  return 0;
}

// u and x as plain allocations: all new arrays are allocated and filled before any old one is released, so a failed allocation
// or copy leaves the history exactly as it was (the two arrays share one leading dimension and can only be swapped together)
static int history_reserve_plain(tph_ctx* ctx, int64_t need) {
  int64_t nc = ctx->cap ? ctx->cap : 1024;
  while (nc < need) nc *= 2;
  nc = (nc + 255) / 256 * 256;
  const size_t d = (size_t)ctx->d;
  const size_t w = sizeof(double) * (size_t)ctx->size;
  double** arrs[2] = {&ctx->u, &ctx->x};
  double* fresh[2] = {nullptr, nullptr};
  for (int a = 0; a < 2; ++a) {
    hipError_t e = hipMalloc((void**)&fresh[a], sizeof(double) * d * (size_t)nc);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      for (int b = 0; b < a; ++b) (void)hipFree(fresh[b]);
      tph_set_error("history_reserve: cannot grow the history to %lld rows (%s)", (long long)nc, hipGetErrorString(e));
      return -1;
    }
    if (ctx->size > 0) {
      hipError_t c = hipMemcpy2DAsync(fresh[a], sizeof(double) * nc, *arrs[a], sizeof(double) * ctx->cap, w, d, hipMemcpyDeviceToDevice, ctx->stream);
      if (c == hipSuccess) c = hipStreamSynchronize(ctx->stream);
      if (c != hipSuccess) {
        for (int b = 0; b <= a; ++b) (void)hipFree(fresh[b]);
        tph_set_error("history_reserve: copy failed (%s)", hipGetErrorString(c));
        return -1;
      }
      ctx->stat_mem[3] += 1;
    }
  }
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  for (int a = 0; a < 2; ++a) {
    if (*arrs[a]) (void)hipFree(*arrs[a]);
    *arrs[a] = fresh[a];
  }
  ctx->cap = nc;
  ctx->hist_mapped = nc;
  return 0;
}

// u and x in a mapped range (state_manager.py:356-416 appends without bound: the reference's lists of arrays never move either).
// Growth: the next `need` rows, and at least an eighth more than is mapped, so that the driver calls (2 n_dim allocations and
// mappings per step) stay a handful per run; under memory pressure the mirror goes first (it is a cache), then exactly `need`.
static int history_reserve_vm(tph_ctx* ctx, int64_t need) {
  tph_vm_set& v = ctx->hist_vm;
  const size_t g = v.on() ? v.piece : tph_vm_piece(ctx->device, sizeof(double) * (size_t)need);
  if (g == 0) return 2;                                        // not available: the caller falls back to plain allocations
  const int64_t grow_rows = (int64_t)(g / sizeof(double));     // rows per piece
  auto round_rows = [&](int64_t r) { return (r + grow_rows - 1) / grow_rows * grow_rows; };
  if (!v.on()) {
    // address space for four times what is asked for (it costs nothing); the rows held so far move in with ONE copy
    const int64_t rows_va = round_rows(need > (1ll << 40) / 4 ? need : 4 * need);
    tph_vm_set fresh;
    if (tph_vm_reserve(&fresh, ctx->device, 2 * ctx->d, sizeof(double) * (size_t)rows_va, g)) { (void)hipGetLastError(); return 2; }
    const int rc = tph_vm_grow(&fresh, sizeof(double) * (size_t)round_rows(need));
    if (rc) { tph_vm_release(&fresh); return rc == 1 ? -1 : 2; }
    if (ctx->size > 0) {
      const size_t w = sizeof(double) * (size_t)ctx->size;
      const int64_t fld = (int64_t)(fresh.stride / sizeof(double));
      int crc = copy2d(ctx, (double*)fresh.base, fld, ctx->u, ctx->cap, ctx->size, ctx->d);
      if (!crc) crc = copy2d(ctx, (double*)(fresh.base + (size_t)ctx->d * fresh.stride), fld, ctx->x, ctx->cap, ctx->size, ctx->d);
      hipError_t c = crc ? hipErrorUnknown : hipStreamSynchronize(ctx->stream);
      if (c != hipSuccess) {
        tph_vm_release(&fresh);
        tph_set_error("history_reserve: copy into the mapped range failed (%s)", hipGetErrorString(c));
        return -1;
      }
      (void)w;
      ctx->stat_mem[3] += 1;
    }
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->u) (void)hipFree(ctx->u);
    if (ctx->x) (void)hipFree(ctx->x);
    v = fresh;
    ctx->u = (double*)v.base;
    ctx->x = (double*)(v.base + (size_t)ctx->d * v.stride);
    ctx->cap = (int64_t)(v.stride / sizeof(double));
    ctx->hist_mapped = (int64_t)(v.mapped / sizeof(double));
    ctx->stat_mem[0] += 1;
    return 0;
  }
  if (need > ctx->cap) {                                       // outgrown the reserved range: a wider one, mappings moved, nothing copied
    int64_t rows_va = ctx->cap;
    while (rows_va < need) rows_va *= 2;
    if (tph_vm_restride(&v, sizeof(double) * (size_t)rows_va, ctx->stream)) return -1;
    ctx->u = (double*)v.base;
    ctx->x = (double*)(v.base + (size_t)ctx->d * v.stride);
    ctx->cap = rows_va;
    ctx->stat_mem[1] += 1;
  }
  int64_t want = ctx->hist_mapped + ctx->hist_mapped / 8;
  if (want < need) want = need;
  want = round_rows(want);
  if (want > ctx->cap) want = ctx->cap;
  int rc = tph_vm_grow(&v, sizeof(double) * (size_t)want);
  if (rc == 1 && ctx->rows) {                                  // out of memory: the mirror's share goes back first
    tph_rows_drop(ctx);
    rc = tph_vm_grow(&v, sizeof(double) * (size_t)want);
  }
  if (rc == 1 && want > round_rows(need)) rc = tph_vm_grow(&v, sizeof(double) * (size_t)round_rows(need));
  if (rc) return -1;
  ctx->hist_mapped = (int64_t)(v.mapped / sizeof(double));
  ctx->stat_mem[0] += 1;
  return 0;
}

static int history_reserve(tph_ctx* ctx, int64_t need) {
  if (history_reserve_scalars(ctx, need)) return -1;
  if (need <= ctx->hist_mapped) return 0;
  if (history_wants_vm(ctx, need)) {
    const int rc = history_reserve_vm(ctx, need);
    if (rc != 2) return rc;
    ctx->hist_vm_mode = 0;                                     // the device cannot map memory this way: plain allocations from now on
  }
  return history_reserve_plain(ctx, need);
}

extern "C" int tph_ctx_create(int device, int n_dim, int64_t capacity_hint, void* hip_stream, tph_ctx** out) {
  TPH_REQUIRE(out != nullptr, "tph_ctx_create: out is NULL");
  TPH_REQUIRE(n_dim > 0 && n_dim <= 4096, "tph_ctx_create: n_dim=%d out of range", n_dim);
  int ndev = 0;
  TPH_HIP(hipGetDeviceCount(&ndev));
  TPH_REQUIRE(device >= 0 && device < ndev, "tph_ctx_create: device %d not present (%d visible)", device, ndev);
  TPH_HIP(hipSetDevice(device));
  tph_ctx* c = new tph_ctx();
  c->device = device;
  if (const char* env = getenv("TEMPEST_AMD_ROW_MIRROR")) c->rows_mode = atoi(env) ? 1 : 0;     // debugging aid (TPH_OPT_ROW_MIRROR)
  if (const char* env = getenv("TEMPEST_AMD_SORTED_DRAWS")) c->mc_sorted = atoi(env) ? 1 : 0;  // debugging aid (TPH_OPT_SORTED_DRAWS)
  if (const char* env = getenv("TEMPEST_AMD_BLK_STAGE")) c->blk_stage = atoi(env) ? 1 : 0;       // debugging aid (TPH_OPT_BLK_STAGE)
  if (const char* env = getenv("TEMPEST_AMD_COV_KERNEL")) c->cov_kernel = atoi(env);             // debugging aid (TPH_OPT_COV_KERNEL)
  if (const char* env = getenv("TEMPEST_AMD_GMM_KERNEL")) c->gmm_kernel = atoi(env);             // debugging aid (TPH_OPT_GMM_KERNEL)
  if (const char* env = getenv("TEMPEST_AMD_FORMS_MFMA")) c->forms_mfma = atoi(env) ? 1 : 0;     // debugging aid (TPH_OPT_FORMS_MFMA)
  if (const char* env = getenv("TEMPEST_AMD_MF_DEAL")) c->mf_deal = atoi(env) < 0 ? 0 : (atoi(env) > 2 ? 2 : atoi(env));          // debugging aid (TPH_OPT_MF_DEAL)
  if (const char* env = getenv("TEMPEST_AMD_HISTORY_VM")) c->hist_vm_mode = atoi(env) < 0 ? 0 : (atoi(env) > 2 ? 2 : atoi(env));   // debugging aid (TPH_OPT_HISTORY_VM)
  c->d = n_dim;
  c->stream = (hipStream_t)hip_stream;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_simd = 4 * prop.multiProcessorCount;
  }
  c->partials_bytes = sizeof(double) * (size_t)TPH_RED_BLOCKS * 64;
  if (hipMalloc((void**)&c->partials, c->partials_bytes) != hipSuccess ||
      hipMalloc((void**)&c->small_dev, sizeof(double) * (4096 + 32)) != hipSuccess ||      // [4096]: the ticket of k_reweight_fold
      hipHostMalloc((void**)&c->pinned, sizeof(double) * 4096) != hipSuccess) {
    tph_set_error("tph_ctx_create: scratch allocation failed");
    delete c;
    return -1;
  }
  memset(c->pinned, 0, sizeof(double) * 4096);      // [4095] is the sequence word tph_reweight_eval polls
  if (hipMemset(c->small_dev + 4096, 0, sizeof(double) * 32) != hipSuccess) {
    tph_set_error("tph_ctx_create: scratch initialisation failed");
    delete c;
    return -1;
  }
  if (capacity_hint > 0 && history_reserve(c, capacity_hint)) {
    delete c;
    return -1;
  }
  if (capacity_hint > 0) {
    // the big scratch (sort buffers, scans, compaction lists) for a history of that size too: grown on demand it doubles a
    // dozen times during a run, each time behind a stream synchronisation, a hipFree and a hipMalloc of up to gigabytes.
    // 48 B per reserved row covers the largest user (the radix sorts of the trim and of the up-sampling draws); never more
    // than a sixteenth of the free memory, and a failure here is not an error (the on-demand growth still works).
    size_t free_b = 0, total_b = 0;
    size_t want = (size_t)48 * (size_t)capacity_hint;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 16) want = free_b / 16;
    if (want > ((size_t)1 << 20) && tph_scratch_reserve(c, want)) (void)hipGetLastError();
  }
  *out = c;
  return 0;
}

extern "C" int tph_ctx_destroy(tph_ctx* ctx) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  tph_p2p_release(ctx);
  const bool hist_mapped = ctx->hist_vm.on(), rows_mapped = ctx->rows_vm.on();
  tph_vm_release(&ctx->hist_vm);
  tph_vm_release(&ctx->rows_vm);
  void* bufs[] = {hist_mapped ? nullptr : ctx->u, hist_mapped ? nullptr : ctx->x, ctx->logl, ctx->cmix, ctx->table_dev, ctx->partials,
                  ctx->small_dev, ctx->scratch, ctx->winv, ctx->blk_table, ctx->vv_buf, ctx->adapt_buf, ctx->blk_buf, rows_mapped ? nullptr : ctx->rows,
                  ctx->sm_small, ctx->sm_scr, ctx->mf_buf, ctx->bm_buf, ctx->mt_buf};
  for (void* b : bufs) (void)hipFree(b);
  for (void* b : ctx->retired) (void)hipFree(b);
  (void)hipHostFree(ctx->pinned);
  if (ctx->table_host) (void)hipHostFree(ctx->table_host);
  delete ctx;
  return 0;
}

// ------------------------------------------------------------------------------------------ communicator
extern "C" int tph_comm_attach(tph_ctx* ctx, int rank, int world, void* buf_dev, int64_t buf_bytes, tph_allreduce_fn allreduce,
                               tph_allgather_fn allgather, void* user) {
  TPH_REQUIRE(ctx && buf_dev && allreduce && allgather, "tph_comm_attach: NULL argument");
  TPH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tph_comm_attach: rank %d outside world %d", rank, world);
  TPH_REQUIRE(buf_bytes >= (1 << 20), "tph_comm_attach: the staging block must hold at least 1 MiB");
  ctx->rank = rank; ctx->world = world;
  ctx->comm_buf = (char*)buf_dev; ctx->comm_bytes = (size_t)buf_bytes;
  ctx->comm_allreduce = allreduce; ctx->comm_allgather = allgather; ctx->comm_user = user;
  return 0;
}
extern "C" int tph_comm_detach(tph_ctx* ctx) {
  TPH_REQUIRE(ctx, "tph_comm_detach: ctx is NULL");
  tph_p2p_release(ctx);
  ctx->rank = 0; ctx->world = 1;
  ctx->comm_buf = nullptr; ctx->comm_bytes = 0;
  ctx->comm_allreduce = nullptr; ctx->comm_allgather = nullptr; ctx->comm_user = nullptr;
  return 0;
}
int tph_comm_require(tph_ctx* ctx, size_t bytes, const char* who) {
  TPH_REQUIRE(ctx->comm_active(), "%s: no communicator attached", who);
  TPH_REQUIRE(bytes <= ctx->comm_bytes, "%s: needs %zu B of communication staging, %zu attached (tph_comm_attach)", who, bytes,
              ctx->comm_bytes);
  return 0;
}
int tph_comm_allreduce(tph_ctx* ctx, size_t off, int64_t count, int dtype, int op) {
  if (tph_p2p_fits(ctx, count, dtype)) return tph_p2p_exchange(ctx, ctx->comm_buf + off, ctx->comm_buf + off, count, dtype, op);
  const int rc = ctx->comm_allreduce(ctx->comm_user, (int64_t)off, count, dtype, op);
  ctx->stat[1] += 1; ctx->stat[2] += count * (dtype == TPH_DT_I32 ? 4 : 8);
  TPH_REQUIRE(rc == 0, "all-reduce callback failed (%d)", rc);
  return 0;
}
int tph_comm_allgather(tph_ctx* ctx, size_t send_off, size_t recv_off, int64_t count, int dtype) {
  if (tph_p2p_fits(ctx, count, dtype)) return tph_p2p_exchange(ctx, ctx->comm_buf + send_off, ctx->comm_buf + recv_off, count, dtype, -1);
  return tph_comm_allgather_cb(ctx, send_off, recv_off, count, dtype);
}
int tph_comm_allgather_cb(tph_ctx* ctx, size_t send_off, size_t recv_off, int64_t count, int dtype) {
  const int rc = ctx->comm_allgather(ctx->comm_user, (int64_t)send_off, (int64_t)recv_off, count, dtype);
  ctx->stat[1] += 1; ctx->stat[2] += count * (dtype == TPH_DT_I32 ? 4 : 8) * (int64_t)ctx->world;
  TPH_REQUIRE(rc == 0, "all-gather callback failed (%d)", rc);
  return 0;
}
// the local history as T equal blocks of `rows` rows (one per committed iteration): what the global order is built on
int tph_blocks(tph_ctx* ctx, int64_t n, int* T, int64_t* rows) {
  TPH_REQUIRE(n == ctx->size && !ctx->n_local_t.empty(), "global-order functions work on the whole local history (%lld rows given, %lld held)",
              (long long)n, (long long)ctx->size);
  const int64_t r = ctx->n_local_t[0];
  for (int64_t v : ctx->n_local_t)
    TPH_REQUIRE(v == r, "sharded runs need the same number of particles in every iteration (%lld vs %lld)", (long long)v, (long long)r);
  *T = (int)ctx->n_local_t.size();
  *rows = r;
  return 0;
}

int tph_vshards_for(int64_t n_global) { return tph_vshards_inline((long long)n_global); }
tph_part tph_partition(const tph_ctx* ctx, int64_t n) {
  tph_part p{1, n, 1, ctx->world, n, false};
  if (n != ctx->size || ctx->n_local_t.empty()) return p;
  const int64_t nl = ctx->n_local_t[0], ng = ctx->n_global_t[0];
  for (size_t t = 0; t < ctx->n_local_t.size(); ++t)
    if (ctx->n_local_t[t] != nl || ctx->n_global_t[t] != ng) return p;
  p.T = (int)ctx->n_local_t.size();
  p.n_loc = nl;
  p.nv = nl;
  if (ng == nl * (int64_t)ctx->world && ng % 256 == 0) {
    const int V = tph_vshards_for(ng);
    if (V % ctx->world == 0) {
      p.vl = V / ctx->world;
      p.V = V;
      p.nv = ng / V;
      p.canonical = true;
    }
  }
  return p;
}
int tph_partials_reserve(tph_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->partials_bytes) return 0;
  size_t nb = ctx->partials_bytes;
  while (nb < bytes) nb *= 2;
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->partials) TPH_HIP(hipFree(ctx->partials));
  ctx->partials = nullptr;
  TPH_HIP(hipMalloc((void**)&ctx->partials, nb));
  ctx->partials_bytes = nb;
  return 0;
}

extern "C" int tph_comm_stats(tph_ctx* ctx, int64_t* out, int reset) {
  TPH_REQUIRE(ctx && out, "tph_comm_stats: NULL argument");
  for (int i = 0; i < 5; ++i) { out[i] = ctx->stat[i]; if (reset) ctx->stat[i] = 0; }
  return 0;
}

extern "C" int tph_set_stream(tph_ctx* ctx, void* hip_stream) {
  TPH_REQUIRE(ctx, "tph_set_stream: ctx is NULL");
  ctx->stream = (hipStream_t)hip_stream;
  return 0;
}

extern "C" int tph_set_option(tph_ctx* ctx, int option, int value) {
  TPH_REQUIRE(ctx, "tph_set_option: ctx is NULL");
  switch (option) {
    case TPH_OPT_PROPOSE_VARIANT: ctx->propose_variant = value; break;
    case TPH_OPT_REDUCE_GRID: ctx->reduce_grid = value; break;
    case TPH_OPT_REDRAW_LANES: ctx->redraw_lanes = value; break;
    case TPH_OPT_ML_UNSTAGED: ctx->ml_unstaged = value; break;
    case TPH_OPT_ROW_MIRROR: ctx->rows_mode = value ? 1 : 0; break;
    case TPH_OPT_COV_KERNEL: ctx->cov_kernel = value; break;
    case TPH_OPT_SORTED_DRAWS: ctx->mc_sorted = value < 0 ? 0 : value; break;
    case TPH_OPT_BLOCKED: ctx->blocked = value; break;
    case TPH_OPT_MODES_EPOCH: ctx->modes_epoch = value; break;
    case TPH_OPT_STAGED_REDRAW: ctx->staged = value; break;
    case TPH_OPT_SM_LANES: ctx->sm_lanes = value; break;
    case TPH_OPT_SM_THRESHOLD: ctx->sm_thr = value; break;
    case TPH_OPT_SCREEN: ctx->screen = value ? 1 : 0; break;
    case TPH_OPT_MF_LANES: ctx->mf_lanes = value; break;
    case TPH_OPT_MF_AUDIT: ctx->mf_audit = value ? 1 : 0; break;
    case TPH_OPT_BLK_MFMA: ctx->blk_mfma = value ? 1 : 0; break;
    case TPH_OPT_BLK_TRIES: ctx->blk_tries = value < 0 ? 0 : (value > 3 ? 3 : value); break;
    case TPH_OPT_BLK_FAN: ctx->blk_fan = value < 0 ? 0 : (value > 3 ? 3 : value); break;
    case TPH_OPT_BLK_STAGE: ctx->blk_stage = value ? 1 : 0; break;
    case TPH_OPT_GMM_KERNEL: ctx->gmm_kernel = value < 0 || value > 2 ? 0 : value; break;
    case TPH_OPT_FORMS_MFMA: ctx->forms_mfma = value ? 1 : 0; break;
    case TPH_OPT_MF_DEAL: ctx->mf_deal = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    case TPH_OPT_HISTORY_VM: ctx->hist_vm_mode = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    default: TPH_REQUIRE(false, "tph_set_option: unknown option %d", option);
  }
  return 0;
}

// Every translation unit of the library is a code object of its own, loaded by the HIP runtime when one of its kernels is first
// launched -- 20 MB of them (the proposal kernels' template instantiations), tens of milliseconds each as the first process on a
// machine.  A run met them one by one in its first iterations (first reweight, first fit, first resample, first d > 16 proposal,
// first clustering fit).  This launches one empty kernel of each unit: called from the Sampler's start-up thread, the loads
// happen beside torch's own, before the first iteration needs them.
extern "C" int tph_warmup(tph_ctx* ctx) {
  TPH_REQUIRE(ctx, "tph_warmup: ctx is NULL");
  TPH_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, (unsigned int*)nullptr, 0);
  tph_warm_cluster(s);
  tph_warm_modes(s);
  tph_warm_mutate(s);
  tph_warm_p2p(s);
  tph_warm_propose_blkm(s);
  tph_warm_propose_mf(s);
  tph_warm_propose_sm(s);
  tph_warm_resample(s);
  tph_warm_reweight(s);
  tph_warm_student(s);
  TPH_LAUNCH_CHECK();
  TPH_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int tph_synchronize(tph_ctx* ctx) {
  TPH_REQUIRE(ctx, "tph_synchronize: ctx is NULL");
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int64_t tph_history_size(const tph_ctx* ctx) { return ctx ? ctx->size : -1; }
extern "C" int tph_history_iterations(const tph_ctx* ctx) { return ctx ? (int)ctx->beta_t.size() : -1; }

// where the history's memory is: out[0] rows held, [1] rows of u / x backed by memory, [2] rows reserved (leading dimension),
// [3] 1 = u / x in a mapped range, [4] rows of the row-major mirror backed (0: none), [5] growth steps of the mapped ranges,
// [6] re-reservations (mappings moved), [7] times the mirror was given back under memory pressure, [8] copies of the whole history
extern "C" int tph_history_memory(tph_ctx* ctx, int64_t* out9) {
  TPH_REQUIRE(ctx && out9, "tph_history_memory: NULL argument");
  out9[0] = ctx->size; out9[1] = ctx->hist_mapped; out9[2] = ctx->cap; out9[3] = ctx->hist_vm.on() ? 1 : 0;
  out9[4] = ctx->rows ? ctx->rows_cap : 0;
  out9[5] = ctx->stat_mem[0]; out9[6] = ctx->stat_mem[1]; out9[7] = ctx->stat_mem[2]; out9[8] = ctx->stat_mem[3];
  return 0;
}

extern "C" int tph_history_clear(tph_ctx* ctx) {
  TPH_REQUIRE(ctx, "tph_history_clear: ctx is NULL");
  ctx->size = 0;
  ctx->rows_size = 0;
  ctx->beta_t.clear(); ctx->logz_t.clear(); ctx->n_local_t.clear(); ctx->n_global_t.clear();
  ctx->table_uploaded = 0;
  return 0;
}

// K1: log-mixture update.  Old rows fold in the one new term (one logaddexp); new rows take all T terms in iteration order.
// HBM: old rows 24 B (read l, C; write C); new rows 16 B + T table terms (scalar loads).  The old rows go two to a lane in
// 16-byte accesses, two such pairs in flight per lane before the first logaddexp (a row at a time the kernel sat at 0.43 of the
// HBM rate, waiting on one 8-byte load per lane behind ~100 FP64 instructions of exp and log1p).
__global__ void __launch_bounds__(256) k_logmix_append(const double* __restrict__ logl, double* __restrict__ cmix,
                                                       int64_t size_old, int64_t size_new,
                                                       const double* __restrict__ table, int tcap, int T) {
  const double* beta = table;
  const double* logz = table + tcap;
  const double* logn = table + 2 * (size_t)tcap;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double bT = beta[T - 1], zT = logz[T - 1], nT = logn[T - 1];
  const int64_t pairs = size_old >> 1;
  const double2* __restrict__ l2 = reinterpret_cast<const double2*>(logl);
  double2* __restrict__ c2 = reinterpret_cast<double2*>(cmix);
  auto fold = [&](double2 l, double2 c) {
    c.x = tph_logaddexp(c.x, l.x * bT - zT + nT);
    c.y = tph_logaddexp(c.y, l.y * bT - zT + nT);
    return c;
  };
  int64_t p = tid;
  for (; p + stride < pairs; p += 2 * stride) {
    const double2 la = l2[p], ca = c2[p], lb = l2[p + stride], cb = c2[p + stride];
    c2[p] = fold(la, ca);
    c2[p + stride] = fold(lb, cb);
  }
  if (p < pairs) c2[p] = fold(l2[p], c2[p]);
  for (int64_t s = 2 * pairs + tid; s < size_new; s += stride) {
    double l = logl[s];
    if (s < size_old) {
      cmix[s] = tph_logaddexp(cmix[s], l * bT - zT + nT);
    } else {
      // streaming log-sum-exp over the T terms (running maximum m and sum of exp(a_t - m)): one exp per term and one log
      // per row, where the pairwise fold of np.logaddexp.reduce costs an exp AND a log1p per term; same value to rounding
      double m = l * beta[0] - logz[0] + logn[0], ssum = 1.0;
      for (int t = 1; t < T; ++t) {
        const double a = l * beta[t] - logz[t] + logn[t];
        if (a > m) { ssum = ssum * exp(m - a) + 1.0; m = a; }
        else ssum += (a == m) ? 1.0 : exp(a - m);          // a == m also covers -inf, -inf (logaddexp gives -inf)
      }
      cmix[s] = m + log(ssum);
    }
  }
}

static int launch_logmix(tph_ctx* ctx, int64_t size_old, int64_t size_new) {
  int T = (int)ctx->beta_t.size();
  int grid = tph_grid_for(size_new, 256);
  hipLaunchKernelGGL(k_logmix_append, dim3(grid), dim3(256), 0, ctx->stream, ctx->logl, ctx->cmix, size_old,
                     size_new, ctx->table_dev, ctx->table_cap, T);
  TPH_LAUNCH_CHECK();
  return 0;
}

extern "C" int tph_history_append(tph_ctx* ctx, const double* u_dev, const double* x_dev, const double* logl_dev,
                                  int64_t n, int64_t ld, double beta, double logz, int64_t n_global) {
  TPH_REQUIRE(ctx, "tph_history_append: ctx is NULL");
  TPH_REQUIRE(n > 0 && ld >= n, "tph_history_append: bad n=%lld ld=%lld", (long long)n, (long long)ld);
  TPH_REQUIRE(u_dev && x_dev && logl_dev, "tph_history_append: NULL array");
  TPH_REQUIRE(n_global >= n, "tph_history_append: n_global < n");
  TPH_HIP(hipSetDevice(ctx->device));
  if (history_reserve(ctx, ctx->size + n)) return -1;
  size_t w = sizeof(double) * (size_t)n;
  if (copy2d(ctx, ctx->u + ctx->size, ctx->cap, u_dev, ld, n, ctx->d)) return -1;
  if (copy2d(ctx, ctx->x + ctx->size, ctx->cap, x_dev, ld, n, ctx->d)) return -1;
  TPH_HIP(hipMemcpyAsync(ctx->logl + ctx->size, logl_dev, w, hipMemcpyDeviceToDevice, ctx->stream));
  ctx->beta_t.push_back(beta);
  ctx->logz_t.push_back(logz);
  ctx->n_local_t.push_back(n);
  ctx->n_global_t.push_back(n_global);
  if (table_upload(ctx)) return -1;
  int64_t old = ctx->size;
  ctx->size += n;
  return launch_logmix(ctx, old, ctx->size);
}

extern "C" int tph_history_load(tph_ctx* ctx, const double* u_host, const double* x_host, const double* logl_host,
                                int64_t n, int T, const double* beta_t, const double* logz_t,
                                const int64_t* n_t_local, const int64_t* n_t_global) {
  TPH_REQUIRE(ctx, "tph_history_load: ctx is NULL");
  TPH_REQUIRE(n >= 0 && T >= 0, "tph_history_load: bad sizes");
  TPH_HIP(hipSetDevice(ctx->device));
  tph_history_clear(ctx);
  int64_t tot = 0;
  for (int t = 0; t < T; ++t) {
    ctx->beta_t.push_back(beta_t[t]);
    ctx->logz_t.push_back(logz_t[t]);
    ctx->n_local_t.push_back(n_t_local[t]);
    ctx->n_global_t.push_back(n_t_global ? n_t_global[t] : n_t_local[t]);
    tot += n_t_local[t];
  }
  TPH_REQUIRE(tot == n, "tph_history_load: sum(n_t)=%lld != n=%lld", (long long)tot, (long long)n);
  if (n == 0) return 0;
  if (history_reserve(ctx, n)) return -1;
  size_t w = sizeof(double) * (size_t)n;
  // (one 1-D copy per coordinate: hipMemcpy2D rejects addresses inside a mapped range)
  for (int j = 0; j < ctx->d && u_host; ++j)
    TPH_HIP(hipMemcpyAsync(ctx->u + (size_t)j * ctx->cap, u_host + (size_t)j * n, w, hipMemcpyHostToDevice, ctx->stream));
  for (int j = 0; j < ctx->d && x_host; ++j)
    TPH_HIP(hipMemcpyAsync(ctx->x + (size_t)j * ctx->cap, x_host + (size_t)j * n, w, hipMemcpyHostToDevice, ctx->stream));
  TPH_REQUIRE(logl_host, "tph_history_load: logl is NULL");
  TPH_HIP(hipMemcpyAsync(ctx->logl, logl_host, w, hipMemcpyHostToDevice, ctx->stream));
  if (table_upload(ctx)) return -1;
  ctx->size = n;
  if (launch_logmix(ctx, 0, n)) return -1;
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int tph_history_read(tph_ctx* ctx, int key, int64_t off, int64_t n, double* out_host) {
  TPH_REQUIRE(ctx && out_host, "tph_history_read: NULL argument");
  TPH_REQUIRE(off >= 0 && n >= 0 && off + n <= ctx->size, "tph_history_read: range [%lld,%lld) outside history of %lld",
              (long long)off, (long long)(off + n), (long long)ctx->size);
  if (n == 0) return 0;
  TPH_HIP(hipSetDevice(ctx->device));
  size_t w = sizeof(double) * (size_t)n;
  switch (key) {
    case TPH_KEY_U:
    case TPH_KEY_X: {
      const double* src = (key == TPH_KEY_U ? ctx->u : ctx->x) + off;
      for (int j = 0; j < ctx->d; ++j)
        TPH_HIP(hipMemcpyAsync(out_host + (size_t)j * n, src + (size_t)j * ctx->cap, w, hipMemcpyDeviceToHost, ctx->stream));
      break;
    }
    case TPH_KEY_LOGL:
      TPH_HIP(hipMemcpyAsync(out_host, ctx->logl + off, w, hipMemcpyDeviceToHost, ctx->stream));
      break;
    case TPH_KEY_LOGMIX:
      TPH_HIP(hipMemcpyAsync(out_host, ctx->cmix + off, w, hipMemcpyDeviceToHost, ctx->stream));
      break;
    default:
      TPH_REQUIRE(false, "tph_history_read: unknown key %d", key);
  }
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int tph_history_ptr(tph_ctx* ctx, int key, void** dev_ptr, int64_t* ld) {
  TPH_REQUIRE(ctx && dev_ptr, "tph_history_ptr: NULL argument");
  switch (key) {
    case TPH_KEY_U: *dev_ptr = ctx->u; break;
    case TPH_KEY_X: *dev_ptr = ctx->x; break;
    case TPH_KEY_LOGL: *dev_ptr = ctx->logl; break;
    case TPH_KEY_LOGMIX: *dev_ptr = ctx->cmix; break;
    default: TPH_REQUIRE(false, "tph_history_ptr: unknown key %d", key);
  }
  if (ld) *ld = ctx->cap;
  return 0;
}
