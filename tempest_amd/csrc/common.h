// Shared host/device helpers for libtempest_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <math.h>
#include <string>
#include <vector>

#include "../../include/tempest_hip.h"

// ------------------------------------------------------------------------------------------ errors
void tph_set_error(const char* fmt, ...);

#define TPH_HIP(call)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      tph_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));       \
      return -1;                                                                                \
    }                                                                                           \
  } while (0)

#define TPH_REQUIRE(cond, ...)                                                                  \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      tph_set_error(__VA_ARGS__);                                                               \
      return -2;                                                                                \
    }                                                                                           \
  } while (0)

#define TPH_LAUNCH_CHECK() TPH_HIP(hipGetLastError())

// -------------------------------------------------------------------------------------------- ctx
constexpr int TPH_MAX_NB = 16;          // trial betas per reweight pass
constexpr int TPH_RED_BLOCKS = 2048;    // 256 CUs x 8 blocks: grid cap of the streaming reductions
constexpr int TPH_RED_THREADS = 256;
constexpr int TPH_WAVE = 64;

struct tph_p2p;                         // p2p.hip: peer-mapped inboxes of the node's ranks

// A set of equally spaced arrays ("slabs") that grow WITHOUT being copied: one address range reserved for all of them, physical
// memory mapped at the front of every slab as the arrays fill (hipMemAddressReserve / hipMemCreate / hipMemMap; ctx.hip).  The
// dimension-major history u[d][cap], x[d][cap] is one set of 2 d slabs with stride cap * 8 bytes; the row-major mirror is a set
// of one.  Growing maps more memory behind what is there; outgrowing the reserved range moves the MAPPINGS to a wider range
// (no byte is copied either way).
// Every physical piece of a set has the SAME size (2 or 32 MiB, fixed when the range is reserved): on ROCm 7.2 hipMemSetAccess
// fails with "invalid value" once pieces of different sizes meet in one reservation (tools/ubench_vmm_probe.hip,
// profiles/r05_vmm_probe.json); uniform pieces map by the thousand, in ~13-16 us each, and stream as fast as hipMalloc'ed memory
// (5.98-6.03 against 6.03 TB/s read).
struct tph_vm_set {
  int device = 0;
  char* base = nullptr;
  size_t stride = 0;                    // bytes reserved per slab (a multiple of piece)
  int slabs = 0;
  size_t mapped = 0;                    // bytes mapped at the front of EVERY slab (a multiple of piece)
  size_t piece = 0;                     // bytes of one physical allocation
  std::vector<hipMemGenericAllocationHandle_t> handles;      // [piece index][slab]
  bool on() const { return base != nullptr; }
};

struct tph_ctx {
  int device = 0;
  int d = 0;
  hipStream_t stream = nullptr;
  // history, dimension-major with leading dimension `cap`
  int64_t cap = 0, size = 0;        // cap: leading dimension of u and x (rows reserved per coordinate)
  int64_t cap1 = 0;                 // rows logl and cmix are allocated for
  double *u = nullptr, *x = nullptr, *logl = nullptr, *cmix = nullptr;
  // u and x of a large history live in mapped address ranges that grow in place (TPH_OPT_HISTORY_VM): no reallocation spike,
  // no copy of tens of gigabytes when the history outgrows its reservation; small histories are plain allocations
  tph_vm_set hist_vm;
  int64_t hist_mapped = 0;          // rows of u / x backed by memory (== cap without the mapping)
  int hist_vm_mode = 1;             // TPH_OPT_HISTORY_VM: 0 never | 1 histories of >= 524 288 rows (default) | 2 always
  tph_vm_set rows_vm;               // the row-major mirror likewise
  int64_t stat_mem[4] = {0, 0, 0, 0};   // tph_history_memory: growth steps of the mapping, re-reservations, mirror drops, copies
  // row-major MIRROR of (u, x, logl) for the indexed consumers (resample gather, one-sided shuffle): a random history row is
  // 2 d + 1 scattered 8-byte reads in the dimension-major arrays (one 64-byte sector each), but ONE contiguous record here.
  // Filled lazily, up to rows_size, by tph_rows_sync (resample.hip); dropped if it cannot be allocated.
  int mc_sorted = 1;                // TPH_OPT_SORTED_DRAWS
  int cov_kernel = 0;               // TPH_OPT_COV_KERNEL: second moments at n_dim >= 16: 0 auto (= 1) | 1 register blocks | 2 MFMA
  double* rows = nullptr;
  int64_t rows_cap = 0, rows_size = 0;
  int rows_mode = 1;                // TPH_OPT_ROW_MIRROR: 1 = use the mirror (default), 0 = gather from the dimension-major arrays
  // iteration table (host mirrors + device copy of (beta_t, -logZ_t + log n_t))
  std::vector<double> beta_t, logz_t;
  std::vector<int64_t> n_local_t, n_global_t;
  double* table_dev = nullptr;  // [3][table_cap]: beta_t, logZ_t, log n_t
  double* table_host = nullptr; // pinned mirror; rows [0, table_uploaded) are on the device
  int table_cap = 0, table_uploaded = 0;
  // scratch
  double* partials = nullptr;       // streaming-reduction block partials
  size_t partials_bytes = 0;
  double* small_dev = nullptr;      // small device results (<= 4096 doubles)
  double* pinned = nullptr;         // pinned host staging (<= 4096 doubles); [4095] = tph_reweight_eval sequence word
  uint64_t eval_seq = 0;
  void* scratch = nullptr;          // big scratch (sort buffers, scans), grown on demand
  size_t scratch_bytes = 0;
  int reduce_grid = 0;              // 0 auto | blocks of the reweight reduction (experiments)
  int ml_unstaged = 0;              // 1: k_propose_ml reads its matrices from global memory (small LDS footprint, 4x the waves)
  int redraw_lanes = 0;             // 0 auto | tiles per wave of k_propose_reg (experiments)
  int propose_variant = 0;          // 0 auto | 1 one-lane LDS | 2 one-lane registers (d<=16) | 3 multi-lane (tests compare them)
  int n_simd = 1024;                // SIMDs of the device (compute units x 4): sizes one-resident-batch launches
  double* winv = nullptr;           // L^-1 per mode, formed by tph_propose when the caller passes cholinv_dev = NULL
  size_t winv_bytes = 0;
  int blocked = 0;                  // TPH_OPT_BLOCKED: d > 16 proposals through k_propose_blk + straggler pass (late iterations)
  int modes_epoch = 0;              // TPH_OPT_MODES_EPOCH: version of the caller's mode statistics (0 = unversioned)
  int blk_epoch = -1;
  const void* blk_src = nullptr;
  void* blk_buf = nullptr;          // blocked copies of L and L^-1 + the straggler flags
  size_t blk_bytes = 0;
  // stage-machine proposal kernel (propose_sm.hip): TPH_OPT_STAGED_REDRAW / _SM_LANES / _SM_THRESHOLD and its persistent buffers
  int staged = 0, sm_lanes = 0, sm_thr = 0;
  void* sm_small = nullptr;         // queue words | L in 4-row blocks | L^-1 in 8-row blocks
  size_t sm_small_bytes = 0;
  double* sm_scr = nullptr;         // per-lane columns of the rows an attempt has passed
  size_t sm_scr_bytes = 0;
  int sm_epoch = -1, sm_kernel = -1;
  const void* sm_src = nullptr;
  // screened-batch proposal kernel (propose_mf.hip): TPH_OPT_SCREEN / _MF_LANES / _MF_AUDIT and its persistent buffer
  int screen = 1, mf_lanes = 0, mf_audit = 0;
  int mf_deal = 0;                  // TPH_OPT_MF_DEAL: 0 = one global chunk cursor (default) | 1 = per-workgroup ranges with stealing | 2 = eight ranges (one per XCD)
  int mf_checked = 0;               // the screen's FP32 transcendental budget on this device: 0 not measured yet | 1 holds | -1 does not (screen off)
  void* mf_buf = nullptr;           // queue words | blocked L^-1 | FP16 pack of L + row error tables | transposed FP64 L
  size_t mf_bytes = 0;
  int mf_epoch = -1, mf_kernel = -1, mf_K = 0;
  const void* mf_src = nullptr;
  // matrix-core round kernel of the blocked path (propose_blkm.hip): TPH_OPT_BLK_MFMA and its blocked copies of L and L^-1
  int blk_mfma = 1;
  int blk_tries = 0;                // TPH_OPT_BLK_TRIES: attempts a round of the matrix-core kernel gives its failing columns in place (0 = by n_dim)
  int forms_mfma = 0;               // TPH_OPT_FORMS_MFMA: 1 = the forms behind the screened batches (tpCN, one mode) on the matrix cores (tph_blkm_forms)
  int gmm_kernel = 0;               // TPH_OPT_GMM_KERNEL: clustering E-step at n_dim >= 16: 0 auto (= 1) | 1 one lane per row | 2 matrix cores
  int blk_stage = 1;                // TPH_OPT_BLK_STAGE: 1 (default) = one-try rounds of one mode stage the panels' matrix blocks in LDS (k_propose_blkm_lds; bitwise the same draws)
  int blk_fan = 1;                  // TPH_OPT_BLK_FAN: list rounds give a straggler up to 16 attempts side by side (1 = default)
  void* bm_buf = nullptr;
  size_t bm_bytes = 0;
  int bm_epoch = -1, bm_kernel = -1, bm_K = 0;
  const void* bm_src = nullptr;
  void* mt_buf = nullptr;           // several modes: tile table, particle order, per-mode failure lists of the rounds
  size_t mt_bytes = 0;
  const void* mt_assign = nullptr;
  int64_t mt_n = 0;
  int mt_epoch = -1, mt_K = 0;
  std::vector<void*> retired;       // outgrown buffers a captured hipGraph of an earlier step may still address: freed with the ctx
  double* vv_buf = nullptr;         // small persistent buffers of tph_volume_variation (moments, factors, blocked L^-1)
  double* adapt_buf = nullptr;      // shard sums of an MCMC step's acceptance statistics (k_adapt, k_accept_sums): fixed address
  size_t adapt_bytes = 0;
  size_t vv_bytes = 0;
  uint64_t vv_seq = 0;
  // ---- communicator (tph_comm_attach): one process per GPU, this ctx holds one shard of every iteration's particles
  int rank = 0, world = 1;
  char* comm_buf = nullptr;         // caller-owned device staging block the collectives operate on (offsets into it)
  size_t comm_bytes = 0;
  tph_allreduce_fn comm_allreduce = nullptr;
  tph_allgather_fn comm_allgather = nullptr;
  void* comm_user = nullptr;
  double* blk_table = nullptr;      // block table of the last tph_cdf_global: glo[T], ghi[T], total (device)
  int blk_table_cap = 0, blk_T = 0;
  int64_t blk_rows = 0;
  tph_p2p* p2p = nullptr;           // small-message collectives over peer-mapped memory (tph_comm_p2p_attach), or NULL
  int64_t stat[5] = {0, 0, 0, 0, 0};   // tph_comm_stats: p2p exchanges, callback collectives, callback bytes, rows put, bytes put
  bool comm_active() const { return comm_allreduce != nullptr; }
};

// collectives over the attached communicator (ctx.hip); data lives in ctx->comm_buf at byte offset `off`
enum { TPH_DT_F64 = 0, TPH_DT_I64 = 1, TPH_DT_I32 = 2 };
enum { TPH_OP_SUM = 0, TPH_OP_MAX = 1, TPH_OP_MIN = 2 };
int tph_comm_require(tph_ctx* ctx, size_t bytes, const char* who);
int tph_comm_allreduce(tph_ctx* ctx, size_t off, int64_t count, int dtype, int op);
int tph_comm_allgather(tph_ctx* ctx, size_t send_off, size_t recv_off, int64_t count, int dtype);
int tph_comm_allgather_cb(tph_ctx* ctx, size_t send_off, size_t recv_off, int64_t count, int dtype);   // through the callback whatever the size (host-paced callers)
// p2p.hip: one single-block exchange kernel on the ctx stream instead of the callback, for messages of <= 32 KB
bool tph_p2p_fits(const tph_ctx* ctx, int64_t count, int dtype);
int tph_p2p_exchange(tph_ctx* ctx, const void* src, void* dst, int64_t count, int dtype, int op /* < 0: all-gather */);
void tph_p2p_release(tph_ctx* ctx);
int tph_blocks(tph_ctx* ctx, int64_t n, int* T, int64_t* rows);   // equal-sized iteration blocks of the local history

// ---- the canonical partition: what makes a run on G GPUs the SAME floating-point computation as the run on one ---------------
// Every reduction over the particles of an iteration (the reweight triples, the cumulative weights, the moments of the proposal
// fit, the acceptance sums of an MCMC step) is formed per VIRTUAL SHARD and the V per-shard results are folded in shard order.
// A virtual shard is a fixed range of n_particles / V particle slots -- V depends on n_particles alone (tph_vshards_for), never
// on the number of GPUs --; a rank of a G-GPU run owns V / G consecutive ones, the one-GPU run owns all V, and a shard's partial
// result is computed by a procedure that sees only the shard's rows.  So the summation tree of every global quantity is the same
// for every G that divides V: bitwise the same run (tests/test_distributed.py: worlds 1, 2, 3, 4).  History rows of virtual
// shard v: for every committed iteration t the `nv` rows from t * n_loc + v * nv ("piece" (t, v)).
struct tph_part {
  int T;            // committed iterations (pieces per virtual shard); 1 when the history has no block structure
  int64_t n_loc;    // rows per iteration held by this rank
  int vl;           // virtual shards of this rank
  int V;            // virtual shards of the whole run (vl * world when canonical)
  int64_t nv;       // rows of one piece
  bool canonical;   // V was chosen from n_particles alone and divides evenly: results do not depend on the number of ranks
};
__host__ __device__ static inline int tph_vshards_inline(long long n_global) {
  const int cand[9] = {48, 16, 12, 8, 6, 4, 3, 2, 1};
  for (int i = 0; i < 9; ++i)
    if (n_global > 0 && n_global % ((long long)cand[i] * 256) == 0) return cand[i];
  return 1;
}
int tph_vshards_for(int64_t n_global);                 // the largest of 48, 16, 12, 8, 6, 4, 3, 2, 1 with n_global % (256 V) == 0
tph_part tph_partition(const tph_ctx* ctx, int64_t n); // partition of the first n == ctx->size history rows (else: one piece)
int tph_partials_reserve(tph_ctx* ctx, size_t bytes);  // ctx->partials of at least that many bytes

int tph_scratch_reserve(tph_ctx* ctx, size_t bytes);
void tph_warm_cluster(hipStream_t stream);
void tph_warm_modes(hipStream_t stream);
void tph_warm_mutate(hipStream_t stream);
void tph_warm_p2p(hipStream_t stream);
void tph_warm_propose_blkm(hipStream_t stream);
void tph_warm_propose_mf(hipStream_t stream);
void tph_warm_propose_sm(hipStream_t stream);
void tph_warm_resample(hipStream_t stream);
void tph_warm_reweight(hipStream_t stream);
void tph_warm_student(hipStream_t stream);
int tph_cdf_plain(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev, double* cdf_dev);   // resample.hip: the plain scan
// mapped, growing arrays (ctx.hip)
int tph_vm_reserve(tph_vm_set* v, int device, int slabs, size_t stride_bytes, size_t piece_bytes);
int tph_vm_grow(tph_vm_set* v, size_t want_bytes_per_slab);            // 0 ok | 1 out of memory (nothing changed) | -1 error
int tph_vm_restride(tph_vm_set* v, size_t new_stride_bytes, hipStream_t stream);
void tph_vm_release(tph_vm_set* v);
size_t tph_vm_piece(int device, size_t first_bytes_per_slab);            // piece size for a set (0: the device cannot map memory this way)
void tph_rows_drop(tph_ctx* ctx);                                       // resample.hip: give the mirror's memory back (it is a cache)
const double* tph_rows_sync(tph_ctx* ctx);          // resample.hip: mirror up to date for rows [0, size), or NULL (not in use)
int tph_tri_inv(tph_ctx* ctx, const double* chol_dev, int K, double* winv_dev);   // modes.hip
// propose_sm.hip: the whole proposal of a redraw-dominated step at 16 < d <= 100, one mode (pending moves, forms, u')
int tph_propose_sm(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                   const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                   const double* ctl, int64_t item0, double* up, double* maha_u, double* maha_up, uint8_t* pend);
// propose_mf.hip: the same step with the attempts screened on the matrix cores and only the survivors evaluated in FP64
int tph_propose_mf(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                   const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                   const double* ctl, int64_t item0, double* up, double* maha_u, double* maha_up, uint8_t* pend);
// propose_blkm.hip: one round of the blocked path with both triangular products on the FP64 matrix cores
int tph_blkm_tries(const tph_ctx* ctx);      // attempts per round of the matrix-core kernel (TPH_OPT_BLK_TRIES resolved)
int tph_blkm_round(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                   const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                   const double* ctl, int64_t item0, double* up, double* mu_, double* mup, uint8_t* pend, const int32_t* cnt_in,
                   const int32_t* rows_in, int att, int32_t* cnt_out, int32_t* rows_out, const int32_t* att_in, int32_t* att_out);
// several modes: every round over mode-pure tiles; the particles still out of bounds are left in one list for the caller
int tph_blkm_multi(tph_ctx* ctx, int kernel, double* u, const int32_t* assign, int64_t n, int64_t ld, int K, const double* means,
                   const double* chol, const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed,
                   uint32_t tick0, const double* ctl, int64_t item0, double* up, double* mu_, double* mup, uint8_t* pend, int rounds,
                   const int32_t** todo_cnt, const int32_t** todo_rows, const int32_t** todo_att,       // todo_att[mode]: where its list goes on (or NULL)
                   const int32_t** per_mode = nullptr);      // != NULL: no concatenated list; per_mode[0..2] = counts[K], the list array, the modes' offsets into it
// the same over a device-side list of particles (count + rows), from attempt att0: the straggler pass behind the blocked kernel
int tph_propose_mf_list(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                        const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                        const double* ctl, int64_t item0, double* up, double* maha_up, const int32_t* todo_cnt, const int32_t* todo_rows,
                        int att0, const int32_t* att0_dev = nullptr,       // att0_dev: the first attempt read from the device instead
                        int queue_zeroed = 0);                             // 1: the caller zeroed tph_mf_queue_words() on the stream already
bool tph_mf_selftest(tph_ctx* ctx);                  // the device meets the screen's FP32 budget (measured once per context)
bool tph_mf_screen(tph_ctx* ctx);                    // TPH_OPT_SCREEN on, n_dim in range AND the device passed the screen's self-test (run once)
// the screened kernel's queue block: 16 counters (64-bit), then one chunk cursor per workgroup (<= 256); in 32-bit words:
constexpr int TPH_MF_QWORDS = 2 * (16 + 256), TPH_MF_QCURSORS = 32;
unsigned int* tph_mf_queue_words(tph_ctx* ctx);
// several proposal modes: the particles grouped by mode (propose_blkm.hip; device pointers into ctx-owned memory: order[n], mstart[K], mcount[K])
int tph_mode_lists(tph_ctx* ctx, const int32_t* assign, int64_t n, int K, const int32_t** order, const int32_t** mstart, const int32_t** mcount);
// screened batches mode by mode over those lists (a whole redraw-dominated step), and over the modes' failure lists of the matrix-core rounds
int tph_propose_mf_modes(tph_ctx* ctx, int kernel, double* u, const int32_t* assign, int64_t n, int64_t ld, int K, const double* means,
                         const double* chol, const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed,
                         uint32_t tick0, const double* ctl, int64_t item0, double* up, double* maha_u, double* maha_up, uint8_t* pend);
int tph_propose_mf_mode_lists(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, int K, const double* means, const double* chol,
                              const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                              const double* ctl, int64_t item0, double* up, double* maha_up, const int32_t* cnts, const int32_t* rows,
                              const int32_t* offs, int att0, const int32_t* atts);      // the 32 work-queue words of the screened kernel (allocates its buffers; NULL on error)

// --------------------------------------------------------------------------------- device helpers
#if defined(__HIPCC__)

// np.logaddexp (numpy/core/src/npymath: npy_logaddexp) restated
// ---- lean FP64 elementary functions for the proposal kernels (gfx950) ----
// The library log/sqrt/division are IEEE-complete (denormal scaling, special cases, < 1 ulp through double-double steps):
// 60-70 VALU instructions for log, ~14 each for sqrt and a/b.  The proposal kernels are VALU-issue-bound and call them on
// arguments whose range is known (uniforms in (0,1], -2 log u, Gamma candidates), so they use these forms instead:
// hardware seed (v_rcp_f64 / v_rsq_f64) + Newton steps, and the fdlibm log kernel.  Errors stay below 1 ulp (log) /
// 1 ulp (sqrt, division) on normal arguments; NaN/negative inputs are not handled (none occur: see the call sites).
__device__ __forceinline__ double tph_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  e = fma(-b, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double tph_div(double a, double b) {       // b normal, a/b in range
  const double r = tph_rcp(b);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double tph_sqrt(double x) {                // 0 <= x, normal or zero
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  g = fma(fma(-g, g, x), h, g);
  return x == 0.0 ? 0.0 : g;
}
// log of a positive normal double (fdlibm e_log.c kernel: x = 2^k (1+f), sqrt(1/2) <= 1+f < sqrt(2))
__device__ __forceinline__ double tph_log(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);         // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool lowhalf = m < 0.70710678118654752440;
  m = lowhalf ? m + m : m;
  k = lowhalf ? k - 1 : k;
  const double f = m - 1.0;
  const double s = tph_div(f, 2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
  return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// log(1 + e) for 0 <= e <= 1 through the lean log: u = fl(1 + e), the rounding error c = e - (u - 1) of that sum is exact, and
// log(u + c) = log(u) + c / u to second order in c / u < 2^-53 (the library log1p is ~110 FP64 instructions: with the exp in front
// of it K1 was bound by VALU issue at 0.43 of the HBM rate)
__device__ __forceinline__ double tph_log1p_unit(double e) {
  const double u = 1.0 + e;
  return tph_log(u) + tph_div(e - (u - 1.0), u);
}
// np.logaddexp (numpy/_core/src/npymath/npy_math_internal.h.src: npy_logaddexp): max + log1p(exp(-|x - y|))
__device__ __forceinline__ double tph_logaddexp(double x, double y) {
  if (x == y) return x + 0.6931471805599453094;  // also equal infinities
  double t = x - y;
  if (t > 0) return x + tph_log1p_unit(exp(-t));
  if (t <= 0) return y + tph_log1p_unit(exp(t));
  return t;  // NaN
}

// ---- Philox4x32-10 (twin of oracle/philox.py) ----
struct tph_u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ tph_u4 tph_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return tph_u4{c0, c1, c2, c3};
}

constexpr uint32_t TPH_TAG_PRIOR = 1, TPH_TAG_NORMAL = 2, TPH_TAG_GAMMA = 3, TPH_TAG_ACCEPT = 4,
                   TPH_TAG_RESAMPLE = 5, TPH_TAG_UPSAMPLE = 6, TPH_TAG_REPAIR = 7;

__device__ __forceinline__ double tph_k53(uint32_t hi, uint32_t lo) {
  // k = (hi >> 5) * 2^26 + (lo >> 6) < 2^53, formed exactly from two 32-bit conversions and one FMA (the generic
  // u64 -> f64 conversion costs ~50 SIMD cycles more per value)
  return fma((double)(hi >> 5), 67108864.0, (double)(lo >> 6));
}

// sin(pi t), cos(pi t) for 0 <= t <= 2 (the Box-Muller angle 2 u): nearest quarter turn by an EXACT reduction (t - q/2 is
// exact in FP64), then the fdlibm kernels on |x| <= pi/4 -- 35 VALU instructions where the library sincospi, which also
// serves huge and special arguments, spends 61.  Errors < 1 ulp of 1 (checked against long-double references on 10^7 angles);
// a normal moves by at most a few 1e-16 relative to the library form.
__device__ __forceinline__ void tph_sincospi(double t, double& sn, double& cs) {
  const double qf = __builtin_rint(t + t);                     // 0 .. 4
  const int q = (int)qf;
  const double x = fma(-0.5, qf, t) * 3.14159265358979311600;  // (t - q/2) pi, |x| <= pi/4
  const double z = x * x;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                 -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double s0 = fma(x * z, ps, x);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
  sn = (q & 2) ? -s1 : s1;
  cs = ((q + 1) & 2) ? -c1 : c1;
}

struct tph_rng {
  uint32_t k0, k1, tick, tag, item;
  __device__ tph_rng(uint64_t seed, uint32_t tick_, uint32_t tag_, uint64_t item_)
      : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), tick(tick_), tag(tag_), item((uint32_t)item_) {}
  // two U[0,1)
  __device__ __forceinline__ void uniform2(uint32_t draw, double& a, double& b) const {
    tph_u4 r = tph_philox(item, draw, tick, tag, k0, k1);
    a = tph_k53(r.x, r.y) * 0x1.0p-53;
    b = tph_k53(r.z, r.w) * 0x1.0p-53;
  }
  // two N(0,1), Box-Muller with u1 in (0,1]
  __device__ __forceinline__ void normal2(uint32_t draw, double& z0, double& z1) const {
    tph_u4 r = tph_philox(item, draw, tick, tag, k0, k1);
    double u1 = (tph_k53(r.x, r.y) + 1.0) * 0x1.0p-53;
    double u2 = tph_k53(r.z, r.w) * 0x1.0p-53;
    double rad = tph_sqrt(-2.0 * tph_log(u1));
    double s, c;
    tph_sincospi(2.0 * u2, s, c);
    z0 = rad * c;
    z1 = rad * s;
  }
  // the candidate of one Marsaglia-Tsang attempt from ONE Philox call: a normal (53-bit radius uniform, 32-bit angle) and
  // the log of a (0, 1] uniform with 32 bits (twin: oracle/philox.py gamma_mt)
  __device__ __forceinline__ void gamma_candidate(uint32_t att, double& x, double& logu) const {
    tph_u4 r = tph_philox(item, att, tick, tag, k0, k1);
    const double u1 = (tph_k53(r.x, r.y) + 1.0) * 0x1.0p-53;
    const double rad = tph_sqrt(-2.0 * tph_log(u1));
    double s, c;
    tph_sincospi((double)r.z * 0x1.0p-31, s, c);
    x = rad * c;
    logu = tph_log(((double)r.w + 1.0) * 0x1.0p-32);
  }
};

// RNG tick of one MCMC-step launch.  With a device-resident step-control block (tph_adapt's state_dev, see
// tempest_hip.h) the tick is  tick + state[7] + 2 * state[0]: it advances by two per completed step ON THE DEVICE, so
// that one captured hipGraph of a step can be replayed without new kernel arguments (and re-used by the next PS
// iteration after rewriting the block); without one (ctl == NULL) it is the plain by-value tick.
struct tph_stepctl {
  uint32_t tick;
  const double* ctl;
  __device__ __forceinline__ operator uint32_t() const {
    return ctl ? tick + (uint32_t)(unsigned long long)ctl[7] + 2u * (uint32_t)ctl[0] : tick;
  }
  __device__ __forceinline__ bool done() const { return ctl && ctl[1] != 0.0; }
  // After the first step of a run maha_u[i] already holds (u_i - mu)^T Sigma^-1 (u_i - mu): tph_accept copies the
  // proposal's value for accepted rows, rejected rows keep theirs.  The proposal kernels then read it instead of
  // recomputing a d x d quadratic form per particle (same formula on the same doubles: bit-identical).
  __device__ __forceinline__ bool carry() const { return ctl && ctl[0] > 0.0; }
};
// the forms of a finished step on the matrix cores (propose_blkm.hip)
int tph_blkm_forms(tph_ctx* ctx, const double* up, int64_t n, int64_t ld, const double* means, const double* chol, const double* winv,
                   double* maha, tph_stepctl tick, const unsigned long long* queue, const int32_t* todo_cnt, const int32_t* todo_rows,
                   const int32_t* todo_off);

// Gamma(shape,1), Marsaglia-Tsang; attempt a uses draw a (one Philox call: normal and uniform); shape<1 boosted.
__device__ inline double tph_gamma_mt(const tph_rng& g, double shape, int first_attempt = 0) {
  const int max_attempts = 64;
  bool boost = shape < 1.0;
  double a = boost ? shape + 1.0 : shape;
  double d = a - 1.0 / 3.0;
  double c = tph_rcp(tph_sqrt(9.0 * d));
  double out = d;
  for (int att = first_attempt; att < max_attempts; ++att) {
    double x, logu;
    g.gamma_candidate((uint32_t)att, x, logu);
    double v = 1.0 + c * x;
    v = v * v * v;
    if (v > 0.0 && logu < 0.5 * x * x + d - d * v + d * tph_log(v)) {
      out = d * v;
      break;
    }
  }
  if (boost) {
    double ub, u1;
    g.uniform2(2 * max_attempts, ub, u1);
    ub += 0x1.0p-53;
    out *= pow(ub, 1.0 / shape);
  }
  return out;
}

// The scalar part of the step's adaptation -- per-cluster mean alpha (mcmc.py:180-186), sigma update (:281-288 / :320-323),
// adaptive step count and stopping rule (:104-140, 192-194, with the `sigmas[:n_nonempty]` weighting quirk), returned statistics
// (:196-197) -- by ONE thread, on `sigmas[K]` and the step-control block `state` (tempest_hip.h), which may be device memory
// (k_adapt) or a workgroup's own copy (the persistent step kernel of the HIP-callback plugins: every workgroup adapts for
// itself from the same sums, with the same arithmetic).  `mailbox` (or NULL): the step's record into pinned host memory.
__device__ inline void tph_adapt_scalar(int kernel, const double* __restrict__ sums, const double* __restrict__ counts, int K,
                                        double n_global, int d, int n_steps, int n_max, double* __restrict__ sigmas,
                                        double* __restrict__ state, double* __restrict__ mailbox, int slots) {
  const int iteration = (int)state[0] + 1;
  const double sigma_0 = 2.38 / sqrt((double)d);
  const double rate = 1.0 / (double)(iteration + 1);
  double alpha_tot = 0.0;
  for (int c = 0; c < K; ++c) {
    alpha_tot += sums[1 + c];
    if (counts[c] > 0.0) {
      double mean_accept = sums[1 + c] / counts[c];
      double s = sigmas[c] + rate * (mean_accept - 0.234);
      if (kernel == TPH_KERNEL_TPCN) s = fmin(fmax(s, 0.0), fmin(sigma_0, 0.99));
      sigmas[c] = s;
    }
  }
  const double acc = sums[0] / n_global;
  // weighted average of sigmas[:n_nonempty] with the non-empty cluster sizes (mcmc.py:107-117)
  double wsum = 0.0, wsig = 0.0, smean = 0.0;
  int q = 0;
  for (int c = 0; c < K; ++c) {
    smean += sigmas[c];
    if (counts[c] > 0.0) { wsig += sigmas[q] * counts[c]; wsum += counts[c]; ++q; }
  }
  const double weighted_sigma = wsig / wsum;
  const double n_min = (double)n_steps * d;
  double ratio = sigma_0 / fmax(1e-6, weighted_sigma);
  double n_adapt = (double)n_steps * d * (0.234 / fmax(0.01, acc)) * (ratio * ratio);
  double n_final = fmin(fmax(n_min, n_adapt), (double)n_max * d);
  long long n_int = (long long)n_final;  // int() truncation
  state[0] = (double)iteration;
  state[1] = (iteration >= n_int) ? 1.0 : 0.0;
  state[2] = acc;
  state[3] = alpha_tot / n_global;
  state[4] = (smean / K) / sigma_0;
  state[5] = (double)n_int;
  if (mailbox) {
    // the step's record straight into pinned host memory: the host polls the sequence field instead of putting a
    // device-to-host copy (and its cross-engine barrier) between two steps of the stream
    double* rec = mailbox + (size_t)(iteration % slots) * 8;
    for (int j = 0; j < 6; ++j) rec[j] = state[j];
    rec[6] = state[8];                                   // mean redraw attempts seen by the d > 16 proposal kernel (0: not reported)
    __threadfence_system();
    __hip_atomic_store(rec + 7, (double)iteration, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// tpCN's proposal-density ratio in the Metropolis factor (mcmc.py:251-279): -A + B with A, B = -1/2 (d + nu) log(1 + m / nu) at
// u' and u.  The lean log (both arguments are >= 1 for finite forms); a NaN / inf form goes through the library log so that
// alpha stays NaN -> 0 as in the reference.  One definition for k_accept and the HIP-callback plugins (bit-identical paths).
__device__ __forceinline__ double tph_tpcn_factor(double d_plus_nu, double nu, double m_u, double m_up) {
  const double au = 1.0 + tph_div(m_u, nu), ap = 1.0 + tph_div(m_up, nu);
  const bool plain = au < 1e300 && ap < 1e300;
  const double B = -0.5 * d_plus_nu * (plain ? tph_log(au) : log(au));
  const double A = -0.5 * d_plus_nu * (plain ? tph_log(ap) : log(ap));
  return -A + B;
}

// ---- boundary conditions of the proposals (mcmc.py:326-411) and the cap of the redraw-until-in-bounds loop ----
__device__ __forceinline__ double bc_periodic(double v) {  // numpy `v % 1.0` (npy_divmod)
  double r = fmod(v, 1.0);
  if (r != 0.0) { if (r < 0.0) r += 1.0; } else r = 0.0;
  return r;
}
__device__ __forceinline__ double bc_reflective(double v) {  // mcmc.py:357-364
  double nr = floor(v);
  double rem = v - nr;
  return fmod(nr, 2.0) == 0.0 ? rem : 1.0 - rem;
}
constexpr int PROP_MAX_ATTEMPTS = 256;   // then the current point is proposed (the reference loops on)

// ---- shared pieces of the register proposal kernels (mutate.hip, user_plugin.hip.in) ----
// tph_opaque: launders a wave-uniform pointer so that the scalar loads through it stay where they are written (see k_propose_reg)
template <class T>
__device__ __forceinline__ const T* tph_opaque(const T* p) {
  // an opaque ZERO offset in an SGPR: the pointer keeps its address space and provenance (the loads stay scalar
  // s_load's of a kernel argument), but their address now depends on a value defined here, so they cannot be hoisted
  // above this point.  (Laundering the pointer itself turns every load behind it into a flat VECTOR load.)
  int off = 0;
  asm volatile("" : "+s"(off));
  return p + off;
}

// |W (v - mu)|^2, W lower-triangular [D][D] row-major
template <int D, bool UNIFORM>
__device__ __forceinline__ double maha_w(const double* __restrict__ W, const double (&dv)[D]) {
  double m = 0.0;
#pragma unroll
  for (int r = 0; r < D; ++r) {
    if (UNIFORM && (r == D / 2 || r == (3 * D) / 4)) W = tph_opaque(W);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j <= r; ++j) acc = fma(W[r * D + j], dv[j], acc);
    m = fma(acc, acc, m);
  }
  return m;
}

// Zeroing a few device words on the stream: a KERNEL, not hipMemsetAsync -- inside a captured hipGraph the memset node of
// this ROCm release is not reliably ordered against the kernel nodes around it (the work-queue words of propose_sm.hip came
// back holding stale data on the second replay of a captured step; the same launch sequence issued eagerly was fine).
static __global__ void k_zero_words(unsigned int* __restrict__ p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
static __global__ void k_zero_words2(unsigned int* __restrict__ p, int n, unsigned int* __restrict__ q, int m) {      // two blocks of words, one launch
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
  for (int i = threadIdx.x; i < m; i += blockDim.x) q[i] = 0u;
}

// ---- wave / block reductions (wave64) ----
__device__ __forceinline__ double tph_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double tph_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// block-wide sum; result valid in thread 0.  `sh` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double tph_block_sum(double v, double* sh) {
  v = tph_wave_sum(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  if (wid == 0) {
    int nw = (blockDim.x + 63) >> 6;
    v = lane < nw ? sh[lane] : 0.0;
    v = tph_wave_sum(v);
  }
  return v;
}
// NV block-wide sums at once (256 threads): out[k] = the value tph_block_sum(acc[k], .) leaves in thread 0, bit for bit -- the
// same shuffle tree per wave and (w0 + w2) + (w1 + w3) over the four wave sums -- behind ONE barrier pair instead of NV of them
// (a register-accumulator kernel with 55 or 66 sums spent ~7 us per block in its barriers).  `sh` holds 4 NV doubles.
template <int NV>
__device__ __forceinline__ void tph_block_sum_many(double (&acc)[NV], double* __restrict__ sh, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = tph_wave_sum(acc[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) sh[wid * NV + k] = acc[k];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < NV; k += 256) out[k] = (sh[k] + sh[2 * NV + k]) + (sh[NV + k] + sh[3 * NV + k]);
}
__device__ __forceinline__ double tph_block_max(double v, double* sh) {
  v = tph_wave_max(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  if (wid == 0) {
    int nw = (blockDim.x + 63) >> 6;
    v = lane < nw ? sh[lane] : -DBL_MAX;
    v = tph_wave_max(v);
  }
  return v;
}

#endif  // __HIPCC__

static inline int tph_grid_for(int64_t n, int threads, int per_thread = 1, int cap = TPH_RED_BLOCKS) {
  int64_t b = (n + (int64_t)threads * per_thread - 1) / ((int64_t)threads * per_thread);
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}
