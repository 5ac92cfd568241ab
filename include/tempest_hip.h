/*
 * tempest_hip.h -- C ABI of libtempest_hip.so: the MI355X (gfx950) persistent-SMC hot path.
 *
 * Drop-in boundary for the numerics of minaskar/tempest v0.2.1 (reference file:line cited per
 * entry point, relative to the reference tree).  The reference is pure Python/NumPy and has no FFI
 * of its own; these are the calls its four step plugins (tempest/steps/ *.py) and its StateManager
 * (tempest/state_manager.py) would bind if their NumPy bodies were replaced (INTEGRATION.md shows
 * the ctypes stubs).  Host code above this boundary: tempest_amd/ (Python, mirrors the reference's
 * Sampler / steps API).
 *
 * Conventions
 *  - extern "C"; every function returns 0 on success, <0 on error; tph_last_error() returns the
 *    thread-local message of the last failure.
 *  - tph_ctx owns the persistent particle history and all scratch; every other pointer is a
 *    BORROWED raw pointer valid for the call only.  `_dev` = device pointer, `_host` = host pointer.
 *  - Particle arrays are structure-of-arrays, dimension-major: a[j*ld + i] is coordinate j of
 *    particle i (ld >= n).  All reals are FP64, indices int64, labels int32.
 *  - All work is enqueued on the hipStream_t given to tph_ctx_create / tph_set_stream (pass the
 *    caller framework's current stream so its own kernels are ordered with ours).  Functions whose
 *    outputs are `_host` synchronise that stream before returning; all others are asynchronous.
 *  - One ctx per device, not re-entrant.
 *  - Randomness is counter-based Philox4x32-10: (seed, tick, tag, item, draw) -> bits, so results do
 *    not depend on launch geometry or on how particles are sharded (item = global particle index =
 *    item0 + local index).  The reference's global NumPy RNG (mcmc.py:169,236,243,307 ...) cannot be
 *    reproduced on a GPU; parity for RNG-consuming steps is as pure functions of the draws.
 */
#ifndef TEMPEST_HIP_H
#define TEMPEST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tph_ctx tph_ctx;

#define TPH_VERSION 100

/* proposal kernels (tempest/config.py:135-137: only these two exist in the reference) */
#define TPH_KERNEL_TPCN 0
#define TPH_KERNEL_RWM 1

/* per-dimension boundary flags (tempest/mcmc.py:326-411) */
#define TPH_BC_STRICT 0
#define TPH_BC_PERIODIC 1
#define TPH_BC_REFLECTIVE 2

/* history array keys for tph_history_read / tph_history_ptr */
#define TPH_KEY_U 0
#define TPH_KEY_X 1
#define TPH_KEY_LOGL 2
#define TPH_KEY_LOGMIX 3

const char* tph_last_error(void);
int tph_version(void);

/* ---- context ------------------------------------------------------------------------------- */
int tph_ctx_create(int device, int n_dim, int64_t capacity_hint, void* hip_stream, tph_ctx** out);
int tph_ctx_destroy(tph_ctx* ctx);
int tph_set_stream(tph_ctx* ctx, void* hip_stream);
int tph_synchronize(tph_ctx* ctx);
/* loads every code object of the library now (one empty launch per translation unit, then a stream synchronisation) instead of
 * at the first use of each kernel family in the middle of a run's first iterations; safe to call from a start-up thread */
int tph_warmup(tph_ctx* ctx);
/* TPH_OPT_PROPOSE_VARIANT selects the proposal kernel: 0 = automatic (registers for n_dim <= 16, multi-lane above, blocked
 * when TPH_OPT_BLOCKED is set, row walker when TPH_OPT_STAGED_REDRAW is), 1 = one lane per particle with LDS columns (any n_dim), 2 = one lane per particle in
 * registers (n_dim <= 16), 3 = several lanes per particle with the matrices staged in LDS, 4 = blocked + straggler pass
 * (16 < n_dim <= 100, one mode), 5 = row walker (16 < n_dim <= 100, one mode), 6 = screened batches (16 < n_dim <= 112, one
 * mode); the parity tests run every variant */
#define TPH_OPT_PROPOSE_VARIANT 0
/* TPH_OPT_REDUCE_GRID: 0 = automatic grid of the reweight reduction, > 0 = that many blocks (experiments) */
#define TPH_OPT_REDUCE_GRID 1
/* TPH_OPT_REDRAW_LANES: 0 = automatic, t > 0 = 64-particle tiles per wave of the d <= 16 proposal kernel (its redraw work
 * lists are per wave: more tiles = longer lists = fuller redraw passes, fewer waves) */
#define TPH_OPT_REDRAW_LANES 2
/* TPH_OPT_ML_UNSTAGED: 1 = the d > 16 proposal kernel reads Sigma^-1 and L from global memory instead of staging them in LDS
 * (a quarter of the LDS footprint, four times the resident waves: the right trade while most attempts are redraws that stop
 * after a few rows); 0 = staged (default) */
#define TPH_OPT_ML_UNSTAGED 3
/* TPH_OPT_BLOCKED: R >= 1 = n_dim > 16, one mode: attempt 0 of every particle by the blocked kernel (lane = particle, matrix
 * operands through the scalar cache: the fast form of a step that is one attempt), then R - 1 further ROUNDS of it over the
 * particles still out of bounds (attempt k of each, compacted lists; R <= 24), and whoever is left by the multi-lane kernel
 * from attempt R on; 0 (default) = multi-lane kernel for everything.  The host switches on the redraw probe and sizes R from
 * it (a few attempts per particle: ~3 x the mean), like TPH_OPT_ML_UNSTAGED. */
#define TPH_OPT_BLOCKED 4
/* TPH_OPT_MODES_EPOCH: v > 0 = version of the mode statistics AND of the assignments passed to tph_propose: the blocked copies
 * of L and L^-1, the screening packs and -- with several modes -- the grouping of the particles by mode (tile table, per-mode
 * order) are rebuilt only when it (or the chol / assign_dev pointer, or n) changes.  A caller who keeps v > 0 must bump it
 * whenever the CONTENTS of chol_dev, cholinv_dev or assign_dev change, also at an unchanged address (a caching allocator hands
 * the same address back routinely).  0 (default) = everything rebuilt on every call. */
#define TPH_OPT_MODES_EPOCH 5
/* TPH_OPT_ROW_MIRROR: 1 (default) = tph_gather and tph_resample_put_global read a row-major mirror of (u, x, logl) that the
 * library keeps beside the dimension-major history (filled lazily, + (2 n_dim + 1) * 8 bytes per row; dropped by itself when
 * it cannot be allocated): a gathered row is one contiguous record instead of 2 n_dim + 1 scattered sectors; 0 = gather from
 * the dimension-major arrays */
#define TPH_OPT_ROW_MIRROR 6
/* TPH_OPT_COV_KERNEL: centred second moments at 16 <= n_dim <= 100: 0 = automatic (= 1), 1 = 4 x 4 register blocks per thread
 * (VALU), 2 = FP64 matrix cores (v_mfma_f64_16x16x4: the product is a SYRK; measured no faster -- both are bound by the fill of
 * the staged tile); the parity tests run both */
#define TPH_OPT_COV_KERNEL 7
/* TPH_OPT_SORTED_DRAWS: 1 (default) = tph_multinomial_counts with >= 2^21 draws generates them as 53-bit integers, sorts them
 * and merges them against the cdf (the counts do not depend on the order of the draws; one 8-byte device-to-host read of the
 * kept count sizes the sort); 0 = one indexed lookup per draw in draw order; a value > 1 = that many draws as the threshold
 * instead of 2^21 (tests).  Same counts either way. */
#define TPH_OPT_SORTED_DRAWS 8
/* TPH_OPT_STAGED_REDRAW: 1 = 16 < n_dim <= 100, one mode, steps that are mostly REDRAWS (mcmc.py:239-249: early iterations of a
 * high-dimensional run): the row-walker kernel (propose_sm.hip: a lane per attempt with its normals in its column of a tile,
 * attempts advance four rows at a time and stop at their first out-of-bounds coordinate, several attempts of a particle in
 * flight, the first in-bounds one in attempt order wins).  0 = those steps through the multi-lane kernel.  The host (mcmc.py:
 * StepEngine) switches on the redraw probe, like TPH_OPT_BLOCKED.  TPH_OPT_SM_LANES: log2 of the lanes per particle (0 = by
 * size); TPH_OPT_SM_THRESHOLD: rows of an attempt's normals kept in LDS, rounded up to 16 (0 = 32; the rows beyond live in
 * global scratch -- fewer rows, more waves per CU). */
#define TPH_OPT_STAGED_REDRAW 9
#define TPH_OPT_SM_LANES 10
#define TPH_OPT_SM_THRESHOLD 11
/* TPH_OPT_SCREEN: 1 (default) = the redraw-dominated steps of TPH_OPT_STAGED_REDRAW run as SCREENED BATCHES (propose_mf.hip,
 * 16 < n_dim <= 112): a wave holds 64 attempts as columns of 16 x 16 tiles and walks them in lockstep through panels of 16
 * rows on the matrix cores (FP16 copy of L, FP32 Box-Muller pairs from the same Philox blocks); an attempt is dropped only
 * when a coordinate is out of bounds by more than a rigorous bound of the low-precision error, every attempt that is not
 * dropped is evaluated in FP64 by the arithmetic of the other kernels, and the first of those in bounds, in attempt order, is
 * the proposal -- the proposal of the sequential loop of mcmc.py:239-249, equal to the row walker's bit for bit.  0 = the FP64
 * row walker.  TPH_OPT_MF_LANES: log2 of the attempts a particle keeps in flight (2..6; 0 = by size).  TPH_OPT_MF_AUDIT: 1 =
 * every dropped attempt is ALSO evaluated in FP64 and contradictions are counted (tph_bench_mf_counters word 3; tests). */
#define TPH_OPT_SCREEN 12
#define TPH_OPT_MF_LANES 13
#define TPH_OPT_MF_AUDIT 14
/* TPH_OPT_BLK_MFMA: 1 (default) = the rounds of the blocked path (TPH_OPT_BLOCKED, 16 < n_dim <= 112) run on the FP64 matrix
 * cores (propose_blkm.hip: a wave per 16 particles, L z and |L^-1 (u' - mu)|^2 as v_mfma_f64_16x16x4_f64 tiles, the normals
 * generated straight into the operand layout); 0 = lane = particle with the matrix through the scalar cache (k_propose_blk). */
#define TPH_OPT_BLK_MFMA 15
/* TPH_OPT_BLK_TRIES: t (1..3; 0 = default: 2 up to n_dim 32, 1 above) = a round of the matrix-core kernel evaluates up to t consecutive attempts of its failing
 * columns in place (round k: attempts k t .. k t + t - 1) before a particle is listed for the next round / the straggler pass */
#define TPH_OPT_BLK_TRIES 16
/* TPH_OPT_BLK_FAN: 1 (default) = a round of the matrix-core kernel over a LIST gives every listed particle G consecutive
 * attempts side by side (G = the largest power of two <= 16 with G x list <= n_particles / 2; the first in bounds in attempt order
 * is the proposal), with the screened kernel as the straggler pass (TPH_OPT_SCREEN); 0 = one attempt per particle and try;
 * 2 / 3 = the fanned-out list may fill all / a quarter of the ensemble's columns (experiments). */
#define TPH_OPT_BLK_FAN 17
/* TPH_OPT_HISTORY_VM: where u and x of the history live.  1 (default) = a history asked to hold >= 16 GB of u and x -- or a
 * plain one whose next doubling would not fit comfortably -- lives in a MAPPED address range (hipMemAddressReserve / hipMemCreate
 * / hipMemMap, uniform pieces of 2 or 32 MiB) that grows in place: memory is mapped behind what is there, an eighth at a time,
 * nothing is reallocated or copied, so a history of 100 GB grows without a 2-3 x spike (tempest/state_manager.py:356-416
 * appends without bound); smaller ones are plain allocations grown by doubling.  0 = plain allocations always; 2 = mapped
 * always (tests).  The row-major mirror follows the history (within a fifth of the device, and never below 15 % free memory:
 * beyond that it is given back); logl and the cached mixture (16 bytes per row) are always plain. */
#define TPH_OPT_HISTORY_VM 18
/* TPH_OPT_MF_DEAL: how the screened kernel deals its particles (chunks of 4) to the waves: 0 (default) = one global cursor;
 * 1 = per-workgroup contiguous ranges with stealing; 2 = eight ranges, one per XCD's workgroups (the 8 particles behind one
 * 64-byte sector of u then go through one L2: half the HBM reads, not faster -- csrc/propose_mf.hip).  Same proposals. */
#define TPH_OPT_MF_DEAL 19
/* TPH_OPT_BLK_STAGE: 1 = the one-try rounds of the matrix-core blocked kernel (one mode) stage every panel's matrix blocks in
 * LDS once per workgroup -- loaded while the previous panel's products run, one barrier per panel, two buffers -- instead of
 * streaming them to each of its four waves (default, 1: 160 -> 142 us per round at 100-D x 131072, 48.8 -> 47.3 at 50-D x 65536,
 * profiles/r05_blkm_stage.json); 0 = streamed.  Same proposals bit for bit (same instructions on the same operands). */
#define TPH_OPT_BLK_STAGE 20
/* TPH_OPT_GMM_KERNEL: the clustering E-step (tph_gmm_estep, tph_gmm_em_run) at 16 <= n_dim <= 48: 0 = automatic (= 1), 1 = one
 * lane per row (row in LDS, precision matrix through scalar loads), 2 = 16 rows per wave on the FP64 matrix cores (measured slower on config 3: 158 against
 * 107 us per pass; kept for the parity test).  Same values to rounding (different summation order). */
#define TPH_OPT_GMM_KERNEL 21
/* TPH_OPT_FORMS_MFMA: 1 = the forms |L^-1 (u' - mu)|^2 behind the screened batches (tpCN, one mode, 16 < n_dim <= 112) come from the
 * matrix-core blocks of the blocked rounds (the same instructions as the rounds' own forms) instead of the lane-per-particle
 * pass through the scalar cache; 0 (default) = the latter.  Same values to rounding; measured 49.6 against 54.0 us per launch at
 * 131 072 x 100-D. */
#define TPH_OPT_FORMS_MFMA 22
int tph_set_option(tph_ctx* ctx, int option, int value);

/* ---- multi-GPU: one process per GPU --------------------------------------
 * Every iteration's particles are split evenly over the ranks: slot i of the global active set belongs to rank
 * i / (N / world), and each rank appends its slice to its LOCAL history shard (the history never moves).  The GLOBAL
 * history order is the reference's: iteration-major, and inside an iteration by slot -- i.e. block (t, rank 0), (t, rank 1)
 * ... -- so that a sharded run draws, trims and fits exactly what the one-GPU run does (same counter-based draws mapped
 * through the same cumulative weights; only floating-point summation order differs).
 * The library does not link a communication library: the host attaches two collectives (RCCL through its framework's
 * process group, or anything else) that operate IN PLACE on a device staging block it owns, addressed by byte offset:
 *   allreduce(user, offset, count, dtype, op)                  dtype 0 = f64, 1 = i64, 2 = i32; op 0 = sum, 1 = max, 2 = min
 *   allgather(user, send_offset, recv_offset, count, dtype)    recv block = world x count elements, rank order
 * Both must be ordered with the ctx stream (enqueue on it, or synchronise) and return 0 on success.  With a communicator
 * attached the *_global entry points below and tph_reweight_eval / tph_reweight_partials return GLOBAL results on every
 * rank; without one they are the single-GPU functions. */
typedef int (*tph_allreduce_fn)(void* user, int64_t offset, int64_t count, int dtype, int op);
typedef int (*tph_allgather_fn)(void* user, int64_t send_offset, int64_t recv_offset, int64_t count, int dtype);
int tph_comm_attach(tph_ctx* ctx, int rank, int world, void* buf_dev, int64_t buf_bytes, tph_allreduce_fn allreduce,
                    tph_allgather_fn allgather, void* user);
int tph_comm_detach(tph_ctx* ctx);
/* Optional, ranks on ONE node: small-message collectives (<= 32 KB: the reweight triples, the per-step acceptance sums, block
 * totals, moments) over peer-mapped device memory instead of the callbacks -- one single-block kernel on the ctx stream per
 * collective: every rank stores its values into its slot of every peer's inbox (xGMI is point to point), raises a sequence
 * flag, waits for the world's flags in its own inbox and reduces the slots in rank order (bit-identical sums on all ranks).
 * No host call, no second stream; the kernel takes its sequence number from device memory, so a step that contains one can
 * be captured in a graph.  Protocol: every rank calls _export (allocates its inbox, returns a 64-byte HIP IPC handle), the
 * host all-gathers the handles by any means, every rank calls _attach with all of them (rank order).  _attach maps the peers,
 * runs a self-test exchange and agrees with the other ranks through the attached all-reduce: *ok_out = 1 on every rank, or 0
 * on every rank (peer access or IPC unavailable: everything keeps going through the callbacks).  A peer that never arrives
 * makes the waiting kernel give up after TEMPEST_AMD_P2P_TIMEOUT seconds (default 120, or a longer
 * TEMPEST_AMD_STEP_TIMEOUT); the next collective, or
 * tph_comm_p2p_status, then fails.  tph_comm_detach / tph_ctx_destroy unmap. */
#define TPH_P2P_HANDLE_BYTES 64
int tph_comm_p2p_export(tph_ctx* ctx, void* handle_out /* [TPH_P2P_HANDLE_BYTES] host */);
int tph_comm_p2p_attach(tph_ctx* ctx, const void* handles /* [world][TPH_P2P_HANDLE_BYTES] host */, int* ok_out);
int tph_comm_p2p_active(const tph_ctx* ctx);
int tph_comm_p2p_status(tph_ctx* ctx);
/* Traffic counters of this ctx since creation (or since the last call with reset != 0): out[0] = small collectives that went
 * through the peer-to-peer exchange kernels (incl. those folded into tph_adapt), out[1] = collectives through the attached
* callbacks, out[2] = bytes through the callbacks, out[3] = slots of this rank refilled by the one-sided resample shuffle (on
 * average (world-1)/world of them arrive from peers), out[4] = bytes of those records.  What `bench.py --gpus N` reports as its `comm` block. */
int tph_comm_stats(tph_ctx* ctx, int64_t* out /*[5] host*/, int reset);
/* in-place all-reduce of a small device array that lives outside the staging block (MCMC step: the (accepted, sum alpha_c)
 * sums between tph_accept and tph_adapt; replaces a framework call per step).  Peer-to-peer when attached and count fits,
 * else staged through the all-reduce callback.  No-op without a communicator. */
int tph_comm_allreduce_dev(tph_ctx* ctx, void* data_dev, int64_t count, int dtype, int op);

/* ---- persistent ensemble: StateManager history (state_manager.py:171-176,356-416) ------------
 * tph_history_append = commit_current_to_history for the array keys u, x, logl, plus the cached
 * per-particle log-mixture  C_s = log sum_t n_t exp(beta_t l_s - logZ_t)  (state_manager.py:466-471,
 * kept up to the common -log N_h) updated incrementally: one logaddexp term for the old particles,
 * the full T-term fold for the n new ones.  n_global is n_t of the mixture (the iteration's particle
 * count over all shards; = n on one GPU). */
int tph_history_append(tph_ctx* ctx, const double* u_dev, const double* x_dev, const double* logl_dev,
                       int64_t n, int64_t ld, double beta, double logz, int64_t n_global);
int64_t tph_history_size(const tph_ctx* ctx);        /* N_h held by this ctx */
int tph_history_iterations(const tph_ctx* ctx);      /* T */
int tph_history_clear(tph_ctx* ctx);
/* where the history's memory is (TPH_OPT_HISTORY_VM): out9 = { rows held, rows of u / x backed by memory, rows reserved (the
 * leading dimension), 1 if u / x live in a mapped range, rows of the row-major mirror backed (0: no mirror), growth steps of the
 * mapped ranges, re-reservations (mappings moved to a wider range), times the mirror was given back under memory pressure,
 * copies of the whole history made while growing } */
int tph_history_memory(tph_ctx* ctx, int64_t* out9_host);
/* copy rows [off, off+n) to host: u/x as [n_dim][n] (dimension-major), logl/logmix as [n] */
int tph_history_read(tph_ctx* ctx, int key, int64_t off, int64_t n, double* out_host);
/* device base pointer and leading dimension (capacity) of a history array */
int tph_history_ptr(tph_ctx* ctx, int key, void** dev_ptr, int64_t* ld);
/* replace the history wholesale (resume / tests): arrays [n_dim][n] / [n] on host, tables of length T */
int tph_history_load(tph_ctx* ctx, const double* u_host, const double* x_host, const double* logl_host,
                     int64_t n, int T, const double* beta_t, const double* logz_t, const int64_t* n_t_local,
                     const int64_t* n_t_global);

/* ---- reweighting (state_manager.py:418-480, steps/reweight.py:88-118, tools.py:120-135) -------
 * For each trial beta_b:  v_s = beta_b*l_s - C_s ;  out[b] = (max_s v, sum_s e^{v-max}, sum_s e^{2(v-max)}).
 * Then ESS = s1^2/s2 (tools.py:134-135) and logZ = max + log s1 (state_manager.py:475; the two
 * log N_h cancel).  nb <= 16.  _partials leaves the triples on the device, _eval synchronises and returns them.
 * With a communicator attached both return the GLOBAL triples (all-gather of the ranks' triples + device-side merge). */
int tph_reweight_partials(tph_ctx* ctx, const double* betas_host, int nb, double* out_dev /*[nb][3]*/);
int tph_reweight_eval(tph_ctx* ctx, const double* betas_host, int nb, double* out_host /*[nb][3]*/);
/* normalised weights w_s = e^{beta l_s - C_s - vmax}/s1 for all N_h particles (reweight.py:106,328) */
int tph_weights(tph_ctx* ctx, double beta, double vmax, double s1, double* w_dev);
/* unnormalised log-weights beta*l - C + log(n_h_global)  (state_manager.py:473) */
int tph_logw(tph_ctx* ctx, double beta, int64_t n_h_global, double* logw_dev);

/* ---- the same steps over a SHARDED history (tph_comm_attach; without a communicator each is the plain function) ----------
 * They work on the whole local history (n = tph_history_size) and return on every rank what the one-GPU run computes on
 * the global history in the reference's order (iteration-major, then particle slot). */
/* global trimming threshold: exact global percentiles by a radix descent over all-reduced counts (8 rounds) */
int tph_trim_threshold_global(tph_ctx* ctx, const double* w_dev, int64_t n, double ess, int bins,
                              double* out_dev /*[4]*/, double* out_host /*[4] or NULL*/);
/* this rank's slice of the GLOBAL cumulative weight (one all-gather of the per-iteration block totals); records the block
 * table the next two calls use.  total_host (optional, synchronises) = the global sum. */
int tph_cdf_global(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev /*or NULL*/, double* cdf_dev,
                   double* total_host /*or NULL*/);
/* global resampling: of the n_slots GLOBAL output slots, idx = local row for those whose draw lands in one of this rank's
 * blocks, -1 otherwise (the blocks partition the cumulative axis: exactly one rank claims each slot).
 * scheme 0: multinomial, position U_i * total; 1: systematic, position (u0 + i) / n_slots * pscale (pscale = the renorm
 * divisor of tph_resample_systematic: the total when tools.py:214-217 renormalises, else 1). */
int tph_resample_select_global(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_slots, int scheme, uint64_t seed,
                               uint32_t tick, uint32_t tag, double u0, double pscale, int64_t* idx_dev);
/* The shuffle that follows the selection, ONE-SIDED (needs tph_comm_p2p_attach): every rank writes the rows it holds -- the
 * slots k with idx[k] >= 0 -- straight into the window of the slot's owner (rank k / n_local) over the peer mapping, one small
 * exchange is the barrier, and every owner unpacks its window into u / x / logl of its n_local slots (dimension-major, ld_out).
 * Replaces count exchange + pack + all-to-all-v + scatter; no host synchronisation.  Collective: all ranks call it with the
 * idx of the same selection.  A slot nobody wrote makes the next collective (or tph_comm_p2p_status) fail.
 * Returns 0, or 1 on EVERY rank when the row windows cannot be allocated / mapped (agreed between the ranks; nothing was
 * written: shuffle through an all-to-all instead), or < 0 on error. */
int tph_resample_put_global(tph_ctx* ctx, const int64_t* idx_dev /* [n_slots] */, int64_t n_slots, int64_t n_local,
                            double* u_out, double* x_out, double* logl_out, int64_t ld_out);
/* multiplicities of the LOCAL rows among factor * (*kept_count_dev) global multinomial draws (modes.py:196-201) */
int tph_multinomial_counts_global(tph_ctx* ctx, const double* cdf_dev, int64_t n, const double* kept_count_dev, int factor,
                                  int64_t n_draw_max, uint64_t seed, uint32_t tick, uint32_t tag, int32_t* counts_dev);
/* tph_fit_modes of the global up-sampled set: all-reduced first/second moments and median histograms, gathered candidates */
int tph_fit_modes_global(tph_ctx* ctx, const int32_t* counts_dev, const int32_t* labels_dev, int64_t n, int K,
                         double* means_dev, double* covs_dev, double* chol_dev, double* inv_dev, double* cholinv_dev /*or NULL*/);

/* ---- generic reductions on a device vector ---------------------------------------------------- */
/* out_host = (sum w, sum w^2, max w) */
int tph_sum_sq_max(tph_ctx* ctx, const double* w_dev, int64_t n, double* out_host /*[3]*/);

/* ---- trimming (tools.py:10-55) ----------------------------------------------------------------
 * Decides the percentile threshold exactly as the reference's 99->0 walk over a `bins`-point grid:
 * out = (threshold, kept_sum, kept_count, ess_total).  Weights must be normalised. */
int tph_trim_threshold(tph_ctx* ctx, const double* w_dev, int64_t n, double ess, int bins,
                       double* out_dev /*[4]*/, double* out_host /*[4] or NULL*/);

/* ---- resampling (tools.py:178-228, steps/resample.py:52-99) ----------------------------------- */
/* inclusive prefix sum of w (optionally masked: w_s < *thr_dev -> 0) */
int tph_cdf(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev /*or NULL*/, double* cdf_dev);
/* idx_i = #{k : cdf_k < (u0+i0+i)/size_global}   (strict walk of tools.py:222-226) */
int tph_resample_systematic(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_out, int64_t i0,
                            int64_t size_global, double u0, double renorm, int64_t* idx_dev);
/* idx_i = #{k : cdf_k/cdf_last <= U_i}, U_i = Philox(seed, tick, tag, item0+i)  (np.random.choice) */
int tph_resample_multinomial(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_out, uint64_t seed,
                             uint32_t tick, uint32_t tag, int64_t item0, int64_t* idx_dev);
/* Sharded variant: of the n_slots GLOBAL output slots, the ones whose position in the global cumulative weight
 * falls in this rank's span [w_before, w_upto) get idx = local row, all others -1 (scheme 0 multinomial, 1 systematic).
 * Every rank evaluates the same counter-based draws, so the spans partition the slots without communication.
 * is_last: bit 0 set on the rank owning the last span, bit 1 on the rank owning the first. */
int tph_resample_select(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_slots, int scheme, uint64_t seed,
                        uint32_t tick, uint32_t tag, double u0, double w_before, double w_upto, double w_total,
                        int is_last, int64_t* idx_dev);
/* u_out[j][i] = u_hist[j][idx_i] etc. (steps/resample.py:86-99).  Reads the row-major mirror of the history (one contiguous
 * record per gathered row; TPH_OPT_ROW_MIRROR) after bringing it up to date with the rows appended since its last use. */
int tph_gather(tph_ctx* ctx, const int64_t* idx_dev, int64_t n_out, double* u_out, double* x_out,
               double* logl_out, int64_t ld_out);
/* posterior extraction (core.py:187-242): rows idx_i (idx_dev NULL = the first m rows) of the history's u or x
 * (key) as ROW-MAJOR (m, d) x_out, with logl_out[i] = logl[idx_i] and, if w_out, w_out[i] = w_dev[idx_i] / wdiv:
 * only the kept rows cross PCIe, already in the layout Sampler.posterior() returns. */
int tph_posterior_rows(tph_ctx* ctx, int key, const int64_t* idx_dev, int64_t m, const double* w_dev, double wdiv,
                       double* x_out /*[m][d]*/, double* logl_out /*[m]*/, double* w_out /*[m] or NULL*/);
/* out[i] = a[b[i]]: indices of a resampling drawn over an already compacted selection */
int tph_index_compose(tph_ctx* ctx, const int64_t* a_dev, const int64_t* b_dev, int64_t m, int64_t* out_dev);
/* multiplicity of each history row among factor*(*kept_count_dev) multinomial draws from cdf (modes.py:196-201);
 * kept_count_dev NULL = n_draw_max draws.  The count stays on the device: no host sync. */
int tph_multinomial_counts(tph_ctx* ctx, const double* cdf_dev, int64_t n, const double* kept_count_dev, int factor,
                           int64_t n_draw_max, uint64_t seed, uint32_t tick, uint32_t tag, int32_t* counts_dev);

/* ---- mutation (steps/mutate.py:76-200, mcmc.py:142-411) ---------------------------------------- */
/* u ~ U(0,1)^d (mutate.py:102) */
int tph_prior_draw(tph_ctx* ctx, double* u_dev, int64_t n, int64_t ld, uint64_t seed, uint32_t tick, int64_t item0);
/* rows with +-inf logl replaced by uniformly chosen finite rows; stats_dev = (n_finite, n) (mutate.py:122-148) */
int tph_inf_repair(tph_ctx* ctx, double* u_dev, double* x_dev, double* logl_dev, int64_t n, int64_t ld,
                   uint64_t seed, uint32_t tick, int64_t item0, double* stats_dev /*[2]*/);
/* the same, and src_dev[i] = the row that now sits in row i (i itself for untouched rows): what the caller needs to move
 * per-particle data the library does not hold -- the likelihood's blobs, mutate.py:135-136 */
int tph_inf_repair_src(tph_ctx* ctx, double* u_dev, double* x_dev, double* logl_dev, int64_t n, int64_t ld,
                       uint64_t seed, uint32_t tick, int64_t item0, double* stats_dev /*[2]*/, int64_t* src_dev /*[n] or NULL*/);
/* Step control (optional, ctl_dev may be NULL): tph_adapt's state_dev block, extended to TPH_STEP_STATE_LEN doubles,
 * passed to tph_propose / tph_accept makes one MCMC step replayable as a captured hipGraph with no per-step kernel
 * arguments: the RNG tick used is  tick + state[7] + 2 * state[0]  (state[0] = steps completed, advanced by tph_adapt;
 * state[7] = the run's tick base, < 2^32), beta is read from state[6], and once state[1] (the stopping rule of mcmc.py:119-140) is set, tph_accept and tph_adapt of any
 * further (speculatively launched) step leave every buffer untouched.  state[8] receives, from the d > 16 proposal kernel, the mean
 * number of redraw attempts per particle of its first block (tph_adapt forwards it as field [6] of the mailbox record): the host
 * uses it to switch TPH_OPT_ML_UNSTAGED. */
#define TPH_STEP_STATE_LEN 10
/* proposals for all particles (mcmc.py:225-249 tpCN, :301-312 RWM) incl. boundary handling and the
 * redraw-until-in-bounds loop; maha_u/maha_up receive (u-mu)^T S^-1 (u-mu) at u and u' (tpCN), evaluated as
 * |L^-1 (u-mu)|^2: cholinv_dev = the K inverse Cholesky factors (tph_chol_inv / tph_fit_modes produce them), or NULL --
 * the library then inverts chol_dev itself on every call (a caller holding only ModeStatistics.chol_covariances). */
int tph_propose(tph_ctx* ctx, int kernel, double* u_dev /* in; with pending_dev also out */, const int32_t* assign_dev, int64_t n, int64_t ld,
                int K, const double* means_dev, const double* chol_dev, const double* cholinv_dev,
                const double* dof_dev, const double* sigmas_dev, const uint8_t* bc_dev,
                uint64_t seed, uint32_t tick, int64_t item0,
                double* uprime_dev, double* maha_u_dev, double* maha_up_dev, const double* ctl_dev,
                uint8_t* pending_dev /* NULL, or the mask written by the previous step's deferred tph_accept: particles with
                                        pending[i] != 0 first take uprime[i] (that step's accepted proposal) as their current
                                        point -- u is updated in place, the flag cleared -- and then propose from it */);
/* Metropolis step (mcmc.py:163-177 with the factor of :251-279): masked overwrite of u,x,logl and
 * per-rank sums  sums_dev = (n_accepted, sum alpha_0 .. sum alpha_{K-1}).  x_dev and xprime_dev may both be NULL: x is
 * then not maintained during the run (it is a function of u: the caller re-evaluates prior_transform once at the end, which
 * halves this kernel's masked-write traffic). */
int tph_accept(tph_ctx* ctx, int kernel, double beta, double* u_dev, double* x_dev, double* logl_dev,
               const double* uprime_dev, const double* xprime_dev, const double* loglprime_dev,
               double* maha_u_dev /* in; accepted rows take maha_up (tpCN): see ctl_dev */, const double* maha_up_dev,
               const int32_t* assign_dev,
               int64_t n, int64_t ld, int K, const double* dof_dev,
               uint64_t seed, uint32_t tick, int64_t item0,
               double* sums_dev /*[1+K]; NULL = leave the block partials in partials_dev for tph_adapt to sum*/,
               const double* ctl_dev,
               double* partials_dev /* NULL = library scratch, or ceil(n/256)*(1+K) doubles owned by the caller: needed
                                       when the launch is captured in a graph (the scratch may move when it grows) */,
               uint8_t* pending_dev /* NULL = update u (and x) in place; else DEFERRED mode (x_dev must be NULL): the decision
                                       goes to pending[i] (1 = accepted), logl and maha_u take their new values, u is left
                                       alone and the next tph_propose given this mask moves the accepted proposals into place.
                                       The masked in-place copy is a read-modify-write of every line of u (the accepted rows
                                       are scattered): deferred, that traffic rides under the proposal kernel's FP64 work */);
/* sigma adaptation + adaptive stopping rule (mcmc.py:104-140,180-194,281-288,320-323) from GLOBAL sums.
 * state_dev (6 doubles; TPH_STEP_STATE_LEN when it doubles as step control or a mailbox is given):
 *            [0]=iteration (in/out) [1]=done flag [2]=accepted fraction [3]=mean alpha [4]=mean(sigma)/sigma_0
 *            [5]=adaptive step target ([6]=beta, [7]=tick base, [8]=redraw-regime probe of the proposal kernel);
 * counts_dev = particles per cluster (global).  A call with the done flag already set is a no-op. */
int tph_adapt(tph_ctx* ctx, int kernel, double* sums_dev /* in; out when partials_dev is given */, const double* counts_dev, int K,
              double n_global, int n_dim, int n_steps, int n_max, double* sigmas_dev, double* state_dev,
              double* mailbox_host /* NULL, or mailbox_slots x 8 doubles of PINNED host memory (device-accessible):
                                      the record of step s = state[0..5] (+ state[8] as field [6]) goes to slot
                                      s % mailbox_slots, its field [7] = s is stored last (system-scope release), so the host can poll for it */,
              int mailbox_slots,
              const double* partials_dev /* NULL, or tph_accept's block partials of n particles: their column sums
                                            are formed here (into sums_dev) instead of by a kernel of their own, in the
                                            canonical order (per virtual shard, then in shard order: tph_accept_sums_global).
                                            With a communicator attached they are THIS RANK's shard sums and the kernel exchanges
                                            them with the peers in place (needs tph_comm_p2p_attach): a sharded step then has the
                                            launches of a single-GPU step and no host call.  Returns 1 (nothing launched) when
                                            that exchange is not attached or the rank's shard sums exceed its 32 KB slot:
                                            combine the ranks with tph_accept_sums_global and call again with partials_dev = NULL */,
              int64_t n);
/* The step's sums (#accepted, sum alpha_c) over ALL ranks from this rank's block partials (tph_accept with sums_dev = NULL),
 * in the CANONICAL order: the particle slots are cut into V virtual shards (V from the global particle count alone: the largest
 * of 48, 16, 12, 8, 6, 4, 3, 2, 1 with n_global % (256 V) == 0), a shard's sums are formed by a fixed procedure that sees only
 * the shard's blocks, and the V shard sums are added in shard order -- the same summation tree on any number of ranks that
 * divides V, so the adapted sigma does not depend on it.  host_paced != 0: the exchange goes through the all-gather callback even
 * where the peer-to-peer exchange is attached (user callbacks on the host may keep the ranks minutes apart). */
int tph_accept_sums_global(tph_ctx* ctx, const double* partials_dev, int64_t n, int K, double* sums_dev, int host_paced);
int tph_cluster_counts(tph_ctx* ctx, const int32_t* assign_dev, int64_t n, int K, double* counts_dev);

/* ---- proposal fit (student.py:6-116 effective form, modes.py:58-119,131-288) -------------------- */
/* From multiplicities counts_s over the first n history rows (labels_dev NULL = one global mode):
 * per mode k: mean = per-dimension median, cov = MLE covariance + diag(var)/n of the up-sampled set,
 * then chol/inv (and L^-1) with the 1e-6 ridge on failure.  Outputs [K][d], [K][d][d] x4 on the device. */
int tph_fit_modes(tph_ctx* ctx, const int32_t* counts_dev, const int32_t* labels_dev, int64_t n, int K,
                  double* means_dev, double* covs_dev, double* chol_dev, double* inv_dev, double* cholinv_dev /*or NULL*/);
/* Cholesky L, inverse Sigma^-1 = L^-T L^-1 and (cholinv_dev, optional) L^-1 of K d x d matrices with the reference's
 * ridge rule (modes.py:105-119) */
int tph_chol_inv(tph_ctx* ctx, double* covs_dev, int K, double* chol_dev, double* inv_dev, double* cholinv_dev /*or NULL*/);

/* ---- volume variation (tools.py:58-117) ---------------------------------------------------------
 * stage 1: weighted mean and covariance of the history's u with weights w (device -> host, tiny);
 * stage 2 (after the host's rank check / inverse, d x d): 0.25 * sum w^2 clip(d2-n,+-1e6)^2 partial. */
int tph_weighted_moments(tph_ctx* ctx, const double* w_dev, int64_t n, double* mean_cov_dev /*[d + d*d]*/);
/* sums_dev = (sum w, sum w u_0 .. sum w u_{d-1}) over the first n history rows (per-rank partial sums) */
int tph_weighted_sums(tph_ctx* ctx, const double* w_dev, int64_t n, double* sums_dev /*[1+d]*/);
/* cov_dev[a][b] = sum_s w_s (u_a - mean_a)(u_b - mean_b)  (raw per-rank sums, not divided by sum w) */
int tph_weighted_cov_centered(tph_ctx* ctx, const double* w_dev, int64_t n, const double* mean_dev,
                              double* cov_dev /*[d*d]*/);
/* one-pass variant for n_dim <= 12: moments about a centre close to the mean (cancellation-free finish);
 * out_dev = (sum w, mean[d], cov[d][d]) with cov already divided by sum w */
int tph_weighted_moments_shifted(tph_ctx* ctx, const double* w_dev, int64_t n, const double* centre_dev,
                                 double* out_dev /*[1 + d + d*d]*/);
int tph_cv_sum(tph_ctx* ctx, const double* w_dev, int64_t n, const double* mean_dev, const double* covinv_dev,
               double* out_dev /*[1]: sum w^2 dev^2*/);
/* The whole statistic in one call (tools.py:58-117), d x d work on the device: weighted mean / covariance of the history's
 * u under the (normalised) weights, the reference's rank rule (rank-deficient: cov += 1e-6 trace I; test = pivoted Cholesky
 * with the threshold n_dim * eps * trace), Cholesky + triangular inverse, sum_s w_s^2 clip(|L^-1 (u_s - mean)|^2 - n_dim)^2
 * and 0.5 sqrt(sum / S0^2), or 1e10 where the reference returns it (singular, non-finite).  centre_dev (n_dim doubles, in/out,
 * optional): a point near the mean (it receives the new mean) enabling the one-pass moments for n_dim <= 12.  One host wait
 * (pinned mailbox).  With a communicator attached: the statistic of the global weighted history. */
int tph_volume_variation(tph_ctx* ctx, const double* w_dev, int64_t n, double* centre_dev /*or NULL*/, double* value_host);

/* ---- clustering working set (cluster.py) ---------------------------------------------------------------
 * The hierarchical GMM runs on a compact, [0,1]-normalised SoA copy x[j*ld+i] of the kept (trimmed) history rows. */
/* idx = ascending history rows with w >= *thr_dev (their number is the kept_count of tph_trim_threshold) */
int tph_compact_indices(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev, int64_t* idx_dev);
/* out[j][i] = (u_hist[j][idx_i] - shift_j) * scale_j (shift NULL = copy), wout[i] = w[idx_i] (cluster.py:373-379) */
int tph_gather_u_affine(tph_ctx* ctx, const int64_t* idx_dev, int64_t m, const double* shift_dev, const double* scale_dev,
                        const double* w_dev, double* out_dev, int64_t ld, double* wout_dev);
int tph_affine(tph_ctx* ctx, double* x_dev, int64_t ld, int64_t n, const double* shift_dev, const double* scale_dev);
/* sums (1+d) [+ range (2d): min,max of rows with w>0] and raw centred second moments of an explicit SoA array */
int tph_x_weighted_sums(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* w_dev, double* sums_dev,
                        double* range_dev /*or NULL*/);
int tph_x_weighted_cov(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* w_dev, const double* mean_dev,
                       double* cov_dev);
/* Gaussian-mixture pass over the rows with labels[i]==label (labels NULL = all).  params per component:
 * [log-weight term, mean(d), precision(d*d), logdet].  mode 0: wr[k][i] = sw_i w_k N_k / (sum_k w_k N_k + eps) and
 * stats = (sum sw log(p+1e-10), sum log(p+1e-10), #rows) (cluster.py:178-198,287-304,330-340); mode 1: wr[i] = sw_i min_k
 * maha_k (k-means++ seeding, :146-157); mode 2: label_out[i] = argmax_k term_k + logN_k (:306-328,600-696).
 * shift/scale (NULL = none): rows are normalised on the fly as (x - shift_j) * scale_j. */
int tph_gmm_estep(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* sw_dev, const int32_t* labels_dev,
                  int label, int K, const double* params_dev, int mode, double eps, const double* shift_dev,
                  const double* scale_dev, double* wr_dev, int32_t* label_out_dev, double* stats_dev);

/* Device-paced EM for covariance_type "full" (cluster.py:104-133,174-304): the loop's state lives in ONE device block of
 * tph_gmm_em_state_doubles(n_dim, K) doubles -- control words [0..15] = {iteration, done flag, lower bound, n_iter, the last
 * E-step's three sums, ...}, the M-step's moments, the packed E-step parameters, and the weights / means / covariances those
 * parameters were formed from (offsets: 16 + K(1+d) + K d + K d^2 + K(2+d+d^2), then K, K d, K d^2).  tph_gmm_em_begin: the
 * fit's first M-step from the initial responsibilities in wr_dev (K x n) and a fresh control block.  tph_gmm_em_run ENQUEUES
 * `iters` iterations (parameters incl. precisions and log-determinants by a kernel; E-step; convergence test on the device;
 * M-step); passes behind a raised done flag are no-ops.  The host reads the control words once per call. */
int64_t tph_gmm_em_state_doubles(int n_dim, int K);
int tph_gmm_em_begin(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int K, const double* wr_dev, double* state_dev);
int tph_gmm_em_run(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* sw_dev, const int32_t* labels_dev,
                   int label, int K, double* wr_dev, double* state_dev, double reg_covar, double tol, int max_iter, int iters);

/* ---- Student-t EM (opt-in extension: tempest/student.py:40-57, 96-102; the reference's own loop returns its start values,
 * SURVEY F5 -- that estimator is tph_fit_modes) ------------------------------------------------------------------------
 * Rows carry int32 multiplicities (counts) and an optional label filter; delta_i = (x_i - mu)^T Sigma^-1 (x_i - mu) per row.
 * tph_student_sums: for nb <= 16 trial nu: out[b] = (sum_i c_i log w_i, sum_i c_i w_i), w_i = (nu_b + d) / (nu_b + delta_i)
 * (the data-dependent terms of the digamma equation).  tph_student_weights: v_i = c_i (nu + d) / (nu + delta_i). */
int tph_student_sums(tph_ctx* ctx, const double* delta_dev, const int32_t* counts_dev, const int32_t* labels_dev, int label, int64_t n,
                     const double* nus_host, int nb, double* out_host /*[nb][2]*/);
int tph_student_weights(tph_ctx* ctx, const double* delta_dev, const int32_t* counts_dev, const int32_t* labels_dev, int label, int64_t n,
                        double nu, double* v_dev);

/* ---- measurement aids (tph_bench_*): used by bench.py, tools/ and the tests; NOT part of the drop-in surface ------------
 * tph_bench_reweight_time: average duration (ms, HIP events on the ctx stream) of `reps` back-to-back launches of the
 * reweight reduction kernel alone for nb trial betas.  tph_bench_membw_time: the box's own ceiling, timed in the same
 * process -- a streaming READ (mode 0: n_doubles * 8 bytes per launch, the reduction's access pattern) or COPY (mode 1:
 * 2 * n_doubles * 8 bytes) over freshly allocated buffers.  tph_bench_fp64_time: sustained FP64 vector-FMA rate (TFLOP/s).
 * tph_bench_mf_normals: max |z~ - z| of the screened kernel's FP32 Box-Muller pair against the FP64 pair of the same
 * Philox blocks over n_blocks blocks starting at block `first` (edge = 1: hand-made blocks at the ends of u1 and around the
 * switch of the logarithm); out = (max error, max |z|, blocks).  tph_bench_mf_counters: the counters of the last screened
 * launch: next chunk, sum of attempts, particles, audit contradictions, FP64 verifications, attempts screened, pair jobs. */
int tph_bench_reweight_time(tph_ctx* ctx, double beta, int nb, int reps, double* avg_ms_host);
int tph_bench_membw_time(tph_ctx* ctx, int mode, int64_t n_doubles, int reps, double* avg_ms_host);
int tph_bench_fp64_time(tph_ctx* ctx, int reps, double* tflops_host);
int tph_bench_mf_normals(tph_ctx* ctx, uint64_t seed, uint64_t first, uint64_t n_blocks, int edge, double* out3_host);
int tph_bench_mf_counters(tph_ctx* ctx, unsigned long long* out7_host);

#ifdef __cplusplus
}
#endif
#endif
