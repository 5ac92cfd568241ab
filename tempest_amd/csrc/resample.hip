// Resampling of the active set from the weighted history.
// Reference: tempest/tools.py:178-228 (systematic_resample), tempest/steps/resample.py:52-99
// (Resampler.run: np.random.choice(p=w) | systematic, then u,x,logl[idx]), tempest/modes.py:196-201
// (x4 multinomial up-sampling before the proposal fit).
#include "common.h"
#include "scan.h"
#include "cdf_index.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

// the plain three-pass scan (library-internal prefix counts: flags of kept rows, of finite rows; their scratch layout counts on the
// num_tiles(n) doubles this takes at the front of ctx->scratch)
int tph_cdf_plain(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev, double* cdf_dev) {
  if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)tph_scan::num_tiles(n))) return -1;
  double* tiles = (double*)ctx->scratch;
  return thr_dev ? tph_scan::inclusive<tph_scan::MASKED>(ctx, w_dev, n, thr_dev, tiles, cdf_dev)
                 : tph_scan::inclusive<tph_scan::PLAIN>(ctx, w_dev, n, nullptr, tiles, cdf_dev);
}
static int cdf_pieces(tph_ctx* ctx, const double* w_dev, const double* thr_dev, double* cdf_dev, const tph_part& part);
extern "C" int tph_cdf(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev, double* cdf_dev) {
  TPH_REQUIRE(ctx && w_dev && cdf_dev && n > 0, "tph_cdf: bad argument");
  // (an input that lives in ctx->scratch is a library-internal array, never the weights of the history)
  const bool internal = ctx->scratch && (const char*)w_dev >= (const char*)ctx->scratch && (const char*)w_dev < (const char*)ctx->scratch + ctx->scratch_bytes;
  if (!ctx->comm_active() && !internal) {
    // the weights of the whole history on one GPU: cumulative sums piece by piece of the canonical partition (common.h), as the
    // ranks of a sharded run form them -- the same numbers for any number of GPUs (a weight vector that is not the history's,
    // or a history without the partition: the plain three-pass scan below)
    const tph_part part = tph_partition(ctx, n);
    if (part.canonical) return cdf_pieces(ctx, w_dev, thr_dev, cdf_dev, part);
  }
  if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)tph_scan::num_tiles(n))) return -1;
  double* tiles = (double*)ctx->scratch;
  return thr_dev ? tph_scan::inclusive<tph_scan::MASKED>(ctx, w_dev, n, thr_dev, tiles, cdf_dev)
                 : tph_scan::inclusive<tph_scan::PLAIN>(ctx, w_dev, n, nullptr, tiles, cdf_dev);
}

// #{k in [0,n) : pred(cdf_k)}, pred monotone (true ... true false ... false)
template <bool STRICT>
__device__ __forceinline__ int64_t count_below(const double* __restrict__ cdf, int64_t n, double div, double pos) {
  int64_t lo = 0, hi = n;  // answer in [lo, hi]
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    double c = cdf[mid] / div;
    bool below = STRICT ? (c < pos) : (c <= pos);
    if (below) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) k_resample_systematic(const double* __restrict__ cdf, int64_t n, int64_t n_out,
                                                             int64_t i0, int64_t size_global, double u0, double renorm,
                                                             int64_t* __restrict__ idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  // (the position scaled to the cumulative weights, not the weights divided down to the position: the form the sharded
  // selection uses, k_select_global -- one predicate on every number of GPUs)
  const double p = (u0 + (double)(i0 + i)) / (double)size_global * renorm;
  int64_t k = count_below<true>(cdf, n, 1.0, p);
  idx[i] = k < n ? k : n - 1;
}

extern "C" int tph_resample_systematic(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_out, int64_t i0,
                                       int64_t size_global, double u0, double renorm, int64_t* idx_dev) {
  TPH_REQUIRE(ctx && cdf_dev && idx_dev && n > 0 && n_out > 0, "tph_resample_systematic: bad argument");
  TPH_REQUIRE(renorm > 0 && size_global >= n_out, "tph_resample_systematic: bad renorm/size");
  hipLaunchKernelGGL(k_resample_systematic, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx->stream, cdf_dev, n,
                     n_out, i0, size_global, u0, renorm, idx_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) k_resample_multinomial(tph_cdf_index ix, int64_t n_out, uint64_t seed, uint32_t tick,
                                                              uint32_t tag, int64_t item0, int64_t* __restrict__ idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  tph_rng g(seed, tick, tag, (uint64_t)(item0 + i));
  double U, U1;
  g.uniform2(0, U, U1);
  const int64_t n = ix.n[0];
  const double p = U * ix.lvl[0][n - 1];       // as k_select_global: the draw scaled to the total, compared with the sums themselves
  int64_t k = tph_count_below<false>(ix, 1.0, p);
  idx[i] = k < n ? k : n - 1;
}

extern "C" int tph_resample_multinomial(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_out, uint64_t seed,
                                        uint32_t tick, uint32_t tag, int64_t item0, int64_t* idx_dev) {
  TPH_REQUIRE(ctx && cdf_dev && idx_dev && n > 0 && n_out > 0, "tph_resample_multinomial: bad argument");
  if (tph_scratch_reserve(ctx, sizeof(double) * tph_cdf_index_doubles(n))) return -1;
  tph_cdf_index ix;
  if (tph_cdf_index_build(ctx, cdf_dev, n, (double*)ctx->scratch, &ix)) return -1;
  hipLaunchKernelGGL(k_resample_multinomial, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx->stream, ix, n_out, seed,
                     tick, tag, item0, idx_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// Sharded selection (multi-GPU): every rank walks ALL n_slots global output slots (the draws are a pure
// function of (seed, tick, slot), so every rank sees the same ones) and keeps those whose position falls in
// its own span [w_before, w_upto) of the global cumulative weight; idx_out[i] = local row or -1.
// scheme 0 = multinomial (position U_i * w_total, `<=` walk), 1 = systematic ((u0+i)/n_slots * w_total, `<` walk).
// is_last: bit 0 = this rank owns the last span, bit 1 = the first.
__global__ void __launch_bounds__(256) k_resample_select(tph_cdf_index ix, int64_t n_slots, int scheme, uint64_t seed, uint32_t tick, uint32_t tag, double u0,
                                                         double w_before, double w_upto, double w_total, int is_last,
                                                         int64_t* __restrict__ idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  double p;
  if (scheme == 0) {
    tph_rng g(seed, tick, tag, (uint64_t)i);
    double U, U1;
    g.uniform2(0, U, U1);
    p = U * w_total;
  } else {
    p = (u0 + (double)i) / (double)n_slots * w_total;
  }
  // span membership uses the same boundaries on every rank (computed from the all-gathered totals), so the
  // spans partition the slots: multinomial row = #{cum <= p}, systematic row = #{cum < p}
  const bool first = (is_last & 2) != 0, last = (is_last & 1) != 0;
  bool mine = scheme == 0 ? ((first || p >= w_before) && (last || p < w_upto))
                          : ((first || p > w_before) && (last || p <= w_upto));
  if (!mine) { idx[i] = -1; return; }
  double q = p - w_before;
  const int64_t n = ix.n[0];
  int64_t k = scheme == 0 ? tph_count_below<false>(ix, 1.0, q) : tph_count_below<true>(ix, 1.0, q);
  idx[i] = k < n ? k : n - 1;
}

extern "C" int tph_resample_select(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_slots, int scheme,
                                   uint64_t seed, uint32_t tick, uint32_t tag, double u0, double w_before, double w_upto,
                                   double w_total, int is_last, int64_t* idx_dev) {
  TPH_REQUIRE(ctx && cdf_dev && idx_dev && n > 0 && n_slots > 0, "tph_resample_select: bad argument");
  TPH_REQUIRE(scheme == 0 || scheme == 1, "tph_resample_select: scheme must be 0 (multinomial) or 1 (systematic)");
  if (tph_scratch_reserve(ctx, sizeof(double) * tph_cdf_index_doubles(n))) return -1;
  tph_cdf_index ix;
  if (tph_cdf_index_build(ctx, cdf_dev, n, (double*)ctx->scratch, &ix)) return -1;
  hipLaunchKernelGGL(k_resample_select, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, ctx->stream, ix,
                     n_slots, scheme, seed, tick, tag, u0, w_before, w_upto, w_total, is_last, idx_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// GLOBAL cumulative weights of a sharded history (one process per GPU, tph_comm_attach).
// The reference resamples from ONE history ordered by iteration and, inside an iteration, by particle slot
// (state_manager.py:267-320 flat=True; steps/resample.py:80-84; modes.py:196-201).  With the slots of every iteration split
// evenly over the ranks that order is: block (t = 0, rank 0), (0, 1) ... (0, G-1), (1, 0) ...; a rank holds the T blocks
// (t, rank) of `rows` rows each.  Its slice of the global cumulative sum is therefore
//     gcdf[i] = glo[t(i)] + (inclusive scan of w inside block t(i)) ,   glo[t] = mass of all blocks before (t, rank),
// obtained with ONE all-gather of the T block totals.  Rows are clamped into [glo[t], ghi[t]] and the last row of a block is
// set to ghi[t] = the next block's glo exactly (every rank computes the table from the same gathered numbers with the same
// arithmetic), so the blocks partition the global axis without gaps or overlaps whatever the rounding inside a block: each
// draw position belongs to exactly one rank.  A draw is then mapped exactly as on one GPU: same counter-based uniform, same
// predicate, on the same (to rounding) cumulative values -- a sharded run selects the same history rows as the one-GPU run.
template <int MODE>
__global__ void __launch_bounds__(tph_scan::THREADS) k_seg_tile_sums(const double* __restrict__ w, int64_t rows, int tpb,
                                                                     const double* __restrict__ thr_dev, double* __restrict__ tiles) {
  using namespace tph_scan;
  const double thr = MODE == MASKED ? thr_dev[0] : 0.0;
  const int64_t t = blockIdx.x / tpb, j = blockIdx.x % tpb;
  double x[ITEMS];
  load_tile<MODE>(w, t * rows + j * TILE, (t + 1) * rows, thr, x);
  double s = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
  __shared__ double sh[THREADS / 64];
  s = tph_block_sum(s, sh);
  if (threadIdx.x == 0) tiles[blockIdx.x] = s;
}
// totals[t] = sum of the block's tile sums (one workgroup per block, fixed order)
__global__ void __launch_bounds__(256) k_seg_totals(const double* __restrict__ tiles, int tpb, double* __restrict__ totals) {
  double s = 0.0;
  for (int j = threadIdx.x; j < tpb; j += 256) s += tiles[(size_t)blockIdx.x * tpb + j];
  __shared__ double sh[4];
  s = tph_block_sum(s, sh);
  if (threadIdx.x == 0) totals[blockIdx.x] = s;
}
// table = (glo[T], ghi[T], total) of rank `rank` from the gathered totals all[G][T], accumulated in global (iteration-major)
// order by ONE thread: identical on every rank
// (pieces: a rank holds vl virtual shards of every iteration -- common.h: tph_part --, its piece (t, v') is entry t * vl + v' of
// its list; the global order is iteration, rank, shard.  The totals are added in that order whatever G is: the table, and with it
// every cumulative weight, is the same on any number of GPUs that divides V.)
// (the running sum is ONE thread's serial chain: that order is the definition.  Everything around the chain is done by the
// whole workgroup: the totals are staged in LDS in the global order, thread 0 turns them into running sums in place -- eight
// LDS reads requested ahead of the eight dependent additions --, and the entries of this rank are written out by all lanes.
// One thread doing loads, index arithmetic and stores around each addition took 64 us at 25 iterations x 16 shards.)
__global__ void __launch_bounds__(256) k_block_table(const double* __restrict__ all, int G, int T, int vl, int rank, double* __restrict__ table) {
  extern __shared__ double s_run[];
  const int P = T * vl, tot = G * P;
  if (tot <= 8000) {
    // position i of the global order <-> (t, g, v): i = (t * G + g) * vl + v
    for (int i = threadIdx.x; i < tot; i += blockDim.x) {
      const int v = i % vl, tg = i / vl, g = tg % G, t = tg / G;
      s_run[i] = all[(size_t)g * P + t * vl + v];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double run = 0.0;
      int i = 0;
      for (; i + 8 <= tot; i += 8) {
        double a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = s_run[i + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { run += a[k]; s_run[i + k] = run; }
      }
      for (; i < tot; ++i) { run += s_run[i]; s_run[i] = run; }
      table[2 * P] = run;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      const int t = p / vl, v = p - t * vl, i = (t * G + rank) * vl + v;
      table[p] = i > 0 ? s_run[i - 1] : 0.0;
      table[P + p] = s_run[i];
    }
    return;
  }
  if (threadIdx.x) return;
  double run = 0.0;
  for (int t = 0; t < T; ++t)
    for (int g = 0; g < G; ++g)
      for (int v = 0; v < vl; ++v) {
        const int p = t * vl + v;
        if (g == rank) table[p] = run;
        run += all[(size_t)g * P + p];
        if (g == rank) table[P + p] = run;
      }
  table[2 * P] = run;
}
// tile sums of a block -> exclusive offsets + glo[t] (one workgroup per block; sequential over chunks of 256 tiles)
__global__ void __launch_bounds__(256) k_seg_offsets(double* __restrict__ tiles, int tpb, const double* __restrict__ table) {
  __shared__ double wsum[4];
  __shared__ double carry;
  double* mine = tiles + (size_t)blockIdx.x * tpb;
  if (threadIdx.x == 0) carry = table[blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int j0 = 0; j0 < tpb; j0 += 256) {
    const int j = j0 + threadIdx.x;
    const double v = j < tpb ? mine[j] : 0.0;
    const double inc = tph_scan::wave_incl(v);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    double off = carry;
    for (int k = 0; k < wid; ++k) off += wsum[k];
    const double prev = __shfl_up(inc, 1, 64);
    if (lane > 0) off += prev;                      // exclusive prefix (never `inclusive - own`: see scan.h)
    if (j < tpb) mine[j] = off;
    __syncthreads();
    if (threadIdx.x == 255) carry = off + v;
    __syncthreads();
  }
}
template <int MODE>
__global__ void __launch_bounds__(tph_scan::THREADS) k_seg_apply(const double* __restrict__ w, int64_t rows, int tpb, int T,
                                                                 const double* __restrict__ thr_dev, const double* __restrict__ tiles,
                                                                 const double* __restrict__ table, double* __restrict__ out) {
  using namespace tph_scan;
  const double thr = MODE == MASKED ? thr_dev[0] : 0.0;
  const int64_t t = blockIdx.x / tpb, j = blockIdx.x % tpb;
  const int64_t lim = (t + 1) * rows, base = t * rows + j * TILE;
  double x[ITEMS];
  load_tile<MODE>(w, base, lim, thr, x);
  __shared__ double wsum[THREADS / 64];
  scan_tile(x, tiles[blockIdx.x], wsum);
  const double lo = table[t], hi = table[T + t];
  store_tile(out, base, lim, x, [=](double v, int64_t p) { return p == lim - 1 ? hi : fmin(fmax(v, lo), hi); });
}

// the segmented scan over the P = T * vl pieces of the canonical partition (one rank: all V shards; `comm`: the totals of all
// ranks gathered first).  Leaves the piece table in ctx->blk_table.
static int cdf_pieces(tph_ctx* ctx, const double* w_dev, const double* thr_dev, double* cdf_dev, const tph_part& part) {
  const int G = ctx->comm_active() ? ctx->world : 1;
  const int P = part.T * part.vl;
  const int64_t rows = part.nv;
  const int tpb = (int)tph_scan::num_tiles(rows);
  const size_t a_tiles = (sizeof(double) * (size_t)P * tpb + 255) / 256 * 256;
  if (tph_scratch_reserve(ctx, a_tiles + sizeof(double) * (size_t)P)) return -1;
  if (G > 1 || ctx->comm_active())
    if (tph_comm_require(ctx, sizeof(double) * (size_t)P * (G + 1), "tph_cdf_global")) return -2;
  if (ctx->blk_table_cap < 2 * P + 1) {
    int nc = ctx->blk_table_cap ? ctx->blk_table_cap : 513;
    while (nc < 2 * P + 1) nc = 2 * nc + 1;
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->blk_table) TPH_HIP(hipFree(ctx->blk_table));
    ctx->blk_table = nullptr; ctx->blk_table_cap = 0;
    TPH_HIP(hipMalloc((void**)&ctx->blk_table, sizeof(double) * (size_t)nc));
    ctx->blk_table_cap = nc;
  }
  double* tiles = (double*)ctx->scratch;
  double* mine = ctx->comm_active() ? (double*)ctx->comm_buf : (double*)((char*)ctx->scratch + a_tiles);   // [P] this rank's piece totals
  const double* all = mine;                                                                                // [G][P]
  const dim3 grid((unsigned)((size_t)P * tpb)), blk(tph_scan::THREADS);
  if (thr_dev) hipLaunchKernelGGL(k_seg_tile_sums<tph_scan::MASKED>, grid, blk, 0, ctx->stream, w_dev, rows, tpb, thr_dev, tiles);
  else hipLaunchKernelGGL(k_seg_tile_sums<tph_scan::PLAIN>, grid, blk, 0, ctx->stream, w_dev, rows, tpb, thr_dev, tiles);
  hipLaunchKernelGGL(k_seg_totals, dim3(P), dim3(256), 0, ctx->stream, tiles, tpb, mine);
  TPH_LAUNCH_CHECK();
  if (ctx->comm_active()) {
    if (tph_comm_allgather(ctx, 0, sizeof(double) * (size_t)P, P, TPH_DT_F64)) return -2;
    all = mine + P;
  }
  hipLaunchKernelGGL(k_block_table, dim3(1), dim3(256), (size_t)G * P <= 8000 ? sizeof(double) * (size_t)G * P : 0, ctx->stream, all, G, part.T, part.vl,
                     ctx->comm_active() ? ctx->rank : 0, ctx->blk_table);
  hipLaunchKernelGGL(k_seg_offsets, dim3(P), dim3(256), 0, ctx->stream, tiles, tpb, ctx->blk_table);
  if (thr_dev) hipLaunchKernelGGL(k_seg_apply<tph_scan::MASKED>, grid, blk, 0, ctx->stream, w_dev, rows, tpb, P, thr_dev, tiles, ctx->blk_table, cdf_dev);
  else hipLaunchKernelGGL(k_seg_apply<tph_scan::PLAIN>, grid, blk, 0, ctx->stream, w_dev, rows, tpb, P, thr_dev, tiles, ctx->blk_table, cdf_dev);
  TPH_LAUNCH_CHECK();
  ctx->blk_T = P; ctx->blk_rows = rows;
  return 0;
}

extern "C" int tph_cdf_global(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev, double* cdf_dev,
                              double* total_host) {
  TPH_REQUIRE(ctx && w_dev && cdf_dev && n > 0, "tph_cdf_global: bad argument");
  if (!ctx->comm_active()) {
    int rc = tph_cdf(ctx, w_dev, n, thr_dev, cdf_dev);
    if (rc) return rc;
    if (total_host) {
      TPH_HIP(hipMemcpyAsync(ctx->pinned, cdf_dev + (n - 1), sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      TPH_HIP(hipStreamSynchronize(ctx->stream));
      *total_host = ctx->pinned[0];
    }
    return 0;
  }
  int T; int64_t rows;
  if (tph_blocks(ctx, n, &T, &rows)) return -2;
  const tph_part part = tph_partition(ctx, n);
  if (cdf_pieces(ctx, w_dev, thr_dev, cdf_dev, part)) return -2;
  if (total_host) {
    TPH_HIP(hipMemcpyAsync(ctx->pinned, ctx->blk_table + 2 * (size_t)ctx->blk_T, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    *total_host = ctx->pinned[0];
  }
  return 0;
}

// which of my blocks owns global position p?  -1: another rank's.  INCL: multinomial walk (#{c <= p}: block with
// glo <= p < ghi), else systematic (#{c < p}: glo < p <= ghi).  Positions beyond either end go to the first / last block.
template <bool INCL>
__device__ __forceinline__ int owner_block(const double* __restrict__ table, int T, int rank, int world, double p) {
  int lo = 0, hi = T;                      // first t with glo[t] > p (INCL) or >= p
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const bool before = INCL ? (table[mid] <= p) : (table[mid] < p);
    if (before) lo = mid + 1; else hi = mid;
  }
  int t = lo - 1;
  if (t < 0) return rank == 0 ? 0 : -1;    // p at (or below) the origin: the first block of the global order
  const double ghi = table[T + t];
  const bool inside = INCL ? (p < ghi) : (p <= ghi);
  if (inside) return t;
  return (rank == world - 1 && t == T - 1) ? t : -1;   // p at (or beyond) the total: the last block of the global order
}

__global__ void __launch_bounds__(256) k_select_global(tph_cdf_index ix, const double* __restrict__ table, int T, int64_t rows,
                                                       int rank, int world, int64_t n_slots, int scheme, uint64_t seed,
                                                       uint32_t tick, uint32_t tag, double u0, double pscale,
                                                       int64_t* __restrict__ idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  double p;
  if (scheme == 0) {
    tph_rng g(seed, tick, tag, (uint64_t)i);
    double U, U1;
    g.uniform2(0, U, U1);
    p = U * table[2 * T];
  } else {
    p = (u0 + (double)i) / (double)n_slots * pscale;
  }
  const int t = scheme == 0 ? owner_block<true>(table, T, rank, world, p) : owner_block<false>(table, T, rank, world, p);
  if (t < 0) { idx[i] = -1; return; }
  int64_t k = scheme == 0 ? tph_count_below<false>(ix, 1.0, p) : tph_count_below<true>(ix, 1.0, p);
  const int64_t first = (int64_t)t * rows, last = first + rows - 1;
  idx[i] = k < first ? first : (k > last ? last : k);
}

extern "C" int tph_resample_select_global(tph_ctx* ctx, const double* cdf_dev, int64_t n, int64_t n_slots, int scheme,
                                          uint64_t seed, uint32_t tick, uint32_t tag, double u0, double pscale, int64_t* idx_dev) {
  TPH_REQUIRE(ctx && cdf_dev && idx_dev && n > 0 && n_slots > 0, "tph_resample_select_global: bad argument");
  TPH_REQUIRE(scheme == 0 || scheme == 1, "tph_resample_select_global: scheme must be 0 (multinomial) or 1 (systematic)");
  if (!ctx->comm_active()) {
    if (scheme == 0) return tph_resample_multinomial(ctx, cdf_dev, n, n_slots, seed, tick, tag, 0, idx_dev);
    return tph_resample_systematic(ctx, cdf_dev, n, n_slots, 0, n_slots, u0, pscale, idx_dev);
  }
  TPH_REQUIRE(ctx->blk_T > 0 && (int64_t)ctx->blk_T * ctx->blk_rows == n, "tph_resample_select_global: call tph_cdf_global on this history first");
  if (tph_scratch_reserve(ctx, sizeof(double) * tph_cdf_index_doubles(n))) return -1;
  tph_cdf_index ix;
  if (tph_cdf_index_build(ctx, cdf_dev, n, (double*)ctx->scratch, &ix)) return -1;
  hipLaunchKernelGGL(k_select_global, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, ctx->stream, ix, ctx->blk_table,
                     ctx->blk_T, ctx->blk_rows, ctx->rank, ctx->world, n_slots, scheme, seed, tick, tag, u0, pscale, idx_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) k_counts_global(tph_cdf_index ix, const double* __restrict__ table, int T, int64_t rows,
                                                       int rank, int world, const double* __restrict__ kept_count_dev, int factor,
                                                       int64_t n_draw_max, uint64_t seed, uint32_t tick, uint32_t tag,
                                                       int32_t* __restrict__ counts) {
  int64_t n_draw = kept_count_dev ? (int64_t)(kept_count_dev[0]) * factor : n_draw_max;
  if (n_draw > n_draw_max) n_draw = n_draw_max;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_draw; r += (int64_t)gridDim.x * blockDim.x) {   // as above
    tph_rng g(seed, tick, tag, (uint64_t)r);
    double U, U1;
    g.uniform2(0, U, U1);
    const double p = U * table[2 * T];
    const int t = owner_block<true>(table, T, rank, world, p);
    if (t < 0) continue;
    int64_t k = tph_count_below<false>(ix, 1.0, p);
    const int64_t first = (int64_t)t * rows, last = first + rows - 1;
    k = k < first ? first : (k > last ? last : k);
    atomicAdd(&counts[k], 1);
  }
}

extern "C" int tph_multinomial_counts_global(tph_ctx* ctx, const double* cdf_dev, int64_t n, const double* kept_count_dev,
                                             int factor, int64_t n_draw_max, uint64_t seed, uint32_t tick, uint32_t tag,
                                             int32_t* counts_dev) {
  TPH_REQUIRE(ctx && cdf_dev && counts_dev && n > 0 && n_draw_max > 0, "tph_multinomial_counts_global: bad argument");
  if (!ctx->comm_active())
    return tph_multinomial_counts(ctx, cdf_dev, n, kept_count_dev, factor, n_draw_max, seed, tick, tag, counts_dev);
  TPH_REQUIRE(ctx->blk_T > 0 && (int64_t)ctx->blk_T * ctx->blk_rows == n, "tph_multinomial_counts_global: call tph_cdf_global on this history first");
  TPH_HIP(hipMemsetAsync(counts_dev, 0, sizeof(int32_t) * (size_t)n, ctx->stream));
  TPH_HIP(hipMemsetAsync(counts_dev, 0, sizeof(int32_t) * (size_t)n, ctx->stream));
  if (tph_scratch_reserve(ctx, sizeof(double) * tph_cdf_index_doubles(n))) return -1;
  tph_cdf_index ix;
  if (tph_cdf_index_build(ctx, cdf_dev, n, (double*)ctx->scratch, &ix)) return -1;
  int64_t blocks = (n_draw_max + 255) / 256;
  if (blocks > 16 * (int64_t)ctx->n_simd) blocks = 16 * (int64_t)ctx->n_simd;
  hipLaunchKernelGGL(k_counts_global, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ix, ctx->blk_table,
                     ctx->blk_T, ctx->blk_rows, ctx->rank, ctx->world, kept_count_dev, factor, n_draw_max, seed, tick, tag, counts_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// ---- row-major mirror of the history (ctx->rows): records of 2 d + 1 doubles (u, x, logl) ------------------------------------
// A random history row costs 2 d + 1 scattered 8-byte reads in the dimension-major arrays -- one 64-byte sector each, 8.4x the
// algorithmic bytes by the counters (profiles/r02_roofline_table.json) -- but one contiguous 16 d + 8 byte record here.  The
// mirror is filled lazily from the dimension-major arrays (the rows appended since the last use), 64 rows per block through an
// LDS tile so that both sides are coalesced.
constexpr int ROWS_TILE = 64;
__global__ void __launch_bounds__(256) k_rows_pack(const double* __restrict__ hu, const double* __restrict__ hx,
                                                   const double* __restrict__ hl, int64_t cap, int d, int64_t off, int64_t n,
                                                   double* __restrict__ rows) {
  extern __shared__ double tile[];                                // [64][rec + 1]
  const int rec = 2 * d + 1, pitch = rec + 1;
  const int64_t i0 = off + (int64_t)blockIdx.x * ROWS_TILE;
  const int64_t end = off + n;
  for (int e = threadIdx.x; e < ROWS_TILE * rec; e += 256) {      // consecutive lanes: consecutive rows of one column
    const int c = e / ROWS_TILE, r = e - c * ROWS_TILE;
    const int64_t i = i0 + r;
    if (i < end) tile[r * pitch + c] = c < d ? hu[(size_t)c * cap + i] : (c < 2 * d ? hx[(size_t)(c - d) * cap + i] : hl[i]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ROWS_TILE * rec; e += 256) {      // consecutive lanes: consecutive fields of one record
    const int r = e / rec, c = e - r * rec;
    if (i0 + r < end) rows[(size_t)(i0 + r) * rec + c] = tile[r * pitch + c];
  }
}

// the mirror's memory goes back (it is a cache of the dimension-major arrays): under memory pressure, or when it would take more
// than its share of the device
void tph_rows_drop(tph_ctx* ctx) {
  if (!ctx->rows) return;
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->rows_vm.on()) tph_vm_release(&ctx->rows_vm);
  else (void)hipFree(ctx->rows);
  (void)hipGetLastError();
  ctx->rows = nullptr; ctx->rows_cap = 0; ctx->rows_size = 0;
  ctx->rows_mode = 0;
  ctx->stat_mem[2] += 1;
}

// may the mirror take `more` bytes on top of the `have` it holds?  Never more than a fifth of the device for the whole mirror,
// and never so much that less than 15 % of the device stays free (the fit's working set, the sort buffers and the caller's own
// tensors come out of that): a 100-D history of 10^8 rows does without a mirror -- its gather is a per-mille of an iteration.
static bool rows_budget_ok(size_t have, size_t more) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return true; }
  if (have + more > total_b / 5) return false;
  return free_b >= more && free_b - more >= total_b / 20 * 3;
}

const double* tph_rows_sync(tph_ctx* ctx) {
  if (!ctx->rows_mode || ctx->size <= 0) return nullptr;
  const int d = ctx->d, rec = 2 * d + 1;
  const size_t recb = sizeof(double) * (size_t)rec;
  if (ctx->rows_cap < ctx->size && ctx->hist_vm.on()) {
    // the history grows in a mapped range: so does the mirror (a set of one array), an eighth at a time
    tph_vm_set& v = ctx->rows_vm;
    const size_t g = v.on() ? v.piece : tph_vm_piece(ctx->device, (size_t)ctx->size * recb);
    int64_t want_rows = ctx->rows_cap + ctx->rows_cap / 8;
    if (want_rows < ctx->size) want_rows = ctx->size;
    size_t want = g ? ((size_t)want_rows * recb + g - 1) / g * g : 0;
    bool ok = g > 0 && rows_budget_ok(v.on() ? v.mapped : 0, want - (v.on() ? v.mapped : 0));
    if (!ok) tph_set_error("row-major mirror: %zu bytes would exceed its share of the device memory", want);
    if (ok && !v.on()) {
      if (ctx->rows) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->rows); ctx->rows = nullptr; ctx->rows_cap = 0; ctx->rows_size = 0; }
      const size_t va = ((size_t)ctx->cap * recb + g - 1) / g * g;
      ok = tph_vm_reserve(&v, ctx->device, 1, va > want ? va : want, g) == 0;
    }
    if (ok && want > v.stride) {
      size_t ns = v.stride;
      while (ns < want) ns *= 2;
      ok = tph_vm_restride(&v, ns, ctx->stream) == 0;
    }
    if (ok) ok = tph_vm_grow(&v, want) == 0;
    if (!ok) {
      (void)hipGetLastError();
      tph_rows_drop(ctx);
      ctx->rows_mode = 0;
      return nullptr;
    }
    ctx->rows = (double*)v.base;
    ctx->rows_cap = (int64_t)(v.mapped / recb);
  } else if (ctx->rows_cap < ctx->size) {                                // a plain allocation, grown with the history's own capacity
    double* fresh = nullptr;
    const int64_t nc = ctx->cap > ctx->size ? ctx->cap : ctx->size;
    if (!rows_budget_ok(0, (size_t)nc * recb) || hipMalloc((void**)&fresh, (size_t)nc * recb) != hipSuccess) {
      (void)hipGetLastError();
      tph_rows_drop(ctx);
      ctx->rows_mode = 0;                                         // no room for a mirror: the dimension-major gather from now on
      return nullptr;
    }
    if (ctx->rows) {
      if (ctx->rows_size > 0 &&
          hipMemcpyAsync(fresh, ctx->rows, (size_t)ctx->rows_size * recb, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        ctx->rows_size = 0;
      }
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipFree(ctx->rows);
    }
    ctx->rows = fresh;
    ctx->rows_cap = nc;
  }
  if (ctx->rows_size < ctx->size) {
    const int64_t n = ctx->size - ctx->rows_size;
    const size_t lds = sizeof(double) * ROWS_TILE * (size_t)(rec + 1);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k_rows_pack, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    hipLaunchKernelGGL(k_rows_pack, dim3((unsigned)((n + ROWS_TILE - 1) / ROWS_TILE)), dim3(256), lds, ctx->stream, ctx->u, ctx->x,
                       ctx->logl, ctx->cap, d, ctx->rows_size, n, ctx->rows);
    if (hipGetLastError() != hipSuccess) return nullptr;
    ctx->rows_size = ctx->size;
  }
  return ctx->rows;
}

// K7: gather rows of the history.  From the mirror: 64 output rows per block, records in (contiguous), LDS transpose,
// dimension-major out (coalesced).
__global__ void __launch_bounds__(256) k_gather_rows(const double* __restrict__ rows, int d, const int64_t* __restrict__ idx,
                                                     int64_t n_out, double* __restrict__ u, double* __restrict__ x,
                                                     double* __restrict__ l, int64_t ld) {
  extern __shared__ double tile[];
  const int rec = 2 * d + 1, pitch = rec + 1;
  const int64_t i0 = (int64_t)blockIdx.x * ROWS_TILE;
  for (int e0 = threadIdx.x; e0 < ROWS_TILE * rec; e0 += 4 * 256) {      // four indexed loads in flight per lane
    double v[4];
    int at[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = e0 + k * 256;
      const int r = e / rec, c = e - r * rec;
      const bool in = e < ROWS_TILE * rec && i0 + r < n_out;
      at[k] = in ? r * pitch + c : -1;
      v[k] = in ? rows[(size_t)idx[i0 + r] * rec + c] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (at[k] >= 0) tile[at[k]] = v[k];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ROWS_TILE * rec; e += 256) {
    const int c = e / ROWS_TILE, r = e - c * ROWS_TILE;
    const int64_t i = i0 + r;
    if (i >= n_out) continue;
    const double v = tile[r * pitch + c];
    if (c < d) u[(size_t)c * ld + i] = v;
    else if (c < 2 * d) x[(size_t)(c - d) * ld + i] = v;
    else l[i] = v;
  }
}
// ... and from the dimension-major arrays (no mirror): coalesced writes, indexed reads
__global__ void __launch_bounds__(256) k_gather(const double* __restrict__ hu, const double* __restrict__ hx,
                                                const double* __restrict__ hl, int64_t cap, int d,
                                                const int64_t* __restrict__ idx, int64_t n_out,
                                                double* __restrict__ u, double* __restrict__ x, double* __restrict__ l,
                                                int64_t ld) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int64_t s = idx[i];
  for (int j = 0; j < d; ++j) {
    u[(size_t)j * ld + i] = hu[(size_t)j * cap + s];
    x[(size_t)j * ld + i] = hx[(size_t)j * cap + s];
  }
  l[i] = hl[s];
}

extern "C" int tph_gather(tph_ctx* ctx, const int64_t* idx_dev, int64_t n_out, double* u_out, double* x_out,
                          double* logl_out, int64_t ld_out) {
  TPH_REQUIRE(ctx && idx_dev && u_out && x_out && logl_out, "tph_gather: NULL argument");
  TPH_REQUIRE(n_out > 0 && ld_out >= n_out && ctx->size > 0, "tph_gather: bad sizes");
  // the mirror pays once the gather is a sizeable fraction of the rows it has to pack first
  const double* rows = (ctx->rows_mode && 8 * n_out >= ctx->size - ctx->rows_size) ? tph_rows_sync(ctx) : nullptr;
  if (rows) {
    const int rec = 2 * ctx->d + 1;
    const size_t lds = sizeof(double) * ROWS_TILE * (size_t)(rec + 1);
    if (lds > 64 * 1024) TPH_HIP(hipFuncSetAttribute((const void*)k_gather_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n_out + ROWS_TILE - 1) / ROWS_TILE)), dim3(256), lds, ctx->stream, rows, ctx->d,
                       idx_dev, n_out, u_out, x_out, logl_out, ld_out);
  } else {
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx->stream, ctx->u, ctx->x, ctx->logl,
                       ctx->cap, ctx->d, idx_dev, n_out, u_out, x_out, logl_out, ld_out);
  }
  TPH_LAUNCH_CHECK();
  return 0;
}

// posterior extraction (core.py:187-242 after the trim): selected history rows as ROW-MAJOR (m, d) parameters, their
// log-likelihoods and (rescaled) weights, so that only the kept rows cross PCIe and the host gets the layout it
// returns.  64 rows x 32 dimensions per LDS tile: indexed reads along the rows, 256-byte row segments out.
constexpr int POST_R = 64, POST_C = 32;
__global__ void __launch_bounds__(256) k_posterior_rows(const double* __restrict__ hx, const double* __restrict__ hl,
                                                        int64_t cap, int d, const int64_t* __restrict__ idx, int64_t m,
                                                        const double* __restrict__ w, double wdiv,
                                                        double* __restrict__ x_out, double* __restrict__ l_out,
                                                        double* __restrict__ w_out) {
  __shared__ double tile[POST_R][POST_C + 1];
  __shared__ int64_t rows[POST_R];
  const int64_t row0 = (int64_t)blockIdx.x * POST_R;
  const int t = threadIdx.x;
  if (t < POST_R) {
    int64_t i = row0 + t;
    int64_t s = i < m ? (idx ? idx[i] : i) : -1;
    rows[t] = s;
    if (s >= 0) {
      l_out[i] = hl[s];
      if (w_out) w_out[i] = w[s] / wdiv;
    }
  }
  __syncthreads();
  for (int j0 = 0; j0 < d; j0 += POST_C) {
    for (int e = t; e < POST_R * POST_C; e += 256) {      // row fastest: neighbouring lanes read neighbouring history rows
      int r = e % POST_R, c = e / POST_R;
      int64_t s = rows[r];
      if (s >= 0 && j0 + c < d) tile[r][c] = hx[(size_t)(j0 + c) * cap + s];
    }
    __syncthreads();
    for (int e = t; e < POST_R * POST_C; e += 256) {      // dimension fastest: contiguous segments of the output rows
      int c = e % POST_C, r = e / POST_C;
      if (rows[r] >= 0 && j0 + c < d) x_out[(size_t)(row0 + r) * d + j0 + c] = tile[r][c];
    }
    __syncthreads();
  }
}

extern "C" int tph_posterior_rows(tph_ctx* ctx, int key, const int64_t* idx_dev, int64_t m, const double* w_dev, double wdiv,
                                  double* x_out, double* logl_out, double* w_out) {
  TPH_REQUIRE(ctx && x_out && logl_out, "tph_posterior_rows: NULL argument");
  TPH_REQUIRE(key == TPH_KEY_U || key == TPH_KEY_X, "tph_posterior_rows: key must be TPH_KEY_U or TPH_KEY_X");
  TPH_REQUIRE(m > 0 && (idx_dev || m <= ctx->size), "tph_posterior_rows: bad sizes");
  TPH_REQUIRE(!w_out || w_dev, "tph_posterior_rows: w_out needs w_dev");
  hipLaunchKernelGGL(k_posterior_rows, dim3((unsigned)((m + POST_R - 1) / POST_R)), dim3(256), 0, ctx->stream,
                     key == TPH_KEY_U ? ctx->u : ctx->x, ctx->logl, ctx->cap, ctx->d, idx_dev, m, w_dev, wdiv, x_out, logl_out,
                     w_out);
  TPH_LAUNCH_CHECK();
  return 0;
}

// out[i] = a[b[i]]  (indices of a resampling drawn over an already compacted selection)
__global__ void __launch_bounds__(256) k_index_compose(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t m,
                                                       int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) out[i] = a[b[i]];
}
extern "C" int tph_index_compose(tph_ctx* ctx, const int64_t* a_dev, const int64_t* b_dev, int64_t m, int64_t* out_dev) {
  TPH_REQUIRE(ctx && a_dev && b_dev && out_dev && m > 0, "tph_index_compose: bad argument");
  hipLaunchKernelGGL(k_index_compose, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, a_dev, b_dev, m, out_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// multiplicities of `factor * kept_count` multinomial draws (kept_count lives on the device: no host sync)
__global__ void __launch_bounds__(256) k_multinomial_counts(tph_cdf_index ix,
                                                            const double* __restrict__ kept_count_dev, int factor,
                                                            int64_t n_draw_max, uint64_t seed, uint32_t tick, uint32_t tag,
                                                            int32_t* __restrict__ counts) {
  // grid strides: the number of draws is only known on the device, and a grid sized for n_draw_max (4 x the history) whose
  // blocks find nothing to do is not free -- 10^6 empty workgroups are ~0.25 ms
  int64_t n_draw = kept_count_dev ? (int64_t)(kept_count_dev[0]) * factor : n_draw_max;
  if (n_draw > n_draw_max) n_draw = n_draw_max;
  const int64_t n = ix.n[0];
  const double total = ix.lvl[0][n - 1];
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_draw; r += (int64_t)gridDim.x * blockDim.x) {
    tph_rng g(seed, tick, tag, (uint64_t)r);
    double U, U1;
    g.uniform2(0, U, U1);
    int64_t k = tph_count_below<false>(ix, 1.0, U * total);      // (as k_counts_global)
    if (k >= n) k = n - 1;
    atomicAdd(&counts[k], 1);
  }
}

// Many draws (millions: the x4 up-sampling of a 10^6-particle run): each lookup above is ~4.7 random 64-byte sectors
// (FETCH_SIZE: 3.0 GB per call at 2.6 x 10^7 rows) and the kernel is bound by exactly that.  The counts do not depend on the
// ORDER of the draws, so they are generated as their 53-bit integers k (U = k 2^-53, tph_k53), sorted on their top MC_SORT_BITS
// bits, and counted by the OWNERS of the rows: a workgroup takes a tile of MC_TILE consecutive rows of the cdf into LDS, reads
// the stretch of the sorted draws that can fall into it (k_mc_bounds: the buckets of the tile's first and last cumulative weight,
// one bucket of margin either side -- inside a bucket the draws are in no order, and the bucket of a cumulative weight is only
// known to rounding), keeps those whose row IS in the tile (the predicate of every other lookup: a_i <= U total, the tile's first
// and last rows decide membership, so a draw is counted by exactly one tile), finds the row by bisection in LDS and counts it with
// an LDS atomic; the tile's counts leave in one coalesced store.  Every byte of the cdf, of the draws and of the counts crosses
// HBM once, in streams (the draws of the margins twice); no global atomic, no dependent global load outside k_mc_bounds.  (Until
// round 5 a thread walked 16 consecutive sorted draws by galloping searches in global memory: 482 us at 2.6 x 10^7 rows, 0.12 of
// the HBM rate.)
constexpr int MC_TILE = 2048;             // rows of the cdf per workgroup: 16 KiB + 8 KiB of counters in LDS, six workgroups per CU
constexpr int64_t MC_SORT_MIN = 1 << 21;      // from 2 x 10^6 draws on the sorted path wins at every history size (1 - 27 x 10^6 rows:
                                              // 123-215 us + the host read against 158-307 us; below, the sort alone costs more: profiles/r05_upsample_ab.jsonl)
constexpr int MC_SORT_BITS = 16;          // the draws are sorted on the top MC_SORT_BITS bits of their 53-bit integers ...
constexpr int MC_SORT_LO = 53 - MC_SORT_BITS;      // ... i.e. on bits [MC_SORT_LO, 53): 2^16 buckets, two passes of the radix sort
__global__ void __launch_bounds__(256) k_mc_draws(int64_t n_draw, uint64_t seed, uint32_t tick, uint32_t tag,
                                                  uint64_t* __restrict__ keys) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_draw; r += (int64_t)gridDim.x * blockDim.x) {
    tph_rng g(seed, tick, tag, (uint64_t)r);
    const tph_u4 q = tph_philox(g.item, 0, g.tick, g.tag, g.k0, g.k1);      // uniform2(0, U, .): U = tph_k53(q.x, q.y) 2^-53
    keys[r] = ((uint64_t)(q.x >> 5) << 26) | (uint64_t)(q.y >> 6);
  }
}
// first position of the sorted draws whose bucket is >= b
__device__ __forceinline__ int64_t mc_lower_bound(const uint64_t* __restrict__ keys, int64_t n_draw, int64_t b) {
  int64_t lo = 0, hi = n_draw;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)(keys[mid] >> MC_SORT_LO) < b) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// pos[2 j], pos[2 j + 1]: where the draws that may belong to the tiles below / above boundary j (between rows j MC_TILE - 1 and
// j MC_TILE) end / begin.  The boundary's cumulative weight v lies in bucket floor(v / total 2^53) >> MC_SORT_LO up to a few units
// of 2^-53 (the predicate rounds U total once): tiles below it read through the NEXT bucket, tiles above it from the one BEFORE.
__global__ void __launch_bounds__(256) k_mc_bounds(const double* __restrict__ a, int64_t n, int64_t ntiles,
                                                   const uint64_t* __restrict__ keys, int64_t n_draw, int64_t* __restrict__ pos) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * (ntiles + 1)) return;
  const int64_t j = e >> 1;
  const bool upper = e & 1;                     // 0: end of the tiles below; 1: start of the tiles above
  const double total = a[n - 1];
  if (j == 0 || j == ntiles || !(total > 0.0) || !(total < INFINITY)) {      // the ends; a degenerate cdf: every tile reads every draw
    pos[e] = (j == 0 || (j < ntiles && upper)) ? 0 : n_draw;
    return;
  }
  const double f = (a[j * MC_TILE - 1] / total) * 0x1.0p53;
  const int64_t kb = (f >= 0x1.0p53 ? ((int64_t)1 << 53) - 1 : (int64_t)f) >> MC_SORT_LO;
  pos[e] = upper ? mc_lower_bound(keys, n_draw, kb - 1) : mc_lower_bound(keys, n_draw, kb + 2);
}
constexpr int MC_BATCH = 8;               // draws a lane requests before it waits for the first (a tile's stretch is ~2 x 10^3 draws)
__global__ void __launch_bounds__(256) k_mc_tiles(const double* __restrict__ a, int64_t n, const uint64_t* __restrict__ keys,
                                                  const int64_t* __restrict__ pos, int32_t* __restrict__ counts) {
  __shared__ double s_a[MC_TILE];
  __shared__ int s_c[MC_TILE];
  const int64_t t = blockIdx.x, r0 = t * MC_TILE;
  const int rows = (int)(n - r0 < MC_TILE ? n - r0 : MC_TILE);
  const int64_t k0 = pos[2 * t + 1], k1 = pos[2 * (t + 1)];
  if (k0 >= k1) {                                           // no draw can fall here (the flat stretches of a trimmed history)
    for (int i = threadIdx.x; i < rows; i += 256) counts[r0 + i] = 0;
    return;
  }
  // the first batch of draws is requested together with the tile (one memory round trip for both)
  uint64_t kv[MC_BATCH];
#pragma unroll
  for (int j = 0; j < MC_BATCH; ++j) {
    const int64_t k = k0 + threadIdx.x + (int64_t)j * 256;
    kv[j] = k < k1 ? keys[k] : ~0ull;
  }
  for (int i = threadIdx.x; i < MC_TILE; i += 256) {
    s_a[i] = i < rows ? a[r0 + i] : INFINITY;               // (beyond the end: never <= a draw)
    s_c[i] = 0;
  }
  const double total = a[n - 1];
  const bool has_below = r0 > 0, last = r0 + rows >= n;
  const double below = has_below ? a[r0 - 1] : 0.0;
  __syncthreads();
  const double top = s_a[rows - 1];
  for (int64_t kb = k0; kb < k1; kb += 256 * MC_BATCH) {
    // the batch's bisections side by side, a fixed eleven steps each (the tile is padded with +inf): eight independent LDS reads
    // per step instead of a chain of eleven per draw
    double pp[MC_BATCH];
    bool mine[MC_BATCH];
    int lo[MC_BATCH];
#pragma unroll
    for (int j = 0; j < MC_BATCH; ++j) {
      pp[j] = ((double)kv[j] * 0x1.0p-53) * total;          // (as k_counts_global: one predicate on any number of GPUs)
      mine[j] = kv[j] != ~0ull                              // (a draw is < 2^53)
                && !(has_below && !(below <= pp[j]))        // else its row lies in a tile below
                && !(!last && top <= pp[j]);                // ... above
      lo[j] = 0;                                            // #{i in the tile : a_i <= pp}
    }
#pragma unroll
    for (int half = MC_TILE / 2; half >= 1; half >>= 1)
#pragma unroll
      for (int j = 0; j < MC_BATCH; ++j) lo[j] += (s_a[lo[j] + half - 1] <= pp[j]) ? half : 0;
#pragma unroll
    for (int j = 0; j < MC_BATCH; ++j)
      if (mine[j]) atomicAdd(&s_c[lo[j] < rows ? lo[j] : rows - 1], 1);     // (only the last tile: a draw at the very top counts for the last row)
    const int64_t nb = kb + 256 * MC_BATCH;
    if (nb < k1) {
#pragma unroll
      for (int j = 0; j < MC_BATCH; ++j) {
        const int64_t k = nb + threadIdx.x + (int64_t)j * 256;
        kv[j] = k < k1 ? keys[k] : ~0ull;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows; i += 256) counts[r0 + i] = s_c[i];
}

extern "C" int tph_multinomial_counts(tph_ctx* ctx, const double* cdf_dev, int64_t n, const double* kept_count_dev,
                                      int factor, int64_t n_draw_max, uint64_t seed, uint32_t tick, uint32_t tag,
                                      int32_t* counts_dev) {
  TPH_REQUIRE(ctx && cdf_dev && counts_dev && n > 0 && n_draw_max > 0, "tph_multinomial_counts: bad argument");
  const int64_t sort_min = ctx->mc_sorted > 1 ? (int64_t)ctx->mc_sorted : MC_SORT_MIN;     // > 1: the threshold itself (tests)
  if (ctx->mc_sorted && n_draw_max >= sort_min) {
    // the sort is sized on the host: one 8-byte read of the count (the stream has to drain once; ~20 us against ~1 ms)
    int64_t n_draw = n_draw_max;
    if (kept_count_dev) {
      TPH_HIP(hipMemcpyAsync(ctx->pinned, kept_count_dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      TPH_HIP(hipStreamSynchronize(ctx->stream));
      n_draw = (int64_t)ctx->pinned[0] * factor;
      if (n_draw > n_draw_max) n_draw = n_draw_max;
    }
    if (n_draw <= 0) {
      TPH_HIP(hipMemsetAsync(counts_dev, 0, sizeof(int32_t) * (size_t)n, ctx->stream));
      return 0;
    }
    if (n_draw >= sort_min) {                               // (the tiles' owners write every row's count: no clearing)
      size_t temp_bytes = 0;
      uint64_t* nullk = nullptr;
      TPH_HIP(rocprim::radix_sort_keys(nullptr, temp_bytes, nullk, nullk, (size_t)n_draw, MC_SORT_LO, 53, ctx->stream));
      const int64_t ntiles = (n + MC_TILE - 1) / MC_TILE;
      const size_t a_pos = (sizeof(int64_t) * 2 * (size_t)(ntiles + 1) + 255) / 256 * 256;
      const size_t a_keys = (sizeof(uint64_t) * (size_t)n_draw + 255) / 256 * 256;
      if (tph_scratch_reserve(ctx, a_pos + 2 * a_keys + temp_bytes)) return -1;
      char* base = (char*)ctx->scratch;
      int64_t* pos = (int64_t*)base;
      uint64_t* keys = (uint64_t*)(base + a_pos);
      uint64_t* sorted = (uint64_t*)(base + a_pos + a_keys);
      void* tmp = base + a_pos + 2 * a_keys;
      int64_t blocks = (n_draw + 255) / 256;
      if (blocks > 16 * (int64_t)ctx->n_simd) blocks = 16 * (int64_t)ctx->n_simd;
      hipLaunchKernelGGL(k_mc_draws, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, n_draw, seed, tick, tag, keys);
      TPH_HIP(rocprim::radix_sort_keys(tmp, temp_bytes, keys, sorted, (size_t)n_draw, MC_SORT_LO, 53, ctx->stream));
      hipLaunchKernelGGL(k_mc_bounds, dim3((unsigned)((2 * (ntiles + 1) + 255) / 256)), dim3(256), 0, ctx->stream, cdf_dev, n, ntiles,
                         (const uint64_t*)sorted, n_draw, pos);
      hipLaunchKernelGGL(k_mc_tiles, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, cdf_dev, n, (const uint64_t*)sorted,
                         (const int64_t*)pos, counts_dev);
      TPH_LAUNCH_CHECK();
      return 0;
    }
  }
  TPH_HIP(hipMemsetAsync(counts_dev, 0, sizeof(int32_t) * (size_t)n, ctx->stream));
  if (tph_scratch_reserve(ctx, sizeof(double) * tph_cdf_index_doubles(n))) return -1;
  tph_cdf_index ix;
  if (tph_cdf_index_build(ctx, cdf_dev, n, (double*)ctx->scratch, &ix)) return -1;
  int64_t blocks = (n_draw_max + 255) / 256;
  if (blocks > 16 * (int64_t)ctx->n_simd) blocks = 16 * (int64_t)ctx->n_simd;      // 8 resident workgroups per CU, twice over
  hipLaunchKernelGGL(k_multinomial_counts, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ix, kept_count_dev, factor,
                     n_draw_max, seed, tick, tag, counts_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_resample(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
