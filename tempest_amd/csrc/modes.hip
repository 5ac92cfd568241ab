// Weight trimming, proposal fit (median + covariance of the up-sampled set, Cholesky/inverse) and the
// volume-variation diagnostic.
// Reference: tempest/tools.py:10-55 (trim_weights), :58-117 (volume_variation); tempest/student.py:6-116
// (fit_mvstud; effective form = per-dimension median, MLE covariance + diag(var)/n, nu=inf -- SURVEY.md F5);
// tempest/modes.py:58-119 (chol/inv with ridge), :131-288 (x4 multinomial up-sampling then fit).
#include "common.h"
#include "scan.h"
#include "tri.h"

#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

// ------------------------------------------------------------------------------------------- trimming
// All `bins` candidates of the reference's 99 -> 0 percentile walk at once on the sorted weights S with
// prefix sums P1 (of S) and P2 (of S^2): candidate i keeps {w >= thr_i}; it passes if
// ESS(kept)/ESS(all) >= ess; the answer is the largest passing i (tools.py:42-53).
__device__ __forceinline__ void trim_threshold_of(const double* __restrict__ S, int64_t n, int bins, double step, int i, double& thr) {
  double p = (i == bins - 1 && bins > 1) ? 99.0 : __dmul_rn((double)i, step);   // np.linspace(0, 99, bins)[i]
  double q = p / 100.0;                                                         // np.percentile: q / 100
  // numpy method 'linear': virtual index = (n - 1) * q  (lib/_function_base_impl.py, _QuantileMethods)
  double vi = __dmul_rn((double)(n - 1), q);
  if (vi >= (double)(n - 1)) thr = S[n - 1];
  else if (vi < 0.0) thr = S[0];
  else {
    double pf = floor(vi);
    int64_t pi = (int64_t)pf;
    double g = vi - pf;
    double a = S[pi], b = S[pi + 1];
    double diff = b - a;   // numpy _lerp
    thr = (g >= 0.5) ? __dadd_rn(b, -__dmul_rn(diff, __dadd_rn(1.0, -g))) : __dadd_rn(a, __dmul_rn(diff, g));
  }
}

// The pass test needs, per candidate, the sums of w and w^2 over the kept rows [first_kept_i, n) of the sorted array: the
// candidates' first-kept rows cut S into <= bins contiguous segments, so ONE streaming pass that sums every segment replaces
// the two prefix scans this used to take (8 bytes read per row instead of 2 x (8 read + 8 read + 8 written); 0.35 -> 0.06 ms
// per iteration at 1 048 576 particles).  Deterministic: a block sums its contiguous range of rows split at the segment
// boundaries inside it (slot = block + segment is unique along the staircase of overlapping pairs), a wave per segment adds its
// slots, suffix sums from the top segment down give the kept sums.
// (one wave per candidate: its 64 lanes probe the ends of 64 equal stretches of the remaining range at once -- five rounds of
// one memory round trip each at 2.6 x 10^7 rows, where a lane bisecting alone makes twenty-five; same predicate, same answer)
__global__ void __launch_bounds__(256) k_trim_bounds(const double* __restrict__ S, int64_t n, int bins, int64_t* __restrict__ first,
                                                     double* __restrict__ thrs) {
  const double step = bins > 1 ? 99.0 / (double)(bins - 1) : 0.0;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x == 0) first[bins] = n;
  if (i >= bins) return;                               // (the whole wave)
  double thr;
  trim_threshold_of(S, n, bins, step, i, thr);
  int64_t lo = 0, hi = n;                              // first k with S[k] >= thr lies in [lo, hi]
  for (;;) {
    const int64_t width = hi - lo;
    if (width <= 0) break;
    const int64_t st = (width + 63) / 64;
    const int64_t end = lo + (int64_t)(lane + 1) * st;                    // one past my stretch
    const int64_t at = (end < hi ? end : hi) - 1;
    const bool valid = lo + (int64_t)lane * st < hi;
    const bool below = valid && S[at] < thr;                              // then the whole stretch is below
    const int c = __popcll(__ballot(below));                              // S ascending: the first c stretches
    const int64_t nlo = lo + (int64_t)c * st;
    if (st == 1) { lo = nlo < hi ? nlo : hi; break; }
    if (nlo >= hi) { lo = hi; break; }
    hi = nlo + st < hi ? nlo + st : hi;
    lo = nlo;
  }
  if (lane == 0) { first[i] = lo; thrs[i] = thr; }
}

// segment of row r: the last i with first[i] <= r (first[] is non-decreasing, first[0] = 0)
__device__ __forceinline__ int trim_segment(const int64_t* __restrict__ first, int bins, int64_t r) {
  int lo = 0, hi = bins;                       // answer in [lo, hi)
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (first[mid] <= r) lo = mid; else hi = mid; }
  return lo;
}

__global__ void __launch_bounds__(256) k_trim_segsums(const double* __restrict__ S, int64_t n, int64_t per,
                                                      const int64_t* __restrict__ first, int bins, double* __restrict__ slots) {
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
  if (r0 >= r1) return;
  __shared__ double sh[4];
  const int j0 = trim_segment(first, bins, r0), j1 = trim_segment(first, bins, r1 - 1);
  for (int j = j0; j <= j1; ++j) {
    const int64_t a = first[j] > r0 ? first[j] : r0, b = first[j + 1] < r1 ? first[j + 1] : r1;
    if (a >= b) continue;                          // an empty segment (two candidates with the same first kept row)
    double t1 = 0.0, t2 = 0.0;
    int64_t i = a + threadIdx.x;
    for (; i + 768 < b; i += 1024) {               // four loads in flight per lane
      const double v0 = S[i], v1 = S[i + 256], v2 = S[i + 512], v3 = S[i + 768];
      t1 += (v0 + v1) + (v2 + v3);
      t2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
    for (; i < b; i += 256) {
      const double v = S[i];
      t1 += v;
      t2 = fma(v, v, t2);
    }
    t1 = tph_block_sum(t1, sh);
    t2 = tph_block_sum(t2, sh);
    if (threadIdx.x == 0) {
      const size_t slot = (size_t)blockIdx.x + (size_t)j;
      slots[2 * slot] = t1;
      slots[2 * slot + 1] = t2;
    }
  }
}

// sums of segment j: its slots are those of the consecutive blocks first[j] / per ... (first[j + 1] - 1) / per, i.e. slots
// block + j; one wave per segment adds them, lanes striding over the run and a fixed shuffle tree at the end (deterministic)
__global__ void __launch_bounds__(64) k_trim_segreduce(const double* __restrict__ slots, const int64_t* __restrict__ first,
                                                       int64_t per, double* __restrict__ seg) {
  const int j = blockIdx.x;
  const int64_t a = first[j], b = first[j + 1];
  double t1 = 0.0, t2 = 0.0;
  if (b > a) {
    const int64_t s0 = a / per + j, s1 = (b - 1) / per + j;
    for (int64_t s = s0 + threadIdx.x; s <= s1; s += 64) { t1 += slots[2 * s]; t2 += slots[2 * s + 1]; }
  }
  t1 = tph_wave_sum(t1);
  t2 = tph_wave_sum(t2);
  if (threadIdx.x == 0) { seg[2 * j] = t1; seg[2 * j + 1] = t2; }
}

// kept sums of candidate i = segments i .. bins-1 (suffix sums from the top), the pass test, the largest passing candidate.
// With the reference's 1000 candidates a single thread walking them costs 115 us: every thread takes a run of consecutive
// candidates (suffix inside the run, then the totals of the runs above it, added in a fixed order).
__global__ void __launch_bounds__(256) k_trim_decide(const double* __restrict__ seg, const int64_t* __restrict__ first,
                                                     const double* __restrict__ thrs, int64_t n, double ess, int bins,
                                                     double* __restrict__ out) {
  extern __shared__ double sh[];
  double* s_seg = sh;                                    // [2 bins]
  double* s_tot = sh + 2 * (size_t)bins;                 // [2 * 256] run totals
  __shared__ int s_best;
  for (int e = threadIdx.x; e < 2 * bins; e += blockDim.x) s_seg[e] = seg[e];
  if (threadIdx.x == 0) s_best = 0;                      // i = 0 keeps everything and always passes
  __syncthreads();
  const int run = (bins + 255) / 256;
  const int lo = threadIdx.x * run < bins ? threadIdx.x * run : bins;
  const int hi = lo + run < bins ? lo + run : bins;
  double k1 = 0.0, k2 = 0.0;
  for (int i = hi - 1; i >= lo; --i) {
    k1 += s_seg[2 * i];
    k2 += s_seg[2 * i + 1];
    s_seg[2 * i] = k1;
    s_seg[2 * i + 1] = k2;
  }
  s_tot[2 * threadIdx.x] = k1;
  s_tot[2 * threadIdx.x + 1] = k2;
  __syncthreads();
  double o1 = 0.0, o2 = 0.0;                             // the runs above mine, from the top down
  for (int t = 255; t > (int)threadIdx.x; --t) { o1 += s_tot[2 * t]; o2 += s_tot[2 * t + 1]; }
  for (int i = lo; i < hi; ++i) { s_seg[2 * i] += o1; s_seg[2 * i + 1] += o2; }
  __syncthreads();
  const double ess_total = (s_seg[0] * s_seg[0]) / s_seg[1];
  int mine = 0;
  for (int i = lo; i < hi; ++i)
    if (i > 0 && ((s_seg[2 * i] * s_seg[2 * i]) / s_seg[2 * i + 1]) / ess_total >= ess) mine = i;
  if (mine > 0) atomicMax(&s_best, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int best = s_best;
    out[0] = thrs[best];
    out[1] = s_seg[2 * best];
    out[2] = (double)(n - first[best]);
    out[3] = ess_total;
  }
}

extern "C" int tph_trim_threshold(tph_ctx* ctx, const double* w_dev, int64_t n, double ess, int bins, double* out_dev,
                                  double* out_host) {
  TPH_REQUIRE(ctx && w_dev && out_dev && n > 0 && bins >= 1, "tph_trim_threshold: bad argument");
  size_t temp_bytes = 0;
  double* nullk = nullptr;
  TPH_HIP(rocprim::radix_sort_keys(nullptr, temp_bytes, w_dev, nullk, (size_t)n, 0, 64, ctx->stream));
  int64_t want = (n + 256 * 64 - 1) / (256 * 64);                 // >= 16 384 rows per block, at most 1024 blocks
  const int nblk = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  const int64_t per = (n + nblk - 1) / nblk;
  const int nslots = nblk + bins;
  auto al = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t a_n = al((size_t)n * sizeof(double)), a_tmp = al(temp_bytes), a_first = al(sizeof(int64_t) * (size_t)(bins + 1)),
               a_thr = al(sizeof(double) * (size_t)bins), a_slots = al(sizeof(double) * 2 * (size_t)nslots),
               a_seg = al(sizeof(double) * 2 * (size_t)bins);
  const size_t lds = sizeof(double) * (2 * (size_t)bins + 512);
  TPH_REQUIRE(lds <= 150 * 1024, "tph_trim_threshold: bins=%d: at most ~9000 candidates", bins);
  if (tph_scratch_reserve(ctx, a_n + a_tmp + a_first + a_thr + a_slots + a_seg)) return -1;
  char* base = (char*)ctx->scratch;
  double* S = (double*)base;
  void* tmp = base + a_n;
  int64_t* first = (int64_t*)(base + a_n + a_tmp);
  double* thrs = (double*)((char*)first + a_first);
  double* slots = (double*)((char*)thrs + a_thr);
  double* seg = (double*)((char*)slots + a_slots);
  TPH_HIP(rocprim::radix_sort_keys(tmp, temp_bytes, w_dev, S, (size_t)n, 0, 64, ctx->stream));
  hipLaunchKernelGGL(k_trim_bounds, dim3((bins + 3) / 4), dim3(256), 0, ctx->stream, S, n, bins, first, thrs);
  hipLaunchKernelGGL(k_trim_segsums, dim3(nblk), dim3(256), 0, ctx->stream, S, n, per, first, bins, slots);
  hipLaunchKernelGGL(k_trim_segreduce, dim3(bins), dim3(64), 0, ctx->stream, slots, first, per, seg);
  if (lds > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_trim_decide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_trim_decide, dim3(1), dim3(256), lds, ctx->stream, seg, first, thrs, n, ess, bins, out_dev);
  TPH_LAUNCH_CHECK();
  if (out_host) {
    TPH_HIP(hipMemcpyAsync(ctx->pinned, out_dev, sizeof(double) * 4, hipMemcpyDeviceToHost, ctx->stream));
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) out_host[i] = ctx->pinned[i];
  }
  return 0;
}

// ---- GLOBAL trimming threshold of a sharded weight vector (tph_comm_attach) -------------------------------------------------
// The reference's threshold is a percentile of ALL weights (np.percentile, linear interpolation between two neighbouring
// order statistics) and its pass test needs the sums of w and w^2 above it (tools.py:42-53).  Every rank sorts its own
// weights; the 2 x bins global order statistics are found EXACTLY by a radix descent over the 64-bit patterns of the
// (non-negative) doubles, 8 bits per round: per target and digit, each rank counts its weights whose pattern continues the
// target's prefix with that digit (two binary searches in its sorted array per digit boundary), the counts are all-reduced,
// and every rank picks the digit holding the target's rank -- 8 all-reduces of [2 bins][256] counters, no weight ever moves.
// Then the candidates' thresholds are interpolated with NumPy's arithmetic on the exact order statistics (bit-identical to
// the one-GPU threshold), the local suffix sums above each threshold are all-reduced, and the largest passing candidate wins.
constexpr int SEL_BITS = 8, SEL_DIGITS = 1 << SEL_BITS, SEL_ROUNDS = 64 / SEL_BITS;

__device__ __forceinline__ int64_t lower_bound_key(const double* __restrict__ S, int64_t n, unsigned long long key) {
  int64_t lo = 0, hi = n;                  // first k with bits(S[k]) >= key   (S ascending, non-negative)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((unsigned long long)__double_as_longlong(S[mid]) < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// target ranks: candidate i -> virtual index (n-1) q_i of numpy's 'linear' method: order statistics floor(vi), floor(vi)+1
__global__ void k_trim_targets(const long long* __restrict__ n_global_dev, int bins, long long* __restrict__ remaining,
                               unsigned long long* __restrict__ prefix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bins) return;
  const long long n = n_global_dev[0];
  const double step = bins > 1 ? 99.0 / (double)(bins - 1) : 0.0;
  const double p = (i == bins - 1 && bins > 1) ? 99.0 : __dmul_rn((double)i, step);
  const double vi = __dmul_rn((double)(n - 1), p / 100.0);
  long long pi = (long long)floor(vi);
  if (pi < 0) pi = 0;
  if (pi > n - 1) pi = n - 1;
  remaining[2 * i] = pi;
  remaining[2 * i + 1] = pi + 1 < n ? pi + 1 : n - 1;
  prefix[2 * i] = 0ull;
  prefix[2 * i + 1] = 0ull;
}
// counts[q][v] = #{local weights whose pattern lies in [prefix_q + v << shift, prefix_q + (v+1) << shift)}
template <typename CT>
__global__ void __launch_bounds__(256) k_select_count(const double* __restrict__ S, int64_t n, const unsigned long long* __restrict__ prefix,
                                                      int round, CT* __restrict__ counts) {
  const int q = blockIdx.x, v = threadIdx.x;
  const int shift = 64 - SEL_BITS * (round + 1);
  const unsigned long long base = prefix[q] + ((unsigned long long)v << shift);
  // one search per digit boundary: the upper edge of digit v is the lower edge of digit v + 1
  __shared__ long long edge[SEL_DIGITS + 1];
  edge[v] = lower_bound_key(S, n, base);
  if (v == SEL_DIGITS - 1) {
    const unsigned long long upper = base + (1ull << shift);
    edge[SEL_DIGITS] = upper < base ? n : lower_bound_key(S, n, upper);   // wrapped past 2^64: everything from there on
  }
  __syncthreads();
  counts[(size_t)q * SEL_DIGITS + v] = (CT)(edge[v + 1] - edge[v]);
}
template <typename CT>
__global__ void __launch_bounds__(256) k_select_update(const CT* __restrict__ counts, int round, long long* __restrict__ remaining,
                                                       unsigned long long* __restrict__ prefix) {
  const int q = blockIdx.x;
  __shared__ long long cum[SEL_DIGITS];
  const long long mine = (long long)counts[(size_t)q * SEL_DIGITS + threadIdx.x];
  cum[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 1; o < SEL_DIGITS; o <<= 1) {
    const long long t = threadIdx.x >= o ? cum[threadIdx.x - o] : 0;
    __syncthreads();
    cum[threadIdx.x] += t;
    __syncthreads();
  }
  const long long before = cum[threadIdx.x] - mine, rem = remaining[q];
  const bool owner = rem >= before && rem < cum[threadIdx.x];
  const bool overflow = threadIdx.x == SEL_DIGITS - 1 && rem >= cum[SEL_DIGITS - 1];   // cannot happen with consistent counts
  __syncthreads();
  if (owner || overflow) {
    const int shift = 64 - SEL_BITS * (round + 1);
    prefix[q] += (unsigned long long)threadIdx.x << shift;
    remaining[q] = overflow ? 0 : rem - before;
  }
}
__device__ __forceinline__ double trim_lerp(const long long* __restrict__ n_global_dev, int bins, int i, const unsigned long long* __restrict__ keys) {
  const long long n = n_global_dev[0];
  const double step = bins > 1 ? 99.0 / (double)(bins - 1) : 0.0;
  const double p = (i == bins - 1 && bins > 1) ? 99.0 : __dmul_rn((double)i, step);
  const double vi = __dmul_rn((double)(n - 1), p / 100.0);
  const double a = __longlong_as_double((long long)keys[2 * i]), b = __longlong_as_double((long long)keys[2 * i + 1]);
  if (vi >= (double)(n - 1)) return b;
  if (vi < 0.0) return a;
  const double g = vi - floor(vi), diff = b - a;   // numpy _lerp, as trim_threshold_of
  return (g >= 0.5) ? __dadd_rn(b, -__dmul_rn(diff, __dadd_rn(1.0, -g))) : __dadd_rn(a, __dmul_rn(diff, g));
}
// part[3 i ..] = local (count, sum w, sum w^2) of the weights >= threshold_i ; part[3 bins ..] = local (sum w, sum w^2)
__global__ void __launch_bounds__(256) k_trim_partials(const double* __restrict__ S, const double* __restrict__ P1,
                                                       const double* __restrict__ P2, int64_t n, const long long* __restrict__ n_global_dev,
                                                       int bins, const unsigned long long* __restrict__ keys, double* __restrict__ part) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double tot1 = P1[n - 1], tot2 = P2[n - 1];
  if (i == 0) { part[3 * bins] = tot1; part[3 * bins + 1] = tot2; }
  if (i >= bins) return;
  const double thr = trim_lerp(n_global_dev, bins, i, keys);
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (S[mid] < thr) lo = mid + 1; else hi = mid; }
  part[3 * i] = (double)(n - lo);
  part[3 * i + 1] = tot1 - (lo > 0 ? P1[lo - 1] : 0.0);
  part[3 * i + 2] = tot2 - (lo > 0 ? P2[lo - 1] : 0.0);
}
__global__ void __launch_bounds__(256) k_trim_decide(const double* __restrict__ part, const long long* __restrict__ n_global_dev, double ess,
                                                     int bins, const unsigned long long* __restrict__ keys, double* __restrict__ out) {
  __shared__ int best;
  if (threadIdx.x == 0) best = 0;
  __syncthreads();
  const double tot1 = part[3 * bins], tot2 = part[3 * bins + 1];
  const double ess_total = (tot1 * tot1) / tot2;
  for (int i = threadIdx.x; i < bins; i += blockDim.x) {
    const double k1 = part[3 * i + 1], k2 = part[3 * i + 2];
    if (((k1 * k1) / k2) / ess_total >= ess) atomicMax(&best, i);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = trim_lerp(n_global_dev, bins, best, keys);
    out[1] = part[3 * best + 1];
    out[2] = part[3 * best];
    out[3] = ess_total;
  }
}
__global__ void k_set_i64(long long* p, long long v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }

extern "C" int tph_trim_threshold_global(tph_ctx* ctx, const double* w_dev, int64_t n, double ess, int bins, double* out_dev,
                                         double* out_host) {
  TPH_REQUIRE(ctx && w_dev && out_dev && n > 0 && bins >= 1, "tph_trim_threshold_global: bad argument");
  if (!ctx->comm_active()) return tph_trim_threshold(ctx, w_dev, n, ess, bins, out_dev, out_host);
  const int Q = 2 * bins;
  // staging: [0] n_global (i64, padded to 256 B) | counts [Q][256] i64  (later re-used for the [3 bins + 2] partial sums)
  const size_t o_cnt = 256, counts_bytes = sizeof(long long) * (size_t)Q * SEL_DIGITS;
  if (tph_comm_require(ctx, o_cnt + counts_bytes, "tph_trim_threshold_global")) return -2;
  size_t temp_bytes = 0;
  double* nullk = nullptr;
  TPH_HIP(rocprim::radix_sort_keys(nullptr, temp_bytes, w_dev, nullk, (size_t)n, 0, 64, ctx->stream));
  int64_t ntiles = tph_scan::num_tiles(n);
  size_t a_tiles = ((size_t)ntiles * sizeof(double) + 255) / 256 * 256;
  size_t a_n = ((size_t)n * sizeof(double) + 255) / 256 * 256;
  size_t a_tmp = (temp_bytes + 255) / 256 * 256;
  size_t a_q = ((size_t)Q * sizeof(long long) + 255) / 256 * 256;
  if (tph_scratch_reserve(ctx, a_tiles + 3 * a_n + a_tmp + 2 * a_q)) return -1;
  char* base = (char*)ctx->scratch;
  double* tiles = (double*)base;
  double* S = (double*)(base + a_tiles);
  double* P1 = (double*)(base + a_tiles + a_n);
  double* P2 = (double*)(base + a_tiles + 2 * a_n);
  void* tmp = base + a_tiles + 3 * a_n;
  long long* remaining = (long long*)(base + a_tiles + 3 * a_n + a_tmp);
  unsigned long long* prefix = (unsigned long long*)(base + a_tiles + 3 * a_n + a_tmp + a_q);
  TPH_HIP(rocprim::radix_sort_keys(tmp, temp_bytes, w_dev, S, (size_t)n, 0, 64, ctx->stream));
  if (tph_scan::inclusive<tph_scan::PLAIN>(ctx, S, n, nullptr, tiles, P1)) return -1;
  if (tph_scan::inclusive<tph_scan::SQUARE>(ctx, S, n, nullptr, tiles, P2)) return -1;
  long long* n_global = (long long*)ctx->comm_buf;
  long long* counts = (long long*)(ctx->comm_buf + o_cnt);
  hipLaunchKernelGGL(k_set_i64, dim3(1), dim3(1), 0, ctx->stream, n_global, (long long)n);
  TPH_LAUNCH_CHECK();
  if (tph_comm_allreduce(ctx, 0, 1, TPH_DT_I64, TPH_OP_SUM)) return -2;
  hipLaunchKernelGGL(k_trim_targets, dim3((bins + 255) / 256), dim3(256), 0, ctx->stream, n_global, bins, remaining, prefix);
  // 32-bit counters (half the all-reduce: 8 x [2 bins][256]) whenever the GLOBAL history is known to hold < 2^31 rows
  long long n_all = 0;
  for (int64_t v : ctx->n_global_t) n_all += v;
  const bool narrow = n == ctx->size && n_all > 0 && n_all < (1ll << 31);
  for (int round = 0; round < SEL_ROUNDS; ++round) {
    if (narrow) {
      hipLaunchKernelGGL(k_select_count<int>, dim3(Q), dim3(SEL_DIGITS), 0, ctx->stream, S, n, prefix, round, (int*)counts);
      TPH_LAUNCH_CHECK();
      if (tph_comm_allreduce(ctx, o_cnt, (int64_t)Q * SEL_DIGITS, TPH_DT_I32, TPH_OP_SUM)) return -2;
      hipLaunchKernelGGL(k_select_update<int>, dim3(Q), dim3(SEL_DIGITS), 0, ctx->stream, (const int*)counts, round, remaining, prefix);
    } else {
      hipLaunchKernelGGL(k_select_count<long long>, dim3(Q), dim3(SEL_DIGITS), 0, ctx->stream, S, n, prefix, round, counts);
      TPH_LAUNCH_CHECK();
      if (tph_comm_allreduce(ctx, o_cnt, (int64_t)Q * SEL_DIGITS, TPH_DT_I64, TPH_OP_SUM)) return -2;
      hipLaunchKernelGGL(k_select_update<long long>, dim3(Q), dim3(SEL_DIGITS), 0, ctx->stream, counts, round, remaining, prefix);
    }
  }
  double* part = (double*)(ctx->comm_buf + o_cnt);
  hipLaunchKernelGGL(k_trim_partials, dim3((bins + 255) / 256), dim3(256), 0, ctx->stream, S, P1, P2, n, n_global, bins, prefix, part);
  TPH_LAUNCH_CHECK();
  if (tph_comm_allreduce(ctx, o_cnt, 3 * (int64_t)bins + 2, TPH_DT_F64, TPH_OP_SUM)) return -2;
  hipLaunchKernelGGL(k_trim_decide, dim3(1), dim3(256), 0, ctx->stream, part, n_global, ess, bins, prefix, out_dev);
  TPH_LAUNCH_CHECK();
  if (out_host) {
    TPH_HIP(hipMemcpyAsync(ctx->pinned, out_dev, sizeof(double) * 4, hipMemcpyDeviceToHost, ctx->stream));
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) out_host[i] = ctx->pinned[i];
  }
  return 0;
}

// ---- segments: the stretches of a working set that belong to the virtual shards of the canonical partition (common.h) ----------
// seg[2 z], seg[2 z + 1] = first row and number of rows of segment z.  A kernel launched with gridDim.z segments works on each
// as if it were the whole input: same row partition over gridDim.x blocks, block partials of segment z behind those of z - 1.
#define SEG_SHIFT(hu, wt, labels, n, seg)                               \
  do {                                                                  \
    if (seg) {                                                          \
      const long long off_ = seg[2 * blockIdx.z];                       \
      n = seg[2 * blockIdx.z + 1];                                      \
      hu += off_;                                                       \
      wt += off_;                                                       \
      if (labels) labels += off_;                                       \
    }                                                                   \
  } while (0)
// out[c] = ((rows[0][c] + rows[1][c]) + rows[2][c]) + ... : the V per-shard results in shard order (deterministic, and the same
// tree for every number of ranks that divides V)
__global__ void __launch_bounds__(256) k_fold_cols(const double* __restrict__ rows, int V, int ncol, double* __restrict__ out) {
  rows += (size_t)blockIdx.y * V * ncol;      // (blockIdx.y: group -- its V rows, its output row: the pieces of one virtual shard)
  out += (size_t)blockIdx.y * ncol;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncol; c += gridDim.x * blockDim.x) {
    double s = rows[c];
    for (int v = 1; v < V; ++v) s += rows[(size_t)v * ncol + c];
    out[c] = s;
  }
}
// (min, max) per coordinate over the segments' ranges (order-free)
__global__ void __launch_bounds__(256) k_fold_range(const double* __restrict__ vr, int V, int d, double* __restrict__ range) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d) return;
  double mn = DBL_MAX, mx = -DBL_MAX;
  for (int v = 0; v < V; ++v) { mn = fmin(mn, vr[(size_t)v * 2 * d + 2 * j]); mx = fmax(mx, vr[(size_t)v * 2 * d + 2 * j + 1]); }
  range[2 * j] = mn; range[2 * j + 1] = mx;
}

// the pieces of the canonical partition as segments, shard-major: segment v * T + t = rows [t * n_loc + v * nv, + nv)
__global__ void k_piece_segments(int T, int vl, long long n_loc, long long nv, long long* __restrict__ seg) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= T * vl) return;
  const int v = p / T, t = p - v * T;
  seg[2 * p] = (long long)t * n_loc + (long long)v * nv;
  seg[2 * p + 1] = nv;
}

// -------------------------------------------------------------------------- weighted first moments
// sums[0] = sum wt ; sums[1+j] = sum wt * u_j ; range[2j], range[2j+1] = min, max of u_j over rows with wt > 0
// (wt = counts or real weights, optional label filter)
template <typename WT>
__global__ void __launch_bounds__(256) k_wsum(const double* __restrict__ hu, int64_t cap, int d, const WT* __restrict__ wt,
                                              const int32_t* __restrict__ labels, int label, int64_t n,
                                              double* __restrict__ partials, const long long* __restrict__ seg = nullptr) {
  // grid: (row blocks, 1 + d [, segments]): blockIdx.y == 0 -> sum of weights, else coordinate blockIdx.y-1.
  // seg != NULL: blockIdx.z picks the segment [seg[2z], seg[2z] + seg[2z+1]) of the rows (a virtual shard's stretch of the
  // working set, see tph_fit_modes): the block partials of segment z depend on that segment's rows and on gridDim.x alone
  SEG_SHIFT(hu, wt, labels, n, seg);
  const int col = blockIdx.y;
  const double* src = col ? hu + (size_t)(col - 1) * cap : nullptr;
  double s = 0.0, mn = DBL_MAX, mx = -DBL_MAX;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // four rows in flight per lane (the loads of a trip are issued before the first use); accumulated in row order, as one by one
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
    double w[4], v[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      ok[k] = i < n && (!labels || labels[i] == label);
      w[k] = ok[k] ? (double)wt[i] : 0.0;
      v[k] = (ok[k] && col) ? src[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!ok[k]) continue;
      if (col) {
        s += w[k] * v[k];
        if (w[k] > 0.0) { mn = fmin(mn, v[k]); mx = fmax(mx, v[k]); }
      } else {
        s += w[k];
      }
    }
  }
  __shared__ double sh[4];
  s = tph_block_sum(s, sh);
  mx = tph_block_max(mx, sh);
  mn = -tph_block_max(-mn, sh);
  if (threadIdx.x == 0) {
    double* p = partials + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + col) * 3;
    p[0] = s; p[1] = mn; p[2] = mx;
  }
}

// reduce the (sum, min, max) block partials: one block per column
__global__ void __launch_bounds__(256) k_wsum_final(const double* __restrict__ partials, int nblocks, int ncol,
                                                    double* __restrict__ sums, double* __restrict__ range) {
  int c = blockIdx.x;
  // (blockIdx.y: segment -- its partials, its sums[ncol] and range[2 (ncol - 1)])
  partials += (size_t)blockIdx.y * nblocks * ncol * 3;
  sums += (size_t)blockIdx.y * ncol;
  if (range) range += (size_t)blockIdx.y * 2 * (ncol - 1);
  double s = 0.0, mn = DBL_MAX, mx = -DBL_MAX;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    const double* p = partials + ((size_t)b * ncol + c) * 3;
    s += p[0]; mn = fmin(mn, p[1]); mx = fmax(mx, p[2]);
  }
  __shared__ double sh[4];
  s = tph_block_sum(s, sh);
  mx = tph_block_max(mx, sh);
  mn = -tph_block_max(-mn, sh);
  if (threadIdx.x == 0) {
    sums[c] = s;
    if (c > 0 && range) { range[2 * (c - 1)] = mn; range[2 * (c - 1) + 1] = mx; }
  }
}

// ... with many columns (ncol > 256: the triangle of a covariance at n_dim >= 23): a thread per COLUMN, the rows in order through four
// interleaved accumulators -- neighbouring lanes read neighbouring columns of a row (coalesced), where a block per column reads one
// 8-byte word per 64-byte sector and, since the per-shard segments of round 5, runs 16 x ncol blocks of 128 rows each
// (50-D: 17 -> 111 us, 100-D: 71 -> 316 us per launch before this kernel).  Deterministic: a fixed order per column.
__global__ void __launch_bounds__(256) k_colsum2_wide(const double* __restrict__ partials, int nblocks, int ncol,
                                                      double* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncol) return;
  partials += (size_t)blockIdx.y * nblocks * ncol;      // (blockIdx.y: segment)
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  int b = 0;
  for (; b + 4 <= nblocks; b += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += partials[(size_t)(b + k) * ncol + c];
  }
  for (; b < nblocks; ++b) a[b & 3] += partials[(size_t)b * ncol + c];
  out[(size_t)blockIdx.y * ncol + c] = (a[0] + a[1]) + (a[2] + a[3]);
}
static void launch_colsum2(tph_ctx* ctx, const double* partials, int nblocks, int ncol, int nseg, double* out);
__global__ void __launch_bounds__(256) k_colsum2(const double* __restrict__ partials, int nblocks, int ncol,
                                                 double* __restrict__ out) {
  int c = blockIdx.x;
  partials += (size_t)blockIdx.y * nblocks * ncol;      // (blockIdx.y: segment)
  out += (size_t)blockIdx.y * ncol;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[(size_t)b * ncol + c];
  __shared__ double sh[4];
  s = tph_block_sum(s, sh);
  if (threadIdx.x == 0) out[c] = s;
}

static void launch_colsum2(tph_ctx* ctx, const double* partials, int nblocks, int ncol, int nseg, double* out) {
  // (few rows per segment and many columns: the per-shard partials of a fit at n_dim >= 23; one segment of 2048 rows keeps the block per column)
  if (ncol > 256 && nblocks <= 256) hipLaunchKernelGGL(k_colsum2_wide, dim3((ncol + 255) / 256, nseg), dim3(256), 0, ctx->stream, partials, nblocks, ncol, out);
  else hipLaunchKernelGGL(k_colsum2, dim3(ncol, nseg), dim3(256), 0, ctx->stream, partials, nblocks, ncol, out);
}

// mean_j = sums[1+j] / sums[0]
__global__ void k_mean_from_sums(const double* __restrict__ sums, int d, double* __restrict__ mean) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d) mean[j] = sums[1 + j] / sums[0];
}

// --------------------------------------------------------------- centred weighted second moments
// C[a][b] = sum_s wt_s (u_a - m_a)(u_b - m_b), lower triangle only (npl = d(d+1)/2 pairs).  A tile of 64 rows is
// staged in LDS ([d][65] padded + weights); thread t owns row slice t % S of the tile and the pairs
// t / S + k * (256 / S): for small d (few pairs) the 64 rows are split over S lanes per pair so that all 256
// threads work.  Block partials [blocks * S][npl] are reduced in fixed order by k_colsum2 (deterministic).
constexpr int COV_ROWS = 64;
constexpr int COV_LD = 65;   // padded row: conflict-free ds_read_b64 across pairs
constexpr int COV_NPT = 20;  // pairs per thread: 20 * 256 >= 5050 = npl(d = 100)

__host__ __device__ inline int cov_slices(int npl) {
  int S = 1;
  while (S < 16 && npl * (S * 2) <= 256) S *= 2;
  return S;
}

template <typename WT>
__global__ void __launch_bounds__(256) k_wcov(const double* __restrict__ hu, int64_t cap, int d, const WT* __restrict__ wt,
                                              const int32_t* __restrict__ labels, int label, int64_t n,
                                              const double* __restrict__ mean, double* __restrict__ partials,
                                              const long long* __restrict__ seg = nullptr) {
  SEG_SHIFT(hu, wt, labels, n, seg);
  extern __shared__ double sh[];
  double* xs = sh;                               // [d][65]
  double* ws = sh + (size_t)d * COV_LD;          // [64]
  const int npl = d * (d + 1) / 2;
  const int S = cov_slices(npl);
  const int slots = 256 / S;
  const int slice = threadIdx.x % S, pslot = threadIdx.x / S;
  const int rows_per = COV_ROWS / S, r_lo = slice * rows_per;
  int pa[COV_NPT], pb[COV_NPT];
  double acc[COV_NPT];
#pragma unroll
  for (int k = 0; k < COV_NPT; ++k) {
    acc[k] = 0.0;
    int p = pslot + k * slots;
    int a = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    while (a * (a + 1) / 2 > p) --a;
    pa[k] = a;
    pb[k] = p - a * (a + 1) / 2;
  }
  const int64_t ntiles = (n + COV_ROWS - 1) / COV_ROWS;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t r0 = t * COV_ROWS;
    __syncthreads();
    for (int e = threadIdx.x; e < d * COV_ROWS; e += blockDim.x) {
      int j = e / COV_ROWS, r = e % COV_ROWS;
      int64_t i = r0 + r;
      xs[j * COV_LD + r] = i < n ? hu[(size_t)j * cap + i] - mean[j] : 0.0;
    }
    if (threadIdx.x < COV_ROWS) {
      int64_t i = r0 + threadIdx.x;
      double w = 0.0;
      if (i < n && (!labels || labels[i] == label)) w = (double)wt[i];
      ws[threadIdx.x] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < COV_NPT; ++k) {
      if (pslot + k * slots < npl) {
        const double* xa = xs + pa[k] * COV_LD + r_lo;
        const double* xb = xs + pb[k] * COV_LD + r_lo;
        const double* wr = ws + r_lo;
        double s = 0.0;
        for (int r = 0; r < rows_per; ++r) s += wr[r] * xa[r] * xb[r];
        acc[k] += s;
      }
    }
  }
  double* mine = partials + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * S + slice) * npl;
#pragma unroll
  for (int k = 0; k < COV_NPT; ++k) {
    int p = pslot + k * slots;
    if (p < npl) mine[p] = acc[k];
  }
}

// n_dim >= 16: the same staged tile, but a thread owns a 4 x 4 BLOCK of the lower triangle over a slice of the tile's rows:
// per row 1 + 4 + 4 LDS reads feed 16 FMAs (the pair-per-thread form above reads w, x_a, x_b for every single FMA and is
// LDS-bound: 0.5 TFLOP/s at d = 32).  Items = (block pair, row slice); slices exist only while there are fewer block pairs
// than threads (d <= 64) and meet in a fixed order through an LDS copy of the triangle (no float atomics).
constexpr int COV_TB = 4;
__host__ __device__ inline int cov_tile_slices(int d) {
  const int nbk = (d + COV_TB - 1) / COV_TB, pairs = nbk * (nbk + 1) / 2;
  int sl = 1;
  while (sl < 16 && pairs * sl * 2 <= 256) sl *= 2;
  return sl;
}
// (NV: see the tile's fill below.)
template <typename WT, int NV>
__global__ void __launch_bounds__(256) k_wcov_tiled(const double* __restrict__ hu, int64_t cap, int d, const WT* __restrict__ wt,
                                                    const int32_t* __restrict__ labels, int label, int64_t n,
                                                    const double* __restrict__ mean, double* __restrict__ partials,
                                                    const long long* __restrict__ seg = nullptr) {
  SEG_SHIFT(hu, wt, labels, n, seg);
  extern __shared__ double sh[];
  double* xs = sh;                               // [d][65]
  double* ws = sh + (size_t)d * COV_LD;          // [64]
  double* tri = ws + COV_ROWS;                   // [npl], only when the tile's rows are sliced
  const int npl = d * (d + 1) / 2;
  const int nbk = (d + COV_TB - 1) / COV_TB, pairs = nbk * (nbk + 1) / 2;
  const int SL = cov_tile_slices(d), rows_per = COV_ROWS / SL;
  const int items = pairs * SL;
  constexpr int MAXI = 2;                         // items per thread: 325 block pairs at d = 100
  int ia[MAXI], ib[MAXI], isl[MAXI];
  double acc[MAXI][COV_TB][COV_TB];
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const int it = threadIdx.x + k * 256;
    const int p = it / SL;
    isl[k] = it - p * SL;
    int a = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    while (a * (a + 1) / 2 > p) --a;
    ia[k] = it < items ? a : -1;
    ib[k] = p - a * (a + 1) / 2;
#pragma unroll
    for (int q = 0; q < COV_TB; ++q)
#pragma unroll
      for (int r = 0; r < COV_TB; ++r) acc[k][q][r] = 0.0;
  }
  const int64_t ntiles = (n + COV_ROWS - 1) / COV_ROWS;
  // the tile's fill: wave w takes columns w, w + 4, ... (lane = row of the tile), eight of them requested before the first is
  // used.  NV = 8 (n_dim <= 32): all of a thread's columns, and those of the NEXT tile are requested before this tile's products
  // start (config 3: 132 -> 114 us per launch); NV = 0 (above 32-D): tile by tile -- carrying 16 or 32 columns in registers
  // across the products costs more occupancy than the overlap returns (measured: 50-D 388 -> 463 us, 100-D 1.6 -> 3.5 ms)
  const int fr = threadIdx.x & 63, jw = threadIdx.x >> 6;
  constexpr int NVR = NV > 0 ? NV : 1;
  double nx[NVR], nw = 0.0;
  auto request = [&](int64_t t) {
    const int64_t i = t * COV_ROWS + fr;
#pragma unroll
    for (int k = 0; k < NVR; ++k) {
      const int j = jw + 4 * k;
      nx[k] = (j < d && i < n) ? hu[(size_t)j * cap + i] : 0.0;
    }
    if (jw == 0) nw = (i < n && (!labels || labels[i] == label)) ? (double)wt[i] : 0.0;
  };
  if (NV > 0 && (int64_t)blockIdx.x < ntiles) request(blockIdx.x);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t r0 = t * COV_ROWS;
    __syncthreads();
    if (NV > 0) {
#pragma unroll
      for (int k = 0; k < NVR; ++k) {
        const int j = jw + 4 * k;
        if (j < d) xs[j * COV_LD + fr] = r0 + fr < n ? nx[k] - mean[j] : 0.0;
      }
      if (jw == 0) ws[fr] = nw;
    } else {
      const int64_t i = r0 + fr;
      for (int j0 = jw; j0 < d; j0 += 32) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = j0 + 4 * k;
          v[k] = (j < d && i < n) ? hu[(size_t)j * cap + i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = j0 + 4 * k;
          if (j < d) xs[j * COV_LD + fr] = i < n ? v[k] - mean[j] : 0.0;
        }
      }
      if (jw == 0) ws[fr] = (i < n && (!labels || labels[i] == label)) ? (double)wt[i] : 0.0;
    }
    __syncthreads();
    if (NV > 0 && t + gridDim.x < ntiles) request(t + gridDim.x);
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      if (ia[k] < 0) continue;
      // slice s takes rows s, s + SL, s + 2 SL ... of the tile: the slices of one block pair read neighbouring LDS words
      const int a0 = ia[k] * COV_TB, b0 = ib[k] * COV_TB, rlo = isl[k];
      // rows of the blocks beyond d (last block of an n_dim that is no multiple of 4) read row d-1: computed, never stored
      const double* xa[COV_TB];
      const double* xb[COV_TB];
#pragma unroll
      for (int q = 0; q < COV_TB; ++q) {
        xa[q] = xs + (size_t)(a0 + q < d ? a0 + q : d - 1) * COV_LD + rlo;
        xb[q] = xs + (size_t)(b0 + q < d ? b0 + q : d - 1) * COV_LD + rlo;
      }
      for (int r = 0; r < rows_per; ++r) {
        const double w = ws[rlo + r * SL];
        double va[COV_TB], vb[COV_TB];
#pragma unroll
        for (int q = 0; q < COV_TB; ++q) { va[q] = w * xa[q][r * SL]; vb[q] = xb[q][r * SL]; }
#pragma unroll
        for (int q = 0; q < COV_TB; ++q)
#pragma unroll
          for (int c = 0; c < COV_TB; ++c) acc[k][q][c] = fma(va[q], vb[c], acc[k][q][c]);
      }
    }
  }
  double* mine = partials + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * npl;
  if (SL == 1) {
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      if (ia[k] < 0) continue;
#pragma unroll
      for (int q = 0; q < COV_TB; ++q)
#pragma unroll
        for (int c = 0; c < COV_TB; ++c) {
          const int a = ia[k] * COV_TB + q, b = ib[k] * COV_TB + c;
          if (a < d && b <= a) mine[a * (a + 1) / 2 + b] = acc[k][q][c];
        }
    }
    return;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < npl; e += blockDim.x) tri[e] = 0.0;
  for (int s = 0; s < SL; ++s) {                  // slice by slice: every entry has one owner per slice -> fixed order, no atomics
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      if (ia[k] < 0 || isl[k] != s) continue;
#pragma unroll
      for (int q = 0; q < COV_TB; ++q)
#pragma unroll
        for (int c = 0; c < COV_TB; ++c) {
          const int a = ia[k] * COV_TB + q, b = ib[k] * COV_TB + c;
          if (a < d && b <= a) tri[a * (a + 1) / 2 + b] += acc[k][q][c];
        }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < npl; e += blockDim.x) mine[e] = tri[e];
}

// n_dim >= 16 on the matrix cores: C = X^T diag(w) X is a genuine FP64 SYRK, the one GEMM-shaped product of the path.
// v_mfma_f64_16x16x4: D(16 x 16) += A(16 x 4) B(4 x 16) per instruction, lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]
// and holds D[4 v + l / 16][l % 16], v = 0..3 (layout probed on gfx950, scratch test).  The staged tile of 64 rows is xs[dim][row];
// for the block pair (Ab >= Bb) of 16 x 16 blocks of the triangle, A = (w x)[Ab dims][4 rows], B = x[4 rows][Bb dims]: two LDS
// reads and one multiply per 1024 multiply-adds.  The 4 waves of a block share the block pairs.  MEASURED: no faster than the
// register-blocked VALU kernel above (74 vs 74 us at 262 144 x 32-D, 218 vs 224 us at 131 072 x 100-D, incl. the column sums):
// with either, the arithmetic hides behind the fill of the 64-row tile, which is what bounds both (0.9 TB/s: two barriers
// per 16 KB tile, no overlap of the next tile's loads).  Kept as TPH_OPT_COV_KERNEL = 2 and in the parity tests; the default is
// the VALU kernel.
typedef double tph_v4d __attribute__((ext_vector_type(4)));
constexpr int COV_MF_MAXP = 9;                  // block pairs per wave: 36 pairs (n_dim <= 128) over 4 waves
template <typename WT, int NV>
__global__ void __launch_bounds__(256) k_wcov_mfma(const double* __restrict__ hu, int64_t cap, int d, const WT* __restrict__ wt,
                                                   const int32_t* __restrict__ labels, int label, int64_t n,
                                                   const double* __restrict__ mean, double* __restrict__ partials,
                                                   const long long* __restrict__ seg = nullptr) {
  SEG_SHIFT(hu, wt, labels, n, seg);
  extern __shared__ double sh[];
  const int NB = (d + 15) / 16, dp = NB * 16, pairs = NB * (NB + 1) / 2;
  double* xs = sh;                               // [dp][65], dims >= d are zero
  double* ws = sh + (size_t)dp * COV_LD;         // [64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  int offa[COV_MF_MAXP], offb[COV_MF_MAXP];      // LDS offsets of this lane's A / B element at row 0
  tph_v4d acc[COV_MF_MAXP];
#pragma unroll
  for (int k = 0; k < COV_MF_MAXP; ++k) {
    const int p = wave + 4 * k;
    int a = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    while (a * (a + 1) / 2 > p) --a;
    const int b = p - a * (a + 1) / 2;
    offa[k] = p < pairs ? (a * 16 + li) * COV_LD + lk : -1;
    offb[k] = (b * 16 + li) * COV_LD + lk;
    acc[k] = tph_v4d{0.0, 0.0, 0.0, 0.0};
  }
  const int64_t ntiles = (n + COV_ROWS - 1) / COV_ROWS;
  // the tile's fill as in k_wcov_tiled: NV = 8 (n_dim <= 32) carries the next tile's columns in registers, NV = 0 fills tile by tile
  const int fr = threadIdx.x & 63, jw = threadIdx.x >> 6;
  constexpr int NVR = NV > 0 ? NV : 1;
  double nx[NVR], nw = 0.0;
  auto request = [&](int64_t t) {
    const int64_t i = t * COV_ROWS + fr;
#pragma unroll
    for (int k = 0; k < NVR; ++k) {
      const int j = jw + 4 * k;
      nx[k] = (j < d && i < n) ? hu[(size_t)j * cap + i] : 0.0;
    }
    if (jw == 0) nw = (i < n && (!labels || labels[i] == label)) ? (double)wt[i] : 0.0;
  };
  if (NV > 0 && (int64_t)blockIdx.x < ntiles) request(blockIdx.x);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t r0 = t * COV_ROWS;
    __syncthreads();
    if (NV > 0) {
#pragma unroll
      for (int k = 0; k < NVR; ++k) {
        const int j = jw + 4 * k;
        if (j < dp) xs[j * COV_LD + fr] = (j < d && r0 + fr < n) ? nx[k] - mean[j] : 0.0;
      }
      if (jw == 0) ws[fr] = nw;
    } else {
      const int64_t i = r0 + fr;
      for (int j0 = jw; j0 < dp; j0 += 32) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = j0 + 4 * k;
          v[k] = (j < d && i < n) ? hu[(size_t)j * cap + i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = j0 + 4 * k;
          if (j < dp) xs[j * COV_LD + fr] = (j < d && i < n) ? v[k] - mean[j] : 0.0;
        }
      }
      if (jw == 0) ws[fr] = (i < n && (!labels || labels[i] == label)) ? (double)wt[i] : 0.0;
    }
    __syncthreads();
    if (NV > 0 && t + gridDim.x < ntiles) request(t + gridDim.x);
    for (int r = 0; r < COV_ROWS; r += 4) {
      const double w = ws[r + lk];
#pragma unroll
      for (int k = 0; k < COV_MF_MAXP; ++k) {
        if (offa[k] >= 0) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(w * xs[offa[k] + r], xs[offb[k] + r], acc[k], 0, 0, 0);
      }
    }
  }
  const int npl = d * (d + 1) / 2;
  double* mine = partials + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * npl;
#pragma unroll
  for (int k = 0; k < COV_MF_MAXP; ++k) {
    if (offa[k] < 0) continue;
    const int a0 = (offa[k] - lk) / COV_LD - li, b0 = (offb[k] - lk) / COV_LD - li;     // first dims of the two blocks
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int a = a0 + 4 * v + lk, b = b0 + li;
      if (a < d && b <= a) mine[a * (a + 1) / 2 + b] = acc[k][v];
    }
  }
}

// launches the second-moment kernel for n_dim > 12 (block partials [nblk][npl] when tiled, [nblk * S][npl] otherwise):
// returns the number of partial rows per block
template <typename WT>
static int launch_wcov(tph_ctx* ctx, const double* src, int64_t src_ld, const WT* wt, const int32_t* labels, int label, int64_t n,
                       const double* mean, double* partials, int nblk, int* rows_per_block, const long long* seg = nullptr, int nseg = 1) {
  const int d = ctx->d, npl = d * (d + 1) / 2;
  if (d >= 16 && d <= 128 && ctx->cov_kernel == 2) {      // matrix cores on request (TPH_OPT_COV_KERNEL: 0 auto = 1 register blocks | 2 MFMA)
    const size_t lds = sizeof(double) * ((size_t)((d + 15) / 16 * 16) * COV_LD + COV_ROWS);
#define TPH_WCOV_MFMA(NV_)                                                                                                    \
    do {                                                                                                                      \
      if (lds > 64 * 1024)                                                                                                    \
        TPH_HIP(hipFuncSetAttribute((const void*)k_wcov_mfma<WT, NV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((k_wcov_mfma<WT, NV_>), dim3(nblk, 1, nseg), dim3(256), lds, ctx->stream, src, src_ld, d, wt, labels, label, n, mean, partials, seg); \
    } while (0)
    if (d <= 32) TPH_WCOV_MFMA(8); else TPH_WCOV_MFMA(0);
#undef TPH_WCOV_MFMA
    *rows_per_block = 1;
    return 0;
  }
  if (d >= 16) {
    const int SL = cov_tile_slices(d);
    const size_t lds = sizeof(double) * ((size_t)d * COV_LD + COV_ROWS + (SL > 1 ? npl : 0));
#define TPH_WCOV_TILED(NV_)                                                                                                   \
    do {                                                                                                                      \
      if (lds > 64 * 1024)                                                                                                    \
        TPH_HIP(hipFuncSetAttribute((const void*)k_wcov_tiled<WT, NV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((k_wcov_tiled<WT, NV_>), dim3(nblk, 1, nseg), dim3(256), lds, ctx->stream, src, src_ld, d, wt, labels, label, n, mean, partials, seg); \
    } while (0)
    if (d <= 32) TPH_WCOV_TILED(8); else TPH_WCOV_TILED(0);
#undef TPH_WCOV_TILED
    *rows_per_block = 1;
    return 0;
  }
  const size_t lds = sizeof(double) * ((size_t)d * COV_LD + COV_ROWS);
  if (lds > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_wcov<WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_wcov<WT>, dim3(nblk, 1, nseg), dim3(256), lds, ctx->stream, src, src_ld, d, wt, labels, label, n, mean, partials, seg);
  *rows_per_block = cov_slices(npl);
  return 0;
}

// Small n_dim (<= 12): one lane per row, the d(d+1)/2 accumulators live in registers, the stream over u is
// coalesced per coordinate and nothing is staged: HBM-bound.  One block partial of npl sums per block.
template <typename WT, int D>
__global__ void __launch_bounds__(256) k_wcov_small(const double* __restrict__ hu, int64_t cap, const WT* __restrict__ wt,
                                                    const int32_t* __restrict__ labels, int label, int64_t n,
                                                    const double* __restrict__ mean, double* __restrict__ partials,
                                                    const long long* __restrict__ seg = nullptr) {
  SEG_SHIFT(hu, wt, labels, n, seg);
  constexpr int NPL = D * (D + 1) / 2;
  double acc[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) acc[k] = 0.0;
  double m[D];
#pragma unroll
  for (int j = 0; j < D; ++j) m[j] = mean[j];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // two rows in flight per lane (2 D loads issued before the first use); accumulated in row order, as one by one
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 2 * stride) {
    double w[2], xc[2][D];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t i = i0 + r * stride;
      const bool in = i < n;
      w[r] = (in && (!labels || labels[i] == label)) ? (double)wt[i] : 0.0;
#pragma unroll
      for (int j = 0; j < D; ++j) xc[r][j] = (in ? hu[(size_t)j * cap + i] : m[j]) - m[j];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r == 1 && i0 + stride >= n) break;
      int k = 0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double wa = w[r] * xc[r][a];
#pragma unroll
        for (int b = 0; b <= a; ++b) acc[k++] += wa * xc[r][b];
      }
    }
  }
  __shared__ double sh[4 * NPL];
  tph_block_sum_many<NPL>(acc, sh, partials + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * NPL);
}

// grid of the register kernel: at 168 VGPRs three of its blocks fit a CU, and a grid past the 768 resident ones only adds
// rounds of (start-up latency + 55 block reductions) -- measured on a 3.3 M-row working set: 2048 blocks 143 us
static inline int wcov_small_blocks(int nblk) { return nblk < 768 ? nblk : 768; }

template <typename WT>
static bool launch_wcov_small(tph_ctx* ctx, const double* src, int64_t src_ld, const WT* wt, const int32_t* labels, int label,
                              int64_t n, const double* mean, double* partials, int nblk, const long long* seg = nullptr, int nseg = 1) {
  switch (ctx->d) {
#define C(DD) case DD: hipLaunchKernelGGL((k_wcov_small<WT, DD>), dim3(nblk, 1, nseg), dim3(256), 0, ctx->stream, src, src_ld, wt, labels, label, n, mean, partials, seg); return true;
    C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
    default: return false;
  }
}

// Weighted mean AND covariance in ONE pass over u (n_dim <= 12): sums of w, w (u - c) and w (u - c)(u - c)^T about a
// caller-supplied centre c close to the mean (the previous iteration's mean), finished as
//     mean = c + S1/S0 ,  cov = S2/S0 - (S1/S0)(S1/S0)^T .
// With |mean - c| a fraction of the spread the subtraction loses nothing (relative error ~eps (mean-c)^2/var), and the
// volume-variation diagnostic of tools.py:94-99 needs two streaming passes over the history instead of three.
template <int D>
__global__ void __launch_bounds__(256) k_wmom_small(const double* __restrict__ hu, int64_t cap, const double* __restrict__ wt,
                                                    int64_t n, const double* __restrict__ centre, double* __restrict__ partials,
                                                    const long long* __restrict__ seg = nullptr) {
  const int32_t* nolab_ = nullptr;
  SEG_SHIFT(hu, wt, nolab_, n, seg);
  constexpr int NPL = D * (D + 1) / 2, NC = NPL + D + 1;
  double acc[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) acc[k] = 0.0;
  double c[D];
#pragma unroll
  for (int j = 0; j < D; ++j) c[j] = centre[j];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double w = wt[i];
    // (no test for weightless rows here, unlike k_cv_sum_small: a branch in front of the coordinate loads costs this kernel
    // 10 % on a history without such rows -- the loads of consecutive trips no longer overlap -- and it runs mostly on those)
    double xc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) xc[j] = hu[(size_t)j * cap + i] - c[j];
    int k = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const double wa = w * xc[a];
      acc[NPL + 1 + a] += wa;
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[k++] += wa * xc[b];
    }
    acc[NPL] += w;
  }
  __shared__ double sh[4 * NC];
  tph_block_sum_many<NC>(acc, sh, partials + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * NC);
}
// out = (S0, mean[d], cov[d][d]) from the column sums csum = (S2 lower triangle, S0, S1[d])
__global__ void k_wmom_finish(const double* __restrict__ csum, const double* __restrict__ centre, int d, double* __restrict__ out) {
  const int npl = d * (d + 1) / 2;
  const double s0 = csum[npl];
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) out[0] = s0;
  if (e < d) out[1 + e] = centre[e] + csum[npl + 1 + e] / s0;
  if (e < d * d) {
    int a = e / d, b = e % d;
    int hi = a >= b ? a : b, lo = a >= b ? b : a;
    double da = csum[npl + 1 + a] / s0, db = csum[npl + 1 + b] / s0;
    out[1 + d + e] = csum[hi * (hi + 1) / 2 + lo] / s0 - da * db;
  }
}

// symmetrise the lower triangle and (optionally) apply student.py:62-63:
//   Sigma = C/n + diag(C/n)/n      (np.cov*(n-1)/n + diag(np.var)/n)
__global__ void k_cov_finish(const double* __restrict__ csum, const double* __restrict__ sums, int d, int student,
                             double* __restrict__ cov) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d * d) return;
  int a = e / d, b = e % d;
  int hi = a >= b ? a : b, lo = a >= b ? b : a;
  double v = csum[hi * (hi + 1) / 2 + lo];
  if (student == 1) {
    double ntot = sums[0];
    v = v / ntot;
    if (a == b) v = v + v / ntot;
  } else if (student == 2) {
    v = v / sums[0];   // tools.py:94: weights normalised to sum 1
  }
  cov[e] = v;
}

static int moments_launch_cov(tph_ctx* ctx, const void* wt, bool wt_is_int, const int32_t* labels, int label, int64_t n,
                              const double* mean_dev, const double* sums_dev, int student, double* cov_dev,
                              double* partials, int nblk, const double* xsrc = nullptr, int64_t xld = 0) {
  const int d = ctx->d;
  const double* src = xsrc ? xsrc : ctx->u;
  const int64_t src_ld = xsrc ? xld : ctx->cap;
  const int npl = d * (d + 1) / 2;
  const int S = cov_slices(npl);
  TPH_REQUIRE(npl <= COV_NPT * 256, "covariance kernel supports n_dim <= 100 (got %d)", d);
  if (d <= 12) {
    // register kernel: block partials [nblk][npl] (no row slices)
    const int nb = wcov_small_blocks(nblk);
    bool ok = wt_is_int ? launch_wcov_small<int32_t>(ctx, src, src_ld, (const int32_t*)wt, labels, label, n, mean_dev, partials, nb)
                        : launch_wcov_small<double>(ctx, src, src_ld, (const double*)wt, labels, label, n, mean_dev, partials, nb);
    TPH_REQUIRE(ok, "covariance: no register kernel for n_dim=%d", d);
    double* csum1 = partials + (size_t)nblk * npl;
    launch_colsum2(ctx, partials, nb, npl, 1, csum1);
    hipLaunchKernelGGL(k_cov_finish, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, csum1, sums_dev, d, student, cov_dev);
    TPH_LAUNCH_CHECK();
    return 0;
  }
  int rpb = S;
  if (wt_is_int ? launch_wcov<int32_t>(ctx, src, src_ld, (const int32_t*)wt, labels, label, n, mean_dev, partials, nblk, &rpb)
                : launch_wcov<double>(ctx, src, src_ld, (const double*)wt, labels, label, n, mean_dev, partials, nblk, &rpb))
    return -1;
  double* csum = partials + (size_t)nblk * S * npl;
  launch_colsum2(ctx, partials, nblk * rpb, npl, 1, csum);
  hipLaunchKernelGGL(k_cov_finish, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, csum, sums_dev, d, student, cov_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

static int cov_blocks(int64_t n) {
  int64_t t = (n + COV_ROWS - 1) / COV_ROWS;
  return (int)(t < 2048 ? (t < 1 ? 1 : t) : 2048);
}
// doubles of scratch the covariance stage needs: block partials (<= nblk * 16 slices * npl <= nblk * 256 for small d,
// nblk * npl otherwise) + the reduced triangle
static size_t cov_scratch_doubles(int d, int nblk) {
  int npl = d * (d + 1) / 2;
  return (size_t)nblk * cov_slices(npl) * npl + (size_t)npl;
}

// -------------------------------------------------------------------- exact weighted median select
// Per (dimension j, target rank t in {lower, upper middle}): two 4096-bin histogram levels over the
// coordinate's data range (2^24 bins in all), then the <= MED_CAP entries of the final bin are
// collected and the order statistic is found exactly.  Multiplicities are the up-sampling counts.
constexpr int MED_BINS = 4096;
constexpr int MED_CAP = 2048;

// Monotone (not necessarily exact) two-level bin of u inside the data range [lo, hi] of its coordinate:
// selection only needs the same non-decreasing map in every pass.
__device__ __forceinline__ void med_digits(double u, double lo, double scale, int& d1, int& d2) {
  double a = (u - lo) * scale;
  double f1 = floor(a);
  if (f1 > 4095.0) f1 = 4095.0;
  if (!(f1 >= 0.0)) f1 = 0.0;
  double b = (a - f1) * 4096.0;
  double f2 = floor(b);
  if (f2 > 4095.0) f2 = 4095.0;
  if (!(f2 >= 0.0)) f2 = 0.0;
  d1 = (int)f1;
  d2 = (int)f2;
}
__device__ __forceinline__ double med_scale(double lo, double hi) {
  double w = hi - lo;
  return w > 0.0 ? 4096.0 / w : 0.0;
}

// level 1: hist1[j][bin] += count
__global__ void __launch_bounds__(256) k_med_hist1(const double* __restrict__ hu, int64_t cap, const int32_t* __restrict__ cnt,
                                                   const int32_t* __restrict__ labels, int label, int64_t n,
                                                   const double* __restrict__ range, unsigned int* __restrict__ hist1) {
  __shared__ unsigned int h[MED_BINS];
  const int j = blockIdx.y;
  const double lo = range[2 * j], scale = med_scale(range[2 * j], range[2 * j + 1]);
  for (int b = threadIdx.x; b < MED_BINS; b += blockDim.x) h[b] = 0;
  __syncthreads();
  const double* src = hu + (size_t)j * cap;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {     // four rows in flight per lane
    int c[4];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      c[k] = (i < n && (!labels || labels[i] == label)) ? cnt[i] : 0;
      v[k] = i < n ? src[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c[k] == 0) continue;
      int d1, d2;
      med_digits(v[k], lo, scale, d1, d2);
      atomicAdd(&h[d1], (unsigned int)c[k]);
    }
  }
  __syncthreads();
  unsigned int* g = hist1 + (size_t)j * MED_BINS;
  for (int b = threadIdx.x; b < MED_BINS; b += blockDim.x)
    if (h[b]) atomicAdd(&g[b], h[b]);
}

// pick the bin holding each target rank.  sel[j][t] = {bin1, bin2, rank_in_bin(after level), _}
// level==1 reads hist1[j][.] ; level==2 reads hist2[j][t][.].  256 threads x 16 bins + a block scan.
__global__ void __launch_bounds__(256) k_med_select(const unsigned int* __restrict__ hist, int level,
                                                   const double* __restrict__ sums, long long* __restrict__ sel) {
  const int j = blockIdx.x, t = blockIdx.y;
  long long* s = sel + ((size_t)j * 2 + t) * 4;
  long long rank;
  if (level == 1) {
    long long ntot = (long long)sums[0];
    rank = t == 0 ? (ntot - 1) / 2 : ntot / 2;   // equal when ntot is odd
  } else {
    rank = s[2];
  }
  const unsigned int* h = level == 1 ? hist + (size_t)j * MED_BINS : hist + ((size_t)j * 2 + t) * MED_BINS;
  constexpr int PER = MED_BINS / 256;
  long long mine = 0;
  for (int b = 0; b < PER; ++b) mine += h[threadIdx.x * PER + b];
  __shared__ long long cum[256];
  cum[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {            // inclusive Hillis-Steele scan
    long long v = threadIdx.x >= o ? cum[threadIdx.x - o] : 0;
    __syncthreads();
    cum[threadIdx.x] += v;
    __syncthreads();
  }
  const long long before = cum[threadIdx.x] - mine;
  const bool owner = rank >= before && rank < cum[threadIdx.x];
  const bool overflow_owner = threadIdx.x == 255 && rank >= cum[255];   // rank beyond the histogram: last bin
  if (owner || overflow_owner) {
    long long c = before;
    int b = 0;
    for (; b < PER; ++b) {
      long long hb = h[threadIdx.x * PER + b];
      if (rank < c + hb) break;
      c += hb;
    }
    if (b >= PER) { b = PER - 1; c -= h[threadIdx.x * PER + b]; }
    s[level - 1] = threadIdx.x * PER + b;
    s[2] = rank - c;
  }
}

// level 2: hist2[j][t][bin2] += count for rows whose first digit is the target's
__global__ void __launch_bounds__(256) k_med_hist2(const double* __restrict__ hu, int64_t cap, const int32_t* __restrict__ cnt,
                                                   const int32_t* __restrict__ labels, int label, int64_t n,
                                                   const double* __restrict__ range, const long long* __restrict__ sel,
                                                   unsigned int* __restrict__ hist2) {
  __shared__ unsigned int h[2 * MED_BINS];
  const int j = blockIdx.y;
  const double lo = range[2 * j], scale = med_scale(range[2 * j], range[2 * j + 1]);
  for (int b = threadIdx.x; b < 2 * MED_BINS; b += blockDim.x) h[b] = 0;
  __syncthreads();
  const int b0 = (int)sel[((size_t)j * 2 + 0) * 4], b1 = (int)sel[((size_t)j * 2 + 1) * 4];
  const double* src = hu + (size_t)j * cap;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {     // four rows in flight per lane
    int c[4];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      c[k] = (i < n && (!labels || labels[i] == label)) ? cnt[i] : 0;
      v[k] = i < n ? src[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c[k] == 0) continue;
      int d1, d2;
      med_digits(v[k], lo, scale, d1, d2);
      if (d1 == b0) atomicAdd(&h[d2], (unsigned int)c[k]);
      if (d1 == b1) atomicAdd(&h[MED_BINS + d2], (unsigned int)c[k]);
    }
  }
  __syncthreads();
  unsigned int* g = hist2 + (size_t)j * 2 * MED_BINS;
  for (int b = threadIdx.x; b < 2 * MED_BINS; b += blockDim.x)
    if (h[b]) atomicAdd(&g[b], h[b]);
}

// collect (value, count) of the rows in each target's final bin
__global__ void __launch_bounds__(256) k_med_collect(const double* __restrict__ hu, int64_t cap, const int32_t* __restrict__ cnt,
                                                     const int32_t* __restrict__ labels, int label, int64_t n,
                                                     const double* __restrict__ range, const long long* __restrict__ sel,
                                                     double* __restrict__ vals, int* __restrict__ cnts, int* __restrict__ fill) {
  const int j = blockIdx.y;
  const double lo = range[2 * j], scale = med_scale(range[2 * j], range[2 * j + 1]);
  const long long* s0 = sel + ((size_t)j * 2 + 0) * 4;
  const long long* s1 = sel + ((size_t)j * 2 + 1) * 4;
  const int a0 = (int)s0[0], a1 = (int)s0[1], c0 = (int)s1[0], c1 = (int)s1[1];
  const double* src = hu + (size_t)j * cap;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {     // four rows in flight per lane
    int cs[4];
    double vs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      cs[k] = (i < n && (!labels || labels[i] == label)) ? cnt[i] : 0;
      vs[k] = i < n ? src[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cs[k];
      if (c == 0) continue;
      const double v = vs[k];
      int d1, d2;
      med_digits(v, lo, scale, d1, d2);
      if (d1 == a0 && d2 == a1) {
        int slot = atomicAdd(&fill[j * 2 + 0], 1);
        if (slot < MED_CAP) { vals[((size_t)j * 2 + 0) * MED_CAP + slot] = v; cnts[((size_t)j * 2 + 0) * MED_CAP + slot] = c; }
      }
      if (d1 == c0 && d2 == c1) {
        int slot = atomicAdd(&fill[j * 2 + 1], 1);
        if (slot < MED_CAP) { vals[((size_t)j * 2 + 1) * MED_CAP + slot] = v; cnts[((size_t)j * 2 + 1) * MED_CAP + slot] = c; }
      }
    }
  }
}

// exact order statistic inside the collected bin; median_j = (v_lo + v_hi)/2 (np.median)
__global__ void __launch_bounds__(256) k_med_finish(const long long* __restrict__ sel, const double* __restrict__ range,
                                                    const double* __restrict__ vals, const int* __restrict__ cnts,
                                                    const int* __restrict__ fill, double* __restrict__ median,
                                                    int* __restrict__ overflow) {
  const int j = blockIdx.x;
  const double lo = range[2 * j], width = range[2 * j + 1] - range[2 * j];
  __shared__ double res[2];
  __shared__ double sv[MED_CAP];
  __shared__ int sc[MED_CAP];
  for (int t = 0; t < 2; ++t) {
    const long long* s = sel + ((size_t)j * 2 + t) * 4;
    int m = fill[j * 2 + t];
    long long rank = s[2];
    __syncthreads();
    if (threadIdx.x == 0) res[t] = lo + width * (((double)s[0] + (double)s[1] / 4096.0) / 4096.0);  // bin edge if overflow
    if (m > MED_CAP) { if (threadIdx.x == 0) atomicAdd(overflow, 1); m = 0; }
    for (int e = threadIdx.x; e < m; e += blockDim.x) {
      sv[e] = vals[((size_t)j * 2 + t) * MED_CAP + e];
      sc[e] = cnts[((size_t)j * 2 + t) * MED_CAP + e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < m; e += blockDim.x) {
      double v = sv[e];
      long long below = 0, eq = 0;
      for (int f = 0; f < m; ++f) {
        double o = sv[f];
        below += o < v ? sc[f] : 0;
        eq += o == v ? sc[f] : 0;
      }
      if (below <= rank && rank < below + eq) res[t] = v;  // all writers hold the same value
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) median[j] = (res[0] + res[1]) / 2.0;
}

// -------------------------------------------------------------------------- Cholesky + inverse
// One block per mode.  LinAlgError <=> a non-positive (or NaN) pivot; then the reference adds
// max(1e-6, 1e-6*|trace|) to the diagonal and retries (student.py:75-79, modes.py:111-119).
// inv = L^-T L^-1 with W = L^-1 in scratch.
// The factor (and, up to 96-D, its inverse) are built in LDS: every element of L is read O(d) times by serial chains -- thread 0's
// pivot sum, each thread's column of the triangular solve -- and from global memory each of those reads was a trip to L2 (350 us
// for one 50 x 50 matrix, 1.6 ms at 100-D).  Same operations in the same order: the results are those of the global-memory form.
__global__ void __launch_bounds__(256) k_chol_inv(double* __restrict__ covs, int d, double* __restrict__ chols,
                                                  double* __restrict__ invs, double* __restrict__ work, int w_lds) {
  extern __shared__ double cl[];
  double* A = covs + (size_t)blockIdx.x * d * d;
  double* Lg = chols + (size_t)blockIdx.x * d * d;
  double* Ainv = invs + (size_t)blockIdx.x * d * d;
  double* Wg = work + (size_t)blockIdx.x * d * d;
  double* L = cl;
  double* W = w_lds ? cl + (size_t)d * d : Wg;
  __shared__ int fail;
  __shared__ double piv;
  for (int round = 0; round < 3; ++round) {
    __syncthreads();
    if (threadIdx.x == 0) fail = 0;
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) L[e] = 0.0;
    __syncthreads();
    for (int j = 0; j < d; ++j) {
      if (threadIdx.x == 0) {
        double s = A[j * d + j];
        for (int k = 0; k < j; ++k) s -= L[j * d + k] * L[j * d + k];
        if (!(s > 0.0)) fail = 1;
        piv = sqrt(s);
        L[j * d + j] = piv;
      }
      __syncthreads();
      if (fail) break;
      for (int i = j + 1 + threadIdx.x; i < d; i += blockDim.x) {
        double s = A[i * d + j];
        for (int k = 0; k < j; ++k) s -= L[i * d + k] * L[j * d + k];
        L[i * d + j] = s / piv;
      }
      __syncthreads();
    }
    __syncthreads();
    if (!fail || round == 2) break;
    if (threadIdx.x == 0) {
      double tr = 0.0;
      for (int j = 0; j < d; ++j) tr += A[j * d + j];
      double reg = fmax(1e-6, 1e-6 * fabs(tr));
      for (int j = 0; j < d; ++j) A[j * d + j] += reg;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) Lg[e] = L[e];
  for (int c = threadIdx.x; c < d; c += blockDim.x) {  // column c of W solves L y = e_c
    for (int i = 0; i < d; ++i) {
      if (i < c) { W[i * d + c] = 0.0; continue; }
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= L[i * d + k] * W[k * d + c];
      W[i * d + c] = s / L[i * d + i];
    }
  }
  __syncthreads();
  if (w_lds)
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) Wg[e] = W[e];
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
    int i = e / d, j = e % d;
    int m = i > j ? i : j;
    double s = 0.0;
    for (int k = m; k < d; ++k) s += W[k * d + i] * W[k * d + j];
    Ainv[e] = s;
  }
}

extern "C" int tph_chol_inv(tph_ctx* ctx, double* covs_dev, int K, double* chol_dev, double* inv_dev, double* cholinv_dev) {
  TPH_REQUIRE(ctx && covs_dev && chol_dev && inv_dev && K >= 1, "tph_chol_inv: bad argument");
  double* work = cholinv_dev;            // W = L^-1: an output when asked for, scratch otherwise
  if (!work) {
    size_t need = sizeof(double) * (size_t)K * ctx->d * ctx->d;
    if (tph_scratch_reserve(ctx, need)) return -1;
    work = (double*)ctx->scratch;
  }
  const int d_ = ctx->d;
  const int w_lds = 2 * sizeof(double) * (size_t)d_ * d_ <= 150 * 1024 ? 1 : 0;
  const size_t lds = sizeof(double) * (size_t)d_ * d_ * (w_lds ? 2 : 1);
  if (lds > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_chol_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_chol_inv, dim3(K), dim3(256), lds, ctx->stream, covs_dev, ctx->d, chol_dev, inv_dev, work, w_lds);
  TPH_LAUNCH_CHECK();
  return 0;
}

// W = L^-1 for K lower-triangular factors (column c of W solves L y = e_c): for callers of tph_propose that hold only L
__global__ void __launch_bounds__(256) k_tri_inv(const double* __restrict__ chols, int d, double* __restrict__ winv) {
  const double* L = chols + (size_t)blockIdx.x * d * d;
  double* W = winv + (size_t)blockIdx.x * d * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    for (int i = 0; i < d; ++i) {
      if (i < c) { W[i * d + c] = 0.0; continue; }
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= L[i * d + k] * W[k * d + c];
      W[i * d + c] = s / L[i * d + i];
    }
  }
}
int tph_tri_inv(tph_ctx* ctx, const double* chol_dev, int K, double* winv_dev) {
  hipLaunchKernelGGL(k_tri_inv, dim3(K), dim3(256), 0, ctx->stream, chol_dev, ctx->d, winv_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------ fit_modes
// ---- the working set of the fit: the rows with a multiplicity, shard by shard -------------------------------------------------
// The fit reads u five times (first moments, covariance, two histogram levels, candidate collection), but only rows with a
// non-zero multiplicity matter, and their share falls as the history grows (4 n_particles kept rows of N_h: 60 % at iteration
// 6, 13 % at iteration 24 of the bench run).  One order-preserving stream compaction (block counts -> offsets -> scatter)
// gathers those rows into a dense SoA working set; the five passes then stream 8d B per KEPT row instead of per history row.
// The compaction walks the history VIRTUAL SHARD BY VIRTUAL SHARD (common.h: tph_part; inside a shard in iteration order), so
// every shard's kept rows are ONE stretch of the working set -- a segment.  The moment kernels reduce each segment as if it
// were the whole input (SEG_SHIFT above) and the per-shard results are added in shard order: the same summation tree whether
// the V shards live on one GPU or on G, i.e. a fitted covariance that does not depend on the number of ranks.
constexpr int NZ_ROWS = 1024;   // history rows per block (4 tiles of 256)
constexpr int64_t FIT_COMPACT_MIN = 262144;   // below this the working set is sized for the whole history (no host read of the kept count)
struct nz_geom {
  long long n_loc, nv, n;         // rows per iteration on this rank | rows per piece | rows of the history
  int T, vl, bpp;                 // pieces per shard | shards | blocks per piece
};
// block b (shard-major: shard, iteration, block of the piece) -> its rows [r0, r0 + cnt)
__device__ __forceinline__ void nz_block_rows(const nz_geom& g, int b, long long& r0, int& cnt) {
  const int per_shard = g.T * g.bpp;
  const int v = b / per_shard, rem = b - v * per_shard;
  const int t = rem / g.bpp, sb = rem - t * g.bpp;
  const long long q0 = (long long)sb * NZ_ROWS;
  r0 = (long long)t * g.n_loc + (long long)v * g.nv + q0;
  const long long left = g.nv - q0;
  cnt = (int)(left < NZ_ROWS ? left : NZ_ROWS);
}
__global__ void __launch_bounds__(256) k_nz_count(const int32_t* __restrict__ counts, nz_geom g, int* __restrict__ blockcnt) {
  __shared__ int s_c;
  if (threadIdx.x == 0) s_c = 0;
  __syncthreads();
  long long base; int rows;
  nz_block_rows(g, blockIdx.x, base, rows);
  int c = 0;
#pragma unroll
  for (int t = 0; t < NZ_ROWS / 256; ++t) {
    const int q = t * 256 + threadIdx.x;
    c += (q < rows && counts[base + q] > 0) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_c, c);     // integer: exact, order-free
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = s_c;
}
// exclusive scan of the block counts in place (one block); total[0] = number of kept rows.  A thread owns a run of consecutive
// blocks; it requests them eight at a time (one by one the run is a chain of memory round trips: 44 us at 25 600 blocks)
__global__ void __launch_bounds__(1024) k_nz_offsets(int* __restrict__ blockcnt, int nblocks, long long* __restrict__ total) {
  __shared__ long long s_part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = lo + per < nblocks ? lo + per : nblocks;
  long long sum = 0;
  for (int b = lo; b < hi; b += 8) {
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = b + k < hi ? blockcnt[b + k] : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  // exclusive scan of the 1024 run totals: inside each wave by shuffles, the 16 wave totals by the first lanes
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  long long incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  __shared__ long long s_wtot[16];
  if (lane == 63) s_wtot[wid] = incl;
  __syncthreads();
  long long woff = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { const long long v = s_wtot[w]; if (w < wid) woff += v; all += v; }
  if (threadIdx.x == 0) total[0] = all;
  long long run = woff + incl - sum;
  for (int b = lo; b < hi; b += 8) {
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = b + k < hi ? blockcnt[b + k] : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (b + k < hi) { blockcnt[b + k] = (int)run; run += v[k]; }
  }
}
// seg[2 v], seg[2 v + 1] = first kept row and number of kept rows of shard v (from the exclusive offsets of its first block)
__global__ void k_nz_segments(const int* __restrict__ offsets, const long long* __restrict__ total, int vl, int per_shard,
                              long long* __restrict__ seg) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= vl) return;
  const long long a = offsets[(size_t)v * per_shard], b = v + 1 < vl ? (long long)offsets[(size_t)(v + 1) * per_shard] : total[0];
  seg[2 * v] = a;
  seg[2 * v + 1] = b - a;
}
__global__ void k_seg_whole(long long n, long long* __restrict__ seg) { if (!threadIdx.x && !blockIdx.x) { seg[0] = 0; seg[1] = n; } }
// A block first lists its kept rows in LDS (order preserved: ballots and the 16 wave counts), then moves them with EVERY lane
// busy: lane k of a pass takes kept row k -- neighbouring lanes read neighbouring kept rows of one coordinate and write
// neighbouring elements of the working set.  (With one lane per HISTORY row, as until round 5, a wave's load instruction carried
// as many useful loads as the wave had kept rows: 8 of 64 where 13 % of the rows are kept.)
__global__ void __launch_bounds__(256) k_nz_scatter(const double* __restrict__ u, int64_t cap, int d,
                                                    const int32_t* __restrict__ counts, const int32_t* __restrict__ labels,
                                                    nz_geom g, const int* __restrict__ offsets, double* __restrict__ uc,
                                                    int64_t ldc, int32_t* __restrict__ cc, int32_t* __restrict__ lc) {
  constexpr int NT = NZ_ROWS / 256;
  __shared__ int s_wave[NT * 4];
  __shared__ int s_row[NZ_ROWS];
  __shared__ int s_cnt[NZ_ROWS];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t out = offsets[blockIdx.x];
  long long base; int rows;
  nz_block_rows(g, blockIdx.x, base, rows);
  int cnt[NT], below[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = t * 256 + threadIdx.x;
    cnt[t] = q < rows ? counts[base + q] : 0;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const unsigned long long mask = __ballot(cnt[t] > 0);
    below[t] = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[t * 4 + wid] = __popcll(mask);
  }
  __syncthreads();
  int nk = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int c = s_wave[t * 4 + w]; if (w < wid) before += c; nk += c; }
    if (cnt[t] > 0) {
      const int k = nk - (s_wave[t * 4] + s_wave[t * 4 + 1] + s_wave[t * 4 + 2] + s_wave[t * 4 + 3]) + before + below[t];
      s_row[k] = t * 256 + threadIdx.x;
      s_cnt[k] = cnt[t];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nk; k += 256) {
    const int64_t i = base + s_row[k], pos = out + k;
    cc[pos] = s_cnt[k];
    if (lc) lc[pos] = labels[i];
    for (int j0 = 0; j0 < d; j0 += 8) {          // eight coordinates in flight per kept row
      double v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = j0 + c < d ? u[(size_t)(j0 + c) * cap + i] : 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (j0 + c < d) uc[(size_t)(j0 + c) * ldc + pos] = v[c];
    }
  }
}

// ---- proposal fit over a SHARDED history (tph_comm_attach): the fit of the global up-sampled set ---------------------------
// The reference fits ONE Gaussian per mode to the whole weighted history (train.py:91-122, modes.py:131-288).  Every
// statistic of that fit is a sum or an order statistic over rows, so each rank runs the same kernels on its shard and the
// small intermediate results are combined: (count, sum) and (min, max) per coordinate, the centred second moments about
// the GLOBAL mean, the two histogram levels of the median select, and the (value, multiplicity) candidates of the final
// median bins (gathered from all ranks; they are as few as on one GPU).  mean/covariance equal the one-GPU values to
// rounding, the medians exactly.
__global__ void k_pack_range(const double* __restrict__ range, int d, double* __restrict__ out) {   // (-min, max): one MAX all-reduce
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d) { out[2 * j] = -range[2 * j]; out[2 * j + 1] = range[2 * j + 1]; }
}
__global__ void k_unpack_range(const double* __restrict__ in, int d, double* __restrict__ range) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d) { range[2 * j] = -in[2 * j]; range[2 * j + 1] = in[2 * j + 1]; }
}
// candidates of all ranks -> exact order statistic; lists [G][d*2][cap] with fill[G][d*2]
__global__ void __launch_bounds__(256) k_med_finish_g(const long long* __restrict__ sel, const double* __restrict__ range,
                                                      const double* __restrict__ vals, const int* __restrict__ cnts,
                                                      const int* __restrict__ fill, int G, int d, int cap,
                                                      double* __restrict__ median, int* __restrict__ overflow) {
  const int j = blockIdx.x;
  const double lo = range[2 * j], width = range[2 * j + 1] - range[2 * j];
  __shared__ double res[2];
  __shared__ double sv[MED_CAP];
  __shared__ int sc[MED_CAP];
  __shared__ int s_m;
  for (int t = 0; t < 2; ++t) {
    const long long* s = sel + ((size_t)j * 2 + t) * 4;
    const long long rank = s[2];
    __syncthreads();
    if (threadIdx.x == 0) {
      res[t] = lo + width * (((double)s[0] + (double)s[1] / 4096.0) / 4096.0);  // bin edge if overflow
      int m = 0;
      for (int g = 0; g < G; ++g) m += fill[(size_t)g * d * 2 + j * 2 + t];
      if (m > MED_CAP) { atomicAdd(overflow, 1); m = 0; }
      s_m = m;
    }
    __syncthreads();
    const int m = s_m;
    if (m > 0) {
      int base = 0;
      for (int g = 0; g < G; ++g) {
        const int f = fill[(size_t)g * d * 2 + j * 2 + t];
        for (int e = threadIdx.x; e < f; e += blockDim.x) {
          sv[base + e] = vals[((size_t)g * d * 2 + j * 2 + t) * cap + e];
          sc[base + e] = cnts[((size_t)g * d * 2 + j * 2 + t) * cap + e];
        }
        base += f;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < m; e += blockDim.x) {
      const double v = sv[e];
      long long below = 0, eq = 0;
      for (int f = 0; f < m; ++f) {
        const double o = sv[f];
        below += o < v ? sc[f] : 0;
        eq += o == v ? sc[f] : 0;
      }
      if (below <= rank && rank < below + eq) res[t] = v;  // all writers hold the same value
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) median[j] = (res[0] + res[1]) / 2.0;
}
// pack the first `cap` candidates of every (dimension, target) list contiguously: [d*2][cap]
__global__ void k_med_pack(const double* __restrict__ vals, const int* __restrict__ cnts, int d, int cap, double* __restrict__ pv,
                           int* __restrict__ pc) {
  const int jt = blockIdx.x;
  for (int e = threadIdx.x; e < cap; e += blockDim.x) {
    pv[(size_t)jt * cap + e] = vals[(size_t)jt * MED_CAP + e];
    pc[(size_t)jt * cap + e] = cnts[(size_t)jt * MED_CAP + e];
  }
}


// The proposal fit of the up-sampled set (tempest/train.py:91-122, modes.py:131-288, student.py:60-64), on one GPU or over a
// sharded history: every statistic is a sum or an order statistic over rows.  Sums are formed per virtual shard and folded in
// shard order (with a communicator the per-shard sums of all ranks are all-gathered first: rank order = shard order); ranges and
// the histogram levels of the median select are order-free (max / integer sums: all-reduced), the (value, multiplicity)
// candidates of the final median bins are gathered.  `use_comm` false with a communicator attached: the fit of this rank's rows.
static int fit_modes_impl(tph_ctx* ctx, const int32_t* counts_dev, const int32_t* labels_dev, int64_t n, int K, double* means_dev,
                          double* covs_dev, double* chol_dev, double* inv_dev, double* cholinv_dev, bool use_comm) {
  const int d = ctx->d, G = use_comm ? ctx->world : 1;
  const int npl = d * (d + 1) / 2;
  TPH_REQUIRE(npl <= COV_NPT * 256, "covariance kernel supports n_dim <= 100 (got %d)", d);
  const tph_part part = tph_partition(ctx, n);
  const int vl = part.vl, V = use_comm ? part.vl * ctx->world : part.vl;
  const int64_t n_hist = n;
  // ---- working set
  nz_geom g;
  g.n_loc = part.n_loc; g.nv = part.nv; g.n = n_hist; g.T = part.T; g.vl = vl;
  g.bpp = (int)((part.nv + NZ_ROWS - 1) / NZ_ROWS);
  const int per_shard = g.T * g.bpp, nzb = vl * per_shard;
  const bool pieces = part.T * vl > 1;
  const bool compact = pieces || n_hist >= FIT_COMPACT_MIN;
  const double* src = ctx->u;
  int64_t src_ld = ctx->cap;
  int64_t m_cap = n_hist;                         // rows of the working set (its leading dimension); kept rows <= m_cap
  int* blockcnt = nullptr;
  long long* total = (long long*)ctx->small_dev;
  if (compact) {
    if (tph_partials_reserve(ctx, sizeof(int) * (size_t)nzb)) return -1;
    blockcnt = (int*)ctx->partials;
    hipLaunchKernelGGL(k_nz_count, dim3(nzb), dim3(256), 0, ctx->stream, counts_dev, g, blockcnt);
    hipLaunchKernelGGL(k_nz_offsets, dim3(1), dim3(1024), 0, ctx->stream, blockcnt, nzb, total);
    TPH_LAUNCH_CHECK();
    if (n_hist >= FIT_COMPACT_MIN) {              // large: the working set is sized by the kept rows (one 8-byte host read)
      TPH_HIP(hipMemcpyAsync(ctx->pinned, total, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
      TPH_HIP(hipStreamSynchronize(ctx->stream));
      const long long m = *(long long*)ctx->pinned;
      m_cap = m > 0 ? m : 1;
      if (!pieces && 2 * m > n_hist) m_cap = -1;  // one segment and most rows kept: streaming the history itself is as cheap
    }
  }
  const bool dense = compact && m_cap > 0;
  if (!dense) m_cap = n_hist;
  // block counts of the reductions: functions of the SHARD's shape (rows per shard of the history) and of V, never of vl
  const int64_t rows_v = part.nv * (int64_t)part.T;
  const int Vt = part.canonical ? part.V : 1;
  int rblk = tph_grid_for(rows_v, 256, 4, 512 / Vt > 1 ? 512 / Vt : 1);
  int nblk = cov_blocks(rows_v);
  { const int capb = (d <= 12 ? 768 : 2048) / Vt; if (nblk > capb) nblk = capb > 1 ? capb : 1; }
  const int rpb = d <= 12 ? 1 : ((d >= 16) ? 1 : cov_slices(npl));
  // scratch layout
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
  size_t o_part = take(sizeof(double) * ((size_t)vl * nblk * (d <= 12 ? 1 : cov_slices(npl)) * npl));   // cov block partials
  size_t o_part1 = take(sizeof(double) * (size_t)vl * rblk * (1 + d) * 3);                                // first-moment partials
  size_t o_vs = take(sizeof(double) * (size_t)V * (size_t)(npl > 1 + d ? npl : 1 + d));                   // per-shard sums (local, or gathered)
  size_t o_vr = take(sizeof(double) * (size_t)vl * 2 * d);
  size_t o_sums = take(sizeof(double) * (1 + d));
  size_t o_csum = take(sizeof(double) * npl);
  size_t o_range = take(sizeof(double) * 2 * d);
  size_t o_mean = take(sizeof(double) * d);
  size_t o_seg = take(sizeof(long long) * 2 * (size_t)vl);
  size_t o_h1 = take(sizeof(unsigned int) * (size_t)d * MED_BINS);
  size_t o_h2 = take(sizeof(unsigned int) * (size_t)d * 2 * MED_BINS);
  size_t o_sel = take(sizeof(long long) * (size_t)d * 2 * 4);
  size_t o_vals = take(sizeof(double) * (size_t)d * 2 * MED_CAP);
  size_t o_cnts = take(sizeof(int) * (size_t)d * 2 * MED_CAP);
  size_t o_fill = take(sizeof(int) * ((size_t)d * 2 + 1));
  size_t o_uc = 0, o_cc = 0, o_lc = 0;
  if (dense) {
    o_uc = take(sizeof(double) * (size_t)d * (size_t)m_cap);
    o_cc = take(sizeof(int32_t) * (size_t)m_cap);
    o_lc = take(sizeof(int32_t) * (size_t)m_cap);
  }
  if (tph_scratch_reserve(ctx, o)) return -1;
  char* base = (char*)ctx->scratch;
  long long* seg = (long long*)(base + o_seg);
  int64_t n_work = n_hist;                        // rows the order-free passes (histograms) walk
  if (dense) {
    double* uc = (double*)(base + o_uc);
    int32_t* cc = (int32_t*)(base + o_cc);
    int32_t* lc = K > 1 ? (int32_t*)(base + o_lc) : nullptr;
    if (n_hist < FIT_COMPACT_MIN) TPH_HIP(hipMemsetAsync(cc, 0, sizeof(int32_t) * (size_t)m_cap, ctx->stream));   // rows behind the kept ones count 0
    hipLaunchKernelGGL(k_nz_scatter, dim3(nzb), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, d, counts_dev, labels_dev, g, blockcnt, uc,
                       m_cap, cc, lc);
    hipLaunchKernelGGL(k_nz_segments, dim3((vl + 63) / 64), dim3(64), 0, ctx->stream, (const int*)blockcnt, (const long long*)total, vl, per_shard, seg);
    TPH_LAUNCH_CHECK();
    src = uc; src_ld = m_cap; counts_dev = cc;
    if (K > 1) labels_dev = lc;
    n_work = m_cap;
  } else {
    hipLaunchKernelGGL(k_seg_whole, dim3(1), dim3(1), 0, ctx->stream, (long long)n_hist, seg);
    TPH_LAUNCH_CHECK();
  }
  double* part_c = (double*)(base + o_part);
  double* part1 = (double*)(base + o_part1);
  double* vs = (double*)(base + o_vs);
  double* vr = (double*)(base + o_vr);
  double* sums = (double*)(base + o_sums);
  double* csum = (double*)(base + o_csum);
  double* range = (double*)(base + o_range);
  double* mean = (double*)(base + o_mean);
  unsigned int* h1 = (unsigned int*)(base + o_h1);
  unsigned int* h2 = (unsigned int*)(base + o_h2);
  long long* sel = (long long*)(base + o_sel);
  double* vals = (double*)(base + o_vals);
  int* cnts = (int*)(base + o_cnts);
  int* fill = (int*)(base + o_fill);
  int* overflow = fill + (size_t)d * 2;
  // staging of the sharded fit (fixed offsets; the largest user is the second histogram level / the gathered candidates / the
  // gathered covariance sums)
  const size_t c_rng = 0;                                                            // 2d f64
  const size_t c_big = ((sizeof(double) * 2 * d) + 255) / 256 * 256;                 // shard sums | histograms | candidate lists
  if (use_comm) {
    const size_t gather_need = sizeof(double) * (size_t)(vl + V) * (size_t)(npl > 1 + d ? npl : 1 + d) + 512;
    const size_t med_need = sizeof(unsigned int) * (size_t)d * 2 * MED_BINS + sizeof(int) * (size_t)d * 2 * (G + 1) + 4096;
    if (tph_comm_require(ctx, c_big + (gather_need > med_need ? gather_need : med_need), "tph_fit_modes_global")) return -2;
    TPH_HIP(hipMemsetAsync(overflow, 0, sizeof(int), ctx->stream));
  }
  // per-shard sums [vl][ncol] -> the run's sums: all-gather (rank order = shard order) when sharded, then the fold in shard order
  auto fold_shards = [&](const double* mine, int ncol, double* out) -> int {
    const double* rows = mine;
    if (use_comm) {
      const size_t one = sizeof(double) * (size_t)vl * ncol, all_off = c_big + (one + 255) / 256 * 256;
      TPH_HIP(hipMemcpyAsync(ctx->comm_buf + c_big, mine, one, hipMemcpyDeviceToDevice, ctx->stream));
      if (tph_comm_allgather(ctx, c_big, all_off, (int64_t)vl * ncol, TPH_DT_F64)) return -2;
      rows = (const double*)(ctx->comm_buf + all_off);
    }
    hipLaunchKernelGGL(k_fold_cols, dim3((ncol + 255) / 256), dim3(256), 0, ctx->stream, rows, V, ncol, out);
    TPH_LAUNCH_CHECK();
    return 0;
  };
  for (int k = 0; k < K; ++k) {
    const int32_t* lab = K > 1 ? labels_dev : nullptr;
    // first moments -> arithmetic mean (centre of np.cov), data range
    hipLaunchKernelGGL(k_wsum<int32_t>, dim3(rblk, 1 + d, vl), dim3(256), 0, ctx->stream, src, src_ld, d, counts_dev, lab, k, n_work, part1,
                       (const long long*)seg);
    hipLaunchKernelGGL(k_wsum_final, dim3(1 + d, vl), dim3(256), 0, ctx->stream, part1, rblk, 1 + d, vs, vr);
    hipLaunchKernelGGL(k_fold_range, dim3((d + 255) / 256), dim3(256), 0, ctx->stream, (const double*)vr, vl, d, range);
    TPH_LAUNCH_CHECK();
    if (fold_shards(vs, 1 + d, sums)) return -2;
    if (use_comm) {
      double* rng = (double*)(ctx->comm_buf + c_rng);
      hipLaunchKernelGGL(k_pack_range, dim3((d + 63) / 64), dim3(64), 0, ctx->stream, range, d, rng);
      TPH_LAUNCH_CHECK();
      if (tph_comm_allreduce(ctx, c_rng, 2 * d, TPH_DT_F64, TPH_OP_MAX)) return -2;
      hipLaunchKernelGGL(k_unpack_range, dim3((d + 63) / 64), dim3(64), 0, ctx->stream, rng, d, range);
    }
    hipLaunchKernelGGL(k_mean_from_sums, dim3((d + 63) / 64), dim3(64), 0, ctx->stream, sums, d, mean);
    // covariance (student.py:62-63): centred second moments about the (global) mean, per shard, folded
    {
      int rows_pb = 1;
      if (d <= 12) {
        bool ok = launch_wcov_small<int32_t>(ctx, src, src_ld, counts_dev, lab, k, n_work, mean, part_c, nblk, seg, vl);
        TPH_REQUIRE(ok, "covariance: no register kernel for n_dim=%d", d);
      } else {
        if (launch_wcov<int32_t>(ctx, src, src_ld, counts_dev, lab, k, n_work, mean, part_c, nblk, &rows_pb, seg, vl)) return -1;
      }
      launch_colsum2(ctx, part_c, nblk * rows_pb, npl, vl, vs);
      TPH_LAUNCH_CHECK();
      if (fold_shards(vs, npl, csum)) return -2;
      hipLaunchKernelGGL(k_cov_finish, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, csum, sums, d, 1, covs_dev + (size_t)k * d * d);
      TPH_LAUNCH_CHECK();
    }
    (void)rpb;
    // per-dimension median (student.py:61): two histogram levels (all-reduced when sharded), then the candidates of the final bins
    dim3 hg(tph_grid_for(n_work, 256, 8, 256), d);
    if (!use_comm) {
      TPH_HIP(hipMemsetAsync(h1, 0, sizeof(unsigned int) * (size_t)d * MED_BINS, ctx->stream));
      TPH_HIP(hipMemsetAsync(h2, 0, sizeof(unsigned int) * (size_t)d * 2 * MED_BINS, ctx->stream));
      TPH_HIP(hipMemsetAsync(fill, 0, sizeof(int) * ((size_t)d * 2 + 1), ctx->stream));
      hipLaunchKernelGGL(k_med_hist1, hg, dim3(256), 0, ctx->stream, src, src_ld, counts_dev, lab, k, n_work, range, h1);
      hipLaunchKernelGGL(k_med_select, dim3(d, 2), dim3(256), 0, ctx->stream, h1, 1, sums, sel);
      hipLaunchKernelGGL(k_med_hist2, hg, dim3(256), 0, ctx->stream, src, src_ld, counts_dev, lab, k, n_work, range, sel, h2);
      hipLaunchKernelGGL(k_med_select, dim3(d, 2), dim3(256), 0, ctx->stream, h2, 2, sums, sel);
      hipLaunchKernelGGL(k_med_collect, hg, dim3(256), 0, ctx->stream, src, src_ld, counts_dev, lab, k, n_work, range, sel, vals, cnts, fill);
      hipLaunchKernelGGL(k_med_finish, dim3(d), dim3(256), 0, ctx->stream, sel, range, vals, cnts, fill, means_dev + (size_t)k * d, overflow);
      TPH_LAUNCH_CHECK();
    } else {
      unsigned int* h = (unsigned int*)(ctx->comm_buf + c_big);
      TPH_HIP(hipMemsetAsync(h, 0, sizeof(unsigned int) * (size_t)d * MED_BINS, ctx->stream));
      hipLaunchKernelGGL(k_med_hist1, hg, dim3(256), 0, ctx->stream, src, src_ld, counts_dev, lab, k, n_work, range, h);
      TPH_LAUNCH_CHECK();
      if (tph_comm_allreduce(ctx, c_big, (int64_t)d * MED_BINS, TPH_DT_I32, TPH_OP_SUM)) return -2;
      hipLaunchKernelGGL(k_med_select, dim3(d, 2), dim3(256), 0, ctx->stream, h, 1, sums, sel);
      TPH_HIP(hipMemsetAsync(h, 0, sizeof(unsigned int) * (size_t)d * 2 * MED_BINS, ctx->stream));
      hipLaunchKernelGGL(k_med_hist2, hg, dim3(256), 0, ctx->stream, src, src_ld, counts_dev, lab, k, n_work, range, sel, h);
      TPH_LAUNCH_CHECK();
      if (tph_comm_allreduce(ctx, c_big, (int64_t)d * 2 * MED_BINS, TPH_DT_I32, TPH_OP_SUM)) return -2;
      hipLaunchKernelGGL(k_med_select, dim3(d, 2), dim3(256), 0, ctx->stream, h, 2, sums, sel);
      // candidates: fill counts first (their maximum sizes the gather), then the packed lists
      int* fill_mine = (int*)(ctx->comm_buf + c_big);                 // [d*2]
      int* fill_all = fill_mine + (size_t)d * 2;                      // [G][d*2]
      TPH_HIP(hipMemsetAsync(fill_mine, 0, sizeof(int) * (size_t)d * 2, ctx->stream));
      hipLaunchKernelGGL(k_med_collect, hg, dim3(256), 0, ctx->stream, src, src_ld, counts_dev, lab, k, n_work, range, sel, vals, cnts, fill_mine);
      TPH_LAUNCH_CHECK();
      if (tph_comm_allgather(ctx, c_big, c_big + sizeof(int) * (size_t)d * 2, (int64_t)d * 2, TPH_DT_I32)) return -2;
      std::vector<int> fh((size_t)G * d * 2);
      TPH_HIP(hipMemcpyAsync(fh.data(), fill_all, sizeof(int) * fh.size(), hipMemcpyDeviceToHost, ctx->stream));
      TPH_HIP(hipStreamSynchronize(ctx->stream));
      int cap = 1;
      for (int v : fh) cap = v > cap ? v : cap;
      if (cap > MED_CAP) cap = MED_CAP;
      cap = (cap + 7) / 8 * 8;
      // lists: values [d*2][cap] f64 then counts [d*2][cap] i32, gathered separately behind the fill table
      const size_t c_fill = c_big + sizeof(int) * (size_t)d * 2;                                   // fill_all stays here
      const size_t c_pv = (c_fill + sizeof(int) * (size_t)G * d * 2 + 255) / 256 * 256;            // my values
      const size_t pv_bytes = sizeof(double) * (size_t)d * 2 * cap, pc_bytes = sizeof(int) * (size_t)d * 2 * cap;
      const size_t c_av = c_pv + (pv_bytes + 255) / 256 * 256;                                     // all values [G][d*2][cap]
      const size_t c_pc = c_av + ((size_t)G * pv_bytes + 255) / 256 * 256;                         // my counts
      const size_t c_ac = c_pc + (pc_bytes + 255) / 256 * 256;                                     // all counts
      if (tph_comm_require(ctx, c_ac + (size_t)G * pc_bytes, "tph_fit_modes_global (median candidates)")) return -2;
      hipLaunchKernelGGL(k_med_pack, dim3(d * 2), dim3(256), 0, ctx->stream, vals, cnts, d, cap, (double*)(ctx->comm_buf + c_pv),
                         (int*)(ctx->comm_buf + c_pc));
      TPH_LAUNCH_CHECK();
      if (tph_comm_allgather(ctx, c_pv, c_av, (int64_t)d * 2 * cap, TPH_DT_F64)) return -2;
      if (tph_comm_allgather(ctx, c_pc, c_ac, (int64_t)d * 2 * cap, TPH_DT_I32)) return -2;
      hipLaunchKernelGGL(k_med_finish_g, dim3(d), dim3(256), 0, ctx->stream, sel, range, (const double*)(ctx->comm_buf + c_av),
                         (const int*)(ctx->comm_buf + c_ac), (const int*)(ctx->comm_buf + c_fill), G, d, cap,
                         means_dev + (size_t)k * d, overflow);
      TPH_LAUNCH_CHECK();
    }
  }
  return tph_chol_inv(ctx, covs_dev, K, chol_dev, inv_dev, cholinv_dev);
}

extern "C" int tph_fit_modes(tph_ctx* ctx, const int32_t* counts_dev, const int32_t* labels_dev, int64_t n, int K,
                             double* means_dev, double* covs_dev, double* chol_dev, double* inv_dev, double* cholinv_dev) {
  TPH_REQUIRE(ctx && counts_dev && means_dev && covs_dev && chol_dev && inv_dev, "tph_fit_modes: NULL argument");
  TPH_REQUIRE(n > 0 && n <= ctx->size && K >= 1, "tph_fit_modes: bad sizes");
  TPH_REQUIRE(K == 1 || labels_dev, "tph_fit_modes: K>1 needs labels");
  return fit_modes_impl(ctx, counts_dev, labels_dev, n, K, means_dev, covs_dev, chol_dev, inv_dev, cholinv_dev, false);
}

extern "C" int tph_fit_modes_global(tph_ctx* ctx, const int32_t* counts_dev, const int32_t* labels_dev, int64_t n, int K,
                                    double* means_dev, double* covs_dev, double* chol_dev, double* inv_dev, double* cholinv_dev) {
  TPH_REQUIRE(ctx && counts_dev && means_dev && covs_dev && chol_dev && inv_dev, "tph_fit_modes_global: NULL argument");
  TPH_REQUIRE(n > 0 && n <= ctx->size && K >= 1, "tph_fit_modes_global: bad sizes");
  TPH_REQUIRE(K == 1 || labels_dev, "tph_fit_modes_global: K>1 needs labels");
  return fit_modes_impl(ctx, counts_dev, labels_dev, n, K, means_dev, covs_dev, chol_dev, inv_dev, cholinv_dev, ctx->comm_active());
}

// ------------------------------------------------------------------------------ volume variation
extern "C" int tph_weighted_moments(tph_ctx* ctx, const double* w_dev, int64_t n, double* mean_cov_dev) {
  TPH_REQUIRE(ctx && w_dev && mean_cov_dev && n > 0 && n <= ctx->size, "tph_weighted_moments: bad argument");
  const int d = ctx->d;
  const int nblk = cov_blocks(n);
  const int rblk = tph_grid_for(n, 256, 4, 512);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
  size_t o_part = take(sizeof(double) * cov_scratch_doubles(d, nblk));
  size_t o_part1 = take(sizeof(double) * (size_t)rblk * (1 + d) * 3);
  size_t o_sums = take(sizeof(double) * (1 + d));
  if (tph_scratch_reserve(ctx, o)) return -1;
  char* base = (char*)ctx->scratch;
  double* part = (double*)(base + o_part);
  double* part1 = (double*)(base + o_part1);
  double* sums = (double*)(base + o_sums);
  hipLaunchKernelGGL(k_wsum<double>, dim3(rblk, 1 + d), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, d, w_dev,
                     (const int32_t*)nullptr, 0, n, part1);
  hipLaunchKernelGGL(k_wsum_final, dim3(1 + d), dim3(256), 0, ctx->stream, part1, rblk, 1 + d, sums, (double*)nullptr);
  // tools.py:94-96: weights are normalised first, so mean = sum(w u)/sum(w)
  hipLaunchKernelGGL(k_mean_from_sums, dim3((d + 63) / 64), dim3(64), 0, ctx->stream, sums, d, mean_cov_dev);
  TPH_LAUNCH_CHECK();
  return moments_launch_cov(ctx, w_dev, false, nullptr, 0, n, mean_cov_dev, sums, 2, mean_cov_dev + d, part, nblk);
}

extern "C" int tph_weighted_sums(tph_ctx* ctx, const double* w_dev, int64_t n, double* sums_dev) {
  TPH_REQUIRE(ctx && w_dev && sums_dev && n > 0 && n <= ctx->size, "tph_weighted_sums: bad argument");
  const int d = ctx->d;
  const int rblk = tph_grid_for(n, 256, 4, 512);
  if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)rblk * (1 + d) * 3)) return -1;
  double* part1 = (double*)ctx->scratch;
  hipLaunchKernelGGL(k_wsum<double>, dim3(rblk, 1 + d), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, d, w_dev,
                     (const int32_t*)nullptr, 0, n, part1);
  hipLaunchKernelGGL(k_wsum_final, dim3(1 + d), dim3(256), 0, ctx->stream, part1, rblk, 1 + d, sums_dev, (double*)nullptr);
  TPH_LAUNCH_CHECK();
  return 0;
}

// the same two reductions on an explicit SoA array x[j*ld + i] (the clustering working set)
extern "C" int tph_x_weighted_sums(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* w_dev,
                                   double* sums_dev, double* range_dev) {
  TPH_REQUIRE(ctx && x_dev && w_dev && sums_dev && n > 0 && ld >= n, "tph_x_weighted_sums: bad argument");
  const int d = ctx->d;
  const int rblk = tph_grid_for(n, 256, 4, 512);
  if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)rblk * (1 + d) * 3)) return -1;
  double* part1 = (double*)ctx->scratch;
  hipLaunchKernelGGL(k_wsum<double>, dim3(rblk, 1 + d), dim3(256), 0, ctx->stream, x_dev, ld, d, w_dev, (const int32_t*)nullptr, 0,
                     n, part1);
  hipLaunchKernelGGL(k_wsum_final, dim3(1 + d), dim3(256), 0, ctx->stream, part1, rblk, 1 + d, sums_dev, range_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

extern "C" int tph_x_weighted_cov(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* w_dev,
                                  const double* mean_dev, double* cov_dev) {
  TPH_REQUIRE(ctx && x_dev && w_dev && mean_dev && cov_dev && n > 0 && ld >= n, "tph_x_weighted_cov: bad argument");
  const int d = ctx->d;
  const int nblk = cov_blocks(n);
  if (tph_scratch_reserve(ctx, sizeof(double) * cov_scratch_doubles(d, nblk))) return -1;
  return moments_launch_cov(ctx, w_dev, false, nullptr, 0, n, mean_dev, nullptr, 0, cov_dev, (double*)ctx->scratch, nblk, x_dev, ld);
}

extern "C" int tph_weighted_cov_centered(tph_ctx* ctx, const double* w_dev, int64_t n, const double* mean_dev, double* cov_dev) {
  TPH_REQUIRE(ctx && w_dev && mean_dev && cov_dev && n > 0 && n <= ctx->size, "tph_weighted_cov_centered: bad argument");
  const int d = ctx->d;
  const int nblk = cov_blocks(n);
  size_t need = sizeof(double) * cov_scratch_doubles(d, nblk);
  if (tph_scratch_reserve(ctx, need)) return -1;
  return moments_launch_cov(ctx, w_dev, false, nullptr, 0, n, mean_dev, nullptr, 0, cov_dev, (double*)ctx->scratch, nblk);
}

// `global`: the shifted raw sums (additive over rows) of all ranks are all-reduced before the finish; the centre must be the
// same on every rank (it is: the previous GLOBAL mean)
static int moments_shifted(tph_ctx* ctx, const double* w_dev, int64_t n, const double* centre_dev, double* out_dev, bool global) {
  const int d = ctx->d;
  TPH_REQUIRE(d <= 12, "tph_weighted_moments_shifted: n_dim=%d > 12 (use tph_weighted_sums + tph_weighted_cov_centered)", d);
  const int nc = d * (d + 1) / 2 + d + 1;
  const int nblk = tph_grid_for(n, 256, 8, 1024);
  if (tph_scratch_reserve(ctx, sizeof(double) * ((size_t)nblk + 1) * nc)) return -1;
  if (global && tph_comm_require(ctx, sizeof(double) * nc, "tph_volume_variation")) return -2;
  double* part = (double*)ctx->scratch;
  double* csum = global ? (double*)ctx->comm_buf : part + (size_t)nblk * nc;
  switch (d) {
#define C(DD) case DD: hipLaunchKernelGGL((k_wmom_small<DD>), dim3(nblk), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, w_dev, n, centre_dev, part); break;
    C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
  }
  launch_colsum2(ctx, part, nblk, nc, 1, csum);
  TPH_LAUNCH_CHECK();
  if (global && tph_comm_allreduce(ctx, 0, nc, TPH_DT_F64, TPH_OP_SUM)) return -2;
  hipLaunchKernelGGL(k_wmom_finish, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, csum, centre_dev, d, out_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

extern "C" int tph_weighted_moments_shifted(tph_ctx* ctx, const double* w_dev, int64_t n, const double* centre_dev,
                                            double* out_dev) {
  TPH_REQUIRE(ctx && w_dev && centre_dev && out_dev && n > 0 && n <= ctx->size, "tph_weighted_moments_shifted: bad argument");
  return moments_shifted(ctx, w_dev, n, centre_dev, out_dev, false);
}

// sum_s w_s^2 clip(d2_s - n_dim, +-1e6)^2 with d2 the Mahalanobis distance (tools.py:111-115).
// One lane per row, its centred coordinates in LDS as [d][64]; Sigma^-1 read with wave-uniform loads.
__global__ void __launch_bounds__(64) k_cv_sum(const double* __restrict__ hu, int64_t cap, int d, const double* __restrict__ w,
                                               int64_t n, const double* __restrict__ mean, const double* __restrict__ P,
                                               double* __restrict__ partials) {
  extern __shared__ double sh[];
  double* xc = sh + threadIdx.x;
  double acc = 0.0;
  const int64_t ntiles = (n + 63) / 64;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int64_t i = t * 64 + threadIdx.x;
    if (i < n) {
      for (int j = 0; j < d; ++j) xc[j * 64] = hu[(size_t)j * cap + i] - mean[j];
      double d2 = 0.0;
      for (int r = 0; r < d; ++r) {
        double s = 0.0;
        for (int j = 0; j < d; ++j) s += P[r * d + j] * xc[j * 64];
        d2 += xc[r * 64] * s;
      }
      double dev = fmin(fmax(d2 - (double)d, -1e6), 1e6);
      double ww = w[i];
      acc += (ww * ww) * (dev * dev);
    }
  }
  acc = tph_wave_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

extern "C" int tph_cv_sum(tph_ctx* ctx, const double* w_dev, int64_t n, const double* mean_dev, const double* covinv_dev,
                          double* out_dev) {
  TPH_REQUIRE(ctx && w_dev && mean_dev && covinv_dev && out_dev && n > 0 && n <= ctx->size, "tph_cv_sum: bad argument");
  const int d = ctx->d;
  int64_t ntiles = (n + 63) / 64;
  int nblk = (int)(ntiles < 4096 ? ntiles : 4096);
  if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)nblk)) return -1;
  double* part = (double*)ctx->scratch;
  size_t lds = sizeof(double) * (size_t)d * 64;
  if (lds > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_cv_sum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_cv_sum, dim3(nblk), dim3(64), lds, ctx->stream, ctx->u, ctx->cap, d, w_dev, n, mean_dev, covinv_dev, part);
  hipLaunchKernelGGL(k_colsum2, dim3(1), dim3(256), 0, ctx->stream, part, nblk, 1, out_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}


// ---- volume variation in ONE call, d x d work included (tools.py:58-117) -------------------------------------
// weighted mean / covariance of the history's u  ->  rank test, ridge, Cholesky and triangular inverse of the covariance on
// the device (one workgroup)  ->  sum_s w_s^2 clip(|L^-1 (u_s - mean)|^2 - d, +-1e6)^2 by the blocked triangular kernel
// ->  0.5 sqrt(sum / S0^2) delivered through the pinned mailbox.  One host wait per evaluation (the first version made two
// or three round trips for the host's matrix_rank / inv), and |L^-1 y|^2 costs half the FMAs of y^T Sigma^-1 y with the
// matrix operand in SGPRs (tri.h).  With a communicator attached the moments and the sum are all-reduced: the statistic of
// the GLOBAL weighted history.
//
// Rank test: numpy.linalg.matrix_rank(cov) < d (tools.py:102-104) compares singular values with s_max * d * eps.  Here: a
// diagonally pivoted Cholesky of the covariance; the rank is the number of pivots above d * eps * trace (for a positive
// semi-definite matrix s_max <= trace, and the pivots of the pivoted factorisation bound the trailing singular values), i.e.
// the same verdict on the degenerate ensembles the rule exists for.  Rank-deficient: cov += 1e-6 trace I as the reference.
// (the working copy, the factor and -- up to 96-D -- its inverse live in LDS, as in k_chol_inv: same operations, same order)
__global__ void __launch_bounds__(256) k_vv_prepare(double* __restrict__ cov, int d, double* __restrict__ Lg,
                                                    double* __restrict__ Wg, double* __restrict__ flag, int w_lds) {
  extern __shared__ double vl[];
  double* work = vl;                    // pivoted phase: the working copy; afterwards the same words hold L
  double* L = vl;
  double* W = w_lds ? vl + (size_t)d * d : Wg;
  __shared__ double s_val[256];
  __shared__ int s_idx[256];
  __shared__ int s_rank, s_fail;
  __shared__ double s_piv, s_tr;
  const int tid = threadIdx.x;
  for (int e = tid; e < d * d; e += 256) work[e] = cov[e];
  if (tid == 0) { s_rank = d; s_fail = 0; }
  __syncthreads();
  {
    double tr = 0.0;
    for (int j = tid; j < d; j += 256) tr += work[j * d + j];
    s_val[tid] = tr;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_val[tid] += s_val[tid + o]; __syncthreads(); }
    if (tid == 0) s_tr = s_val[0];
    __syncthreads();
  }
  const double tol = (double)d * DBL_EPSILON * fabs(s_tr);
  // ---- diagonally pivoted Cholesky on `work` (only its verdict is kept)
  for (int k = 0; k < d; ++k) {
    double best = -DBL_MAX; int bi = k;
    for (int i = k + tid; i < d; i += 256) { const double v = work[i * d + i]; if (v > best) { best = v; bi = i; } }
    s_val[tid] = best; s_idx[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o && (s_val[tid + o] > s_val[tid] || (s_val[tid + o] == s_val[tid] && s_idx[tid + o] < s_idx[tid]))) {
        s_val[tid] = s_val[tid + o]; s_idx[tid] = s_idx[tid + o];
      }
      __syncthreads();
    }
    const double pv = s_val[0]; const int p = s_idx[0];
    __syncthreads();
    if (!(pv > tol)) { if (tid == 0) s_rank = k; break; }      // also NaN
    if (p != k) {                                               // symmetric swap k <-> p
      for (int j = tid; j < d; j += 256) { const double t = work[k * d + j]; work[k * d + j] = work[p * d + j]; work[p * d + j] = t; }
      __syncthreads();
      for (int i = tid; i < d; i += 256) { const double t = work[i * d + k]; work[i * d + k] = work[i * d + p]; work[i * d + p] = t; }
      __syncthreads();
    }
    if (tid == 0) s_piv = sqrt(work[k * d + k]);
    __syncthreads();
    for (int i = k + 1 + tid; i < d; i += 256) work[i * d + k] /= s_piv;
    __syncthreads();
    const int m = d - k - 1;
    for (int e = tid; e < m * m; e += 256) {
      const int i = k + 1 + e / m, j = k + 1 + e % m;
      if (j <= i) { const double v = work[i * d + j] - work[i * d + k] * work[j * d + k]; work[i * d + j] = v; work[j * d + i] = v; }
    }
    __syncthreads();
  }
  __syncthreads();
  if (s_rank < d) {
    const double reg = 1e-6 * s_tr;
    for (int j = tid; j < d; j += 256) cov[j * d + j] += reg;
  }
  __syncthreads();
  // ---- plain Cholesky of the (possibly ridged) covariance, then W = L^-1
  for (int e = tid; e < d * d; e += 256) L[e] = 0.0;
  __syncthreads();
  for (int j = 0; j < d; ++j) {
    if (tid == 0) {
      double sd = cov[j * d + j];
      for (int k = 0; k < j; ++k) sd -= L[j * d + k] * L[j * d + k];
      if (!(sd > 0.0)) s_fail = 1;
      s_piv = sqrt(sd);
      L[j * d + j] = s_piv;
    }
    __syncthreads();
    if (s_fail) break;
    for (int i = j + 1 + tid; i < d; i += 256) {
      double sd = cov[i * d + j];
      for (int k = 0; k < j; ++k) sd -= L[i * d + k] * L[j * d + k];
      L[i * d + j] = sd / s_piv;
    }
    __syncthreads();
  }
  __syncthreads();
  if (!s_fail)
    for (int c = tid; c < d; c += 256) {
      for (int i = 0; i < d; ++i) {
        if (i < c) { W[i * d + c] = 0.0; continue; }
        double sd = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; ++k) sd -= L[i * d + k] * W[k * d + c];
        W[i * d + c] = sd / L[i * d + i];
      }
    }
  else
    for (int e = tid; e < d * d; e += 256) W[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < d * d; e += 256) Lg[e] = L[e];
  if (w_lds)
    for (int e = tid; e < d * d; e += 256) Wg[e] = W[e];
  if (tid == 0) { flag[0] = (double)s_fail; flag[1] = (double)s_rank; }
}

// sum_s w_s^2 clip(|W (u_s - mean)|^2 - d, +-1e6)^2 : 64 rows per workgroup tile, 4 waves share the tile's LDS columns and
// split the row chunks of W
__global__ void __launch_bounds__(256) k_cv_sum_blk(const double* __restrict__ hu, int64_t cap, int d, const double* __restrict__ w,
                                                    int64_t n, const double* __restrict__ mean, const double* __restrict__ Wb,
                                                    double* __restrict__ partials, const long long* __restrict__ seg = nullptr) {
  const int32_t* nolab_ = nullptr;
  SEG_SHIFT(hu, w, nolab_, n, seg);
  extern __shared__ double sh[];
  double* xs = sh;                           // [d][64]
  double* part = sh + (size_t)d * 64;        // [4][64]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: see k_propose_blk
  double acc = 0.0;
  const int64_t ntiles = (n + 63) / 64;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t i = t * 64 + lane;
    const double ww = i < n ? w[i] : 0.0;
    if (__ballot(ww != 0.0) == 0ull) continue;     // the four waves see the same 64 weights: uniform over the block
    __syncthreads();
    for (int j0 = wid; j0 < d; j0 += 32) {        // eight coordinates requested before the first is stored
      double v[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = j0 + 4 * a;
        v[a] = (j < d && i < n) ? hu[(size_t)j * cap + i] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = j0 + 4 * a;
        if (j < d) xs[(size_t)j * 64 + lane] = i < n ? v[a] - mean[j] : 0.0;
      }
    }
    __syncthreads();
    double d2 = 0.0;
    tri_apply(Wb, d, xs, lane, wid, 4, [&](int, double y) { d2 = fma(y, y, d2); });
    part[wid * 64 + lane] = d2;
    __syncthreads();
    if (wid == 0 && i < n) {
      const double tot = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
      const double dev = fmin(fmax(tot - (double)d, -1e6), 1e6);
      acc += (ww * ww) * (dev * dev);
    }
  }
  if (wid == 0) {
    acc = tph_wave_sum(acc);
    if (lane == 0) partials[(size_t)blockIdx.z * gridDim.x + blockIdx.x] = acc;
  }
}
// n_dim <= 12: one lane per row, the row in registers, W wave-uniform (scalar loads): d(d+1)/2 FMAs per 8d + 8 bytes -> HBM-bound
template <int D>
__global__ void __launch_bounds__(256) k_cv_sum_small(const double* __restrict__ hu, int64_t cap, const double* __restrict__ w,
                                                      int64_t n, const double* __restrict__ mean, const double* __restrict__ W,
                                                      double* __restrict__ partials, const long long* __restrict__ seg = nullptr) {
  const int32_t* nolab_ = nullptr;
  SEG_SHIFT(hu, w, nolab_, n, seg);
  double m[D];
#pragma unroll
  for (int j = 0; j < D; ++j) m[j] = mean[j];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double ww = w[i];
    // 64 consecutive rows without weight (whole early iterations, once exp(logw - max) underflows) add exactly zero: their
    // coordinates are not read (8 instead of 8 D + 8 bytes per row; same sum bit for bit).  Measured neutral without such rows.
    if (__ballot(ww != 0.0) == 0ull) continue;
    double xc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) xc[j] = hu[(size_t)j * cap + i] - m[j];
    double d2 = 0.0;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      double y = 0.0;
#pragma unroll
      for (int j = 0; j <= r; ++j) y = fma(W[r * D + j], xc[j], y);
      d2 = fma(y, y, d2);
    }
    const double dev = fmin(fmax(d2 - (double)D, -1e6), 1e6);
    acc += (ww * ww) * (dev * dev);
  }
  __shared__ double sh[4];
  acc = tph_block_sum(acc, sh);
  if (threadIdx.x == 0) partials[(size_t)blockIdx.z * gridDim.x + blockIdx.x] = acc;
}

// value = singular ? 1e10 : 0.5 sqrt(s / S0^2) -> pinned mailbox (value, then the sequence word)
__global__ void k_vv_finish(const double* __restrict__ s, const double* __restrict__ s0, const double* __restrict__ flag,
                            double* __restrict__ out_host, double* __restrict__ seq_host, double seq, double* __restrict__ out_dev) {
  if (threadIdx.x || blockIdx.x) return;
  const double S0 = s0[0];
  double v = 1e10;
  if (flag[0] == 0.0 && isfinite(S0) && S0 > 0.0) v = 0.5 * sqrt(s[0] / (S0 * S0));
  if (out_dev) out_dev[0] = v;
  out_host[0] = v;
  out_host[1] = flag[1];
  __threadfence_system();
  __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_vv_split(const double* __restrict__ mom, int d, double* __restrict__ s0, double* __restrict__ mean, double* __restrict__ cov) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;        // (S0, mean[d], cov[d*d]) -> separate buffers
  if (e == 0) s0[0] = mom[0];
  if (e < d) mean[e] = mom[1 + e];
  if (e < d * d) cov[e] = mom[1 + d + e];
}

// ---- the same statistic over the CANONICAL partition (common.h: tph_part): every sum per piece, the pieces of a virtual shard
// folded in iteration order, the shards in shard order (all-gathered first when sharded) -- one summation tree for any number
// of ranks, like the reweight triples, the proposal fit and the step sums.  Pieces as segments of the moment kernels.
struct vv_canon {
  tph_part part;
  int P, bps;                 // pieces of this rank | blocks per piece (from the piece's rows and V T: world-invariant)
  long long* seg;             // [P][2]
  double* segsums;            // [P][ncol_max]
  double* vs;                 // [vl + V][ncol_max]
  size_t c_off;               // staging offset of the gathers in ctx->comm_buf
};
// block partials [P][nblocks][ncol] -> out[ncol]
static int vv_reduce(tph_ctx* ctx, const vv_canon& c, const double* partials, int nblocks, int ncol, double* out) {
  const int vl = c.part.vl, T = c.part.T;
  const bool comm = ctx->comm_active();
  const int V = comm ? vl * ctx->world : vl;
  launch_colsum2(ctx, partials, nblocks, ncol, c.P, c.segsums);
  hipLaunchKernelGGL(k_fold_cols, dim3((ncol + 255) / 256, vl), dim3(256), 0, ctx->stream, (const double*)c.segsums, T, ncol, c.vs);
  TPH_LAUNCH_CHECK();
  const double* rows = c.vs;
  if (comm) {
    const size_t one = sizeof(double) * (size_t)vl * ncol, all_off = c.c_off + (one + 255) / 256 * 256;
    if (tph_comm_require(ctx, all_off + one * ctx->world, "tph_volume_variation")) return -2;
    TPH_HIP(hipMemcpyAsync(ctx->comm_buf + c.c_off, c.vs, one, hipMemcpyDeviceToDevice, ctx->stream));
    if (tph_comm_allgather(ctx, c.c_off, all_off, (int64_t)vl * ncol, TPH_DT_F64)) return -2;
    rows = (const double*)(ctx->comm_buf + all_off);
  }
  hipLaunchKernelGGL(k_fold_cols, dim3((ncol + 255) / 256, 1), dim3(256), 0, ctx->stream, rows, V, ncol, out);
  TPH_LAUNCH_CHECK();
  return 0;
}

extern "C" int tph_volume_variation(tph_ctx* ctx, const double* w_dev, int64_t n, double* centre_dev, double* value_host) {
  TPH_REQUIRE(ctx && w_dev && value_host && n > 0 && n <= ctx->size, "tph_volume_variation: bad argument");
  const int d = ctx->d;
  TPH_REQUIRE(d <= 100, "tph_volume_variation: n_dim=%d > 100", d);
  const bool comm = ctx->comm_active();
  // persistent small buffers behind small_dev: s0 | mean[d] | flag[2] | s | then d x d matrices in the winv-style block
  const size_t mat = (size_t)d * d;
  const size_t need_small = sizeof(double) * (8 + d + 4 * mat + tri_blocked_doubles(d) + (1 + d + mat));
  if (ctx->vv_bytes < need_small) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->vv_buf) TPH_HIP(hipFree(ctx->vv_buf));
    ctx->vv_buf = nullptr; ctx->vv_bytes = 0;
    TPH_HIP(hipMalloc((void**)&ctx->vv_buf, need_small));
    ctx->vv_bytes = need_small;
  }
  double* s0 = ctx->vv_buf;
  double* flag = s0 + 1;
  double* ssum = s0 + 3;
  double* mean = s0 + 8;
  double* cov = mean + d;
  double* L = cov + 2 * mat;            // (cov + mat: free -- the pivoted factorisation's working copy lives in LDS)
  double* W = L + mat;
  double* Wb = W + mat;
  double* mom = Wb + tri_blocked_doubles(d);
  // ---- the canonical partition of the history (sums per piece, folded per shard, then across shards), when it has one
  vv_canon cn;
  cn.part = tph_partition(ctx, n);
  const bool canon = cn.part.canonical || (comm && cn.part.T * cn.part.vl > 1);
  const int npl_ = d * (d + 1) / 2, ncmax = npl_ + d + 1;
  double* cpart = nullptr;              // block partials of the canonical passes
  if (canon) {
    cn.P = cn.part.T * cn.part.vl;
    const int Vt = cn.part.canonical ? cn.part.V : ctx->world;
    // blocks per piece: >= 1024 rows each, <= 4096 blocks in all (one block per 4096-row piece left a 65 536-particle history of
    // 46 iterations with 736 blocks -- three per CU -- and the covariance pass at 592 us against 388 us before the partition)
    long long b = cn.part.nv / 1024;
    const long long bcap = 4096 / ((long long)Vt * cn.part.T) > 1 ? 4096 / ((long long)Vt * cn.part.T) : 1;
    cn.bps = (int)(b < 1 ? 1 : (b > bcap ? bcap : b));
    const int rp = d <= 12 ? 1 : (d >= 16 ? 1 : cov_slices(npl_));
    size_t per_block = (size_t)(d <= 12 ? ncmax : (rp * npl_ > (1 + d) * 3 ? rp * npl_ : (1 + d) * 3));
    size_t o2 = 0;
    auto take2 = [&](size_t bytes) { size_t r = o2; o2 += (bytes + 255) / 256 * 256; return r; };
    const size_t o_seg = take2(sizeof(long long) * 2 * (size_t)cn.P);
    const size_t o_ss = take2(sizeof(double) * (size_t)cn.P * ncmax);
    const size_t o_vs2 = take2(sizeof(double) * (size_t)(cn.part.vl + Vt) * ncmax);
    const size_t o_cp = take2(sizeof(double) * (size_t)cn.P * cn.bps * per_block);
    const size_t o_small = take2(sizeof(double) * (size_t)(2 * ncmax + 8));
    if (tph_scratch_reserve(ctx, o2)) return -1;
    char* sb = (char*)ctx->scratch;
    cn.seg = (long long*)(sb + o_seg); cn.segsums = (double*)(sb + o_ss); cn.vs = (double*)(sb + o_vs2); cpart = (double*)(sb + o_cp);
    cn.c_off = 0;
    double* csum_c = (double*)(sb + o_small);                 // [ncmax]
    double* sums_c = csum_c + ncmax;                          // [1 + d]
    hipLaunchKernelGGL(k_piece_segments, dim3((cn.P + 255) / 256), dim3(256), 0, ctx->stream, cn.part.T, cn.part.vl, (long long)cn.part.n_loc,
                       (long long)cn.part.nv, cn.seg);
    if (d <= 12 && centre_dev) {
      switch (d) {
#define C(DD) case DD: hipLaunchKernelGGL((k_wmom_small<DD>), dim3(cn.bps, 1, cn.P), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, w_dev, n, centre_dev, cpart, (const long long*)cn.seg); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
      }
      TPH_LAUNCH_CHECK();
      if (vv_reduce(ctx, cn, cpart, cn.bps, ncmax, csum_c)) return -2;
      hipLaunchKernelGGL(k_wmom_finish, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, csum_c, centre_dev, d, mom);
      hipLaunchKernelGGL(k_vv_split, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, mom, d, s0, mean, cov);
      TPH_HIP(hipMemcpyAsync(centre_dev, mean, sizeof(double) * d, hipMemcpyDeviceToDevice, ctx->stream));   // next call's centre
    } else {
      hipLaunchKernelGGL(k_wsum<double>, dim3(cn.bps, 1 + d, cn.P), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, d, w_dev, (const int32_t*)nullptr, 0,
                         n, cpart, (const long long*)cn.seg);
      hipLaunchKernelGGL(k_wsum_final, dim3(1 + d, cn.P), dim3(256), 0, ctx->stream, cpart, cn.bps, 1 + d, cn.segsums, (double*)nullptr);
      hipLaunchKernelGGL(k_fold_cols, dim3((1 + d + 255) / 256, cn.part.vl), dim3(256), 0, ctx->stream, (const double*)cn.segsums, cn.part.T, 1 + d, cn.vs);
      TPH_LAUNCH_CHECK();
      {
        const int vl = cn.part.vl, ncol = 1 + d, Vv = comm ? vl * ctx->world : vl;
        const double* rows = cn.vs;
        if (comm) {
          const size_t one = sizeof(double) * (size_t)vl * ncol, all_off = (one + 255) / 256 * 256;
          if (tph_comm_require(ctx, all_off + one * ctx->world, "tph_volume_variation")) return -2;
          TPH_HIP(hipMemcpyAsync(ctx->comm_buf, cn.vs, one, hipMemcpyDeviceToDevice, ctx->stream));
          if (tph_comm_allgather(ctx, 0, all_off, (int64_t)vl * ncol, TPH_DT_F64)) return -2;
          rows = (const double*)(ctx->comm_buf + all_off);
        }
        hipLaunchKernelGGL(k_fold_cols, dim3((ncol + 255) / 256, 1), dim3(256), 0, ctx->stream, rows, Vv, ncol, sums_c);
      }
      hipLaunchKernelGGL(k_mean_from_sums, dim3((d + 63) / 64), dim3(64), 0, ctx->stream, (const double*)sums_c, d, mean);
      TPH_HIP(hipMemcpyAsync(s0, sums_c, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      int rows_pb = 1;
      if (d <= 12) {
        bool ok = launch_wcov_small<double>(ctx, ctx->u, ctx->cap, w_dev, nullptr, 0, n, mean, cpart, cn.bps, cn.seg, cn.P);
        TPH_REQUIRE(ok, "covariance: no register kernel for n_dim=%d", d);
      } else {
        if (launch_wcov<double>(ctx, ctx->u, ctx->cap, w_dev, (const int32_t*)nullptr, 0, n, mean, cpart, cn.bps, &rows_pb, cn.seg, cn.P)) return -1;
      }
      TPH_LAUNCH_CHECK();
      if (vv_reduce(ctx, cn, cpart, cn.bps * rows_pb, npl_, csum_c)) return -2;
      hipLaunchKernelGGL(k_cov_finish, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, (const double*)csum_c, (const double*)sums_c, d, 2, cov);
      TPH_LAUNCH_CHECK();
      if (centre_dev) TPH_HIP(hipMemcpyAsync(centre_dev, mean, sizeof(double) * d, hipMemcpyDeviceToDevice, ctx->stream));
    }
  } else
  // ---- moments
  if (d <= 12 && centre_dev) {
    if (moments_shifted(ctx, w_dev, n, centre_dev, mom, comm)) return -1;
    hipLaunchKernelGGL(k_vv_split, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, mom, d, s0, mean, cov);
    TPH_HIP(hipMemcpyAsync(centre_dev, mean, sizeof(double) * d, hipMemcpyDeviceToDevice, ctx->stream));   // next call's centre
  } else {
    const int nblk = cov_blocks(n);
    const int rblk = tph_grid_for(n, 256, 4, 512);
    const int npl = d * (d + 1) / 2;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    size_t o_part = take(sizeof(double) * cov_scratch_doubles(d, nblk));
    size_t o_part1 = take(sizeof(double) * (size_t)rblk * (1 + d) * 3);
    size_t o_sums = take(sizeof(double) * (1 + d));
    size_t o_csum = take(sizeof(double) * npl);
    if (tph_scratch_reserve(ctx, o)) return -1;
    char* base = (char*)ctx->scratch;
    double* part = (double*)(base + o_part);
    double* part1 = (double*)(base + o_part1);
    double* sums = comm ? (double*)ctx->comm_buf : (double*)(base + o_sums);
    double* csum = comm ? (double*)(ctx->comm_buf + 4096) : (double*)(base + o_csum);
    if (comm && tph_comm_require(ctx, 4096 + sizeof(double) * npl, "tph_volume_variation")) return -2;
    hipLaunchKernelGGL(k_wsum<double>, dim3(rblk, 1 + d), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, d, w_dev,
                       (const int32_t*)nullptr, 0, n, part1);
    hipLaunchKernelGGL(k_wsum_final, dim3(1 + d), dim3(256), 0, ctx->stream, part1, rblk, 1 + d, sums, (double*)nullptr);
    TPH_LAUNCH_CHECK();
    if (comm && tph_comm_allreduce(ctx, 0, 1 + d, TPH_DT_F64, TPH_OP_SUM)) return -2;
    hipLaunchKernelGGL(k_mean_from_sums, dim3((d + 63) / 64), dim3(64), 0, ctx->stream, sums, d, mean);
    TPH_HIP(hipMemcpyAsync(s0, sums, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    const int S = cov_slices(npl);
    if (d <= 12) {
      bool ok = launch_wcov_small<double>(ctx, ctx->u, ctx->cap, w_dev, nullptr, 0, n, mean, part, nblk);
      TPH_REQUIRE(ok, "covariance: no register kernel for n_dim=%d", d);
      launch_colsum2(ctx, part, nblk, npl, 1, csum);
    } else {
      int rpb = S;
      if (launch_wcov<double>(ctx, ctx->u, ctx->cap, w_dev, (const int32_t*)nullptr, 0, n, mean, part, nblk, &rpb)) return -1;
      launch_colsum2(ctx, part, nblk * rpb, npl, 1, csum);
    }
    TPH_LAUNCH_CHECK();
    if (comm && tph_comm_allreduce(ctx, 4096, npl, TPH_DT_F64, TPH_OP_SUM)) return -2;
    hipLaunchKernelGGL(k_cov_finish, dim3((d * d + 255) / 256), dim3(256), 0, ctx->stream, csum, sums, d, 2, cov);
    TPH_LAUNCH_CHECK();
    if (centre_dev) TPH_HIP(hipMemcpyAsync(centre_dev, mean, sizeof(double) * d, hipMemcpyDeviceToDevice, ctx->stream));
  }
  // ---- d x d work on the device, blocked layout of L^-1
  const int w_lds = 2 * sizeof(double) * (size_t)d * d <= 150 * 1024 ? 1 : 0;
  const size_t lds_vv = sizeof(double) * (size_t)d * d * (w_lds ? 2 : 1);
  if (lds_vv > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_vv_prepare, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_vv));
  hipLaunchKernelGGL(k_vv_prepare, dim3(1), dim3(256), lds_vv, ctx->stream, cov, d, L, W, flag, w_lds);
  hipLaunchKernelGGL(k_tri_block, dim3(1), dim3(256), 0, ctx->stream, W, d, Wb);
  // ---- the statistic
  const int64_t ntiles = (n + 63) / 64;
  int nb;
  double* partials;
  double* sdst = comm ? (double*)ctx->comm_buf : ssum;
  if (canon) {
    if (d <= 12) {
      switch (d) {
#define C(DD) case DD: hipLaunchKernelGGL((k_cv_sum_small<DD>), dim3(cn.bps, 1, cn.P), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, w_dev, n, mean, W, cpart, (const long long*)cn.seg); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
      }
    } else {
      const size_t lds = sizeof(double) * ((size_t)d * 64 + 256);
      if (lds > 64 * 1024)
        TPH_HIP(hipFuncSetAttribute((const void*)k_cv_sum_blk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_cv_sum_blk, dim3(cn.bps, 1, cn.P), dim3(256), lds, ctx->stream, ctx->u, ctx->cap, d, w_dev, n, mean, Wb, cpart,
                         (const long long*)cn.seg);
    }
    TPH_LAUNCH_CHECK();
    sdst = ssum;
    cn.c_off = 4096;
    if (vv_reduce(ctx, cn, cpart, cn.bps, 1, ssum)) return -2;
  } else {
  if (d <= 12) {
    nb = tph_grid_for(n, 256, 8, 2048);
    if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)nb)) return -1;
    partials = (double*)ctx->scratch;
    switch (d) {
#define C(DD) case DD: hipLaunchKernelGGL((k_cv_sum_small<DD>), dim3(nb), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, w_dev, n, mean, W, partials); break;
      C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
    }
  } else {
    nb = (int)(ntiles < 2048 ? ntiles : 2048);
    if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)nb)) return -1;
    partials = (double*)ctx->scratch;
    const size_t lds = sizeof(double) * ((size_t)d * 64 + 256);
    if (lds > 64 * 1024)
      TPH_HIP(hipFuncSetAttribute((const void*)k_cv_sum_blk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_cv_sum_blk, dim3(nb), dim3(256), lds, ctx->stream, ctx->u, ctx->cap, d, w_dev, n, mean, Wb, partials);
  }
  hipLaunchKernelGGL(k_colsum2, dim3(1), dim3(256), 0, ctx->stream, partials, nb, 1, sdst);
  TPH_LAUNCH_CHECK();
  if (comm && tph_comm_allreduce(ctx, 0, 1, TPH_DT_F64, TPH_OP_SUM)) return -2;
  }
  // ---- result through the pinned mailbox: [64] value, [65] rank, [4094] sequence word
  volatile double* seqp = ctx->pinned + 4094;
  const double seq = (double)(++ctx->vv_seq);
  hipLaunchKernelGGL(k_vv_finish, dim3(1), dim3(1), 0, ctx->stream, sdst, s0, flag, ctx->pinned + 64, ctx->pinned + 4094, seq,
                     (double*)nullptr);
  TPH_LAUNCH_CHECK();
  uint64_t spins = 0;
  while (*seqp != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFF) == 0) {
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipErrorNotReady) {
        TPH_HIP(hipStreamSynchronize(ctx->stream));
        TPH_REQUIRE(*seqp == seq, "tph_volume_variation: the device never delivered evaluation %.0f", seq);
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  *value_host = ctx->pinned[64];
  return 0;
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_modes(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
