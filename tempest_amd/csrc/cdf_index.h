// Cache-line-fanout index over a cumulative-weight array (gfx950).
//
// Inverse-CDF lookups (np.random.choice(p=w), tempest/steps/resample.py:80-84; the x4 multinomial up-sampling of
// tempest/modes.py:196-201) are binary searches over a cdf of N_h doubles.  At 4x10^7 rows a plain search misses in
// DRAM ~15 times per draw (measured with FETCH_SIZE: 1 KB fetched per draw).  Level l+1 of this index keeps the LAST
// element of every group of 16 consecutive elements of level l (one 128-byte line per group); the top level (<= 2048
// entries) stays cached, and each level below costs one line: ~3 misses per draw.  The predicate is evaluated on the same
// values with the same arithmetic as the plain search, so the result is identical.
#pragma once
#include "common.h"

constexpr int TPH_CDFI_FAN = 16;
constexpr int TPH_CDFI_MAXLVL = 7;        // 16^6 * 2048 rows
constexpr int64_t TPH_CDFI_TOP = 2048;

struct tph_cdf_index {
  const double* lvl[TPH_CDFI_MAXLVL];     // lvl[0] = the cdf itself
  int64_t n[TPH_CDFI_MAXLVL];
  int levels;
};

static inline size_t tph_cdf_index_doubles(int64_t n) {
  size_t tot = 0;
  while (n > TPH_CDFI_TOP) { n = (n + TPH_CDFI_FAN - 1) / TPH_CDFI_FAN; tot += (size_t)n; }
  return tot;
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(256) k_cdf_coarsen(const double* __restrict__ in, int64_t n_in, double* __restrict__ out,
                                                     int64_t n_out) {
  int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_out) return;
  int64_t j = k * TPH_CDFI_FAN + (TPH_CDFI_FAN - 1);
  out[k] = in[j < n_in ? j : n_in - 1];
}

// builds the coarse levels into `work` (tph_cdf_index_doubles(n) doubles) on the ctx stream
static inline int tph_cdf_index_build(tph_ctx* ctx, const double* cdf, int64_t n, double* work, tph_cdf_index* ix) {
  ix->lvl[0] = cdf;
  ix->n[0] = n;
  ix->levels = 1;
  while (ix->n[ix->levels - 1] > TPH_CDFI_TOP && ix->levels < TPH_CDFI_MAXLVL) {
    const int l = ix->levels;
    const int64_t n_in = ix->n[l - 1], n_out = (n_in + TPH_CDFI_FAN - 1) / TPH_CDFI_FAN;
    hipLaunchKernelGGL(k_cdf_coarsen, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx->stream, ix->lvl[l - 1], n_in,
                       work, n_out);
    ix->lvl[l] = work;
    ix->n[l] = n_out;
    work += n_out;
    ix->levels = l + 1;
  }
  TPH_LAUNCH_CHECK();
  return 0;
}

// #{k in [0,n) : pred(cdf_k)}, pred(c) = c/div < pos (STRICT) or <= pos: monotone (true ... true false ... false)
template <bool STRICT>
__device__ __forceinline__ int64_t tph_count_below(const tph_cdf_index& ix, double div, double pos) {
  auto below = [&](double c) { c /= div; return STRICT ? (c < pos) : (c <= pos); };
  const int top = ix.levels - 1;
  int64_t lo = 0, hi = ix.n[top];
  {
    const double* __restrict__ a = ix.lvl[top];
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (below(a[mid])) lo = mid + 1; else hi = mid;
    }
  }
  int64_t t = lo;                          // groups 0 .. t-1 of the level below are entirely `true`
  for (int l = top - 1; l >= 0; --l) {
    const double* __restrict__ a = ix.lvl[l];
    lo = t * TPH_CDFI_FAN;
    if (lo >= ix.n[l]) { t = ix.n[l]; continue; }
    hi = lo + TPH_CDFI_FAN < ix.n[l] ? lo + TPH_CDFI_FAN : ix.n[l];
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (below(a[mid])) lo = mid + 1; else hi = mid;
    }
    t = lo;
  }
  return t;
}
#endif
