// Three-pass inclusive prefix sum of FP64 values (tile sums -> exclusive scan of the tile sums -> local scan +
// offset), shared by resampling (cdf of the weights, optionally masked by a device-side threshold) and by the
// trimming step (prefix sums of the sorted weights and of their squares).
//
// Accuracy note: exclusive prefixes are obtained by shuffling the inclusive ones, never as `inclusive - own`:
// importance weights span tens of orders of magnitude and that subtraction would wipe out a small prefix in front
// of a dominant weight.  Neighbouring outputs still come from different summation trees, so the result is monotone
// only up to rounding (<= 1 ulp dips where a weight is below the running sum's ulp); the binary searches tolerate it.
#pragma once
#include "common.h"

namespace tph_scan {

constexpr int THREADS = 256;
constexpr int ITEMS = 8;
constexpr int TILE = THREADS * ITEMS;

enum Mode { PLAIN = 0, MASKED = 1, SQUARE = 2 };

template <int MODE>
__device__ __forceinline__ double load(const double* __restrict__ w, int64_t i, double thr) {
  double v = w[i];
  if (MODE == MASKED) return (v >= thr) ? v : 0.0;
  if (MODE == SQUARE) return v * v;
  return v;
}

__device__ __forceinline__ double wave_incl(double v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

template <int MODE>
__global__ void __launch_bounds__(THREADS) k_tile_sums(const double* __restrict__ w, int64_t n,
                                                       const double* __restrict__ thr_dev, double* __restrict__ tiles) {
  const double thr = MODE == MASKED ? thr_dev[0] : 0.0;
  const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * ITEMS;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (base + k < n) s += load<MODE>(w, base + k, thr);
  __shared__ double sh[THREADS / 64];
  s = tph_block_sum(s, sh);
  if (threadIdx.x == 0) tiles[blockIdx.x] = s;
}

// exclusive scan of the tile sums in place: one block of 1024 threads, each owning a contiguous run
static __global__ void __launch_bounds__(1024) k_tile_offsets(double* __restrict__ tiles, int64_t ntiles) {
  const int64_t per = (ntiles + 1023) / 1024;
  const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
  double s = 0.0;
  for (int64_t i = lo; i < hi; ++i) s += tiles[i];
  __shared__ double wsum[16];
  const double inc = wave_incl(s);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    double v = lane < 16 ? wsum[lane] : 0.0;
    v = wave_incl(v);
    if (lane < 16) wsum[lane] = v;
  }
  __syncthreads();
  const double prev = __shfl_up(inc, 1, 64);
  double excl = (lane > 0 ? prev : 0.0) + (wid > 0 ? wsum[wid - 1] : 0.0);
  for (int64_t i = lo; i < hi; ++i) {
    const double t = tiles[i];
    tiles[i] = excl;
    excl += t;
  }
}

template <int MODE>
__global__ void __launch_bounds__(THREADS) k_apply(const double* __restrict__ w, int64_t n,
                                                   const double* __restrict__ thr_dev, const double* __restrict__ tiles,
                                                   double* __restrict__ out) {
  const double thr = MODE == MASKED ? thr_dev[0] : 0.0;
  const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * ITEMS;
  double v[ITEMS];
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    s += base + k < n ? load<MODE>(w, base + k, thr) : 0.0;
    v[k] = s;
  }
  __shared__ double wsum[THREADS / 64];
  const double inc = wave_incl(s);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  double off = tiles[blockIdx.x];
  for (int k = 0; k < wid; ++k) off += wsum[k];
  const double prev = __shfl_up(inc, 1, 64);
  if (lane > 0) off += prev;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (base + k < n) out[base + k] = off + v[k];
}

static inline int64_t num_tiles(int64_t n) { return (n + TILE - 1) / TILE; }

// `tiles` must hold num_tiles(n) doubles of scratch
template <int MODE>
static int inclusive(tph_ctx* ctx, const double* in, int64_t n, const double* thr_dev, double* tiles, double* out) {
  const int64_t nt = num_tiles(n);
  hipLaunchKernelGGL(k_tile_sums<MODE>, dim3((unsigned)nt), dim3(THREADS), 0, ctx->stream, in, n, thr_dev, tiles);
  hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(1024), 0, ctx->stream, tiles, nt);
  hipLaunchKernelGGL(k_apply<MODE>, dim3((unsigned)nt), dim3(THREADS), 0, ctx->stream, in, n, thr_dev, tiles, out);
  TPH_LAUNCH_CHECK();
  return 0;
}

}  // namespace tph_scan
