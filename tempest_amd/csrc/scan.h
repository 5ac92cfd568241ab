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
template <int MODE>
__device__ __forceinline__ double xform(double v, double thr) {
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

// The value `CTRL` lanes away through the data-parallel-primitive path of the vector ALU (no LDS crossbar); lanes the pattern
// or the masks leave out read 0.0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_or_zero(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
// Inclusive prefix sum over the 64 lanes in 7 DPP steps: three shifts inside groups of 4, then 4 and 8 inside a row of 16,
// then lane 15 / lane 31 broadcast into the following rows.
__device__ __forceinline__ double wave_incl_dpp(double v) {
  double s = v + dpp_or_zero<0x111, 0xf, 0xf>(v);      // row_shr:1
  s += dpp_or_zero<0x112, 0xf, 0xf>(v);                // row_shr:2
  s += dpp_or_zero<0x113, 0xf, 0xf>(v);                // row_shr:3
  s += dpp_or_zero<0x114, 0xf, 0xe>(s);                // row_shr:4, lanes 4..15 of a row
  s += dpp_or_zero<0x118, 0xf, 0xc>(s);                // row_shr:8, lanes 8..15 of a row
  s += dpp_or_zero<0x142, 0xa, 0xf>(s);                // row_bcast:15 into rows 1 and 3
  s += dpp_or_zero<0x143, 0xc, 0xf>(s);                // row_bcast:31 into rows 2 and 3
  return s;
}
__device__ __forceinline__ double wave_prev(double v) { return dpp_or_zero<0x138, 0xf, 0xf>(v); }   // wave_shr:1, lane 0 reads 0.0
__device__ __forceinline__ double wave_last(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Tile layout in registers: wave `wid` of the block owns values [512 wid, 512 wid + 512) of the tile as 4 segments of 128, and
// lane l holds values 2l and 2l + 1 of each segment -- every load and store instruction of a wave is one contiguous 1 KB.
// `base` = first row of the tile, `lim` = one past the last valid row.
template <int MODE>
__device__ __forceinline__ void load_tile(const double* __restrict__ w, int64_t base, int64_t lim, double thr, double (&x)[ITEMS]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t p0 = base + wid * 512 + 2 * lane;
  if (base + TILE <= lim && (((uintptr_t)(w + base)) & 15) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double2 v = *reinterpret_cast<const double2*>(w + p0 + j * 128);
      x[2 * j] = xform<MODE>(v.x, thr);
      x[2 * j + 1] = xform<MODE>(v.y, thr);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = p0 + j * 128;
      x[2 * j] = p < lim ? xform<MODE>(w[p], thr) : 0.0;
      x[2 * j + 1] = p + 1 < lim ? xform<MODE>(w[p + 1], thr) : 0.0;
    }
  }
}

// Inclusive prefixes of the tile in the same layout, `off` (the sum of everything before the tile) included.  Exclusive
// prefixes come from shifting the inclusive ones, never from `inclusive - own` (see the accuracy note above).
__device__ __forceinline__ void scan_tile(double (&x)[ITEMS], double off, double* __restrict__ wsum) {
  const int wid = threadIdx.x >> 6;
  double b[4], ex[4], tot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = x[2 * j] + x[2 * j + 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double inc = wave_incl_dpp(b[j]);
    ex[j] = wave_prev(inc);
    tot[j] = wave_last(inc);
  }
  if ((threadIdx.x & 63) == 0) wsum[wid] = (tot[0] + tot[1]) + (tot[2] + tot[3]);
  __syncthreads();
  for (int k = 0; k < wid; ++k) off += wsum[k];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double o = off + ex[j];
    x[2 * j] = o + x[2 * j];
    x[2 * j + 1] = o + b[j];
    off += tot[j];
  }
}

template <int MODE>
__global__ void __launch_bounds__(THREADS) k_tile_sums(const double* __restrict__ w, int64_t n,
                                                       const double* __restrict__ thr_dev, double* __restrict__ tiles) {
  const double thr = MODE == MASKED ? thr_dev[0] : 0.0;
  double x[ITEMS];
  load_tile<MODE>(w, (int64_t)blockIdx.x * TILE, n, thr, x);
  double s = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
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

// stores the tile (layout of load_tile); `fix(value, row)` post-processes a value on its way out
template <typename F>
__device__ __forceinline__ void store_tile(double* __restrict__ out, int64_t base, int64_t lim, const double (&x)[ITEMS], F fix) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t p0 = base + wid * 512 + 2 * lane;
  if (base + TILE <= lim && (((uintptr_t)(out + base)) & 15) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = p0 + j * 128;
      *reinterpret_cast<double2*>(out + p) = make_double2(fix(x[2 * j], p), fix(x[2 * j + 1], p + 1));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = p0 + j * 128;
      if (p < lim) out[p] = fix(x[2 * j], p);
      if (p + 1 < lim) out[p + 1] = fix(x[2 * j + 1], p + 1);
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(THREADS) k_apply(const double* __restrict__ w, int64_t n,
                                                   const double* __restrict__ thr_dev, const double* __restrict__ tiles,
                                                   double* __restrict__ out) {
  const double thr = MODE == MASKED ? thr_dev[0] : 0.0;
  const int64_t base = (int64_t)blockIdx.x * TILE;
  double x[ITEMS];
  load_tile<MODE>(w, base, n, thr, x);
  __shared__ double wsum[THREADS / 64];
  scan_tile(x, tiles[blockIdx.x], wsum);
  store_tile(out, base, n, x, [](double v, int64_t) { return v; });
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
