// Blocked lower-triangular matrix times a tile of 64 column vectors, the FP64 contraction of the d > 16 paths (gfx950).
//
//   y_r = sum_{j <= r} T[r][j] x_j      for 64 vectors at once, lane = vector
//
// is what the proposal (L z), both Mahalanobis forms (|L^-1 (v - mu)|^2) and the volume-variation statistic
// (|L_cov^-1 (u - mean)|^2) spend their time in once a step is one attempt.  The first version of those kernels read BOTH
// operands of every FMA from LDS (a lane group per vector, rows dealt to lanes): ~2 TFLOP/s.  Here a lane owns a vector,
// so the matrix element is WAVE-UNIFORM and comes through the scalar cache into an SGPR operand; the vector elements sit
// in LDS as conflict-free columns xs[j * 64 + lane]; 8 consecutive rows are accumulated together so that each x_j read
// feeds 8 FMAs, and the matrix is stored in the matching layout  Tb[c][j][q] = T[8 c + q][j]  (one 64-byte scalar load per
// 8 FMAs).  Rows above the diagonal inside a chunk are zeros of T: multiplied, not branched around.
#pragma once
#include "common.h"

constexpr int TRI_RB = 8;      // rows per chunk

__host__ __device__ static inline size_t tri_blocked_doubles(int d) { return (size_t)((d + TRI_RB - 1) / TRI_RB) * d * TRI_RB; }

#if defined(__HIPCC__)
// Tb[k][c][j][q] = T[k][8c+q][j] (0 beyond the matrix); one workgroup per matrix
static __global__ void __launch_bounds__(256) k_tri_block(const double* __restrict__ T, int d, double* __restrict__ Tb) {
  const int nch = (d + TRI_RB - 1) / TRI_RB;
  const double* A = T + (size_t)blockIdx.x * d * d;
  double* B = Tb + (size_t)blockIdx.x * tri_blocked_doubles(d);
  for (int e = threadIdx.x; e < nch * d * TRI_RB; e += blockDim.x) {
    const int q = e % TRI_RB, j = (e / TRI_RB) % d, c = e / (TRI_RB * d);
    const int r = c * TRI_RB + q;
    B[e] = (r < d && j <= r) ? A[(size_t)r * d + j] : 0.0;
  }
}

// the chunks of  y = T x  that fall to wave c0 of cstep; f(r, y_r) is called for every row r < d of those chunks.
// Four columns per trip: their 4 x 64 B of matrix come as back-to-back scalar loads behind ONE wait (a scalar load that
// is waited for alone costs its full ~200-cycle latency per 8 FMAs), then 32 FMAs.
template <class F>
__device__ __forceinline__ void tri_apply(const double* __restrict__ Tb, int d, const double* __restrict__ xs, int lane, int c0,
                                          int cstep, F&& f) {
  const int nch = (d + TRI_RB - 1) / TRI_RB;
  constexpr int JU = 4;
  // chunk c costs ~(c + 1) columns: the chunks are dealt from the LAST one down, boustrophedon over the waves (wave c0 of
  // cstep takes positions c0, 2 cstep - 1 - c0, 2 cstep + c0, ... of the descending order), so the waves of a workgroup finish
  // together (d = 50 on 4 waves: 7 + 7 + 7 + 7 columns-of-8 instead of 6 + 8 + 10 + 4)
  for (int k = 0;; ++k) {
    const int pos = k * cstep + ((k & 1) ? cstep - 1 - c0 : c0);
    if (pos >= nch) break;
    const int c = nch - 1 - pos;
    const double* __restrict__ Tc = Tb + (size_t)c * d * TRI_RB;
    const int jmax = (c + 1) * TRI_RB < d ? (c + 1) * TRI_RB : d;
    double acc[TRI_RB];
#pragma unroll
    for (int q = 0; q < TRI_RB; ++q) acc[q] = 0.0;
    int j = 0;
    for (; j + JU <= jmax; j += JU) {
      double t[JU][TRI_RB], x[JU];
#pragma unroll
      for (int a = 0; a < JU; ++a) {
#pragma unroll
        for (int q = 0; q < TRI_RB; ++q) t[a][q] = Tc[(size_t)(j + a) * TRI_RB + q];
        x[a] = xs[(size_t)(j + a) * 64 + lane];
      }
#pragma unroll
      for (int a = 0; a < JU; ++a)
#pragma unroll
        for (int q = 0; q < TRI_RB; ++q) acc[q] = fma(t[a][q], x[a], acc[q]);
    }
    for (; j < jmax; ++j) {
      const double x0 = xs[(size_t)j * 64 + lane];
#pragma unroll
      for (int q = 0; q < TRI_RB; ++q) acc[q] = fma(Tc[(size_t)j * TRI_RB + q], x0, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < TRI_RB; ++q)
      if (c * TRI_RB + q < d) f(c * TRI_RB + q, acc[q]);
  }
}
#endif
