// Device pieces of the weighted Gaussian-mixture clustering (reference: tempest/cluster.py).
// E-step / BIC log-likelihood / label prediction over a compact SoA working set, compaction of the
// trimmed history rows, and the M-step sums on an explicit SoA pointer.  The EM control loop, the BIC
// split search and the d x d inverses stay on the host exactly as the reference structures them
// (cluster.py:56-133,420-520).
#include "common.h"

// params per component k: [0] log-weight term, [1..d] mean, [1+d .. 1+d+d*d) precision P, [1+d+d*d] logdet
__device__ __forceinline__ size_t gmm_stride(int d) { return (size_t)2 + d + (size_t)d * d; }

constexpr int GMM_THREADS = 64;
constexpr int GMM_KMAX_RESP = 8;

// One lane per row; the row sits in LDS as [d][64].  mode 0: weighted responsibilities + log-likelihood
// sums (cluster.py:178-198,287-304), mode 1: sw * min_k |x-mu_k|^2_P (k-means++ seeding, :146-157),
// mode 2: argmax_k log(w_k + 1e-10) + logN_k (:306-328, :600-696).
__global__ void __launch_bounds__(GMM_THREADS) k_gmm_estep(const double* __restrict__ x, int64_t ld, int64_t n, int d,
                                                           const double* __restrict__ sw, const int32_t* __restrict__ labels,
                                                           int label, int K, const double* __restrict__ params, int mode,
                                                           double eps, const double* __restrict__ shift,
                                                           const double* __restrict__ scale, double* __restrict__ wr,
                                                           int32_t* __restrict__ label_out, double* __restrict__ partials) {
  extern __shared__ double sh[];
  double* xs = sh + threadIdx.x;
  const size_t ps = gmm_stride(d);
  double acc_w = 0.0, acc_u = 0.0, acc_n = 0.0;
  const int64_t ntiles = (n + GMM_THREADS - 1) / GMM_THREADS;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int64_t i = t * GMM_THREADS + threadIdx.x;
    if (i >= n) continue;
    bool member = !labels || labels[i] == label;
    if (!member) {
      if (mode == 0) for (int k = 0; k < K; ++k) wr[(size_t)k * n + i] = 0.0;
      if (mode == 1) wr[i] = 0.0;
      if (mode == 2 && label_out) label_out[i] = -1;
      continue;
    }
    for (int j = 0; j < d; ++j) {
      double v = x[(size_t)j * ld + i];
      xs[j * GMM_THREADS] = shift ? (v - shift[j]) * scale[j] : v;
    }
    double pk[GMM_KMAX_RESP];
    double best = -INFINITY, minm = INFINITY;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
      const double* pr = params + (size_t)k * ps;
      const double* mu = pr + 1;
      const double* P = pr + 1 + d;
      // (x - mu)^T P (x - mu), 8 rows of P at a time: P is symmetric, so the 8 entries P[r0 .. r0+7][j] are the contiguous
      // P[j][r0 .. r0+7] -- one wave-uniform 64-byte (scalar) load and one LDS read of x_j feed 8 FMAs.  (The row-by-row form
      // re-read x_j, mu_j and one matrix element per FMA: 1.3 TFLOP/s at d = 32.)
      double maha = 0.0;
      for (int r0 = 0; r0 < d; r0 += 8) {
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (r0 + 8 <= d) {
          // four columns per trip: their 4 x 64 B of matrix come as back-to-back scalar loads behind ONE wait (a scalar load
          // waited for alone costs its full latency per 8 FMAs -- tri.h); same order of the FMAs in every accumulator
          int j = 0;
          for (; j + 4 <= d; j += 4) {
            double t[4][8], xc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const double* __restrict__ Pj = P + (size_t)(j + a) * d + r0;
#pragma unroll
              for (int q = 0; q < 8; ++q) t[a][q] = Pj[q];
              xc[a] = xs[(j + a) * GMM_THREADS] - mu[j + a];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int q = 0; q < 8; ++q) acc[q] = fma(t[a][q], xc[a], acc[q]);
          }
          for (; j < d; ++j) {
            const double xc = xs[j * GMM_THREADS] - mu[j];
            const double* __restrict__ Pj = P + (size_t)j * d + r0;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fma(Pj[q], xc, acc[q]);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) maha = fma(xs[(r0 + q) * GMM_THREADS] - mu[r0 + q], acc[q], maha);
        } else {
          const int nq = d - r0;
          for (int j = 0; j < d; ++j) {
            const double xc = xs[j * GMM_THREADS] - mu[j];
            const double* __restrict__ Pj = P + (size_t)j * d + r0;
            for (int q = 0; q < nq; ++q) acc[q] = fma(Pj[q], xc, acc[q]);
          }
          for (int q = 0; q < nq; ++q) maha = fma(xs[(r0 + q) * GMM_THREADS] - mu[r0 + q], acc[q], maha);
        }
      }
      double logpdf = -0.5 * ((double)d * 1.8378770664093454836 + pr[1 + d + d * d] + maha);
      if (mode == 0) { if (k < GMM_KMAX_RESP) pk[k] = exp(pr[0]) * exp(logpdf); }
      else if (mode == 1) minm = fmin(minm, maha);
      else { double v = pr[0] + logpdf; if (v > best) { best = v; arg = k; } }
    }
    const double swi = sw ? sw[i] : 1.0;
    if (mode == 0) {
      double tot = 0.0;
      for (int k = 0; k < K && k < GMM_KMAX_RESP; ++k) tot += pk[k];
      for (int k = 0; k < K && k < GMM_KMAX_RESP; ++k) wr[(size_t)k * n + i] = swi * (pk[k] / (tot + eps));
      double l = log(tot + 1e-10);
      acc_w += swi * l;
      acc_u += l;
      acc_n += 1.0;
    } else if (mode == 1) {
      wr[i] = swi * minm;
    } else {
      label_out[i] = arg;
    }
  }
  acc_w = tph_wave_sum(acc_w);
  acc_u = tph_wave_sum(acc_u);
  acc_n = tph_wave_sum(acc_n);
  if (threadIdx.x == 0) {
    partials[(size_t)blockIdx.x * 3] = acc_w;
    partials[(size_t)blockIdx.x * 3 + 1] = acc_u;
    partials[(size_t)blockIdx.x * 3 + 2] = acc_n;
  }
}

__global__ void __launch_bounds__(256) k_colsum3(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[(size_t)b * 3 + c];
  __shared__ double sh[4];
  s = tph_block_sum(s, sh);
  if (threadIdx.x == 0) out[c] = s;
}

extern "C" int tph_gmm_estep(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* sw_dev,
                             const int32_t* labels_dev, int label, int K, const double* params_dev, int mode, double eps,
                             const double* shift_dev, const double* scale_dev, double* wr_dev, int32_t* label_out_dev,
                             double* stats_dev) {
  TPH_REQUIRE(ctx && x_dev && params_dev && n > 0 && ld >= n && K >= 1, "tph_gmm_estep: bad argument");
  TPH_REQUIRE(mode >= 0 && mode <= 2, "tph_gmm_estep: mode must be 0 (responsibilities), 1 (seeding), 2 (predict)");
  TPH_REQUIRE(mode != 0 || (K <= GMM_KMAX_RESP && wr_dev && stats_dev), "tph_gmm_estep: mode 0 needs K<=8, wr, stats");
  TPH_REQUIRE(mode != 1 || wr_dev, "tph_gmm_estep: mode 1 needs wr");
  TPH_REQUIRE(mode != 2 || label_out_dev, "tph_gmm_estep: mode 2 needs label_out");
  const int d = ctx->d;
  int64_t ntiles = (n + GMM_THREADS - 1) / GMM_THREADS;
  int nblk = (int)(ntiles < 8192 ? ntiles : 8192);
  if (tph_scratch_reserve(ctx, sizeof(double) * 3 * (size_t)nblk)) return -1;
  double* part = (double*)ctx->scratch;
  size_t lds = sizeof(double) * (size_t)d * GMM_THREADS;
  if (lds > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_gmm_estep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_gmm_estep, dim3(nblk), dim3(GMM_THREADS), lds, ctx->stream, x_dev, ld, n, d, sw_dev, labels_dev, label,
                     K, params_dev, mode, eps, shift_dev, scale_dev, wr_dev, label_out_dev, part);
  if (stats_dev) hipLaunchKernelGGL(k_colsum3, dim3(3), dim3(256), 0, ctx->stream, part, nblk, stats_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// ---- compaction of the kept history rows (w >= *thr) into a normalised compact working set ----
__global__ void __launch_bounds__(256) k_flag_kept(const double* __restrict__ w, int64_t n, const double* __restrict__ thr,
                                                   double* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (w[i] >= thr[0]) ? 1.0 : 0.0;
}
__global__ void __launch_bounds__(256) k_kept_list(const double* __restrict__ flag, const double* __restrict__ rank, int64_t n,
                                                   int64_t* __restrict__ list) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i] != 0.0) list[(int64_t)rank[i] - 1] = i;
}

extern "C" int tph_cdf(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev, double* cdf_dev);

extern "C" int tph_compact_indices(tph_ctx* ctx, const double* w_dev, int64_t n, const double* thr_dev, int64_t* idx_dev) {
  TPH_REQUIRE(ctx && w_dev && thr_dev && idx_dev && n > 0, "tph_compact_indices: bad argument");
  size_t tiles_bytes = sizeof(double) * (size_t)((n + 2047) / 2048 + 1);
  size_t off = (tiles_bytes + 255) / 256 * 256;
  if (tph_scratch_reserve(ctx, off + sizeof(double) * 2 * (size_t)n)) return -1;
  double* flag = (double*)((char*)ctx->scratch + off);
  double* rank = flag + n;
  unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_flag_kept, dim3(grid), dim3(256), 0, ctx->stream, w_dev, n, thr_dev, flag);
  int rc = tph_cdf(ctx, flag, n, nullptr, rank);
  if (rc) return rc;
  hipLaunchKernelGGL(k_kept_list, dim3(grid), dim3(256), 0, ctx->stream, flag, rank, n, idx_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// out[j][i] = (u_hist[j][idx_i] - shift_j) * scale_j ; wout[i] = w[idx_i]   (cluster.py:373-379,436-441)
__global__ void __launch_bounds__(256) k_gather_affine(const double* __restrict__ hu, int64_t cap, int d,
                                                       const int64_t* __restrict__ idx, int64_t m,
                                                       const double* __restrict__ shift, const double* __restrict__ scale,
                                                       const double* __restrict__ w, double* __restrict__ out, int64_t ld,
                                                       double* __restrict__ wout) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  int64_t s = idx[i];
  for (int j = 0; j < d; ++j) {
    double v = hu[(size_t)j * cap + s];
    out[(size_t)j * ld + i] = shift ? (v - shift[j]) * scale[j] : v;
  }
  if (wout) wout[i] = w ? w[s] : 1.0;
}

extern "C" int tph_gather_u_affine(tph_ctx* ctx, const int64_t* idx_dev, int64_t m, const double* shift_dev,
                                   const double* scale_dev, const double* w_dev, double* out_dev, int64_t ld, double* wout_dev) {
  TPH_REQUIRE(ctx && idx_dev && out_dev && m > 0 && ld >= m && ctx->size > 0, "tph_gather_u_affine: bad argument");
  hipLaunchKernelGGL(k_gather_affine, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, ctx->u, ctx->cap, ctx->d,
                     idx_dev, m, shift_dev, scale_dev, w_dev, out_dev, ld, wout_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// in place: x[j][i] = (x[j][i] - shift_j) * scale_j on an explicit SoA array
__global__ void __launch_bounds__(256) k_affine(double* __restrict__ x, int64_t ld, int64_t n, const double* __restrict__ shift,
                                                const double* __restrict__ scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  if (i < n) x[(size_t)j * ld + i] = (x[(size_t)j * ld + i] - shift[j]) * scale[j];
}

extern "C" int tph_affine(tph_ctx* ctx, double* x_dev, int64_t ld, int64_t n, const double* shift_dev, const double* scale_dev) {
  TPH_REQUIRE(ctx && x_dev && shift_dev && scale_dev && n > 0 && ld >= n, "tph_affine: bad argument");
  hipLaunchKernelGGL(k_affine, dim3((unsigned)((n + 255) / 256), ctx->d), dim3(256), 0, ctx->stream, x_dev, ld, n, shift_dev,
                     scale_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}
