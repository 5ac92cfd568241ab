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
                                                           int32_t* __restrict__ label_out, double* __restrict__ partials,
                                                           const double* __restrict__ done) {
  if (done && done[0] != 0.0) return;            // device-paced EM (tph_gmm_em_run): the fit has converged, later passes are no-ops
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
    for (int j0 = 0; j0 < d; j0 += 8) {            // eight coordinates requested before the first is stored (one by one the fill
      double v[8];                                 // of a row was a chain of d memory round trips)
#pragma unroll
      for (int a = 0; a < 8; ++a) v[a] = j0 + a < d ? x[(size_t)(j0 + a) * ld + i] : 0.0;
#pragma unroll
      for (int a = 0; a < 8; ++a)
        if (j0 + a < d) xs[(j0 + a) * GMM_THREADS] = shift ? (v[a] - shift[j0 + a]) * scale[j0 + a] : v[a];
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

// n_dim >= 16: the same three modes on the FP64 matrix cores.  A wave takes 16 rows; with Xc = X - mu (d x 16) the quadratic forms
// are the column sums of Xc .* (P Xc), and P Xc is d/16 x d/4 products v_mfma_f64_16x16x4: for the 16 rows rb of P and the four
// coordinates 4s .. 4s+3, lane (g = lane / 16, c = lane % 16) supplies A = P[16 rb + c][4s + g] -- read as P[4s + g][16 rb + c], P is
// symmetric: 128 contiguous bytes per lane group -- and B = Xc[4s + g][row c], and receives Y[16 rb + 4v + g][row c], v = 0..3:
// the very coordinates 4 (4 rb + v) + g whose Xc it already holds as the B operand of step 4 rb + v.  So the lane multiplies its own
// registers and two shuffles add the four lane groups; no LDS, no barrier, the rows come straight from global memory in 128-byte
// segments.  (The one-lane-per-row form above keeps a row in LDS and streams P through scalar loads: 107 us per pass at 7 x 10^5
// rows x 32-D, a third of the VALU issue rate, waves waiting half their cycles.)  NB = ceil(n_dim / 16).
typedef double gmm_v4d __attribute__((ext_vector_type(4)));
template <int NB>
__global__ void __launch_bounds__(256) k_gmm_estep_mf(const double* __restrict__ x, int64_t ld, int64_t n, int d,
                                                      const double* __restrict__ sw, const int32_t* __restrict__ labels,
                                                      int label, int K, const double* __restrict__ params, int mode,
                                                      double eps, const double* __restrict__ shift,
                                                      const double* __restrict__ scale, double* __restrict__ wr,
                                                      int32_t* __restrict__ label_out, double* __restrict__ partials,
                                                      const double* __restrict__ done) {
  if (done && done[0] != 0.0) return;            // (as k_gmm_estep)
  constexpr int KS = 4 * NB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const size_t ps = gmm_stride(d);
  double acc_w = 0.0, acc_u = 0.0, acc_n = 0.0;
  const int64_t ntiles = (n + 15) / 16;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t i = t * 16 + c;
    const bool in = i < n;
    double xr[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const int j = 4 * s_ + g;
      const double v = (in && j < d) ? x[(size_t)j * ld + i] : 0.0;
      xr[s_] = (shift && j < d) ? (v - shift[j]) * scale[j] : v;
    }
    const bool member = in && (!labels || labels[i] == label);
    double pk[GMM_KMAX_RESP];
    double best = -INFINITY, minm = INFINITY;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
      const double* pr = params + (size_t)k * ps;
      const double* mu = pr + 1;
      const double* P = pr + 1 + d;
      double xc[KS];
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        const int j = 4 * s_ + g;
        xc[s_] = j < d ? xr[s_] - mu[j] : 0.0;
      }
      double part = 0.0;
#pragma unroll
      for (int rb = 0; rb < NB; ++rb) {
        gmm_v4d y = {0.0, 0.0, 0.0, 0.0};
        const int r = 16 * rb + c;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
          const int j = 4 * s_ + g;
          const double a = (j < d && r < d) ? P[(size_t)j * d + r] : 0.0;
          y = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xc[s_], y, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) part = fma(xc[4 * rb + v], y[v], part);
      }
      part += __shfl_xor(part, 16, 64);
      const double maha = part + __shfl_xor(part, 32, 64);
      const double logpdf = -0.5 * ((double)d * 1.8378770664093454836 + pr[1 + d + d * d] + maha);
      if (mode == 0) { if (k < GMM_KMAX_RESP) pk[k] = exp(pr[0]) * exp(logpdf); }
      else if (mode == 1) minm = fmin(minm, maha);
      else { const double v = pr[0] + logpdf; if (v > best) { best = v; arg = k; } }
    }
    if (g != 0 || !in) continue;                   // lanes 0..15 of the wave write the tile's rows
    if (!member) {
      if (mode == 0) for (int k = 0; k < K; ++k) wr[(size_t)k * n + i] = 0.0;
      if (mode == 1) wr[i] = 0.0;
      if (mode == 2 && label_out) label_out[i] = -1;
      continue;
    }
    const double swi = sw ? sw[i] : 1.0;
    if (mode == 0) {
      double tot = 0.0;
      for (int k = 0; k < K && k < GMM_KMAX_RESP; ++k) tot += pk[k];
      for (int k = 0; k < K && k < GMM_KMAX_RESP; ++k) wr[(size_t)k * n + i] = swi * (pk[k] / (tot + eps));
      const double l = log(tot + 1e-10);
      acc_w += swi * l;
      acc_u += l;
      acc_n += 1.0;
    } else if (mode == 1) {
      wr[i] = swi * minm;
    } else {
      label_out[i] = arg;
    }
  }
  __shared__ double sh[4];
  acc_w = tph_block_sum(acc_w, sh);
  acc_u = tph_block_sum(acc_u, sh);
  acc_n = tph_block_sum(acc_n, sh);
  if (threadIdx.x == 0) {
    partials[(size_t)blockIdx.x * 3] = acc_w;
    partials[(size_t)blockIdx.x * 3 + 1] = acc_u;
    partials[(size_t)blockIdx.x * 3 + 2] = acc_n;
  }
}

__global__ void __launch_bounds__(256) k_colsum3(const double* __restrict__ partials, int nblocks, double* __restrict__ out,
                                                 const double* __restrict__ done) {
  if (done && done[0] != 0.0) return;
  int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[(size_t)b * 3 + c];
  __shared__ double sh[4];
  s = tph_block_sum(s, sh);
  if (threadIdx.x == 0) out[c] = s;
}

static int gmm_estep_launch(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* sw_dev,
                            const int32_t* labels_dev, int label, int K, const double* params_dev, int mode, double eps,
                            const double* shift_dev, const double* scale_dev, double* wr_dev, int32_t* label_out_dev,
                            double* stats_dev, const double* done_dev) {
  TPH_REQUIRE(ctx && x_dev && params_dev && n > 0 && ld >= n && K >= 1, "tph_gmm_estep: bad argument");
  TPH_REQUIRE(mode >= 0 && mode <= 2, "tph_gmm_estep: mode must be 0 (responsibilities), 1 (seeding), 2 (predict)");
  TPH_REQUIRE(mode != 0 || (K <= GMM_KMAX_RESP && wr_dev && stats_dev), "tph_gmm_estep: mode 0 needs K<=8, wr, stats");
  TPH_REQUIRE(mode != 1 || wr_dev, "tph_gmm_estep: mode 1 needs wr");
  TPH_REQUIRE(mode != 2 || label_out_dev, "tph_gmm_estep: mode 2 needs label_out");
  const int d = ctx->d;
  // matrix cores on request, up to 48-D (TPH_OPT_GMM_KERNEL: 0 auto = 1 one lane per row | 2 MFMA; above 48-D the unrolled products
  // of a 16-row tile need more than 192 registers).  MEASURED on config 3 (705 passes over ~7 x 10^5 rows x 32-D): 157.6 us per
  // pass against 107.1 us of the lane-per-row kernel -- 16 rows per wave leave the exp / log epilogue on a quarter of the lanes
  // and the precision matrix is re-read per tile; the default stays the lane-per-row kernel (profiles/r05_c3_fit_kernels.json)
  if (d >= 16 && d <= 48 && ctx->gmm_kernel == 2) {
    const int64_t nt = (n + 63) / 64;
    const int nb = (int)(nt < 4096 ? nt : 4096);
    if (tph_scratch_reserve(ctx, sizeof(double) * 3 * (size_t)nb)) return -1;
    double* part = (double*)ctx->scratch;
    switch ((d + 15) / 16) {
#define C(NB_) case NB_: hipLaunchKernelGGL((k_gmm_estep_mf<NB_>), dim3(nb), dim3(256), 0, ctx->stream, x_dev, ld, n, d, sw_dev, labels_dev, label, K, params_dev, mode, eps, shift_dev, scale_dev, wr_dev, label_out_dev, part, done_dev); break;
      C(1) C(2) C(3)
#undef C
    }
    if (stats_dev) hipLaunchKernelGGL(k_colsum3, dim3(3), dim3(256), 0, ctx->stream, part, nb, stats_dev, done_dev);
    TPH_LAUNCH_CHECK();
    return 0;
  }
  int64_t ntiles = (n + GMM_THREADS - 1) / GMM_THREADS;
  int nblk = (int)(ntiles < 8192 ? ntiles : 8192);
  if (tph_scratch_reserve(ctx, sizeof(double) * 3 * (size_t)nblk)) return -1;
  double* part = (double*)ctx->scratch;
  size_t lds = sizeof(double) * (size_t)d * GMM_THREADS;
  if (lds > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_gmm_estep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_gmm_estep, dim3(nblk), dim3(GMM_THREADS), lds, ctx->stream, x_dev, ld, n, d, sw_dev, labels_dev, label,
                     K, params_dev, mode, eps, shift_dev, scale_dev, wr_dev, label_out_dev, part, done_dev);
  if (stats_dev) hipLaunchKernelGGL(k_colsum3, dim3(3), dim3(256), 0, ctx->stream, part, nblk, stats_dev, done_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

extern "C" int tph_gmm_estep(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* sw_dev,
                             const int32_t* labels_dev, int label, int K, const double* params_dev, int mode, double eps,
                             const double* shift_dev, const double* scale_dev, double* wr_dev, int32_t* label_out_dev,
                             double* stats_dev) {
  return gmm_estep_launch(ctx, x_dev, ld, n, sw_dev, labels_dev, label, K, params_dev, mode, eps, shift_dev, scale_dev, wr_dev,
                          label_out_dev, stats_dev, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-paced EM (cluster.py:104-133, 174-304 for covariance_type "full").  The reference's loop -- pack the parameters
// (precisions and log-determinants of the K covariances), E-step, convergence test on the weighted log-likelihood, M-step --
// ran with the host in it twice per iteration (the E-step's sums, then the M-step's moments back for the d x d inverses and the
// parameters forward again): 0.3 ms per iteration whatever the size of the working set, 3.1 of the 3.7 s of a 1 000-particle
// run (10 500 EM iterations in 1 200 fits).  Here the loop's state lives in one device block and a call ENQUEUES a batch of
// iterations: the parameters are formed by a kernel (Cholesky, inverse and log-determinant in LDS), the test is a one-thread
// kernel that raises a `done` flag, and every pass behind a raised flag is a no-op.  The host reads 16 doubles per batch.
// State block (doubles): ctl[16] = {iteration, done, lower bound, n_iter, stats[3], -, ...}; the M-step's moments (sums
// K x (1+d), means K x d, scatter K x d x d); the packed parameters of the E-step; the weights / means / covariances those
// parameters were formed from (what a fit returns: the ones of its LAST E-step).
struct em_layout { size_t ctl, sums, means, scat, params, w_used, m_used, c_used, total; };
static inline em_layout em_lay(int d, int K) {
  em_layout L;
  size_t o = 0;
  L.ctl = o; o += 16;
  L.sums = o; o += (size_t)K * (1 + d);
  L.means = o; o += (size_t)K * d;
  L.scat = o; o += (size_t)K * d * d;
  L.params = o; o += (size_t)K * (2 + d + (size_t)d * d);
  L.w_used = o; o += (size_t)K;
  L.m_used = o; o += (size_t)K * d;
  L.c_used = o; o += (size_t)K * d * d;
  L.total = o;
  return L;
}
extern "C" int64_t tph_gmm_em_state_doubles(int d, int K) { return (int64_t)em_lay(d, K).total; }

__global__ void k_em_begin(double* __restrict__ ctl) {
  if (threadIdx.x < 16) ctl[threadIdx.x] = 0.0;
  if (threadIdx.x == 0) ctl[2] = -INFINITY;
}
// means[k][j] = sums[k][1+j] / (sums[k][0] + 1e-10)   (cluster.py:205-209); one workgroup per component
__global__ void __launch_bounds__(128) k_em_means(const double* __restrict__ sums, int d, double* __restrict__ means) {
  const double* s = sums + (size_t)blockIdx.x * (1 + d);
  for (int j = threadIdx.x; j < d; j += blockDim.x) means[(size_t)blockIdx.x * d + j] = s[1 + j] / (s[0] + 1e-10);
}
// weights, covariances and the packed E-step parameters of component blockIdx.x from the M-step's moments
// (cluster.py:200-235 and _inv_logdet's rule :186-193: precision and log-determinant of cov + reg I, reg I alone when that is
// not positive definite).  Factor and inverse factor in LDS.
__global__ void __launch_bounds__(256) k_em_params(double* __restrict__ st, em_layout L, int d, int K, double reg) {
  extern __shared__ double pl[];
  double* C = pl;                          // cov + reg I, then its Cholesky factor (lower)
  double* W = pl + (size_t)d * d;          // L^-1
  const double* ctl = st + L.ctl;
  if (ctl[1] != 0.0) return;
  const int k = blockIdx.x;
  const double* sums = st + L.sums;
  const double tot = sums[(size_t)k * (1 + d)];
  double totsum = 0.0;
  for (int c = 0; c < K; ++c) totsum += sums[(size_t)c * (1 + d)];
  const double weight = tot / totsum;
  const double* mean = st + L.means + (size_t)k * d;
  const double* scat = st + L.scat + (size_t)k * d * d;
  double* cu = st + L.c_used + (size_t)k * d * d;
  double* par = st + L.params + (size_t)k * (2 + d + (size_t)d * d);
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
    const double c = scat[e] / (tot + 1e-10);
    cu[e] = c;
    C[e] = c + ((e / d == e % d) ? reg : 0.0);
  }
  for (int j = threadIdx.x; j < d; j += blockDim.x) { st[L.m_used + (size_t)k * d + j] = mean[j]; par[1 + j] = mean[j]; }
  if (threadIdx.x == 0) { st[L.w_used + k] = weight; par[0] = log(weight); }
  __shared__ int fail;
  __shared__ double piv;
  if (threadIdx.x == 0) fail = 0;
  __syncthreads();
  for (int j = 0; j < d; ++j) {            // in-place Cholesky of the lower triangle
    if (threadIdx.x == 0) {
      double sd = C[j * d + j];
      for (int q = 0; q < j; ++q) sd -= C[j * d + q] * C[j * d + q];
      if (!(sd > 0.0)) fail = 1;
      piv = sqrt(sd);
      C[j * d + j] = piv;
    }
    __syncthreads();
    if (fail) break;
    for (int i = j + 1 + threadIdx.x; i < d; i += blockDim.x) {
      double sd = C[i * d + j];
      for (int q = 0; q < j; ++q) sd -= C[i * d + q] * C[j * d + q];
      C[i * d + j] = sd / piv;
    }
    __syncthreads();
  }
  __syncthreads();
  if (fail) {                              // cov + reg I is not positive definite: reg I (cluster.py:190-193)
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) par[1 + d + e] = (e / d == e % d) ? 1.0 / reg : 0.0;
    if (threadIdx.x == 0) par[1 + d + (size_t)d * d] = (double)d * log(reg);
    return;
  }
  for (int c = threadIdx.x; c < d; c += blockDim.x) {      // column c of W solves L y = e_c
    for (int i = 0; i < d; ++i) {
      if (i < c) { W[i * d + c] = 0.0; continue; }
      double sd = (i == c) ? 1.0 : 0.0;
      for (int q = c; q < i; ++q) sd -= C[i * d + q] * W[q * d + c];
      W[i * d + c] = sd / C[i * d + i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) {  // precision = W^T W
    const int i = e / d, j = e % d, m = i > j ? i : j;
    double sd = 0.0;
    for (int q = m; q < d; ++q) sd += W[q * d + i] * W[q * d + j];
    par[1 + d + e] = sd;
  }
  if (threadIdx.x == 0) {
    double ld = 0.0;
    for (int j = 0; j < d; ++j) ld += log(C[j * d + j]);
    par[1 + d + (size_t)d * d] = 2.0 * ld;
  }
}
// the loop's bookkeeping (cluster.py:104-121): iteration 0 has no test; from iteration 1 on the fit stops when the weighted
// log-likelihood gained less than tol (or at max_iter), keeping the PREVIOUS bound; otherwise the bound is updated
__global__ void k_em_control(double* __restrict__ ctl, double tol, int max_iter) {
  if (ctl[1] != 0.0) return;
  const int it = (int)ctl[0];
  if (it > 0) {
    ctl[3] = (double)it;
    if (ctl[4] - ctl[2] < tol || it == max_iter) { ctl[1] = 1.0; return; }
    ctl[2] = ctl[4];
  }
  ctl[0] = (double)(it + 1);
}

extern "C" int tph_x_weighted_sums(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* w_dev, double* sums_dev,
                                   double* range_dev);
extern "C" int tph_x_weighted_cov(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* w_dev,
                                  const double* mean_dev, double* cov_dev);

static int em_mstep(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int K, const double* wr_dev, double* st, const em_layout& L) {
  const int d = ctx->d;
  for (int k = 0; k < K; ++k)
    if (int rc = tph_x_weighted_sums(ctx, x_dev, ld, n, wr_dev + (size_t)k * n, st + L.sums + (size_t)k * (1 + d), nullptr)) return rc;
  hipLaunchKernelGGL(k_em_means, dim3(K), dim3(128), 0, ctx->stream, (const double*)(st + L.sums), d, st + L.means);
  for (int k = 0; k < K; ++k)
    if (int rc = tph_x_weighted_cov(ctx, x_dev, ld, n, wr_dev + (size_t)k * n, st + L.means + (size_t)k * d, st + L.scat + (size_t)k * d * d))
      return rc;
  return 0;
}

// the fit's first M-step (from the initial responsibilities in wr) and a fresh control block
extern "C" int tph_gmm_em_begin(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, int K, const double* wr_dev,
                                double* state_dev) {
  TPH_REQUIRE(ctx && x_dev && wr_dev && state_dev && n > 0 && ld >= n && K >= 1 && K <= GMM_KMAX_RESP, "tph_gmm_em_begin: bad argument");
  const em_layout L = em_lay(ctx->d, K);
  hipLaunchKernelGGL(k_em_begin, dim3(1), dim3(64), 0, ctx->stream, state_dev + L.ctl);
  if (int rc = em_mstep(ctx, x_dev, ld, n, K, wr_dev, state_dev, L)) return rc;
  TPH_LAUNCH_CHECK();
  return 0;
}

// `iters` EM iterations enqueued back to back; state_dev[0..15] tells the host how far the fit is (done flag [1])
extern "C" int tph_gmm_em_run(tph_ctx* ctx, const double* x_dev, int64_t ld, int64_t n, const double* sw_dev,
                              const int32_t* labels_dev, int label, int K, double* wr_dev, double* state_dev, double reg,
                              double tol, int max_iter, int iters) {
  TPH_REQUIRE(ctx && x_dev && sw_dev && wr_dev && state_dev && n > 0 && ld >= n && K >= 1 && K <= GMM_KMAX_RESP && iters >= 1,
              "tph_gmm_em_run: bad argument");
  const int d = ctx->d;
  const em_layout L = em_lay(d, K);
  const size_t lds = sizeof(double) * 2 * (size_t)d * d;
  TPH_REQUIRE(lds <= 160 * 1024, "tph_gmm_em_run: n_dim=%d too large", d);
  if (lds > 64 * 1024)
    TPH_HIP(hipFuncSetAttribute((const void*)k_em_params, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double* ctl = state_dev + L.ctl;
  for (int i = 0; i < iters; ++i) {
    hipLaunchKernelGGL(k_em_params, dim3(K), dim3(256), lds, ctx->stream, state_dev, L, d, K, reg);
    if (int rc = gmm_estep_launch(ctx, x_dev, ld, n, sw_dev, labels_dev, label, K, state_dev + L.params, 0, 1e-10, nullptr, nullptr,
                                  wr_dev, nullptr, ctl + 4, ctl + 1))
      return rc;
    hipLaunchKernelGGL(k_em_control, dim3(1), dim3(1), 0, ctx->stream, ctl, tol, max_iter);
    if (int rc = em_mstep(ctx, x_dev, ld, n, K, wr_dev, state_dev, L)) return rc;
  }
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
  int rc = tph_cdf_plain(ctx, flag, n, nullptr, rank);      // a prefix COUNT of 0 / 1 flags: exact in any order
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

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_cluster(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
