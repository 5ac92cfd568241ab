// The reductions of the Student-t EM (opt-in extension): tempest/student.py:40-57 (the digamma equation for nu) and
// :96-102 (the weights w_i = (nu + d) / (nu + delta_i) of the covariance / mean update).  The reference's own loop never gets
// here -- with the NumPy / SciPy versions it pins, func0(1e300) evaluates to >= 0 on its first pass, nu becomes inf and the START
// values are returned (that effective estimator is K11, tph_fit_modes).  These kernels serve `fit_mvstud(..., em=True)` and
// `Sampler(..., student_em=True)`: rows carry integer multiplicities (the x4 up-sampling of modes.py:196-201 as counts) and an
// optional label filter; delta_i = |L^-1 (x_i - mu)|^2 comes from tph_gmm_estep (mode 1, one component).
#include "common.h"

// per trial nu (up to 16 per pass, like the reweight reduction): sum_i c_i log w_i and sum_i c_i w_i -- block partials, combined
// in a fixed order (deterministic)
__global__ void __launch_bounds__(256) k_student_sums(const double* __restrict__ delta, const int32_t* __restrict__ counts,
                                                      const int32_t* __restrict__ labels, int label, int64_t n, int d, int nb,
                                                      const double* __restrict__ nus, double* __restrict__ partials /* [grid][nb][2] */) {
  __shared__ double sh[4];
  double sl[16], sw[16];
  for (int b = 0; b < 16; ++b) { sl[b] = 0.0; sw[b] = 0.0; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double c = (!labels || labels[i] == label) ? (double)counts[i] : 0.0;
    if (c == 0.0) continue;
    const double dl = delta[i];
    for (int b = 0; b < nb; ++b) {
      const double w = (nus[b] + (double)d) / (nus[b] + dl);
      sl[b] = fma(c, log(w), sl[b]);
      sw[b] = fma(c, w, sw[b]);
    }
  }
  for (int b = 0; b < nb; ++b) {
    const double a = tph_block_sum(sl[b], sh);
    if (threadIdx.x == 0) partials[((size_t)blockIdx.x * nb + b) * 2] = a;
    const double w = tph_block_sum(sw[b], sh);
    if (threadIdx.x == 0) partials[((size_t)blockIdx.x * nb + b) * 2 + 1] = w;
  }
}
__global__ void __launch_bounds__(64) k_student_sums_finish(const double* __restrict__ partials, int grid, int nb, double* __restrict__ out) {
  const int t = threadIdx.x;
  if (t >= 2 * nb) return;
  double s = 0.0;
  for (int g = 0; g < grid; ++g) s += partials[(size_t)g * nb * 2 + t];
  out[t] = s;
}
// v_i = c_i (nu + d) / (nu + delta_i) (0 outside the label): the weights of the covariance and mean update
__global__ void __launch_bounds__(256) k_student_weights(const double* __restrict__ delta, const int32_t* __restrict__ counts,
                                                         const int32_t* __restrict__ labels, int label, int64_t n, int d, double nu,
                                                         double* __restrict__ v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double c = (!labels || labels[i] == label) ? (double)counts[i] : 0.0;
  v[i] = c == 0.0 ? 0.0 : c * (nu + (double)d) / (nu + delta[i]);
}

extern "C" int tph_student_sums(tph_ctx* ctx, const double* delta_dev, const int32_t* counts_dev, const int32_t* labels_dev, int label,
                                int64_t n, const double* nus_host, int nb, double* out_host /* [nb][2] */) {
  TPH_REQUIRE(ctx && delta_dev && counts_dev && nus_host && out_host && n > 0 && nb >= 1 && nb <= 16, "tph_student_sums: bad argument");
  const int grid = tph_grid_for(n, 256, 4, 1024);
  const size_t need = sizeof(double) * ((size_t)grid * nb * 2 + 64);
  if (tph_scratch_reserve(ctx, need)) return -1;
  double* part = (double*)ctx->scratch;
  double* nus_dev = part + (size_t)grid * nb * 2;
  double* out_dev = nus_dev + 16;
  TPH_HIP(hipMemcpyAsync(nus_dev, nus_host, sizeof(double) * nb, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_student_sums, dim3(grid), dim3(256), 0, ctx->stream, delta_dev, counts_dev, labels_dev, label, n, ctx->d, nb,
                     (const double*)nus_dev, part);
  hipLaunchKernelGGL(k_student_sums_finish, dim3(1), dim3(64), 0, ctx->stream, (const double*)part, grid, nb, out_dev);
  TPH_LAUNCH_CHECK();
  TPH_HIP(hipMemcpyAsync(out_host, out_dev, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, ctx->stream));
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int tph_student_weights(tph_ctx* ctx, const double* delta_dev, const int32_t* counts_dev, const int32_t* labels_dev, int label,
                                   int64_t n, double nu, double* v_dev) {
  TPH_REQUIRE(ctx && delta_dev && counts_dev && v_dev && n > 0 && nu > 0.0, "tph_student_weights: bad argument");
  hipLaunchKernelGGL(k_student_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, delta_dev, counts_dev, labels_dev,
                     label, n, ctx->d, nu, v_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_student(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
