// What does rocprofv3's FETCH_SIZE count on gfx950 for the access patterns of this library's kernels?  The guide calibrates one
// pattern (16 B per lane, streaming: the counter shows HALF the bytes); the roofline tables of rounds 2-4 doubled FETCH_SIZE for
// every kernel and called the result "an upper bound" for the indexed ones.  This probe runs each pattern once over 2 GiB (beyond
// the 256 MiB Infinity Cache) with a known number of bytes / sectors / lines touched; run it under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -- ./ubench_fetch
// and compare per kernel (tools/fetch_calibration.py).  Prints the expected figures as one JSON object.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_fetch tools/ubench_fetch.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_stream16(const double2* __restrict__ p, size_t n2, double* sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 1.2345e300) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream8(const double* __restrict__ p, size_t n, double* sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 1.2345e300) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream4(const int* __restrict__ p, size_t n, double* sink) {
  int acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 123456789) sink[0] = acc;
}
// lane i reads the 8 bytes at element i * S: S = 8 -> one element of every 64-byte sector, 16 -> of every 128-byte line, ...
template <int S>
__global__ void __launch_bounds__(256) k_stride(const double* __restrict__ p, size_t m, double* sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) acc += p[i * S];
  if (acc == 1.2345e300) sink[0] = acc;
}
// m reads of 8 bytes at pseudo-random elements (a multiplicative hash: a permutation of [0, 2^28))
__global__ void __launch_bounds__(256) k_random8(const double* __restrict__ p, size_t m, double* sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256)
    acc += p[(i * 0x9E3779B1ull) & ((1ull << 28) - 1)];
  if (acc == 1.2345e300) sink[0] = acc;
}
// rows of 80 bytes (10 doubles) gathered whole from random row numbers, 10 lanes to a row (the row-major mirror's pattern)
__global__ void __launch_bounds__(256) k_rows80(const double* __restrict__ p, size_t m, double* sink) {
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < m * 10; e += (size_t)gridDim.x * 256) {
    const size_t r = e / 10, c = e - r * 10;
    acc += p[((r * 0x9E3779B1ull) % 26843545ull) * 10 + c];
  }
  if (acc == 1.2345e300) sink[0] = acc;
}
// 13 % of the rows of a column-major table, every kept row all 10 columns (the compaction's pattern: a Bernoulli mask by hash)
__global__ void __launch_bounds__(256) k_sparse_cols(const double* __restrict__ p, size_t rows, double* sink, unsigned long long* kept) {
  double acc = 0.0;
  unsigned long long k = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (size_t)gridDim.x * 256) {
    const unsigned h = (unsigned)((i * 0x9E3779B97F4A7C15ull) >> 40);
    if (h % 100 < 13) {
      ++k;
      for (int j = 0; j < 10; ++j) acc += p[(size_t)j * rows + i];
    }
  }
  if (acc == 1.2345e300) sink[0] = acc;
  atomicAdd(kept, k);
}
__global__ void __launch_bounds__(256) k_fill(double* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.0;
}

int main() {
  const size_t N = 1ull << 28;                 // doubles: 2 GiB
  double *buf, *sink;
  unsigned long long* kept;
  CHECK(hipMalloc((void**)&buf, N * 8));
  CHECK(hipMalloc((void**)&sink, 64));
  CHECK(hipMalloc((void**)&kept, 8));
  CHECK(hipMemset(kept, 0, 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, N);
  CHECK(hipDeviceSynchronize());
  const dim3 g(4096), b(256);
  const size_t m_rand = 1ull << 24, m_rows = 1ull << 22, rows_sparse = N / 10;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_stream16, g, b, 0, 0, (const double2*)buf, N / 2, sink);
    hipLaunchKernelGGL(k_stream8, g, b, 0, 0, buf, N, sink);
    hipLaunchKernelGGL(k_stream4, g, b, 0, 0, (const int*)buf, 2 * N, sink);
    hipLaunchKernelGGL(k_stride<8>, g, b, 0, 0, buf, N / 8, sink);
    hipLaunchKernelGGL(k_stride<16>, g, b, 0, 0, buf, N / 16, sink);
    hipLaunchKernelGGL(k_stride<32>, g, b, 0, 0, buf, N / 32, sink);
    hipLaunchKernelGGL(k_random8, g, b, 0, 0, buf, m_rand, sink);
    hipLaunchKernelGGL(k_rows80, g, b, 0, 0, buf, m_rows, sink);
    hipLaunchKernelGGL(k_sparse_cols, g, b, 0, 0, buf, rows_sparse, sink, kept);
    CHECK(hipDeviceSynchronize());
  }
  unsigned long long k = 0;
  CHECK(hipMemcpy(&k, kept, 8, hipMemcpyDeviceToHost));
  k /= 2;
  printf("{\"bytes_total\": %zu,\n", N * 8);
  printf(" \"k_stream16\": {\"bytes\": %zu},\n \"k_stream8\": {\"bytes\": %zu},\n \"k_stream4\": {\"bytes\": %zu},\n", N * 8, N * 8, N * 8);
  printf(" \"k_stride<8>\": {\"elements\": %zu, \"bytes_used\": %zu, \"sectors64\": %zu, \"lines128\": %zu},\n", N / 8, N, N / 8, N / 16);
  printf(" \"k_stride<16>\": {\"elements\": %zu, \"bytes_used\": %zu, \"sectors64\": %zu, \"lines128\": %zu},\n", N / 16, N / 2, N / 16, N / 16);
  printf(" \"k_stride<32>\": {\"elements\": %zu, \"bytes_used\": %zu, \"sectors64\": %zu, \"lines128\": %zu},\n", N / 32, N / 4, N / 32, N / 32);
  printf(" \"k_random8\": {\"elements\": %zu, \"bytes_used\": %zu, \"sectors64\": %zu, \"lines128\": %zu},\n", m_rand, m_rand * 8, m_rand, m_rand);
  printf(" \"k_rows80\": {\"rows\": %zu, \"bytes_used\": %zu, \"sectors64_expected\": %.0f, \"lines128_expected\": %.0f},\n", m_rows, m_rows * 80,
         (double)m_rows * 2.0, (double)m_rows * 1.5);      // rows start at multiples of 16 B
  printf(" \"k_sparse_cols\": {\"rows\": %zu, \"kept\": %llu, \"bytes_used\": %llu, \"sectors64_expected\": %.0f, \"lines128_expected\": %.0f}}\n", rows_sparse, k,
         k * 80ull, 10.0 * (double)(rows_sparse / 8) * (1.0 - 0.3282), 10.0 * (double)(rows_sparse / 16) * (1.0 - 0.1077));
  return 0;
}
