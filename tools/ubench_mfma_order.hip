// What arithmetic does v_mfma_f64_16x16x4_f64 perform inside one instruction?  Dumps random A (16 x 4), B (4 x 16), C (16 x 16) and
// the instruction's D for a number of trials; tools/mfma_order_check.py tests hypotheses (chained FMAs in k order, one rounding
// of the exact sum, ...) in exact rational arithmetic.  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_order tools/ubench_mfma_order.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_one(const double* A, const double* B, const double* C, double* D, int trials) {
  const int lane = threadIdx.x & 63, n = lane & 15, k = lane >> 4;
  for (int t = 0; t < trials; ++t) {
    const double a = A[(size_t)t * 64 + n * 4 + k];          // A[i = n][k]
    const double b = B[(size_t)t * 64 + k * 16 + n];         // B[k][j = n]
    d4 c;
    for (int v = 0; v < 4; ++v) c[v] = C[(size_t)t * 256 + (size_t)(4 * v + k) * 16 + n];      // C[i = 4v + k][j = n]
    d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(size_t)t * 256 + (size_t)(4 * v + k) * 16 + n] = r[v];
  }
}

int main(int argc, char** argv) {
  const int trials = 64;
  const char* out = argc > 1 ? argv[1] : "mfma_order.bin";
  size_t na = (size_t)trials * 64, nc = (size_t)trials * 256;
  double *hA = (double*)malloc(8 * na), *hB = (double*)malloc(8 * na), *hC = (double*)malloc(8 * nc), *hD = (double*)malloc(8 * nc);
  srand(12345);
  auto rnd = [&](int spread) {
    double m = (double)rand() / RAND_MAX * 2.0 - 1.0 + ((double)rand() / RAND_MAX) * 1e-9;
    return ldexp(m, (rand() % (2 * spread + 1)) - spread);
  };
  for (size_t i = 0; i < na; ++i) { hA[i] = rnd(i < na / 2 ? 2 : 30); hB[i] = rnd(i < na / 2 ? 2 : 30); }
  for (size_t i = 0; i < nc; ++i) hC[i] = (i % 3 == 0) ? 0.0 : rnd(i < nc / 2 ? 2 : 30);
  double *A, *B, *C, *D;
  hipMalloc(&A, 8 * na); hipMalloc(&B, 8 * na); hipMalloc(&C, 8 * nc); hipMalloc(&D, 8 * nc);
  hipMemcpy(A, hA, 8 * na, hipMemcpyHostToDevice); hipMemcpy(B, hB, 8 * na, hipMemcpyHostToDevice); hipMemcpy(C, hC, 8 * nc, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, A, B, C, D, trials);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
  hipMemcpy(hD, D, 8 * nc, hipMemcpyDeviceToHost);
  FILE* f = fopen(out, "wb");
  int32_t hdr[2] = {trials, 0};
  fwrite(hdr, 4, 2, f); fwrite(hA, 8, na, f); fwrite(hB, 8, na, f); fwrite(hC, 8, nc, f); fwrite(hD, 8, nc, f);
  fclose(f);
  printf("wrote %s (%d trials)\n", out, trials);
  return 0;
}
