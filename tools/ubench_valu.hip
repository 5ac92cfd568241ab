// Issue-cost microbenchmarks for the VALU-bound proposal kernels (gfx950): how many SIMD cycles does one wave64 instruction
// of each kind in the Philox / Box-Muller / Gamma chain cost when the chip is full (1 048 576 lanes, 4 waves per SIMD resident)?
//   hipcc -O3 --offload-arch=gfx950 -I tempest_amd/csrc tools/ubench_valu.hip -o scratch/ubench_valu && scratch/ubench_valu
// Output: one line per probe: us per launch and SIMD cycles per wave per call at the measured launch time (2.4 GHz nominal).
#include "common.h"
#include <cstdio>

constexpr int N = 1 << 20;
constexpr int ITER = 16;

template <int OP>
__global__ void __launch_bounds__(256) k_probe(double* __restrict__ out, uint64_t seed, double x0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  uint32_t iacc = 0;
  tph_rng g(seed, 1, 2, (uint64_t)i);
  double x = x0 + 1e-9 * (double)(i & 1023);
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    if (OP == 0) {                 // Philox4x32-10 alone
      tph_u4 r = tph_philox(g.item, (uint32_t)it, g.tick, g.tag, g.k0, g.k1);
      iacc ^= r.x ^ r.y ^ r.z ^ r.w;
    } else if (OP == 1) {          // uniform2 (Philox + 2 x 53-bit conversion)
      double a, b; g.uniform2((uint32_t)it, a, b); acc += a + b;
    } else if (OP == 2) {          // normal2 (lean log / sqrt)
      double a, b; g.normal2((uint32_t)it, a, b); acc += a + b;
    } else if (OP == 3) {          // library log
      x = log(x) * 1e-3 + x0; acc += x;
    } else if (OP == 4) {          // lean log
      x = tph_log(x) * 1e-3 + x0; acc += x;
    } else if (OP == 5) {          // library sqrt
      x = sqrt(x) * 1e-3 + x0; acc += x;
    } else if (OP == 6) {          // lean sqrt
      x = tph_sqrt(x) * 1e-3 + x0; acc += x;
    } else if (OP == 7) {          // sincospi
      double s, c; sincospi(x, &s, &c); x = (s + c) * 1e-3 + x0; acc += x;
    } else if (OP == 8) {          // 64 dependent-free FMAs (8 chains x 8)
      double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a0 = fma(a0, x0, 1.0); a1 = fma(a1, x0, 1.0); a2 = fma(a2, x0, 1.0); a3 = fma(a3, x0, 1.0);
        a4 = fma(a4, x0, 1.0); a5 = fma(a5, x0, 1.0); a6 = fma(a6, x0, 1.0); a7 = fma(a7, x0, 1.0);
      }
      x = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); acc += x; x = x * 1e-9 + x0;
    } else if (OP == 9) {          // library division
      x = (1.0 / x) * 1e-3 + x0; acc += x;
    } else if (OP == 10) {         // lean division
      x = tph_rcp(x) * 1e-3 + x0; acc += x;
    } else if (OP == 11) {         // 64 v_mad_u64_u32 (dependent pairs as in Philox)
      uint32_t c0 = g.item + it, c2 = i ^ 0x9e3779b9u;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        c0 = (uint32_t)(p1 >> 32) ^ (uint32_t)p0; c2 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1;
      }
      iacc ^= c0 ^ c2;
    } else if (OP == 12) {         // exp (library)
      x = exp(-x) * 1e-3 + x0; acc += x;
    } else if (OP == 13) {         // Gamma(shape 500005) Marsaglia-Tsang draw
      acc += tph_gamma_mt(tph_rng(seed, (uint32_t)it, 3, (uint64_t)i), 500005.0);
    }
  }
  out[i] = acc + (double)iacc;
}

template <int OP>
static void run(const char* name, double* out, double x0, int calls_per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_probe<OP>, dim3(N / 256), dim3(256), 0, 0, out, 1234567ull, x0);
  hipEventRecord(e0, 0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe<OP>, dim3(N / 256), dim3(256), 0, 0, out, 1234567ull, x0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double waves_per_simd = (double)N / 64 / 1024;
  const double cyc = us * 1e-6 * 2.4e9 / waves_per_simd / ITER / calls_per_iter;
  printf("%-28s %9.2f us/launch  %8.1f SIMD-cycles per wave per call (2.4 GHz nominal)\n", name, us, cyc);
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * N);
  run<0>("philox4x32-10", out, 0.5, 1);
  run<1>("uniform2", out, 0.5, 1);
  run<2>("normal2 (lean)", out, 0.5, 1);
  run<3>("log (ocml)", out, 0.3, 1);
  run<4>("log (lean)", out, 0.3, 1);
  run<5>("sqrt (ocml)", out, 3.0, 1);
  run<6>("sqrt (lean)", out, 3.0, 1);
  run<7>("sincospi (ocml)", out, 0.3, 1);
  run<8>("fma f64 x64", out, 0.999, 64);
  run<9>("1/x (ocml)", out, 3.0, 1);
  run<10>("1/x (lean)", out, 3.0, 1);
  run<11>("v_mad_u64_u32 x64", out, 0.5, 64);
  run<12>("exp (ocml)", out, 0.3, 1);
  run<13>("gamma_mt(5e5)", out, 0.3, 1);
  hipFree(out);
  return 0;
}
