// The chores around the redraw kernels of 16 < n_dim <= 112 (propose_sm.hip: the FP64 row walker; propose_mf.hip: the
// screened batches): pending moves, the Mahalanobis forms of tpCN through the blocked product of tri.h, the Gamma step
// scale, the redraw probe.  Reference: tempest/mcmc.py:225-236 (forms and scale), :163-194 (the deferred moves).
#pragma once
#include "common.h"
#include "tri.h"

// ---- Mahalanobis forms of a tile of 64 particles, |L^-1 (v - mu)|^2 through the blocked product of tri.h, and the chores
// around the stage-machine kernel.  MODE 0 (before it): resolve the deferred Metropolis moves (u <- u' where pending); tpCN:
// the form at u -> maha_u on the first step of a run (afterwards it is carried), then the Gamma draw and the step scale
// b = sigma sqrt(s) of every particle -> bfac_out (mcmc.py:228-236; one draw per particle and step, reused by its redraws).
// MODE 1 (after it): the form at u' -> maha_up (tpCN; 0 for RWM), and the step's mean attempts per particle -> state[8]
// (regime probe for the host).
template <int KERNEL, int WV, int MODE>
__global__ void __launch_bounds__(64 * WV) k_maha_tile(double* __restrict__ u, int64_t n, int64_t ld, int d,
                                                       const double* __restrict__ means, const double* __restrict__ Wb,
                                                       double* __restrict__ up, double* __restrict__ maha, tph_stepctl tick,
                                                       uint8_t* __restrict__ pend, const unsigned long long* __restrict__ queue,
                                                       const double* __restrict__ dof, const double* __restrict__ sigmas,
                                                       uint64_t seed, int64_t item0, double* __restrict__ bfac_out,
                                                       const int32_t* __restrict__ todo_cnt, const int32_t* __restrict__ todo_rows,
                                                       const int32_t* __restrict__ todo_off) {
  // todo_cnt != NULL: only the particles LISTED in todo_rows[0 .. *todo_cnt) (the closing pass behind a screened launch over
  // the blocked kernel's failures; one MODE's particles of a run with several proposal modes: the list then starts *todo_off
  // entries into todo_rows, and means / Wb / dof / sigmas are that mode's); a block beyond the list exits at once
  if (todo_off) todo_rows += *todo_off;
  extern __shared__ double sh[];
  double* xs = sh;                                 // [d][64]
  double* sc = sh + (size_t)d * 64;                // [WV][64]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t i = (int64_t)blockIdx.x * 64 + lane;
  bool live = i < n;
  if (todo_cnt) {
    const int64_t cnt = *todo_cnt;
    if ((int64_t)blockIdx.x * 64 >= cnt) return;          // the whole block (uniform)
    live = i < cnt;
    i = (int64_t)todo_rows[live ? i : 0];
  }
  const int64_t ii = live ? i : (todo_cnt ? i : n - 1);
  if (MODE == 1 && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl && queue)
    const_cast<double*>(tick.ctl)[8] = queue[2] ? (double)queue[1] / (double)queue[2] : 0.0;
#ifdef SM_PROFILE
  if (MODE == 1 && blockIdx.x == 0 && threadIdx.x == 0 && queue)
    printf("SM_PROFILE attempts %llu particles %llu | wave-cycles: bm %llu rows(z in LDS) %llu rows(z global) %llu in %llu steps, epilogue %llu (wait %llu + bounds %llu + rest) refill %llu | steps %llu\n", queue[1],
           queue[2], queue[3], queue[4], queue[10], queue[11], queue[7] + queue[8] + queue[9], queue[8], queue[9], queue[5], queue[6]);
#endif
  const bool form = KERNEL == TPH_KERNEL_TPCN && (MODE == 1 || !tick.carry());
  if (MODE == 0) {
    // (the rows of a lane's particle are requested eight at a time: one by one they are a chain of d / WV memory round trips)
    const bool pd = pend && live && pend[i];
    for (int j0 = wid; j0 < d; j0 += 8 * WV) {
      double uj[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = j0 + a * WV;
        uj[a] = j < d ? (pd ? up[(size_t)j * ld + i] : u[(size_t)j * ld + ii]) : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = j0 + a * WV;
        if (j < d) {
          if (pd) u[(size_t)j * ld + i] = uj[a];
          if (form) xs[(size_t)j * 64 + lane] = uj[a] - means[j];
        }
      }
    }
    __syncthreads();
    if (pd && wid == 0) pend[i] = 0;
  } else if (form) {
    for (int j0 = wid; j0 < d; j0 += 8 * WV) {
      double v[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = j0 + a * WV;
        v[a] = j < d ? up[(size_t)j * ld + ii] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = j0 + a * WV;
        if (j < d) xs[(size_t)j * 64 + lane] = v[a] - means[j];
      }
    }
    __syncthreads();
  }
  if (KERNEL != TPH_KERNEL_TPCN) {
    if (maha && wid == 0 && live) maha[i] = 0.0;
    return;
  }
  double m = 0.0;
  if (form) {
    double part = 0.0;
    tri_apply(Wb, d, xs, lane, wid, WV, [&](int, double y) { part = fma(y, y, part); });
    sc[(size_t)wid * 64 + lane] = part;
    __syncthreads();
    if (wid == 0) {
      for (int w = 0; w < WV; ++w) m += sc[(size_t)w * 64 + lane];
      if (live) maha[i] = m;
    }
  } else if (wid == 0) {
    m = maha[ii];
  }
  if (MODE == 0 && wid == 0 && live) {
    const double nu = dof[0], sigma = sigmas[0];
    tph_rng gr(seed, tick, TPH_TAG_GAMMA, (uint64_t)(item0 + i));
    const double gam = tph_gamma_mt(gr, 0.5 * ((double)d + nu)) * tph_div(2.0, nu + m);
    bfac_out[i] = sigma * tph_sqrt(tph_rcp(gam));
  }
}

template <int KERNEL, int MODE>
static int launch_maha_tile(tph_ctx* ctx, double* u, int64_t n, int64_t ld, const double* means, const double* Wb, double* up,
                            double* maha, tph_stepctl tick, uint8_t* pend, const unsigned long long* queue, const double* dof,
                            const double* sigmas, uint64_t seed, int64_t item0, double* bfac_out, const int32_t* todo_cnt = nullptr,
                            const int32_t* todo_rows = nullptr, const int32_t* todo_off = nullptr) {
  const int d = ctx->d;
  const int wv = d <= 32 ? 4 : d <= 64 ? 8 : 16;
  const size_t lds = sizeof(double) * ((size_t)d * 64 + (size_t)wv * 64);
  const dim3 grid((unsigned)((n + 63) / 64));
#define TPH_MT(WV)                                                                                                       \
  do {                                                                                                                   \
    if (lds > 64 * 1024)                                                                                                 \
      TPH_HIP(hipFuncSetAttribute((const void*)k_maha_tile<KERNEL, WV, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_maha_tile<KERNEL, WV, MODE>), grid, dim3(64 * WV), lds, ctx->stream, u, n, ld, d, means, Wb, up, maha, \
                       tick, pend, queue, dof, sigmas, seed, item0, bfac_out, todo_cnt, todo_rows, todo_off);            \
  } while (0)
  if (wv == 4) TPH_MT(4); else if (wv == 8) TPH_MT(8); else TPH_MT(16);
#undef TPH_MT
  TPH_LAUNCH_CHECK();
  return 0;
}

