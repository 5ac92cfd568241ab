// One lockstep attempt of every (listed) particle at 16 < d <= 112, one mode, with BOTH triangular products on the FP64 matrix
// cores: the round kernel of the blocked proposal path (mutate.hip: launch_propose_blk drives the rounds, the lists and the
// straggler pass).
// Reference: tempest/mcmc.py:225-249 (tpCN), :301-312 (RWM).
//
// k_propose_blk evaluates an attempt with lane = particle and the matrix element as a scalar operand: one 64-byte scalar
// load per 8 FMAs -- 11 % of the FP64 vector peak at 262 144 x 100-D, the scalar cache is the limiter.  L z for a TILE of
// particles is a triangular matrix product, and v_mfma_f64_16x16x4_f64 issues it at the full FP64 rate with ONE 8-byte
// operand per lane and instruction:
//   * a wave owns 16 particles (the columns n = lane & 15 of the tile); lane (k = lane >> 4, n) holds, for every step s of the
//     contraction, ONE normal of particle n as its B operand.  The contraction index is PERMUTED so that no normal ever changes
//     lanes: lane (k, n) draws the Box-Muller pairs q = k, k + 4, k + 8, ... of particle n (the same Philox blocks as every other
//     kernel), pair q = k + 4c is rows 8c + 2k and 8c + 2k + 1, and those are this lane's operands of steps s = 2c and 2c + 1;
//     the copy of L is blocked to match (block (p, s): rows 16p .., columns {8c + 2k' (+1)}: k_blkm_pack);
//   * rows 16p .. 16p+15 of the tile are one accumulator of 4p + 4 instructions (L is lower-triangular); the panels run from
//     the LAST one down, so that the operands of steps 4p .. 4p+3 are dead once panel p is done and the panel's results --
//     lane (k, n) holds rows 16p + k + 4i, i = 0..3 -- overwrite them IN PLACE: they are exactly the B operands of steps 4p + i
//     of the next product, |L^-1 (u' - mu)|^2, in the natural order of the contraction index (no lane movement, no LDS);
//   * no LDS at all: ~110 VGPRs, four waves per SIMD; the matrix blocks stream from L2 (512 bytes per instruction and wave).
// Same draws, same formulas as the other kernels (v = fma(b, (L z)_r, mu_r + a (u_r - mu_r)) as in the row walker); the
// summation order inside a row is the matrix core's, so proposals agree with the other kernels to rounding.
#include "common.h"

typedef double bm_d4 __attribute__((ext_vector_type(4)));

__host__ __device__ static inline int bm_panels(int d) { return (d + 15) / 16; }
__host__ __device__ static inline int bm_blocks(int np) { return 2 * np * (np + 1); }       // sum over p of 4p + 4
__host__ __device__ static inline int bm_blk(int p, int s) { return 2 * p * (p + 1) + s; }

// Tb[blk(p, s)][lane] = T[16p + (lane & 15)][col(s, lane >> 4)], zero outside the lower triangle / beyond d.
// perm = 1: col(s, k) = 8 (s >> 1) + 2k + (s & 1) (the Box-Muller layout of L z); perm = 0: col = 4s + k (natural: L^-1 x)
static __global__ void __launch_bounds__(256) k_blkm_pack(const double* __restrict__ T, int d, int np, int perm, double* __restrict__ Tb) {
  const int total = bm_blocks(np) * 64;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int b = e >> 6, lane = e & 63;
    int p = 0;
    while (bm_blk(p + 1, 0) <= b) ++p;
    const int s = b - bm_blk(p, 0), k = lane >> 4;
    const int r = 16 * p + (lane & 15), c = perm ? 8 * (s >> 1) + 2 * k + (s & 1) : 4 * s + k;
    Tb[e] = (r < d && c <= r) ? T[(size_t)r * d + c] : 0.0;
  }
}

template <int KERNEL, int NP, bool HAS_BC>
__global__ void __launch_bounds__(256) k_propose_blkm(double* __restrict__ u, int64_t n, int64_t ld, int d, const double* __restrict__ means,
                                                      const double* __restrict__ Lm, const double* __restrict__ Wm,
                                                      const double* __restrict__ dof, const double* __restrict__ sigmas,
                                                      const uint8_t* __restrict__ bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                                                      double* __restrict__ up, double* __restrict__ maha_u, double* __restrict__ maha_up,
                                                      uint8_t* __restrict__ pend, const int32_t* __restrict__ cnt_in,
                                                      const int32_t* __restrict__ rows_in, int att, int32_t* __restrict__ cnt_out,
                                                      int32_t* __restrict__ rows_out) {
  // cnt_in == NULL: round 0, attempt 0 of every particle and the chores of the step (pending moves, form at u, Gamma scale);
  // cnt_in != NULL: attempt `att` of the particles listed by the round before; the step scale comes from where round 0 parked it
  constexpr int NS = 4 * NP;
  __shared__ int s_fail[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = lane >> 4, nn = lane & 15;
  const bool first = cnt_in == nullptr;
  int64_t slot = ((int64_t)blockIdx.x * 4 + wid) * 16 + nn;
  int64_t total = n;
  if (!first) {
    total = *cnt_in;
    if (att == 1 && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl)      // the redraw probe from ALL first attempts
      const_cast<double*>(tick.ctl)[8] = total < n ? (double)n / (double)(n - total) : 256.0;
    if ((int64_t)blockIdx.x * 64 >= total) return;                        // the whole block (uniform): nothing listed for it
  }
  // (a wave beyond the list inside the last block runs along on a shadow particle: the block meets at two barriers below)
  const bool live = slot < total;
  const int64_t i = first ? (live ? slot : n - 1) : (int64_t)rows_in[live ? slot : 0];     // dead columns shadow a particle, never store
  const int npairs = (d + 1) >> 1;
  const double sigma = sigmas[0];
  const bool carry = KERNEL == TPH_KERNEL_TPCN && tick.carry();
  const double a_fac = (KERNEL == TPH_KERNEL_TPCN) ? tph_sqrt(1.0 - sigma * sigma) : 1.0;
  double X[NS];                                  // the B operands: normals (permuted steps), then rows of the proposal (natural steps)

  // sum over the rows of |T x|^2 for the tile (x in X as natural-step operands): every lane ends with ITS column's value
  auto form = [&](const double* __restrict__ Tb) -> double {
    double part = 0.0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      bm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4 * p + 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Tb[(size_t)bm_blk(p, s) * 64 + lane], X[s], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fma(acc[q], acc[q], part);
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    return part;
  };

  // ---- round 0: pending accepted move (deferred tph_accept), form at u (first step of a run; afterwards carried), Gamma scale
  double b_fac = sigma;
  if (first) {
    const bool pd = pend && live && pend[i];
    if (pd || (KERNEL == TPH_KERNEL_TPCN && !carry)) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int r = 4 * s + k;
        double uj = 0.0;
        if (r < d) {
          uj = u[(size_t)r * ld + i];
          if (pd) { uj = up[(size_t)r * ld + i]; u[(size_t)r * ld + i] = uj; }
          uj -= (KERNEL == TPH_KERNEL_TPCN) ? means[r] : 0.0;
        }
        X[s] = uj;
      }
    }
    if (KERNEL == TPH_KERNEL_TPCN) {
      double m_u;
      if (carry) {
        m_u = maha_u[i];
      } else {
        m_u = form(Wm);
        if (live && k == 0 && maha_u) maha_u[i] = m_u;
      }
      const double nu = dof[0];
      tph_rng gg(seed, tick, TPH_TAG_GAMMA, (uint64_t)(item0 + i));
      const double gam = tph_gamma_mt(gg, 0.5 * ((double)d + nu)) * tph_div(2.0, nu + m_u);
      b_fac = sigma * tph_sqrt(tph_rcp(gam));
      if (live && k == 0) maha_up[i] = b_fac;        // parked for the later rounds of this particle
    } else if (live && k == 0 && maha_u) {
      maha_u[i] = 0.0;
    }
    // every lane of a column has read the flag (four lanes, one wave: the store below is ordered behind their loads by the
    // s_waitcnt of the loads' first use above)
    if (pd && k == 0) pend[i] = 0;
  } else if (KERNEL == TPH_KERNEL_TPCN) {
    b_fac = maha_up[i];
  }

  // ---- the normals of this round's attempt: lane (k, n) draws pairs k, k + 4, ... of particle n = its operands of steps 2c, 2c + 1
  {
    tph_rng gz(seed, tick, TPH_TAG_NORMAL, (uint64_t)(item0 + i));
    const uint32_t d0 = (uint32_t)att * (uint32_t)npairs;
#pragma unroll
    for (int c = 0; c < NS / 2; ++c) {
      const int q = k + 4 * c;
      double z0 = 0.0, z1 = 0.0;
      if (q < npairs) gz.normal2(d0 + (uint32_t)q, z0, z1);
      X[2 * c] = z0;
      X[2 * c + 1] = z1;             // (row 2q + 1 == d for odd d: its column of L is zero)
    }
  }
  // ---- rows of the attempt, last panel first: v = fma(b, (L z)_r, mu_r + a (u_r - mu_r)), bounds; results in place
  bool ok = true;
#pragma unroll
  for (int p = NP - 1; p >= 0; --p) {
    bm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * p + 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lm[(size_t)bm_blk(p, s) * 64 + lane], X[s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 16 * p + k + 4 * q;
      double v = 0.0;
      if (r < d) {
        const double ur = u[(size_t)r * ld + i];
        const double mr = (KERNEL == TPH_KERNEL_TPCN) ? means[r] : 0.0;
        const double base = (KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, ur - mr, mr) : ur;
        v = fma(b_fac, acc[q], base);
        const uint8_t f = HAS_BC ? bc[r] : (uint8_t)TPH_BC_STRICT;
        if (f == TPH_BC_PERIODIC) v = bc_periodic(v);
        else if (f == TPH_BC_REFLECTIVE) v = bc_reflective(v);
        else ok = ok && (v >= 0.0) && (v <= 1.0);
      }
      X[4 * p + q] = v;
    }
  }
  // a column is in bounds when its four lanes are
  const unsigned long long okb = __ballot(ok);
  const unsigned int okc = (unsigned int)(okb & (okb >> 16) & (okb >> 32) & (okb >> 48)) & 0xFFFFu;
  const bool all_ok = (okc >> nn) & 1u;
  // ---- outputs of the particles whose attempt is in bounds; the others are listed for the next round / the straggler pass
  if (live && all_ok) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int r = 4 * s + k;
      if (r < d) up[(size_t)r * ld + i] = X[s];
    }
  }
  {
    // the block's failures take ONE slot range of the list (an atomic per wave -- 16 384 of them on one address at 262 144
    // particles -- cost 110 us of a 195 us launch with a fifth of the first attempts out of bounds)
    const unsigned long long failb = __ballot(live && !all_ok && k == 0);
    const int nf = __popcll(failb);
    if (lane == 0) s_fail[wid] = nf | (__popcll(__ballot(live && k == 0)) << 8);
    __syncthreads();
    int before = 0, nfail = 0, nlive = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int f = s_fail[w] & 255;
      before += w < wid ? f : 0;
      nfail += f;
      nlive += s_fail[w] >> 8;
    }
    if (threadIdx.x == 0 && nfail) s_base = atomicAdd(cnt_out, nfail);
    __syncthreads();
    if (live && !all_ok && k == 0) rows_out[s_base + before + __popcll(failb & ((1ull << lane) - 1ull))] = (int32_t)i;
    if (first && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl) {      // regime probe: mean attempts implied by this block's failures, 1 / (1 - f)
      const double f = (double)nfail / fmax(1.0, (double)nlive);
      const_cast<double*>(tick.ctl)[8] = f < 0.99 ? 1.0 / (1.0 - f) : 100.0;
    }
  }
  if (KERNEL == TPH_KERNEL_TPCN) {
    if (okc == 0u) return;                          // (wave-uniform) no column needs its form
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int r = 4 * s + k;
      X[s] = r < d ? X[s] - means[r] : 0.0;
    }
    const double m_up = form(Wm);
    if (live && all_ok && k == 0) maha_up[i] = m_up;
  } else if (live && all_ok && k == 0 && maha_up) {
    maha_up[i] = 0.0;
  }
}

// One round of the matrix-core blocked kernel on the ctx stream (mutate.hip drives the rounds).  The blocked copies of L (permuted
// steps) and L^-1 (natural steps) live in a ctx-owned buffer, rebuilt when the caller's mode statistics change (and always under
// stream capture: a replayed step never re-enters this host code).
template <int KERNEL>
static int blkm_round(tph_ctx* ctx, double* u, int64_t n, int64_t ld, const double* means, const double* chol, const double* winv,
                      const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                      double* up, double* mu_, double* mup, uint8_t* pend, const int32_t* cnt_in, const int32_t* rows_in, int att,
                      int32_t* cnt_out, int32_t* rows_out) {
  const int d = ctx->d, np = bm_panels(d);
  TPH_REQUIRE(d > 16 && d <= 112, "tph_propose (blocked, matrix cores): n_dim=%d outside 17..112", d);
  const size_t one = sizeof(double) * (size_t)bm_blocks(np) * 64;
  if (ctx->bm_bytes < 2 * one) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->bm_buf) ctx->retired.push_back(ctx->bm_buf);
    ctx->bm_buf = nullptr; ctx->bm_bytes = 0; ctx->bm_epoch = -1;
    TPH_HIP(hipMalloc((void**)&ctx->bm_buf, 2 * one));
    ctx->bm_bytes = 2 * one;
  }
  double* Lm = (double*)ctx->bm_buf;
  double* Wm = Lm + (size_t)bm_blocks(np) * 64;
  if (att == 0) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    TPH_HIP(hipStreamIsCapturing(ctx->stream, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (capturing || ctx->modes_epoch <= 0 || ctx->bm_epoch != ctx->modes_epoch || ctx->bm_src != (const void*)chol || ctx->bm_kernel != KERNEL) {
      hipLaunchKernelGGL(k_blkm_pack, dim3(1), dim3(256), 0, ctx->stream, chol, d, np, 1, Lm);
      if (KERNEL == TPH_KERNEL_TPCN) hipLaunchKernelGGL(k_blkm_pack, dim3(1), dim3(256), 0, ctx->stream, winv, d, np, 0, Wm);
      ctx->bm_epoch = capturing ? -1 : ctx->modes_epoch; ctx->bm_src = (const void*)chol; ctx->bm_kernel = KERNEL;
    }
  }
  const dim3 grid((unsigned)((n + 63) / 64));
#define TPH_BM(NPV, BC)                                                                                                  \
  hipLaunchKernelGGL((k_propose_blkm<KERNEL, NPV, BC>), grid, dim3(256), 0, ctx->stream, u, n, ld, d, means, (const double*)Lm, \
                     (const double*)Wm, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend, cnt_in, rows_in, att, cnt_out, rows_out)
#define TPH_BM_NP(NPV) do { if (bc) TPH_BM(NPV, true); else TPH_BM(NPV, false); } while (0)
  switch (np) {
    case 2: TPH_BM_NP(2); break;
    case 3: TPH_BM_NP(3); break;
    case 4: TPH_BM_NP(4); break;
    case 5: TPH_BM_NP(5); break;
    case 6: TPH_BM_NP(6); break;
    default: TPH_BM_NP(7); break;
  }
#undef TPH_BM_NP
#undef TPH_BM
  TPH_LAUNCH_CHECK();
  return 0;
}

int tph_blkm_round(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                   const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                   const double* ctl, int64_t item0, double* up, double* mu_, double* mup, uint8_t* pend, const int32_t* cnt_in,
                   const int32_t* rows_in, int att, int32_t* cnt_out, int32_t* rows_out) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return blkm_round<TPH_KERNEL_TPCN>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend, cnt_in,
                                       rows_in, att, cnt_out, rows_out);
  return blkm_round<TPH_KERNEL_RWM>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend, cnt_in,
                                    rows_in, att, cnt_out, rows_out);
}
