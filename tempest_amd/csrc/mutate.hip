// MCMC mutation of the active particle set.
// Reference: tempest/steps/mutate.py:76-200 (Mutator.run), tempest/mcmc.py:104-208 (runner loop,
// adaptive step count), :211-288 (tpCN), :291-323 (RWM), :326-411 (boundary conditions).
#include "common.h"
#include <stdlib.h>
#include "tri.h"
#include "p2p.h"

// ------------------------------------------------------------------------------- prior draw (beta=0)
// u ~ U(0,1)^d (mutate.py:102): one Philox call per coordinate pair.
__global__ void __launch_bounds__(256) k_prior_draw(double* __restrict__ u, int64_t n, int64_t ld, int d, uint64_t seed,
                                                    uint32_t tick, int64_t item0) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int p = blockIdx.y;
  if (i >= n) return;
  tph_rng g(seed, tick, TPH_TAG_PRIOR, (uint64_t)(item0 + i));
  double a, b;
  g.uniform2((uint32_t)p, a, b);
  u[(size_t)(2 * p) * ld + i] = a;
  if (2 * p + 1 < d) u[(size_t)(2 * p + 1) * ld + i] = b;
}

extern "C" int tph_prior_draw(tph_ctx* ctx, double* u_dev, int64_t n, int64_t ld, uint64_t seed, uint32_t tick,
                              int64_t item0) {
  TPH_REQUIRE(ctx && u_dev && n > 0 && ld >= n, "tph_prior_draw: bad argument");
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)((ctx->d + 1) / 2));
  hipLaunchKernelGGL(k_prior_draw, grid, dim3(256), 0, ctx->stream, u_dev, n, ld, ctx->d, seed, tick, item0);
  TPH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------- +-inf repair (mutate.py:122-148)
__global__ void __launch_bounds__(256) k_flag_finite(const double* __restrict__ logl, int64_t n, double* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = isinf(logl[i]) ? 0.0 : 1.0;
}
__global__ void __launch_bounds__(256) k_finite_list(const double* __restrict__ flag, const double* __restrict__ rank,
                                                     int64_t n, int64_t* __restrict__ list) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i] != 0.0) list[(int64_t)rank[i] - 1] = i;
}
__global__ void __launch_bounds__(256) k_inf_repair(double* __restrict__ u, double* __restrict__ x, double* __restrict__ logl,
                                                    int64_t n, int64_t ld, int d, const double* __restrict__ flag,
                                                    const double* __restrict__ rank, const int64_t* __restrict__ list,
                                                    uint64_t seed, uint32_t tick, int64_t item0, double* __restrict__ stats,
                                                    int64_t* __restrict__ src) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n_fin = (int64_t)rank[n - 1];
  if (i == 0) { stats[0] = (double)n_fin; stats[1] = (double)n; }
  if (i < n && src) src[i] = i;
  if (i >= n || flag[i] != 0.0 || n_fin == 0) return;
  tph_rng g(seed, tick, TPH_TAG_REPAIR, (uint64_t)(item0 + i));
  double U, U1;
  g.uniform2(0, U, U1);
  int64_t pick = (int64_t)(U * (double)n_fin);
  if (pick > n_fin - 1) pick = n_fin - 1;
  int64_t s = list[pick];  // finite rows are never written, infinite rows never read: no hazard
  for (int j = 0; j < d; ++j) {
    u[(size_t)j * ld + i] = u[(size_t)j * ld + s];
    x[(size_t)j * ld + i] = x[(size_t)j * ld + s];
  }
  logl[i] = logl[s];
  if (src) src[i] = s;
}

extern "C" int tph_inf_repair_src(tph_ctx* ctx, double* u_dev, double* x_dev, double* logl_dev, int64_t n, int64_t ld,
                                  uint64_t seed, uint32_t tick, int64_t item0, double* stats_dev, int64_t* src_dev) {
  TPH_REQUIRE(ctx && u_dev && x_dev && logl_dev && stats_dev && n > 0 && ld >= n, "tph_inf_repair: bad argument");
  // layout inside the big scratch: [tile sums (used by tph_cdf)] ... we take our arrays after a 1 MiB offset
  size_t tiles_bytes = sizeof(double) * (size_t)((n + 2047) / 2048 + 1);
  size_t off = (tiles_bytes + 255) / 256 * 256;
  size_t need = off + sizeof(double) * 3 * (size_t)n;
  if (tph_scratch_reserve(ctx, need)) return -1;
  double* flag = (double*)((char*)ctx->scratch + off);
  double* rank = flag + n;
  int64_t* list = (int64_t*)(rank + n);
  unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_flag_finite, dim3(grid), dim3(256), 0, ctx->stream, logl_dev, n, flag);
  int rc = tph_cdf_plain(ctx, flag, n, nullptr, rank);  // scratch already large enough: no reallocation
  if (rc) return rc;
  hipLaunchKernelGGL(k_finite_list, dim3(grid), dim3(256), 0, ctx->stream, flag, rank, n, list);
  hipLaunchKernelGGL(k_inf_repair, dim3(grid), dim3(256), 0, ctx->stream, u_dev, x_dev, logl_dev, n, ld, ctx->d, flag, rank,
                     list, seed, tick, item0, stats_dev, src_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}
extern "C" int tph_inf_repair(tph_ctx* ctx, double* u_dev, double* x_dev, double* logl_dev, int64_t n, int64_t ld,
                              uint64_t seed, uint32_t tick, int64_t item0, double* stats_dev) {
  return tph_inf_repair_src(ctx, u_dev, x_dev, logl_dev, n, ld, seed, tick, item0, stats_dev, nullptr);
}

// ------------------------------------------------------------------------------------------- proposals
constexpr int PROP_THREADS = 64;

// One lane per particle; its z[d] and diff[d] columns live in LDS as [d][64] (conflict-free).
// tpCN (mcmc.py:225-249): m = diff^T S^-1 diff ; s = 1/Gamma((d+nu)/2, 2/(nu+m)) (one draw, reused across
// redraws) ; u' = mu + sqrt(1-sigma^2) diff + sigma sqrt(s) L z, redrawn until the strict dims lie in [0,1].
// RWM (mcmc.py:301-312): u' = u + sigma L z.
// Mahalanobis forms as |W (v - mu)|^2 with W = L^-1 (lower-triangular).
// HBM per particle: read 8d+4, write 8d+16; FLOP 2*(d^2/2) per attempt (+ 2*(d^2/2) per Mahalanobis form).
template <int KERNEL>
__global__ void __launch_bounds__(PROP_THREADS) k_propose(double* __restrict__ u, const int32_t* __restrict__ assign,
                                                          int64_t n, int64_t ld, int d, const double* __restrict__ means,
                                                          const double* __restrict__ chol, const double* __restrict__ winv,
                                                          const double* __restrict__ dof, const double* __restrict__ sigmas,
                                                          const uint8_t* __restrict__ bc, uint64_t seed, tph_stepctl tick,
                                                          int64_t item0, double* __restrict__ up,
                                                          double* __restrict__ maha_u, double* __restrict__ maha_up,
                                                          uint8_t* __restrict__ pend) {
  extern __shared__ double sh[];
  const int tid = threadIdx.x;
  double* zs = sh + tid;                       // zs[j*64]
  double* df = sh + (size_t)d * PROP_THREADS + tid;  // df[j*64]
  int64_t i = (int64_t)blockIdx.x * PROP_THREADS + tid;
  if (i >= n) return;
  const int c = assign ? assign[i] : 0;
  const double* __restrict__ mu = means + (size_t)c * d;
  const double* __restrict__ L = chol + (size_t)c * d * d;
  const double* __restrict__ W = winv + (size_t)c * d * d;
  const double sigma = sigmas[c];

  const bool pd = pend && pend[i];      // the previous step's move was accepted and is still pending (deferred tph_accept)
  for (int j = 0; j < d; ++j) {
    double uj = u[(size_t)j * ld + i];
    if (pd) { uj = up[(size_t)j * ld + i]; u[(size_t)j * ld + i] = uj; }
    df[j * PROP_THREADS] = (KERNEL == TPH_KERNEL_TPCN) ? uj - mu[j] : uj;
  }
  if (pd) pend[i] = 0;
  double a_fac = 1.0, b_fac = sigma, m_u = 0.0;
  if (KERNEL == TPH_KERNEL_TPCN) {
    if (tick.carry()) {
      m_u = maha_u[i];
    } else {
      for (int r = 0; r < d; ++r) {
        double acc = 0.0;
        for (int j = 0; j <= r; ++j) acc = fma(W[r * d + j], df[j * PROP_THREADS], acc);
        m_u = fma(acc, acc, m_u);
      }
    }
    const double nu = dof[c];
    tph_rng gg(seed, tick, TPH_TAG_GAMMA, (uint64_t)(item0 + i));
    const double gam = tph_gamma_mt(gg, 0.5 * ((double)d + nu)) * tph_div(2.0, nu + m_u);
    a_fac = tph_sqrt(1.0 - sigma * sigma);
    b_fac = sigma * tph_sqrt(tph_rcp(gam));
  }
  tph_rng gz(seed, tick, TPH_TAG_NORMAL, (uint64_t)(item0 + i));
  const int npairs = (d + 1) >> 1;
  bool ok = false;
  for (int att = 0; att < PROP_MAX_ATTEMPTS && !ok; ++att) {
    for (int p = 0; p < npairs; ++p) {
      double z0, z1;
      gz.normal2((uint32_t)(att * npairs + p), z0, z1);
      zs[(2 * p) * PROP_THREADS] = z0;
      if (2 * p + 1 < d) zs[(2 * p + 1) * PROP_THREADS] = z1;
    }
    ok = true;
    for (int r = d - 1; r >= 0; --r) {  // descending: slot r is free once row r is done
      double acc = 0.0;
      for (int j = 0; j <= r; ++j) acc = fma(L[r * d + j], zs[j * PROP_THREADS], acc);
      double v;
      if (KERNEL == TPH_KERNEL_TPCN) v = mu[r] + a_fac * df[r * PROP_THREADS] + b_fac * acc;
      else v = df[r * PROP_THREADS] + b_fac * acc;
      const uint8_t f = bc ? bc[r] : (uint8_t)TPH_BC_STRICT;
      if (f == TPH_BC_PERIODIC) v = bc_periodic(v);
      else if (f == TPH_BC_REFLECTIVE) v = bc_reflective(v);
      else ok = ok && (v >= 0.0) && (v <= 1.0);
      zs[r * PROP_THREADS] = v;
    }
  }
  if (!ok) {  // the reference would loop forever; after 256 redraws we propose the current point
    for (int j = 0; j < d; ++j)
      zs[j * PROP_THREADS] = (KERNEL == TPH_KERNEL_TPCN) ? df[j * PROP_THREADS] + mu[j] : df[j * PROP_THREADS];
  }
  double m_up = 0.0;
  for (int j = 0; j < d; ++j) up[(size_t)j * ld + i] = zs[j * PROP_THREADS];
  if (KERNEL == TPH_KERNEL_TPCN) {
    for (int j = 0; j < d; ++j) zs[j * PROP_THREADS] -= mu[j];
    for (int r = 0; r < d; ++r) {
      double acc = 0.0;
      for (int j = 0; j <= r; ++j) acc = fma(W[r * d + j], zs[j * PROP_THREADS], acc);
      m_up = fma(acc, acc, m_up);
    }
  }
  if (maha_u) maha_u[i] = m_u;
  if (maha_up) maha_up[i] = m_up;
}

// ---- multi-lane proposals: LPP lanes cooperate on one particle ---------------------------------------------
// The active set (1e5-1e6 particles) is far smaller than the chip's 524 288 thread slots and each particle needs
// O(d^2) FP64 work plus ~d/2 Philox/Box-Muller draws, so one lane per particle leaves the SIMDs latency-bound.
// Here a group of LPP lanes (a power of two <= 64, inside one wave) shares a particle: lane q draws normal pair q
// (and two lanes draw the Gamma candidate's normal and uniform in the same instruction stream), rows of the
// Sigma^-1 / L products are dealt round-robin to the lanes, partial sums meet by xor-shuffles.  The draws are the
// same counter-based ones as in the one-lane kernels, so results agree to rounding (different summation tree).
constexpr int ML_THREADS = 256;

template <int LPP>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPP>
__device__ __forceinline__ int group_and(int v) {
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) v &= __shfl_xor(v, o, 64);
  return v;
}

template <int KERNEL, int LPP, int STAGE>
__global__ void __launch_bounds__(ML_THREADS) k_propose_ml(double* __restrict__ u, const int32_t* __restrict__ assign,
                                                           int64_t n, int64_t ld, int d, const double* __restrict__ means,
                                                           const double* __restrict__ chol, const double* __restrict__ winv,
                                                           const double* __restrict__ dof, const double* __restrict__ sigmas,
                                                           const uint8_t* __restrict__ bc, uint64_t seed, tph_stepctl tick,
                                                           int64_t item0, double* __restrict__ up,
                                                           double* __restrict__ maha_u, double* __restrict__ maha_up,
                                                           uint8_t* __restrict__ pend, const int32_t* __restrict__ todo,
                                                           const int32_t* __restrict__ todo_rows, int att0,
                                                           const int32_t* __restrict__ att_by_mode) {
  // todo != NULL: straggler pass behind k_propose_blk -- only the particles it LISTED (todo[0] = count, todo_rows[] = the rows
  // whose attempts 0 .. att0-1 left the cube; a block beyond the list exits at once: 16 384 blocks scanning flags cost 69 us at
  // 262 144 x 32-D), starting from attempt att0; everything else of the step (pending moves, the others' proposals) is done already
  extern __shared__ double sh[];
  constexpr int PPB = ML_THREADS / LPP;
  const int l = threadIdx.x % LPP, p = threadIdx.x / LPP;
  const int dp = d | 1;                               // odd per-particle stride: particles land on different LDS banks
  double* zs = sh + (size_t)p * dp;                   // normals of the current attempt
  double* df = sh + (size_t)PPB * dp + (size_t)p * dp;  // u - mu (tpCN) or u (RWM)
  double* vs = sh + (size_t)2 * PPB * dp + (size_t)p * dp;   // rows of the current attempt; the proposal once it is complete
  // (W = L^-1, lower-triangular: the Mahalanobis forms are |W (v - mu)|^2)
  // STAGE 1: W and L both resident in LDS (rows padded to d+1); STAGE 2: one matrix slot, refilled per phase;
  // STAGE 0: read from global/L2 (several modes, or matrices too large).  Staging amortises the matrix reads over the
  // PPB particles of the block instead of re-reading d*d doubles per particle.
  double* mat0 = sh + (size_t)3 * PPB * dp;
  double* mat1 = mat0 + (size_t)d * (d + 1);
  int64_t i = (int64_t)blockIdx.x * PPB + p;
  bool live = i < n;
  if (todo) {
    const int64_t cnt = todo[0];
    if ((int64_t)blockIdx.x * PPB >= cnt) return;     // the whole block (uniform): nothing listed for it
    live = i < cnt;
    i = live ? (int64_t)todo_rows[i] : n - 1;
  }
  const int64_t ii = i < n ? i : n - 1;              // dead groups shadow a particle, never store
  const int c = (STAGE == 0 && assign) ? assign[ii] : 0;
  const double* __restrict__ mu = means + (size_t)c * d;
  const double* __restrict__ Lg = chol + (size_t)c * d * d;
  const double* __restrict__ Pg = winv + (size_t)c * d * d;
  const double sigma = sigmas[c];
  const int npairs = (d + 1) >> 1;
  const int ms = STAGE == 0 ? d : d + 1;              // row stride of the matrices as read below
  const double* P = STAGE == 0 ? Pg : mat0;
  const double* L = STAGE == 0 ? Lg : (STAGE == 1 ? mat1 : mat0);
  auto stage = [&](double* dst, const double* src) {
    for (int e = threadIdx.x; e < d * d; e += ML_THREADS) dst[(e / d) * (d + 1) + (e % d)] = src[e];
  };
  const bool carry = KERNEL == TPH_KERNEL_TPCN && tick.carry();   // maha_u already holds the form at u (see tph_stepctl)
  if (STAGE != 0) {
    if (KERNEL == TPH_KERNEL_TPCN && (STAGE == 1 || !carry)) stage(mat0, Pg);
    if (STAGE == 1 || KERNEL != TPH_KERNEL_TPCN || carry) stage(STAGE == 1 ? mat1 : mat0, Lg);
  }

  const bool pd = pend && live && pend[i];      // pending accepted move of the previous step (deferred tph_accept)
  for (int j = l; j < d; j += LPP) {
    double uj = u[(size_t)j * ld + ii];
    if (pd) { uj = up[(size_t)j * ld + i]; u[(size_t)j * ld + i] = uj; }
    df[j] = (KERNEL == TPH_KERNEL_TPCN) ? uj - mu[j] : uj;
  }
  __syncthreads();
  if (pd && l == 0) pend[i] = 0;                 // every lane of the group has read the flag before the barrier
  double a_fac = 1.0, b_fac = sigma, m_u = 0.0;
  double nu = 0.0, gshape = 1.0;
  if (KERNEL == TPH_KERNEL_TPCN) {
    if (carry) {
      m_u = maha_u[ii];
    } else {
      double part = 0.0;
      for (int r = l; r < d; r += LPP) {
        const double* Pr = P + (size_t)r * ms;
        double acc = 0.0;
        for (int j = 0; j <= r; ++j) acc = fma(Pr[j], df[j], acc);
        part = fma(acc, acc, part);
      }
      m_u = group_sum<LPP>(part);
    }
    nu = dof[c];
    gshape = 0.5 * ((double)d + nu);
    a_fac = tph_sqrt(1.0 - sigma * sigma);
  }
  // first attempt: normal pairs (tag NORMAL, draws 0..npairs-1) and, for tpCN, the Gamma candidate (tag GAMMA, draw 0: one
  // Philox call) are generated by different lanes at once
  const uint64_t item = (uint64_t)(item0 + ii);
  tph_rng gz(seed, tick, TPH_TAG_NORMAL, item);
  tph_rng gg(seed, tick, TPH_TAG_GAMMA, item);
  double g_x = 0.0, g_logu = 0.0;
  const int nq = npairs + (KERNEL == TPH_KERNEL_TPCN ? 1 : 0);
  for (int q = todo ? npairs + l : l; q < nq; q += LPP) {     // straggler pass: attempt 0 is over, only the Gamma candidate
    if (q < npairs) {
      double z0, z1;
      gz.normal2((uint32_t)q, z0, z1);
      zs[2 * q] = z0;
      if (2 * q + 1 < d) zs[2 * q + 1] = z1;
    } else {
      gg.gamma_candidate(0u, g_x, g_logu);
    }
  }
  if (KERNEL == TPH_KERNEL_TPCN) {
    // bring the candidate's two pieces to every lane of the group (owner: lane npairs % LPP; the others hold zeros)
    g_x = group_sum<LPP>(g_x);
    g_logu = group_sum<LPP>(g_logu);
    double gam;
    if (gshape < 1.0) {
      gam = tph_gamma_mt(gg, gshape);            // boosted small-shape path: serial (d = 1 and nu < 1 only)
    } else {
      const double dd = gshape - 1.0 / 3.0, cc = tph_rcp(tph_sqrt(9.0 * dd));
      double v = 1.0 + cc * g_x;
      v = v * v * v;
      if (v > 0.0 && g_logu < 0.5 * g_x * g_x + dd - dd * v + dd * tph_log(v)) gam = dd * v;
      else gam = tph_gamma_mt(gg, gshape, 1);    // rare: continue with attempt 1, 2, ... as the one-lane kernels do
    }
    gam *= tph_div(2.0, nu + m_u);
    b_fac = sigma * tph_sqrt(tph_rcp(gam));
  }
  __syncthreads();
  if (STAGE == 2 && KERNEL == TPH_KERNEL_TPCN && !carry) {      // Sigma^-1 is done with for now: the slot takes L
    stage(mat0, Lg);
    __syncthreads();
  }

  // ---- attempts (mcmc.py:239-249).  Every particle's lane group walks its OWN sequence of (attempt, row chunk) steps; nothing
  // is block-synchronous here (a group lives inside one wave: LDS traffic between its lanes is ordered by the wave's own
  // instruction stream).  Rows are taken in ascending order, LPP at a time: L is lower-triangular, row r needs z_0..z_r only, so
  // the normals of a redraw attempt are generated as the rows reach them and an attempt is ABANDONED AT ITS FIRST out-of-bounds
  // coordinate -- a violation at row r* costs ~r*^2/2 FMAs and r*/2 Box-Muller pairs instead of d^2/2 and d/2.  In the first
  // iterations of a high-dimensional run (50-D: ~98 % of the attempts leave the unit cube) that is most of the work.  Draws,
  // attempt order and arithmetic are those of the sequential loop: the accepted proposal is bit-identical.
  const int nchunks = (d + LPP - 1) / LPP;
  // (att_by_mode: behind fanned-out rounds over several modes every mode's list has gone through its own number of attempts)
  int att = todo ? (att_by_mode ? att_by_mode[c] : att0) : 0, ch = 0, zgen = todo ? 0 : 2 * npairs;   // attempt 0: all normals are in zs already (generated above)
  bool active = !todo || live;
  while (__any(active)) {
    if (active) {
      const int need = (ch + 1) * LPP < d ? (ch + 1) * LPP : d;
      if (zgen < need) {                              // 2*LPP more normals of this attempt: enough for this chunk and the next
        const int q = zgen / 2 + l;
        if (q < npairs) {
          double z0, z1;
          gz.normal2((uint32_t)(att * npairs + q), z0, z1);
          zs[2 * q] = z0;
          if (2 * q + 1 < d) zs[2 * q + 1] = z1;
        }
        zgen += 2 * LPP;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r = ch * LPP + l;
    int okl = 1;
    double v = 0.0;
    if (active && r < d) {
      const double* Lr = L + (size_t)r * ms;
      double acc = 0.0;
      for (int j = 0; j <= r; ++j) acc = fma(Lr[j], zs[j], acc);
      if (KERNEL == TPH_KERNEL_TPCN) v = mu[r] + a_fac * df[r] + b_fac * acc;
      else v = df[r] + b_fac * acc;
      const uint8_t f = bc ? bc[r] : (uint8_t)TPH_BC_STRICT;
      if (f == TPH_BC_PERIODIC) v = bc_periodic(v);
      else if (f == TPH_BC_REFLECTIVE) v = bc_reflective(v);
      else okl = (v >= 0.0) && (v <= 1.0);
    }
    const int okg = group_and<LPP>(okl);              // all lanes of the wave take part in the shuffles
    if (active) {
      if (okg) {
        if (r < d) vs[r] = v;
        if (++ch == nchunks) active = false;          // every row in bounds: this is the proposal
      } else {
        ++att; ch = 0; zgen = 0;
        if (att >= PROP_MAX_ATTEMPTS) {               // redraw cap reached (the reference would loop on): propose the current point
          for (int j = l; j < d; j += LPP) vs[j] = (KERNEL == TPH_KERNEL_TPCN) ? df[j] + mu[j] : df[j];
          active = false;
        }
      }
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && tick.ctl && !todo) {
    // regime probe for the host (StepEngine): mean attempts per particle in this block -> state[8].  While most attempts are
    // redraws the kernel should run un-staged (TPH_OPT_ML_UNSTAGED: a quarter of the LDS, four times the waves).
    double mine = (l == 0 && live) ? (double)(att + 1) : 0.0;
    double* probe = sh;                                  // zs of particle 0 is free now
    __syncthreads();
    if (threadIdx.x == 0) { probe[0] = 0.0; probe[1] = 0.0; }
    __syncthreads();
    if (l == 0 && live) { atomicAdd(&probe[0], mine); atomicAdd(&probe[1], 1.0); }   // integer-valued: exact, order-free
    __syncthreads();
    if (threadIdx.x == 0) const_cast<double*>(tick.ctl)[8] = probe[0] / fmax(probe[1], 1.0);
    __syncthreads();
  }
  if (live)
    for (int j = l; j < d; j += LPP) up[(size_t)j * ld + i] = vs[j];
  double m_up = 0.0;
  if (KERNEL == TPH_KERNEL_TPCN) {
    __syncthreads();
    if (STAGE == 2) stage(mat0, Pg);
    for (int j = l; j < d; j += LPP) vs[j] -= mu[j];
    __syncthreads();
    double part = 0.0;
    for (int r = l; r < d; r += LPP) {
      const double* Pr = P + (size_t)r * ms;
      double acc = 0.0;
      for (int j = 0; j <= r; ++j) acc = fma(Pr[j], vs[j], acc);
      part = fma(acc, acc, part);
    }
    m_up = group_sum<LPP>(part);
  }
  if (live && l == 0) {
    if (maha_u) maha_u[i] = m_u;
    if (maha_up) maha_up[i] = m_up;
  }
}

template <int KERNEL, int LPP>
static int launch_propose_ml(tph_ctx* ctx, double* u, const int32_t* assign, int64_t n, int64_t ld, const double* means,
                             const double* chol, const double* winv, const double* dof, const double* sigmas,
                             const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0, double* up, double* mu_,
                             double* mup, uint8_t* pend, const int32_t* todo = nullptr, const int32_t* todo_rows = nullptr,
                             int att0 = 1, const int32_t* att_by_mode = nullptr) {
  constexpr int PPB = ML_THREADS / LPP;
  const int d = ctx->d;
  const size_t base = sizeof(double) * 3 * (size_t)PPB * (d | 1);
  const size_t one = sizeof(double) * (size_t)d * (d + 1);
  const size_t budget = 150 * 1024;
  int stage = 0;
  if (assign == nullptr && ctx->ml_unstaged == 0) stage = (base + 2 * one <= budget) ? 1 : ((base + one <= budget) ? 2 : 0);
  const size_t lds = base + (stage == 1 ? 2 * one : (stage == 2 ? one : 0));
  const dim3 grid((unsigned)((n + PPB - 1) / PPB));
#define TPH_ML_LAUNCH(ST)                                                                                              \
  do {                                                                                                                 \
    if (lds > 64 * 1024)                                                                                               \
      TPH_HIP(hipFuncSetAttribute((const void*)k_propose_ml<KERNEL, LPP, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds));                                                                          \
    hipLaunchKernelGGL((k_propose_ml<KERNEL, LPP, ST>), grid, dim3(ML_THREADS), lds, ctx->stream, u, assign, n, ld, d, means, \
                       chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend, todo, todo_rows, att0, att_by_mode); \
  } while (0)
  if (stage == 1) TPH_ML_LAUNCH(1);
  else if (stage == 2) TPH_ML_LAUNCH(2);
  else TPH_ML_LAUNCH(0);
#undef TPH_ML_LAUNCH
  return 0;
}

// ---- d > 16, one mode, a step that is (mostly) ONE attempt: the blocked kernel --------------------------------------------
// 64 particles per workgroup, lane = particle, WV waves.  The Box-Muller pairs are dealt to the waves and land in LDS as
// columns zs[j][lane]; L z and, for tpCN, |L^-1 (u' - mu)|^2 are the blocked triangular products of tri.h (matrix
// element wave-uniform through the scalar cache, 8 rows per x_j read), the row chunks dealt to the waves.  Every particle
// gets attempt 0 -- in lockstep, which is what makes the matrix operand uniform; the particles whose attempt 0 leaves the cube
// are LISTED, and the kernel runs again over the list with attempt 1, over that round's list with attempt 2, ... (R rounds,
// TPH_OPT_BLOCKED; the lists are compact, so a round costs what its share of the particles costs: 131 072 x 100-D tpCN at 5.4
// attempts per particle: multi-lane kernel 3.3 ms un-staged / 5.8 ms staged, one round + stragglers 2.7 ms, ten rounds 2.3 ms;
// a round that still has work costs at least one tile's latency, 30-45 us at 100-D, so rounds pay while their list fills the chip).
// Whoever is still listed after R rounds is finished by k_propose_ml in straggler mode (attempt R, R+1, ... with early exit).
// The host uses this path while the redraw probe says that a step is a few attempts per particle (all but the first
// iterations of a run); redraw-DOMINATED steps go to the row walker (propose_sm.hip), where attempts stop at their first
// out-of-bounds row.  Same draws, same formulas: the proposal equals the other kernels' to rounding (different summation
// order in the products).
template <int KERNEL, int WV>
__global__ void __launch_bounds__(64 * WV) k_propose_blk(double* __restrict__ u, int64_t n, int64_t ld, int d,
                                                         const double* __restrict__ means, const double* __restrict__ Lb,
                                                         const double* __restrict__ Wb, const double* __restrict__ dof,
                                                         const double* __restrict__ sigmas, const uint8_t* __restrict__ bc,
                                                         uint64_t seed, tph_stepctl tick, int64_t item0, double* __restrict__ up,
                                                         double* __restrict__ maha_u, double* __restrict__ maha_up,
                                                         uint8_t* __restrict__ pend, const int32_t* __restrict__ cnt_in,
                                                         const int32_t* __restrict__ rows_in, int att,
                                                         int32_t* __restrict__ cnt_out, int32_t* __restrict__ rows_out) {
  // cnt_in == NULL: round 0, attempt 0 of every particle (and the chores of the step: pending moves, form at u, Gamma scale).
  // cnt_in != NULL: round `att` >= 1 over the particles the round before LISTED (rows_in[0 .. *cnt_in)): attempt `att` of each,
  // again in lockstep; the step scale b comes from where round 0 parked it (maha_up; RWM: sigma).  A round lists its own
  // failures in rows_out / cnt_out.  Blocks beyond the list exit at once.
  extern __shared__ double sh[];
  double* zs = sh;                               // [d][64] normals, later u' - mu
  double* vs = sh + (size_t)d * 64;              // [d][64] u - mu (first step of a run), then the proposal
  double* sc = sh + (size_t)2 * d * 64;          // [WV][64] per-wave partials; [WV*64 ..] b_fac[64]
  double* bf = sc + (size_t)WV * 64;
  __shared__ int s_ok[WV][64];
  // the wave index as a SCALAR (readfirstlane): the row chunks a wave takes, hence the matrix addresses, are then provably
  // wave-uniform and the matrix comes through scalar loads (with threadIdx.x >> 6 the compiler emits vector loads)
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t i = (int64_t)blockIdx.x * 64 + lane;
  bool live = i < n;
  int64_t ii = live ? i : n - 1;
  const bool first = cnt_in == nullptr;
  if (!first) {
    const int64_t cnt = *cnt_in;
    // the redraw probe from ALL first attempts (round 0's own estimate comes from its first 64 particles)
    if (att == 1 && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl)
      const_cast<double*>(tick.ctl)[8] = cnt < n ? (double)n / (double)(n - cnt) : 256.0;
    if ((int64_t)blockIdx.x * 64 >= cnt) return;        // the whole block (uniform): nothing listed for it
    live = i < cnt;
    i = ii = (int64_t)rows_in[live ? i : 0];             // dead lanes shadow the list's first particle, never store
  }
  const int npairs = (d + 1) >> 1;
  const double sigma = sigmas[0];
  const bool carry = KERNEL == TPH_KERNEL_TPCN && tick.carry();
  // ---- current point: resolve a pending accepted move (deferred tph_accept), rows dealt to the waves
  const bool pd = first && pend && live && pend[i];
  if (first)
    for (int j = wid; j < d; j += WV) {
      double uj = u[(size_t)j * ld + ii];
      if (pd) { uj = up[(size_t)j * ld + i]; u[(size_t)j * ld + i] = uj; }
      if (KERNEL == TPH_KERNEL_TPCN && !carry) vs[(size_t)j * 64 + lane] = uj - means[j];
    }
  // ---- the normals of this round's attempt, pairs dealt to the waves
  {
    tph_rng gz(seed, tick, TPH_TAG_NORMAL, (uint64_t)(item0 + ii));
    const uint32_t d0 = (uint32_t)att * (uint32_t)npairs;
    // two pairs per trip, as independent straight-line chains (the second one clamped, stored only when it exists): the
    // Box-Muller chain is latency-bound at this kernel's occupancy, like the d <= 16 kernel's
    for (int p = wid; p < npairs; p += 2 * WV) {
      const int p2 = p + WV < npairs ? p + WV : p;
      double z0, z1, y0, y1;
      gz.normal2(d0 + (uint32_t)p, z0, z1);
      gz.normal2(d0 + (uint32_t)p2, y0, y1);
      zs[(size_t)(2 * p) * 64 + lane] = z0;
      if (2 * p + 1 < d) zs[(size_t)(2 * p + 1) * 64 + lane] = z1;
      if (p2 != p) {
        zs[(size_t)(2 * p2) * 64 + lane] = y0;
        if (2 * p2 + 1 < d) zs[(size_t)(2 * p2 + 1) * 64 + lane] = y1;
      }
    }
  }
  __syncthreads();
  if (pd && wid == 0) pend[i] = 0;               // every wave has read the flag
  // ---- tpCN: Mahalanobis form at u (first step of a run; afterwards carried) and the Gamma scale
  double a_fac = 1.0;
  if (KERNEL == TPH_KERNEL_TPCN && !first) {
    if (wid == 0) bf[lane] = maha_up[ii];          // parked by round 0 (a success overwrites it with the form at u')
    a_fac = tph_sqrt(1.0 - sigma * sigma);
    __syncthreads();
  } else if (KERNEL == TPH_KERNEL_TPCN) {
    if (!carry) {
      double part = 0.0;
      tri_apply(Wb, d, vs, lane, wid, WV, [&](int, double y) { part = fma(y, y, part); });
      sc[(size_t)wid * 64 + lane] = part;
      __syncthreads();
    }
    if (wid == 0) {
      double m_u;
      if (carry) {
        m_u = maha_u[ii];
      } else {
        m_u = 0.0;
        for (int w = 0; w < WV; ++w) m_u += sc[(size_t)w * 64 + lane];
        if (live && maha_u) maha_u[i] = m_u;
      }
      const double nu = dof[0];
      tph_rng gg(seed, tick, TPH_TAG_GAMMA, (uint64_t)(item0 + ii));
      const double gam = tph_gamma_mt(gg, 0.5 * ((double)d + nu)) * tph_div(2.0, nu + m_u);
      const double b = sigma * tph_sqrt(tph_rcp(gam));
      bf[lane] = b;
      if (live) maha_up[i] = b;                    // parked for the later rounds of this particle
    }
    a_fac = tph_sqrt(1.0 - sigma * sigma);
    __syncthreads();
  } else if (wid == 0) {
    bf[lane] = sigma;
    if (first && live && maha_u) maha_u[i] = 0.0;
  }
  if (KERNEL != TPH_KERNEL_TPCN) __syncthreads();
  const double b_fac = bf[lane];
  // ---- rows of the attempt: v = mu + a (u - mu) + b (L z)_r, bounds
  int ok = 1;
  tri_apply(Lb, d, zs, lane, wid, WV, [&](int r, double acc) {
    const double ur = u[(size_t)r * ld + ii];
    double v;
    if (KERNEL == TPH_KERNEL_TPCN) v = means[r] + a_fac * (ur - means[r]) + b_fac * acc;
    else v = ur + b_fac * acc;
    const uint8_t f = bc ? bc[r] : (uint8_t)TPH_BC_STRICT;
    if (f == TPH_BC_PERIODIC) v = bc_periodic(v);
    else if (f == TPH_BC_REFLECTIVE) v = bc_reflective(v);
    else ok &= (v >= 0.0) && (v <= 1.0);
    vs[(size_t)r * 64 + lane] = v;
  });
  s_ok[wid][lane] = ok;
  __syncthreads();
  int all_ok = 1;
#pragma unroll
  for (int w = 0; w < WV; ++w) all_ok &= s_ok[w][lane];
  // ---- outputs of the particles whose attempt is in bounds; the others are left to the next round / the straggler pass
  for (int j = wid; j < d; j += WV) {
    const double v = vs[(size_t)j * 64 + lane];
    if (live && all_ok) up[(size_t)j * ld + i] = v;
    if (KERNEL == TPH_KERNEL_TPCN) zs[(size_t)j * 64 + lane] = v - means[j];      // the normals are done with
  }
  if (wid == 0 && live && !all_ok) rows_out[atomicAdd(cnt_out, 1)] = (int32_t)i;      // the straggler list (order is irrelevant)
  if (first && blockIdx.x == 0 && tick.ctl) {             // regime probe: mean attempts implied by this block's failures, 1 / (1 - f)
    __syncthreads();
    if (wid == 0) {
      const unsigned long long fm = __ballot(live && !all_ok), lm = __ballot(live);
      if (lane == 0) {
        const double f = (double)__popcll(fm) / fmax(1.0, (double)__popcll(lm));
        const_cast<double*>(tick.ctl)[8] = f < 0.99 ? 1.0 / (1.0 - f) : 100.0;
      }
    }
  }
  if (KERNEL == TPH_KERNEL_TPCN) {
    __syncthreads();
    double part = 0.0;
    tri_apply(Wb, d, zs, lane, wid, WV, [&](int, double y) { part = fma(y, y, part); });
    sc[(size_t)wid * 64 + lane] = part;
    __syncthreads();
    if (wid == 0 && live && all_ok) {
      double m_up = 0.0;
      for (int w = 0; w < WV; ++w) m_up += sc[(size_t)w * 64 + lane];
      maha_up[i] = m_up;
    }
  } else if (wid == 0 && live && all_ok && maha_up) {
    maha_up[i] = 0.0;
  }
}

template <int KERNEL>
static int launch_propose_blk(tph_ctx* ctx, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                              const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed,
                              tph_stepctl tick, int64_t item0, double* up, double* mu_, double* mup, uint8_t* pend) {
  const int d = ctx->d;
  // blocked copies of L and L^-1 (tri.h) and the straggler flags, owned by the ctx.  The copies are rebuilt on every call
  // unless the caller versions its mode statistics (TPH_OPT_MODES_EPOCH > 0 and unchanged since the last call)
  const size_t tb = tri_blocked_doubles(d);
  TPH_REQUIRE(n < (1ll << 31), "tph_propose (blocked): %lld particles on one device", (long long)n);
  // rounds of the blocked kernel before the straggler pass (TPH_OPT_BLOCKED = R: attempts 0 .. R-1 in lockstep)
  constexpr int BLK_MAX_ROUNDS = 24;
  const int rounds = ctx->blocked < 1 ? 1 : (ctx->blocked > BLK_MAX_ROUNDS ? BLK_MAX_ROUNDS : ctx->blocked);
  const size_t need = sizeof(double) * 2 * tb + sizeof(int32_t) * (2 * (size_t)n + 64);
  if (ctx->blk_bytes < need) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->blk_buf) ctx->retired.push_back(ctx->blk_buf);   // a captured step of a smaller engine may still point here
    ctx->blk_buf = nullptr; ctx->blk_bytes = 0; ctx->blk_epoch = -1;
    TPH_HIP(hipMalloc((void**)&ctx->blk_buf, need));
    ctx->blk_bytes = need;
  }
  double* Lb = (double*)ctx->blk_buf;
  double* Wb = Lb + tb;
  int32_t* cnts = (int32_t*)(Wb + tb);               // [k] = particles whose attempts 0..k all left the cube (k < rounds)
  int32_t* atts = cnts + 32;                         // [k] = the first attempt the particles listed by round k have not tried yet
  int32_t* rows[2] = {cnts + 64, cnts + 64 + n};     // their rows: round k writes rows[k & 1], round k + 1 reads it
  // (with the screened kernel as the straggler pass its work-queue words are zeroed by the same launch: one launch less per step)
  const bool screen = tph_mf_screen(ctx);
  unsigned int* mfq = screen ? tph_mf_queue_words(ctx) : nullptr;
  if (mfq) hipLaunchKernelGGL(k_zero_words2, dim3(1), dim3(64), 0, ctx->stream, (unsigned int*)cnts, 64, mfq, TPH_MF_QWORDS);
  else hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, ctx->stream, (unsigned int*)cnts, 64);
  // A launch that is being CAPTURED into a hipGraph always records the rebuild: a replayed step never re-enters this host
  // code, so an epoch test made here would freeze the copies of the capture-time statistics while the caller refreshes the
  // fixed-address chol / winv between runs (the straggler pass and tpCN's carried form read those) -- two covariances inside
  // one Metropolis ratio.  The copies made under capture are not trusted by later eager calls either.
  const bool mfma = ctx->blk_mfma && d <= 112;      // TPH_OPT_BLK_MFMA: the rounds on the FP64 matrix cores (propose_blkm.hip)
  int att_next = rounds;                            // the straggler pass starts here
  // the list rounds give a straggler several attempts side by side (propose_blkm.hip: fan-out) when the screened kernel finishes
  // the list: it reads the attempt to go on from on the device (the multi-lane straggler pass takes it by value)
  const bool fan = mfma && screen && ctx->blk_fan;
  if (mfma) {
    const int tries = tph_blkm_tries(ctx);      // TPH_OPT_BLK_TRIES: attempts per round, in place
    for (int k = 0; k < rounds; ++k)
      if (tph_blkm_round(ctx, KERNEL, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick.tick, tick.ctl, item0, up, mu_, mup, pend,
                         k ? cnts + (k - 1) : (const int32_t*)nullptr, (const int32_t*)rows[(k + 1) & 1], k * tries, cnts + k, rows[k & 1],
                         (fan && k) ? atts + (k - 1) : (const int32_t*)nullptr, fan ? atts + k : (int32_t*)nullptr))
        return -1;
    att_next = rounds * tries;
  } else {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  TPH_HIP(hipStreamIsCapturing(ctx->stream, &cap));
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (capturing || ctx->modes_epoch <= 0 || ctx->blk_epoch != ctx->modes_epoch || ctx->blk_src != (const void*)chol) {
    hipLaunchKernelGGL(k_tri_block, dim3(1), dim3(256), 0, ctx->stream, chol, d, Lb);
    if (KERNEL == TPH_KERNEL_TPCN) hipLaunchKernelGGL(k_tri_block, dim3(1), dim3(256), 0, ctx->stream, winv, d, Wb);
    ctx->blk_epoch = capturing ? -1 : ctx->modes_epoch; ctx->blk_src = (const void*)chol;
  }
  // waves per 64-particle tile: one 8-row chunk per wave up to 128 rows (the chunks of a triangular matrix are unequal: with
  // fewer waves the longest chain of chunks bounds the tile; measured 65 536 x 50-D: 4 -> 8 waves 76 -> 70 us (tpCN), 53 -> 46 us
  // (RWM); 262 144 x 100-D: 8 -> 16 waves 720 -> 644 us, 467 -> 393 us; at d = 32 four waves have a chunk each and win)
  const int wv = d <= 32 ? 4 : d <= 64 ? 8 : 16;
  const size_t lds = sizeof(double) * ((size_t)2 * d * 64 + (size_t)(wv + 1) * 64);
  const dim3 grid((unsigned)((n + 63) / 64));
#define TPH_BLK_LAUNCH(WV)                                                                                              \
  do {                                                                                                                  \
    if (lds > 64 * 1024)                                                                                                \
      TPH_HIP(hipFuncSetAttribute((const void*)k_propose_blk<KERNEL, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    for (int k = 0; k < rounds; ++k)                                                                                    \
      hipLaunchKernelGGL((k_propose_blk<KERNEL, WV>), grid, dim3(64 * WV), lds, ctx->stream, u, n, ld, d, means, Lb, Wb, dof, \
                         sigmas, bc, seed, tick, item0, up, mu_, mup, pend, k ? cnts + (k - 1) : (const int32_t*)nullptr,  \
                         (const int32_t*)rows[(k + 1) & 1], k, cnts + k, rows[k & 1]);                                    \
  } while (0)
  if (wv == 4) TPH_BLK_LAUNCH(4); else if (wv == 8) TPH_BLK_LAUNCH(8); else TPH_BLK_LAUNCH(16);
#undef TPH_BLK_LAUNCH
  TPH_LAUNCH_CHECK();
  }
  // straggler pass: the particles still listed continue with attempt `rounds`, ...  Screened windows over the list
  // (propose_mf.hip, TPH_OPT_SCREEN: 8 attempts of a straggler in flight, the first in bounds in attempt order wins; a
  // launch over a short list costs its workgroups' table loads, where the multi-lane kernel's straggler pass cost the chain
  // of its hardest particle's attempts: config 5's shard 357 us per launch, config 2 47-56 us) -- or, with the screen off or
  // beyond its range, the multi-lane kernel
  if (screen)
    return tph_propose_mf_list(ctx, KERNEL, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick.tick, tick.ctl, item0, up, mup,
                               cnts + (rounds - 1), rows[(rounds - 1) & 1], att_next, fan ? atts + (rounds - 1) : (const int32_t*)nullptr,
                               mfq ? 1 : 0);
  // as many lanes per straggler as it has Box-Muller pairs: a straggler's chain of attempts is latency-bound (few blocks
  // have any work), so the pairs of an attempt are generated in ONE round and the rows spread over more lanes
  int lpp = 4;
  while (lpp < 64 && lpp < (d + 1) / 2) lpp *= 2;
  const int keep = ctx->ml_unstaged;
  ctx->ml_unstaged = 1;        // (staged, measured: 65 536 x 50-D +-0, 131 072 x 100-D +8 ... +28 %)
  int rc = 0;
  switch (lpp) {
#define TPH_ML_S(LL) case LL: rc = launch_propose_ml<KERNEL, LL>(ctx, u, nullptr, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, nullptr, cnts + (rounds - 1), rows[(rounds - 1) & 1], att_next); break;
    TPH_ML_S(4) TPH_ML_S(8) TPH_ML_S(16) TPH_ML_S(32) TPH_ML_S(64)
#undef TPH_ML_S
  }
  ctx->ml_unstaged = keep;
  if (rc) return rc;
  TPH_LAUNCH_CHECK();
  return 0;
}

// Small-dimension variant (d <= 16 known at compile time): everything in registers, loops fully unrolled, so
// one lane's independent Philox / Box-Muller / matvec chains overlap.  The redraw-until-in-bounds loop
// (mcmc.py:239-249) is run as a block-level WORK LIST: after each attempt the particles still out of bounds are
// compacted into an LDS list and dealt to the first lanes again, so the cost follows the total number of redraws
// instead of 64 x the worst lane of every wave (early iterations redraw 30-60 % of the proposals).
// A block owns 256 * PPT particles: with PPT = 4 the straggler lists of the later rounds are four times as long and fill
// whole waves (a round with 3 busy lanes costs the issue slots of a full wave; measured, 1 048 576 particles: PPT 1 -> 4
// = -20 % of the kernel's VALU instructions).
// Mahalanobis forms: (v - mu)^T Sigma^-1 (v - mu) = |W (v - mu)|^2 with W = L^-1 lower-triangular -- d(d+1)/2 FMAs and as
// many wave-uniform operands instead of d^2 (Sigma^-1 is formed as W^T W anyway, tph_chol_inv), same value to rounding.
// With ONE_MODE the matrices are wave-uniform and travel through scalar loads; the pointers are laundered inside the
// round loop (tph_opaque) so that the loads stay next to their use: hoisted out of the loop their 2 x 110 SGPRs do not fit
// the scalar file and came back as v_readlane/v_writelane spill traffic (15 % of the instructions of the first version).
// wave-local ordering of LDS traffic between the lanes of ONE wave (no s_barrier: the workgroup is a single wave)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One WAVE per workgroup; a wave owns the 64-particle tiles  blockIdx.x + k * gridDim.x  (k < tiles <= REG_MAX_TILES).
// The work list of the redraw rounds is wave-local (ballot + popcount compaction into LDS, deterministic order, no
// atomics, no s_barrier).  The first version kept one list per 256-thread block: in the straggler rounds three of its
// four waves sat at the barrier while one computed, and with three resident blocks per CU the SIMDs idled -- PMC on
// 1 048 576 particles, half of the first attempts out of bounds: SQ_WAIT_ANY 55 % of the wave-cycles, VALU busy 38 %
// of the launch.  Independent waves never wait for each other, several tiles per wave make the straggler lists long
// enough to fill whole passes, and the launch is sized to ONE resident batch (WPE waves per SIMD x the chip's SIMDs)
// so that no under-filled second batch trails behind.
constexpr int REG_MAX_TILES = 8;

template <int KERNEL, int D, bool ONE_MODE, int WPE, bool HAS_BC>
__global__ void __launch_bounds__(64, WPE) k_propose_reg(double* __restrict__ u, const int32_t* __restrict__ assign,
                                                         int64_t n, int64_t ld, const double* __restrict__ means,
                                                         const double* __restrict__ chol, const double* __restrict__ winv,
                                                         const double* __restrict__ dof, const double* __restrict__ sigmas,
                                                         const uint8_t* __restrict__ bc, uint64_t seed, tph_stepctl tick,
                                                         int64_t item0, double* __restrict__ up, double* __restrict__ maha_u,
                                                         double* __restrict__ maha_up, int tiles, uint8_t* __restrict__ pend) {
  constexpr int NPW = 64 * REG_MAX_TILES;
  __shared__ double s_bfac[NPW];
  __shared__ int s_list[2][NPW];
  const int lane = threadIdx.x;
  constexpr int NP = (D + 1) / 2;
  // local particle id pid = k * 64 + lane  <->  global row (blockIdx.x + k * gridDim.x) * 64 + lane
  auto row_of = [&](int pid) { return ((int64_t)blockIdx.x + (int64_t)(pid >> 6) * gridDim.x) * 64 + (pid & 63); };

  // One attempt `att` for particle i (mode c) from its current point uc: z <- the proposal; returns in-bounds?
  // (mcmc.py:239-249 tpCN / :301-312 RWM; att == PROP_MAX_ATTEMPTS: the redraw cap, the current point is proposed.)
  auto attempt = [&](int64_t i, int c, int att, double b_fac, double (&z)[D]) -> bool {
    const double* __restrict__ mu = means + (size_t)c * D;
    const double* __restrict__ L = chol + (size_t)c * D * D;
    bool ok = true;
    if (att < PROP_MAX_ATTEMPTS) {
      {
      // the normals of this attempt: a ROLLED loop over the Box-Muller pairs writing a private array (dynamic index ->
      // scratch memory, 8 B per value and lane, L1/L2-resident): one Philox / log / sincospi body with ~30 live
      // registers instead of ceil(D/2) interleaved copies
      tph_rng gz(seed, tick, TPH_TAG_NORMAL, (uint64_t)(item0 + i));
      if constexpr (WPE <= 2) {
        // the pairs as independent, interleaved chains: the kernel is bound by the LATENCY of one particle's dependent chain
        // as much as by issue slots (see launch_propose_reg for where this form is used)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          double z0, z1;
          gz.normal2((uint32_t)(att * NP + p), z0, z1);
          z[2 * p] = z0;
          if (2 * p + 1 < D) z[2 * p + 1] = z1;
        }
      } else {
        double zb[2 * NP];
#pragma unroll 1
        for (int p = 0; p < NP; ++p) {
          double z0, z1;
          gz.normal2((uint32_t)(att * NP + p), z0, z1);
          zb[2 * p] = z0;
          zb[2 * p + 1] = z1;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) z[j] = zb[j];
      }
      }
      if (ONE_MODE) L = tph_opaque(L);
      const double sigma = sigmas[c];
      const double a_fac = (KERNEL == TPH_KERNEL_TPCN) ? tph_sqrt(1.0 - sigma * sigma) : 1.0;
#pragma unroll
      for (int r = D - 1; r >= 0; --r) {  // descending: slot r is free once row r is done
        if (ONE_MODE && (r == (2 * D) / 3 || r == D / 3)) L = tph_opaque(L);
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j <= r; ++j) acc = fma(L[r * D + j], z[j], acc);
        const double ur = u[(size_t)r * ld + i];      // loaded where it is used: the current point is not held across the
                                                      // Box-Muller loop (20 registers for d = 10)
        double v;
        if (KERNEL == TPH_KERNEL_TPCN) v = mu[r] + a_fac * (ur - mu[r]) + b_fac * acc;
        else v = ur + b_fac * acc;
        if (HAS_BC) {       // periodic / reflective dimensions (mcmc.py:326-366): a separate instantiation, so that
                            // the usual all-strict case carries none of the fmod code in its unrolled rows
          const uint8_t f = bc[r];
          if (f == TPH_BC_PERIODIC) v = bc_periodic(v);
          else if (f == TPH_BC_REFLECTIVE) v = bc_reflective(v);
          else ok = ok && (v >= 0.0) && (v <= 1.0);
        } else {
          ok = ok && (v >= 0.0) && (v <= 1.0);
        }
        z[r] = v;
      }
    } else {  // redraw cap reached (the reference would loop on): propose the current point
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double uj = u[(size_t)j * ld + i];
        z[j] = (KERNEL == TPH_KERNEL_TPCN) ? (uj - mu[j]) + mu[j] : uj;
      }
    }
    return ok;
  };
  // the winning attempt's outputs: u' and its Mahalanobis form
  auto commit = [&](int64_t i, int c, double (&z)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) up[(size_t)j * ld + i] = z[j];
    double m_up = 0.0;
    if (KERNEL == TPH_KERNEL_TPCN) {
      const double* __restrict__ mu = means + (size_t)c * D;
      const double* __restrict__ W = winv + (size_t)c * D * D;
      if (ONE_MODE) W = tph_opaque(W);
#pragma unroll
      for (int j = 0; j < D; ++j) z[j] -= mu[j];
      m_up = maha_w<D, ONE_MODE>(W, z);
    }
    if (maha_up) maha_up[i] = m_up;
  };

  // ---- phase A, tile by tile: resolve a pending accepted move; Mahalanobis at u and the Gamma scale (one draw, reused
  // by the redraws).  (Fusing attempt 0 into this loop -- the current point stays in registers -- was measured SLOWER:
  // the longer live ranges spill, 79.5 -> 84 us at 1 048 576 particles.)
  int count = 0;
#pragma unroll 1
  for (int t = 0; t < tiles; ++t) {
    const int pid = t * 64 + lane;
    const int64_t i = row_of(pid);
    double b_fac = 0.0;
    if (i < n) {
      // deferred Metropolis update (tph_accept with a pending mask): the previous step's accepted proposal becomes the
      // current point HERE.  As a kernel of its own the masked in-place copy is a bandwidth-bound read-modify-write of
      // every line of u (50 us at 1 048 576 x 10-D); here it costs ~25 us beside this kernel's FP64 work.
      if (pend && pend[i]) {
#pragma unroll
        for (int j = 0; j < D; ++j) u[(size_t)j * ld + i] = up[(size_t)j * ld + i];
        pend[i] = 0;
      }
      const int c = ONE_MODE ? 0 : assign[i];
      const double sigma = sigmas[c];
      b_fac = sigma;
      if (KERNEL == TPH_KERNEL_TPCN) {
        double m_u;
        if (tick.carry()) {
          m_u = maha_u[i];
        } else {
          const double* __restrict__ mu = means + (size_t)c * D;
          const double* __restrict__ W = winv + (size_t)c * D * D;
          double df[D];
#pragma unroll
          for (int j = 0; j < D; ++j) df[j] = u[(size_t)j * ld + i] - mu[j];
          m_u = maha_w<D, ONE_MODE>(W, df);
          if (maha_u) maha_u[i] = m_u;
        }
        const double nu = dof[c];
        tph_rng gg(seed, tick, TPH_TAG_GAMMA, (uint64_t)(item0 + i));
        const double g0 = tph_gamma_mt(gg, 0.5 * ((double)D + nu));
        const double gam = g0 * tph_div(2.0, nu + m_u);
        b_fac = sigma * tph_sqrt(tph_rcp(gam));
      } else if (maha_u) {
        maha_u[i] = 0.0;
      }
    }
    s_bfac[pid] = b_fac;
    if (row_of(t * 64) < n) count = (t + 1) * 64;     // slots of round 0 (rows >= n inside the last tile are skipped below)
  }
  wave_sync();
  // ---- phase B: attempts over the shrinking work list.  Round 0 takes every particle of the wave (one pass of 64
  // lanes per tile); afterwards only the particles still out of bounds, packed into as few passes as possible.  Once
  // the list is shorter than half a wave, the spare lanes try the NEXT attempts of the same particles at the same time
  // -- G lanes per particle evaluate attempts a0 .. a0+G-1 (independent counter-based draws) and the first in-bounds one
  // in attempt order wins, which is exactly the proposal the sequential loop would have returned.
  int a0 = 0;                      // attempts [0, a0) have failed for every particle still on the list
#pragma unroll 1
  for (int round = 0; a0 <= PROP_MAX_ATTEMPTS && count > 0; ++round) {
    const int cur = round & 1, nxt = cur ^ 1;
    int G = 1;
    if (round > 0)
      while (G < 64 && 2 * G * count <= 64) G *= 2;
    const int n_work = count * G;
    int n_next = 0;
#pragma unroll 1
    for (int w0 = 0; w0 < n_work; w0 += 64) {
      const int wl = w0 + lane;
      const int slot = wl / G, att = a0 + (wl % G);
      bool busy = slot < count && att <= PROP_MAX_ATTEMPTS;
      bool ok = false;
      int pid = 0, c = 0;
      int64_t i = 0;
      double z[D];
      if (busy) {
        pid = round == 0 ? slot : s_list[cur][slot];
        i = row_of(pid);
        busy = i < n;
      }
      if (busy) {
        c = ONE_MODE ? 0 : assign[i];
        ok = attempt(i, c, att, s_bfac[pid], z);
      }
      // first in-bounds attempt of each particle: its G lanes are consecutive lanes of the wave (G <= 64 divides 64)
      const unsigned long long okmask = __ballot(ok);
      const int g0 = lane & ~(G - 1);
      const unsigned long long grp = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << g0;
      const unsigned long long mine = okmask & grp;
      const bool winner = ok && (mine & ((1ull << lane) - 1ull)) == 0ull;
      if (winner) commit(i, c, z);
      // particles whose G attempts all left the cube go to the next round's list, in lane order
      const bool again = busy && mine == 0ull && (lane & (G - 1)) == 0;
      const unsigned long long amask = __ballot(again);
      if (again) s_list[nxt][n_next + __popcll(amask & ((1ull << lane) - 1ull))] = pid;
      n_next += __popcll(amask);
    }
    wave_sync();
    count = n_next;
    a0 += G;
  }
}

template <int KERNEL, int D>
static void launch_propose_reg(tph_ctx* ctx, double* u, const int32_t* assign, int64_t n, int64_t ld,
                               const double* means, const double* chol, const double* winv, const double* dof,
                               const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                               double* up, double* mu_, double* mup, uint8_t* pend) {
  // one resident batch: waves = min(tiles, 4 per SIMD x SIMDs), tiles of a wave = ceil(tiles / waves) <= REG_MAX_TILES
  // (TPH_OPT_REDRAW_LANES = t > 0 forces t tiles per wave: experiments)
  constexpr int WPE = 4;
  const int64_t ntiles = (n + 63) / 64;
  int64_t waves = (int64_t)WPE * ctx->n_simd;
  if (waves > ntiles) waves = ntiles;
  if (ntiles > waves * REG_MAX_TILES) waves = (ntiles + REG_MAX_TILES - 1) / REG_MAX_TILES;
  int tiles = (int)((ntiles + waves - 1) / waves);
  if (ctx->redraw_lanes > 0) tiles = ctx->redraw_lanes < REG_MAX_TILES ? ctx->redraw_lanes : REG_MAX_TILES;
  waves = (ntiles + tiles - 1) / tiles;
#define TPH_REG_LAUNCH(ONE, BC)                                                                                         \
  hipLaunchKernelGGL((k_propose_reg<KERNEL, D, ONE, WPE, BC>), dim3((unsigned)waves), dim3(64), 0, ctx->stream, u, assign, n, ld, \
                     means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, tiles, pend)
  // The instantiation with the Box-Muller pairs as independent, interleaved chains (template argument 2): wherever it stays
  // within 128 VGPRs -- RWM at every n_dim, tpCN up to n_dim = 11 -- it keeps four waves per SIMD AND fills their issue slots
  // (1 048 576 x 10-D: 82.8 -> 71.3 us with every first attempt in bounds, 197 -> 157 us with half of them out); above that
  // (tpCN, n_dim 12..16: 129-168 VGPRs) only where occupancy does not matter, i.e. at most two waves per SIMD.
  constexpr bool ilp_fits = KERNEL == TPH_KERNEL_RWM || D <= 11;
  if (assign == nullptr && !bc && ctx->redraw_lanes == 0 && (ilp_fits || waves <= 2 * (int64_t)ctx->n_simd)) {
    hipLaunchKernelGGL((k_propose_reg<KERNEL, D, true, 2, false>), dim3((unsigned)waves), dim3(64), 0, ctx->stream, u, assign, n, ld,
                       means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, tiles, pend);
  } else if (assign == nullptr) { if (bc) TPH_REG_LAUNCH(true, true); else TPH_REG_LAUNCH(true, false); }
  else { if (bc) TPH_REG_LAUNCH(false, true); else TPH_REG_LAUNCH(false, false); }
#undef TPH_REG_LAUNCH
}

#define TPH_PROPOSE_CASE(DD)                                                                                       \
  case DD:                                                                                                         \
    if (kernel == TPH_KERNEL_TPCN)                                                                                 \
      launch_propose_reg<TPH_KERNEL_TPCN, DD>(ctx, u_dev, assign_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, \
                                              sigmas_dev, bc_dev, seed, tick, item0, uprime_dev, maha_u_dev,       \
                                              maha_up_dev, pending_dev);                                           \
    else                                                                                                           \
      launch_propose_reg<TPH_KERNEL_RWM, DD>(ctx, u_dev, assign_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev,  \
                                             sigmas_dev, bc_dev, seed, tick, item0, uprime_dev, maha_u_dev,        \
                                             maha_up_dev, pending_dev);                                            \
    break;

extern "C" int tph_propose(tph_ctx* ctx, int kernel, double* u_dev, const int32_t* assign_dev, int64_t n, int64_t ld,
                           int K, const double* means_dev, const double* chol_dev, const double* cholinv_dev,
                           const double* dof_dev, const double* sigmas_dev, const uint8_t* bc_dev, uint64_t seed,
                           uint32_t tick0, int64_t item0, double* uprime_dev, double* maha_u_dev, double* maha_up_dev,
                           const double* ctl_dev, uint8_t* pending_dev) {
  TPH_REQUIRE(ctx && u_dev && uprime_dev && chol_dev && sigmas_dev, "tph_propose: NULL argument");
  const tph_stepctl tick{tick0, ctl_dev};
  TPH_REQUIRE(n > 0 && ld >= n && K >= 1, "tph_propose: bad sizes");
  TPH_REQUIRE(kernel == TPH_KERNEL_TPCN || kernel == TPH_KERNEL_RWM, "tph_propose: unknown kernel %d", kernel);
  if (kernel == TPH_KERNEL_TPCN)
    TPH_REQUIRE(means_dev && dof_dev && maha_u_dev && maha_up_dev, "tph_propose: tpCN needs means/dof/maha");
  TPH_REQUIRE(K == 1 || assign_dev, "tph_propose: K>1 needs assignments");
  const double* inv_dev = cholinv_dev;     // W = L^-1 per mode
  if (kernel == TPH_KERNEL_TPCN && !inv_dev) {
    // the caller has only the Cholesky factors (the reference's ModeStatistics.chol_covariances): invert them here
    const size_t bytes = sizeof(double) * (size_t)K * ctx->d * ctx->d;
    if (ctx->winv_bytes < bytes) {
      TPH_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->winv) ctx->retired.push_back(ctx->winv);
      ctx->winv = nullptr; ctx->winv_bytes = 0;
      TPH_HIP(hipMalloc((void**)&ctx->winv, bytes));
      ctx->winv_bytes = bytes;
    }
    if (tph_tri_inv(ctx, chol_dev, K, ctx->winv)) return -1;
    inv_dev = ctx->winv;
  }
  const int variant = ctx->propose_variant;   // 0 auto | 1 one-lane LDS | 2 one-lane registers (d<=16) | 3 multi-lane | 4 blocked
  const bool use_reg = (variant == 2 || variant == 0) && ctx->d <= 16;
  // redraw-dominated steps: attempts screened on the matrix cores, survivors in FP64 (propose_mf.hip; TPH_OPT_SCREEN, default)
  // or every attempt walked row by row in FP64 (propose_sm.hip)
  if (!use_reg && ctx->d > 16 && ctx->d <= 112 && !assign_dev && K == 1 &&
      ((variant == 6 && tph_mf_selftest(ctx)) || (variant == 0 && ctx->staged && !ctx->blocked && tph_mf_screen(ctx))))
    return tph_propose_mf(ctx, kernel, u_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed, tick0, ctl_dev,
                          item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev);
  if (!use_reg && ctx->d > 16 && ctx->d <= 100 && !assign_dev && K == 1 &&
      (variant == 5 || (variant == 0 && ctx->staged && !ctx->blocked)))
    return tph_propose_sm(ctx, kernel, u_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed, tick0, ctl_dev,
                          item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev);
  // ... with several modes: the same batches mode by mode over the particles of each (propose_mf.hip: tph_propose_mf_modes)
  if (!use_reg && ctx->d > 16 && ctx->d <= 112 && assign_dev && K > 1 && K <= 64 &&
      ((variant == 6 && tph_mf_selftest(ctx)) || (variant == 0 && ctx->staged && !ctx->blocked && tph_mf_screen(ctx))))
    return tph_propose_mf_modes(ctx, kernel, u_dev, assign_dev, n, ld, K, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed, tick0,
                                ctl_dev, item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev);
  if (!use_reg && ctx->d > 16 && ctx->d <= (ctx->blk_mfma ? 112 : 100) && !assign_dev && K == 1 && (variant == 4 || (variant == 0 && ctx->blocked))) {
    if (kernel == TPH_KERNEL_TPCN)
      return launch_propose_blk<TPH_KERNEL_TPCN>(ctx, u_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed,
                                                 tick, item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev);
    return launch_propose_blk<TPH_KERNEL_RWM>(ctx, u_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed,
                                              tick, item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev);
  }
  // several modes, a step of a few attempts per particle: the rounds of the blocked path over mode-pure tiles of 16 particles
  // (propose_blkm.hip), then the multi-lane kernel over the particles still out of bounds (it reads its particle's mode)
  if (!use_reg && ctx->d > 16 && ctx->d <= 112 && assign_dev && K > 1 && K <= 64 && ctx->blk_mfma &&
      (variant == 4 || (variant == 0 && ctx->blocked))) {
    const int rounds = ctx->blocked < 1 ? 1 : ctx->blocked;
    const int32_t *todo_cnt = nullptr, *todo_rows = nullptr, *todo_att = nullptr;
    const int32_t* per_mode[3] = {nullptr, nullptr, nullptr};
    const bool screen = tph_mf_screen(ctx);
    if (tph_blkm_multi(ctx, kernel, u_dev, assign_dev, n, ld, K, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed, tick0,
                       ctl_dev, item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev, rounds, &todo_cnt, &todo_rows, &todo_att,
                       screen ? per_mode : nullptr))
      return -1;
    const int att0 = (rounds > 24 ? 24 : rounds) * tph_blkm_tries(ctx);
    // the stragglers of the rounds: screened windows per mode (each mode's failure list with its own matrices), as the one-mode
    // path finishes its list; with the screen off, the multi-lane kernel over the concatenated list
    if (screen)
      return tph_propose_mf_mode_lists(ctx, kernel, u_dev, n, ld, K, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed, tick0, ctl_dev,
                                       item0, uprime_dev, maha_up_dev, per_mode[0], per_mode[1], per_mode[2], att0, todo_att);
    int lpp = 4;
    while (lpp < 64 && lpp < (ctx->d + 1) / 2) lpp *= 2;
    const int keep = ctx->ml_unstaged;
    ctx->ml_unstaged = 1;
    int rc = 0;
    switch (lpp) {
#define TPH_ML_M(LL)                                                                                                     \
  case LL:                                                                                                               \
    rc = kernel == TPH_KERNEL_TPCN                                                                                       \
             ? launch_propose_ml<TPH_KERNEL_TPCN, LL>(ctx, u_dev, assign_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev,   \
                                                      bc_dev, seed, tick, item0, uprime_dev, maha_u_dev, maha_up_dev, nullptr, todo_cnt,  \
                                                      todo_rows, att0, todo_att)                                                            \
             : launch_propose_ml<TPH_KERNEL_RWM, LL>(ctx, u_dev, assign_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev,    \
                                                     bc_dev, seed, tick, item0, uprime_dev, maha_u_dev, maha_up_dev, nullptr, todo_cnt,   \
                                                     todo_rows, att0, todo_att);                                                            \
    break;
      TPH_ML_M(4) TPH_ML_M(8) TPH_ML_M(16) TPH_ML_M(32) TPH_ML_M(64)
#undef TPH_ML_M
    }
    ctx->ml_unstaged = keep;
    if (rc) return rc;
    TPH_LAUNCH_CHECK();
    return 0;
  }
  if (!use_reg && (variant == 3 || variant == 0) && ctx->d <= 8 * 64) {
    // lanes per particle: the fewest with <= 8 rows per lane, so that as many particles as possible share one
    // block's staged matrices (the draws are dealt round-robin over the lanes in as many passes as needed)
    int lpp = 4;
    while (lpp < 64 && (ctx->d + lpp - 1) / lpp > 8) lpp *= 2;
#define TPH_ML(LL)                                                                                                      \
  case LL:                                                                                                               \
    if (kernel == TPH_KERNEL_TPCN)                                                                                       \
      rc_ml = launch_propose_ml<TPH_KERNEL_TPCN, LL>(ctx, u_dev, assign_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, \
                                             sigmas_dev, bc_dev, seed, tick, item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev); \
    else                                                                                                                 \
      rc_ml = launch_propose_ml<TPH_KERNEL_RWM, LL>(ctx, u_dev, assign_dev, n, ld, means_dev, chol_dev, inv_dev, dof_dev, \
                                            sigmas_dev, bc_dev, seed, tick, item0, uprime_dev, maha_u_dev, maha_up_dev, pending_dev);  \
    break;
    int rc_ml = 0;
    switch (lpp) { TPH_ML(4) TPH_ML(8) TPH_ML(16) TPH_ML(32) TPH_ML(64) }
    if (rc_ml) return rc_ml;
#undef TPH_ML
    TPH_LAUNCH_CHECK();
    return 0;
  }
  if (use_reg) {
    switch (ctx->d) {
      TPH_PROPOSE_CASE(1) TPH_PROPOSE_CASE(2) TPH_PROPOSE_CASE(3) TPH_PROPOSE_CASE(4) TPH_PROPOSE_CASE(5)
      TPH_PROPOSE_CASE(6) TPH_PROPOSE_CASE(7) TPH_PROPOSE_CASE(8) TPH_PROPOSE_CASE(9) TPH_PROPOSE_CASE(10)
      TPH_PROPOSE_CASE(11) TPH_PROPOSE_CASE(12) TPH_PROPOSE_CASE(13) TPH_PROPOSE_CASE(14) TPH_PROPOSE_CASE(15)
      TPH_PROPOSE_CASE(16)
    }
    TPH_LAUNCH_CHECK();
    return 0;
  }
  size_t lds = sizeof(double) * 2 * (size_t)ctx->d * PROP_THREADS;
  TPH_REQUIRE(lds <= 160 * 1024, "tph_propose: n_dim=%d needs %zu B of LDS (>160 KiB)", ctx->d, lds);
  unsigned grid = (unsigned)((n + PROP_THREADS - 1) / PROP_THREADS);
  if (kernel == TPH_KERNEL_TPCN) {
    if (lds > 64 * 1024)
      TPH_HIP(hipFuncSetAttribute((const void*)k_propose<TPH_KERNEL_TPCN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_propose<TPH_KERNEL_TPCN>, dim3(grid), dim3(PROP_THREADS), lds, ctx->stream, u_dev, assign_dev, n, ld,
                       ctx->d, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed, tick, item0, uprime_dev,
                       maha_u_dev, maha_up_dev, pending_dev);
  } else {
    if (lds > 64 * 1024)
      TPH_HIP(hipFuncSetAttribute((const void*)k_propose<TPH_KERNEL_RWM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_propose<TPH_KERNEL_RWM>, dim3(grid), dim3(PROP_THREADS), lds, ctx->stream, u_dev, assign_dev, n, ld,
                       ctx->d, means_dev, chol_dev, inv_dev, dof_dev, sigmas_dev, bc_dev, seed, tick, item0, uprime_dev,
                       maha_u_dev, maha_up_dev, pending_dev);
  }
  TPH_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------ accept
constexpr int ACC_THREADS = 256;

// Metropolis step (mcmc.py:163-177): alpha = min(1, exp(beta (l'-l) + factor)), NaN -> 0,
// factor (tpCN, mcmc.py:251-279) = -A + B with A,B = -0.5 (d+nu) log(1 + m/nu) at u', u.
// Accepted rows overwrite u, x, logl in place.  Block partials: [block][1+K] = (#accepted, sum alpha_c).
template <int KERNEL>
__global__ void __launch_bounds__(ACC_THREADS) k_accept(double beta, double* __restrict__ u, double* __restrict__ x,
                                                        double* __restrict__ logl, const double* __restrict__ up,
                                                        const double* __restrict__ xp, const double* __restrict__ lp,
                                                        double* __restrict__ maha_u, const double* __restrict__ maha_up,
                                                        const int32_t* __restrict__ assign, int64_t n, int64_t ld, int d, int K,
                                                        const double* __restrict__ dof, uint64_t seed, tph_stepctl tick,
                                                        int64_t item0, double* __restrict__ partials,
                                                        uint8_t* __restrict__ pend) {
  if (tick.done()) return;              // a step launched past the stopping rule (replayed graph) changes nothing
  if (tick.ctl) beta = tick.ctl[6];
  int64_t i = (int64_t)blockIdx.x * ACC_THREADS + threadIdx.x;
  double alpha = 0.0, acc = 0.0;
  int c = 0;
  if (i < n) {
    c = assign ? assign[i] : 0;
    double l0 = logl[i], l1 = lp[i];
    double factor = 0.0;
    if (KERNEL == TPH_KERNEL_TPCN) {
      const double nu = dof[c];
      factor = tph_tpcn_factor((double)d + nu, nu, maha_u[i], maha_up[i]);
    }
    double a = exp(beta * (l1 - l0) + factor);
    a = isnan(a) ? 0.0 : fmin(1.0, a);
    alpha = a;
    tph_rng g(seed, tick, TPH_TAG_ACCEPT, (uint64_t)(item0 + i));
    double U, U1;
    g.uniform2(0, U, U1);
    if (pend) {
      // deferred mode: the decision is recorded, u stays as it is -- the next tph_propose (given the same mask) moves the
      // accepted proposals into place under its own compute.  What remains here are whole-line streams: 8 B scalars
      // read and written for EVERY row (no partial lines), 1 B of mask.
      const bool take = U < a;
      acc = take ? 1.0 : 0.0;
      pend[i] = take ? 1 : 0;
      logl[i] = take ? l1 : l0;
      if (KERNEL == TPH_KERNEL_TPCN) maha_u[i] = take ? maha_up[i] : maha_u[i];
    } else if (U < a) {
      acc = 1.0;
      for (int j = 0; j < d; ++j) u[(size_t)j * ld + i] = up[(size_t)j * ld + i];
      if (x)
        for (int j = 0; j < d; ++j) x[(size_t)j * ld + i] = xp[(size_t)j * ld + i];
      logl[i] = l1;
      if (KERNEL == TPH_KERNEL_TPCN) maha_u[i] = maha_up[i];   // the form at the new position: next step's proposal reads it
    }
  }
  __shared__ double sh[ACC_THREADS / 64];
  double* out = partials + (size_t)blockIdx.x * (1 + K);
  double t = tph_block_sum(acc, sh);
  if (threadIdx.x == 0) out[0] = t;
  for (int k = 0; k < K; ++k) {
    t = tph_block_sum((i < n && c == k) ? alpha : 0.0, sh);
    if (threadIdx.x == 0) out[1 + k] = t;
  }
}

// ---- the step's sums (#accepted, sum alpha_c) from the block partials of k_accept, in the CANONICAL order (common.h: tph_part) ----
// The particle slots of a step are cut into the virtual shards of the canonical partition (vl of them on this rank, each a whole
// number `mper` of 256-particle blocks); a shard's column sums are formed by ONE wave -- lane l adds partials l, l + 64, ... in
// order, then a fixed shuffle tree -- and the V shard sums of the whole run are added in shard order.  The same tree on one GPU
// and on G: the adapted sigma, and with it every later proposal, does not depend on the number of ranks.  (Folding the sums
// into k_accept behind a last-block ticket was measured SLOWER: the agent-scope release fence writes back the L2 lines the
// accepted rows have just dirtied, +13 us, against a 5 us kernel that follows with no gap.)
static tph_part active_partition(const tph_ctx* ctx, int64_t n) {
  tph_part p{1, n, 1, ctx->world, n, false};
  const int64_t ng = n * (int64_t)ctx->world;
  if (ng % 256 == 0) {
    const int V = tph_vshards_for(ng);
    if (V % ctx->world == 0) { p.vl = V / ctx->world; p.V = V; p.nv = ng / V; p.canonical = true; }
  }
  return p;
}
// vs[v * ncol + col] = sum over the blocks of local shard v of partials[block][col]; every wave of the workgroup takes pairs
__device__ __forceinline__ void vshard_colsums(const double* __restrict__ partials, int nblocks, int vl, int ncol, double* __restrict__ vs) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int mper = nblocks / vl;                 // (vl == 1: all blocks)
  for (int p = wid; p < vl * ncol; p += nw) {
    const int v = p / ncol, col = p - v * ncol;
    const double* src = partials + (size_t)v * mper * ncol + col;
    double s = 0.0;
    for (int b = lane; b < mper; b += 64) s += src[(size_t)b * ncol];
    s = tph_wave_sum(s);
    if (lane == 0) vs[p] = s;
  }
}
// out[col] = ((vs[0][col] + vs[1][col]) + vs[2][col]) + ...  over V shards
__device__ __forceinline__ void vshard_fold(const double* __restrict__ vs, int V, int ncol, double* __restrict__ out) {
  for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
    double s = vs[c];
    for (int v = 1; v < V; ++v) s += vs[(size_t)v * ncol + c];
    out[c] = s;
  }
}
__global__ void __launch_bounds__(1024) k_accept_sums(const double* __restrict__ partials, int nblocks, int vl, int ncol,
                                                      double* __restrict__ vs, double* __restrict__ out, tph_stepctl tick) {
  if (tick.done()) return;
  vshard_colsums(partials, nblocks, vl, ncol, vs);
  __syncthreads();
  if (out) vshard_fold(vs, vl, ncol, out);
}
__global__ void __launch_bounds__(256) k_fold_shards(const double* __restrict__ vs, int V, int ncol, double* __restrict__ out) {
  vshard_fold(vs, V, ncol, out);
}
// scratch of the shard sums: [vl][ncol] of this rank, then [V][ncol] of the run; a fixed address per ctx (captured steps keep it)
static double* adapt_scratch(tph_ctx* ctx, size_t doubles) {
  if (ctx->adapt_bytes < sizeof(double) * doubles) {
    size_t nb = ctx->adapt_bytes ? ctx->adapt_bytes : 4096;
    while (nb < sizeof(double) * doubles) nb *= 2;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
    if (ctx->adapt_buf) ctx->retired.push_back(ctx->adapt_buf);      // a captured step of an earlier engine may still point here
    ctx->adapt_buf = nullptr; ctx->adapt_bytes = 0;
    if (hipMalloc((void**)&ctx->adapt_buf, nb) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->adapt_bytes = nb;
  }
  return ctx->adapt_buf;
}

extern "C" int tph_accept(tph_ctx* ctx, int kernel, double beta, double* u_dev, double* x_dev, double* logl_dev,
                          const double* uprime_dev, const double* xprime_dev, const double* loglprime_dev,
                          double* maha_u_dev, const double* maha_up_dev, const int32_t* assign_dev, int64_t n,
                          int64_t ld, int K, const double* dof_dev, uint64_t seed, uint32_t tick0, int64_t item0,
                          double* sums_dev, const double* ctl_dev, double* partials_dev, uint8_t* pending_dev) {
  const tph_stepctl tick{tick0, ctl_dev};
  TPH_REQUIRE(!pending_dev || !x_dev, "tph_accept: deferred mode (pending_dev) does not maintain x: pass x_dev = xprime_dev = NULL");
  TPH_REQUIRE(ctx && u_dev && logl_dev && uprime_dev && loglprime_dev, "tph_accept: NULL argument");
  TPH_REQUIRE((x_dev == nullptr) == (xprime_dev == nullptr), "tph_accept: x_dev and xprime_dev go together (both NULL = x is not maintained)");
  TPH_REQUIRE(sums_dev || partials_dev, "tph_accept: without sums_dev the block partials must go to partials_dev");
  TPH_REQUIRE(n > 0 && ld >= n && K >= 1, "tph_accept: bad sizes");
  TPH_REQUIRE(K == 1 || assign_dev, "tph_accept: K>1 needs assignments");
  TPH_REQUIRE(K <= 4096, "tph_accept: K=%d too large", K);
  TPH_REQUIRE(kernel == TPH_KERNEL_TPCN || kernel == TPH_KERNEL_RWM, "tph_accept: unknown kernel %d", kernel);
  if (kernel == TPH_KERNEL_TPCN) TPH_REQUIRE(maha_u_dev && maha_up_dev && dof_dev, "tph_accept: tpCN needs maha/dof");
  unsigned grid = (unsigned)((n + ACC_THREADS - 1) / ACC_THREADS);
  double* partials = partials_dev;       // caller-owned (fixed address: required when the launch is captured in a graph)
  if (!partials) {
    size_t need = sizeof(double) * (size_t)grid * (1 + K);
    if (tph_scratch_reserve(ctx, need)) return -1;
    partials = (double*)ctx->scratch;
  }
  if (kernel == TPH_KERNEL_TPCN)
    hipLaunchKernelGGL(k_accept<TPH_KERNEL_TPCN>, dim3(grid), dim3(ACC_THREADS), 0, ctx->stream, beta, u_dev, x_dev, logl_dev,
                       uprime_dev, xprime_dev, loglprime_dev, maha_u_dev, maha_up_dev, assign_dev, n, ld, ctx->d, K, dof_dev,
                       seed, tick, item0, partials, pending_dev);
  else
    hipLaunchKernelGGL(k_accept<TPH_KERNEL_RWM>, dim3(grid), dim3(ACC_THREADS), 0, ctx->stream, beta, u_dev, x_dev, logl_dev,
                       uprime_dev, xprime_dev, loglprime_dev, maha_u_dev, maha_up_dev, assign_dev, n, ld, ctx->d, K, dof_dev,
                       seed, tick, item0, partials, pending_dev);
  if (sums_dev) {
    // this rank's sums in the canonical order (a sharded caller combines the ranks with tph_accept_sums_global instead)
    const tph_part part = active_partition(ctx, n);
    double* vs = adapt_scratch(ctx, (size_t)(part.vl + part.V) * (1 + K));
    TPH_REQUIRE(vs, "tph_accept: cannot allocate the shard sums");
    hipLaunchKernelGGL(k_accept_sums, dim3(1), dim3(((size_t)part.vl * (1 + K)) > 4 ? 1024 : 256), 0, ctx->stream, (const double*)partials, (int)grid,
                       part.vl, 1 + K, vs, sums_dev, tick);
  }
  TPH_LAUNCH_CHECK();
  return 0;
}

// The step's sums over ALL ranks from this rank's block partials (tph_accept with sums_dev = NULL): per-shard sums -> all-gather
// (rank order = shard order; the peer-to-peer exchange when attached and not `host_paced`, else the callback) -> fold in shard
// order.  For callers that run tph_adapt without the folded exchange (no peer mapping, or user callbacks on the host between the
// collectives).  Without a communicator: the same sums of the one rank.
extern "C" int tph_accept_sums_global(tph_ctx* ctx, const double* partials_dev, int64_t n, int K, double* sums_dev, int host_paced) {
  TPH_REQUIRE(ctx && partials_dev && sums_dev && n > 0 && K >= 1 && K <= 4096, "tph_accept_sums_global: bad argument");
  const tph_part part = active_partition(ctx, n);
  const int ncol = 1 + K, nblocks = (int)((n + ACC_THREADS - 1) / ACC_THREADS);
  const tph_stepctl tick{0u, nullptr};
  const int threads = ((size_t)part.vl * ncol) > 4 ? 1024 : 256;
  if (!ctx->comm_active()) {
    double* vs = adapt_scratch(ctx, (size_t)(part.vl + part.V) * ncol);
    TPH_REQUIRE(vs, "tph_accept_sums_global: cannot allocate the shard sums");
    hipLaunchKernelGGL(k_accept_sums, dim3(1), dim3(threads), 0, ctx->stream, partials_dev, nblocks, part.vl, ncol, vs, sums_dev, tick);
    TPH_LAUNCH_CHECK();
    return 0;
  }
  const size_t one = sizeof(double) * (size_t)part.vl * ncol, all = (one + 255) / 256 * 256;
  if (tph_comm_require(ctx, all + one * ctx->world, "tph_accept_sums_global")) return -2;
  hipLaunchKernelGGL(k_accept_sums, dim3(1), dim3(threads), 0, ctx->stream, partials_dev, nblocks, part.vl, ncol, (double*)ctx->comm_buf,
                     (double*)nullptr, tick);
  TPH_LAUNCH_CHECK();
  if (host_paced ? tph_comm_allgather_cb(ctx, 0, all, (int64_t)part.vl * ncol, TPH_DT_F64)
                 : tph_comm_allgather(ctx, 0, all, (int64_t)part.vl * ncol, TPH_DT_F64))
    return -2;
  hipLaunchKernelGGL(k_fold_shards, dim3(1), dim3(256), 0, ctx->stream, (const double*)(ctx->comm_buf + all), part.vl * ctx->world, ncol, sums_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------- sigma adaptation + stopping rule
// mcmc.py:180-186 (per-cluster mean alpha), :281-288 / :320-323 (sigma update), :104-140,192-194
// (adaptive step count incl. the `sigmas[:n_nonempty]` weighting quirk), :196-197 (returned stats).
__global__ void __launch_bounds__(1024) k_adapt(int kernel, const double* __restrict__ sums, const double* __restrict__ counts, int K, double n_global,
                        int d, int n_steps, int n_max, double* __restrict__ sigmas, double* __restrict__ state,
                        double* __restrict__ mailbox, int slots, const double* __restrict__ partials, int nblocks,
                        double* __restrict__ sums_out, int exchange, p2p_args peers, int vl, double* __restrict__ vs) {
  if (state[1] != 0.0) return;          // stopping rule already fired: later (speculative) steps are no-ops (on every rank)
  if (partials) {                        // the sums of tph_accept's block partials, folded in here (canonical order: see above)
    const int ncol = 1 + K;
    __shared__ double s_vs[1024];          // the rank's shard sums stay in LDS when they fit (one GPU: no trip through memory)
    if (vl * ncol <= 1024) vs = exchange ? vs : s_vs;
    vshard_colsums(partials, nblocks, vl, ncol, vs);
    __syncthreads();
    const double* all = vs;
    int V = vl;
    // sharded run: the ranks' shard sums meet here, through the peer-mapped inboxes (p2p.h) -- the exchange of the step costs
    // no launch and no host call; every rank folds the V shard sums in shard order, so all adapt identically, and as on one GPU
    if (exchange) {
      double* gathered = vs + (size_t)vl * ncol;
      if (!p2p_block_exchange(peers, (const double*)vs, gathered, vl * ncol, -1)) return;
      all = gathered;
      V = vl * peers.world;
    }
    vshard_fold(all, V, ncol, sums_out);
    __syncthreads();
    sums = sums_out;
  }
  if (threadIdx.x != 0) return;
  tph_adapt_scalar(kernel, sums, counts, K, n_global, d, n_steps, n_max, sigmas, state, mailbox, slots);
}

extern "C" int tph_adapt(tph_ctx* ctx, int kernel, double* sums_dev, const double* counts_dev, int K, double n_global,
                         int n_dim, int n_steps, int n_max, double* sigmas_dev, double* state_dev,
                         double* mailbox_host, int mailbox_slots, const double* partials_dev, int64_t n) {
  TPH_REQUIRE(ctx && sums_dev && counts_dev && sigmas_dev && state_dev && K >= 1, "tph_adapt: bad argument");
  TPH_REQUIRE(!partials_dev || n > 0, "tph_adapt: partials need the particle count");
  TPH_REQUIRE(!mailbox_host || mailbox_slots >= 1, "tph_adapt: mailbox needs at least one slot");
  const int nparts = (int)((n + ACC_THREADS - 1) / ACC_THREADS);
  tph_part part{1, n, 1, ctx->world, n, false};
  double* vs = nullptr;
  if (partials_dev) {
    part = active_partition(ctx, n);
    vs = adapt_scratch(ctx, (size_t)(part.vl + part.V) * (1 + K));
    TPH_REQUIRE(vs, "tph_adapt: cannot allocate the shard sums");
  }
  const int threads = (partials_dev && (size_t)part.vl * (1 + K) > 4) ? 1024 : 256;      // a wave per (shard, column) pair
  p2p_args peers{};
  int exchange = 0;
  if (partials_dev && ctx->comm_active()) {
    // not attached, or this rank's shard sums do not fit one 32 KB slot: a usage condition the caller can act on (code 1: combine
    // the ranks with tph_accept_sums_global and pass partials_dev = NULL); an exchange that FAILED earlier keeps its own message
    const int64_t cnt = (int64_t)part.vl * (1 + K);
    if (!tph_p2p_fits(ctx, cnt, TPH_DT_F64)) {
      tph_set_error("tph_adapt: folding the block partials of a SHARDED step needs the peer-to-peer exchange (tph_comm_p2p_attach) and "
                    "vl (1 + K) = %lld doubles within one %zu-byte slot; combine the ranks with tph_accept_sums_global and pass partials_dev = NULL",
                    (long long)cnt, (size_t)TPH_P2P_SLOT);
      return 1;
    }
    const p2p_args* a = tph_p2p_ready(ctx, cnt, TPH_DT_F64);
    if (!a) return -2;                    // the sticky error of a timed-out exchange (text set by tph_p2p_ready)
    peers = *a;
    exchange = 1;
    ctx->stat[0] += 1;                    // the step's exchange, folded into k_adapt
  }
  hipLaunchKernelGGL(k_adapt, dim3(1), dim3(threads), 0, ctx->stream, kernel, (const double*)sums_dev, counts_dev, K, n_global,
                     n_dim, n_steps, n_max, sigmas_dev, state_dev, mailbox_host, mailbox_slots, partials_dev, nparts, sums_dev,
                     exchange, peers, part.vl, vs);
  TPH_LAUNCH_CHECK();
  return 0;
}

// particles per cluster (mcmc.py:107-112); int atomics: exact and order-independent
__global__ void __launch_bounds__(256) k_cluster_counts(const int32_t* __restrict__ assign, int64_t n, int K,
                                                        unsigned long long* __restrict__ cnt) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int c = assign[i];
    if (c >= 0 && c < K) atomicAdd(&cnt[c], 1ull);
  }
}
__global__ void k_u64_to_double(const unsigned long long* __restrict__ in, int K, double* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < K) out[c] = (double)in[c];
}

__global__ void k_fill_count(double n, int K, double* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < K) out[c] = c == 0 ? n : 0.0;
}

extern "C" int tph_cluster_counts(tph_ctx* ctx, const int32_t* assign_dev, int64_t n, int K, double* counts_dev) {
  TPH_REQUIRE(ctx && counts_dev && n > 0 && K >= 1 && K <= 4096, "tph_cluster_counts: bad argument");
  unsigned long long* cnt = (unsigned long long*)(ctx->small_dev);  // 4096 slots of 8 B
  if (!assign_dev) {            // one mode: every particle is in cluster 0 (no copy, no host synchronisation)
    hipLaunchKernelGGL(k_fill_count, dim3((K + 255) / 256), dim3(256), 0, ctx->stream, (double)n, K, counts_dev);
    TPH_LAUNCH_CHECK();
    return 0;
  }
  TPH_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * (size_t)K, ctx->stream));
  hipLaunchKernelGGL(k_cluster_counts, dim3(tph_grid_for(n, 256, 4)), dim3(256), 0, ctx->stream, assign_dev, n, K, cnt);
  hipLaunchKernelGGL(k_u64_to_double, dim3((K + 255) / 256), dim3(256), 0, ctx->stream, cnt, K, counts_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_mutate(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
