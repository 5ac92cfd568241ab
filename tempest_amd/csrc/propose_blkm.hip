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

// the redraw probe of the blocked path: mean attempts per particle implied by the share fs of the particles whose first `tries`
// attempts all left the cube -- per-attempt failure rate f = fs^(1/tries), geometric mean 1 / (1 - f) (capped like the redraw loop)
__device__ __forceinline__ double bm_probe(double fs, int tries) {
  const double f = tries > 1 ? pow(fs, 1.0 / (double)tries) : fs;
  return f < 1.0 - 1.0 / 256.0 ? 1.0 / (1.0 - f) : 256.0;
}

__host__ __device__ static inline int bm_panels(int d) { return (d + 15) / 16; }
__host__ __device__ static inline int bm_blocks(int np) { return 2 * np * (np + 1); }       // sum over p of 4p + 4
__host__ __device__ static inline int bm_blk(int p, int s) { return 2 * p * (p + 1) + s; }

// Tb[blk(p, s)][lane] = T[16p + (lane & 15)][col(s, lane >> 4)], zero outside the lower triangle / beyond d.
// perm = 1: col(s, k) = 8 (s >> 1) + 2k + (s & 1) (the Box-Muller layout of L z); perm = 0: col = 4s + k (natural: L^-1 x)
static __global__ void __launch_bounds__(256) k_blkm_pack(const double* __restrict__ T_all, int d, int np, int perm, double* __restrict__ Tb_all) {
  const int total = bm_blocks(np) * 64;
  const double* __restrict__ T = T_all + (size_t)blockIdx.x * d * d;           // one workgroup per mode
  double* __restrict__ Tb = Tb_all + (size_t)blockIdx.x * total;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int b = e >> 6, lane = e & 63;
    int p = 0;
    while (bm_blk(p + 1, 0) <= b) ++p;
    const int s = b - bm_blk(p, 0), k = lane >> 4;
    const int r = 16 * p + (lane & 15), c = perm ? 8 * (s >> 1) + 2 * k + (s & 1) : 4 * s + k;
    Tb[e] = (r < d && c <= r) ? T[(size_t)r * d + c] : 0.0;
  }
}


// ---- several modes: tiles of 16 particles of ONE mode ------------------------------------------------------------------
// The matrix operand of a tile is its mode's factor, so a tile must not mix modes.  Once per set of mode statistics (the
// assignments are fixed during an iteration's MCMC steps) the particles are grouped by mode: order[] lists the rows mode by
// mode, and a tile table cuts every mode's stretch into tiles {mode, start in order[], particles, start of the mode}.  The
// failure lists of the rounds are kept PER MODE too (mode m owns the stretch [mstart_m, mstart_m + n_m) of the list arrays,
// its counter is cnt[m]), so that the tiles of a later round are again pure: tile (m, j) takes entries 16j .. 16j+15 of mode
// m's list.  Layout of the ctx-owned block (int32): hdr[4] = {tiles, K, -, -} | mstart[64] | mcount[64] | cursor[64] |
// tiles[][4] | order[n].
constexpr int BM_KMAX = 64;
// (... | order[n] | blockcnt[ceil(n / 256)][K]: particles of every mode per 256-particle block, then their exclusive prefix over the
// blocks.  order[] is a STABLE partition -- a mode's particles in index order --: with an atomic cursor the order inside a mode,
// and with it the sample of particles behind the first workgroup's redraw probe, changed from launch to launch, and two runs
// with the same seed chose different numbers of rounds: equal to rounding only.)
__host__ __device__ static inline size_t bm_mt_words(int64_t n, int K) {
  return 4 + 3 * BM_KMAX + 4 * (size_t)((n + 15) / 16 + K + 1) + (size_t)n + (size_t)((n + 255) / 256) * (size_t)K;
}

static __global__ void __launch_bounds__(256) k_mt_count(const int32_t* __restrict__ assign, int64_t n, int K, int32_t* __restrict__ mt,
                                                         int32_t* __restrict__ blockcnt) {
  __shared__ int s_cnt[BM_KMAX];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int a = i < n ? assign[i] : -1;
  int32_t* mcount = mt + 4 + BM_KMAX;
  if (threadIdx.x < K) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int m = 0; m < K; ++m) {
    const unsigned long long b = __ballot(a == m);
    if (b && (threadIdx.x & 63) == 0) atomicAdd(&s_cnt[m], __popcll(b));
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int c = s_cnt[threadIdx.x];
    blockcnt[(size_t)blockIdx.x * K + threadIdx.x] = c;
    if (c) atomicAdd(&mcount[threadIdx.x], c);
  }
}
static __global__ void __launch_bounds__(256) k_mt_layout(int64_t n, int K, int32_t* __restrict__ mt, int32_t* __restrict__ blockcnt,
                                                          int nblocks) {
  __shared__ int s_start[BM_KMAX + 1], s_toff[BM_KMAX + 1];
  int32_t* mstart = mt + 4;
  const int32_t* mcount = mt + 4 + BM_KMAX;
  int32_t* tiles = mt + 4 + 3 * BM_KMAX;
  if (threadIdx.x == 0) {
    int st = 0, to = 0;
    for (int m = 0; m < K; ++m) {
      s_start[m] = st; s_toff[m] = to;
      mstart[m] = st;
      st += mcount[m];
      to += (mcount[m] + 15) / 16;
    }
    s_start[K] = st; s_toff[K] = to;
    mt[0] = to; mt[1] = K;
  }
  if (threadIdx.x < K) {                       // exclusive prefix of the mode's block counts, in block order
    int run = 0;
    for (int b = 0; b < nblocks; ++b) {
      const int c = blockcnt[(size_t)b * K + threadIdx.x];
      blockcnt[(size_t)b * K + threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
  const int ntiles = s_toff[K];
  for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
    int m = 0;
    while (s_toff[m + 1] <= t) ++m;
    const int j = t - s_toff[m], nm = s_start[m + 1] - s_start[m];
    tiles[4 * t] = m;
    tiles[4 * t + 1] = s_start[m] + 16 * j;
    tiles[4 * t + 2] = nm - 16 * j < 16 ? nm - 16 * j : 16;
    tiles[4 * t + 3] = s_start[m];
  }
}
static __global__ void __launch_bounds__(256) k_mt_scatter(const int32_t* __restrict__ assign, int64_t n, int K, int32_t* __restrict__ mt,
                                                           int64_t tiles_max, const int32_t* __restrict__ blockoff) {
  __shared__ int s_w[4][BM_KMAX];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int a = i < n ? assign[i] : -1;
  const int32_t* mstart = mt + 4;
  int32_t* order = mt + 4 + 3 * BM_KMAX + 4 * tiles_max;
  int below = 0;
  for (int m = 0; m < K; ++m) {
    const unsigned long long b = __ballot(a == m);
    if (lane == 0) s_w[wid][m] = __popcll(b);
    if (a == m) below = __popcll(b & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (a >= 0 && a < K) {
    int off = blockoff[(size_t)blockIdx.x * K + a] + below;
    for (int w = 0; w < wid; ++w) off += s_w[w][a];
    order[mstart[a] + off] = (int32_t)i;
  }
}
// the modes' failure lists of the last round, strung together for the straggler pass: rows_cat[0 .. *cnt_cat)
static __global__ void __launch_bounds__(256) k_mt_concat(const int32_t* __restrict__ mt, const int32_t* __restrict__ cnt_in,
                                                          const int32_t* __restrict__ rows_in, int32_t* __restrict__ cnt_cat,
                                                          int32_t* __restrict__ rows_cat) {
  __shared__ int s_base;
  const int m = blockIdx.x, c = cnt_in[m], st = mt[4 + m];
  if (threadIdx.x == 0) s_base = c ? atomicAdd(cnt_cat, c) : 0;
  __syncthreads();
  for (int e = threadIdx.x; e < c; e += blockDim.x) rows_cat[s_base + e] = rows_in[st + e];
}

template <int KERNEL, int NP, bool HAS_BC, bool MULTI, int MAXT>
__global__ void __launch_bounds__(256) k_propose_blkm(double* __restrict__ u, int64_t n, int64_t ld, int d, const double* __restrict__ means_all,
                                                      const double* __restrict__ Lm_all, const double* __restrict__ Wm_all,
                                                      const double* __restrict__ dof, const double* __restrict__ sigmas,
                                                      const uint8_t* __restrict__ bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                                                      double* __restrict__ up, double* __restrict__ maha_u, double* __restrict__ maha_up,
                                                      uint8_t* __restrict__ pend, const int32_t* __restrict__ cnt_in,
                                                      const int32_t* __restrict__ rows_in, int att, int32_t* __restrict__ cnt_out,
                                                      int32_t* __restrict__ rows_out, const int32_t* __restrict__ mt, int64_t tiles_max, int tries,
                                                      const int32_t* __restrict__ att_in, int32_t* __restrict__ att_out, int fan_div) {
  // cnt_in == NULL: round 0, attempt 0 of every particle and the chores of the step (pending moves, form at u, Gamma scale);
  // cnt_in != NULL: attempt `att` of the particles listed by the round before; the step scale comes from where round 0 parked it.
  // MULTI: several modes -- the wave's tile, its mode and the mode's stretch of the lists come from the tile table mt (above).
  constexpr int NS = 4 * NP;
  __shared__ int s_fail[4], s_mode[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = lane >> 4, nn = lane & 15;
  const bool first = cnt_in == nullptr;
  const int64_t T = (int64_t)blockIdx.x * 4 + wid;
  int mode = 0, lbase = 0;                // the wave's mode; where its mode's stretch of the list arrays starts
  bool live;
  int64_t i;
  // FAN-OUT (one mode, att_out given): a listed particle gets G consecutive columns of a tile -- its next G attempts side by side,
  // the first in bounds in attempt order wins -- with G the largest power of two (<= 16) that keeps the fanned-out list within a
  // fraction 1 / fan_div of the ensemble's columns.  A round over a short list costs one tile's chain of latencies whatever it holds; spent on G attempts
  // per particle instead of one it empties the list G times as fast (config 2 mid-run: 4-6 rounds -> 2).  The attempt the NEXT
  // round starts from travels through att_out / att_in (the width is chosen on the device, from the list's length).
  int G = 1, lgG = 0, att_base = att;
  if (!MULTI) {
    const int64_t slot = T * 16 + nn;
    int64_t total = n;
    if (!first) {
      total = *cnt_in;
      if (att_out) {
        att_base = *att_in;
        while (G < 16 && 2 * (int64_t)fan_div * G * total <= n) { G *= 2; ++lgG; }
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (att == tries && tick.ctl)          // the redraw probe from ALL particles' first `tries` attempts
          const_cast<double*>(tick.ctl)[8] = bm_probe((double)total / (double)n, tries);
        if (att_out) *att_out = att_base + tries * G;
      }
      total *= G;
      if ((int64_t)blockIdx.x * 64 >= total) return;                        // the whole block (uniform): nothing listed for it
    } else if (att_out && blockIdx.x == 0 && threadIdx.x == 0) {
      *att_out = tries;
    }
    live = slot < total;
    i = first ? (live ? slot : n - 1) : (int64_t)rows_in[live ? (slot >> lgG) : 0];  // dead columns shadow a particle, never store
  } else {
    const int ntiles = mt[0];
    const int32_t* tiles = mt + 4 + 3 * BM_KMAX;
    const int32_t* order = tiles + 4 * tiles_max;
    const bool valid = T < ntiles;
    const int t4 = valid ? 4 * (int)T : 0;
    mode = tiles[t4];
    lbase = tiles[t4 + 3];
    const int start = tiles[t4 + 1], count = tiles[t4 + 2];
    if (first) {
      live = valid && nn < count;
      i = order[start + (live ? nn : 0)];
      if (att_out && valid && start == lbase && lane == 0) att_out[mode] = tries;     // (the mode's first tile speaks for it)
    } else {
      const int j16 = start - lbase, cm = cnt_in[mode];
      if (att_out) {                                   // fan-out, per mode: its list against its own particle count
        att_base = att_in[mode];
        const int nm = mt[4 + BM_KMAX + mode];
        while (G < 16 && 2 * (int64_t)fan_div * G * cm <= nm) { G *= 2; ++lgG; }
        if (valid && j16 == 0 && lane == 0) att_out[mode] = att_base + tries * G;
      }
      live = valid && j16 + nn < cm * G;
      i = live ? (int64_t)rows_in[lbase + ((j16 + nn) >> lgG)] : (int64_t)order[start];
      if (att == tries && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl) {
        int64_t total = 0;
        for (int m = 0; m < mt[1]; ++m) total += cnt_in[m];
        const_cast<double*>(tick.ctl)[8] = bm_probe((double)total / (double)n, tries);
      }
    }
  }
  const bool wave_on = __ballot(live) != 0ull;     // a wave without particles skips the arithmetic and only meets the barriers
  if (!first) {                                   // (uniform) blocks whose four tiles are all beyond their lists
    if (lane == 0) s_fail[wid] = wave_on;
    __syncthreads();
    if (!(s_fail[0] | s_fail[1] | s_fail[2] | s_fail[3])) return;
    __syncthreads();
  }
  const double* __restrict__ means = means_all + (size_t)mode * d;
  const double* __restrict__ Lm = Lm_all + (size_t)mode * bm_blocks(NP) * 64;
  const double* __restrict__ Wm = Wm_all + (size_t)mode * bm_blocks(NP) * 64;
  const int npairs = (d + 1) >> 1;
  const double sigma = sigmas[mode];
  const bool carry = KERNEL == TPH_KERNEL_TPCN && tick.carry();
  const double a_fac = (KERNEL == TPH_KERNEL_TPCN) ? tph_sqrt(1.0 - sigma * sigma) : 1.0;
  double X[NS];                                  // the B operands: normals (permuted steps), then rows of the proposal (natural steps)
  bool all_ok = false;

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

  if (wave_on) {
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
        const double nu = dof[mode];
        tph_rng gg(seed, tick, TPH_TAG_GAMMA, (uint64_t)(item0 + i));
        const double gam = tph_gamma_mt(gg, 0.5 * ((double)d + nu)) * tph_div(2.0, nu + m_u);
        b_fac = sigma * tph_sqrt(tph_rcp(gam));
        if (live && k == 0) maha_up[i] = b_fac;        // parked for the later rounds of this particle
      } else if (live && k == 0 && maha_u) {
        maha_u[i] = 0.0;
      }
      // (the four lanes of a column read the flag in the same instruction of one wave: the store is behind their loads)
      if (pd && k == 0) pend[i] = 0;
    } else if (KERNEL == TPH_KERNEL_TPCN) {
      b_fac = maha_up[i];
    }

    // ---- the round's attempts att, att + 1, ... att + tries - 1: every column still out of bounds gets the next one IN PLACE (a
    // tile with one failing column pays a whole pass for it, but late in a run that is a fifth of the tiles for one more pass --
    // against a list launch, a closing pass and their launch latencies for a handful of particles: config 2, 35-70 us per step)
    // (Written out, not looped: inside a loop the compiler keeps every address of the matrix blocks and of the current point
    // live across the body -- 256 VGPRs, one wave per SIMD, 705 against 198 us at 131 072 x 100-D -- where the straight-line form
    // needs 125.)
    bool pending = live;                               // this lane's column still has no in-bounds attempt
    const bool want_form = KERNEL == TPH_KERNEL_TPCN;
    auto attempt = [&](const int t) {
      // the normals of the attempt: lane (k, n) draws pairs k, k + 4, ... of particle n = its operands of steps 2c, 2c + 1
      {
        tph_rng gz(seed, tick, TPH_TAG_NORMAL, (uint64_t)(item0 + i));
        const uint32_t d0 = (uint32_t)(att_base + t * G + (nn & (G - 1))) * (uint32_t)npairs;
#pragma unroll
        for (int c = 0; c < NS / 2; ++c) {
          const int q = k + 4 * c;
          double z0 = 0.0, z1 = 0.0;
          if (pending && q < npairs) gz.normal2(d0 + (uint32_t)q, z0, z1);
          X[2 * c] = z0;
          X[2 * c + 1] = z1;             // (row 2q + 1 == d for odd d: its column of L is zero)
        }
      }
      // rows of the attempt, last panel first: v = fma(b, (L z)_r, mu_r + a (u_r - mu_r)), bounds; results in place.  (The
      // matrix blocks, the current point and the mean are the SAME loads in every attempt: behind an opaque offset each attempt
      // loads them where it uses them -- otherwise the values of the first attempt are kept for the next, 224 + 112 VGPRs.)
      const double* __restrict__ Lq = tph_opaque(Lm);
      const double* __restrict__ uq = tph_opaque((const double*)u);
      const double* __restrict__ mq = tph_opaque(means);
      bool ok = true;
#pragma unroll
      for (int p = NP - 1; p >= 0; --p) {
        bm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4 * p + 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lq[(size_t)bm_blk(p, s) * 64 + lane], X[s], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * p + k + 4 * q;
          double v = 0.0;
          if (r < d) {
            const double ur = uq[(size_t)r * ld + i];
            const double mr = (KERNEL == TPH_KERNEL_TPCN) ? mq[r] : 0.0;
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
      // a column is in bounds when its four lanes are; of a particle's G columns the lowest one in bounds is its proposal
      // (attempts past the redraw cap do not count: the straggler pass proposes the current point for those)
      const unsigned long long okb = __ballot(ok && pending && att_base + t * G + (nn & (G - 1)) < PROP_MAX_ATTEMPTS);
      const unsigned int okt = (unsigned int)(okb & (okb >> 16) & (okb >> 32) & (okb >> 48)) & 0xFFFFu;
      const unsigned int grp = okt & (((1u << G) - 1u) << (nn & ~(G - 1)));
      const bool now_ok = ((okt >> nn) & 1u) && (grp & ((1u << nn) - 1u)) == 0u;
      // outputs of the columns whose attempt is in bounds
      if (now_ok) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int r = 4 * s + k;
          if (r < d) up[(size_t)r * ld + i] = X[s];
        }
      }
      if (want_form) {
        if (__ballot(now_ok) != 0ull) {               // (wave-uniform) some column needs its form |L^-1 (u' - mu)|^2
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const int r = 4 * s + k;
            X[s] = r < d ? X[s] - mq[r] : 0.0;
          }
          const double m_up = form(tph_opaque(Wm));
          if (now_ok && k == 0) maha_up[i] = m_up;
        }
      } else if (now_ok && k == 0 && maha_up) {
        maha_up[i] = 0.0;
      }
      if (grp) { pending = false; all_ok = true; }      // the particle is settled (by this column or by a lower one of its group)
    };
    attempt(0);
    if (MAXT > 1 && tries > 1 && __ballot(pending) != 0ull) attempt(1);
    if (MAXT > 2 && tries > 2 && __ballot(pending) != 0ull) attempt(2);
  }
  {
    // the block's failures take ONE slot range of their mode's list (an atomic per wave -- 16 384 of them on one address at
    // 262 144 particles -- cost 110 us of a 195 us launch with a fifth of the first attempts out of bounds); a block whose
    // four tiles belong to different modes (at most K - 1 of them) falls back to one atomic per wave
    const unsigned long long failb = __ballot(live && !all_ok && k == 0 && (nn & (G - 1)) == 0);
    const int nf = __popcll(failb);
    if (lane == 0) { s_fail[wid] = nf | (__popcll(__ballot(live && k == 0)) << 8); s_mode[wid] = mode; }
    __syncthreads();
    int before = 0, nfail = 0, nlive = 0, m0 = mode;
    bool same = true;
    if (MULTI) {                                         // the mode of the block's failures (tiles without failures do not count)
#pragma unroll
      for (int w = 3; w >= 0; --w) if (s_fail[w] & 255) m0 = s_mode[w];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int f = s_fail[w] & 255;
      before += w < wid ? f : 0;
      nfail += f;
      nlive += s_fail[w] >> 8;
      same = same && (!MULTI || f == 0 || s_mode[w] == m0);
    }
    int slot0;
    if (same) {
      if (threadIdx.x == 0 && nfail) s_base = atomicAdd(cnt_out + m0, nfail);
      __syncthreads();
      slot0 = s_base + before;
    } else {
      __syncthreads();
      int b0 = 0;
      if (lane == 0 && nf) b0 = atomicAdd(cnt_out + mode, nf);
      slot0 = __shfl(b0, 0, 64);
    }
    if ((failb >> lane) & 1ull) rows_out[lbase + slot0 + __popcll(failb & ((1ull << lane) - 1ull))] = (int32_t)i;
    if (first && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl) {      // regime probe: mean attempts implied by this block's failures, 1 / (1 - f)
      const_cast<double*>(tick.ctl)[8] = bm_probe((double)nfail / fmax(1.0, (double)nlive), tries);
    }
  }
}

template <int KERNEL, int NP, bool HAS_BC>
__global__ void __launch_bounds__(256) k_propose_blkm_lds(double* __restrict__ u, int64_t n, int64_t ld, int d, const double* __restrict__ means_all,
                                                      const double* __restrict__ Lm_all, const double* __restrict__ Wm_all,
                                                      const double* __restrict__ dof, const double* __restrict__ sigmas,
                                                      const uint8_t* __restrict__ bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                                                      double* __restrict__ up, double* __restrict__ maha_u, double* __restrict__ maha_up,
                                                      uint8_t* __restrict__ pend, const int32_t* __restrict__ cnt_in,
                                                      const int32_t* __restrict__ rows_in, int att, int32_t* __restrict__ cnt_out,
                                                      int32_t* __restrict__ rows_out, const int32_t* __restrict__ mt, int64_t tiles_max, int tries,
                                                      const int32_t* __restrict__ att_in, int32_t* __restrict__ att_out, int fan_div) {
  // cnt_in == NULL: round 0, attempt 0 of every particle and the chores of the step (pending moves, form at u, Gamma scale);
  // cnt_in != NULL: attempt `att` of the particles listed by the round before; the step scale comes from where round 0 parked it.
  // MULTI: several modes -- the wave's tile, its mode and the mode's stretch of the lists come from the tile table mt (above).
  constexpr int NS = 4 * NP;
  constexpr bool MULTI = false;
  extern __shared__ double bm_lds[];          // two panel buffers of NS blocks (64 doubles each)
  __shared__ int s_fail[4], s_mode[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = lane >> 4, nn = lane & 15;
  const bool first = cnt_in == nullptr;
  const int64_t T = (int64_t)blockIdx.x * 4 + wid;
  int mode = 0, lbase = 0;                // the wave's mode; where its mode's stretch of the list arrays starts
  bool live;
  int64_t i;
  // FAN-OUT (one mode, att_out given): a listed particle gets G consecutive columns of a tile -- its next G attempts side by side,
  // the first in bounds in attempt order wins -- with G the largest power of two (<= 16) that keeps the fanned-out list within a
  // fraction 1 / fan_div of the ensemble's columns.  A round over a short list costs one tile's chain of latencies whatever it holds; spent on G attempts
  // per particle instead of one it empties the list G times as fast (config 2 mid-run: 4-6 rounds -> 2).  The attempt the NEXT
  // round starts from travels through att_out / att_in (the width is chosen on the device, from the list's length).
  int G = 1, lgG = 0, att_base = att;
  if (!MULTI) {
    const int64_t slot = T * 16 + nn;
    int64_t total = n;
    if (!first) {
      total = *cnt_in;
      if (att_out) {
        att_base = *att_in;
        while (G < 16 && 2 * (int64_t)fan_div * G * total <= n) { G *= 2; ++lgG; }
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (att == tries && tick.ctl)          // the redraw probe from ALL particles' first `tries` attempts
          const_cast<double*>(tick.ctl)[8] = bm_probe((double)total / (double)n, tries);
        if (att_out) *att_out = att_base + tries * G;
      }
      total *= G;
      if ((int64_t)blockIdx.x * 64 >= total) return;                        // the whole block (uniform): nothing listed for it
    } else if (att_out && blockIdx.x == 0 && threadIdx.x == 0) {
      *att_out = tries;
    }
    live = slot < total;
    i = first ? (live ? slot : n - 1) : (int64_t)rows_in[live ? (slot >> lgG) : 0];  // dead columns shadow a particle, never store
  } else {
    const int ntiles = mt[0];
    const int32_t* tiles = mt + 4 + 3 * BM_KMAX;
    const int32_t* order = tiles + 4 * tiles_max;
    const bool valid = T < ntiles;
    const int t4 = valid ? 4 * (int)T : 0;
    mode = tiles[t4];
    lbase = tiles[t4 + 3];
    const int start = tiles[t4 + 1], count = tiles[t4 + 2];
    if (first) {
      live = valid && nn < count;
      i = order[start + (live ? nn : 0)];
      if (att_out && valid && start == lbase && lane == 0) att_out[mode] = tries;     // (the mode's first tile speaks for it)
    } else {
      const int j16 = start - lbase, cm = cnt_in[mode];
      if (att_out) {                                   // fan-out, per mode: its list against its own particle count
        att_base = att_in[mode];
        const int nm = mt[4 + BM_KMAX + mode];
        while (G < 16 && 2 * (int64_t)fan_div * G * cm <= nm) { G *= 2; ++lgG; }
        if (valid && j16 == 0 && lane == 0) att_out[mode] = att_base + tries * G;
      }
      live = valid && j16 + nn < cm * G;
      i = live ? (int64_t)rows_in[lbase + ((j16 + nn) >> lgG)] : (int64_t)order[start];
      if (att == tries && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl) {
        int64_t total = 0;
        for (int m = 0; m < mt[1]; ++m) total += cnt_in[m];
        const_cast<double*>(tick.ctl)[8] = bm_probe((double)total / (double)n, tries);
      }
    }
  }
  const bool wave_on = __ballot(live) != 0ull;     // a wave without particles skips the arithmetic and only meets the barriers
  if (!first) {                                   // (uniform) blocks whose four tiles are all beyond their lists
    if (lane == 0) s_fail[wid] = wave_on;
    __syncthreads();
    if (!(s_fail[0] | s_fail[1] | s_fail[2] | s_fail[3])) return;
    __syncthreads();
  }
  const double* __restrict__ means = means_all + (size_t)mode * d;
  const double* __restrict__ Lm = Lm_all + (size_t)mode * bm_blocks(NP) * 64;
  const double* __restrict__ Wm = Wm_all + (size_t)mode * bm_blocks(NP) * 64;
  const int npairs = (d + 1) >> 1;
  const double sigma = sigmas[mode];
  const bool carry = KERNEL == TPH_KERNEL_TPCN && tick.carry();
  const double a_fac = (KERNEL == TPH_KERNEL_TPCN) ? tph_sqrt(1.0 - sigma * sigma) : 1.0;
  double X[NS];                                  // the B operands: normals (permuted steps), then rows of the proposal (natural steps)
  bool all_ok = false;

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

  double b_fac_s = sigma;
  if (wave_on) {
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
        const double nu = dof[mode];
        tph_rng gg(seed, tick, TPH_TAG_GAMMA, (uint64_t)(item0 + i));
        const double gam = tph_gamma_mt(gg, 0.5 * ((double)d + nu)) * tph_div(2.0, nu + m_u);
        b_fac = sigma * tph_sqrt(tph_rcp(gam));
        if (live && k == 0) maha_up[i] = b_fac;        // parked for the later rounds of this particle
      } else if (live && k == 0 && maha_u) {
        maha_u[i] = 0.0;
      }
      // (the four lanes of a column read the flag in the same instruction of one wave: the store is behind their loads)
      if (pd && k == 0) pend[i] = 0;
    } else if (KERNEL == TPH_KERNEL_TPCN) {
      b_fac = maha_up[i];
    }

    // ---- the round's attempts att, att + 1, ... att + tries - 1: every column still out of bounds gets the next one IN PLACE (a
    // tile with one failing column pays a whole pass for it, but late in a run that is a fifth of the tiles for one more pass --
    // against a list launch, a closing pass and their launch latencies for a handful of particles: config 2, 35-70 us per step)
    // (Written out, not looped: inside a loop the compiler keeps every address of the matrix blocks and of the current point
    // live across the body -- 256 VGPRs, one wave per SIMD, 705 against 198 us at 131 072 x 100-D -- where the straight-line form
    // needs 125.)
    b_fac_s = b_fac;
  }
  // ---- the round's ONE attempt, every wave of the workgroup in step (a wave without particles multiplies zeros): the matrix
  // blocks of a panel are staged in LDS ONCE per workgroup -- loaded into registers while the previous panel's products run,
  // stored behind them, one barrier per panel, two buffers -- instead of streamed from L1 / L2 by each of the four waves
  // (512 bytes per instruction and wave: the operand stream, not the matrix cores, bounded the un-staged kernel).
  {
    const double b_fac = b_fac_s;
    bool pending = live;
    const bool want_form = KERNEL == TPH_KERNEL_TPCN;
    double stg[NP];                                           // this thread's share of the next panel: (4 p + 4) blocks x 64 / 256 = p + 1 values
    auto stage_load = [&](const double* __restrict__ Tb, const int p) {
      const double* src = Tb + (size_t)bm_blk(p, 0) * 64;
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if (k <= p) stg[k] = src[threadIdx.x + 256 * k];
    };
    auto stage_store = [&](const int p, const int buf) {
      double* dst = bm_lds + (size_t)buf * NS * 64;
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if (k <= p) dst[threadIdx.x + 256 * k] = stg[k];
    };
    {
      // the normals of the attempt: lane (k, n) draws pairs k, k + 4, ... of particle n = its operands of steps 2c, 2c + 1
      stage_load(Lm, NP - 1);
      tph_rng gz(seed, tick, TPH_TAG_NORMAL, (uint64_t)(item0 + i));
      const uint32_t d0 = (uint32_t)(att_base + (nn & (G - 1))) * (uint32_t)npairs;
#pragma unroll
      for (int c = 0; c < NS / 2; ++c) {
        const int q = k + 4 * c;
        double z0 = 0.0, z1 = 0.0;
        if (pending && q < npairs) gz.normal2(d0 + (uint32_t)q, z0, z1);
        X[2 * c] = z0;
        X[2 * c + 1] = z1;
      }
      stage_store(NP - 1, 0);
      __syncthreads();
    }
    const double* __restrict__ uq = (const double*)u;
    const double* __restrict__ mq = means;
    bool ok = true;
    int buf = 0;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p) {
      if (p > 0) stage_load(Lm, p - 1);
      else if (want_form) stage_load(Wm, 0);
      const double* __restrict__ Ls = bm_lds + (size_t)buf * NS * 64;
      bm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s2 = 0; s2 < 4 * p + 4; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[s2 * 64 + lane], X[s2], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * p + k + 4 * q;
        double v = 0.0;
        if (r < d) {
          const double ur = uq[(size_t)r * ld + i];
          const double mr = (KERNEL == TPH_KERNEL_TPCN) ? mq[r] : 0.0;
          const double base = (KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, ur - mr, mr) : ur;
          v = fma(b_fac, acc[q], base);
          const uint8_t f = HAS_BC ? bc[r] : (uint8_t)TPH_BC_STRICT;
          if (f == TPH_BC_PERIODIC) v = bc_periodic(v);
          else if (f == TPH_BC_REFLECTIVE) v = bc_reflective(v);
          else ok = ok && (v >= 0.0) && (v <= 1.0);
        }
        X[4 * p + q] = v;
      }
      if (p > 0) stage_store(p - 1, buf ^ 1);
      else if (want_form) stage_store(0, buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    const unsigned long long okb = __ballot(ok && pending && att_base + (nn & (G - 1)) < PROP_MAX_ATTEMPTS);
    const unsigned int okt = (unsigned int)(okb & (okb >> 16) & (okb >> 32) & (okb >> 48)) & 0xFFFFu;
    const unsigned int grp = okt & (((1u << G) - 1u) << (nn & ~(G - 1)));
    const bool now_ok = ((okt >> nn) & 1u) && (grp & ((1u << nn) - 1u)) == 0u;
    if (now_ok) {
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const int r = 4 * s2 + k;
        if (r < d) up[(size_t)r * ld + i] = X[s2];
      }
    }
    if (want_form) {
      // |L^-1 (u' - mu)|^2 of every column (those out of bounds are not stored): the panels of L^-1 in ascending order, staged alike
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const int r = 4 * s2 + k;
        X[s2] = r < d ? X[s2] - mq[r] : 0.0;
      }
      double part = 0.0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (p + 1 < NP) stage_load(Wm, p + 1);
        const double* __restrict__ Ws = bm_lds + (size_t)buf * NS * 64;
        bm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < 4 * p + 4; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[s2 * 64 + lane], X[s2], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) part = fma(acc[q], acc[q], part);
        if (p + 1 < NP) stage_store(p + 1, buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      if (now_ok && k == 0) maha_up[i] = part;
    } else if (now_ok && k == 0 && maha_up) {
      maha_up[i] = 0.0;
    }
    if (grp) { pending = false; all_ok = true; }
  }
  {
    // the block's failures take ONE slot range of their mode's list (an atomic per wave -- 16 384 of them on one address at
    // 262 144 particles -- cost 110 us of a 195 us launch with a fifth of the first attempts out of bounds); a block whose
    // four tiles belong to different modes (at most K - 1 of them) falls back to one atomic per wave
    const unsigned long long failb = __ballot(live && !all_ok && k == 0 && (nn & (G - 1)) == 0);
    const int nf = __popcll(failb);
    if (lane == 0) { s_fail[wid] = nf | (__popcll(__ballot(live && k == 0)) << 8); s_mode[wid] = mode; }
    __syncthreads();
    int before = 0, nfail = 0, nlive = 0, m0 = mode;
    bool same = true;
    if (MULTI) {                                         // the mode of the block's failures (tiles without failures do not count)
#pragma unroll
      for (int w = 3; w >= 0; --w) if (s_fail[w] & 255) m0 = s_mode[w];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int f = s_fail[w] & 255;
      before += w < wid ? f : 0;
      nfail += f;
      nlive += s_fail[w] >> 8;
      same = same && (!MULTI || f == 0 || s_mode[w] == m0);
    }
    int slot0;
    if (same) {
      if (threadIdx.x == 0 && nfail) s_base = atomicAdd(cnt_out + m0, nfail);
      __syncthreads();
      slot0 = s_base + before;
    } else {
      __syncthreads();
      int b0 = 0;
      if (lane == 0 && nf) b0 = atomicAdd(cnt_out + mode, nf);
      slot0 = __shfl(b0, 0, 64);
    }
    if ((failb >> lane) & 1ull) rows_out[lbase + slot0 + __popcll(failb & ((1ull << lane) - 1ull))] = (int32_t)i;
    if (first && blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl) {      // regime probe: mean attempts implied by this block's failures, 1 / (1 - f)
      const_cast<double*>(tick.ctl)[8] = bm_probe((double)nfail / fmax(1.0, (double)nlive), tries);
    }
  }
}


// TPH_OPT_BLK_TRIES (0 = by dimension: a retry in place costs a whole tile pass, which pays below n_dim 64 -- regime sweep:
// 65 536 x 50-D 71 -> 62 us, 262 144 x 32-D 99 -> 79 us at one attempt per particle; 131 072 x 100-D 479 -> 598 us at 1.13)
// (With the list rounds fanned out -- TPH_OPT_BLK_FAN -- a retry in place only pays at n_dim <= 32: 65 536 x 50-D at 1.3 / 1.8
// estimated attempts per particle 145 / 205 us with one try against 171 / 235 us with two; 262 144 x 32-D at 2.6: 426 against 376.)
static inline int bm_tries(const tph_ctx* ctx) { return ctx->blk_tries > 0 ? (ctx->blk_tries > 3 ? 3 : ctx->blk_tries) : (ctx->d > 32 ? 1 : 2); }

// One round of the matrix-core blocked kernel on the ctx stream (mutate.hip drives the rounds of the one-mode path; several
// modes: tph_blkm_propose_multi below).  The blocked copies of L (permuted steps) and L^-1 (natural steps) of every mode live in
// a ctx-owned buffer, rebuilt when the caller's mode statistics change (and always under stream capture: a replayed step never
// re-enters this host code).
template <int KERNEL>
static int blkm_pack(tph_ctx* ctx, int K, const double* chol, const double* winv, double** Lm, double** Wm) {
  const int d = ctx->d, np = bm_panels(d);
  const size_t one = sizeof(double) * (size_t)bm_blocks(np) * 64 * (size_t)K;
  if (ctx->bm_bytes < 2 * one) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->bm_buf) ctx->retired.push_back(ctx->bm_buf);
    ctx->bm_buf = nullptr; ctx->bm_bytes = 0; ctx->bm_epoch = -1;
    TPH_HIP(hipMalloc((void**)&ctx->bm_buf, 2 * one));
    ctx->bm_bytes = 2 * one;
  }
  *Lm = (double*)ctx->bm_buf;
  *Wm = *Lm + (size_t)bm_blocks(np) * 64 * (size_t)K;
  return 0;
}
template <int KERNEL>
static int blkm_refresh(tph_ctx* ctx, int K, const double* chol, const double* winv, double* Lm, double* Wm, bool* rebuilt) {
  const int d = ctx->d, np = bm_panels(d);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  TPH_HIP(hipStreamIsCapturing(ctx->stream, &cap));
  const bool capturing = cap != hipStreamCaptureStatusNone;
  *rebuilt = capturing || ctx->modes_epoch <= 0 || ctx->bm_epoch != ctx->modes_epoch || ctx->bm_src != (const void*)chol ||
             ctx->bm_kernel != KERNEL || ctx->bm_K != K;
  if (*rebuilt) {
    hipLaunchKernelGGL(k_blkm_pack, dim3(K), dim3(256), 0, ctx->stream, chol, d, np, 1, Lm);
    if (KERNEL == TPH_KERNEL_TPCN) hipLaunchKernelGGL(k_blkm_pack, dim3(K), dim3(256), 0, ctx->stream, winv, d, np, 0, Wm);
    ctx->bm_epoch = capturing ? -1 : ctx->modes_epoch; ctx->bm_src = (const void*)chol; ctx->bm_kernel = KERNEL; ctx->bm_K = K;
  }
  return 0;
}

template <int KERNEL, bool MULTI>
static int blkm_launch(tph_ctx* ctx, int64_t blocks, double* u, int64_t n, int64_t ld, const double* means, const double* Lm, const double* Wm,
                       const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                       double* up, double* mu_, double* mup, uint8_t* pend, const int32_t* cnt_in, const int32_t* rows_in, int att,
                       int32_t* cnt_out, int32_t* rows_out, const int32_t* mt, int64_t tiles_max, const int32_t* att_in = nullptr,
                       int32_t* att_out = nullptr) {
  const int d = ctx->d, np = bm_panels(d);
  const int tries = bm_tries(ctx);
  const int fan_div = ctx->blk_fan == 3 ? 4 : (ctx->blk_fan == 2 ? 1 : 2);      // TPH_OPT_BLK_FAN: 1 = half of the columns, 2 = all, 3 = a quarter
  const dim3 grid((unsigned)blocks);
#define TPH_BM(NPV, BC, MT)                                                                                              \
  hipLaunchKernelGGL((k_propose_blkm<KERNEL, NPV, BC, MULTI, MT>), grid, dim3(256), 0, ctx->stream, u, n, ld, d, means, Lm, Wm, dof, \
                     sigmas, bc, seed, tick, item0, up, mu_, mup, pend, cnt_in, rows_in, att, cnt_out, rows_out, mt, tiles_max, tries, att_in, att_out, fan_div)
  // (the one-attempt instantiation keeps four waves per SIMD at n_dim = 100: 198 against 247 us at 131 072 particles)
#define TPH_BM_NP(NPV)                                                                                                   \
  do {                                                                                                                   \
    if (tries > 1) { if (bc) TPH_BM(NPV, true, 3); else TPH_BM(NPV, false, 3); }                                         \
    else { if (bc) TPH_BM(NPV, true, 1); else TPH_BM(NPV, false, 1); }                                                   \
  } while (0)
  if (!MULTI && tries == 1 && ctx->blk_stage) {
    // TPH_OPT_BLK_STAGE: the panels' matrix blocks staged in LDS once per workgroup (k_propose_blkm_lds)
    const size_t lds = sizeof(double) * 2 * (size_t)(4 * np) * 64;
#define TPH_BML(NPV, BC)                                                                                                 \
  hipLaunchKernelGGL((k_propose_blkm_lds<KERNEL, NPV, BC>), grid, dim3(256), lds, ctx->stream, u, n, ld, d, means, Lm, Wm, dof, \
                     sigmas, bc, seed, tick, item0, up, mu_, mup, pend, cnt_in, rows_in, att, cnt_out, rows_out, mt, tiles_max, tries, att_in, att_out, fan_div)
#define TPH_BML_NP(NPV) do { if (bc) TPH_BML(NPV, true); else TPH_BML(NPV, false); } while (0)
    switch (np) {
      case 2: TPH_BML_NP(2); break;
      case 3: TPH_BML_NP(3); break;
      case 4: TPH_BML_NP(4); break;
      case 5: TPH_BML_NP(5); break;
      case 6: TPH_BML_NP(6); break;
      default: TPH_BML_NP(7); break;
    }
#undef TPH_BML_NP
#undef TPH_BML
    TPH_LAUNCH_CHECK();
    return 0;
  }
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

template <int KERNEL>
static int blkm_round(tph_ctx* ctx, double* u, int64_t n, int64_t ld, const double* means, const double* chol, const double* winv,
                      const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                      double* up, double* mu_, double* mup, uint8_t* pend, const int32_t* cnt_in, const int32_t* rows_in, int att,
                      int32_t* cnt_out, int32_t* rows_out, const int32_t* att_in, int32_t* att_out) {
  const int d = ctx->d;
  TPH_REQUIRE(d > 16 && d <= 112, "tph_propose (blocked, matrix cores): n_dim=%d outside 17..112", d);
  double *Lm, *Wm;
  if (blkm_pack<KERNEL>(ctx, 1, chol, winv, &Lm, &Wm)) return -1;
  if (cnt_in == nullptr) {
    bool rebuilt;
    if (blkm_refresh<KERNEL>(ctx, 1, chol, winv, Lm, Wm, &rebuilt)) return -1;
  }
  return blkm_launch<KERNEL, false>(ctx, (n + 63) / 64, u, n, ld, means, Lm, Wm, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend,
                                    cnt_in, rows_in, att, cnt_out, rows_out, nullptr, 0, att_in, att_out);
}

// ---- the forms |L^-1 (u' - mu)|^2 of a finished step on the matrix cores (tpCN, one mode) -------------------------------------
// What the closing pass behind the screened batches needs (maha_tile.h, MODE 1: a lane per particle, L^-1 through the scalar cache,
// waves waiting on it 80 % of their cycles) in the arithmetic of the blocked rounds: a wave per 16 particles, the natural-step
// blocks of L^-1 (k_blkm_pack, perm = 0), panels ascending -- the `form` of k_propose_blkm instruction for instruction, so that
// the form of a proposal no longer depends on which kernel found it.  todo_cnt != NULL: the listed particles only.
// MEASURED on config 5's shard (131 072 x 100-D, 2929 launches): 49.6 us per launch against 54.0 us for the lane-per-particle pass --
// every wave streams the 57 KB of blocks from L2, 470 MB per launch; with the blocks staged in LDS per workgroup (two workgroups per
// CU at 100-D) 65.3 us.  A gain of half a per cent of the run: an option (TPH_OPT_FORMS_MFMA), off by default.
template <int NP>
__global__ void __launch_bounds__(256) k_forms_mfma(const double* __restrict__ up, int64_t n, int64_t ld, int d, const double* __restrict__ means,
                                                    const double* __restrict__ Wm, double* __restrict__ maha, tph_stepctl tick,
                                                    const unsigned long long* __restrict__ queue, const int32_t* __restrict__ todo_cnt,
                                                    const int32_t* __restrict__ todo_rows, const int32_t* __restrict__ todo_off) {
  constexpr int NS = 4 * NP;
  if (todo_off) todo_rows += *todo_off;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, k = lane >> 4, nn = lane & 15;
  if (blockIdx.x == 0 && threadIdx.x == 0 && tick.ctl && queue)      // the step's mean attempts per particle (regime probe)
    const_cast<double*>(tick.ctl)[8] = queue[2] ? (double)queue[1] / (double)queue[2] : 0.0;
  int64_t idx = ((int64_t)blockIdx.x * 4 + wid) * 16 + nn, lim = n;
  if (todo_cnt) lim = *todo_cnt;
  if (((int64_t)blockIdx.x * 4 + wid) * 16 >= lim) return;           // (the whole wave)
  const bool live = idx < lim;
  int64_t i = live ? idx : lim - 1;
  if (todo_cnt) i = (int64_t)todo_rows[i];
  double X[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = 4 * s + k;
    X[s] = r < d ? up[(size_t)r * ld + i] - means[r] : 0.0;
  }
  double part = 0.0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    bm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * p + 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wm[(size_t)bm_blk(p, s) * 64 + lane], X[s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) part = fma(acc[q], acc[q], part);
  }
  part += __shfl_xor(part, 16, 64);
  part += __shfl_xor(part, 32, 64);
  if (live && k == 0) maha[i] = part;
}

int tph_blkm_forms(tph_ctx* ctx, const double* up, int64_t n, int64_t ld, const double* means, const double* chol, const double* winv,
                   double* maha, tph_stepctl tick, const unsigned long long* queue, const int32_t* todo_cnt, const int32_t* todo_rows,
                   const int32_t* todo_off) {
  const int d = ctx->d, np = bm_panels(d);
  TPH_REQUIRE(d > 16 && d <= 112, "tph_blkm_forms: n_dim=%d outside 17..112", d);
  double *Lm, *Wm;
  if (blkm_pack<TPH_KERNEL_TPCN>(ctx, 1, chol, winv, &Lm, &Wm)) return -1;
  bool rebuilt;
  if (blkm_refresh<TPH_KERNEL_TPCN>(ctx, 1, chol, winv, Lm, Wm, &rebuilt)) return -1;
  const dim3 grid((unsigned)((n + 63) / 64));
#define TPH_FM(NPV) hipLaunchKernelGGL((k_forms_mfma<NPV>), grid, dim3(256), 0, ctx->stream, up, n, ld, d, means, (const double*)Wm, maha, tick, queue, todo_cnt, todo_rows, todo_off)
  switch (np) {
    case 2: TPH_FM(2); break;
    case 3: TPH_FM(3); break;
    case 4: TPH_FM(4); break;
    case 5: TPH_FM(5); break;
    case 6: TPH_FM(6); break;
    default: TPH_FM(7); break;
  }
#undef TPH_FM
  TPH_LAUNCH_CHECK();
  return 0;
}

int tph_blkm_tries(const tph_ctx* ctx) { return bm_tries(ctx); }

int tph_blkm_round(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                   const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                   const double* ctl, int64_t item0, double* up, double* mu_, double* mup, uint8_t* pend, const int32_t* cnt_in,
                   const int32_t* rows_in, int att, int32_t* cnt_out, int32_t* rows_out, const int32_t* att_in, int32_t* att_out) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return blkm_round<TPH_KERNEL_TPCN>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend, cnt_in,
                                       rows_in, att, cnt_out, rows_out, att_in, att_out);
  return blkm_round<TPH_KERNEL_RWM>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend, cnt_in,
                                    rows_in, att, cnt_out, rows_out, att_in, att_out);
}

// The grouping of the particles by mode in the ctx-owned block (layout above), rebuilt when the caller's statistics version
// (TPH_OPT_MODES_EPOCH), the assignments' address or n change -- the assignments are fixed while the statistics are.
static int mode_table(tph_ctx* ctx, const int32_t* assign, int64_t n, int K) {
  const int64_t tiles_max = (n + 15) / 16 + K + 1;
  const size_t mtw = bm_mt_words(n, K), cntw = (size_t)2 * (24 + 2) * BM_KMAX;
  const size_t need = sizeof(int32_t) * (mtw + cntw + 3 * (size_t)n + 64);
  bool rebuild = ctx->modes_epoch <= 0 || ctx->mt_epoch != ctx->modes_epoch || ctx->mt_assign != (const void*)assign || ctx->mt_n != n ||
                 ctx->mt_K != K;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  TPH_HIP(hipStreamIsCapturing(ctx->stream, &cap));
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (capturing) rebuild = true;                    // a replayed step never re-enters this host code
  if (ctx->mt_bytes < need) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->mt_buf) ctx->retired.push_back(ctx->mt_buf);
    ctx->mt_buf = nullptr; ctx->mt_bytes = 0;
    TPH_HIP(hipMalloc((void**)&ctx->mt_buf, need));
    ctx->mt_bytes = need;
    rebuild = true;
  }
  if (rebuild) {
    int32_t* mt = (int32_t*)ctx->mt_buf;
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, ctx->stream, (unsigned int*)mt, 4 + 3 * BM_KMAX);
    const unsigned gb = (unsigned)((n + 255) / 256);
    int32_t* blockcnt = mt + 4 + 3 * BM_KMAX + 4 * tiles_max + n;
    hipLaunchKernelGGL(k_mt_count, dim3(gb), dim3(256), 0, ctx->stream, assign, n, K, mt, blockcnt);
    hipLaunchKernelGGL(k_mt_layout, dim3(1), dim3(256), 0, ctx->stream, n, K, mt, blockcnt, (int)gb);
    hipLaunchKernelGGL(k_mt_scatter, dim3(gb), dim3(256), 0, ctx->stream, assign, n, K, mt, tiles_max, (const int32_t*)blockcnt);
    TPH_LAUNCH_CHECK();
    ctx->mt_assign = (const void*)assign; ctx->mt_n = n; ctx->mt_K = K;
    ctx->mt_epoch = capturing ? -1 : ctx->modes_epoch;
  }
  return 0;
}
int tph_mode_lists(tph_ctx* ctx, const int32_t* assign, int64_t n, int K, const int32_t** order, const int32_t** mstart, const int32_t** mcount) {
  TPH_REQUIRE(assign && K >= 1 && K <= BM_KMAX && n > 0, "mode lists: bad argument");
  if (mode_table(ctx, assign, n, K)) return -1;
  const int32_t* mt = (const int32_t*)ctx->mt_buf;
  const int64_t tiles_max = (n + 15) / 16 + K + 1;
  *mstart = mt + 4;
  *mcount = mt + 4 + BM_KMAX;
  *order = mt + 4 + 3 * BM_KMAX + 4 * tiles_max;
  return 0;
}

// Several modes: all `rounds` rounds of the blocked path over mode-pure tiles; whoever is still out of bounds afterwards is left
// in ONE list (*todo_cnt entries of todo_rows, both device pointers into ctx-owned memory) for the caller's straggler pass.
template <int KERNEL>
static int blkm_multi(tph_ctx* ctx, double* u, const int32_t* assign, int64_t n, int64_t ld, int K, const double* means, const double* chol,
                      const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick,
                      int64_t item0, double* up, double* mu_, double* mup, uint8_t* pend, int rounds, const int32_t** todo_cnt,
                      const int32_t** todo_rows, const int32_t** todo_att, const int32_t** per_mode) {
  const int d = ctx->d;
  TPH_REQUIRE(d > 16 && d <= 112 && K >= 1 && K <= BM_KMAX, "tph_propose (blocked, several modes): n_dim=%d / K=%d out of range", d, K);
  TPH_REQUIRE(n < (1ll << 31), "tph_propose (blocked): %lld particles on one device", (long long)n);
  double *Lm, *Wm;
  if (blkm_pack<KERNEL>(ctx, K, chol, winv, &Lm, &Wm)) return -1;
  bool rebuilt;
  if (blkm_refresh<KERNEL>(ctx, K, chol, winv, Lm, Wm, &rebuilt)) return -1;
  // tile table + order | counters of the rounds [rounds + 1][K] | the two list arrays | the concatenated list
  const int64_t tiles_max = (n + 15) / 16 + K + 1;
  const size_t mtw = bm_mt_words(n, K), cntw = (size_t)2 * (24 + 2) * BM_KMAX;      // counters [26][K] | next attempts [26][K]
  if (mode_table(ctx, assign, n, K)) return -1;
  int32_t* mt = (int32_t*)ctx->mt_buf;
  int32_t* cnts = mt + mtw;
  int32_t* rows[2] = {cnts + cntw, cnts + cntw + n};
  int32_t* rows_cat = cnts + cntw + 2 * n;
  if (rounds < 1) rounds = 1;
  if (rounds > 24) rounds = 24;
  hipLaunchKernelGGL(k_zero_words, dim3(4), dim3(64), 0, ctx->stream, (unsigned int*)cnts, (int)cntw);
  const int64_t blocks = (tiles_max + 3) / 4;
  int32_t* atts = cnts + (size_t)26 * BM_KMAX;        // [round][mode]: the first attempt the mode's list of that round has not tried
  const bool fan = ctx->blk_fan != 0;
  for (int k = 0; k < rounds; ++k)
    if (blkm_launch<KERNEL, true>(ctx, blocks, u, n, ld, means, Lm, Wm, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend,
                                  k ? cnts + (size_t)(k - 1) * BM_KMAX : (const int32_t*)nullptr, (const int32_t*)rows[(k + 1) & 1],
                                  k * bm_tries(ctx), cnts + (size_t)k * BM_KMAX, rows[k & 1], mt, tiles_max,
                                  (fan && k) ? atts + (size_t)(k - 1) * BM_KMAX : (const int32_t*)nullptr,
                                  fan ? atts + (size_t)k * BM_KMAX : (int32_t*)nullptr))
      return -1;
  *todo_att = fan ? atts + (size_t)(rounds - 1) * BM_KMAX : nullptr;
  if (per_mode) {
    // the modes' failure lists as they are: mode m's entries [mstart[m], mstart[m] + cnt[m]) of the last round's list array
    // (the per-mode screened launches of propose_mf.hip read offset and length on the device)
    per_mode[0] = cnts + (size_t)(rounds - 1) * BM_KMAX;
    per_mode[1] = rows[(rounds - 1) & 1];
    per_mode[2] = mt + 4;
    *todo_cnt = nullptr;
    *todo_rows = nullptr;
    return 0;
  }
  int32_t* cnt_cat = cnts + (size_t)25 * BM_KMAX;
  hipLaunchKernelGGL(k_mt_concat, dim3(K), dim3(256), 0, ctx->stream, (const int32_t*)mt, (const int32_t*)(cnts + (size_t)(rounds - 1) * BM_KMAX),
                     (const int32_t*)rows[(rounds - 1) & 1], cnt_cat, rows_cat);
  TPH_LAUNCH_CHECK();
  *todo_cnt = cnt_cat;
  *todo_rows = rows_cat;
  return 0;
}

int tph_blkm_multi(tph_ctx* ctx, int kernel, double* u, const int32_t* assign, int64_t n, int64_t ld, int K, const double* means,
                   const double* chol, const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed,
                   uint32_t tick0, const double* ctl, int64_t item0, double* up, double* mu_, double* mup, uint8_t* pend, int rounds,
                   const int32_t** todo_cnt, const int32_t** todo_rows, const int32_t** todo_att, const int32_t** per_mode) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return blkm_multi<TPH_KERNEL_TPCN>(ctx, u, assign, n, ld, K, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend,
                                       rounds, todo_cnt, todo_rows, todo_att, per_mode);
  return blkm_multi<TPH_KERNEL_RWM>(ctx, u, assign, n, ld, K, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, mu_, mup, pend,
                                    rounds, todo_cnt, todo_rows, todo_att, per_mode);
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_propose_blkm(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
