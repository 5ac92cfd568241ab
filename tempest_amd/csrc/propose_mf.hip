// Redraw-dominated proposals at 16 < d <= 112, one mode: attempts are SCREENED in batches on the matrix cores and only the
// attempts that pass are evaluated in FP64.
// Reference: tempest/mcmc.py:225-249 (tpCN), :301-312 (RWM): a walker's proposal is REDRAWN until it lies in the unit cube.
//
// Early in a high-dimensional run a step is tens to hundreds of attempts per particle (50-D at sigma_0: ~60; 100-D: ~290), and
// all but one of them are thrown away -- the only thing the algorithm wants to know about them is THAT some coordinate left
// [0, 1].  That question does not need FP64: with x~_r a low-precision value of coordinate r and m_r a bound of
// |x~_r - x_r| (every term derived, except the accuracy of the hardware's FP32 transcendentals in the Box-Muller pair, which is
// MEASURED on gfx950 and re-checked on every context before the screen is first used: mf_selftest below), "x~_r < -m_r or x~_r > 1 + m_r" implies that the FP64 kernels find row r out of bounds too, i.e. the attempt
// fails in FP64 as well.  An attempt the screen cannot kill is evaluated in FP64, by the arithmetic of the other proposal
// kernels (same Philox counters, ascending-j FMA chain, v = fma(b, (L z)_r, base_r)), and THAT evaluation decides whether
// it is the proposal.  So the screen only ever removes work: the proposal is the one the sequential loop returns, bit for bit
// equal to the FP64 row walker's (propose_sm.hip).  DESIGN.md section 3k has the error budget behind m_r; TPH_OPT_MF_AUDIT
// re-evaluates every screened-out attempt in FP64 and counts contradictions (tests: zero over millions of attempts).
//
// What is cheap in low precision:
//   * the normals.  One Philox block gives a Box-Muller pair in every kernel of the library; here the pair is formed from
//     the SAME block with v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 (|z~ - z| <= 2^-13: measured over 2^32 blocks by
//     tph_bench_mf_normals, budget in DESIGN 3k) -- ~100 SIMD cycles per wave-call beside Philox's 240, where the FP64 pair
//     costs 848;
//   * the rows.  A wave holds 64 attempts as the columns of four 16 x 16 tiles and walks them in LOCKSTEP through panels of
//     16 rows: rows 16p..16p+15 of all its attempts are  L[16p:16p+16, 0:16p+16] . Z[0:16p+16, :]  = p + 1
//     v_mfma_f32_16x16x16_f16 per tile (8 cycles each), the panel of L as FP16 blocks in LDS (scaled by a power of two into
//     FP16's normal range, rounding error of every row summed exactly when the blocks are packed: that sum IS part of
//     m_r), the normals as FP16 in registers in the B-operand layout.  Dead columns are multiplied along (the
//     matrix cores are idle otherwise); the Box-Muller pairs are generated for LIVE columns only: the live (column, pair)
//     jobs of a panel are dealt to the 64 lanes, their results bounced through 2 KB of LDS into the owners' registers.
//   * lockstep means no per-lane stage, no staircase copy of L, no per-lane LDS reads of L (the FP64 walker's bound:
//     32 sixteen-byte LDS reads per 16 columns and wave whether one lane needs them or 64).
// A particle keeps G consecutive attempts in flight (a window: G columns); when a window's panels are through, its
// surviving columns are verified in FP64 in attempt order (wave-cooperative, lane = row, the transposed factor in LDS) and
// the first that is in bounds is the proposal; a window without one moves on by G attempts; after 256 attempts the current
// point is proposed (as everywhere in the library).  Particles come from a global queue in chunks of 4.
#include "common.h"
#include "tri.h"
#include "maha_tile.h"
#include <stdlib.h>
#include <stdio.h>

constexpr int MF_CAP = PROP_MAX_ATTEMPTS;      // attempts 0 .. MF_CAP-1, then the current point is proposed
constexpr int MF_CHUNK = 4;                    // particles per queue grab
constexpr int MF_DIRECT = 2;                   // list mode: attempts of a straggler evaluated in FP64 before any window is screened
constexpr int MF_WAVES = 8;                    // waves per workgroup (they share the LDS copy of the packed factor)
// the transposed FP64 factor of the verification, LT: rows 0..63 as T0[j][r] (j < min(d, 64), 64 rows per column, zeros above
// the diagonal), rows 64..d-1 as T1[j][r - 64] (j < d, d - 64 rows per column): lane = row reads column j without conflicts
__host__ __device__ static inline int mf_lt_doubles(int d) { return (d < 64 ? d : 64) * 64 + (d > 64 ? d * (d - 64) : 0); }
constexpr int MF_MAX_DIM = 112;                // 7 panels of 16 rows

typedef _Float16 mf_h4 __attribute__((ext_vector_type(4)));
typedef float mf_f4 __attribute__((ext_vector_type(4)));
typedef __fp16 mf_h2 __attribute__((ext_vector_type(2)));

__host__ __device__ static inline int mf_panels(int d) { return (d + 15) / 16; }
__host__ __device__ static inline int mf_nblk(int np) { return np * (np + 1) / 2; }
// one mode's screening pack: header (16 B: 1 / scale as float) | FP16 blocks [nblk][64 lanes][4] | EA[16 np] | EB[16 np] (float)
__host__ __device__ static inline size_t mf_pack_bytes(int np) { return 16 + (size_t)mf_nblk(np) * 512 + 2 * (size_t)(16 * np) * 4; }
// LDS of one wave: base records of its 64 / G particles | pair bounce | column table | column scale | column max |z| | FP64 normals | list
__host__ __device__ static inline size_t mf_wave_bytes(int npw, int dpad) {
  return (size_t)npw * dpad * 4 + 64 * 8 * 4 + 64 * 8 + 256 + 256 + (size_t)dpad * 8 + 64;
}
__host__ __device__ static inline size_t mf_tables_bytes(int np) { return (size_t)mf_nblk(np) * 512 + 2 * (size_t)(16 * np) * 4 + (size_t)(16 * np); }
// (rounded up to 16 B: for odd n_dim > 64 the factor is an odd number of doubles, and the per-wave regions behind it are read as
// 16-byte vectors)
__host__ __device__ static inline size_t mf_shared_bytes(int np, int d) {
  return (mf_tables_bytes(np) + sizeof(double) * (size_t)mf_lt_doubles(d) + 15) & ~(size_t)15;
}

__device__ __forceinline__ void mf_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The Box-Muller pair of tph_rng::normal2 (common.h) from the SAME Philox block, in FP32 with the hardware's
// transcendental instructions.  u1 = (k53 + 1) 2^-53 and the angle u2 are the FP64 kernel's up to the conversions:
//   * -2 ln u1 through v_log_f32 of k 2^-53 (k rounded to 24 bits: relative 2^-23, i.e. 2^-23 absolute in the logarithm);
//     for u1 within 2^-8 of 1, where the logarithm of a rounded argument has no relative accuracy left, from the exact
//     complement t = 1 - u1 = (2^53 - 1 - k53) 2^-53 and three terms of -ln(1 - t);
//   * the angle from the top 32 bits of u2 (2^-25 revolutions), v_sin_f32 / v_cos_f32 take revolutions.
// |z~ - z| against normal2 stays below 2^-13 (tph_bench_mf_normals measures it on the device).
__device__ __forceinline__ void mf_normal2(const tph_rng& g, uint32_t draw, float& z0, float& z1) {
  const tph_u4 r = tph_philox(g.item, draw, g.tick, g.tag, g.k0, g.k1);
  float s;
  if (r.x < 0xFF000000u) {
    const float k = fmaf((float)(r.x >> 5), 67108864.0f, (float)(r.y >> 6) + 1.0f);
    s = -1.3862943611198906f * __builtin_amdgcn_logf(k * 0x1.0p-53f);
  } else {
    const float tk = fmaf((float)(0x7FFFFFFu - (r.x >> 5)), 67108864.0f, (float)(0x3FFFFFFu - (r.y >> 6)));
    const float t = tk * 0x1.0p-53f;
    s = 2.0f * t * fmaf(t, fmaf(t, 0.33333334f, 0.5f), 1.0f);
  }
  const float rad = __builtin_amdgcn_sqrtf(s);
  const float rev = (float)r.z * 0x1.0p-32f;
  z0 = rad * __builtin_amdgcn_cosf(rev);
  z1 = rad * __builtin_amdgcn_sinf(rev);
}

// One workgroup per mode: the FP16 blocks of L (block (p, k), k <= p: rows 16p.., columns 16k.., lane l holds row l & 15,
// columns 4 (l >> 4) .. +3 -- the A operand of v_mfma_f32_16x16x16_f16), the per-row error tables and the transposed FP64
// factor for the verification.  S = 2^e puts the largest |L| into [2^13, 2^14): every entry within 2^-27 of it is a normal
// FP16 number; what is smaller is charged its full magnitude (whether the matrix core flushes a subnormal input or not).
//   acc' = sum_j L~'_rj z~_j  against  S (L z)_r:   |acc' - S (L z)_r| <= zmax EA_r + EB_r   with
//   EA_r = E_r + N_r (2^-10 + 2^-13),  EB_r = N_r 2^-12,   E_r = sum_j |L~'_rj - S L_rj|,  N_r = sum_j |L~'_rj|
// (2^-10: the normals' conversion to FP16, round to zero; 2^-13: the FP32 accumulation of <= 112 exact products, twice
// over; EB: the FP32 normals' own error 2^-13 and a flushed FP16 normal 2^-14, with slack; zmax >= max |z~| + 2^-10).
static __global__ void __launch_bounds__(256) k_mf_pack(const double* __restrict__ chol, int d, int np, unsigned char* __restrict__ pack_all,
                                                        double* __restrict__ LT_all) {
  __shared__ double red[256];
  __shared__ double s_scale;
  const int t = threadIdx.x;
  const double* L = chol + (size_t)blockIdx.x * d * d;
  unsigned char* pack = pack_all + (size_t)blockIdx.x * mf_pack_bytes(np);
  double* LT = LT_all + (size_t)blockIdx.x * mf_lt_doubles(d);
  double mx = 0.0;
  for (int e = t; e < d * d; e += 256) {
    const int r = e / d, j = e % d;
    if (j <= r) mx = fmax(mx, fabs(L[e]));
  }
  red[t] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] = fmax(red[t], red[t + o]);
    __syncthreads();
  }
  if (t == 0) {
    const double m = red[0];
    int e = 0;
    if (m > 0.0 && m < 1e300) e = 13 - ilogb(m);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    s_scale = ldexp(1.0, e);
    ((float*)pack)[0] = (float)ldexp(1.0, -e);
    ((float*)pack)[1] = 0.0f; ((float*)pack)[2] = 0.0f; ((float*)pack)[3] = 0.0f;
  }
  __syncthreads();
  const double S = s_scale;
  _Float16* L16 = (_Float16*)(pack + 16);
  const int nblk = mf_nblk(np), dpad = 16 * np;
  for (int e = t; e < nblk * 256; e += 256) {
    const int b = e >> 8, lane = (e >> 2) & 63, jj = e & 3;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= b) ++p;
    const int k = b - p * (p + 1) / 2;
    const int r = 16 * p + (lane & 15), c = 16 * k + 4 * (lane >> 4) + jj;
    const double v = (r < d && c <= r) ? L[(size_t)r * d + c] * S : 0.0;
    L16[e] = (_Float16)v;
  }
  float* EA = (float*)(pack + 16 + (size_t)nblk * 512);
  float* EB = EA + dpad;
  for (int r = t; r < dpad; r += 256) {
    double E = 0.0, N = 0.0;
    if (r < d)
      for (int c = 0; c <= r; ++c) {
        const double v = L[(size_t)r * d + c] * S;
        const double h = (double)(_Float16)v;
        double err = fabs(h - v);
        if (fabs(v) < 0x1.0p-14) err = fmax(err, fabs(v));
        E += err;
        N += fabs(h);
      }
    EA[r] = (float)((E + N * (0x1.0p-10 + 0x1.0p-13)) * (1.0 + 0x1.0p-20));
    EB[r] = (float)((N * 0x1.0p-12) * (1.0 + 0x1.0p-20));
  }
  const int j0n = d < 64 ? d : 64, r1n = d - 64;
  for (int e = t; e < j0n * 64; e += 256) {
    const int j = e >> 6, r = e & 63;
    LT[e] = (r < d && j <= r) ? L[(size_t)r * d + j] : 0.0;
  }
  if (d > 64)
    for (int e = t; e < d * r1n; e += 256) {
      const int j = e / r1n, r = 64 + e % r1n;
      LT[(size_t)j0n * 64 + e] = (j <= r) ? L[(size_t)r * d + j] : 0.0;
    }
}

// queue words (64-bit): [1] sum of attempts, [2] particles decided, [3] audit contradictions, [4] FP64 verifications,
// [5] attempts screened, [6] Box-Muller pair jobs; [16 + g] chunk cursor of workgroup g.
// Particles are dealt to the waves in chunks of 4 from ONE global cursor (queue[16]).  Where the launch's HBM reads come from
// (FETCH_SIZE 1.09 GB per launch at 131 072 x 100-D from the prior, against 0.105 GB of u; measured round 5,
// profiles/r05_mf_ab.json): the kernel reads u BY PARTICLE -- lane = row, 8 bytes from each of n_dim different lines, once
// when a particle takes a slot and once per FP64 verification -- so a (particle, row) read touches a whole 64-byte sector
// (131 072 x 100 x 64 B = 0.84 GB if nobody shares), and the 8 particles behind one sector are dealt to waves of different
// XCDs, each with its own L2.  Dealing from ranges that keep sector neighbours in one L2 (TPH_OPT_MF_DEAL = 2: eight ranges, one
// per XCD's workgroups) halves the reads (0.51 GB) -- and makes the launch no faster at 100-D and 14-22 % SLOWER at 32-D / 50-D;
// per-workgroup ranges (= 1): reads -19 %, 32-D 14 % faster, 50-D 8 % slower, 100-D +-0.  The loads overlap with the vector work
// that bounds the kernel; what the ranges cost is balance (the attempt counts of particles differ by orders of magnitude).
// The global cursor stays the default.
template <int KERNEL, bool HAS_BC, int NP>
__global__ void __launch_bounds__(64 * MF_WAVES) k_propose_mf(const double* __restrict__ u, int64_t n, int64_t ld, int d,
                                                             const double* __restrict__ means, const unsigned char* __restrict__ pack,
                                                             const double* __restrict__ LT, const double* __restrict__ sigmas,
                                                             const uint8_t* __restrict__ bc, uint64_t seed, tph_stepctl tick,
                                                             int64_t item0, double* __restrict__ up, const double* __restrict__ bfac,
                                                             int lgG, unsigned long long* __restrict__ queue, int audit,
                                                             const int32_t* __restrict__ todo_cnt, const int32_t* __restrict__ todo_rows,
                                                             int att0, const int32_t* __restrict__ att0_dev,
                                                             const int32_t* __restrict__ todo_off, int direct_tries, int local_deal) {
  if (att0_dev) att0 = *att0_dev;      // (list mode behind fanned-out rounds: the first untried attempt was decided on the device)
  if (todo_off) todo_rows += *todo_off; // (several modes: this mode's stretch of the list array; means / pack / LT / sigmas are the mode's)
  // todo_cnt != NULL: only the particles LISTED in todo_rows[0 .. *todo_cnt) (those the blocked kernel's rounds left out of
  // bounds), from attempt att0 on; workgroups beyond the list exit before they load anything
  extern __shared__ __attribute__((aligned(16))) unsigned char mf_lds[];
  int chunk = MF_CHUNK;
  if (todo_cnt) {
    // a short list is SPREAD: one straggler per wave while there are waves (its windows are a chain of latencies, and nothing
    // else would use the other waves), with 16 attempts in flight instead of 8
    n = *todo_cnt;
    const int64_t waves = (int64_t)gridDim.x * MF_WAVES;
    const int64_t ppw = (n + waves - 1) / waves;
    chunk = ppw < 1 ? 1 : (ppw > MF_CHUNK ? MF_CHUNK : (int)ppw);
    if ((int64_t)blockIdx.x * (MF_WAVES * chunk) >= n) return;
    const int lg = ppw <= 1 ? 4 : 3;
    lgG = lg > lgG ? lg : lgG;
  }
  constexpr int DPAD = 16 * NP, NBLK = NP * (NP + 1) / 2;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = 1 << lgG, NPW = 64 >> lgG;
  const int npairs = (d + 1) >> 1;
  // ---- workgroup-shared tables: the packed factor, the row error tables, the boundary flags
  const mf_h4* L16 = (const mf_h4*)mf_lds;
  const float* EA = (const float*)(mf_lds + (size_t)NBLK * 512);
  const float* EB = EA + DPAD;
  uint8_t* bcs = (uint8_t*)(mf_lds + (size_t)NBLK * 512 + 2 * (size_t)DPAD * 4);
  const double* LTs = (const double*)(mf_lds + mf_tables_bytes(NP));     // the FP64 factor, transposed (verification)
  {
    double* dstl = (double*)(mf_lds + mf_tables_bytes(NP));
    const int nl = mf_lt_doubles(d);
    // (eight values requested before the first is stored: one by one the copy is a chain of nl / 512 memory round trips in front
    // of every launch -- 15 at 100-D --, which a launch over a short straggler list consists of)
    for (int e0 = threadIdx.x; e0 < nl; e0 += 8 * (int)blockDim.x) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int e = e0 + k * (int)blockDim.x; v[k] = e < nl ? LT[e] : 0.0; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int e = e0 + k * (int)blockDim.x; if (e < nl) dstl[e] = v[k]; }
    }
  }
  {
    const uint32_t* src = (const uint32_t*)(pack + 16);
    uint32_t* dst = (uint32_t*)mf_lds;
    const int words = NBLK * 128 + 2 * DPAD;
    for (int e0 = threadIdx.x; e0 < words; e0 += 8 * (int)blockDim.x) {
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int e = e0 + k * (int)blockDim.x; v[k] = e < words ? src[e] : 0u; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int e = e0 + k * (int)blockDim.x; if (e < words) dst[e] = v[k]; }
    }
    for (int e = threadIdx.x; e < DPAD; e += blockDim.x) bcs[e] = (HAS_BC && e < d) ? bc[e] : (uint8_t)TPH_BC_STRICT;
  }
  const float inv_scale = ((const float*)pack)[0];
  // ---- this wave's LDS
  unsigned char* w = mf_lds + mf_shared_bytes(NP, d) + (size_t)wid * mf_wave_bytes(NPW, DPAD);
  float* brec = (float*)w;             w += (size_t)NPW * DPAD * 4;     // [NPW][DPAD] base coordinates (FP32) of the wave's particles
  uint32_t* zscr = (uint32_t*)w;       w += 64 * 8 * 4;                 // [column][pair of the panel] two FP16 normals
  int2* coltab = (int2*)w;             w += 64 * 8;                     // [column] (particle, attempt)
  float* colb = (float*)w;             w += 256;                        // [column] step scale b / S
  uint32_t* colzm = (uint32_t*)w;      w += 256;                        // [column] max |z~| so far (float bits)
  double* zv = (double*)w;             w += (size_t)DPAD * 8;           // FP64 normals of the attempt under verification
  uint8_t* list = (uint8_t*)w;                                         // live columns of the panel, packed
  for (int e = lane; e < 64 * 8; e += 64) zscr[e] = 0u;
  __syncthreads();

  const uint32_t tk = tick;
  const double sigma = sigmas[0];
  const double a_fac = (KERNEL == TPH_KERNEL_TPCN) ? tph_sqrt(1.0 - sigma * sigma) : 1.0;
  const int64_t nchunks = (n + chunk - 1) / chunk;
  const unsigned long long wmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  // particle slots: lane e < NPW holds slot e
  int ps_row = -1, ps_a0 = 0;
  bool ps_fresh = false;              // list mode: the particle has not had its direct tries yet (see (B))
  float ps_bs = 0.0f;
  int64_t pool_row = 0;
  int pool_next = 0, pool_cnt = 0;
  bool exhausted = false;
  unsigned long long n_att = 0, n_dec = 0, n_bad = 0, n_ver = 0, n_scr = 0, n_job = 0;
  // chunk dealing (lane 0; ~0ull: nothing left anywhere).  local_deal = 0 (default): ONE cursor for all workgroups.
  // 1: every workgroup owns a contiguous share of the chunks and deals it in order, then takes from the others' (stealing at
  // the tail); 2: eight shares, one per group of workgroups blockIdx.x mod 8 (the XCDs take workgroups round-robin).
  const int ranges = local_deal == 1 ? (int)gridDim.x : (local_deal == 2 ? ((int)gridDim.x < 8 ? (int)gridDim.x : 8) : 1);
  const int64_t per_group = (nchunks + ranges - 1) / ranges;
  int steal_k = 0;
  auto grab = [&]() -> unsigned long long {
    for (int k = steal_k; k < ranges; ++k) {
      const int g = (int)((blockIdx.x + (unsigned)k) % (unsigned)ranges);
      const int64_t lo = (int64_t)g * per_group;
      const int64_t cnt = (lo + per_group < nchunks ? lo + per_group : nchunks) - lo;
      if (cnt <= 0) continue;
      if (k > 0 && (int64_t)__hip_atomic_load(&queue[16 + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cnt) continue;
      const unsigned long long c = atomicAdd(&queue[16 + g], 1ull);
      if ((int64_t)c < cnt) { steal_k = k; return (unsigned long long)(lo + (int64_t)c); }
    }
    steal_k = ranges;
    return ~0ull;
  };
  unsigned long long ahead = 0;
  if (lane == 0) ahead = grab();
#ifdef MF_PROFILE
  long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_t = clock64();      // refill, setup, jobs, mfma + check, verify, cap, batches
#ifdef MF_PROFILE_VERIFY      // (the four counters are the parts of the FP64 evaluation instead: Box-Muller round, wait for u, rows, tail)
#define MF_PF(k) do { } while (0)
#define MF_PV(k) do { const long long now_ = clock64(); pf[k] += now_ - pf_t; pf_t = now_; } while (0)
#else
#define MF_PF(k) do { const long long now_ = clock64(); pf[k] += now_ - pf_t; pf_t = now_; } while (0)
#endif
#else
#define MF_PF(k) do { } while (0)
#endif
#ifndef MF_PV
#define MF_PV(k) do { } while (0)
#endif

  // FP64 evaluation of the attempt in column c, by the arithmetic of the other proposal kernels (lane = row; the normals of
  // the whole attempt in one Box-Muller round, lane = pair).  In bounds -> (commit) the rows are the proposal.
  auto verify = [&](int c, bool commit) -> bool {
    const int2 ra = coltab[c];
    const int64_t R = ra.x;
    // the current point and the step scale are requested first: their round trips run beside the Box-Muller round below
    const int rr0 = lane < d ? lane : d - 1, rr1 = lane + 64 < d ? lane + 64 : d - 1;
    const double uj0 = u[(size_t)rr0 * ld + R], uj1 = d > 64 ? u[(size_t)rr1 * ld + R] : 0.0;
    const double mu0 = (KERNEL == TPH_KERNEL_TPCN) ? means[rr0] : 0.0, mu1 = (KERNEL == TPH_KERNEL_TPCN) ? means[rr1] : 0.0;
    const double b = (KERNEL == TPH_KERNEL_TPCN) ? bfac[R] : sigma;
#ifdef MF_PROFILE_VERIFY
    pf_t = clock64();
#endif
    tph_rng gz(seed, tk, TPH_TAG_NORMAL, (uint64_t)(item0 + R));
    for (int q = lane; q < npairs; q += 64) {
      double z0, z1;
      gz.normal2((uint32_t)ra.y * (uint32_t)npairs + (uint32_t)q, z0, z1);
      zv[2 * q] = z0;
      zv[2 * q + 1] = z1;
    }
    mf_wave_sync();
    MF_PV(0);
#ifdef MF_PROFILE_VERIFY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MF_PV(1);
#endif
    double acc0 = 0.0, acc1 = 0.0;
    const int jm = d < 64 ? d : 64;
    if (d > 64) {
      // rows lane and lane + 64 side by side: two independent chains of fused multiply-adds, each in column order as everywhere
      // else (lanes without a second row walk the last one's and drop the sum)
      const int r1n = d - 64;
      const double* l0 = LTs + lane;
      const double* l1 = LTs + jm * 64 + (lane < r1n ? lane : r1n - 1);
#pragma unroll 8
      for (int j = 0; j < 64; ++j) {
        const double z = zv[j];
        acc0 = fma(l0[j * 64], z, acc0);
        acc1 = fma(l1[j * r1n], z, acc1);
      }
#pragma unroll 4
      for (int j = 64; j < d; ++j) acc1 = fma(l1[j * r1n], zv[j], acc1);
    } else {
      const double* l0 = LTs + lane;
#pragma unroll 8
      for (int j = 0; j < jm; ++j) acc0 = fma(l0[j * 64], zv[j], acc0);
    }
    MF_PV(2);
    bool ok = true;
    double x0 = 0.0, x1 = 0.0;
    if (lane < d) {
      const double base = (KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, uj0 - mu0, mu0) : uj0;
      x0 = fma(b, acc0, base);
      const uint8_t f = HAS_BC ? bcs[lane] : (uint8_t)TPH_BC_STRICT;
      if (f == TPH_BC_PERIODIC) x0 = bc_periodic(x0);
      else if (f == TPH_BC_REFLECTIVE) x0 = bc_reflective(x0);
      else ok = (x0 >= 0.0) && (x0 <= 1.0);
    }
    if (lane + 64 < d) {
      const int r = lane + 64;
      const double base = (KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, uj1 - mu1, mu1) : uj1;
      x1 = fma(b, acc1, base);
      const uint8_t f = HAS_BC ? bcs[r] : (uint8_t)TPH_BC_STRICT;
      if (f == TPH_BC_PERIODIC) x1 = bc_periodic(x1);
      else if (f == TPH_BC_REFLECTIVE) x1 = bc_reflective(x1);
      else ok = ok && (x1 >= 0.0) && (x1 <= 1.0);
    }
    const bool all = __ballot(!ok) == 0ull;
    if (all && commit) {
      if (lane < d) up[(size_t)lane * ld + R] = x0;
      if (lane + 64 < d) up[(size_t)(lane + 64) * ld + R] = x1;
    }
    mf_wave_sync();
    MF_PV(3);
    return all;
  };

#pragma unroll 1
  for (;;) {
    // ---- (A) empty slots take the next particles of the queue; their base records mu + a (u - mu) (tpCN) / u (RWM) go to LDS
    {
      unsigned long long need = __ballot(lane < NPW && ps_row < 0), fresh = 0ull;
#pragma unroll 1
      while (need) {
        if (pool_next >= pool_cnt) {
          if (exhausted) break;
          const unsigned long long c = __shfl(ahead, 0, 64);
          if (c == ~0ull) { exhausted = true; break; }
          if (lane == 0) ahead = grab();
          pool_row = (int64_t)c * chunk;
          pool_cnt = (int)((n - pool_row) < chunk ? (n - pool_row) : chunk);
          pool_next = 0;
        }
        const int e = __ffsll((long long)need) - 1;
        need &= need - 1ull;
        if (lane == e) { ps_row = todo_rows ? todo_rows[pool_row + pool_next] : (int)(pool_row + pool_next); ps_a0 = att0; ps_fresh = todo_cnt != nullptr && direct_tries > 0; }
        ++pool_next;
        fresh |= 1ull << e;
      }
      if ((fresh >> lane) & 1ull) ps_bs = (float)((KERNEL == TPH_KERNEL_TPCN) ? bfac[ps_row] : sigma) * inv_scale;
      // the records of the fresh slots, four slots' loads in flight at a time (a window of a few attempts per particle
      // refills every slot in every pass: one slot after the other, each behind its own memory round trip, was half of such a step)
      const int r1 = lane + 64 < d ? lane + 64 : d - 1, r0 = lane < d ? lane : d - 1;
      const double m0 = (KERNEL == TPH_KERNEL_TPCN) ? means[r0] : 0.0, m1 = (KERNEL == TPH_KERNEL_TPCN) ? means[r1] : 0.0;
#pragma unroll 1
      while (fresh) {
        int es[4];
        double v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          es[q] = fresh ? __ffsll((long long)fresh) - 1 : -1;
          fresh &= fresh - (fresh ? 1ull : 0ull);
          const int64_t R = es[q] >= 0 ? (int64_t)__shfl(ps_row, es[q], 64) : 0;
          v0[q] = es[q] >= 0 ? u[(size_t)r0 * ld + R] : 0.0;
          v1[q] = (es[q] >= 0 && DPAD > 64) ? u[(size_t)r1 * ld + R] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (es[q] >= 0) {
            float* rec = brec + (size_t)es[q] * DPAD;
            const double b0 = (KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, v0[q] - m0, m0) : v0[q];
            if (lane < DPAD) rec[lane] = lane < d ? (float)b0 : 0.5f;      // padding rows: L row zero, base 0.5 -> always in bounds
            if (DPAD > 64 && lane + 64 < DPAD) {
              const double b1 = (KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, v1[q] - m1, m1) : v1[q];
              rec[lane + 64] = lane + 64 < d ? (float)b1 : 0.5f;
            }
          }
      }
      if (__ballot(lane < NPW && ps_row >= 0) == 0ull) break;
    }
    MF_PF(0);
    // ---- (B) the windows: column c = lane is attempt a0 + (c mod G) of slot c / G
    const int my_ps = lane >> lgG;
    const int my_row = __shfl(ps_row, my_ps, 64);
    const int my_att = __shfl(ps_a0, my_ps, 64) + (lane & (G - 1));
    coltab[lane] = make_int2(my_row, my_att);
    colb[lane] = __shfl(ps_bs, my_ps, 64);
    colzm[lane] = 0u;
    // List mode, a straggler's first pass: its next MF_DIRECT attempts go STRAIGHT to the FP64 evaluation, no screen.  Late in a
    // run the blocked kernel's failures mostly succeed at their next attempt, and a screened window (every panel of every live
    // column, a chain of LDS round trips per panel) costs several times the one evaluation that settles them.
    const bool direct = todo_cnt != nullptr && __ballot(lane < NPW && ps_row >= 0 && ps_fresh) != 0ull;
    const bool my_fresh = __shfl((int)ps_fresh, my_ps, 64) != 0;
    unsigned long long alive = __ballot(my_row >= 0 && my_att < MF_CAP && (!direct || (my_fresh && (lane & (G - 1)) < MF_DIRECT)));
    if (!direct) n_scr += (unsigned long long)__popcll(alive);
    mf_wave_sync();
    float bs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bs[t] = colb[16 * t + (lane & 15)];
    MF_PF(1);
    // ---- (C) panels of 16 rows, all columns in lockstep
    unsigned long long audited = 0ull;
    mf_h4 Z[4][NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (alive != 0ull && !direct) {
        // live (column, pair) jobs -> FP16 normals in the bounce buffer
        const int A = __popcll(alive);
        if ((alive >> lane) & 1ull) list[__popcll(alive & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        mf_wave_sync();
        const int nq = (npairs - 8 * p) < 8 ? (npairs - 8 * p) : 8;
#pragma unroll 1
        for (int j0 = 0; j0 < 8 * A; j0 += 64) {
          const int j = j0 + lane, i = j >> 3, q = j & 7;
          if (i < A && q < nq) {
            const int c = list[i];
            const int2 ra = coltab[c];
            tph_rng gz(seed, tk, TPH_TAG_NORMAL, (uint64_t)(item0 + (int64_t)ra.x));
            float z0, z1;
            mf_normal2(gz, (uint32_t)ra.y * (uint32_t)npairs + (uint32_t)(8 * p + q), z0, z1);
            const mf_h2 h = __builtin_amdgcn_cvt_pkrtz(z0, z1);
            zscr[c * 8 + q] = __builtin_bit_cast(uint32_t, h);
            // (an LDS atomic without a return value: its 8 jobs per column serialise in the LDS unit -- 66 M bank-conflict cycles per
            // launch at 131 072 x 100-D -- but BESIDE the vector work that bounds the kernel.  Replaced by a DPP butterfly over the 8
            // lanes and one plain update, the conflicts fell by 95 % and the launch got 4-7 % SLOWER: nine more VALU instructions
            // per job batch on the critical path.  Measured round 5, profiles/r05_mf_ab.json; the atomic stays.)
            atomicMax(&colzm[c], __float_as_uint(fmaxf(fabsf(z0), fabsf(z1))));
          }
        }
        n_job += (unsigned long long)(A * nq);
        mf_wave_sync();
        MF_PF(2);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if ((alive >> (16 * t)) & 0xFFFFull) Z[t][p] = *(const mf_h4*)&zscr[(16 * t + (lane & 15)) * 8 + 2 * (lane >> 4)];
        // rows 16p .. 16p+15 of every column
        mf_f4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mf_f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k <= p; ++k) {
          const mf_h4 a = L16[(size_t)(p * (p + 1) / 2 + k) * 64 + lane];
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if ((alive >> (16 * t)) & 0xFFFFull) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, Z[t][k], acc[t], 0, 0, 0);
        }
        // bounds with the error margin: lane (kg, n) holds rows 16p + 4 kg .. +3 of column 16 t + n
        unsigned long long dead = 0ull;
        const int r0 = 16 * p + 4 * (lane >> 4);
        const mf_f4 ea = *(const mf_f4*)&EA[r0], eb = *(const mf_f4*)&EB[r0];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if ((alive >> (16 * t)) & 0xFFFFull) {
            const int c = 16 * t + (lane & 15);
            const float zm = __uint_as_float(colzm[c]) + 0x1.0p-10f;
            const mf_f4 bb = *(const mf_f4*)&brec[(size_t)(c >> lgG) * DPAD + r0];
            const float b = bs[t], ab = fabsf(b);
            bool kill = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              // |x~ - x| <= m0 + 2^-22 (1 + |x~|) (base, product and sum rounded in FP32); "x~ < -m or x~ > 1 + m" with that m
              // follows from |x~ - 1/2| > 1/2 + m1, m1 = (m0 + 2^-22)(1 + 2^-21) + 2^-21.  Non-finite values never compare true.
              const float x = fmaf(b, acc[t][i], bb[i]);
              const float m0 = ab * fmaf(zm, ea[i], eb[i]);
              const float m1 = fmaf(m0, 1.0f + 0x1.0p-21f, 0x1.0p-20f);
              bool k1 = fabsf(x - 0.5f) > 0.5f + m1;
              if (HAS_BC) k1 = k1 && bcs[r0 + i] == (uint8_t)TPH_BC_STRICT;
              kill = kill || k1;
            }
            const unsigned long long bal = __ballot(kill);
            dead |= ((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull) << (16 * t);
          }
        dead &= alive;
        audited |= audit ? dead : 0ull;      // every attempt the screen removes must fail in FP64 too: checked with the candidates below
        alive &= ~dead;
        MF_PF(3);
      }
    }
    // ---- (D) the columns that passed every panel, in attempt order: the first that is in bounds in FP64 is the proposal.
    // (TPH_OPT_MF_AUDIT: the columns the screen removed are evaluated too, nothing is committed for them, and one that turns
    // out to be in bounds is counted as a contradiction.)
    {
      unsigned long long cand = alive, todo = alive | audited;
#pragma unroll 1
      while (todo) {
        const int c = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const bool is_cand = (cand >> c) & 1ull;
        if (!is_cand && !((audited >> c) & 1ull)) continue;      // a candidate behind its window's winner
        const bool ok = verify(c, is_cand);
        if (is_cand) {
          n_ver += 1ull;
          if (ok) {
            const int e = c >> lgG;
            cand &= ~(wmask << (e << lgG));
            todo &= ~(wmask << (e << lgG)) | audited;
            n_att += (unsigned long long)(coltab[c].y + 1);
            n_dec += 1ull;
            if (lane == e) ps_row = -1;
          }
        } else if (ok) {
          n_bad += 1ull;
        }
      }
    }
    MF_PF(4);
    if (lane < NPW && ps_row >= 0) {
      if (!direct) ps_a0 += G;
      else if (ps_fresh) { ps_a0 += MF_DIRECT; ps_fresh = false; }
    }
    {
      unsigned long long capped = __ballot(lane < NPW && ps_row >= 0 && ps_a0 >= MF_CAP);
#pragma unroll 1
      while (capped) {                  // redraw cap reached (the reference would loop on): the current point is proposed
        const int e = __ffsll((long long)capped) - 1;
        capped &= capped - 1ull;
        const int64_t R = __shfl(ps_row, e, 64);
        for (int r = lane; r < d; r += 64) {
          const double uj = u[(size_t)r * ld + R];
          up[(size_t)r * ld + R] = (KERNEL == TPH_KERNEL_TPCN) ? (uj - means[r]) + means[r] : uj;
        }
        n_att += (unsigned long long)MF_CAP;
        n_dec += 1ull;
        if (lane == e) ps_row = -1;
      }
    }
    MF_PF(5);
#ifdef MF_PROFILE
    pf[6] += 1;
#endif
  }
#ifdef MF_PROFILE
  if (lane == 0)
    for (int k = 0; k < 7; ++k) atomicAdd(&queue[8 + k], (unsigned long long)pf[k]);
#endif
  if (lane == 0) {
    atomicAdd(&queue[1], n_att);
    atomicAdd(&queue[2], n_dec);
    if (n_bad) atomicAdd(&queue[3], n_bad);
    atomicAdd(&queue[4], n_ver);
    atomicAdd(&queue[5], n_scr);
    atomicAdd(&queue[6], n_job);
  }
}

// persistent buffers of the screened kernel (queue words | blocked L^-1 (tri.h, tpCN forms) | screening pack | transposed FP64
// factor), rebuilt when the caller's mode statistics change
// (K modes: K blocked inverse factors, K screening packs, K transposed factors, each behind the other)
struct mf_bufs { unsigned long long* queue; double* Wb; unsigned char* pack; double* LT; };
static int mf_alloc(tph_ctx* ctx, size_t* off_wb_, size_t* off_pack_, size_t* off_lt_, int K = 1) {
  const int d = ctx->d, np = mf_panels(d);
  const size_t tb8 = tri_blocked_doubles(d);
  const size_t off_wb = 4096, off_pack = off_wb + sizeof(double) * tb8 * (size_t)K;      // (the queue block: TPH_MF_QWORDS words)
  const size_t off_lt = (off_pack + mf_pack_bytes(np) * (size_t)K + 255) & ~(size_t)255;
  const size_t need = off_lt + sizeof(double) * (size_t)mf_lt_doubles(d) * (size_t)K;
  if (ctx->mf_bytes < need) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->mf_buf) ctx->retired.push_back(ctx->mf_buf);
    ctx->mf_buf = nullptr; ctx->mf_bytes = 0; ctx->mf_epoch = -1;
    TPH_HIP(hipMalloc((void**)&ctx->mf_buf, need));
    ctx->mf_bytes = need;
  }
  if (off_wb_) { *off_wb_ = off_wb; *off_pack_ = off_pack; *off_lt_ = off_lt; }
  return 0;
}
unsigned int* tph_mf_queue_words(tph_ctx* ctx) {
  if (ctx->d <= 16 || ctx->d > MF_MAX_DIM || mf_alloc(ctx, nullptr, nullptr, nullptr)) return nullptr;
  return (unsigned int*)ctx->mf_buf;
}
template <int KERNEL>
static int mf_prepare(tph_ctx* ctx, const double* chol, const double* winv, mf_bufs* b, bool zero = true, int K = 1) {
  const int d = ctx->d, np = mf_panels(d);
  size_t off_wb, off_pack, off_lt;
  if (mf_alloc(ctx, &off_wb, &off_pack, &off_lt, K)) return -1;
  b->queue = (unsigned long long*)ctx->mf_buf;
  b->Wb = (double*)((char*)ctx->mf_buf + off_wb);
  b->pack = (unsigned char*)ctx->mf_buf + off_pack;
  b->LT = (double*)((char*)ctx->mf_buf + off_lt);
  // a launch being captured into a hipGraph records the rebuild (a replayed step never re-enters this host code)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  TPH_HIP(hipStreamIsCapturing(ctx->stream, &cap));
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (capturing || ctx->modes_epoch <= 0 || ctx->mf_epoch != ctx->modes_epoch || ctx->mf_src != (const void*)chol ||
      ctx->mf_kernel != KERNEL || ctx->mf_K != K) {
    hipLaunchKernelGGL(k_mf_pack, dim3(K), dim3(256), 0, ctx->stream, chol, d, np, b->pack, b->LT);
    if (KERNEL == TPH_KERNEL_TPCN) hipLaunchKernelGGL(k_tri_block, dim3(K), dim3(256), 0, ctx->stream, winv, d, b->Wb);
    ctx->mf_epoch = capturing ? -1 : ctx->modes_epoch; ctx->mf_src = (const void*)chol; ctx->mf_kernel = KERNEL; ctx->mf_K = K;
  }
  if (zero) hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, ctx->stream, (unsigned int*)b->queue, TPH_MF_QWORDS);
  return 0;
}

// the screened kernel over n particles (todo_cnt == NULL) or over a device-side list of at most n (from attempt att0)
template <int KERNEL>
static int mf_launch(tph_ctx* ctx, const mf_bufs& b, const double* u, int64_t n, int64_t ld, const double* means, const double* sigmas,
                     const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0, double* up, const double* bfac,
                     const int32_t* todo_cnt, const int32_t* todo_rows, int att0, const int32_t* att0_dev = nullptr,
                     const int32_t* todo_off = nullptr, int direct_tries = MF_DIRECT) {
  const int d = ctx->d, np = mf_panels(d), dpad = 16 * np;
  const int64_t nchunks = (n + MF_CHUNK - 1) / MF_CHUNK;
  const int cus = ctx->n_simd / 4;
  // workgroups: one per CU (8 waves of up to 256 VGPRs -- the FP16 normals of all panels stay in registers -- around one LDS
  // copy of the packed factor and of the FP64 factor; compiled for four waves per SIMD the normals of the earlier panels spill
  // to scratch: 3.5 against 3.3 ms at 131 072 x 100-D from the prior), never more waves than there are chunks
  int64_t groups = (nchunks + MF_WAVES - 1) / MF_WAVES;
  if (groups > cus) groups = cus;
  // attempts of a particle in flight (log2; TPH_OPT_MF_LANES, 0 = by size as in the row walker: 8 when a wave gets >= 48 particles)
  int lgG = ctx->mf_lanes;
  if (lgG <= 0) lgG = (double)n / (double)(groups * MF_WAVES) >= 48.0 ? 3 : 4;
  if (todo_cnt && ctx->mf_lanes <= 0) lgG = 3;
  if (lgG < 3) lgG = 3;
  if (lgG > 6) lgG = 6;
  const int npw = 64 >> lgG;
  const size_t lds = mf_shared_bytes(np, d) + (size_t)MF_WAVES * mf_wave_bytes(npw, dpad);
  TPH_REQUIRE(lds <= 160 * 1024, "tph_propose (screened batches): n_dim=%d needs %zu B of LDS", d, lds);
  const int audit = ctx->mf_audit;
#define TPH_MF(BC, NPV)                                                                                                  \
  do {                                                                                                                   \
    if (lds > 64 * 1024)                                                                                                 \
      TPH_HIP(hipFuncSetAttribute((const void*)k_propose_mf<KERNEL, BC, NPV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_propose_mf<KERNEL, BC, NPV>), dim3((unsigned)groups), dim3(64 * MF_WAVES), lds, ctx->stream,    \
                       u, n, ld, d, means, (const unsigned char*)b.pack, (const double*)b.LT, sigmas, bc, seed,           \
                       tick, item0, up, bfac, lgG, b.queue, audit, todo_cnt, todo_rows, att0, att0_dev, todo_off, direct_tries, ctx->mf_deal);   \
  } while (0)
#define TPH_MF_NP(NPV) do { if (bc) TPH_MF(true, NPV); else TPH_MF(false, NPV); } while (0)
  switch (np) {
    case 2: TPH_MF_NP(2); break;
    case 3: TPH_MF_NP(3); break;
    case 4: TPH_MF_NP(4); break;
    case 5: TPH_MF_NP(5); break;
    case 6: TPH_MF_NP(6); break;
    default: TPH_MF_NP(7); break;
  }
#undef TPH_MF_NP
#undef TPH_MF
  TPH_LAUNCH_CHECK();
  return 0;
}

template <int KERNEL>
static int propose_mf(tph_ctx* ctx, double* u, int64_t n, int64_t ld, const double* means, const double* chol, const double* winv,
                      const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                      double* up, double* maha_u, double* maha_up, uint8_t* pend) {
  const int d = ctx->d;
  TPH_REQUIRE(d > 16 && d <= MF_MAX_DIM, "tph_propose (screened batches): n_dim=%d outside 17..%d", d, MF_MAX_DIM);
  TPH_REQUIRE(n < (1ll << 31), "tph_propose (screened batches): %lld particles on one device", (long long)n);
  mf_bufs b;
  if (mf_prepare<KERNEL>(ctx, chol, winv, &b)) return -1;
  // pending moves; tpCN: the form at u (first step of a run) and every particle's step scale, parked in maha_up until the
  // closing pass overwrites it with the form at u'
  if (pend || KERNEL == TPH_KERNEL_TPCN || maha_u)
    if (launch_maha_tile<KERNEL, 0>(ctx, u, n, ld, means, b.Wb, up, maha_u, tick, pend, nullptr, dof, sigmas, seed, item0, maha_up)) return -1;
  if (mf_launch<KERNEL>(ctx, b, u, n, ld, means, sigmas, bc, seed, tick, item0, up, maha_up, nullptr, nullptr, 0)) return -1;
  if (KERNEL == TPH_KERNEL_TPCN && ctx->forms_mfma && d <= 112)
    return tph_blkm_forms(ctx, up, n, ld, means, chol, winv, maha_up, tick, b.queue, nullptr, nullptr, nullptr);
  if (KERNEL == TPH_KERNEL_TPCN || maha_up || tick.ctl)
    if (launch_maha_tile<KERNEL, 1>(ctx, u, n, ld, means, b.Wb, up, maha_up, tick, nullptr, b.queue, dof, sigmas, seed, item0, nullptr)) return -1;
  return 0;
}

// The straggler pass behind the blocked kernel's rounds (mutate.hip: launch_propose_blk): the particles it LISTED continue from
// attempt att0 here -- screened windows, first in-bounds attempt in attempt order -- and, for tpCN, get the form at u'.  The
// chores of the step (pending moves, form at u, step scale parked in maha_up) were done by the blocked kernel's first round.
template <int KERNEL>
static int propose_mf_list(tph_ctx* ctx, double* u, int64_t n, int64_t ld, const double* means, const double* chol, const double* winv,
                           const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                           double* up, double* maha_up, const int32_t* todo_cnt, const int32_t* todo_rows, int att0,
                           const int32_t* att0_dev, bool queue_zeroed) {
  mf_bufs b;
  if (mf_prepare<KERNEL>(ctx, chol, winv, &b, !queue_zeroed)) return -1;
  if (mf_launch<KERNEL>(ctx, b, u, n, ld, means, sigmas, bc, seed, tick, item0, up, maha_up, todo_cnt, todo_rows, att0, att0_dev)) return -1;
  if (KERNEL == TPH_KERNEL_TPCN && ctx->forms_mfma && ctx->d <= 112)
    return tph_blkm_forms(ctx, up, n, ld, means, chol, winv, maha_up, tick, nullptr, todo_cnt, todo_rows, nullptr);
  if (KERNEL == TPH_KERNEL_TPCN || maha_up)
    if (launch_maha_tile<KERNEL, 1>(ctx, u, n, ld, means, b.Wb, up, maha_up, tick, nullptr, nullptr, dof, sigmas, seed, item0, nullptr,
                                    todo_cnt, todo_rows)) return -1;
  return 0;
}

// ---- several proposal modes (tempest/mcmc.py:225-249 applies each walker's own cluster mean / factor in its redraw loop) --------
// The screening pack, the transposed factor and the blocked inverse factor of ONE mode fill a workgroup's LDS, so the modes are
// served one after the other, each by launches over the list of ITS particles: the stable partition `order` of the matrix-core
// rounds (propose_blkm.hip: tph_mode_lists -- a mode's particles in index order, mstart[m] entries into order[], mcount[m] of
// them), or a mode's stretch of their failure lists.  A launch reads its list's offset and length on the device (todo_off,
// todo_cnt), so the host needs neither.  Proposals are the single-mode kernel's, attempt for attempt: a particle's arithmetic
// sees its own mode's matrices only.
template <int KERNEL>
static int propose_mf_modes(tph_ctx* ctx, double* u, const int32_t* assign, int64_t n, int64_t ld, int K, const double* means,
                            const double* chol, const double* winv, const double* dof, const double* sigmas, const uint8_t* bc,
                            uint64_t seed, tph_stepctl tick, int64_t item0, double* up, double* maha_u, double* maha_up, uint8_t* pend) {
  const int d = ctx->d, np = mf_panels(d);
  TPH_REQUIRE(d > 16 && d <= MF_MAX_DIM && K >= 1 && K <= 64, "tph_propose (screened batches, several modes): n_dim=%d / K=%d out of range", d, K);
  TPH_REQUIRE(n < (1ll << 31), "tph_propose (screened batches): %lld particles on one device", (long long)n);
  const int32_t *order = nullptr, *mstart = nullptr, *mcount = nullptr;
  if (tph_mode_lists(ctx, assign, n, K, &order, &mstart, &mcount)) return -1;
  mf_bufs b;
  if (mf_prepare<KERNEL>(ctx, chol, winv, &b, true, K)) return -1;
  const size_t tb8 = tri_blocked_doubles(d);
  const bool chores = pend || KERNEL == TPH_KERNEL_TPCN || maha_u;
  const bool closing = KERNEL == TPH_KERNEL_TPCN || maha_up || tick.ctl;
  for (int m = 0; m < K; ++m) {
    mf_bufs bm = b;
    bm.Wb = b.Wb + (size_t)m * tb8;
    bm.pack = b.pack + (size_t)m * mf_pack_bytes(np);
    bm.LT = b.LT + (size_t)m * mf_lt_doubles(d);
    const double* mu = means ? means + (size_t)m * d : nullptr;
    const double* dm = dof ? dof + m : nullptr;
    if (m) hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, ctx->stream, (unsigned int*)b.queue + TPH_MF_QCURSORS, TPH_MF_QWORDS - TPH_MF_QCURSORS);     // the chunk cursors; the counters run on
    if (chores)
      if (launch_maha_tile<KERNEL, 0>(ctx, u, n, ld, mu, bm.Wb, up, maha_u, tick, pend, nullptr, dm, sigmas + m, seed, item0, maha_up,
                                      mcount + m, order, mstart + m)) return -1;
    if (mf_launch<KERNEL>(ctx, bm, u, n, ld, mu, sigmas + m, bc, seed, tick, item0, up, maha_up, mcount + m, order, 0, nullptr, mstart + m, 0)) return -1;
    if (closing)
      if (launch_maha_tile<KERNEL, 1>(ctx, u, n, ld, mu, bm.Wb, up, maha_up, tick, nullptr, m == K - 1 ? b.queue : nullptr, dm, sigmas + m, seed,
                                      item0, nullptr, mcount + m, order, mstart + m)) return -1;
  }
  return 0;
}
int tph_propose_mf_modes(tph_ctx* ctx, int kernel, double* u, const int32_t* assign, int64_t n, int64_t ld, int K, const double* means,
                         const double* chol, const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed,
                         uint32_t tick0, const double* ctl, int64_t item0, double* up, double* maha_u, double* maha_up, uint8_t* pend) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return propose_mf_modes<TPH_KERNEL_TPCN>(ctx, u, assign, n, ld, K, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_u, maha_up, pend);
  return propose_mf_modes<TPH_KERNEL_RWM>(ctx, u, assign, n, ld, K, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_u, maha_up, pend);
}

// the straggler pass behind the matrix-core rounds over SEVERAL modes: mode m's failures are entries [offs[m], offs[m] + cnts[m])
// of `rows`, their first untried attempt is atts[m] (all device-side); screened windows per mode, then (tpCN) the form at u'
template <int KERNEL>
static int propose_mf_mode_lists(tph_ctx* ctx, double* u, int64_t n, int64_t ld, int K, const double* means, const double* chol,
                                 const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed,
                                 tph_stepctl tick, int64_t item0, double* up, double* maha_up, const int32_t* cnts, const int32_t* rows,
                                 const int32_t* offs, int att0, const int32_t* atts) {
  const int d = ctx->d, np = mf_panels(d);
  mf_bufs b;
  if (mf_prepare<KERNEL>(ctx, chol, winv, &b, true, K)) return -1;
  const size_t tb8 = tri_blocked_doubles(d);
  for (int m = 0; m < K; ++m) {
    mf_bufs bm = b;
    bm.Wb = b.Wb + (size_t)m * tb8;
    bm.pack = b.pack + (size_t)m * mf_pack_bytes(np);
    bm.LT = b.LT + (size_t)m * mf_lt_doubles(d);
    const double* mu = means ? means + (size_t)m * d : nullptr;
    if (m) hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, ctx->stream, (unsigned int*)b.queue + TPH_MF_QCURSORS, TPH_MF_QWORDS - TPH_MF_QCURSORS);
    if (mf_launch<KERNEL>(ctx, bm, u, n, ld, mu, sigmas + m, bc, seed, tick, item0, up, maha_up, cnts + m, rows, att0, atts ? atts + m : nullptr,
                          offs + m, MF_DIRECT)) return -1;
    if (KERNEL == TPH_KERNEL_TPCN || maha_up)
      if (launch_maha_tile<KERNEL, 1>(ctx, u, n, ld, mu, bm.Wb, up, maha_up, tick, nullptr, nullptr, dof ? dof + m : nullptr, sigmas + m, seed,
                                      item0, nullptr, cnts + m, rows, offs + m)) return -1;
  }
  return 0;
}
int tph_propose_mf_mode_lists(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, int K, const double* means, const double* chol,
                              const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                              const double* ctl, int64_t item0, double* up, double* maha_up, const int32_t* cnts, const int32_t* rows,
                              const int32_t* offs, int att0, const int32_t* atts) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return propose_mf_mode_lists<TPH_KERNEL_TPCN>(ctx, u, n, ld, K, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_up, cnts, rows, offs, att0, atts);
  return propose_mf_mode_lists<TPH_KERNEL_RWM>(ctx, u, n, ld, K, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_up, cnts, rows, offs, att0, atts);
}

int tph_propose_mf_list(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                        const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                        const double* ctl, int64_t item0, double* up, double* maha_up, const int32_t* todo_cnt, const int32_t* todo_rows,
                        int att0, const int32_t* att0_dev, int queue_zeroed) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return propose_mf_list<TPH_KERNEL_TPCN>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_up, todo_cnt, todo_rows, att0, att0_dev, queue_zeroed != 0);
  return propose_mf_list<TPH_KERNEL_RWM>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_up, todo_cnt, todo_rows, att0, att0_dev, queue_zeroed != 0);
}

int tph_propose_mf(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                   const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                   const double* ctl, int64_t item0, double* up, double* maha_u, double* maha_up, uint8_t* pend) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return propose_mf<TPH_KERNEL_TPCN>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_u, maha_up, pend);
  return propose_mf<TPH_KERNEL_RWM>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_u, maha_up, pend);
}

// ---- counters of the last screened launch (tests, profiles): out[0..6] = the queue words above
extern "C" int tph_bench_mf_counters(tph_ctx* ctx, unsigned long long* out7_host) {
  TPH_REQUIRE(ctx && out7_host, "tph_bench_mf_counters: NULL argument");
  TPH_REQUIRE(ctx->mf_buf, "tph_bench_mf_counters: no screened launch yet");
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  TPH_HIP(hipMemcpy(out7_host, ctx->mf_buf, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#ifdef MF_PROFILE
  {
    unsigned long long pf[8];
    TPH_HIP(hipMemcpy(pf, (char*)ctx->mf_buf + 64, sizeof(pf), hipMemcpyDeviceToHost));
    fprintf(stderr, "MF_PROFILE wave-cycles: refill %llu setup %llu jobs %llu mfma+check %llu verify %llu cap %llu | batches %llu\n", pf[0], pf[1],
            pf[2], pf[3], pf[4], pf[5], pf[6]);
  }
#endif
  return 0;
}

// ---- the FP32 Box-Muller pair against the FP64 one on the same Philox blocks: max |z~ - z| over n_blocks blocks ----
static __global__ void __launch_bounds__(256) k_mf_normals_check(uint64_t seed, uint64_t first, uint64_t n, int edge,
                                                                 double* __restrict__ out /* [blocks][3] */) {
  __shared__ double sh[4];
  double mx = 0.0, mz = 0.0, cnt = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t id = first + i;
    tph_rng g(seed, (uint32_t)(id >> 52), TPH_TAG_NORMAL, (id >> 20) & 0xFFFFFFFFull);
    const uint32_t draw = (uint32_t)(id & 0xFFFFFu);
    double z0, z1;
    float f0, f1;
    if (edge) {
      // hand-made blocks around the switch of the logarithm and at the ends of u1 (the Philox output is bypassed): the same
      // formulas on chosen (x, y, z, w)
      const uint32_t xs[8] = {0xFF000000u, 0xFEFFFFFFu, 0xFFFFFFFFu, 0u, 1u, 0x80000000u, 0xFF000001u, 0xFFFFFFE0u};
      const uint32_t x = xs[i & 7], y = (uint32_t)(id * 2654435761ull), zz = (uint32_t)(id * 40503ull + (i >> 3) * 977ull), ww = (uint32_t)(i * 7919ull);
      const double u1 = (tph_k53(x, y) + 1.0) * 0x1.0p-53, u2 = tph_k53(zz, ww) * 0x1.0p-53;
      const double rad = tph_sqrt(-2.0 * tph_log(u1));
      double s, c;
      tph_sincospi(2.0 * u2, s, c);
      z0 = rad * c; z1 = rad * s;
      float sf;
      if (x < 0xFF000000u) {
        const float k = fmaf((float)(x >> 5), 67108864.0f, (float)(y >> 6) + 1.0f);
        sf = -1.3862943611198906f * __builtin_amdgcn_logf(k * 0x1.0p-53f);
      } else {
        const float tk2 = fmaf((float)(0x7FFFFFFu - (x >> 5)), 67108864.0f, (float)(0x3FFFFFFu - (y >> 6)));
        const float t = tk2 * 0x1.0p-53f;
        sf = 2.0f * t * fmaf(t, fmaf(t, 0.33333334f, 0.5f), 1.0f);
      }
      const float radf = __builtin_amdgcn_sqrtf(sf), rev = (float)zz * 0x1.0p-32f;
      f0 = radf * __builtin_amdgcn_cosf(rev); f1 = radf * __builtin_amdgcn_sinf(rev);
    } else {
      g.normal2(draw, z0, z1);
      mf_normal2(g, draw, f0, f1);
    }
    mx = fmax(mx, fmax(fabs((double)f0 - z0), fabs((double)f1 - z1)));
    mz = fmax(mz, fmax(fabs(z0), fabs(z1)));
    cnt += 1.0;
  }
  mx = tph_block_max(mx, sh);
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * 3] = mx;
  mz = tph_block_max(mz, sh);
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * 3 + 1] = mz;
  cnt = tph_block_sum(cnt, sh);
  if (threadIdx.x == 0) out[(size_t)blockIdx.x * 3 + 2] = cnt;
}

extern "C" int tph_bench_mf_normals(tph_ctx* ctx, uint64_t seed, uint64_t first, uint64_t n_blocks, int edge, double* out3_host) {
  TPH_REQUIRE(ctx && out3_host && n_blocks > 0, "tph_bench_mf_normals: bad argument");
  const int grid = 2048;
  if (tph_scratch_reserve(ctx, sizeof(double) * 3 * grid)) return -1;
  double* part = (double*)ctx->scratch;
  hipLaunchKernelGGL(k_mf_normals_check, dim3(grid), dim3(256), 0, ctx->stream, seed, first, n_blocks, edge, part);
  TPH_LAUNCH_CHECK();
  std::vector<double> h(3 * (size_t)grid);
  TPH_HIP(hipMemcpyAsync(h.data(), part, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  double mx = 0.0, mz = 0.0, cnt = 0.0;
  for (int b = 0; b < grid; ++b) { mx = fmax(mx, h[3 * b]); mz = fmax(mz, h[3 * b + 1]); cnt += h[3 * b + 2]; }
  out3_host[0] = mx; out3_host[1] = mz; out3_host[2] = cnt;
  return 0;
}

// ---- the screen's one empirical assumption, re-checked on every context before its first screened launch ----------------------
// The margin m_r is derived term by term except for |z~ - z| <= 2^-13, the accuracy of v_log_f32 / v_sqrt_f32 / v_sin_f32 /
// v_cos_f32 in the Box-Muller pair: that figure was MEASURED on gfx950 (tph_bench_mf_normals over 2^32 blocks).  A part or a
// ROCm release whose transcendentals are less accurate would let the screen drop an in-bounds attempt without any diagnostic
// (TPH_OPT_MF_AUDIT is off by default).  So the first use of the screen on a context measures the error again -- the hand-made
// edge blocks around the switch of the logarithm and a few million random blocks, ~50 us -- and keeps the screen only if the
// budget holds; otherwise the FP64 kernels take over (one line on stderr says so).  Not run under stream capture (it reads its
// result back): a capture that meets an unchecked context takes the FP64 path.
bool tph_mf_screen(tph_ctx* ctx) { return ctx->screen && tph_mf_selftest(ctx); }
bool tph_mf_selftest(tph_ctx* ctx) {
  if (ctx->d <= 16 || ctx->d > MF_MAX_DIM) return false;
  if (ctx->mf_checked > 0) return true;
  if (ctx->mf_checked < 0) return false;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
  double edge[3] = {1.0, 0.0, 0.0}, rnd[3] = {1.0, 0.0, 0.0};
  const int rc = tph_bench_mf_normals(ctx, 0x5EEDull, 0, 1ull << 16, 1, edge) || tph_bench_mf_normals(ctx, 0x5EEDull, 1ull << 40, 1ull << 22, 0, rnd);
  const double worst = fmax(edge[0], rnd[0]);
  if (rc == 0 && worst <= 0x1.0p-13) {
    ctx->mf_checked = 1;
    return true;
  }
  ctx->mf_checked = -1;
  fprintf(stderr, "tempest_hip: FP32 Box-Muller error %.3g exceeds the screen's budget 2^-13 on this device: redraw-dominated steps "
                  "use the FP64 kernels (TPH_OPT_SCREEN off)\n", worst);
  return false;
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_propose_mf(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
