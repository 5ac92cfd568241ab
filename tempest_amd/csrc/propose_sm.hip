// Redraw-dominated proposals at 16 < d <= 100, one mode: every lane walks the rows of its own attempt.
// Reference: tempest/mcmc.py:225-249 (tpCN), :301-312 (RWM): a walker's proposal is REDRAWN until it lies in the unit cube.
//
// In the first iterations of a high-dimensional run nearly every attempt leaves the cube (50-D at sigma_0: ~98 %, i.e. ~60
// attempts per particle and step; 100-D: ~290), and almost all of them fail early: L is lower-triangular, so coordinate r of
// an attempt needs only the normals z_0..z_r, and an attempt that violates a bound at row r* is decided after r*/2 Box-Muller
// pairs and r*^2/2 FMAs.  What such a step costs is therefore the number of Box-Muller pairs it generates (848 SIMD cycles
// per wave-call for two normals, against ~5 per FMA).  k_propose_ml (lane groups walking their particle's attempts) spends
// several times the pairs an ideal schedule needs: the particles of a wave diverge in their attempt counts and every round
// draws a full group of pairs.  Here:
//   * a LANE owns an attempt; its normals z_0..z_r live in the lane's column of a tile zs[j][lane] (first rows in LDS, the
//     rest in global scratch: "LDS budget" below);
//   * every step, EVERY busy lane draws the next two Box-Muller pairs of its attempt (two independent chains) and evaluates
//     the next four rows -- whatever row it has reached: the Cholesky factor sits in LDS (one copy per workgroup, zero above
//     the diagonal) and a lane reads ITS rows; the column loop runs in blocks of 16 to the deepest lane's row, a lane taking
//     part in the blocks its own rows reach.  An attempt stops at its first out-of-bounds coordinate and the lane starts its
//     next attempt in the next step;
//   * G consecutive lanes work on ONE particle: lane g tries attempts g, g+G, g+2G, ... and the first in-bounds attempt IN
//     ATTEMPT ORDER wins -- exactly the proposal the sequential loop returns (counter-based draws keyed by the attempt
//     number).  Lanes whose attempt number has passed the best success so far stop at once;
//   * particles come from a global queue in chunks of 4 (one atomic per chunk, requested a grab ahead), so waves finish together;
//   * the rows an attempt has passed are parked in the lane's record of a global scratch (32 B per step) and copied to u'
//     when the particle is decided.
// Draws, attempt order and the arithmetic of a row (ascending-j FMA chain, v = fma(b, (L z)_r, base_r)) are those of the
// other proposal kernels: the same proposal to rounding.  tpCN's Mahalanobis forms are left to k_maha_tile (tri.h products).
// Two earlier forms, measured on 65 536 x 50-D from the prior (99 attempts per particle; k_propose_ml 3.5 ms): z in
// REGISTERS with one unrolled case per stage -- the compiler issued every scalar matrix load of a case ahead of its first FMA
// and spilled ~100 SGPRs through v_writelane / v_readlane, four spill instructions per FMA; and a "stage machine" that ran the
// rows stage by stage with wave-uniform matrix rows through the scalar cache -- 4.7 ms: at the 5 waves per CU the z tile
// allows, every stage's serial chain of scalar-load, LDS and store latencies (6 300 cycles per stage pass) was exposed.
#include "common.h"
#include "tri.h"

constexpr int SM_CAP = PROP_MAX_ATTEMPTS;      // attempts 0 .. SM_CAP-1, then the current point is proposed
constexpr int SM_CHUNK = 4;                    // particles per queue grab (small: the last chunks set the kernel's tail);
                                               // the prefetch registers pb0..pb3 of k_propose_sm are one per particle of a chunk
constexpr int SM_ROWS = 4;                     // rows per step (two Box-Muller pairs)

__device__ __forceinline__ void sm_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LDS budget.  The kernel is latency-bound (every step is a chain of Philox, LDS and store latencies), so what matters is how many
// waves fit a CU around the one copy of L.  A full-row copy (100-D: 93 KB) and a d-row z tile per wave (51 KB) leave room for ONE
// wave at 100-D and five at 50-D.  Hence
//   * the copy of L is a STAIRCASE: the four rows of stage s hold 4 (s + 1) columns rounded up to 16, zeros beyond the diagonal
//     (100-D: 50 KB, 50-D: 16 KB), and the column loop masks lanes per 16-column block (a shallow lane skips the blocks its rows
//     do not reach; an idle lane skips them all);
//   * the z tile keeps only its first `zl` rows in LDS (32: 16 KB per wave -- seven waves per CU at 100-D, eight at 50-D); the rows
//     beyond live in a lane-private column of global scratch.  From the prior two attempts in three die before row 32.
// Measured, 65 536 x 50-D RWM from the prior: 1 719 us with full rows and a 52-row tile (5 waves), 1 545 us in this form; 131 072 x
// 100-D tpCN: 8.0 ms (zl = 32; 8.9 ms at 48, 10.4 ms at 64, 9.6 ms at 16) against 14.6 ms for k_propose_ml.
// Stage s starts at smd_off(s): 12 doubles of slack per stage carry a skew of 2 (s mod 6) doubles, so that lanes at six
// consecutive stages read from six different 16-byte bank groups (lanes at the same stage read the same address: a broadcast);
// the rows of a stage are smd_rs(s) apart.  (12, not 16: at 100-D the copy then is 48 992 bytes and SEVEN 16 KB tiles fit beside it.)
__host__ __device__ static inline int smd_rs(int s) { return 16 * ((s >> 2) + 1); }
__host__ __device__ static inline int smd_base(int s) { const int q = s >> 2, t = s & 3; return 64 * (q + 1) * (2 * q + t) + 12 * s; }
__host__ __device__ static inline int smd_off(int s) { return smd_base(s) + 2 * (s % 6); }

// the staircase copy of L in global memory (the workgroups copy it to LDS word for word); one workgroup
static __global__ void __launch_bounds__(256) k_sm_stairs(const double* __restrict__ chol, int d, int nst, double* __restrict__ Lg) {
  const int total = smd_base(nst);
  for (int e = threadIdx.x; e < total; e += blockDim.x) Lg[e] = 0.0;
  __syncthreads();
  for (int e = threadIdx.x; e < SM_ROWS * nst * 128; e += blockDim.x) {
    const int r = e >> 7, j = e & 127;
    if (r < d && j <= r) Lg[smd_off(r >> 2) + (r & 3) * smd_rs(r >> 2) + j] = chol[(size_t)r * d + j];
  }
}

template <int KERNEL, bool HAS_BC>
__global__ void __launch_bounds__(512) k_propose_sm(const double* __restrict__ u, int64_t n, int64_t ld, int d,
                                                    const double* __restrict__ means, const double* __restrict__ Lg,
                                                    const double* __restrict__ sigmas, const uint8_t* __restrict__ bc,
                                                    uint64_t seed, tph_stepctl tick, int64_t item0, double* __restrict__ up,
                                                    const double* __restrict__ bfac, double* __restrict__ vscr,
                                                    double* __restrict__ bscr, int lgG,
                                                    unsigned long long* __restrict__ queue /* [0] next chunk, [1] sum of
                                                    attempts, [2] particles decided */,
                                                    int zl /* rows of z held in LDS (multiple of 16) */,
                                                    double* __restrict__ zscr /* the rows beyond, per wave */) {
  extern __shared__ double sm_lds[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = 1 << lgG;
  const int nst = (d + SM_ROWS - 1) / SM_ROWS, dc = SM_ROWS * nst, npairs = (d + 1) >> 1;
  const int lt_len = smd_base(nst);
  double* Lt = sm_lds;                                         // the workgroup's copy of L (staircase)
  double* zs = Lt + (size_t)lt_len + (size_t)wid * zl * 64;    // [zl][64] normals of this wave's attempts, rows 0 .. zl-1
  for (int e = threadIdx.x; e < lt_len; e += blockDim.x) Lt[e] = Lg[e];
  for (int e = lane; e < zl * 64; e += 64) zs[e] = 0.0;        // stale entries meet zeros of L: they must be finite
  __syncthreads();
  const uint32_t tk = tick;              // the step's RNG tick, read from the control block ONCE: inside the loop its two loads would
                                         // wait for every load in flight (vmcnt counts in order), the prefetched coordinates included
  const size_t wave_id = (size_t)blockIdx.x * (blockDim.x >> 6) + wid;
  double* __restrict__ vrec = vscr + (wave_id * 64 + lane) * (size_t)dc;      // this lane's record of passed rows
  // z rows zl .. of this wave's attempts, [row - zl][lane] (zeroed when allocated, only ever holds normals: finite)
  double* __restrict__ zg = zscr + wave_id * (size_t)(((dc + 15) & ~15) - zl) * 64;
  // the groups' particles as contiguous records base[r] = mu_r + a (u_r - mu_r) (tpCN) or u_r (RWM), written when a group takes a
  // particle: a step reads four consecutive doubles of it (one 32-byte sector, shared by the group's lanes), where the
  // dimension-major u costs four scattered 64-byte sectors per lane and step -- 8 GB per launch at 65 536 x 50-D from the prior
  double* __restrict__ brec = bscr + (wave_id * 32 + (size_t)(lane >> lgG)) * (size_t)dc;
  const double sigma = sigmas[0];
  const double a_fac = (KERNEL == TPH_KERNEL_TPCN) ? tph_sqrt(1.0 - sigma * sigma) : 1.0;
  const int64_t nchunks = (n + SM_CHUNK - 1) / SM_CHUNK;
  const int g = lane >> lgG;
  const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane & ~(G - 1));
  const bool leader = (lane & (G - 1)) == 0;

  // the wave's pool: the chunk grabbed last
  int64_t pool_row = 0;
  int pool_next = 0, pool_cnt = 0;
  bool exhausted = false;
  unsigned long long n_att = 0, n_dec = 0;
#ifdef SM_PROFILE
  long long pf_bm = 0, pf_rows = 0, pf_refill = 0, pf_steps = 0, pf_epi = 0, pf_e1 = 0, pf_e0 = 0, pf_deep = 0, pf_deep_steps = 0, pf_t = clock64();
#define SM_PF(acc) do { const long long now_ = clock64(); acc += now_ - pf_t; pf_t = now_; } while (0)
#else
#define SM_PF(acc) do { } while (0)
#endif

  // per-lane state; `best` (lowest successful attempt of the lane's particle so far) is kept equal on the G lanes of a group
  int64_t row = -1;                // particle this lane works on (-1: none)
  int att = 0, stage = 0, won = -1, best = SM_CAP;
  bool active = false;
  double b_fac = 0.0;
  double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0, p5 = 0.0, p6 = 0.0, p7 = 0.0;   // rows 0..7 of the lane's attempt
  // the pool's particles, prefetched when the chunk is grabbed: lane r holds base coordinate r of each (n_dim <= 64)
  double pb0 = 0.0, pb1 = 0.0, pb2 = 0.0, pb3 = 0.0;
  double pc0 = 0.0, pc1 = 0.0, pc2 = 0.0, pc3 = 0.0;          // coordinate 64 + lane (n_dim > 64)
  // the chunk AFTER the current pool, requested from the queue one grab ahead (lane 0 holds the ticket): the atomic's
  // round trip is over long before the pool runs dry
  unsigned long long ahead = 0;
  if (lane == 0) ahead = atomicAdd(&queue[0], 1ull);

#pragma unroll 1
  for (;;) {
    // ---- groups whose particle is decided: write u', take the next particle
    {
      const unsigned long long act = __ballot(active);
      const bool gdone = (act & gmask) == 0ull;
      const bool more = !exhausted || pool_next < pool_cnt;
      unsigned long long todo = __ballot(leader && gdone && (row >= 0 || more));
      bool refilled = false;
      if (todo) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the parked rows below were stored by other lanes
#pragma unroll 1
      while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int gg = L >> lgG;
        const int64_t orow = __shfl(row, L, 64);
        if (orow >= 0) {
          const int obest = __shfl(best, L, 64);
          if (obest < SM_CAP) {
            const unsigned long long wm = __ballot(won == obest && g == gg);
            const int wl = __ffsll((long long)wm) - 1;
            // rows 0..7 of the winning attempt never left its registers (see below)
            const int rtop = d < 2 * SM_ROWS ? d : 2 * SM_ROWS;
            {
              const double w0 = __shfl(p0, wl, 64), w1 = __shfl(p1, wl, 64), w2 = __shfl(p2, wl, 64), w3 = __shfl(p3, wl, 64);
              const double w4 = __shfl(p4, wl, 64), w5 = __shfl(p5, wl, 64), w6 = __shfl(p6, wl, 64), w7 = __shfl(p7, wl, 64);
              const double mine8 = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : lane == 3 ? w3 : lane == 4 ? w4 : lane == 5 ? w5
                                   : lane == 6 ? w6 : w7;
              if (lane < rtop) up[(size_t)lane * ld + orow] = mine8;
            }
            const double* __restrict__ src = vscr + (wave_id * 64 + wl) * (size_t)dc;
            for (int r = rtop + lane; r < d; r += 64)
              up[(size_t)r * ld + orow] = __hip_atomic_load(src + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n_att += (unsigned long long)(obest + 1);
          } else {                   // redraw cap reached (the reference would loop on): the current point is proposed
            for (int r = lane; r < d; r += 64) {
              const double uj = u[(size_t)r * ld + orow];
              up[(size_t)r * ld + orow] = (KERNEL == TPH_KERNEL_TPCN) ? (uj - means[r]) + means[r] : uj;
            }
            n_att += (unsigned long long)SM_CAP;
          }
          n_dec += 1ull;
        }
        // next particle of the pool; an empty pool takes the chunk requested one grab ago and requests the next one
        if (pool_next >= pool_cnt && !exhausted) {
          const unsigned long long c = __shfl(ahead, 0, 64);
          if ((int64_t)c >= nchunks) {
            exhausted = true;
          } else {
            if (lane == 0) ahead = atomicAdd(&queue[0], 1ull);
            pool_row = (int64_t)c * SM_CHUNK;
            pool_cnt = (int)((n - pool_row) < SM_CHUNK ? (n - pool_row) : SM_CHUNK);
            pool_next = 0;
            // base coordinates mu_r + a (u_r - mu_r) (tpCN) / u_r (RWM) of the chunk's particles, lane = r; padding rows sit
            // at 0.5 (always in bounds, never looked at)
            const int r = lane < d ? lane : d - 1;
            const int64_t last = pool_row + pool_cnt - 1;
            const double u0 = u[(size_t)r * ld + pool_row], u1 = u[(size_t)r * ld + (pool_row + 1 <= last ? pool_row + 1 : last)];
            const double u2 = u[(size_t)r * ld + (pool_row + 2 <= last ? pool_row + 2 : last)];
            const double u3 = u[(size_t)r * ld + (pool_row + 3 <= last ? pool_row + 3 : last)];
            const double mr = (KERNEL == TPH_KERNEL_TPCN) ? means[r] : 0.0;
            pb0 = lane < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, u0 - mr, mr) : u0) : 0.5;
            pb1 = lane < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, u1 - mr, mr) : u1) : 0.5;
            pb2 = lane < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, u2 - mr, mr) : u2) : 0.5;
            pb3 = lane < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, u3 - mr, mr) : u3) : 0.5;
            if (d > 64) {
              const int rh = lane + 64 < d ? lane + 64 : d - 1;
              const double h0 = u[(size_t)rh * ld + pool_row], h1 = u[(size_t)rh * ld + (pool_row + 1 <= last ? pool_row + 1 : last)];
              const double h2 = u[(size_t)rh * ld + (pool_row + 2 <= last ? pool_row + 2 : last)];
              const double h3 = u[(size_t)rh * ld + (pool_row + 3 <= last ? pool_row + 3 : last)];
              const double mh = (KERNEL == TPH_KERNEL_TPCN) ? means[rh] : 0.0;
              pc0 = lane + 64 < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, h0 - mh, mh) : h0) : 0.5;
              pc1 = lane + 64 < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, h1 - mh, mh) : h1) : 0.5;
              pc2 = lane + 64 < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, h2 - mh, mh) : h2) : 0.5;
              pc3 = lane + 64 < d ? ((KERNEL == TPH_KERNEL_TPCN) ? fma(a_fac, h3 - mh, mh) : h3) : 0.5;
            }
          }
        }
        int64_t nrow = -1;
        if (pool_next < pool_cnt) {
          nrow = pool_row + pool_next;
          const double pb = pool_next == 0 ? pb0 : pool_next == 1 ? pb1 : pool_next == 2 ? pb2 : pb3;
          const double pc = pool_next == 0 ? pc0 : pool_next == 1 ? pc1 : pool_next == 2 ? pc2 : pc3;
          ++pool_next;
          if (lane < dc) bscr[(wave_id * 32 + (size_t)gg) * (size_t)dc + lane] = pb;     // the group's record
          if (lane + 64 < dc) bscr[(wave_id * 32 + (size_t)gg) * (size_t)dc + 64 + lane] = pc;
          refilled = true;
        }
        if (g == gg) {
          row = nrow; won = -1; stage = 0; best = SM_CAP;
          b_fac = nrow >= 0 ? (KERNEL == TPH_KERNEL_TPCN ? bfac[nrow] : sigma) : 0.0;
          att = lane & (G - 1);
          active = nrow >= 0 && att < SM_CAP;
        }
      }
      if (refilled) {                  // the records just written are read by other lanes of this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      SM_PF(pf_refill);
      if (!__any(row >= 0)) break;
    }

    // ---- this step's rows r0 .. r0+3 of every busy lane; their base coordinates are requested now and used
    // after the products
    const int r0 = SM_ROWS * stage;
    const double2 b01 = *(const double2*)(brec + r0), b23 = *(const double2*)(brec + r0 + 2);
    // the next two Box-Muller pairs of the attempt (pairs 2 stage, 2 stage + 1), into the lane's column
    if (active) {
      tph_rng gz(seed, tk, TPH_TAG_NORMAL, (uint64_t)(item0 + row));
      const uint32_t base_draw = (uint32_t)att * (uint32_t)npairs;
      const int p0 = 2 * stage, last = npairs - 1;
      double t0, t1, t2, t3;
      gz.normal2(base_draw + (uint32_t)p0, t0, t1);
      gz.normal2(base_draw + (uint32_t)(p0 < last ? p0 + 1 : last), t2, t3);
      if (r0 < zl) {
        double* zw = zs + (size_t)r0 * 64 + lane;
        zw[0] = t0; zw[64] = t1; zw[128] = t2; zw[192] = t3;
      } else {
        double* __restrict__ zw = zg + (size_t)(r0 - zl) * 64 + lane;
        zw[0] = t0; zw[64] = t1; zw[128] = t2; zw[192] = t3;
      }
    }
    sm_wave_sync();
    SM_PF(pf_bm);

    // ---- L z, each lane ITS rows (LDS copy of L); the block loop runs to the deepest busy lane's row.  (Rows r0+2, r0+3 from a
    // padded copy in global memory instead -- two pipes for the two 16-byte loads per row and column pair -- doubled the loop's
    // time: per-lane 16-byte loads through L1 are slower than the LDS reads they were meant to relieve.)
    int smax = active ? stage : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(smax, o, 64); smax = other > smax ? other : smax; }
    const int jm = SM_ROWS * (__builtin_amdgcn_readfirstlane(smax) + 1);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // staircase rows: a lane takes part in the 16-column blocks its rows reach (idle lanes in none)
    const double* __restrict__ Lr = Lt + smd_off(stage);
    const int rs = smd_rs(stage);
    const int mylen = active ? SM_ROWS * (stage + 1) : 0;
#pragma unroll 1
    for (int jb = 0; jb < jm; jb += 16) {
#ifdef SM_PROFILE
      if (jb == zl) { SM_PF(pf_rows); ++pf_deep_steps; }          // from here on: the blocks whose z comes from global scratch -> pf_epi is reused below
#endif
      if (jb < mylen) {
        const double* __restrict__ Lb = Lr + jb;
        double zz[16];
        if (jb < zl) {                       // (wave-uniform) the block's 16 normals: LDS tile or the lane's global column
          const double* zc = zs + (size_t)jb * 64 + lane;
#pragma unroll
          for (int j = 0; j < 16; ++j) zz[j] = zc[(size_t)j * 64];
        } else {
          const double* __restrict__ zc = zg + (size_t)(jb - zl) * 64 + lane;
#pragma unroll
          for (int j = 0; j < 16; ++j) zz[j] = zc[(size_t)j * 64];
        }
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const double2 l0 = *(const double2*)(Lb + j), l1 = *(const double2*)(Lb + rs + j);
          const double2 l2 = *(const double2*)(Lb + 2 * rs + j), l3 = *(const double2*)(Lb + 3 * rs + j);
          a0 = fma(l0.x, zz[j], a0); a1 = fma(l1.x, zz[j], a1); a2 = fma(l2.x, zz[j], a2); a3 = fma(l3.x, zz[j], a3);
          a0 = fma(l0.y, zz[j + 1], a0); a1 = fma(l1.y, zz[j + 1], a1); a2 = fma(l2.y, zz[j + 1], a2); a3 = fma(l3.y, zz[j + 1], a3);
        }
      }
    }
#ifdef SM_PROFILE
    if (jm > zl) SM_PF(pf_deep); else SM_PF(pf_rows);
#else
    SM_PF(pf_rows);
#endif
    int mine = INT32_MAX;             // this lane's successful attempt of the step, if any
#ifdef SM_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SM_PF(pf_e0);
#endif
    if (active) {
      const double acc[SM_ROWS] = {a0, a1, a2, a3};
      const double bs[SM_ROWS] = {b01.x, b01.y, b23.x, b23.y};
      double v[SM_ROWS];
      bool ok = true;
#pragma unroll
      for (int q = 0; q < SM_ROWS; ++q) {            // padding rows of the last stage: L row zero, base 0.5 -> in bounds
        double x = fma(b_fac, acc[q], bs[q]);
        if (HAS_BC) {                                // periodic / reflective dimensions: a separate instantiation
          const int r = r0 + q;
          const uint8_t f = r < d ? bc[r] : (uint8_t)TPH_BC_STRICT;
          if (f == TPH_BC_PERIODIC) x = bc_periodic(x);
          else if (f == TPH_BC_REFLECTIVE) x = bc_reflective(x);
          else ok = ok && (x >= 0.0) && (x <= 1.0);
        } else {
          ok = ok && (x >= 0.0) && (x <= 1.0);
        }
        v[q] = x;
      }
      // park the rows this attempt has passed: rows 0..7 in registers (four in five steps are first or second stages, or die),
      // later ones in the lane's record
      if (stage == 0) { p0 = v[0]; p1 = v[1]; p2 = v[2]; p3 = v[3]; }
      else if (stage == 1) { p4 = v[0]; p5 = v[1]; p6 = v[2]; p7 = v[3]; }
      else if (ok) {
        *(double2*)(vrec + r0) = make_double2(v[0], v[1]);
        *(double2*)(vrec + r0 + 2) = make_double2(v[2], v[3]);
      }
      if (ok) {
        if (++stage == nst) {       // every row in bounds: this attempt is a success
          won = att;
          mine = att;
          active = false;
          stage = 0;                // (an idle lane still walks rows 0..3 with the others: keep its row index inside the tables)
        }
      } else {
        att += G;
        stage = 0;
      }
    }
    SM_PF(pf_e1);
    // the group's lowest success so far (xor-shuffles inside the group); attempts numbered above it are moot, also in flight
    for (int o = 1; o < G; o <<= 1) { const int other = __shfl_xor(mine, o, 64); mine = other < mine ? other : mine; }
    best = mine < best ? mine : best;
    if (row >= 0 && won < 0) active = att < (best < SM_CAP ? best : SM_CAP);
    SM_PF(pf_epi);
#ifdef SM_PROFILE
    ++pf_steps;
#endif
  }
  if (lane == 0) {
    atomicAdd(&queue[1], n_att);
    atomicAdd(&queue[2], n_dec);
#ifdef SM_PROFILE
    atomicAdd(&queue[3], (unsigned long long)pf_bm); atomicAdd(&queue[4], (unsigned long long)pf_rows);
    atomicAdd(&queue[5], (unsigned long long)pf_refill); atomicAdd(&queue[6], (unsigned long long)pf_steps);
    atomicAdd(&queue[7], (unsigned long long)pf_epi);
    atomicAdd(&queue[8], (unsigned long long)pf_e0); atomicAdd(&queue[9], (unsigned long long)pf_e1);
    atomicAdd(&queue[10], (unsigned long long)pf_deep); atomicAdd(&queue[11], (unsigned long long)pf_deep_steps);
#endif
  }
}

#include "maha_tile.h"

template <int KERNEL>
static int propose_sm(tph_ctx* ctx, double* u, int64_t n, int64_t ld, const double* means, const double* chol, const double* winv,
                      const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, tph_stepctl tick, int64_t item0,
                      double* up, double* maha_u, double* maha_up, uint8_t* pend) {
  const int d = ctx->d;
  TPH_REQUIRE(d <= 100, "tph_propose (row walker): n_dim=%d > 100", d);
  const int nst = (d + SM_ROWS - 1) / SM_ROWS, dc = SM_ROWS * nst;
  const int dcp = (dc + 15) & ~15;
  const int64_t nchunks = (n + SM_CHUNK - 1) / SM_CHUNK;
  // workgroup = as many waves as fit around one LDS copy of L with a z tile each (one workgroup per CU)
  // rows of z held in LDS (TPH_OPT_SM_THRESHOLD, a multiple of 16; default 32: see the head of the file)
  int zl = ctx->sm_thr > 0 ? ((ctx->sm_thr + 15) & ~15) : 32;
  if (zl > dcp) zl = dcp;
  const size_t lt_doubles = (size_t)smd_base(nst);
  const size_t lt_bytes = sizeof(double) * lt_doubles, z_bytes = sizeof(double) * (size_t)zl * 64;
  TPH_REQUIRE(lt_bytes + z_bytes <= 160 * 1024, "tph_propose (row walker): n_dim=%d does not fit the LDS", d);
  int wv = (int)((160 * 1024 - lt_bytes) / z_bytes);
  if (wv > 8) wv = 8;
  const int cus = ctx->n_simd / 4;
  int64_t groups = (nchunks + wv - 1) / wv;                  // never more waves than there are chunks
  if (groups > cus) groups = cus;
  const int64_t waves = groups * wv;
  // lanes per particle (log2) = attempts of a particle in flight.  Measured optimum (8 waves per CU, steps of 3 ... 130 attempts
  // per particle): 3 at 85-128 particles per wave (131 072 x 100-D, 262 144 x 32-D), 4 at 8-32 (65 536 and 16 384 x 50-D:
  // 781 vs 825 us at 7 attempts, 989 vs 1 063 us at 23); 2 and 5 lose everywhere (1 249 / 1 133 us at 65 536 x 50-D, 7 attempts)
  int lgG = ctx->sm_lanes;
  if (lgG <= 0) {
    const double pw = (double)n / (double)waves;
    lgG = pw >= 48.0 ? 3 : 4;
  }
  if (lgG > 6) lgG = 6;
  const size_t lds = lt_bytes + (size_t)wv * z_bytes;
  // persistent buffers: blocked copies of L and L^-1 (tri.h), the queue words, the per-lane columns of passed rows
  const size_t tb8 = tri_blocked_doubles(d);
  const size_t need_small = sizeof(double) * (tb8 + lt_doubles) + 128;
  if (ctx->sm_small_bytes < need_small) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->sm_small) ctx->retired.push_back(ctx->sm_small);
    ctx->sm_small = nullptr; ctx->sm_small_bytes = 0; ctx->sm_epoch = -1;
    TPH_HIP(hipMalloc((void**)&ctx->sm_small, need_small));
    ctx->sm_small_bytes = need_small;
  }
  // scratch: passed rows (64 lane records per wave), base records (32 per wave), z rows zl.. (64 columns per wave)
  const size_t zg_doubles = (size_t)(dcp - zl) * 64 * (size_t)waves;
  const size_t need_scr = sizeof(double) * ((size_t)dc * (size_t)waves * (64 + 32) + zg_doubles);
  if (ctx->sm_scr_bytes < need_scr) {
    TPH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->sm_scr) ctx->retired.push_back(ctx->sm_scr);
    ctx->sm_scr = nullptr; ctx->sm_scr_bytes = 0;
    TPH_HIP(hipMalloc((void**)&ctx->sm_scr, need_scr));
    TPH_HIP(hipMemset(ctx->sm_scr, 0, need_scr));          // the z columns must hold finite numbers from the start
    ctx->sm_scr_bytes = need_scr;
  }
  unsigned long long* queue = (unsigned long long*)ctx->sm_small;
  double* Wb = (double*)((char*)ctx->sm_small + 128);
  double* Lg = Wb + tb8;
  // a launch being captured into a hipGraph records the rebuild (a replayed step never re-enters this host code)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  TPH_HIP(hipStreamIsCapturing(ctx->stream, &cap));
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (capturing || ctx->modes_epoch <= 0 || ctx->sm_epoch != ctx->modes_epoch || ctx->sm_src != (const void*)chol ||
      ctx->sm_kernel != KERNEL) {
    hipLaunchKernelGGL(k_sm_stairs, dim3(1), dim3(256), 0, ctx->stream, chol, d, nst, Lg);
    if (KERNEL == TPH_KERNEL_TPCN) hipLaunchKernelGGL(k_tri_block, dim3(1), dim3(256), 0, ctx->stream, winv, d, Wb);
    ctx->sm_epoch = capturing ? -1 : ctx->modes_epoch; ctx->sm_src = (const void*)chol; ctx->sm_kernel = KERNEL;
  }
  hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, ctx->stream, (unsigned int*)queue, 32);
  // pending moves; tpCN: the form at u (first step of a run) and every particle's step scale, parked in maha_up until the
  // closing pass overwrites it with the form at u'
  if (pend || KERNEL == TPH_KERNEL_TPCN || maha_u)
    if (launch_maha_tile<KERNEL, 0>(ctx, u, n, ld, means, Wb, up, maha_u, tick, pend, nullptr, dof, sigmas, seed, item0, maha_up)) return -1;
  double* const zscr = ctx->sm_scr + (size_t)dc * (size_t)waves * (64 + 32);
#define TPH_SM(BC)                                                                                                       \
  do {                                                                                                                   \
    if (lds > 64 * 1024)                                                                                                 \
      TPH_HIP(hipFuncSetAttribute((const void*)k_propose_sm<KERNEL, BC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_propose_sm<KERNEL, BC>), dim3((unsigned)groups), dim3(64 * wv), lds, ctx->stream, (const double*)u, n, ld, d, \
                       means, (const double*)Lg, sigmas, bc, seed, tick, item0, up, (const double*)maha_up, ctx->sm_scr,   \
                       ctx->sm_scr + (size_t)dc * (size_t)waves * 64, lgG, queue, zl, zscr);                              \
  } while (0)
  if (bc) TPH_SM(true); else TPH_SM(false);
#undef TPH_SM
  TPH_LAUNCH_CHECK();
  if (KERNEL == TPH_KERNEL_TPCN || maha_up || tick.ctl)
    if (launch_maha_tile<KERNEL, 1>(ctx, u, n, ld, means, Wb, up, maha_up, tick, nullptr, queue, dof, sigmas, seed, item0, nullptr)) return -1;
  return 0;
}

int tph_propose_sm(tph_ctx* ctx, int kernel, double* u, int64_t n, int64_t ld, const double* means, const double* chol,
                   const double* winv, const double* dof, const double* sigmas, const uint8_t* bc, uint64_t seed, uint32_t tick0,
                   const double* ctl, int64_t item0, double* up, double* maha_u, double* maha_up, uint8_t* pend) {
  const tph_stepctl tick{tick0, ctl};
  if (kernel == TPH_KERNEL_TPCN)
    return propose_sm<TPH_KERNEL_TPCN>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_u, maha_up, pend);
  return propose_sm<TPH_KERNEL_RWM>(ctx, u, n, ld, means, chol, winv, dof, sigmas, bc, seed, tick, item0, up, maha_u, maha_up, pend);
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_propose_sm(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
