// Tempered reweighting over the whole particle history.
// Reference: tempest/state_manager.py:418-480 (compute_logw_and_logz), tempest/steps/reweight.py:88-118
// (_compute_metric_and_weights), tempest/tools.py:120-135 (effective_sample_size).
//
// With the cached log-mixture C_s (ctx.hip) one evaluation of a trial beta is a single streaming pass
//     v_s = beta*l_s - C_s ;  (max v, sum e^{v-max}, sum e^{2(v-max)})
// over 16 B per historical particle: HBM-bound.  ESS = s1^2/s2, logZ = max + log s1.
#include "common.h"

struct tph_betas { double b[TPH_MAX_NB]; };

struct trip { double m, s1, s2; };

__device__ __forceinline__ double2 nt_load2(const double2* p) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d a = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
  return make_double2(a.x, a.y);
}

__device__ __forceinline__ trip trip_merge(trip a, trip b) {
  double M = fmax(a.m, b.m);
  double fa = exp(a.m - M), fb = exp(b.m - M);
  trip r;
  r.m = M;
  r.s1 = a.s1 * fa + b.s1 * fb;
  r.s2 = a.s2 * (fa * fa) + b.s2 * (fb * fb);
  return r;
}

__device__ __forceinline__ trip trip_wave_reduce(trip t) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    trip b;
    b.m = __shfl_down(t.m, o, 64);
    b.s1 = __shfl_down(t.s1, o, 64);
    b.s2 = __shfl_down(t.s2, o, 64);
    t = trip_merge(t, b);
  }
  return t;
}

// block reduce; result valid in thread 0.  sh: 3 * (blockDim/64) doubles.
__device__ __forceinline__ trip trip_block_reduce(trip t, double* sh) {
  t = trip_wave_reduce(t);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) { sh[wid] = t.m; sh[nw + wid] = t.s1; sh[2 * nw + wid] = t.s2; }
  __syncthreads();
  if (wid == 0) {
    if (lane < nw) { t.m = sh[lane]; t.s1 = sh[nw + lane]; t.s2 = sh[2 * nw + lane]; }
    else { t.m = -DBL_MAX; t.s1 = 0.0; t.s2 = 0.0; }
    t = trip_wave_reduce(t);
  }
  return t;
}

// Geometry of the streaming pass (host): the rows of every VIRTUAL SHARD of the canonical partition (common.h: tph_part) are
// cut into blocks that depend on the shard's own shape only -- nv rows per piece, T pieces -- never on how many shards this
// rank holds, so a shard's block partials, and the triple merged from them, are the same numbers on every number of GPUs.
// A workgroup takes either a run of RB rows inside one piece (large pieces) or PPB whole pieces (small ones); RB is at least
// 16 384 rows (contiguous 128 KB per array and block: +18 % over a grid-stride sweep, DESIGN K2) and as many more as keep a
// shard within 2048 / V blocks.
struct k2_geom {
  long long n_loc, nv, RB;
  int T, vl, BPP, PPB, bv;        // blocks per piece | pieces per block | blocks per virtual shard
};
static k2_geom k2_geometry(const tph_ctx* ctx, const tph_part& p) {
  k2_geom g;
  g.n_loc = p.n_loc; g.nv = p.nv; g.T = p.T; g.vl = p.vl;
  const int cap = ctx->reduce_grid > 0 ? ctx->reduce_grid : 2048;
  const long long bv_max = cap / p.V > 1 ? cap / p.V : 1;
  long long rb = (p.nv * (long long)p.T + bv_max - 1) / bv_max;
  rb = (rb + 4095) / 4096 * 4096;
  // at least 16 384 rows (128 KB of l and of C) per block -- or one whole piece where pieces are smaller: a history of small
  // pieces must not end up with fewer blocks than the chip has CUs (65 536 particles x 46 iterations: 184 blocks of 16 384 rows ran
  // the 15-beta pass at 97 us against 47 us before the partition; 736 blocks of one piece each)
  { const long long floor_rows = p.nv < 16384 ? (p.nv + 4095) / 4096 * 4096 : 16384; if (rb < floor_rows) rb = floor_rows; }
  g.RB = rb;
  if (rb >= p.nv) {
    g.PPB = (int)(rb / p.nv);
    if (g.PPB > p.T) g.PPB = p.T;
    g.BPP = 1;
    g.bv = (p.T + g.PPB - 1) / g.PPB;
  } else {
    g.BPP = (int)((p.nv + rb - 1) / rb);
    g.PPB = 1;
    g.bv = p.T * g.BPP;
  }
  return g;
}

// One streaming pass for NB trial betas.  Each block owns a CONTIGUOUS run of rows (measured +20 % over a grid-stride sweep
// on MI355X: fewer DRAM pages open per channel at a time) and keeps U independent 16-B non-temporal loads of l and of C in
// flight per lane.  Each lane keeps a running (m, s1, s2) per beta and rescales only when a 4-row chunk raises its maximum
// (about one exp per row).
template <int NB, int U>
__global__ void __launch_bounds__(TPH_RED_THREADS) k_reweight_reduce(const double* __restrict__ logl,
                                                                      const double* __restrict__ cmix, k2_geom g,
                                                                      tph_betas betas, double* __restrict__ partials) {
  static_assert(U % 2 == 0, "chunks are consumed in pairs (4 rows)");
  double m[NB], s1[NB], s2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) { m[b] = -DBL_MAX; s1[b] = 0.0; s2[b] = 0.0; }

  const double2* __restrict__ l2 = reinterpret_cast<const double2*>(logl);
  const double2* __restrict__ c2 = reinterpret_cast<const double2*>(cmix);
  constexpr int64_t STEP = (int64_t)TPH_RED_THREADS * U;
  const int v = blockIdx.x / g.bv, bb = blockIdx.x - v * g.bv;
  int t0, t1;
  long long q0, q1;
  if (g.BPP == 1) { t0 = bb * g.PPB; t1 = t0 + g.PPB < g.T ? t0 + g.PPB : g.T; q0 = 0; q1 = g.nv; }
  else { t0 = bb / g.BPP; t1 = t0 + 1; q0 = (long long)(bb - t0 * g.BPP) * g.RB; q1 = q0 + g.RB < g.nv ? q0 + g.RB : g.nv; }
  auto one_row = [&](long long r) {          // a row at an odd end of a run (thread 0): pieces of an odd length only
    const double l = logl[r], c = cmix[r];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double vv = betas.b[b] * l - c;
      if (vv > m[b]) { const double f = exp(m[b] - vv); s1[b] *= f; s2[b] *= f * f; m[b] = vv; }
      const double e = exp(vv - m[b]);
      s1[b] += e;
      s2[b] += e * e;
    }
  };
  for (int t = t0; t < t1; ++t) {
    long long r0 = (long long)t * g.n_loc + (long long)v * g.nv + q0, r1 = r0 + (q1 - q0);
    if ((r0 & 1) && r0 < r1) { if (threadIdx.x == 0) one_row(r0); ++r0; }
    if ((r1 & 1) && r0 < r1) { if (threadIdx.x == 0) one_row(r1 - 1); --r1; }
    const int64_t lo = r0 >> 1, hi = r1 >> 1;
    for (int64_t i = lo + threadIdx.x; i < hi; i += STEP) {
      double2 l[U], c[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t j = i + (int64_t)k * TPH_RED_THREADS;
        if (j < hi) { l[k] = nt_load2(l2 + j); c[k] = nt_load2(c2 + j); }
        else { l[k] = make_double2(0.0, 0.0); c[k] = make_double2(INFINITY, INFINITY); }   // v = -inf: contributes 0
      }
#pragma unroll
      for (int k = 0; k < U; k += 2) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const double be = betas.b[b];
          double v0 = be * l[k].x - c[k].x, v1 = be * l[k].y - c[k].y;
          double v2 = be * l[k + 1].x - c[k + 1].x, v3 = be * l[k + 1].y - c[k + 1].y;
          double vm = fmax(fmax(v0, v1), fmax(v2, v3));
          if (vm > m[b]) {
            double f = exp(m[b] - vm);
            s1[b] *= f;
            s2[b] *= f * f;
            m[b] = vm;
          }
          double e0 = exp(v0 - m[b]), e1 = exp(v1 - m[b]), e2 = exp(v2 - m[b]), e3 = exp(v3 - m[b]);
          s1[b] += (e0 + e1) + (e2 + e3);
          s2[b] += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        }
      }
    }
  }
  // Block reduction of all NB triples with ONE exp per lane and beta: block-wide maxima first (shuffles of max only),
  // then every lane rescales its sums to the block maximum and the sums are plain additions.  (Merging triples pairwise
  // costs two exps per shuffle level: at 10^6-row histories that tail was as expensive as the streaming loop itself.)
  constexpr int NW = TPH_RED_THREADS / 64;
  __shared__ double sh_m[NW][NB];
  __shared__ double sh_s[NW][2 * NB];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double M[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    double vv = m[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vv = fmax(vv, __shfl_xor(vv, o, 64));
    M[b] = vv;
    if (lane == 0) sh_m[wid][b] = vv;
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    double vv = sh_m[0][b];
#pragma unroll
    for (int w = 1; w < NW; ++w) vv = fmax(vv, sh_m[w][b]);
    M[b] = vv;
    const double f = exp(m[b] - vv);            // lanes without rows: m = -DBL_MAX, s = 0
    double a1 = s1[b] * f, a2 = s2[b] * (f * f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64); }
    if (lane == 0) { sh_s[wid][2 * b] = a1; sh_s[wid][2 * b + 1] = a2; }
  }
  __syncthreads();
  if (threadIdx.x < NB) {
    const int b = threadIdx.x;
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { a1 += sh_s[w][2 * b]; a2 += sh_s[w][2 * b + 1]; }
    double* p = partials + ((size_t)blockIdx.x * NB + b) * 3;
    p[0] = sh_m[0][b];
#pragma unroll
    for (int w = 1; w < NW; ++w) p[0] = fmax(p[0], sh_m[w][b]);
    p[1] = a1; p[2] = a2;
  }
}

// The triple of ONE virtual shard and one beta from the shard's `bv` block partials, by one wave: the shard's maximum first
// (no exp), then every partial rescaled to it (one exp each) and plain sums -- lane l takes partials l, l + 64, ... in order,
// the lanes meet in a fixed shuffle tree.  The same instructions on the same numbers wherever the shard lives.
__device__ __forceinline__ trip vshard_triple(const double* __restrict__ partials, int v, int bv, int nb, int b, int lane) {
  const double* base = partials + ((size_t)v * bv * nb + b) * 3;
  double mx = -DBL_MAX;
  for (int i = lane; i < bv; i += 64) mx = fmax(mx, base[(size_t)i * nb * 3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  double a1 = 0.0, a2 = 0.0;
  for (int i = lane; i < bv; i += 64) {
    const double* q = base + (size_t)i * nb * 3;
    const double f = exp(q[0] - mx);
    a1 += q[1] * f;
    a2 += q[2] * (f * f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64); }
  return trip{mx, a1, a2};
}

// per-shard triples [vl][nb][3] of this rank (what a sharded run all-gathers): one workgroup per shard, wave b = beta b
__global__ void __launch_bounds__(1024) k_reweight_vshards(const double* __restrict__ partials, int bv, int nb, double* __restrict__ out) {
  const int b = threadIdx.x >> 6, lane = threadIdx.x & 63, v = blockIdx.x;
  if (b >= nb) return;
  const trip t = vshard_triple(partials, v, bv, nb, b, lane);
  if (lane == 0) { double* o = out + ((size_t)v * nb + b) * 3; o[0] = t.m; o[1] = t.s1; o[2] = t.s2; }
}

// The V per-shard triples folded in shard order, one WORKGROUP per beta.  shards != NULL: the triples as stored (gathered from all
// ranks, [V][nb][3]); else they are formed here from this rank's block partials (one GPU: the same values, no trip through memory):
// the 16 waves of the beta's workgroup take the shards -- a shard's triple is one wave's work (vshard_triple) --, park them in LDS,
// and the first lane folds the V triples in shard order (V <= 48: a serial chain of merges, ~1 us).  (One workgroup for ALL betas,
// as first built, gave a wave 15 (shard, beta) pairs one after the other at 15 betas x 16 shards, each two dependent memory round
// trips: 40 us per evaluation at 65 536 particles, 291 evaluations per run.)  out_host != NULL: results into pinned host memory
// with a sequence word behind them, stored by the LAST workgroup to finish (ticket in device memory; every workgroup's results
// are fenced to system scope before its ticket) -- tph_reweight_eval polls that word instead of queueing a device-to-host copy and
// waiting on the stream: the ~25 adaptive-beta passes of one PS iteration are latency-bound, and this removes a copy packet and
// the runtime's completion path from every one of them.
__global__ void __launch_bounds__(1024) k_reweight_fold(const double* __restrict__ partials, int bv, const double* __restrict__ shards,
                                                        int V, int nb, double* __restrict__ out, double* __restrict__ seq_host, double seq,
                                                        unsigned int* __restrict__ ticket) {
  __shared__ double s_t[48 * 3];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6, b = blockIdx.x;
  if (!shards) {
    for (int v = w; v < V; v += nw) {
      const trip t = vshard_triple(partials, v, bv, nb, b, lane);
      if (lane == 0) { s_t[3 * v] = t.m; s_t[3 * v + 1] = t.s1; s_t[3 * v + 2] = t.s2; }
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  trip acc = shards ? trip{shards[(size_t)b * 3], shards[(size_t)b * 3 + 1], shards[(size_t)b * 3 + 2]} : trip{s_t[0], s_t[1], s_t[2]};
  for (int v = 1; v < V; ++v) {
    const double* q = shards ? shards + ((size_t)v * nb + b) * 3 : s_t + 3 * v;
    acc = trip_merge(acc, trip{q[0], q[1], q[2]});
  }
  if (acc.m == -DBL_MAX) acc.m = -INFINITY;  // empty input
  out[b * 3 + 0] = acc.m; out[b * 3 + 1] = acc.s1; out[b * 3 + 2] = acc.s2;
  if (seq_host) {
    __threadfence_system();
    if (atomicAdd(ticket, 1u) == (unsigned int)(nb - 1)) {      // every other workgroup's results are out
      atomicExch(ticket, 0u);
      __threadfence_system();
      __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <int NB>
static void launch_reduce(tph_ctx* ctx, const k2_geom& g, const tph_betas& bt) {
  // loads in flight per lane and array: 8 for one beta (pure streaming), fewer as the per-row exp work grows
  constexpr int U = NB == 1 ? 8 : (NB <= 4 ? 4 : 2);
  hipLaunchKernelGGL((k_reweight_reduce<NB, U>), dim3((unsigned)(g.vl * g.bv)), dim3(TPH_RED_THREADS), 0, ctx->stream, ctx->logl, ctx->cmix,
                     g, bt, ctx->partials);
}

static void launch_reduce_nb(tph_ctx* ctx, const k2_geom& g, const tph_betas& bt, int nb) {
  switch (nb) {
    case 1: launch_reduce<1>(ctx, g, bt); break;
    case 2: launch_reduce<2>(ctx, g, bt); break;
    case 3: launch_reduce<3>(ctx, g, bt); break;
    case 4: launch_reduce<4>(ctx, g, bt); break;
    case 5: launch_reduce<5>(ctx, g, bt); break;
    case 6: launch_reduce<6>(ctx, g, bt); break;
    case 7: launch_reduce<7>(ctx, g, bt); break;
    case 8: launch_reduce<8>(ctx, g, bt); break;
    case 9: launch_reduce<9>(ctx, g, bt); break;
    case 10: launch_reduce<10>(ctx, g, bt); break;
    case 11: launch_reduce<11>(ctx, g, bt); break;
    case 12: launch_reduce<12>(ctx, g, bt); break;
    case 13: launch_reduce<13>(ctx, g, bt); break;
    case 14: launch_reduce<14>(ctx, g, bt); break;
    case 15: launch_reduce<15>(ctx, g, bt); break;
    default: launch_reduce<16>(ctx, g, bt); break;
  }
}

// the streaming pass over the whole local history and the fold of the per-shard triples into `out` (device memory, or pinned
// host memory with the sequence word); with a communicator the per-shard triples of all ranks are gathered in between
static int reweight_pass(tph_ctx* ctx, const tph_betas& bt, int nb, double* out, double* seq_host, double seq) {
  const tph_part part = tph_partition(ctx, ctx->size);
  const k2_geom g = k2_geometry(ctx, part);
  if (tph_partials_reserve(ctx, sizeof(double) * 3 * (size_t)nb * g.vl * g.bv)) return -1;
  launch_reduce_nb(ctx, g, bt, nb);
  TPH_LAUNCH_CHECK();
  if (ctx->comm_active()) {
    // this rank's per-shard triples -> all-gather (rank order = shard order) -> the same fold over the V gathered triples
    // (a shard without a finite row contributes (-DBL_MAX, 0, 0), which the merge ignores)
    const size_t mine = 0, one = sizeof(double) * 3 * (size_t)nb * g.vl, all = (one + 255) / 256 * 256;
    if (tph_comm_require(ctx, all + one * ctx->world, "tph_reweight (sharded)")) return -2;
    hipLaunchKernelGGL(k_reweight_vshards, dim3(g.vl), dim3(1024), 0, ctx->stream, ctx->partials, g.bv, nb, (double*)(ctx->comm_buf + mine));
    TPH_LAUNCH_CHECK();
    if (tph_comm_allgather(ctx, mine, all, 3 * (int64_t)nb * g.vl, TPH_DT_F64)) return -2;
    hipLaunchKernelGGL(k_reweight_fold, dim3(nb), dim3(64), 0, ctx->stream, (const double*)nullptr, 0, (const double*)(ctx->comm_buf + all),
                       g.vl * ctx->world, nb, out, seq_host, seq, (unsigned int*)(ctx->small_dev + 4096));
  } else {
    hipLaunchKernelGGL(k_reweight_fold, dim3(nb), dim3(1024), 0, ctx->stream, (const double*)ctx->partials, g.bv, (const double*)nullptr, g.vl, nb,
                       out, seq_host, seq, (unsigned int*)(ctx->small_dev + 4096));
  }
  TPH_LAUNCH_CHECK();
  return 0;
}

extern "C" int tph_reweight_partials(tph_ctx* ctx, const double* betas_host, int nb, double* out_dev) {
  TPH_REQUIRE(ctx && betas_host && out_dev, "tph_reweight_partials: NULL argument");
  TPH_REQUIRE(nb >= 1 && nb <= TPH_MAX_NB, "tph_reweight_partials: nb=%d outside [1,%d]", nb, TPH_MAX_NB);
  TPH_REQUIRE(ctx->size > 0, "tph_reweight_partials: empty history");
  tph_betas bt;
  for (int i = 0; i < TPH_MAX_NB; ++i) bt.b[i] = i < nb ? betas_host[i] : 0.0;
  return reweight_pass(ctx, bt, nb, out_dev, nullptr, 0.0);
}

extern "C" int tph_reweight_eval(tph_ctx* ctx, const double* betas_host, int nb, double* out_host) {
  TPH_REQUIRE(ctx && betas_host && out_host, "tph_reweight_eval: NULL argument");
  TPH_REQUIRE(nb >= 1 && nb <= TPH_MAX_NB, "tph_reweight_eval: nb=%d outside [1,%d]", nb, TPH_MAX_NB);
  TPH_REQUIRE(ctx->size > 0, "tph_reweight_eval: empty history");
  tph_betas bt;
  for (int i = 0; i < TPH_MAX_NB; ++i) bt.b[i] = i < nb ? betas_host[i] : 0.0;
  // results + sequence word in the ctx's pinned block: [0, 48) triples, [4095] sequence
  volatile double* seqp = ctx->pinned + 4095;
  const double seq = (double)(++ctx->eval_seq);
  const int rc = reweight_pass(ctx, bt, nb, ctx->pinned, ctx->pinned + 4095, seq);
  if (rc) return rc;
  uint64_t spins = 0;
  while (*seqp != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFFF) == 0) {               // every ~1M polls: is the stream still alive?
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipErrorNotReady) {                // idle (or failed): the record is final by now
        TPH_HIP(hipStreamSynchronize(ctx->stream));
        TPH_REQUIRE(*seqp == seq, "tph_reweight_eval: the device never delivered evaluation %.0f", seq);
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int i = 0; i < 3 * nb; ++i) out_host[i] = ctx->pinned[i];
  return 0;
}

// HIP-event timing of the streaming reduction alone (no finalize, no host work between launches):
// what bench.py reports as the kernel's average launch duration.
extern "C" int tph_bench_reweight_time(tph_ctx* ctx, double beta, int nb, int reps, double* avg_ms_host) {
  TPH_REQUIRE(ctx && avg_ms_host && reps > 0, "tph_bench_reweight_time: bad argument");
  TPH_REQUIRE(nb >= 1 && nb <= TPH_MAX_NB && ctx->size > 0, "tph_bench_reweight_time: bad nb / empty history");
  tph_betas bt;
  for (int i = 0; i < TPH_MAX_NB; ++i) bt.b[i] = beta * (1.0 - 0.01 * i);
  const k2_geom g = k2_geometry(ctx, tph_partition(ctx, ctx->size));
  if (tph_partials_reserve(ctx, sizeof(double) * 3 * (size_t)nb * g.vl * g.bv)) return -1;
  hipEvent_t e0, e1;
  TPH_HIP(hipEventCreate(&e0));
  TPH_HIP(hipEventCreate(&e1));
  launch_reduce_nb(ctx, g, bt, nb);  // warm
  TPH_HIP(hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < reps; ++r) launch_reduce_nb(ctx, g, bt, nb);
  TPH_HIP(hipEventRecord(e1, ctx->stream));
  TPH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  TPH_HIP(hipEventElapsedTime(&ms, e0, e1));
  TPH_HIP(hipEventDestroy(e0));
  TPH_HIP(hipEventDestroy(e1));
  *avg_ms_host = (double)ms / reps;
  return 0;
}

// ---- measurement aids: the box's own ceilings, timed in the same process as the kernels they are compared with --------
// mode 0: streaming READ (16-B non-temporal loads, per-block contiguous segments, the access pattern of the reduction
// above; every lane folds its values into one double so that the loads cannot be dropped); mode 1: streaming COPY
// (16-B loads + 16-B non-temporal stores).  Bytes moved per launch: n * 8 (read) or 2 * n * 8 (copy).
template <int MODE>
__global__ void __launch_bounds__(TPH_RED_THREADS) k_membw(const double* __restrict__ src, double* __restrict__ dst, int64_t n,
                                                          double* __restrict__ sink) {
  constexpr int U = 8;
  const int64_t n2 = n >> 1;
  const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src);
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d* __restrict__ d2 = reinterpret_cast<v2d*>(dst);
  constexpr int64_t STEP = (int64_t)TPH_RED_THREADS * U;
  int64_t per = (n2 + gridDim.x - 1) / gridDim.x;
  per = (per + STEP - 1) / STEP * STEP;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += STEP) {
    double2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t j = i + (int64_t)k * TPH_RED_THREADS;
      v[k] = j < hi ? nt_load2(s2 + j) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t j = i + (int64_t)k * TPH_RED_THREADS;
      if (MODE == 1) {
        if (j < hi) { v2d o; o.x = v[k].x; o.y = v[k].y; __builtin_nontemporal_store(o, d2 + j); }
      } else {
        acc += v[k].x + v[k].y;
      }
    }
  }
  if (MODE == 0 && acc == 1.2345e300) sink[0] = acc;     // never true: keeps the loads alive
}
extern "C" int tph_bench_membw_time(tph_ctx* ctx, int mode, int64_t n_doubles, int reps, double* avg_ms_host) {
  TPH_REQUIRE(ctx && avg_ms_host && reps > 0 && n_doubles >= 1024 && (mode == 0 || mode == 1), "tph_bench_membw_time: bad argument");
  const size_t bytes = sizeof(double) * (size_t)n_doubles;
  double *a = nullptr, *b = nullptr;
  TPH_HIP(hipMalloc((void**)&a, bytes));
  if (mode == 1 && hipMalloc((void**)&b, bytes) != hipSuccess) { (void)hipFree(a); TPH_REQUIRE(false, "tph_bench_membw_time: out of memory"); }
  TPH_HIP(hipMemsetAsync(a, 0, bytes, ctx->stream));
  const int grid = tph_grid_for(n_doubles, TPH_RED_THREADS, 16, 1024);
  hipEvent_t e0, e1;
  TPH_HIP(hipEventCreate(&e0));
  TPH_HIP(hipEventCreate(&e1));
  for (int r = -2; r < reps; ++r) {
    if (r == 0) TPH_HIP(hipEventRecord(e0, ctx->stream));
    if (mode == 0) hipLaunchKernelGGL(k_membw<0>, dim3(grid), dim3(TPH_RED_THREADS), 0, ctx->stream, a, b, n_doubles, ctx->small_dev);
    else hipLaunchKernelGGL(k_membw<1>, dim3(grid), dim3(TPH_RED_THREADS), 0, ctx->stream, a, b, n_doubles, ctx->small_dev);
  }
  TPH_HIP(hipEventRecord(e1, ctx->stream));
  TPH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  TPH_HIP(hipEventElapsedTime(&ms, e0, e1));
  TPH_HIP(hipEventDestroy(e0));
  TPH_HIP(hipEventDestroy(e1));
  (void)hipFree(a);
  if (b) (void)hipFree(b);
  *avg_ms_host = (double)ms / reps;
  return 0;
}

// FP64 vector-FMA rate of the box under load (8 independent chains per lane, 4 waves per SIMD): the proposal kernels are
// bound by FP64 VALU issue, and the clock a box holds for such code varies from device to device (MI355X_MICROARCH.md,
// "DVFS give-back"), which is the first thing to look at when two boxes disagree on a VALU-bound figure.
__global__ void __launch_bounds__(256) k_fp64_probe(double* __restrict__ out, double x0, int iters) {
  double a0 = x0, a1 = x0 + 1, a2 = x0 + 2, a3 = x0 + 3, a4 = x0 + 4, a5 = x0 + 5, a6 = x0 + 6, a7 = x0 + 7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      a0 = fma(a0, x0, 1.0); a1 = fma(a1, x0, 1.0); a2 = fma(a2, x0, 1.0); a3 = fma(a3, x0, 1.0);
      a4 = fma(a4, x0, 1.0); a5 = fma(a5, x0, 1.0); a6 = fma(a6, x0, 1.0); a7 = fma(a7, x0, 1.0);
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
extern "C" int tph_bench_fp64_time(tph_ctx* ctx, int reps, double* tflops_host) {
  TPH_REQUIRE(ctx && tflops_host && reps > 0, "tph_bench_fp64_time: bad argument");
  const int blocks = ctx->n_simd;                 // 4 waves per SIMD
  const int iters = 256;
  if (tph_scratch_reserve(ctx, sizeof(double) * (size_t)blocks * 256)) return -1;
  hipEvent_t e0, e1;
  TPH_HIP(hipEventCreate(&e0));
  TPH_HIP(hipEventCreate(&e1));
  for (int r = -2; r < reps; ++r) {
    if (r == 0) TPH_HIP(hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(k_fp64_probe, dim3(blocks), dim3(256), 0, ctx->stream, (double*)ctx->scratch, 0.999, iters);
  }
  TPH_HIP(hipEventRecord(e1, ctx->stream));
  TPH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  TPH_HIP(hipEventElapsedTime(&ms, e0, e1));
  TPH_HIP(hipEventDestroy(e0));
  TPH_HIP(hipEventDestroy(e1));
  const double flop = 2.0 * 128.0 * iters * 256.0 * blocks;
  *tflops_host = flop / ((double)ms / reps * 1e-3) / 1e12;
  return 0;
}

// K3: normalised weights, 24 B per historical particle
__global__ void __launch_bounds__(256) k_weights(const double* __restrict__ logl, const double* __restrict__ cmix,
                                                 int64_t n, double beta, double vmax, double s1,
                                                 double* __restrict__ w) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride)
    w[s] = exp(beta * logl[s] - cmix[s] - vmax) / s1;
}

extern "C" int tph_weights(tph_ctx* ctx, double beta, double vmax, double s1, double* w_dev) {
  TPH_REQUIRE(ctx && w_dev, "tph_weights: NULL argument");
  TPH_REQUIRE(ctx->size > 0, "tph_weights: empty history");
  hipLaunchKernelGGL(k_weights, dim3(tph_grid_for(ctx->size, 256, 2)), dim3(256), 0, ctx->stream, ctx->logl, ctx->cmix,
                     ctx->size, beta, vmax, s1, w_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) k_logw(const double* __restrict__ logl, const double* __restrict__ cmix, int64_t n,
                                              double beta, double lognh, double* __restrict__ out) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride)
    out[s] = beta * logl[s] - (cmix[s] - lognh);
}

extern "C" int tph_logw(tph_ctx* ctx, double beta, int64_t n_h_global, double* logw_dev) {
  TPH_REQUIRE(ctx && logw_dev, "tph_logw: NULL argument");
  TPH_REQUIRE(ctx->size > 0, "tph_logw: empty history");
  hipLaunchKernelGGL(k_logw, dim3(tph_grid_for(ctx->size, 256, 2)), dim3(256), 0, ctx->stream, ctx->logl, ctx->cmix,
                     ctx->size, beta, log((double)n_h_global), logw_dev);
  TPH_LAUNCH_CHECK();
  return 0;
}

// (sum w, sum w^2, max w)
__global__ void __launch_bounds__(256) k_sum_sq_max(const double* __restrict__ w, int64_t n, double* __restrict__ partials) {
  double a = 0.0, b = 0.0, m = -DBL_MAX;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
    double v = w[s];
    a += v; b += v * v; m = fmax(m, v);
  }
  __shared__ double sh[4];
  a = tph_block_sum(a, sh);
  b = tph_block_sum(b, sh);
  m = tph_block_max(m, sh);
  if (threadIdx.x == 0) { partials[blockIdx.x * 3] = a; partials[blockIdx.x * 3 + 1] = b; partials[blockIdx.x * 3 + 2] = m; }
}
__global__ void __launch_bounds__(256) k_sum_sq_max_final(const double* __restrict__ partials, int nblocks, double* out) {
  double a = 0.0, b = 0.0, m = -DBL_MAX;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
    a += partials[i * 3]; b += partials[i * 3 + 1]; m = fmax(m, partials[i * 3 + 2]);
  }
  __shared__ double sh[4];
  a = tph_block_sum(a, sh);
  b = tph_block_sum(b, sh);
  m = tph_block_max(m, sh);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = m; }
}

extern "C" int tph_sum_sq_max(tph_ctx* ctx, const double* w_dev, int64_t n, double* out_host) {
  TPH_REQUIRE(ctx && w_dev && out_host && n > 0, "tph_sum_sq_max: bad argument");
  int grid = tph_grid_for(n, 256, 4);
  hipLaunchKernelGGL(k_sum_sq_max, dim3(grid), dim3(256), 0, ctx->stream, w_dev, n, ctx->partials);
  hipLaunchKernelGGL(k_sum_sq_max_final, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, ctx->small_dev);
  TPH_LAUNCH_CHECK();
  TPH_HIP(hipMemcpyAsync(ctx->pinned, ctx->small_dev, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
  TPH_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3; ++i) out_host[i] = ctx->pinned[i];
  return 0;
}

// (tph_warmup: the first launch of a kernel of this translation unit loads its code object; an empty launch pre-pays that)
void tph_warm_reweight(hipStream_t stream) { hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, (unsigned int*)nullptr, 0); }
