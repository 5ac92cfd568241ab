// Peer-to-peer small-message exchange (p2p.hip): argument block and the block-level device routine, shared with kernels that
// fold an exchange into their own work (k_adapt: the all-reduce of the step's acceptance sums).
#pragma once
#include "common.h"

constexpr int TPH_P2P_MAX = 16;                 // ranks of one node
constexpr size_t TPH_P2P_SLOT = 32768;          // bytes per message and source
constexpr size_t TPH_P2P_FLAG = 128;

struct p2p_args {
  char* inbox[TPH_P2P_MAX];
  int world, rank;
  unsigned long long* seq;                      // device: [0] exchanges completed by this rank, [1] != 0 after a timed-out one
  unsigned int* err;                            // pinned host word (device address): != 0 after a timed-out exchange
  unsigned long long timeout;                   // wall_clock64 ticks (100 MHz)
};

// ready argument block of the ctx, or NULL (no peer-to-peer exchange attached / message too large); also checks the error word
const p2p_args* tph_p2p_ready(tph_ctx* ctx, int64_t count, int dtype);

#if defined(__HIPCC__)
template <typename T> __device__ __forceinline__ T p2p_op(T a, T b, int op) {
  return op == TPH_OP_SUM ? a + b : (op == TPH_OP_MAX ? (a > b ? a : b) : (a < b ? a : b));
}

// One exchange by ONE thread block (any size >= world, all threads call it): op < 0 all-gather (dst = world x count, rank
// order), op >= 0 all-reduce into dst[count]; src may alias dst.  Returns false after a timed-out wait (error word set).
template <typename T>
__device__ __forceinline__ bool p2p_block_exchange(const p2p_args& a, const T* src, T* dst, int count, int op) {
  __shared__ unsigned long long s_seq;
  __shared__ int s_bad;
  const int tid = threadIdx.x, nt = blockDim.x;
  __syncthreads();                              // src may have been written by other threads of the block
  if (tid == 0) { s_seq = *a.seq + 1ull; s_bad = a.seq[1] != 0ull; }
  __syncthreads();
  if (s_bad) return false;                      // an earlier exchange timed out: fail at once instead of waiting again
  const unsigned long long seq = s_seq;
  const size_t e = (size_t)(seq & 1ull);
  const size_t flags = 2 * (size_t)a.world * TPH_P2P_SLOT;
  {
    const size_t mine = (e * a.world + a.rank) * TPH_P2P_SLOT;
    for (int i = tid; i < count; i += nt) {
      const T v = src[i];
      for (int p = 0; p < a.world; ++p)
        __hip_atomic_store((T*)(a.inbox[p] + mine) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < a.world) {
    unsigned long long* raise = (unsigned long long*)(a.inbox[tid] + flags + (e * a.world + a.rank) * TPH_P2P_FLAG);
    __hip_atomic_store(raise, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* wait = (const unsigned long long*)(a.inbox[a.rank] + flags + (e * a.world + tid) * TPH_P2P_FLAG);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(wait, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > a.timeout) { s_bad = tid + 1; break; }      // every wave reaches the end of the kernel
    }
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) { *a.err = (unsigned int)s_bad; a.seq[1] = 1ull; __threadfence_system(); *a.seq = seq; }
    return false;
  }
  const char* in = a.inbox[a.rank] + e * a.world * TPH_P2P_SLOT;
  if (op < 0) {
    for (int k = tid; k < a.world * count; k += nt) {
      const int s = k / count, i = k - s * count;
      dst[k] = __hip_atomic_load((const T*)(in + (size_t)s * TPH_P2P_SLOT) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  } else {
    for (int i = tid; i < count; i += nt) {
      T acc = __hip_atomic_load((const T*)in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      for (int s = 1; s < a.world; ++s)
        acc = p2p_op(acc, __hip_atomic_load((const T*)(in + (size_t)s * TPH_P2P_SLOT) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), op);
      dst[i] = acc;
    }
  }
  if (tid == 0) *a.seq = seq;
  __syncthreads();
  return true;
}
#endif
