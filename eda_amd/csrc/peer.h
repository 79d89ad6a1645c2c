// peer.h -- device-side exchange of BatchNorm column sums between the ranks of one node through peer-mapped memory
// (csrc/peer.hip owns the setup).  The reference trains with SyncBatchNorm (main_utils.py:336-338): every BatchNorm layer
// needs the GLOBAL sums of 2 C numbers per direction.  A collective per layer (RCCL: 136 of them per step) cannot live in a
// captured step on this stack and costs a launch each; here the exchange is a few 8-byte stores and
// polls INSIDE the kernel that has the local sums in registers.
//
// Every rank owns one slab (FINE-GRAINED device memory -- it is polled and written by other GPUs while kernels run, csrc/peer.hip
// alloc_slab -- mapped into every peer process with hipIpc*); slab words are u64:
//   [0] seq   number of exchanging launches this rank has completed (every rank runs the same launches in the same order)
//   [1] done  arrival ticket of the current launch's workgroups (the last one bumps seq)
//   [2] timeouts (sticky count of bounded spins that gave up: results are then garbage, never a hang)
//   [8 ...]  granules: 4 words {a, b, tag, -} at ((parity * PEER_MAXG + g) * PEER_MAXW + source rank) * 4
// A launch with sequence number s exchanges granule g (one channel's pair of sums): the owner thread writes {a, b} into
// EVERY rank's slab (its own included) with system-scope write-through stores, waits for them (vmcnt), then writes
// tag = s + 1; it polls the tags of all source ranks in ITS OWN slab, reads their pairs and adds them in RANK ORDER -- every
// rank gets the same bits.  Two parities suffice: a rank can only be one exchanging launch ahead of the slowest reader of
// its data (the next launch needs that reader's contribution, which is issued behind its read in stream order).
#pragma once
#include <hip/hip_runtime.h>

constexpr int PEER_MAXW = 8;          // ranks of one node
constexpr int PEER_MAXG = 8192;       // granules (channels) per launch
constexpr int PEER_HDR = 8;           // header words
constexpr size_t PEER_SLAB_WORDS = PEER_HDR + (size_t)2 * PEER_MAXG * PEER_MAXW * 4;
// Polls per tag before an exchange gives up: EdaPeer::spin_limit (default 2^24, ~25 s: host-side skew between the ranks --
// one still capturing its graph, a data-loader hiccup -- must not look like a dead peer; EDA_PEER_SPIN_LOG2 sets it).  After
// the FIRST give-up of a process every later exchange skips its spin (the timeout word is sticky): a dead peer costs one
// timeout, not one per exchange.

struct EdaPeer {
  unsigned long long *slab[PEER_MAXW];     // slab[r]: rank r's slab as mapped in this process
  int rank, world;
  unsigned spin_limit;
};

__device__ __forceinline__ unsigned long long eda_peer_seq(const EdaPeer &P) {
  return __hip_atomic_load(P.slab[P.rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one thread, one granule, first half: this rank's pair into every rank's slab, then the tag
__device__ __forceinline__ void eda_peer_publish2(const EdaPeer &P, unsigned long long seq, int g, double a, double b) {
  const unsigned long long tag = seq + 1;
  const size_t base = PEER_HDR + (((seq & 1) * PEER_MAXG + (size_t)g) * PEER_MAXW) * 4;
  const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  for (int r = 0; r < P.world; ++r) {
    unsigned long long *dst = P.slab[r] + base + (size_t)P.rank * 4;
    __hip_atomic_store(dst, ua, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // the pairs have left before any tag does: the wait retires this lane's write-through stores, and the tag is a system-scope
  // RELEASE store (over xGMI a store is posted: the release is what orders the tag behind the pair at the peer; on one device
  // it costs one cache write-back instruction on one lane)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int r = 0; r < P.world; ++r)
    __hip_atomic_store(P.slab[r] + base + (size_t)P.rank * 4 + 2, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// second half: wait for every source rank's tag in the OWN slab, add the pairs in rank order
__device__ __forceinline__ void eda_peer_poll2(const EdaPeer &P, unsigned long long seq, int g, double &a, double &b) {
  const unsigned long long tag = seq + 1;
  const size_t base = PEER_HDR + (((seq & 1) * PEER_MAXG + (size_t)g) * PEER_MAXW) * 4;
  double sa = 0.0, sb = 0.0;
  unsigned long long *mine = P.slab[P.rank];
  const unsigned limit = __hip_atomic_load(mine + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull ? 0u : P.spin_limit;
  for (int q = 0; q < P.world; ++q) {
    unsigned long long *src = mine + base + (size_t)q * 4;
    unsigned spins = 0;
    while (__hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
      if (++spins > limit) {                                  // bounded: count it, carry on with what is there
        __hip_atomic_fetch_add(mine + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");             // (system scope: pairs with the tag's release)
    const unsigned long long va = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long vb = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    sa += __builtin_bit_cast(double, va);
    sb += __builtin_bit_cast(double, vb);
  }
  a = sa; b = sb;
}

// one thread, one granule: (a, b) <- sums over the ranks, rank order
__device__ __forceinline__ void eda_peer_exchange2(const EdaPeer &P, unsigned long long seq, int g, double &a, double &b) {
  eda_peer_publish2(P, seq, g, a, b);
  eda_peer_poll2(P, seq, g, a, b);
}

// every workgroup of an exchanging launch, once, after its last exchange (one thread): the last arriver bumps seq
__device__ __forceinline__ void eda_peer_done(const EdaPeer &P, unsigned nwg) {
  unsigned long long *mine = P.slab[P.rank];
  const unsigned long long t = __hip_atomic_fetch_add(mine + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t == (unsigned long long)nwg - 1) {
    __hip_atomic_store(mine + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(mine, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// host side (peer.hip): the connected peer set, or nullptr when the native exchange is not set up
const EdaPeer *eda_peer_active();
