// fps.hip -- furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel<bs> (reference
// pointnet2/_ext_src/src/sampling_gpu.cu:74-178) and its host wrapper
// (sampling.cpp:70-91).  Same results bit for bit (including the origin-ball
// skip and the shared-memory-tree tie rule), different machine mapping:
//
//  * the reference runs ONE 512-thread block per scene and re-reads xyz and the
//    global `temp` array every round.  Here a scene is owned by a CLUSTER of G
//    workgroups (T = 512/1024 threads each); every thread keeps its P points
//    (x, y, z, running min distance) in registers for the whole kernel, so a
//    round touches no global memory except one 40-byte mailbox record per
//    workgroup.
//  * per round: register update -> DPP wave max -> one 64-bit LDS atomic max per
//    wave (distance bits | inverted tie key) -> ONE barrier -> (cluster only) all-gather of the G workgroup
//    candidates through tagged 8-byte granules (agent-scope relaxed atomics,
//    double-buffered by round parity; cdna_hip_programming.md G16 recipe R2)
//    -> every workgroup picks the same winner and broadcasts it through LDS.
//  * a scene with n <= 8192 points is handled by a single workgroup (G = 1) and
//    skips the mailbox step entirely.
//
// Tie rule.  The reference's winner among equal maxima is the minimum over
// (bitreverse_p(k mod bs), k) with bs = 2^p = opt_n_threads(n) (its thread
// tid = k mod bs keeps its lowest k on strict '>', and the LDS tree with strides
// bs/2..1 keeps slot idx1 on ties).  Because k mod bs are the low p bits of k,
// that order equals the order of   tiekey(k) = rev_p(k & (bs-1)) << (31-p) | k >> p,
// which is what the arg-max below minimises among equal distances.
// A point inside the origin ball (mag <= 1e-3) is never updated nor selectable
// in the reference; here it carries the constant candidate (-1.0f, k = 0), which
// is exactly what a reference thread without a valid point contributes.
#include "eda_common.h"
#include "fps_bucket.h"

#include <limits.h>
#include <stdlib.h>

namespace {

constexpr int kRecWords = 8;          // u64 words per mailbox record (5 used, 64-byte record)
int g_fps_background = -1;            // -1: from EDA_FPS_BACKGROUND (eda_fps_set_background overrides)
int fps_background() { return g_fps_background < 0 ? (int)(eda_knob(EDA_K_FPS_BACKGROUND) != 0) : g_fps_background; }
int g_fps_cu_reserve = INT_MIN;       // INT_MIN: from EDA_FPS_CU_RESERVE (eda_fps_set_cu_reserve overrides)
int fps_cu_reserve() { return g_fps_cu_reserve == INT_MIN ? (int)eda_knob(EDA_K_FPS_CU_RESERVE) : g_fps_cu_reserve; }
constexpr int kMaxG = 64;             // workgroups per scene (one poll lane each)
constexpr size_t kStatusBytes = 256;  // status words in front of the mailboxes
constexpr unsigned kSpinLimit = 1u << 20;   // ~1 s of polling, then give up (per-call flag set, indices zero-filled)
// Status block (64 ints in front of the mailboxes).  int 0: STICKY, some call gave up and nothing recovered it (only
// scenes the bucket sampler cannot take: > 65536 points); int 2: STICKY count of give-ups the bucket sampler recovered;
// int 3: duration of the last launch (10 ns ticks); ints 4..59: diagnostics; int 60 (kFailInt): give-up flag of the
// CURRENT call (zeroed per call with ints 4..63, set by the cluster kernels, consumed by the launches behind them).
constexpr int kFailInt = 60;
int g_fps_policy = EDA_FPS_AUTO;             // EDA_FPS_AUTO / CLUSTER / BUCKET (include/eda_hip.h); EDA_FPS_BUCKET=0|1 overrides

// a give-up that nothing can recover becomes sticky
__global__ void fps_fail_latch_kernel(int *status) {
  if (status[kFailInt] != 0) status[0] = 1;
}
// test hook (EDA_FPS_TEST_GIVEUP=1): what a cluster launch that was not co-resident leaves behind -- the call's give-up
// flag set, zero indices -- without waiting for the spin limit
__global__ void fps_fake_giveup_kernel(int *status, int *idx, long count) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) idx[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) status[kFailInt] = 1;
}

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

__device__ __forceinline__ unsigned fps_tiekey(unsigned k, int p) {
  const unsigned lowmask = (1u << p) - 1u;
  const unsigned hi = p ? (__brev(k & lowmask) >> (32 - p)) : 0u;
  return (hi << (31 - p)) | (k >> p);
}

// Scope note (measured, round 1): all workgroups of a scene can be placed on ONE XCD, whose L2
// would then be a sufficient coherence point -- but gfx950 offers no scope that bypasses the
// per-CU L1 while still hitting the local L2: workgroup scope (sc0) may hit L1 outside
// threadgroup-split mode (polls read stale lines for ~100 us; `buffer_inv sc0` does not help),
// and agent scope (sc1) is coherent across the eight L2s through the fabric, the ~1 us
// hand-off these granules pay.
__device__ __forceinline__ void granule_store(u64 *p, unsigned tag, unsigned value) {
  __hip_atomic_store(p, ((u64)tag << 32) | (u64)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 granule_load(const u64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Inverse of fps_tiekey.
__device__ __forceinline__ unsigned fps_untie(unsigned tk, int p) {
  const unsigned hi = p ? (tk >> (31 - p)) : 0u;
  const unsigned low = p ? (__brev(hi) >> (32 - p)) : 0u;
  const unsigned mask = (1u << (31 - p)) - 1u;
  return ((tk & mask) << p) | low;
}

// LDS layout (dynamic): float lx[T*P], ly[T*P], lz[T*P]; then scratch.
struct FpsShared {
  // Workgroup arg-max of a round: 64-bit key = (order-preserving distance bits << 32) |
  // ~tiekey, merged with ONE LDS atomic max per wave.  Three slots, used round-robin:
  // slot[(r+1)%3] is cleared during round r, when its last readers (round r-2) are
  // provably past (they are separated from this point by the barrier of round r-1).
  u64 slot[3];
  float next_xyz[4];   // cluster only: x, y, z of the point chosen this round (+ chosen index bits)
  int fail;
};

template <int MODE, int T, int P, bool CLUSTER>
__global__ __launch_bounds__(T) void fps_kernel(const float *__restrict__ xyz_all, int n, int m,
                                                int *__restrict__ idx_all, int S, int G, int p_log2,
                                                u64 *mail_all, int *status, const int *__restrict__ only_if) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // only_if (single-workgroup scenes): run the scene only if only_if[scene] != 0 -- the prefix entry point
  // below has already PROVEN the answer 0..m-1 for the other scenes and written it
  if (!CLUSTER && only_if != nullptr && only_if[blockIdx.x % S] == 0) return;
  float *lx = reinterpret_cast<float *>(smem);
  float *ly = lx + T * P;
  float *lz = ly + T * P;
  FpsShared *sh = reinterpret_cast<FpsShared *>(lz + T * P);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int scene = blockIdx.x % S;   // consecutive blocks -> different scenes, so a
  const int w = blockIdx.x / S;       // scene's workgroups share an XCD when S % 8 == 0 (speed only)

  const float *xyz = xyz_all + (size_t)scene * n * 3;
  int *idx = idx_all + (size_t)scene * m;
  u64 *mail = mail_all + (size_t)scene * 2 * kMaxG * kRecWords;

  if (m <= 0) return;

  // ---- load this thread's P points into registers -------------------------
  float px[P], py[P], pz[P], pt[P];
  int pk[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const long long k = ((long long)(j * G + w)) * T + tid;
    float x = 0.f, y = 0.f, z = 0.f;
    bool valid = k < n;
    if (valid) {
      x = xyz[k * 3 + 0];
      y = xyz[k * 3 + 1];
      z = xyz[k * 3 + 2];
      const float mag = eda_sumsq3<MODE>(x, y, z);
      valid = !((double)mag <= 1e-3);          // sampling_gpu.cu:106-107 (double compare)
    }
    px[j] = x; py[j] = y; pz[j] = z;
    pt[j] = valid ? 1e10f : -1.0f;             // sampling.cpp:78-80 / thread init best=-1
    pk[j] = valid ? (int)k : 0;
    lx[j * T + tid] = x; ly[j * T + tid] = y; lz[j * T + tid] = z;
  }
  if (tid == 0) { sh->fail = 0; sh->slot[0] = 0; sh->slot[1] = 0; sh->slot[2] = 0; }

  // first sample is index 0 (sampling_gpu.cu:90-92)
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (w == 0 && tid == 0) idx[0] = 0;
  __syncthreads();

  for (int r = 1; r < m; ++r) {
    // ---- per-thread update + running best (strict '>' keeps the lowest k) --
    int best_bits = __float_as_int(-1.0f);
    int best_k = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const float d = eda_sumsq3<MODE>(px[j] - x1, py[j] - y1, pz[j] - z1);  // point minus centre
      const float t = fminf(d, pt[j]);   // invalid points stay at -1
      pt[j] = t;
      const int tb = __float_as_int(t);  // t >= 0 or == -1: signed int order == float order
      const bool gt = tb > best_bits;
      best_bits = gt ? tb : best_bits;
      best_k = gt ? pk[j] : best_k;
    }
    // ---- workgroup arg-max: DPP wave max, then one 64-bit LDS atomic per wave-max lane
    u64 *slot = &sh->slot[r % 3];
    {
      const int wm = eda_wave_max_i32(best_bits);
      const bool cand = best_bits == wm;
      const u64 ties = __ballot(cand);
      // Exactly ONE lane per wave issues the LDS atomic (hipcc would otherwise serialise
      // all tied lanes -- e.g. 64 padded or already-selected points -- through a readlane loop).
      if (__popcll(ties) == 1) {
        if (cand)
          atomicMax(slot, ((u64)((unsigned)wm ^ 0x80000000u) << 32) |
                              (u64)(~fps_tiekey((unsigned)best_k, p_log2)));
      } else {
        const unsigned tk = cand ? fps_tiekey((unsigned)best_k, p_log2) : 0xFFFFFFFFu;
        const unsigned wt = eda_wave_min_u32(tk);
        if (lane == 0) atomicMax(slot, ((u64)((unsigned)wm ^ 0x80000000u) << 32) | (u64)(~wt));
      }
      if (tid == 0) sh->slot[(r + 1) % 3] = 0;
    }
    __syncthreads();
    if (!CLUSTER) {
      const int kw = (int)fps_untie(~(unsigned)*slot, p_log2);
      x1 = lx[kw]; y1 = ly[kw]; z1 = lz[kw];     // G == 1: local slot index == point index
      if (tid == 0) idx[r] = kw;
      continue;
    }
    if (wave == 0) {       // only wave 0 talks to the other workgroups; the rest wait at the barrier
      const u64 wkey = *slot;
      const int wbits = (int)((unsigned)(wkey >> 32) ^ 0x80000000u);
      const int kw = (int)fps_untie(~(unsigned)wkey, p_log2);
      // coordinates of this workgroup's candidate from the LDS copy
      const int li = ((kw / T) / G) * T + (kw % T);
      const float cx = lx[li], cy = ly[li], cz = lz[li];
      int fk = kw;
      float fx = cx, fy = cy, fz = cz;
      u64 *box = mail + (size_t)(r & 1) * kMaxG * kRecWords;
      if (lane == 0) {
        u64 *rec = box + (size_t)w * kRecWords;
        granule_store(rec + 0, (unsigned)r, (unsigned)wbits);
        granule_store(rec + 1, (unsigned)r, (unsigned)kw);
        granule_store(rec + 2, (unsigned)r, __float_as_uint(cx));
        granule_store(rec + 3, (unsigned)r, __float_as_uint(cy));
        granule_store(rec + 4, (unsigned)r, __float_as_uint(cz));
      }
      // all-gather: lane l < G polls workgroup l's record until its 5 tags match
      unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
      bool ok = lane >= G;
      unsigned spins = 0;
      bool failed = false;
      for (;;) {
        if (!ok) {
          const u64 *rec = box + (size_t)lane * kRecWords;
          const u64 g0 = granule_load(rec + 0), g1 = granule_load(rec + 1);
          const u64 g2 = granule_load(rec + 2), g3 = granule_load(rec + 3);
          const u64 g4 = granule_load(rec + 4);
          v0 = (unsigned)g0; v1 = (unsigned)g1; v2 = (unsigned)g2; v3 = (unsigned)g3; v4 = (unsigned)g4;
          ok = ((unsigned)(g0 >> 32) == (unsigned)r) & ((unsigned)(g1 >> 32) == (unsigned)r) &
               ((unsigned)(g2 >> 32) == (unsigned)r) & ((unsigned)(g3 >> 32) == (unsigned)r) &
               ((unsigned)(g4 >> 32) == (unsigned)r);
        }
        if (__all(ok)) break;
        if (++spins > kSpinLimit) { failed = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (failed) {
        if (lane == 0) { sh->fail = 1; atomicExch(status + kFailInt, 1); }
      } else {
        const int gb = lane < G ? (int)v0 : INT_MIN;
        const int gm = eda_wave_max_i32(gb);
        const unsigned gtk = (lane < G && gb == gm) ? fps_tiekey(v1, p_log2) : 0xFFFFFFFFu;
        const unsigned gt = eda_wave_min_u32(gtk);
        const int gl = __ffsll((long long)__ballot(gtk == gt)) - 1;
        fk = __builtin_amdgcn_readlane((int)v1, gl);
        fx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v2, gl));
        fy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v3, gl));
        fz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v4, gl));
        if (gm < 0) fk = 0;   // every point of the scene is invalid: reference emits index 0
      }
      if (lane == 0) {
        sh->next_xyz[0] = fx; sh->next_xyz[1] = fy; sh->next_xyz[2] = fz;
        if (w == 0) idx[r] = fk;
      }
    }
    __syncthreads();
    if (sh->fail) {
      // give-up (a peer workgroup never answered within the spin limit, i.e. the cluster was not
      // co-resident): the status word is already set; leave VALID indices behind (0) so that the
      // gathers / ball query that follow cannot read out of bounds, then leave.
      if (w == 0) for (int i = r + (int)tid; i < m; i += T) idx[i] = 0;
      return;
    }
    // If the winner is the all-invalid candidate (-1, k = 0) these coordinates are
    // meaningless, but then no point of the scene is ever updated, so they are unused.
    x1 = sh->next_xyz[0]; y1 = sh->next_xyz[1]; z1 = sh->next_xyz[2];
    // next_xyz is rewritten only after the NEXT round's first barrier.
  }
}


// ---------------------------------------------------------------------------------------------
// Speculative cluster kernel: K candidates per hand-off.
//
// The cluster kernel above pays one cross-workgroup hand-off (~1.1 us of its 2.2 us round) per
// sample.  Here every hand-off carries each workgroup's TOP-K candidates under the reference's
// total order (distance bits, then minimum tie key).  From the gathered G*K candidates every
// workgroup derives the global top-K c0 > c1 > ... and accepts a PREFIX of it as the next
// samples, exactly as the sequential algorithm would have chosen them:
//   * c0 is the next sample by definition;
//   * after the update with c0 every point's key can only DROP (d' = min(d, |p-c0|^2), tie key
//     unchanged) and c0's own distance becomes 0.  So if c1's distance is unchanged --
//     |c1-c0|^2 >= d(c1), the same canonical arithmetic the update uses -- and d(c1) > 0, then
//     c1 is still above every other point: it IS the sample after c0.  Inductively c_i is
//     accepted iff all earlier candidates were and |c_i-c_j|^2 >= d(c_i) for every j < i and
//     d(c_i) > 0 (with d = 0 the already-chosen points tie and the tie key decides: stop).
// Furthest points are mutually far apart, so the whole prefix is accepted most of the time: on
// the bench clouds K = 4 needs 592 hand-offs for 2047 samples.  The accepted centres are
// applied in ONE register pass (min over up to K distances), in which each thread also rebuilds
// its sorted top-K list; waves merge by K rounds of (DPP max, pop), workgroups and the cluster
// by K rounds of DPP arg-max over one key per lane (fps_topk_rank; an all-pairs rank through
// LDS broadcasts or v_readlane was measured 1.5-2x slower: 64-bit compares are not cheap).  Results are bit-identical to the sequential
// kernels (tests/test_ops_gpu.py::test_fps_vs_oracle).
constexpr int kSpecMaxG = 16;                    // 64 poll lanes / K=4 candidates each
constexpr int kSpecRecWords = 5 * 4;             // K=4 candidates x (bits, k, x, y, z) granules

template <int K>
struct FpsSpecShared {
  u64 wkey[16][K];        // per wave: its top-K keys (up to 16 waves)
  int gc[K][5];           // global top-K staging: bits, k, x, y, z
  float cand[K][4];       // accepted-prefix candidates: x, y, z, (unused)
  int cand_k[K];
  int nvalid;
  int fail;
};

// Rank (0..K-1) of this lane's key among the K largest keys of the wave, K for all other lanes;
// key 0 = "no entry".  K rounds of wave arg-max (DPP) with the winner retiring; equal keys (only
// the all-invalid candidate) retire in lane order.
template <int K>
__device__ __forceinline__ int fps_topk_rank(u64 key, int lane) {
  int hi = (int)((unsigned)(key >> 32) ^ 0x80000000u);     // signed order == key order; 0 -> INT_MIN
  int lo = (int)((unsigned)key ^ 0x80000000u);
  bool live = key != 0ull;
  int rank = K;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int mh = eda_wave_max_i32(live ? hi : INT_MIN);
    const bool cnd = live && hi == mh;
    const u64 ties = __ballot(cnd);
    int wl = -1;
    if (__popcll(ties) == 1) {
      wl = __ffsll((long long)ties) - 1;
    } else if (ties != 0ull) {
      const int ml = eda_wave_max_i32(cnd ? lo : INT_MIN);
      wl = __ffsll((long long)__ballot(cnd && lo == ml)) - 1;
    }
    if (lane == wl) { rank = i; live = false; }
  }
  return rank;
}

__device__ __forceinline__ u64 fps_key(int bits, unsigned k, int p_log2) {
  return ((u64)((unsigned)bits ^ 0x80000000u) << 32) | (u64)(~fps_tiekey(k, p_log2));
}

template <int MODE, int T, int P, int K>
__global__ __launch_bounds__(T) void fps_spec_kernel(const float *__restrict__ xyz_all, int n, int m,
                                                     int *__restrict__ idx_all, int S, int G, int p_log2,
                                                     u64 *mail_all, int *status, int background) {
  static_assert(K == 4, "record layout and poll-lane mapping are written for K = 4");
  constexpr int NW = T / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *lx = reinterpret_cast<float *>(smem);
  float *ly = lx + T * P;
  float *lz = ly + T * P;
  FpsSpecShared<K> *sh = reinterpret_cast<FpsSpecShared<K> *>(lz + T * P);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scene = blockIdx.x % S;
  const int w = blockIdx.x / S;
  const float *xyz = xyz_all + (size_t)scene * n * 3;
  int *idx = idx_all + (size_t)scene * m;
  u64 *mail = mail_all + (size_t)scene * 2 * kMaxG * kRecWords;     // same footprint as fps_kernel
  if (m <= 0) return;

  float px[P], py[P], pz[P], pt[P];
  int pk[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const long long k = ((long long)(j * G + w)) * T + tid;
    float x = 0.f, y = 0.f, z = 0.f;
    bool valid = k < n;
    if (valid) {
      x = xyz[k * 3 + 0]; y = xyz[k * 3 + 1]; z = xyz[k * 3 + 2];
      const float mag = eda_sumsq3<MODE>(x, y, z);
      valid = !((double)mag <= 1e-3);
    }
    px[j] = x; py[j] = y; pz[j] = z;
    pt[j] = valid ? 1e10f : -1.0f;
    pk[j] = valid ? (int)k : 0;
    lx[j * T + tid] = x; ly[j * T + tid] = y; lz[j * T + tid] = z;
  }
  if (tid == 0) { sh->fail = 0; sh->nvalid = 1; }
  if (tid < 3) sh->cand[0][tid] = xyz[tid];       // first sample is index 0
  if (w == 0 && tid == 0) idx[0] = 0;
  __syncthreads();

  int r = 1;                 // next output position
  unsigned h = 0;            // hand-off counter = mailbox tag
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();     // 100 MHz wall clock (diagnostic below)
  // every round of this kernel is on the step's critical path and the kernel leaves > half of the CUs idle:
  // whatever another stream places next to these waves must not win issue slots from them
  __builtin_amdgcn_s_setprio(3);
#ifdef EDA_FPS_PROFILE
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#define FPS_MARK(i) do { const unsigned long long tn = __builtin_readcyclecounter(); acc[i] += tn - tp; tp = tn; } while (0)
#else
#define FPS_MARK(i) do { } while (0)
#endif
  while (r < m) {
    ++h;
    const int nc = sh->nvalid;
    float cx[K], cy[K], cz[K];
#pragma unroll
    for (int c = 0; c < K; ++c) { cx[c] = sh->cand[c][0]; cy[c] = sh->cand[c][1]; cz[c] = sh->cand[c][2]; }

    // ---- 1. apply the nc accepted centres; rebuild this thread's sorted top-K --------------
    int lb[K], lk[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { lb[i] = INT_MIN; lk[i] = 0; }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      float t = pt[j];
#pragma unroll
      for (int c = 0; c < K; ++c) {
        if (c < nc) {
          const float d = eda_sumsq3<MODE>(px[j] - cx[c], py[j] - cy[c], pz[j] - cz[c]);
          t = fminf(d, t);                 // invalid points stay at -1
        }
      }
      pt[j] = t;
      int vb = __float_as_int(t), vk = pk[j];   // t >= 0 or == -1: signed int order == float order
      // insertion with strict '>' (an equal earlier entry, i.e. a lower k of this thread, stays ahead)
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const bool gt = vb > lb[i];
        const int ob = lb[i], ok = lk[i];
        lb[i] = gt ? vb : ob; lk[i] = gt ? vk : ok;
        vb = gt ? ob : vb;   vk = gt ? ok : vk;
      }
    }

    FPS_MARK(0);
    // ---- 2. wave top-K: K rounds of (wave arg-max of the list heads, winner pops) -----------
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int hb = lb[0];
      const int wm = eda_wave_max_i32(hb);
      const bool cnd = hb == wm;
      const u64 ties = __ballot(cnd);
      int wl;
      if (__popcll(ties) == 1) {
        wl = __ffsll((long long)ties) - 1;
      } else {
        const unsigned tk = cnd ? fps_tiekey((unsigned)lk[0], p_log2) : 0xFFFFFFFFu;
        const unsigned wt = eda_wave_min_u32(tk);
        wl = __ffsll((long long)__ballot(cnd && tk == wt)) - 1;
      }
      const int wk = __builtin_amdgcn_readlane(lk[0], wl);
      if (lane == 0) sh->wkey[wave][i] = wm == INT_MIN ? 0ull : fps_key(wm, (unsigned)wk, p_log2);
      if (lane == wl) {
#pragma unroll
        for (int q = 0; q + 1 < K; ++q) { lb[q] = lb[q + 1]; lk[q] = lk[q + 1]; }
        lb[K - 1] = INT_MIN; lk[K - 1] = 0;
      }
    }
    FPS_MARK(1);
    __syncthreads();
    FPS_MARK(2);

    if (wave == 0) {
      // ---- 3. workgroup top-K of the NW*K wave keys (one per lane); publish -------------------
      u64 *box = mail + (size_t)(h & 1) * kMaxG * kRecWords;
      {
        const u64 *flat = &sh->wkey[0][0];
        const bool have = lane < NW * K;
        const u64 mine = have ? flat[lane] : 0ull;
        const int rank = fps_topk_rank<K>(mine, lane);
        if (have && rank < K) {
          const int bits = (int)((unsigned)(mine >> 32) ^ 0x80000000u);
          int kw = 0;
          float qx = 0.f, qy = 0.f, qz = 0.f;
          if (mine != 0ull) {
            kw = (int)fps_untie(~(unsigned)mine, p_log2);
            const int li = ((kw / T) / G) * T + (kw % T);
            qx = lx[li]; qy = ly[li]; qz = lz[li];
          }
          u64 *rec = box + (size_t)w * kSpecRecWords + (size_t)rank * 5;
          granule_store(rec + 0, h, mine != 0ull ? (unsigned)bits : (unsigned)INT_MIN);
          granule_store(rec + 1, h, (unsigned)kw);
          granule_store(rec + 2, h, __float_as_uint(qx));
          granule_store(rec + 3, h, __float_as_uint(qy));
          granule_store(rec + 4, h, __float_as_uint(qz));
        }
      }
      FPS_MARK(3);
      // ---- 4. all-gather: lane l < G*K polls candidate l%K of workgroup l/K ------------------
      unsigned v0 = (unsigned)INT_MIN, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
      const bool poller = lane < G * K;
      bool ok = !poller;
      unsigned spins = 0;
      bool failed = false;
      for (;;) {
        if (!ok) {
          const u64 *rec = box + (size_t)(lane / K) * kSpecRecWords + (size_t)(lane % K) * 5;
          if (background) {
            // the sampler as a PREFETCH underneath other work (eda_fps_set_background): poll ONE granule, the record's
            // last-written one, and fetch the other four only once its tag is there -- a fifth of the polling traffic.  104
            // resident workgroups polling five granules each took enough HBM / fabric bandwidth from the other stream's
            // kernels to cost the step 0.2 ms (profiles/r05_side_stream_interference.md); the second, dependent fetch
            // costs THIS kernel ~0.1 us per round (3.07 -> 3.29 ms), which is why the in-step form keeps the wide poll.
            const u64 g4 = granule_load(rec + 4);
            if ((unsigned)(g4 >> 32) == h) {
              const u64 g0 = granule_load(rec + 0), g1 = granule_load(rec + 1);
              const u64 g2 = granule_load(rec + 2), g3 = granule_load(rec + 3);
              v0 = (unsigned)g0; v1 = (unsigned)g1; v2 = (unsigned)g2; v3 = (unsigned)g3; v4 = (unsigned)g4;
              ok = ((unsigned)(g0 >> 32) == h) & ((unsigned)(g1 >> 32) == h) & ((unsigned)(g2 >> 32) == h) &
                   ((unsigned)(g3 >> 32) == h);
            }
          } else {
            const u64 g0 = granule_load(rec + 0), g1 = granule_load(rec + 1);
            const u64 g2 = granule_load(rec + 2), g3 = granule_load(rec + 3);
            const u64 g4 = granule_load(rec + 4);
            v0 = (unsigned)g0; v1 = (unsigned)g1; v2 = (unsigned)g2; v3 = (unsigned)g3; v4 = (unsigned)g4;
            ok = ((unsigned)(g0 >> 32) == h) & ((unsigned)(g1 >> 32) == h) & ((unsigned)(g2 >> 32) == h) &
                 ((unsigned)(g3 >> 32) == h) & ((unsigned)(g4 >> 32) == h);
          }
        }
        if (__all(ok)) break;
        if (++spins > kSpinLimit) { failed = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      FPS_MARK(4);
      if (failed) {
        if (lane == 0) { sh->fail = 1; atomicExch(status + kFailInt, 1); }
      } else {
        // ---- 5. global top-K of the G*K gathered keys (one per lane) --------------------------
        const u64 mine = (poller && (int)v0 != INT_MIN) ? fps_key((int)v0, v1, p_log2) : 0ull;
        if (lane < K) sh->gc[lane][0] = __float_as_int(-1.0f);        // "no such candidate"
        const int rank = fps_topk_rank<K>(mine, lane);
        if (poller && rank < K) {       // candidate `rank` of the global order
          sh->gc[rank][0] = (int)v0 == INT_MIN ? __float_as_int(-1.0f) : (int)v0;
          sh->gc[rank][1] = (int)v1; sh->gc[rank][2] = (int)v2; sh->gc[rank][3] = (int)v3; sh->gc[rank][4] = (int)v4;
        }
        int ck[K]; float qx[K], qy[K], qz[K], qd[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {   // LDS broadcasts; a wave's LDS operations execute in order
          qd[i] = __int_as_float(sh->gc[i][0]);
          ck[i] = sh->gc[i][1];
          qx[i] = __int_as_float(sh->gc[i][2]); qy[i] = __int_as_float(sh->gc[i][3]); qz[i] = __int_as_float(sh->gc[i][4]);
        }
        // ---- 6. accept the prefix the sequential algorithm would produce ---------------------
        int nv = 1;
        bool chain = true;
#pragma unroll
        for (int i = 1; i < K; ++i) {
          bool good = chain && qd[i] > 0.f;
#pragma unroll
          for (int j = 0; j < i; ++j) {
            const float d = eda_sumsq3<MODE>(qx[i] - qx[j], qy[i] - qy[j], qz[i] - qz[j]);
            good = good && (d >= qd[i]);
          }
          chain = good;
          nv += good ? 1 : 0;
        }
        if (nv > m - r) nv = m - r;
        if (qd[0] < 0.f) ck[0] = 0;          // every point invalid: the reference emits index 0
        if (lane == 0) {
          sh->nvalid = nv;
#pragma unroll
          for (int i = 0; i < K; ++i) {
            sh->cand[i][0] = qx[i]; sh->cand[i][1] = qy[i]; sh->cand[i][2] = qz[i];
            if (w == 0 && i < nv) idx[r + i] = ck[i];
          }
        }
      }
    }
    FPS_MARK(5);
    __syncthreads();
    FPS_MARK(6);
    if (sh->fail) {
      // give-up (a peer workgroup never answered within the spin limit, i.e. the cluster was not
      // co-resident): the status word is already set; leave VALID indices behind (0) so that the
      // gathers / ball query that follow cannot read out of bounds, then leave.
      if (w == 0) for (int i = r + (int)tid; i < m; i += T) idx[i] = 0;
      return;
    }
    r += sh->nvalid;
    // sh->cand / nvalid are rewritten only after the next hand-off's first barrier
  }
#ifdef EDA_FPS_PROFILE
  if (w == 0 && tid == 0 && scene == 0)
    for (int i = 0; i < 7; ++i) status[16 + i] = (int)(acc[i] >> 4);
#endif
  // diagnostic: hand-offs this scene needed (ints 4.. of the status block; tools/fps_handoffs.py)
  if (w == 0 && tid == 0 && scene < 56) status[4 + scene] = (int)h;
  // and the wall time of scene 0's first workgroup in 10 ns ticks (int 3): what the sampler takes INSIDE a
  // replayed graph, next to whatever runs on other streams (bench.py reports it)
  if (w == 0 && tid == 0 && scene == 0) status[3] = (int)(__builtin_amdgcn_s_memrealtime() - t_begin);
}

template <int MODE, int T, int P>
int launch_fps_spec(const float *xyz, int n, int m, int *idx, int S, int G, int p_log2, u64 *mail,
                    int *status, hipStream_t stream) {
  const size_t lds = (size_t)T * P * 3 * sizeof(float) + sizeof(FpsSpecShared<4>);
  auto kern = fps_spec_kernel<MODE, T, P, 4>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      eda_set_error("fps: cannot raise dynamic LDS to %zu bytes: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(kern, dim3(S * G), dim3(T), lds, stream, xyz, n, m, idx, S, G, p_log2, mail, status, fps_background());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    eda_set_error("fps: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

thread_local const int *tl_fps_only_if = nullptr;     // set by eda_furthest_point_sampling_prefix_f32 around its fallback launch

template <int MODE, int T, int P, bool CLUSTER>
int launch_fps(const float *xyz, int n, int m, int *idx, int S, int G, int p_log2, u64 *mail,
               int *status, hipStream_t stream) {
  const size_t lds = (size_t)T * P * 3 * sizeof(float) + sizeof(FpsShared);
  auto kern = fps_kernel<MODE, T, P, CLUSTER>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      eda_set_error("fps: cannot raise dynamic LDS to %zu bytes: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(kern, dim3(S * G), dim3(T), lds, stream, xyz, n, m, idx, S, G, p_log2, mail,
                     status, tl_fps_only_if);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    eda_set_error("fps: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// ---- "is 0..m-1 the answer?" -----------------------------------------------------------------------------
// The backbone feeds SA2..SA4 with the PREVIOUS level's samples in sampling order
// (models/backbone_module.py:122-131 notes that FPS of such a set returns its prefix 0..m-1).  That
// holds exactly when no tie of the reference's arg-max order interferes (duplicated points do interfere),
// so it cannot be assumed -- but it can be CHECKED without the 1023 dependent rounds that make the sampler
// slow: if the answer is 0..m-1 then the centre of round r is point r-1, known in advance, and every
// point k can run its own min-distance recurrence independently (no per-round arg-max, no barrier):
//   sel[r]  = min_{i<r} d(p_r, p_i)                                  (pass 1, thread per sample)
//   point k, round r:  t = min_{i<r} d(p_k, p_i)  must lose against (sel[r], r) under the reference's
//   total order (distance, then the tie key of sampling_gpu.cu's LDS tree)             (pass 2)
// with the reference's arithmetic (eda_sumsq3<MODE>, point minus centre), the 1e10 initial distance and the
// origin-ball rule.  Scenes that pass get idx = 0..m-1 here; the others set fail[scene] and are sampled by
// the regular kernel (launched with only_if = fail).
template <int MODE>
__global__ __launch_bounds__(256) void fps_prefix_sel_kernel(const float *__restrict__ xyz_all, int n, int m,
                                                             int *__restrict__ idx_all, float *__restrict__ sel_all,
                                                             int *__restrict__ fail) {
  __shared__ float cx[256], cy[256], cz[256];
  const int scene = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  const float *xyz = xyz_all + (size_t)scene * n * 3;
  const bool live = r < m && r < n;
  float x = 0.f, y = 0.f, z = 0.f;
  if (live) { x = xyz[3 * r]; y = xyz[3 * r + 1]; z = xyz[3 * r + 2]; }
  if (r < m) idx_all[(size_t)scene * m + r] = r;
  if (r < m && r >= 1) {
    // sample r must exist and must not be a skipped (origin-ball) point
    if (r >= n || (double)eda_sumsq3<MODE>(x, y, z) <= 1e-3) atomicOr(fail + scene, 1);
  }
  float t0 = 1e10f, t1 = 1e10f;
  const int rmax = blockIdx.x * 256 + 255;                     // largest sample of this block: centres 0 .. rmax-1
  for (int c0 = 0; c0 < rmax && c0 < n; c0 += 256) {
    __syncthreads();
    const int c = c0 + threadIdx.x;
    if (c < n) { cx[threadIdx.x] = xyz[3 * c]; cy[threadIdx.x] = xyz[3 * c + 1]; cz[threadIdx.x] = xyz[3 * c + 2]; }
    __syncthreads();
    int cnt = r - c0;                                          // centres i < r
    if (cnt > 256) cnt = 256;
    if (live) {
      int u = 0;
      for (; u + 1 < cnt; u += 2) {                            // two independent min chains
        t0 = fminf(t0, eda_sumsq3<MODE>(x - cx[u], y - cy[u], z - cz[u]));
        t1 = fminf(t1, eda_sumsq3<MODE>(x - cx[u + 1], y - cy[u + 1], z - cz[u + 1]));
      }
      if (u < cnt) t0 = fminf(t0, eda_sumsq3<MODE>(x - cx[u], y - cy[u], z - cz[u]));
    }
  }
  if (live && r >= 1) sel_all[(size_t)scene * m + r] = fminf(t0, t1);
}

template <int MODE>
__global__ __launch_bounds__(256) void fps_prefix_check_kernel(const float *__restrict__ xyz_all, int n, int m,
                                                               int p_log2, const float *__restrict__ sel_all,
                                                               int *__restrict__ fail) {
  __shared__ float cx[256], cy[256], cz[256], cs[256];
  __shared__ unsigned ck[256];
  const int scene = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  const float *xyz = xyz_all + (size_t)scene * n * 3;
  const float *sel = sel_all + (size_t)scene * m;
  float x = 0.f, y = 0.f, z = 0.f;
  bool valid = k < n;
  if (valid) {
    x = xyz[3 * k]; y = xyz[3 * k + 1]; z = xyz[3 * k + 2];
    valid = !((double)eda_sumsq3<MODE>(x, y, z) <= 1e-3);
  }
  const unsigned tk = fps_tiekey((unsigned)(k < n ? k : 0), p_log2);
  float t = 1e10f;
  bool bad = false;
  // rounds r = 1 .. m-1 in blocks of 256 centres staged through LDS (centre of round r = point r-1)
  for (int r0 = 1; r0 < m; r0 += 256) {
    __syncthreads();
    const int c = r0 - 1 + threadIdx.x;                 // centre index for round r0 + threadIdx.x
    if (c < n && r0 + (int)threadIdx.x < m) {
      cx[threadIdx.x] = xyz[3 * c]; cy[threadIdx.x] = xyz[3 * c + 1]; cz[threadIdx.x] = xyz[3 * c + 2];
      cs[threadIdx.x] = sel[r0 + threadIdx.x];
      ck[threadIdx.x] = fps_tiekey((unsigned)(r0 + threadIdx.x), p_log2);
    }
    __syncthreads();
    const int cnt = m - r0 < 256 ? m - r0 : 256;
    if (valid) {
#pragma unroll 4
      for (int u = 0; u < cnt; ++u) {
        t = fminf(t, eda_sumsq3<MODE>(x - cx[u], y - cy[u], z - cz[u]));
        const float sr = cs[u];
        // point k must lose against sample r: smaller distance, or equal distance and a larger tie key
        // (k == r: t == sr and the keys are equal -> not flagged)
        bad = bad || t > sr || (t == sr && tk < ck[u]);
      }
    }
  }
  if (bad) atomicOr(fail + scene, 1);
}

template <int MODE, int T, bool CLUSTER>
int dispatch_p(int P, const float *xyz, int n, int m, int *idx, int S, int G, int p_log2,
               u64 *mail, int *status, hipStream_t stream) {
  switch (P) {
    case 1: return launch_fps<MODE, T, 1, CLUSTER>(xyz, n, m, idx, S, G, p_log2, mail, status, stream);
    case 2: return launch_fps<MODE, T, 2, CLUSTER>(xyz, n, m, idx, S, G, p_log2, mail, status, stream);
    case 4: return launch_fps<MODE, T, 4, CLUSTER>(xyz, n, m, idx, S, G, p_log2, mail, status, stream);
    case 8: return launch_fps<MODE, T, 8, CLUSTER>(xyz, n, m, idx, S, G, p_log2, mail, status, stream);
    case 16:
      if (T == 512)
        return launch_fps<MODE, 512, 16, CLUSTER>(xyz, n, m, idx, S, G, p_log2, mail, status, stream);
      break;
  }
  eda_set_error("fps: unsupported points-per-thread %d for T=%d", P, T);
  return EDA_ERR_UNSUPPORTED;
}

int round_up_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

void eda_fps_env_reset() { g_fps_cu_reserve = INT_MIN; g_fps_policy = EDA_FPS_AUTO; g_fps_background = -1; }

// status block + mailboxes of the cluster kernels (zeroed per call), then the sorted points of the bucket sampler
static size_t fps_mail_bytes(int b) {
  if (b < 0) b = 0;
  return kStatusBytes + (size_t)b * 2 * kMaxG * kRecWords * sizeof(u64);
}

extern "C" size_t eda_fps_workspace_bytes(int b, int n, int m) {
  (void)m;
  size_t bytes = fps_mail_bytes(b);
  if (eda_fps_bucket_supports(n)) bytes += eda_fps_bucket_workspace_bytes(b, n);
  return bytes;
}

// cuda_utils.h:20-24: 2^floor(log2 n) clamped to [1, 512]; only its log2 is needed.
static int fps_block_log2(int n) {
  if (n <= 0) return 0;
  int p = 0;
  while ((2 << p) <= n) ++p;   // floor(log2 n), exact in integers
  if (p > 9) p = 9;
  return p;
}

extern "C" int eda_furthest_point_sampling_f32(const float *xyz, int b, int n, int m, int *idx,
                                               void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && n >= 0 && m >= 0, "negative dimension");
  if (b == 0 || m == 0) return 0;
  EDA_CHECK_ARG(n > 0, "n must be positive when m > 0 (the reference reads point 0)");
  EDA_CHECK_ARG(xyz && idx, "null pointer");
  EDA_CHECK_ARG(ws != nullptr, "workspace required");
  if (ws_bytes < eda_fps_workspace_bytes(b, n, m)) {
    eda_set_error("fps: workspace too small (%zu < %zu)", ws_bytes, eda_fps_workspace_bytes(b, n, m));
    return EDA_ERR_WORKSPACE;
  }
  const int p_log2 = fps_block_log2(n);

  // Scenes of 8193..65536 points have TWO samplers with identical results: the cluster kernels below (13 workgroups per
  // scene spinning on each other: 3.1 ms for 8 x 50 000 -> 2048, but every workgroup of a launch must be co-resident)
  // and the single-workgroup bucket sampler (csrc/fps_bucket.hip: no co-residency requirement, no spin, 8 CUs; 5.3 ms).
  // Policy (eda_fps_set_policy; the environment variable EDA_FPS_BUCKET=0|1 is read at every call and overrides it):
  //   AUTO     cluster kernels, and behind them the bucket sampler GATED on this call's give-up flag: a cluster that
  //            was not co-resident costs its spin limit plus 5 ms, never a wrong index
  //   CLUSTER  cluster kernels only (a give-up becomes the sticky status word)
  //   BUCKET   bucket sampler only (what a data-parallel job next to RCCL's spinning channel kernels should use)
  int policy = g_fps_policy;
  if (eda_knob_set(EDA_K_FPS_BUCKET)) policy = eda_knob(EDA_K_FPS_BUCKET) != 0 ? EDA_FPS_BUCKET : EDA_FPS_CLUSTER;
  const bool bucket_ok = eda_fps_bucket_supports(n);
  if (bucket_ok && policy == EDA_FPS_BUCKET) {
    { const int zrc__ = eda_zero_async(reinterpret_cast<unsigned char *>(ws) + 16, kStatusBytes - 16, stream); if (zrc__) return zrc__; }
    return eda_fps_bucket_launch(xyz, b, n, m, idx, p_log2, reinterpret_cast<unsigned char *>(ws) + fps_mail_bytes(b),
                                 reinterpret_cast<int *>(ws), nullptr, g_eda_fma_mode, stream);
  }

  // ---- geometry: T threads, P points per thread, G workgroups per scene ----
  int T, P, G;
  const int single_max = 8192;   // 512 threads x 16 points, 96 KiB of LDS coordinates
  if (n <= single_max) {
    T = 512;
    P = round_up_pow2((n + T - 1) / T);
    G = 1;
    if (n > 2048 && eda_knob(EDA_K_FPS_SMALL_T) == 1024) { T = 1024; P = round_up_pow2((n + T - 1) / T); }
  } else {
    // measured on MI355X (B=8, N=50 000 -> 2048).  Speculative K=4 kernel: (512,8) 3.1 ms,
    // (512,16) 3.8 ms.  Sequential cluster kernel (EDA_FPS_SPEC=0): (512,16) 4.41 ms,
    // (1024,8) 4.75 ms, (512,8) 4.79 ms, (1024,4) 5.12 ms.
    const bool want_spec = eda_knob(EDA_K_FPS_SPEC) != 0;
    const int p_dflt = (want_spec && ((n + 511) / 512 + 7) / 8 <= kSpecMaxG) ? 8 : 16;
    T = (int)eda_knob(EDA_K_FPS_T);
    P = eda_knob_set(EDA_K_FPS_P) ? (int)eda_knob(EDA_K_FPS_P) : p_dflt;
    if (T != 512 && T != 1024) T = 1024;
    if (!(P == 1 || P == 2 || P == 4 || P == 8 || (P == 16 && T == 512))) { T = 512; P = 16; }
    const int chunks = (n + T - 1) / T;
    G = (chunks + P - 1) / P;
    if (G > kMaxG) {
      T = 1024; P = 8;
      G = ((n + T - 1) / T + P - 1) / P;
    }
    if (G > kMaxG) {
      eda_set_error("fps: n=%d exceeds the supported %d points per scene", n, kMaxG * 8 * 1024);
      return EDA_ERR_UNSUPPORTED;
    }
  }

  int dev = 0, num_cu = 256;
  EDA_CHECK_HIP(hipGetDevice(&dev));
  EDA_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
  // Every workgroup of a launch must be co-resident (they spin on each other): at most one workgroup per CU is
  // assumed, minus a reserve the host sets aside for other resident spin-kernels (RCCL's channel workgroups at N > 1:
  // eda_fps_set_cu_reserve / EDA_FPS_CU_RESERVE) -- larger batches are then sampled in more launches.
  const int avail_cu = num_cu - fps_cu_reserve() > 16 ? num_cu - fps_cu_reserve() : 16;
  int scenes_per_launch = G == 1 ? b : avail_cu / G;
  if (scenes_per_launch < 1) {
    eda_set_error("fps: a cluster of %d workgroups does not fit %d CUs", G, num_cu);
    return EDA_ERR_UNSUPPORTED;
  }

  // ints 0..3 of the workspace are a STICKY status block (int 0 != 0: some call on this workspace gave
  // up); everything behind them (diagnostics, mailboxes) is zeroed per call
  { const int zrc__ = eda_zero_async(reinterpret_cast<unsigned char *>(ws) + 16, fps_mail_bytes(b) - 16, stream); if (zrc__) return zrc__; }
  int *status = reinterpret_cast<int *>(ws);
  u64 *mail = reinterpret_cast<u64 *>(reinterpret_cast<unsigned char *>(ws) + kStatusBytes);

  const int mode = g_eda_fma_mode;
  // clusters of the default geometry run the speculative K=4 kernel (EDA_FPS_SPEC=0: one sample
  // per hand-off, the kernel above)
  const bool use_spec = G > 1 && G <= kSpecMaxG && T == 512 && (P == 16 || P == 8) && eda_knob(EDA_K_FPS_SPEC) != 0;
  const bool fake_giveup = G > 1 && eda_knob(EDA_K_FPS_TEST_GIVEUP) != 0;
  if (fake_giveup) {
    hipLaunchKernelGGL(fps_fake_giveup_kernel, dim3(64), dim3(256), 0, stream, status, idx, (long)b * m);
    EDA_CHECK_LAUNCH();
  }
  for (int s0 = 0; s0 < b && !fake_giveup; s0 += scenes_per_launch) {
    const int S = (b - s0) < scenes_per_launch ? (b - s0) : scenes_per_launch;
    const float *x = xyz + (size_t)s0 * n * 3;
    int *o = idx + (size_t)s0 * m;
    u64 *mb = mail + (size_t)s0 * 2 * kMaxG * kRecWords;
    int rc;
    if (use_spec && P == 8) {
      rc = mode == 0 ? launch_fps_spec<0, 512, 8>(x, n, m, o, S, G, p_log2, mb, status, stream)
                     : launch_fps_spec<1, 512, 8>(x, n, m, o, S, G, p_log2, mb, status, stream);
    } else if (use_spec) {
      rc = mode == 0 ? launch_fps_spec<0, 512, 16>(x, n, m, o, S, G, p_log2, mb, status, stream)
                     : launch_fps_spec<1, 512, 16>(x, n, m, o, S, G, p_log2, mb, status, stream);
    } else if (G == 1) {
      if (T == 512)
        rc = mode == 0 ? dispatch_p<0, 512, false>(P, x, n, m, o, S, G, p_log2, mb, status, stream)
                       : dispatch_p<1, 512, false>(P, x, n, m, o, S, G, p_log2, mb, status, stream);
      else
        rc = mode == 0 ? dispatch_p<0, 1024, false>(P, x, n, m, o, S, G, p_log2, mb, status, stream)
                       : dispatch_p<1, 1024, false>(P, x, n, m, o, S, G, p_log2, mb, status, stream);
    } else {
      if (T == 512)
        rc = mode == 0 ? dispatch_p<0, 512, true>(P, x, n, m, o, S, G, p_log2, mb, status, stream)
                       : dispatch_p<1, 512, true>(P, x, n, m, o, S, G, p_log2, mb, status, stream);
      else
        rc = mode == 0 ? dispatch_p<0, 1024, true>(P, x, n, m, o, S, G, p_log2, mb, status, stream)
                       : dispatch_p<1, 1024, true>(P, x, n, m, o, S, G, p_log2, mb, status, stream);
    }
    if (rc) return rc;
  }
  if (G > 1) {          // (single-workgroup scenes cannot give up)
    if (bucket_ok && policy == EDA_FPS_AUTO)
      return eda_fps_bucket_launch(xyz, b, n, m, idx, p_log2, reinterpret_cast<unsigned char *>(ws) + fps_mail_bytes(b),
                                   status, status + kFailInt, g_eda_fma_mode, stream);
    hipLaunchKernelGGL(fps_fail_latch_kernel, dim3(1), dim3(1), 0, stream, status);
    EDA_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int eda_fps_set_policy(int policy) {
  EDA_CHECK_ARG(policy == EDA_FPS_AUTO || policy == EDA_FPS_CLUSTER || policy == EDA_FPS_BUCKET, "policy must be EDA_FPS_AUTO / CLUSTER / BUCKET");
  g_fps_policy = policy;
  return 0;
}

extern "C" int eda_fps_set_background(int on) {
  g_fps_background = on != 0;
  return 0;
}

extern "C" int eda_fps_set_cu_reserve(int cus) {
  EDA_CHECK_ARG(cus >= 0 && cus <= 240, "reserve must be 0..240 CUs");
  g_fps_cu_reserve = cus;
  return 0;
}

extern "C" size_t eda_fps_prefix_workspace_bytes(int b, int n, int m) {
  const size_t base = (eda_fps_workspace_bytes(b, n, m) + 15) / 16 * 16;
  return base + (((size_t)(b > 0 ? b : 0) * 4 + 15) / 16 * 16) + (size_t)(b > 0 ? b : 0) * (size_t)(m > 0 ? m : 0) * 4;
}

extern "C" int eda_furthest_point_sampling_prefix_f32(const float *xyz, int b, int n, int m, int *idx, void *ws,
                                                      size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && n >= 0 && m >= 0, "negative dimension");
  if (b == 0 || m == 0) return 0;
  EDA_CHECK_ARG(n > 0 && xyz && idx && ws, "bad arguments");
  // the check costs n*m distance evaluations: only for the single-workgroup sizes (the SA2..SA4 levels)
  if (n > 8192 || m > n || b > 65535) return eda_furthest_point_sampling_f32(xyz, b, n, m, idx, ws, ws_bytes, stream_);
  if (ws_bytes < eda_fps_prefix_workspace_bytes(b, n, m)) {
    eda_set_error("fps: workspace too small");
    return EDA_ERR_WORKSPACE;
  }
  const size_t base = (eda_fps_workspace_bytes(b, n, m) + 15) / 16 * 16;
  int *fail = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(ws) + base);
  float *sel = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(fail) + ((size_t)b * 4 + 15) / 16 * 16);
  { const int zrc = eda_zero_async(fail, (size_t)b * 4, stream); if (zrc) return zrc; }
  const int p_log2 = fps_block_log2(n);
  const dim3 g1((m + 255) / 256, b), g2((n + 255) / 256, b);
  if (g_eda_fma_mode == 0) {
    hipLaunchKernelGGL(fps_prefix_sel_kernel<0>, g1, dim3(256), 0, stream, xyz, n, m, idx, sel, fail);
    hipLaunchKernelGGL(fps_prefix_check_kernel<0>, g2, dim3(256), 0, stream, xyz, n, m, p_log2, sel, fail);
  } else {
    hipLaunchKernelGGL(fps_prefix_sel_kernel<1>, g1, dim3(256), 0, stream, xyz, n, m, idx, sel, fail);
    hipLaunchKernelGGL(fps_prefix_check_kernel<1>, g2, dim3(256), 0, stream, xyz, n, m, p_log2, sel, fail);
  }
  EDA_CHECK_LAUNCH();
  tl_fps_only_if = fail;
  const int rc = eda_furthest_point_sampling_f32(xyz, b, n, m, idx, ws, base, stream_);
  tl_fps_only_if = nullptr;
  return rc;
}
