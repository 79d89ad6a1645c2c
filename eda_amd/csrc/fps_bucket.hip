// fps_bucket.hip -- exact furthest point sampling of a large scene by ONE workgroup, on spatial buckets.
//
// Same contract as fps.hip (reference pointnet2/_ext_src/src/sampling_gpu.cu:74-178, host sampling.cpp:70-91: greedy
// arg-max of the running min distance `temp`, fp32 arithmetic of the selected mode, origin-ball skip, the LDS-tree tie
// rule = minimum over fps_tiekey) and the same results bit for bit; a different algorithm for scenes of 8193..65536
// points (SA1: 50 000 -> 2048), which fps.hip handles with a CLUSTER of 13 workgroups per scene that spin on each
// other (104 co-resident workgroups for 3 ms, a give-up path when they are not co-resident).  Here: no inter-workgroup
// communication at all, 8 scenes = 8 workgroups = 8 CUs (one per XCD: a scene's 1 MB of sorted points stays in that L2).
//
// Idea (bucket / KD-tree FPS, e.g. Han et al., "QuickFPS", restated for exact fp32 reference arithmetic):
//   * the kernel first sorts the scene's valid points by a 16^3 Morton cell code (LDS histogram + scan + scatter) and
//     cuts the sorted array into GROUPS of 64 consecutive points; per group it keeps the bounding box and the group's
//     current arg-max KEY (distance bits, inverted tie key) with the coordinates of that point in the registers of the
//     group's OWNER lane; the points themselves ((x, y, z, temp) + tie key) live in the workspace (L2);
//   * a new sample s changes temp[k] = min(temp[k], |p_k - s|^2) only where |p_k - s|^2 < temp[k].  For a whole group,
//     bound(s, box) = the SAME rounded operation sequence applied to the per-axis gaps between s and the box.  Every
//     fp32 operation of the sequence (subtract, multiply, fma, add) is monotone in its non-negative operands, so
//     bound <= the COMPUTED distance of every point of the group -- exactly, with no safety factor -- and
//     bound >= max temp of the group proves that no temp of the group changes: the group is skipped.  After 2047
//     samples of a 50 000-point scene ~27 N point updates have been done instead of 2047 N;
//   * latency, not work, is what bounds FPS (2047 dependent rounds).  A HAND-OFF loads into registers the groups the
//     pending sample can change plus, speculatively, the groups around the K best group maxima (<= 256 groups = 16 384
//     points, one L2 round trip), then runs rounds entirely out of registers and LDS for as long as each new sample
//     provably touches loaded groups only (the same bound test against the unloaded groups' maxima).  Group g is
//     owned AND processed by wave g mod NW (neighbouring groups of the Morton order go to different waves: the load
//     set of a region spreads evenly), so a round needs no cross-wave work lists: every wave tests its own groups,
//     updates its own affected slots, re-derives its own block maximum -- ONE barrier -- and the NW block maxima give
//     the next sample.  The arg-max is over exact keys of exactly-updated groups, so the emitted sequence IS the
//     sequential one; the speculation only chooses what to load (a wrong guess costs a hand-off, never a result).
//   * the first samples change (almost) every group: those rounds stream all affected groups through registers
//     ("dense" rounds, a handful).
#include "eda_common.h"
#include "fps_bucket.h"

#include <limits.h>
#include <stdlib.h>

namespace {

typedef unsigned long long u64;

constexpr int BGS = 64;             // points per group (one per lane)
constexpr int BMAXG = 1024;         // groups per scene
constexpr int BCAPG = 256;          // groups held in registers during a hand-off
constexpr int BCELLS = 4096;        // 16 x 16 x 16 Morton cells for the sort
constexpr int BWK = 4;              // per-wave entries of the candidate pool

__device__ __forceinline__ unsigned tiekey(unsigned k, int p) {          // = fps.hip fps_tiekey
  const unsigned lowmask = (1u << p) - 1u;
  const unsigned hi = p ? (__brev(k & lowmask) >> (32 - p)) : 0u;
  return (hi << (31 - p)) | (k >> p);
}
__device__ __forceinline__ unsigned untie(unsigned tk, int p) {
  const unsigned hi = p ? (tk >> (31 - p)) : 0u;
  const unsigned low = p ? (__brev(hi) >> (32 - p)) : 0u;
  const unsigned mask = (1u << (31 - p)) - 1u;
  return ((tk & mask) << p) | low;
}

__device__ __forceinline__ float key_dist(u64 key) {     // distance of a group key (key 0 = "no entry" -> -0: nothing compares below it)
  return __int_as_float((int)((unsigned)(key >> 32) ^ 0x80000000u));
}

// wave arg-max of 64-bit keys (0 = no entry); returns the winning key (wave-uniform) and its lane
__device__ __forceinline__ u64 wave_max_key(u64 key, int &wl) {
  const int hi = (int)((unsigned)(key >> 32) ^ 0x80000000u);
  const int lo = (int)((unsigned)key ^ 0x80000000u);
  const int mh = eda_wave_max_i32(hi);
  const bool c = hi == mh;
  const u64 cb = __ballot(c);
  if (__popcll(cb) == 1) {                     // the usual case: one lane holds the largest distance -- no second reduction
    wl = __ffsll((long long)cb) - 1;
    const int l1 = __builtin_amdgcn_readlane(lo, wl);
    return ((u64)((unsigned)mh ^ 0x80000000u) << 32) | (u64)((unsigned)l1 ^ 0x80000000u);
  }
  const int ml = eda_wave_max_i32(c ? lo : INT_MIN);
  const u64 b = __ballot(c && lo == ml);
  wl = __ffsll((long long)b) - 1;
  return ((u64)((unsigned)mh ^ 0x80000000u) << 32) | (u64)((unsigned)ml ^ 0x80000000u);
}

// rank (0..K-1) of this lane's key among the K largest of the wave, K otherwise (key 0 = none).  Heuristic use only.
template <int K>
__device__ __forceinline__ int topk_rank(u64 key, int lane) {
  int rank = K;
  bool live = key != 0ull;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int wl;
    wave_max_key(live ? key : 0ull, wl);
    if (lane == wl && live) { rank = i; live = false; }
  }
  return rank;
}

__device__ __forceinline__ float rd_lane(float v, int l) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}
__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// lower bound of the COMPUTED |p - s|^2 over all points p of a box (see the header: exact by monotonicity)
template <int MODE>
__device__ __forceinline__ float box_bound(const float (&lo)[3], const float (&hi)[3], float sx, float sy, float sz) {
  const float ex = fmaxf(fmaxf(lo[0] - sx, sx - hi[0]), 0.f);
  const float ey = fmaxf(fmaxf(lo[1] - sy, sy - hi[1]), 0.f);
  const float ez = fmaxf(fmaxf(lo[2] - sz, sz - hi[2]), 0.f);
  return eda_sumsq3<MODE>(ex, ey, ez);
}

__device__ __forceinline__ unsigned spread4(unsigned v) {       // bits 0..3 -> bits 0, 3, 6, 9
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

template <int NW>
struct BShared {
  union {
    unsigned hist[BCELLS];          // sort: cell counts -> cursors
    float bbox[BMAXG][6];           // then: group boxes on their way to the owner lanes' registers
  } u;                              // 24 KB
  u64 blk_key[2][NW];               // per-wave block maxima of a round (double-buffered by round parity)
  float blk_xyz[2][NW][3];
  u64 pool_key[NW * BWK];           // candidate pool of a hand-off: the waves' best group maxima
  float pool_xyz[NW * BWK][3];
  unsigned char slot_o[NW][BCAPG / NW];   // per wave: owned index of each register slot
  float sred[NW][6];
  unsigned wtot[NW];
  int unsafe[3];                    // rotating flags (round % 3 / hand-off % 3): see the comments at their use
  int dense[3];
  int nvalid;
};

// Threads: NW waves; wave w owns the groups g = o * NW + w, o = j * 64 + lane (j < GPL): box, key, arg-max coordinates
// in the lane's registers.  Slots: up to BR of a wave's owned groups are resident in registers (one point per lane).
template <int MODE, int NW, int K>
__global__ __launch_bounds__(NW * 64) void fps_bucket_kernel(const float *__restrict__ xyz_all, int n, int m,
                                                             int *__restrict__ idx_all, int p_log2, float *ws_all,
                                                             long ws_stride, int *status, const int *only_if) {
  if (only_if != nullptr && *only_if == 0) return;          // fallback launch: the cluster kernels did not give up
  constexpr int BT = NW * 64;
  constexpr int GPL = BMAXG / BT;          // owned groups per lane
  constexpr int BR = BCAPG / NW;           // register slots per wave
  constexpr int POOL = NW * BWK;
  static_assert(GPL >= 1 && GPL <= 2 && BR <= 32 && BR % 8 == 0 && POOL <= 64 && K <= POOL, "config");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
  BShared<NW> *sh = reinterpret_cast<BShared<NW> *>(smem_);
  // tie keys of the points in the register slots: [wave][slot][lane] (needed only when a group's arg-max is re-derived)
  unsigned *stk_all = reinterpret_cast<unsigned *>(smem_ + ((sizeof(BShared<NW>) + 15) / 16) * 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned *stk = stk_all + (size_t)wave * BR * 64 + lane;
  const int scene = blockIdx.x;
  const float *xyz = xyz_all + (size_t)scene * n * 3;
  int *idx = idx_all + (size_t)scene * m;
  const int NP = (n + BGS - 1) / BGS * BGS;
  float4 *wp4 = reinterpret_cast<float4 *>(ws_all + (size_t)scene * ws_stride);     // (x, y, z, temp) of the sorted points
  unsigned *wtk = reinterpret_cast<unsigned *>(wp4 + NP);                              // their tie keys
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  if (m <= 0) return;
#ifdef EDA_FPS_PROFILE
  // cycles of wave 0 per phase (tools/fps_handoffs.py): 0 prologue, 1 hand-off: need flags + pool, 2 candidates + want
  // flags + slots, 3 dense rounds, 4 load, 5 round: tests + updates, 6 round barrier, 7 arg-max
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
  int n_dense = 0, n_rounds = 0, n_unsafe = 0, n_loaded = 0;
#define BMARK(i) do { const unsigned long long tn = __builtin_readcyclecounter(); acc[i] += tn - tp; tp = tn; } while (0)
#else
#define BMARK(i) do { } while (0)
#endif

  // ================================================================ prologue: sort valid points into groups ====
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = tid; k < n; k += BT) {
    const float x = xyz[3 * (size_t)k], y = xyz[3 * (size_t)k + 1], z = xyz[3 * (size_t)k + 2];
    const float mag = eda_sumsq3<MODE>(x, y, z);
    if (!((double)mag <= 1e-3)) {                       // sampling_gpu.cu:106-107 (double compare)
      mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
      mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { mn[a] = wave_min_f32(mn[a]); mx[a] = wave_max_f32(mx[a]); }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { sh->sred[wave][a] = mn[a]; sh->sred[wave][3 + a] = mx[a]; }
  }
  for (int i = tid; i < BCELLS; i += BT) sh->u.hist[i] = 0u;
  if (tid == 0) { sh->unsafe[0] = sh->unsafe[1] = sh->unsafe[2] = 0; sh->dense[0] = sh->dense[1] = sh->dense[2] = 0; }
  __syncthreads();
  float slo[3], sinv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float lo_ = INFINITY, hi_ = -INFINITY;
    for (int w = 0; w < NW; ++w) { lo_ = fminf(lo_, sh->sred[w][a]); hi_ = fmaxf(hi_, sh->sred[w][3 + a]); }
    slo[a] = lo_;
    const float ext = hi_ - lo_;
    sinv[a] = ext > 0.f ? 16.f / ext : 0.f;
  }
  auto cell_of = [&](float x, float y, float z) -> unsigned {
    const unsigned qx = (unsigned)min(15, max(0, (int)((x - slo[0]) * sinv[0])));
    const unsigned qy = (unsigned)min(15, max(0, (int)((y - slo[1]) * sinv[1])));
    const unsigned qz = (unsigned)min(15, max(0, (int)((z - slo[2]) * sinv[2])));
    return spread4(qx) | (spread4(qy) << 1) | (spread4(qz) << 2);
  };
  for (int k = tid; k < n; k += BT) {
    const float x = xyz[3 * (size_t)k], y = xyz[3 * (size_t)k + 1], z = xyz[3 * (size_t)k + 2];
    const float mag = eda_sumsq3<MODE>(x, y, z);
    if (!((double)mag <= 1e-3)) atomicAdd(&sh->u.hist[cell_of(x, y, z)], 1u);
  }
  __syncthreads();
  {   // exclusive scan of the 4096 counts (BCELLS / BT consecutive cells per thread)
    constexpr int CPT = BCELLS / BT;
    unsigned c[CPT], mine = 0;
#pragma unroll
    for (int i = 0; i < CPT; ++i) { c[i] = sh->u.hist[CPT * tid + i]; mine += c[i]; }
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = __shfl_up(inc, o);
      if (lane >= o) inc += v;
    }
    if (lane == 63) sh->wtot[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += sh->wtot[w];
    if (tid == BT - 1) sh->nvalid = (int)(base + inc);
    unsigned e = base + inc - mine;
#pragma unroll
    for (int i = 0; i < CPT; ++i) { sh->u.hist[CPT * tid + i] = e; e += c[i]; }
  }
  __syncthreads();
  const int nvalid = sh->nvalid;
  const int NG = (nvalid + BGS - 1) / BGS;
  if (tid == 0) idx[0] = 0;                              // sampling_gpu.cu:90-92
  if (nvalid == 0) {                                     // every point inside the origin ball: the reference emits index 0
    for (int i = 1 + tid; i < m; i += BT) idx[i] = 0;
    return;
  }
  for (int k = tid; k < n; k += BT) {
    const float x = xyz[3 * (size_t)k], y = xyz[3 * (size_t)k + 1], z = xyz[3 * (size_t)k + 2];
    const float mag = eda_sumsq3<MODE>(x, y, z);
    if (!((double)mag <= 1e-3)) {
      const unsigned pos = atomicAdd(&sh->u.hist[cell_of(x, y, z)], 1u);
      wp4[pos] = make_float4(x, y, z, 1e10f);            // temp = 1e10: sampling.cpp:78-80
      wtk[pos] = tiekey((unsigned)k, p_log2);
    }
  }
  for (int pos = nvalid + tid; pos < NG * BGS; pos += BT) {     // padding of the last group: never a candidate
    wp4[pos] = make_float4(0.f, 0.f, 0.f, 0.f);
    wtk[pos] = 0xFFFFFFFFu;
  }
  // (one workgroup = one CU = one L1: __syncthreads() -- which waits for the outstanding stores -- is all the waves
  // need to see each other's global writes; no agent-scope fence, which on gfx950 would write back the whole L2)
  __syncthreads();
  // group boxes (the histogram is dead: its LDS is the staging area)
  for (int g = wave; g < NG; g += NW) {
    const int pos = g * BGS + lane;
    const bool live = pos < nvalid;
    const float4 p = wp4[pos];
    const float lx = wave_min_f32(live ? p.x : INFINITY), ly = wave_min_f32(live ? p.y : INFINITY), lz = wave_min_f32(live ? p.z : INFINITY);
    const float hx = wave_max_f32(live ? p.x : -INFINITY), hy = wave_max_f32(live ? p.y : -INFINITY), hz = wave_max_f32(live ? p.z : -INFINITY);
    if (lane == 0) {
      sh->u.bbox[g][0] = lx; sh->u.bbox[g][1] = ly; sh->u.bbox[g][2] = lz;
      sh->u.bbox[g][3] = hx; sh->u.bbox[g][4] = hy; sh->u.bbox[g][5] = hz;
    }
  }
  __syncthreads();
  // owner registers: box, key (placeholder: temp = 1e10 everywhere -- the first sample's dense round computes the real
  // keys), coordinates of the group's arg-max point
  // (a lane without a group carries a zero box and key 0 = distance -0: `bound < -0` never holds, so it is never
  // flagged and never wins -- no predicate in the loops below)
  float blo[GPL][3], bhi[GPL][3], win[GPL][3];
  u64 key[GPL];
#pragma unroll
  for (int j = 0; j < GPL; ++j) {
    const int g = (j * 64 + lane) * NW + wave;
    const bool own = g < NG;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      blo[j][a] = own ? sh->u.bbox[g][a] : 0.f;
      bhi[j][a] = own ? sh->u.bbox[g][3 + a] : 0.f;
      win[j][a] = 0.f;
    }
    key[j] = own ? ((u64)(__float_as_uint(1e10f) ^ 0x80000000u) << 32) : 0ull;
  }

  // ================================================================================================ main loop ===
  float sx = xyz[0], sy = xyz[1], sz = xyz[2];          // the sample to APPLY next (already emitted)
  int emitted = 1, handoffs = 0, rnd = 0;
  int r3 = 0, h3 = 0;      // rnd % 3, handoffs % 3 (flag slots rotate: a slot is cleared one step before it is used)
  // this wave's block maximum (best of its owned groups): key, coordinates, owned index of the group
  u64 bkey = 0ull;
  float bx = 0.f, by = 0.f, bz = 0.f;
  int bown = -1;

  // register slots
  float4 sp[BR];
  int slot_o = 0;          // lane r < BR: owned index of slot r
  int slot_wl = -1;        // lane r < BR: lane of the slot group's arg-max point (-1: unknown)
  int nslots = 0;
  unsigned dirty = 0u;     // slots whose temp changed since they were loaded
  u64 loaded[GPL];
#pragma unroll
  for (int j = 0; j < GPL; ++j) loaded[j] = 0ull;

  // new key / arg-max coordinates of the group in slot registers (p, tk) -> owner lane registers; returns the arg-max lane
  auto refresh_group = [&](const float4 &p, unsigned tk, int o) -> int {       // tk: this lane's tie key of the group
    const int hb = __float_as_int(p.w);                 // temp >= 0: int order == float order
    const int wm = eda_wave_max_i32(hb);
    const bool c = hb == wm;
    const u64 cb = __ballot(c);
    unsigned tmin;
    int wl;
    if (__popcll(cb) == 1) {                   // one point holds the group's largest distance: its tie key decides nothing
      wl = __ffsll((long long)cb) - 1;
      tmin = (unsigned)__builtin_amdgcn_readlane((int)tk, wl);
    } else {
      tmin = eda_wave_min_u32(c ? tk : 0xFFFFFFFFu);
      wl = __ffsll((long long)__ballot(c && tk == tmin)) - 1;
    }
    const u64 nk = ((u64)((unsigned)wm ^ 0x80000000u) << 32) | (u64)(~tmin);
    const float wx = rd_lane(p.x, wl), wy = rd_lane(p.y, wl), wz = rd_lane(p.z, wl);
    const int oj = o >> 6, ol = o & 63;
#pragma unroll
    for (int j = 0; j < GPL; ++j)
      if (oj == j && lane == ol) { key[j] = nk; win[j][0] = wx; win[j][1] = wy; win[j][2] = wz; }
    return wl;
  };
  // block maximum of this wave from the owner registers
  auto refresh_block = [&]() {
    u64 kk = key[0];
    float wx = win[0][0], wy = win[0][1], wz = win[0][2];
    int jj = 0;
#pragma unroll
    for (int j = 1; j < GPL; ++j)
      if (key[j] > kk) { kk = key[j]; wx = win[j][0]; wy = win[j][1]; wz = win[j][2]; jj = j; }
    int bl;
    bkey = wave_max_key(kk, bl);
    bx = rd_lane(wx, bl); by = rd_lane(wy, bl); bz = rd_lane(wz, bl);
    bown = __builtin_amdgcn_readlane(jj, bl) * 64 + bl;
  };
  // end of a round: publish the block maximum, ONE barrier, arg-max over the NW block maxima -> next sample
  auto publish_block = [&]() {
    if (lane == 0) {
      sh->blk_key[rnd & 1][wave] = bkey;
      sh->blk_xyz[rnd & 1][wave][0] = bx; sh->blk_xyz[rnd & 1][wave][1] = by; sh->blk_xyz[rnd & 1][wave][2] = bz;
    }
  };
  auto next_sample = [&]() {       // after the barrier of round `rnd`
    const u64 k2 = lane < NW ? sh->blk_key[rnd & 1][lane] : 0ull;
    int wl2;
    const u64 best = wave_max_key(k2, wl2);
    sx = sh->blk_xyz[rnd & 1][wl2][0]; sy = sh->blk_xyz[rnd & 1][wl2][1]; sz = sh->blk_xyz[rnd & 1][wl2][2];
    if (tid == 0) idx[emitted] = (int)untie(~(unsigned)best, p_log2);
    ++emitted;
  };
  // apply the pending sample to slot R (compile-time index) if its bit is set in `smask`
#define B_SLOT_UPDATE(R)                                                                                   \
  if ((smask >> (R)) & 1u) {                                                                               \
    const float dd = eda_sumsq3<MODE>(sp[R].x - sx, sp[R].y - sy, sp[R].z - sz);   /* point minus centre */ \
    const float nd = fminf(dd, sp[R].w);                                                                   \
    const u64 chm = __ballot(nd != sp[R].w);                                                               \
    sp[R].w = nd;                                                                                          \
    if (chm != 0ull) {                                                                                     \
      dirty |= 1u << (R);                                                                                  \
      const int wl_ = __builtin_amdgcn_readlane(slot_wl, R);                                               \
      if (wl_ < 0 || ((chm >> wl_) & 1ull)) {         /* the group's arg-max point moved (or is unknown) */ \
        const int o_ = __builtin_amdgcn_readlane(slot_o, R);                                               \
        const int nwl = refresh_group(sp[R], stk[64 * (R)], o_);                                                  \
        if (lane == (R)) slot_wl = nwl;                                                                    \
        if (o_ == bown || bown < 0) need_block = true;                                                     \
      }                                                                                                    \
    }                                                                                                      \
  }

  BMARK(0);
  while (emitted < m) {
    ++handoffs;
    h3 = h3 == 2 ? 0 : h3 + 1;
    // ---- H1. groups the pending sample can change; this wave's entries of the candidate pool -------------------
    u64 need[GPL];
    int n_need = 0;
    u64 kk = 0ull;
    float cwx = 0.f, cwy = 0.f, cwz = 0.f;
#pragma unroll
    for (int j = 0; j < GPL; ++j) {
      need[j] = __ballot(box_bound<MODE>(blo[j], bhi[j], sx, sy, sz) < key_dist(key[j]));
      n_need += __popcll(need[j]);
      if (key[j] > kk) { kk = key[j]; cwx = win[j][0]; cwy = win[j][1]; cwz = win[j][2]; }
    }
    if (tid == 0) sh->dense[h3 == 2 ? 0 : h3 + 1] = 0;    // (last read two hand-offs ago; every wave has passed a barrier since)
    if (n_need > BR && lane == 0) sh->dense[h3] = 1;
    {
      const int r4 = topk_rank<BWK>(kk, lane);
      if (lane < BWK) sh->pool_key[wave * BWK + lane] = 0ull;                                        // (fewer than BWK live keys)
      if (r4 < BWK) {                                                                                // (same wave: LDS in order)
        sh->pool_key[wave * BWK + r4] = kk;
        sh->pool_xyz[wave * BWK + r4][0] = cwx; sh->pool_xyz[wave * BWK + r4][1] = cwy; sh->pool_xyz[wave * BWK + r4][2] = cwz;
      }
    }
    __syncthreads();
    BMARK(1);
    if (sh->dense[h3]) {
      // ---- dense round: every wave streams the groups IT owns that the sample can change through a few slots ------
      bool need_block = false;
      constexpr int DC = 8;
#pragma unroll
      for (int j = 0; j < GPL; ++j) {
        u64 todo = need[j];
        while (todo != 0ull) {
          int os[DC];
          unsigned tks[DC];
#pragma unroll
          for (int r = 0; r < DC; ++r) {
            os[r] = -1;
            tks[r] = 0xFFFFFFFFu;
            if (todo != 0ull) {
              const int l = __ffsll((long long)todo) - 1;
              todo &= todo - 1ull;
              os[r] = j * 64 + l;
              const int pos = (os[r] * NW + wave) * BGS + lane;
              sp[r] = wp4[pos]; tks[r] = wtk[pos];
            }
          }
#pragma unroll
          for (int r = 0; r < DC; ++r) {
            if (os[r] < 0) continue;
            const float dd = eda_sumsq3<MODE>(sp[r].x - sx, sp[r].y - sy, sp[r].z - sz);
            const float nd = fminf(dd, sp[r].w);
            if (nd != sp[r].w) { sp[r].w = nd; wp4[(os[r] * NW + wave) * BGS + lane].w = nd; }
            refresh_group(sp[r], tks[r], os[r]);
            need_block = true;
          }
        }
      }
      if (need_block || bown < 0) refresh_block();
      publish_block();
      __syncthreads();
      next_sample();
      ++rnd; r3 = r3 == 2 ? 0 : r3 + 1;
#ifdef EDA_FPS_PROFILE
      ++n_dense;
#endif
      BMARK(3);
      continue;
    }
    // ---- H2. candidates: the K best pool entries; the groups they could change are wanted -- the better half of the
    // candidates first (tier 0), the rest behind them (tier 1) ---------------------------------------------------------
    u64 want[2][GPL];
    {
      const u64 pk = lane < POOL ? sh->pool_key[lane] : 0ull;
      int rank = 0;
      for (int i = 0; i < POOL; ++i) {
        const u64 ki = sh->pool_key[i];
        rank += (ki > pk || (ki == pk && i < lane)) ? 1 : 0;
      }
      u64 cm[2] = {__ballot(pk != 0ull && rank < K / 2), __ballot(pk != 0ull && rank >= K / 2 && rank < K)};
      float kd[GPL];
#pragma unroll
      for (int j = 0; j < GPL; ++j) kd[j] = key_dist(key[j]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bool w_[GPL];
#pragma unroll
        for (int j = 0; j < GPL; ++j) w_[j] = false;
        u64 cmask = cm[t];
        while (cmask != 0ull) {
          const int ci = __ffsll((long long)cmask) - 1;
          cmask &= cmask - 1ull;
          const float cx = sh->pool_xyz[ci][0], cy = sh->pool_xyz[ci][1], cz = sh->pool_xyz[ci][2];
#pragma unroll
          for (int j = 0; j < GPL; ++j) w_[j] = w_[j] || (box_bound<MODE>(blo[j], bhi[j], cx, cy, cz) < kd[j]);
        }
#pragma unroll
        for (int j = 0; j < GPL; ++j) want[t][j] = __ballot(w_[j]) & ~need[j];
      }
#pragma unroll
      for (int j = 0; j < GPL; ++j) want[1][j] &= ~want[0][j];
    }
    // ---- H3. slots: needed groups first, then the wanted ones by tier, up to the capacity ---------------------------
    {
      int cum = 0;
      const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
      for (int j = 0; j < GPL; ++j) loaded[j] = need[j];
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int j = 0; j < GPL; ++j) {
          const u64 mk = pass == 0 ? need[j] : want[pass - 1][j];
          const int rk = cum + __popcll(mk & lt);
          const bool take = ((mk >> lane) & 1ull) && rk < BR;
          if (take) sh->slot_o[wave][rk] = (unsigned char)(j * 64 + lane);
          if (pass > 0) loaded[j] |= __ballot(take);
          cum += __popcll(mk);
        }
      nslots = min(BR, cum);
      slot_o = lane < nslots ? (int)sh->slot_o[wave][lane] : 0;      // (same wave: LDS in order)
      slot_wl = -1;
      dirty = 0u;
    }
    BMARK(2);
    // ---- H4. load the slots: every load is issued before the first one is consumed (a slot beyond nslots re-reads
    // slot 0's group: no branch between the loads) -------------------------------------------------------------------
    {
      int pos[BR];
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        const int o = __builtin_amdgcn_readlane(slot_o, r < nslots ? r : 0);
        pos[r] = (o * NW + wave) * BGS + lane;
      }
#pragma unroll
      for (int r = 0; r < BR; ++r) sp[r] = wp4[pos[r]];
#pragma unroll
      for (int r0 = 0; r0 < BR; r0 += 8) {
        unsigned t8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t8[i] = wtk[pos[r0 + i]];
#pragma unroll
        for (int i = 0; i < 8; ++i) stk[64 * (r0 + i)] = t8[i];
      }
    }
#ifdef EDA_FPS_PROFILE
    n_loaded += nslots;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    BMARK(4);
    // ---- H5. rounds out of registers, while every new sample provably touches loaded groups only ------------------
    for (;;) {
      if (tid == 0) sh->unsafe[r3 == 2 ? 0 : r3 + 1] = 0;      // (last read after the barrier of round rnd - 2)
      u64 aff[GPL];
      bool bad = false;
#pragma unroll
      for (int j = 0; j < GPL; ++j) {
        aff[j] = __ballot(box_bound<MODE>(blo[j], bhi[j], sx, sy, sz) < key_dist(key[j]));
        bad = bad || (aff[j] & ~loaded[j]) != 0ull;
      }
      if (bad && lane == 0) sh->unsafe[r3] = 1;
      // slots whose group the sample can change
      unsigned smask;
      {
        bool hit = false;
#pragma unroll
        for (int j = 0; j < GPL; ++j) hit = hit || ((slot_o >> 6) == j && ((aff[j] >> (slot_o & 63)) & 1ull));
        smask = (unsigned)__ballot(hit && lane < nslots);
      }
      bool need_block = false;
      if (smask & 0x000000FFu) { B_SLOT_UPDATE(0) B_SLOT_UPDATE(1) B_SLOT_UPDATE(2) B_SLOT_UPDATE(3) B_SLOT_UPDATE(4) B_SLOT_UPDATE(5) B_SLOT_UPDATE(6) B_SLOT_UPDATE(7) }
      if constexpr (BR > 8)
        if (smask & 0x0000FF00u) { B_SLOT_UPDATE(8) B_SLOT_UPDATE(9) B_SLOT_UPDATE(10) B_SLOT_UPDATE(11) B_SLOT_UPDATE(12) B_SLOT_UPDATE(13) B_SLOT_UPDATE(14) B_SLOT_UPDATE(15) }
      if constexpr (BR > 16) {
        if (smask & 0x00FF0000u) { B_SLOT_UPDATE(16) B_SLOT_UPDATE(17) B_SLOT_UPDATE(18) B_SLOT_UPDATE(19) B_SLOT_UPDATE(20) B_SLOT_UPDATE(21) B_SLOT_UPDATE(22) B_SLOT_UPDATE(23) }
        if (smask & 0xFF000000u) { B_SLOT_UPDATE(24) B_SLOT_UPDATE(25) B_SLOT_UPDATE(26) B_SLOT_UPDATE(27) B_SLOT_UPDATE(28) B_SLOT_UPDATE(29) B_SLOT_UPDATE(30) B_SLOT_UPDATE(31) }
      }
      if (need_block) refresh_block();
      publish_block();
      BMARK(5);
      __syncthreads();
      BMARK(6);
      const bool unsafe_ = sh->unsafe[r3] != 0;
#ifdef EDA_FPS_PROFILE
      ++n_rounds;
      if (unsafe_) ++n_unsafe;
#endif
      if (unsafe_) { ++rnd; r3 = r3 == 2 ? 0 : r3 + 1; break; }     // the sample stays pending: the next hand-off loads what it needs
      next_sample();
      ++rnd; r3 = r3 == 2 ? 0 : r3 + 1;
      BMARK(7);
      if (emitted >= m) break;
    }
    // ---- H6. write the changed slots' running distances back (this wave reloads only its own groups: program order) ----
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      if ((dirty >> r) & 1u) {
        const int o = __builtin_amdgcn_readlane(slot_o, r);
        wp4[(o * NW + wave) * BGS + lane].w = sp[r].w;
      }
    }
#pragma unroll
    for (int j = 0; j < GPL; ++j) loaded[j] = 0ull;
    nslots = 0;
  }
#ifdef EDA_FPS_PROFILE
  if (tid == 0 && scene == 0) {
    for (int i = 0; i < 7; ++i) status[16 + i] = (int)(acc[i + 1] >> 4);
    status[24] = n_dense; status[25] = n_rounds; status[26] = n_unsafe; status[27] = n_loaded; status[28] = (int)(acc[0] >> 4);
  }
#endif
  if (tid == 0 && scene == 0 && only_if != nullptr) atomicAdd(status + 2, 1);     // sticky: give-ups recovered here
  if (tid == 0 && scene < 56) status[4 + scene] = handoffs;                 // diagnostics (tools/fps_handoffs.py)
  if (tid == 0 && scene == 0) status[3] = (int)(__builtin_amdgcn_s_memrealtime() - t_begin);
#undef B_SLOT_UPDATE
#undef BMARK
}

}  // namespace

size_t eda_fps_bucket_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  const size_t np = (size_t)(n + BGS - 1) / BGS * BGS;
  return (size_t)b * 5 * np * sizeof(float);
}

bool eda_fps_bucket_supports(int n) { return n > 8192 && n <= BMAXG * BGS; }

template <int MODE, int NW, int K>
static int launch_bucket(const float *xyz, int b, int n, int m, int *idx, int p_log2, float *w, long stride, int *status,
                         const int *only_if, hipStream_t stream) {
  const size_t lds = ((sizeof(BShared<NW>) + 15) / 16) * 16 + (size_t)BCAPG * 64 * sizeof(unsigned);
  auto kern = fps_bucket_kernel<MODE, NW, K>;
  // > 64 KB of LDS needs the attribute; it is per device and cheap: set on every launch (no process-wide flag)
  EDA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(b), dim3(NW * 64), lds, stream, xyz, n, m, idx, p_log2, w, stride, status, only_if);
  EDA_CHECK_LAUNCH();
  return 0;
}

int eda_fps_bucket_launch(const float *xyz, int b, int n, int m, int *idx, int p_log2, void *ws, int *status,
                          const int *only_if, int mode, hipStream_t stream) {
  const long stride = 5L * ((n + BGS - 1) / BGS * BGS);
  float *w = reinterpret_cast<float *>(ws);
  // measured (8 x 50 000 -> 2048, MI355X): 16 waves x 16 slots 4.76 ms, 8 waves x 32 slots 5.23 ms (EDA_FPS_BUCKET_NW=8)
  const int nw = (int)eda_knob(EDA_K_FPS_BUCKET_NW);
  if (nw == 8)
    return mode == 0 ? launch_bucket<0, 8, 16>(xyz, b, n, m, idx, p_log2, w, stride, status, only_if, stream)
                     : launch_bucket<1, 8, 16>(xyz, b, n, m, idx, p_log2, w, stride, status, only_if, stream);
  return mode == 0 ? launch_bucket<0, 16, 16>(xyz, b, n, m, idx, p_log2, w, stride, status, only_if, stream)
                   : launch_bucket<1, 16, 16>(xyz, b, n, m, idx, p_log2, w, stride, status, only_if, stream);
}
