// mha2.hip -- fused multi-head attention for gfx950 (fp32 MFMA, or 16-bit contraction quads), forward and ONE-PASS
// backward, and the C entry points eda_mha_fwd/bwd(_f32) (include/eda_hip.h).
//
// The attention the reference runs through torch.nn.MultiheadAttention (models/encoder_decoder_layers.py:87-117,
// 149-153, 179-183, 366-401 -> F.multi_head_attention_forward: q*scale, bmm QK^T, -inf key-padding fill, softmax,
// dropout(0.1), bmm PV).  (The round-1/2 kernels -- 4-wave workgroups on 64-row tiles, two-kernel backward,
// mha.hip / mha16.hip -- were removed in round 4; profiles/r01k_mha_pmc.md is their record.)
//
// Structure (both kernels): ONE big workgroup per CU (up to 16 waves = 4 per SIMD, <= 128 registers), the
// streamed operand arrives in LARGE chunks (64-96 query rows / 192-256 key rows) by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no LDS store pass) into a double buffer, and between two chunk
// barriers every wave works through its 4-16 sixteen-row sub-tiles on its own -- the waves of a SIMD drift apart
// and one wave's softmax / exp / dropout VALU runs under another's MFMAs.
//
//   forward   wave = 16 queries (x a share of the key tiles when Lq is short: the partial (m, l, O) of the KS
//             key shares are merged through LDS at the end); S^T = K Q^T so that a query's softmax statistics
//             are lane-local, exp2 with log2(e) folded into the query scale, 1/(1-p) folded into the final 1/l.
//   backward  wave = 16 keys: K, V rows in registers for the lifetime of the workgroup, dK^T / dV^T
//             accumulators in registers.  Phase A, per 16-query sub-tile: S = Q K^T and dP = dO V^T (9 + 9 MFMA),
//             P = exp2(S - lse), dS = P o (dP - delta); dV^T += dO^T P, dK^T += Q^T dS (8 + 8 + 2 x 4 small); dS
//             goes to the block's LDS tile DS[query][key].  Barrier.  Phase B: work item = one 16-dim x 16-query
//             tile of dQ^T = K^T dS^T contracted over ALL keys of the block (K rows of the block in LDS), written
//             straight to dQ, or to this key block's dense partial when Lk > 256 (summed in split order by
//             mha2_part_reduce_kernel).  45 MFMAs issued per 16 x 16 pair = 45 useful: the last four of the 36
//             head dims run on v_mfma_f32_4x4x1 (see mfma44 below).  delta = rowsum(dO o O) is computed while a
//             chunk is staged.  No atomics anywhere: ds_add_f32 and global fp32 atomics were measured 3x slower
//             (DESIGN.md section 4).
//
// Dropout: counter hash (one 32-bit hash per two keys, 16 bits each), regenerated in the backward.
#include "eda_common.h"
#include "mha2.h"

#include <stdlib.h>

namespace {

constexpr int HD = 36;
constexpr int KSTEPS = 9;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifdef EDA_MHA2_PROFILE
// Phase timing of the backward kernel (experiments only; tools/mha2_phase_profile.py): per-wave s_memtime deltas,
// summed over all waves of all launches since the last read.
__device__ unsigned long long mha2_prof[64 * 8];
#define PSTAMP(slot)                                                   \
  do {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime();     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(now__));                 \
    prof_acc[slot] += now__ - prof_t;                                  \
    prof_t = now__;                                                    \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16 blocks of 4x4x1 (probed on gfx950, tools/probe/mfma44_layout.hip): D[reg i][lane 4b+j] += A[lane 4b+i] * B[lane 4b+j].
// Used for the LAST FOUR of the 36 head dims of every "output = head dim" product (P V, dO^T P, Q^T dS, K^T dS^T):
// with the 16x16x4 form they cost a whole 16-row tile (12 of 16 rows wasted: 54 MFMA-equivalents per backward pair
// for 45 useful); here block b = (lane group g, quad of the lane-local index), the four k-steps t of a sub-tile are
// four 8-cycle instructions, and every lane group accumulates the partial sum over ITS quarter of the contraction
// index -- the four partials are added once, at the very end (grp_sum4).  45 issued for 45 useful.
__device__ __forceinline__ f32x4 mfma44(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// Arithmetic of the contractions: exact fp32 (the parity path) or 16-bit operands with fp32 accumulation (BASELINE.json
// configs[2] / configs[4]).  The 16-bit forms are the SAME kernels: wherever the fp32 code issues four consecutive
// 16x16x4 steps whose operands a lane holds as four values (a contraction quad), the 16-bit code packs the four values
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) and issues ONE v_mfma_f32_16x16x16 (one v_mfma_f32_4x4x4 for the 4-dim tile);
// the 36-deep head-dim contraction (8 + 1 values per lane) becomes three of them, the third carrying one value and three
// zeros.  Tensors stay fp32 in HBM and LDS; softmax, dS and all accumulators stay fp32.
typedef unsigned long long u64;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

struct ArF32 { static constexpr bool is16 = false; };
struct ArBf16 {
  static constexpr bool is16 = true;
  static __device__ __forceinline__ u64 pk4(float a, float b, float c, float d) {
    const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{c, d}, bf16x2));
    return (u64)lo | ((u64)hi << 32);
  }
  static __device__ __forceinline__ f32x4 mma16(u64 a, u64 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mma44(u64 a, u64 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};
struct ArFp16 {
  static constexpr bool is16 = true;
  static __device__ __forceinline__ u64 pk4(float a, float b, float c, float d) {
    const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{c, d}, h16x2));
    return (u64)lo | ((u64)hi << 32);
  }
  static __device__ __forceinline__ f32x4 mma16(u64 a, u64 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mma44(u64 a, u64 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
  }
};

// a lane's 9 contraction values of a head-dim row as three packed quads (the third: one value + zeros)
struct Row16 { u64 p[3]; };
template <class AR>
__device__ __forceinline__ Row16 pack_row(const float (&r)[9]) {
  Row16 o;
  o.p[0] = AR::pk4(r[0], r[1], r[2], r[3]);
  o.p[1] = AR::pk4(r[4], r[5], r[6], r[7]);
  o.p[2] = AR::pk4(r[8], 0.f, 0.f, 0.f);
  return o;
}
// acc += sum over the 36 head dims of a-row x b-row (both lanes' 9-value operands)
template <class AR>
__device__ __forceinline__ f32x4 dot36(const Row16 &a, const Row16 &b, f32x4 acc) {
  acc = AR::mma16(a.p[0], b.p[0], acc);
  acc = AR::mma16(a.p[1], b.p[1], acc);
  return AR::mma16(a.p[2], b.p[2], acc);
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// reductions over the 4 lane groups (lanes l, l^16, l^32, l^48) with the gfx950 row / half swaps
__device__ __forceinline__ float grp_max(float v) {
  u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float grp_sum(float v) {
  u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ f32x4 grp_sum4(f32x4 v) {
  f32x4 r;
  r[0] = grp_sum(v[0]); r[1] = grp_sum(v[1]); r[2] = grp_sum(v[2]); r[3] = grp_sum(v[3]);
  return r;
}

// A lane's 9 contraction values of one 36-float row: MFMA k-step s of lane group g contracts head dim 8g+s
// (s < 8) and 32+g (s = 8) -- any permutation is fine as long as both operands use it -- so the first eight are
// two 16-byte reads (row stride 36 is conflict-free for them).
__device__ __forceinline__ void load_row_operand(float (&r)[KSTEPS], const float *row, int g) {
  const float4 x = *reinterpret_cast<const float4 *>(row + 8 * g);
  const float4 y = *reinterpret_cast<const float4 *>(row + 8 * g + 4);
  r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w;
  r[4] = y.x; r[5] = y.y; r[6] = y.z; r[7] = y.w;
  r[8] = row[32 + g];
}

// The lane's transposed-operand values of one 16-row sub-tile of a ROW-major tile (contracted over the rows):
// v[t][n] = X[row 4g + t][dim c + 16n]; third column tile: dim 32 + (c & 3), its output rows >= 36 are never used.
struct ColOperand { float v[4][3]; };
__device__ __forceinline__ void load_col_operand(ColOperand &o, const float *rows, int c, int g) {
  const float *p = rows + 4 * g * HD;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    o.v[t][0] = p[t * HD + c];
    o.v[t][1] = p[t * HD + 16 + c];
    o.v[t][2] = p[t * HD + 32 + (c & 3)];
  }
}

struct DropCfg { unsigned seed, thresh; float inv_keep; };
__device__ __forceinline__ DropCfg drop_cfg(const Mha2Args &a) {
  DropCfg dc = {0u, 0u, 1.f};
  if (a.p_drop > 0.f) {
    dc.seed = hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt);
    dc.thresh = (unsigned)((double)a.p_drop * 65536.0 + 0.5);
    dc.inv_keep = 1.f / (1.f - a.p_drop);
  }
  return dc;
}

// LDS-DMA of `rows` x 36 floats (row r of the tile = global row min(row0 + r, nrows - 1) of one head) into a
// LINEAR [rows][36] LDS tile: piece p = 64 consecutive 16-byte granules, one wave-instruction each
// (the destination of lane l is base + 1 KiB * p + 16 l; the source address is per lane).
template <int ROWS>
__device__ __forceinline__ void dma_rows(float *lds, const float *base, long row_stride, int row0, int nrows,
                                         int wave, int nwaves, int lane, int first_piece) {
  constexpr int G = ROWS * 9;                  // 16-byte granules of the tile
  constexpr int P = (G + 63) / 64;             // pieces (the last one may be partial: its surplus lanes are masked off)
  for (int p = first_piece + wave; p < first_piece + P; p += nwaves) {
    const int pp = p - first_piece;
    const int i = 64 * pp + lane;
    if (G % 64 == 0 || i < G) {
      const int row = i / 9, c4 = i - row * 9;
      const int grow = min(row0 + row, nrows - 1);
      const float *src = base + (long)grow * row_stride + 4 * c4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(lds + 256 * pp), 16, 0, 0);
    }
  }
}

// ======================================================================================== forward ==========
// One 64-key tile for one wave (16 queries): NSUB live 16-key sub-tiles (compile time).  Scores arrive in the
// log2 domain (the query operand carries scale * log2 e).
template <int NSUB, bool DROP, class AR>
__device__ __forceinline__ void fwd_tile(const float *__restrict__ Kt, const float *__restrict__ Vt,
                                         const unsigned *__restrict__ deadw, bool need_mask,
                                         const float (&qreg)[KSTEPS], const Row16 &q16, int c, int g, int key0,
                                         unsigned rowbase, const DropCfg &dc, float &m, float &lsum, f32x4 (&o)[3]) {
  f32x4 st[NSUB];
#pragma unroll
  for (int j = 0; j < NSUB; ++j) st[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    float kreg[NSUB][KSTEPS];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) load_row_operand(kreg[j], Kt + (16 * j + c) * HD, g);
    if constexpr (AR::is16) {
#pragma unroll
      for (int j = 0; j < NSUB; ++j) st[j] = dot36<AR>(pack_row<AR>(kreg[j]), q16, st[j]);
    } else {
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int j = 0; j < NSUB; ++j) st[j] = mfma4(kreg[j][s], qreg[s], st[j]);
    }
  }
  ColOperand va;
  load_col_operand(va, Vt, c, g);
  float tmax = -INFINITY;
  if (need_mask) {
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
      const unsigned dw = deadw[4 * j + g];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool dead = ((dw >> (8 * r)) & 0xffu) != 0u;
        st[j][r] = dead ? -INFINITY : st[j][r];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NSUB; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, st[j][r]);
  tmax = grp_max(tmax);
  const float m_new = fmaxf(m, tmax);
  const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
  const float alpha = __builtin_amdgcn_exp2f(m - m_safe);       // m = -inf -> 0
  float psum = 0.f;
#pragma unroll
  for (int j = 0; j < NSUB; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __builtin_amdgcn_exp2f(st[j][r] - m_safe);
      st[j][r] = p;
      psum += p;
    }
  lsum = lsum * alpha + psum;          // per lane group; the 4 groups are summed once at the end
  m = m_new;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) o[nt] *= alpha;
  if (DROP) {
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
      for (int r2 = 0; r2 < 4; r2 += 2) {
        const unsigned h = hash32(dc.seed ^ (rowbase + (unsigned)(key0 + 16 * j + 4 * g + r2)));
        st[j][r2] = (h & 0xffffu) >= dc.thresh ? st[j][r2] : 0.f;
        st[j][r2 + 1] = (h >> 16) >= dc.thresh ? st[j][r2 + 1] : 0.f;
      }
  }
  // O^T[dim][query] += V^T P^T
#pragma unroll
  for (int j = 0; j < NSUB; ++j) {
    ColOperand vb;
    if (j + 1 < NSUB) load_col_operand(vb, Vt + 16 * (j + 1) * HD, c, g);
    if constexpr (AR::is16) {
      const u64 pb = AR::pk4(st[j][0], st[j][1], st[j][2], st[j][3]);
      o[0] = AR::mma16(AR::pk4(va.v[0][0], va.v[1][0], va.v[2][0], va.v[3][0]), pb, o[0]);
      o[1] = AR::mma16(AR::pk4(va.v[0][1], va.v[1][1], va.v[2][1], va.v[3][1]), pb, o[1]);
      o[2] = AR::mma44(AR::pk4(va.v[0][2], va.v[1][2], va.v[2][2], va.v[3][2]), pb, o[2]);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float pb = st[j][t];
        o[0] = mfma4(va.v[t][0], pb, o[0]);
        o[1] = mfma4(va.v[t][1], pb, o[1]);
        o[2] = mfma44(va.v[t][2], pb, o[2]);       // dims 32..35: per-lane-group partials
      }
    }
    if (j + 1 < NSUB) va = vb;
  }
}

// NQ query sub-tiles x KS key shares = NW waves; CHK keys per LDS chunk, NBUF chunk buffers.
// SPLIT: the keys of a (scene, head, query block) are divided over a.n_ksplit WORKGROUPS as well (short query sets
// against 1024 point keys: 64 x ceil(Lq / 64) workgroups each staging a head's whole 295 KB of K | V leave half the
// chip idle behind a 12 us staging pass).  A workgroup's merged (O, m, l) goes to a.fwd_part with plain stores, is
// published with ONE agent-scope release and a ticket (cdna_hip_programming.md, in-launch split-K recipe); the
// workgroup that draws the last ticket acquires, merges ALL splits in split order (bit-reproducible whoever is last)
// and writes the output.
template <int NQ, int KS, int CHK, int NBUF, bool DROP, class AR, bool SPLIT = false>
__global__ __launch_bounds__(NQ * KS * 64) void mha2_fwd_kernel(const Mha2Args a) {
  constexpr int NW = NQ * KS, NT = NW * 64;
  constexpr int TILES = CHK / 64;
  static_assert(CHK % 64 == 0 && (NBUF == 1 || NBUF == 2), "config");
  __shared__ __attribute__((aligned(16))) float Ks0[CHK * HD];
  __shared__ __attribute__((aligned(16))) float Vs0[CHK * HD];
  __shared__ __attribute__((aligned(16))) float Ks1[NBUF == 2 ? CHK * HD : 4];
  __shared__ __attribute__((aligned(16))) float Vs1[NBUF == 2 ? CHK * HD : 4];
  __shared__ unsigned dead_s[NBUF][CHK / 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int BH = a.B * a.H;
  const int bh = (int)(blockIdx.x % (unsigned)BH);
  int qb = (int)(blockIdx.x / (unsigned)BH), sp = 0;
  if (SPLIT) { sp = qb / a.n_qs; qb -= sp * a.n_qs; }
  const int b = bh / a.H, h = bh - b * a.H;
  const int qs = wave / KS, ks = wave - qs * KS;
  const int qi = qb * (16 * NQ) + 16 * qs + c;
  const bool qvalid = qi < a.Lq;
  const bool wave_live = qb * (16 * NQ) + 16 * qs < a.Lq;       // uniform: any valid query in this wave

  const float *kbase = a.k + (long)b * a.k_sb + h * HD;
  const float *vbase = a.v + (long)b * a.v_sb + h * HD;
  const unsigned char *mrow = a.mask ? a.mask + (long)b * a.Lk : nullptr;

  float qreg[KSTEPS];
  {
    const float *qrow = a.q + (long)b * a.q_sb + (long)(qvalid ? qi : 0) * a.q_sl + h * HD;
    load_row_operand(qreg, qrow, g);
    const float sc = a.scale * LOG2E;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qreg[s] = qvalid ? qreg[s] * sc : 0.f;
  }
  Row16 q16 = {{0, 0, 0}};
  if constexpr (AR::is16) q16 = pack_row<AR>(qreg);
  DropCfg dc = {0u, 0u, 1.f};
  if (DROP) dc = drop_cfg(a);
  const unsigned rowbase = ((unsigned)bh * (unsigned)a.Lq + (unsigned)qi) * (unsigned)a.Lk;

  auto stage = [&](float *Kd, float *Vd, unsigned *dd, int k0) {
    // dead-key flags first (ordinary loads: keep them out of the span in which the DMA is in flight)
    for (int w = tid; w < CHK / 4; w += NT) {
      unsigned word = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 4 * w + r;
        unsigned dead = key >= a.Lk ? 1u : 0u;
        if (key < a.Lk && mrow) dead = mrow[key] ? 1u : 0u;
        word |= dead << (8 * r);
      }
      dd[w] = word;
    }
    dma_rows<CHK>(Kd, kbase, a.k_sl, k0, a.Lk, wave, NW, lane, 0);
    dma_rows<CHK>(Vd, vbase, a.v_sl, k0, a.Lk, wave, NW, lane, CHK * 9 / 64);
  };

  float m = -INFINITY, lsum = 0.f;
  f32x4 o[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

  auto compute = [&](const float *Kc, const float *Vc, const unsigned *dd, int k0) {
    if (!wave_live) return;
#pragma unroll 1
    for (int t = ks; t < TILES; t += KS) {
      const int key0 = k0 + 64 * t;
      if (key0 >= a.Lk) break;
      const int nsub = min(4, (a.Lk - key0 + 15) >> 4);
      const bool need_mask = (mrow != nullptr) || (key0 + 64 > a.Lk);
      const float *Kt = Kc + 64 * t * HD, *Vt = Vc + 64 * t * HD;
      const unsigned *dw = dd + 16 * t;
      if (nsub == 4) fwd_tile<4, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
      else if (nsub == 3) fwd_tile<3, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
      else if (nsub == 2) fwd_tile<2, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
      else fwd_tile<1, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
    }
  };

  // this workgroup's chunks: all of them, or those of its key split (keys_per_split is a multiple of CHK)
  const int c_lo = SPLIT ? sp * (a.keys_per_split / CHK) : 0;
  const int nchunks = SPLIT ? min((a.Lk + CHK - 1) / CHK, c_lo + a.keys_per_split / CHK) : (a.Lk + CHK - 1) / CHK;
  if (nchunks > c_lo) stage(Ks0, Vs0, dead_s[0], c_lo * CHK);
  EDA_SYNC_DMA();
  for (int ci = c_lo; ci < nchunks; ci += NBUF) {
    // even chunk in buffer 0 (the next one is fetched into buffer 1 meanwhile), odd chunk in buffer 1
    if (NBUF == 2 && ci + 1 < nchunks) stage(Ks1, Vs1, dead_s[1], (ci + 1) * CHK);
    compute(Ks0, Vs0, dead_s[0], ci * CHK);
    EDA_SYNC_DMA();
    if (NBUF == 2) {
      if (ci + 1 >= nchunks) break;
      if (ci + 2 < nchunks) stage(Ks0, Vs0, dead_s[0], (ci + 2) * CHK);
      compute(Ks1, Vs1, dead_s[1], (ci + 1) * CHK);
      EDA_SYNC_DMA();
    } else if (ci + 1 < nchunks) {
      stage(Ks0, Vs0, dead_s[0], (ci + 1) * CHK);
      EDA_SYNC_DMA();
    }
  }

  // merge the KS key shares of a query sub-tile (K/V are dead now: their LDS is the scratch; one slot = 64
  // lanes x 16 floats: O (12), m, l)
  lsum = grp_sum(lsum);
  if (KS > 1) {
    constexpr int PER = 64 * 16, NPER = CHK * HD / PER;
    static_assert((KS - 1) * NQ <= 2 * NPER, "merge scratch must fit the K/V buffers");
    auto slot = [&](int q_, int k_) -> float * {
      const int idx = q_ * (KS - 1) + (k_ - 1);
      return idx < NPER ? Ks0 + idx * PER : Vs0 + (idx - NPER) * PER;
    };
    if (ks > 0 && wave_live) {
      float *s_ = slot(qs, ks) + lane * 16;
      *reinterpret_cast<f32x4 *>(s_) = o[0];
      *reinterpret_cast<f32x4 *>(s_ + 4) = o[1];
      *reinterpret_cast<f32x4 *>(s_ + 8) = o[2];
      s_[12] = m; s_[13] = lsum;
    }
    __syncthreads();
    if (ks == 0 && wave_live) {
#pragma unroll 1
      for (int k_ = 1; k_ < KS; ++k_) {
        const float *s_ = slot(qs, k_) + lane * 16;
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(s_);
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(s_ + 4);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(s_ + 8);
        const float mo = s_[12], lo = s_[13];
        const float mn = fmaxf(m, mo);
        const float ms = (mn == -INFINITY) ? 0.f : mn;
        const float fa = __builtin_amdgcn_exp2f(m - ms), fb = __builtin_amdgcn_exp2f(mo - ms);
        o[0] = o[0] * fa + p0 * fb; o[1] = o[1] * fa + p1 * fb; o[2] = o[2] * fa + p2 * fb;
        lsum = lsum * fa + lo * fb;
        m = mn;
      }
    }
  }
  if (KS == 1 || ks == 0) o[2] = grp_sum4(o[2]);
  if constexpr (SPLIT) {
    // ---- cross-workgroup merge of the key splits.  Hand-off (cdna_hip_programming.md G16, form R1 / "publish-large"):
    // WRITE-THROUGH (sc1) partial stores, every storing wave drains them, one lane takes the block's ticket; the last
    // arriver reads all partials with sc1 loads (served past its L1).  No release / acquire fence anywhere: a release
    // is a write-back of the XCD's whole L2 (measured here with fences: 80 x 1024 in 4 splits 34 us, no faster than the
    // unsplit 34; 256 x 1024 in 2 splits 62 us against 38).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int blk = bh * a.n_qs + qb;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        a.fwd_part + (long)blk * a.n_ksplit * NQ * (64 * 16), 0, a.n_ksplit * NQ * 64 * 16 * 4, 0x00020000);
    const int my_off = ((sp * NQ + qs) * 64 + lane) * 64;           // bytes
    if (ks == 0 && wave_live) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[0]), rs, my_off, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[1]), rs, my_off + 16, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[2]), rs, my_off + 32, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{m, lsum, 0.f, 0.f}), rs, my_off + 48, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its stores
    __syncthreads();                                               // (K/V and the merge scratch are dead now)
    unsigned *flag = dead_s[0];
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(a.fwd_tickets + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = t == (unsigned)a.n_ksplit - 1u;
      if (last) __hip_atomic_store(a.fwd_tickets + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
      flag[0] = last ? 1u : 0u;
    }
    __syncthreads();
    if (flag[0] == 0u || ks != 0 || !wave_live) return;
    m = -INFINITY; lsum = 0.f;
    o[0] = o[1] = o[2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int z = 0; z < a.n_ksplit; ++z) {                          // split order: the result does not depend on who is last
      const int off = ((z * NQ + qs) * 64 + lane) * 64;
      const f32x4 p0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
      const f32x4 p1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16));
      const f32x4 p2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 16));
      const f32x4 ml = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 16));
      const float mo = ml[0], lo = ml[1];
      const float mn = fmaxf(m, mo);
      const float ms = (mn == -INFINITY) ? 0.f : mn;
      const float fa = __builtin_amdgcn_exp2f(m - ms), fb = __builtin_amdgcn_exp2f(mo - ms);
      o[0] = o[0] * fa + p0 * fb; o[1] = o[1] * fa + p1 * fb; o[2] = o[2] * fa + p2 * fb;
      lsum = lsum * fa + lo * fb;
      m = mn;
    }
  }
  if (qvalid && (KS == 1 || ks == 0)) {
    const float inv = dc.inv_keep / lsum;          // all keys masked -> NaN, like the reference
    float *orow = a.o + (long)b * a.o_sb + (long)qi * a.o_sl + h * HD;
    *reinterpret_cast<float4 *>(orow + 4 * g) = make_float4(o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv);
    *reinterpret_cast<float4 *>(orow + 16 + 4 * g) = make_float4(o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv);
    if (g == 0) {
      *reinterpret_cast<float4 *>(orow + 32) = make_float4(o[2][0] * inv, o[2][1] * inv, o[2][2] * inv, o[2][3] * inv);
      a.lse[(long)bh * a.Lq + qi] = (m + __builtin_amdgcn_logf(lsum)) * LN2;     // v_log_f32 = log2
    }
  }
}

// ============================================================ forward with the q-projection fused in front ==
// One launch for [q = x Wq^T + bq | softmax(q K^T) V] of an attention site whose key set is short (Lk <= 192: the text
// tokens and detected boxes of models/encoder_decoder_layers.py:99-117, 375-391) -- for those sites the attention
// core alone is a 10 us launch at ~10 % of the MFMA peak, most of it the launch floor, with an 11 us projection
// launch in front of it.  Workgroup = 4 waves = 64 queries of one (scene, head): every head computes ITS 36 columns
// of q (no redundant product), so the fused launch does the work of both at one launch floor.
//   * the wave's 16 input rows go straight from memory into registers as MFMA operand fragments (lane (c, g) holds
//     x[query c][16 j + 4 g .. + 3], j < 18: all 18 loads in flight at once; the contraction order is a permutation of
//     the input index, which a dot product does not care about);
//   * the head's 36 weight rows (41 KB) are staged once per workgroup into LDS with row stride 292 (conflict-free
//     16-byte operand reads), K and V of the (scene, head) by LDS-DMA as in the plain forward;
//   * q^T[dim][query] accumulates in three 16-row tiles (216 MFMAs), gets its bias, is written to memory (the backward
//     needs it) and, through a wave-private LDS strip, becomes the query operand of the unchanged forward tiles.
constexpr int QP_W = 292;          // LDS row stride of the weight rows
constexpr int QP_D = 288;          // input width of the projection (= 8 heads x 36)
template <bool DROP, class AR>
__global__ __launch_bounds__(256) void mha2_qproj_fwd_kernel(const Mha2Args a) {
  constexpr int CHK = 192, NQ = 4;
  __shared__ __attribute__((aligned(16))) float Ks0[CHK * HD];
  __shared__ __attribute__((aligned(16))) float Vs0[CHK * HD];
  __shared__ __attribute__((aligned(16))) float Ws[HD * QP_W];
  __shared__ __attribute__((aligned(16))) float Qs[NQ][16 * HD];
  __shared__ unsigned dead_s[CHK / 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int BH = a.B * a.H;
  const int bh = (int)(blockIdx.x % (unsigned)BH), qb = (int)(blockIdx.x / (unsigned)BH);
  const int b = bh / a.H, h = bh - b * a.H;
  const int qi = qb * (16 * NQ) + 16 * wave + c;
  const bool qvalid = qi < a.Lq;
  const bool wave_live = qb * (16 * NQ) + 16 * wave < a.Lq;

  // this lane's fragments of its input row (clamped row: no predicate on the loads)
  float4 xf[18];
  {
    const float *xr = a.xq + (long)b * a.xq_sb + (long)min(qi, a.Lq - 1) * a.xq_sl + 4 * g;
#pragma unroll
    for (int j = 0; j < 18; ++j) xf[j] = *reinterpret_cast<const float4 *>(xr + 16 * j);
  }
  // K, V of the (scene, head) by DMA; dead-key flags; the head's weight rows through registers (padded rows)
  const float *kbase = a.k + (long)b * a.k_sb + h * HD;
  const float *vbase = a.v + (long)b * a.v_sb + h * HD;
  const unsigned char *mrow = a.mask ? a.mask + (long)b * a.Lk : nullptr;
  for (int w = tid; w < CHK / 4; w += 256) {
    unsigned word = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 4 * w + r;
      unsigned dead = key >= a.Lk ? 1u : 0u;
      if (key < a.Lk && mrow) dead = mrow[key] ? 1u : 0u;
      word |= dead << (8 * r);
    }
    dead_s[w] = word;
  }
  dma_rows<CHK>(Ks0, kbase, a.k_sl, 0, a.Lk, wave, NQ, lane, 0);
  dma_rows<CHK>(Vs0, vbase, a.v_sl, 0, a.Lk, wave, NQ, lane, CHK * 9 / 64);
  {
    const float *wb = a.wq + (long)(h * HD) * a.ldwq;
    for (int i = tid; i < HD * (QP_D / 4); i += 256) {
      const int row = i / (QP_D / 4), c4 = i - row * (QP_D / 4);
      *reinterpret_cast<float4 *>(Ws + row * QP_W + 4 * c4) = *reinterpret_cast<const float4 *>(wb + (long)row * a.ldwq + 4 * c4);
    }
  }
  __syncthreads();

  // q^T[dim 16 t + 4 g + i][query c] = sum_k W[dim][k] x[query][k]
  f32x4 qa[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  {
    const float *w0 = Ws + c * QP_W + 4 * g, *w1 = w0 + 16 * QP_W, *w2 = Ws + min(32 + c, HD - 1) * QP_W + 4 * g;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      const float4 a0 = *reinterpret_cast<const float4 *>(w0 + 16 * j);
      const float4 a1 = *reinterpret_cast<const float4 *>(w1 + 16 * j);
      const float4 a2 = *reinterpret_cast<const float4 *>(w2 + 16 * j);
      const float xs[4] = {xf[j].x, xf[j].y, xf[j].z, xf[j].w};
      const float s0[4] = {a0.x, a0.y, a0.z, a0.w}, s1[4] = {a1.x, a1.y, a1.z, a1.w}, s2[4] = {a2.x, a2.y, a2.z, a2.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qa[0] = mfma4(s0[i], xs[i], qa[0]);
        qa[1] = mfma4(s1[i], xs[i], qa[1]);
        qa[2] = mfma4(s2[i], xs[i], qa[2]);
      }
    }
  }
  {
    // bias, memory (for the backward), LDS strip (-> operand layout of the forward tiles)
    float *qrow = a.q_out + (long)b * a.qo_sb + (long)(qvalid ? qi : 0) * a.qo_sl + h * HD;
    float *qs = Qs[wave] + c * HD;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t == 2 && g != 0) break;                       // dims 32..35 live in lane group 0 of the third tile
      float4 v = make_float4(qa[t][0], qa[t][1], qa[t][2], qa[t][3]);
      if (a.bq) {
        const float4 bb = *reinterpret_cast<const float4 *>(a.bq + h * HD + 16 * t + 4 * g);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      *reinterpret_cast<float4 *>(qs + 16 * t + 4 * g) = v;
      if (qvalid) *reinterpret_cast<float4 *>(qrow + 16 * t + 4 * g) = v;
    }
  }
  float qreg[KSTEPS];
  load_row_operand(qreg, Qs[wave] + c * HD, g);           // (same wave wrote it: LDS is in order per wave)
  {
    const float sc = a.scale * LOG2E;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qreg[s] = qvalid ? qreg[s] * sc : 0.f;
  }
  Row16 q16 = {{0, 0, 0}};
  if constexpr (AR::is16) q16 = pack_row<AR>(qreg);
  DropCfg dc = {0u, 0u, 1.f};
  if (DROP) dc = drop_cfg(a);
  const unsigned rowbase = ((unsigned)bh * (unsigned)a.Lq + (unsigned)qi) * (unsigned)a.Lk;

  float m = -INFINITY, lsum = 0.f;
  f32x4 o[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  if (wave_live) {
#pragma unroll 1
    for (int t = 0; t < CHK / 64; ++t) {
      const int key0 = 64 * t;
      if (key0 >= a.Lk) break;
      const int nsub = min(4, (a.Lk - key0 + 15) >> 4);
      const bool need_mask = (mrow != nullptr) || (key0 + 64 > a.Lk);
      const float *Kt = Ks0 + 64 * t * HD, *Vt = Vs0 + 64 * t * HD;
      const unsigned *dw = dead_s + 16 * t;
      if (nsub == 4) fwd_tile<4, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
      else if (nsub == 3) fwd_tile<3, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
      else if (nsub == 2) fwd_tile<2, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
      else fwd_tile<1, DROP, AR>(Kt, Vt, dw, need_mask, qreg, q16, c, g, key0, rowbase, dc, m, lsum, o);
    }
  }
  lsum = grp_sum(lsum);
  o[2] = grp_sum4(o[2]);
  if (qvalid) {
    const float inv = dc.inv_keep / lsum;          // all keys masked -> NaN, like the reference
    float *orow = a.o + (long)b * a.o_sb + (long)qi * a.o_sl + h * HD;
    *reinterpret_cast<float4 *>(orow + 4 * g) = make_float4(o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv);
    *reinterpret_cast<float4 *>(orow + 16 + 4 * g) = make_float4(o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv);
    if (g == 0) {
      *reinterpret_cast<float4 *>(orow + 32) = make_float4(o[2][0] * inv, o[2][1] * inv, o[2][2] * inv, o[2][3] * inv);
      a.lse[(long)bh * a.Lq + qi] = (m + __builtin_amdgcn_logf(lsum)) * LN2;
    }
  }
}

// ======================================================================================= backward ==========
// KSUB key sub-tiles (16 keys each) x QG query groups = NW waves; QC queries per chunk, NBUF chunk buffers.
//   phase A  wave (ks, qg): for the chunk's 16-query sub-tiles j = qg, qg + QG, ...: S = Q K^T, dP = dO V^T (18 MFMA),
//            P, dS; dV^T += dO^T P, dK^T += Q^T dS (24 MFMA); dS goes to the LDS tile DS[query][key of the block]
//   barrier
//   phase B  work item (j, n) = one 16-dim x 16-query tile of dQ^T = K^T dS^T, contracted over ALL keys of the block
//            (K^T from the block's K rows in LDS, dS^T as 16-byte reads of DS): 4 x live sub-tiles MFMAs per item, two
//            interleaved accumulators; no partial sums, no atomics (ds_add_f32 from 16 waves measured 3x the whole
//            kernel, scattered global fp32 atomics worse) -- written straight to dQ, or to this key block's partial
//   barrier
template <int KSUB, int QG, int QC, int NBUF, bool DROP, class AR>
__global__ __launch_bounds__(KSUB * QG * 64) void mha2_bwd_kernel(const Mha2Args a) {
  constexpr int NW = KSUB * QG, NT = NW * 64;
  constexpr int KB = 16 * KSUB;        // keys per block
  constexpr int DSS = KB + 4;          // dS tile row stride ([query][key])
  constexpr int NSUBQ = QC / 16;
  static_assert(QC % 16 == 0 && NSUBQ % QG == 0 && (NBUF == 1 || NBUF == 2) && NW <= 16, "config");
  static_assert(QG == 1 || KSUB * 64 * 24 <= NSUBQ * 16 * DSS, "dK/dV merge scratch must fit the dS tile");
  __shared__ __attribute__((aligned(16))) float Qs0[QC * HD];
  __shared__ __attribute__((aligned(16))) float Ds0[QC * HD];
  __shared__ __attribute__((aligned(16))) float Qs1[NBUF == 2 ? QC * HD : 4];
  __shared__ __attribute__((aligned(16))) float Ds1[NBUF == 2 ? QC * HD : 4];
  __shared__ __attribute__((aligned(16))) float lse_s[2][QC];
  __shared__ __attribute__((aligned(16))) float del_s[2][QC];
  __shared__ __attribute__((aligned(16))) float Kb[KB * HD];
  __shared__ __attribute__((aligned(16))) float DS[NSUBQ * 16 * DSS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int BH = a.B * a.H;
  const int bh = (int)(blockIdx.x % (unsigned)BH);
  const int rest = (int)(blockIdx.x / (unsigned)BH);
  const int kb = rest % a.n_kb, qsp = rest / a.n_kb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int qg = wave / KSUB, ks = wave - qg * KSUB;

  const int kblock0 = kb * KB;
  const int key0 = kblock0 + 16 * ks;
  const bool wave_live = key0 < a.Lk;                                   // uniform
  const int nks_live = min(KSUB, (a.Lk - kblock0 + 15) >> 4);           // live key sub-tiles of this block
  const int ki = key0 + c;
  const bool kvalid = ki < a.Lk;
  const bool kdead = !kvalid || (a.mask && a.mask[(long)b * a.Lk + (kvalid ? ki : 0)]);

  DropCfg dc = {0u, 0u, 1.f};
  if (DROP) dc = drop_cfg(a);
#ifdef EDA_MHA2_PROFILE
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prof_t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(prof_t));
#endif

  // the block's K rows -> LDS (operand of phase B), this wave's K / V rows -> registers (lane = key)
  const float *kbase = a.k + (long)b * a.k_sb + h * HD;
  for (int i = tid; i < KB * 9; i += NT) {
    const int row = i / 9, c4 = i - row * 9;
    const int kr = min(kblock0 + row, a.Lk - 1);
    *reinterpret_cast<float4 *>(Kb + row * HD + 4 * c4) =
        *reinterpret_cast<const float4 *>(kbase + (long)kr * a.k_sl + 4 * c4);
  }
  float kreg[KSTEPS], vreg[KSTEPS];
  if (wave_live) {
    const int kr = min(ki, a.Lk - 1);
    load_row_operand(kreg, kbase + (long)kr * a.k_sl, g);
    load_row_operand(vreg, a.v + (long)b * a.v_sb + (long)kr * a.v_sl + h * HD, g);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      kreg[s] = kvalid ? kreg[s] : 0.f;
      vreg[s] = kvalid ? vreg[s] * dc.inv_keep : 0.f;       // dP arrives already divided by the keep rate
    }
  } else {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) { kreg[s] = 0.f; vreg[s] = 0.f; }
  }
  Row16 k16 = {{0, 0, 0}}, v16 = {{0, 0, 0}};
  if constexpr (AR::is16) { k16 = pack_row<AR>(kreg); v16 = pack_row<AR>(vreg); }

  const int qbeg = qsp * a.q_per_wg;
  const int qend = min(a.Lq, qbeg + a.q_per_wg);
  const float *qbase = a.q + (long)b * a.q_sb + h * HD;
  const float *dbase = a.dout + (long)b * a.do_sb + h * HD;
  const float *obase = a.o + (long)b * a.o_sb + h * HD;
  const float sl2 = a.scale * LOG2E;
  // outputs: the tensors themselves, or this workgroup's dense partial (B, L, H*36) when the range is split
  const int D = a.H * HD;
  float *dq_out = a.dq + (long)b * a.dq_sb + h * HD;
  long dq_sl = a.dq_sl;
  if (a.n_kb > 1) { dq_out = a.dq_part + ((long)kb * a.B + b) * a.Lq * D + h * HD; dq_sl = D; }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_dq = __builtin_amdgcn_make_buffer_rsrc(dq_out, 0, a.Lq * D * 4, 0x00020000);

  // stage one chunk: lse (log2 domain, +inf for rows outside the range: their P is exactly 0) and
  // delta = rowsum(dO o O) with ordinary loads, then Q and dO by LDS-DMA
  auto stage = [&](float *Qd, float *Dd, float *lse_d, float *del_d, int q0, int w0, int nw) {
    // executed by waves [w0, w0 + nw).  The DMA goes first and ALL ordinary loads of the wave are issued before
    // the first one is consumed: one memory round trip for the whole stage instead of four in a row.
    constexpr int P = (QC * 9 + 63) / 64;
    dma_rows<QC>(Qd, qbase, a.q_sl, q0, a.Lq, wave - w0, nw, lane, 0);
    dma_rows<QC>(Dd, dbase, a.do_sl, q0, a.Lq, wave - w0, nw, lane, P);
    constexpr int MAXIT = (QC * 8 + 63) / 64;            // items per thread if a single wave stages (upper bound)
    const int nthr = 64 * nw, t0 = tid - 64 * w0;
    const int nit = (QC * 8 + nthr - 1) / nthr;
    float4 d4[4], o4[4], d8[4], o8[4];
    float ls[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      d4[it] = o4[it] = d8[it] = o8[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      ls[it] = 0.f;
      const int item = t0 + it * nthr;
      if (it < nit && item < QC * 8) {
        const int row = item >> 3, sub = item & 7;
        const int gr = min(q0 + row, a.Lq - 1);
        const float *dr = dbase + (long)gr * a.do_sl, *orow = obase + (long)gr * a.o_sl;
        d4[it] = *reinterpret_cast<const float4 *>(dr + 4 * sub);
        o4[it] = *reinterpret_cast<const float4 *>(orow + 4 * sub);
        if (sub == 0) {
          d8[it] = *reinterpret_cast<const float4 *>(dr + 32);
          o8[it] = *reinterpret_cast<const float4 *>(orow + 32);
          ls[it] = a.lse[(long)bh * a.Lq + gr];
        }
      }
    }
    static_assert(MAXIT <= 4 * 16, "stage: chunk too large");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int item = t0 + it * nthr;
      if (it < nit && item < QC * 8) {         // (whole 8-lane groups take this branch together)
        const int row = item >> 3, sub = item & 7;
        float part = d4[it].x * o4[it].x + d4[it].y * o4[it].y + d4[it].z * o4[it].z + d4[it].w * o4[it].w;
        part += d8[it].x * o8[it].x + d8[it].y * o8[it].y + d8[it].z * o8[it].z + d8[it].w * o8[it].w;
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        if (sub == 0) {
          const bool valid = q0 + row < qend;
          lse_d[row] = valid ? ls[it] * LOG2E : INFINITY;
          del_d[row] = valid ? part : 0.f;
        }
      }
    }
    // more than 4 items per thread (few staging waves, large chunk): the remainder, one at a time
    for (int item = t0 + 4 * nthr; item < QC * 8; item += nthr) {
      const int row = item >> 3, sub = item & 7;
      const int gr = min(q0 + row, a.Lq - 1);
      const float *dr = dbase + (long)gr * a.do_sl, *orow = obase + (long)gr * a.o_sl;
      const float4 x = *reinterpret_cast<const float4 *>(dr + 4 * sub), y = *reinterpret_cast<const float4 *>(orow + 4 * sub);
      float part = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      if (sub == 0) {
        const float4 x8 = *reinterpret_cast<const float4 *>(dr + 32), y8 = *reinterpret_cast<const float4 *>(orow + 32);
        part += x8.x * y8.x + x8.y * y8.y + x8.z * y8.z + x8.w * y8.w;
      }
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      if (sub == 0) {
        const bool valid = q0 + row < qend;
        lse_d[row] = valid ? a.lse[(long)bh * a.Lq + gr] * LOG2E : INFINITY;
        del_d[row] = valid ? part : 0.f;
      }
    }
  };

  f32x4 dk[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 dv[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

  auto phase_a = [&](const float *Ql, const float *Dl, const float *lse_l, const float *del_l, int q0) {
    if (!wave_live) return;
    const int nsub = min(NSUBQ, (qend - q0 + 15) >> 4);
#pragma unroll 1
    for (int j = qg; j < nsub; j += QG) {
      // the waves of a SIMD share its matrix pipe; arbitration is priority, then age: a wave that is BEHIND
      // (fewer sub-tiles done) gets the higher priority, so the four finish a phase together instead of one after
      // the other (the last one alone on the pipe runs at half the rate)
      if (a.prio_mode == 1) {
        const int left = (nsub - 1 - j) / QG;          // sub-tiles after this one
        if (left >= 3) __builtin_amdgcn_s_setprio(3);
        else if (left == 2) __builtin_amdgcn_s_setprio(2);
        else if (left == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      f32x4 sacc = {0, 0, 0, 0}, pacc = {0, 0, 0, 0};
      {
        float qa[KSTEPS], da[KSTEPS];
        load_row_operand(qa, Ql + (16 * j + c) * HD, g);
        load_row_operand(da, Dl + (16 * j + c) * HD, g);
        if constexpr (AR::is16) {
          sacc = dot36<AR>(pack_row<AR>(qa), k16, sacc);
          pacc = dot36<AR>(pack_row<AR>(da), v16, pacc);
        } else {
#pragma unroll
          for (int s = 0; s < KSTEPS; ++s) {
            sacc = mfma4(qa[s], kreg[s], sacc);          // S[query 4g+r][key c]
            pacc = mfma4(da[s], vreg[s], pacc);          // dP / keep
          }
        }
      }
      const float4 lse4 = *reinterpret_cast<const float4 *>(lse_l + 16 * j + 4 * g);
      const float4 del4 = *reinterpret_cast<const float4 *>(del_l + 16 * j + 4 * g);
      const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
      const float del_r[4] = {del4.x, del4.y, del4.z, del4.w};
      // dropout bits: lanes c and c^1 (keys 2m, 2m+1) need the same four pair hashes (one per query r);
      // each computes two of them and they swap through a quad-permute DPP move
      unsigned h16[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
      if (DROP) {
        const int odd = c & 1;
        const unsigned qa0 = (unsigned)(q0 + 16 * j + 4 * g + 2 * odd);
        const unsigned kev = (unsigned)(ki & ~1);
        const unsigned hx = hash32(dc.seed ^ (((unsigned)bh * (unsigned)a.Lq + qa0) * (unsigned)a.Lk + kev));
        const unsigned hy = hash32(dc.seed ^ (((unsigned)bh * (unsigned)a.Lq + qa0 + 1u) * (unsigned)a.Lk + kev));
        const unsigned nx = (unsigned)__builtin_amdgcn_mov_dpp((int)hx, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        const unsigned ny = (unsigned)__builtin_amdgcn_mov_dpp((int)hy, 0xB1, 0xf, 0xf, true);
        const unsigned h0 = odd ? nx : hx, h1 = odd ? ny : hy, h2 = odd ? hx : nx, h3 = odd ? hy : ny;
        const int sh = 16 * odd;
        h16[0] = (h0 >> sh) & 0xffffu; h16[1] = (h1 >> sh) & 0xffffu;
        h16[2] = (h2 >> sh) & 0xffffu; h16[3] = (h3 >> sh) & 0xffffu;
      }
      f32x4 pd, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], sl2, -lse_r[r]));     // rows outside the range: exp2(-inf) = 0
        p = kdead ? 0.f : p;
        float dp = pacc[r], pk = p;
        if (DROP) {
          const bool keep = h16[r] >= dc.thresh;
          dp = keep ? dp : 0.f;
          pk = keep ? p : 0.f;
        }
        pd[r] = pk;
        ds[r] = p * (dp - del_r[r]);
      }
      // dS[query 4g+r][key c] -> the block's dS tile
      float *dsw = DS + (16 * j + 4 * g) * DSS + 16 * ks + c;
      dsw[0] = ds[0]; dsw[DSS] = ds[1]; dsw[2 * DSS] = ds[2]; dsw[3 * DSS] = ds[3];
      // dV^T[dim][key] += dO^T pd ; dK^T[dim][key] += Q^T dS
      {
        ColOperand cd;
        load_col_operand(cd, Dl + 16 * j * HD, c, g);
        if constexpr (AR::is16) {
          const u64 pb = AR::pk4(pd[0], pd[1], pd[2], pd[3]);
          dv[0] = AR::mma16(AR::pk4(cd.v[0][0], cd.v[1][0], cd.v[2][0], cd.v[3][0]), pb, dv[0]);
          dv[1] = AR::mma16(AR::pk4(cd.v[0][1], cd.v[1][1], cd.v[2][1], cd.v[3][1]), pb, dv[1]);
          dv[2] = AR::mma44(AR::pk4(cd.v[0][2], cd.v[1][2], cd.v[2][2], cd.v[3][2]), pb, dv[2]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float pb = pd[t];
            dv[0] = mfma4(cd.v[t][0], pb, dv[0]);
            dv[1] = mfma4(cd.v[t][1], pb, dv[1]);
            dv[2] = mfma44(cd.v[t][2], pb, dv[2]);
          }
        }
      }
      {
        ColOperand cq;
        load_col_operand(cq, Ql + 16 * j * HD, c, g);
        if constexpr (AR::is16) {
          const u64 sb = AR::pk4(ds[0], ds[1], ds[2], ds[3]);
          dk[0] = AR::mma16(AR::pk4(cq.v[0][0], cq.v[1][0], cq.v[2][0], cq.v[3][0]), sb, dk[0]);
          dk[1] = AR::mma16(AR::pk4(cq.v[0][1], cq.v[1][1], cq.v[2][1], cq.v[3][1]), sb, dk[1]);
          dk[2] = AR::mma44(AR::pk4(cq.v[0][2], cq.v[1][2], cq.v[2][2], cq.v[3][2]), sb, dk[2]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float sb = ds[t];
            dk[0] = mfma4(cq.v[t][0], sb, dk[0]);
            dk[1] = mfma4(cq.v[t][1], sb, dk[1]);
            dk[2] = mfma44(cq.v[t][2], sb, dk[2]);
          }
        }
      }
    }
  };

  // dQ^T tile (dims of column tile n) x (queries of sub-tile j), contracted over the block's live keys.  Items are
  // ordered full tiles first (n = 0, 1: 4 MFMAs per key sub-tile), then the 4-dim tiles (n = 2: four 4x4x1 steps,
  // a quarter of the time), so that with 16 waves every SIMD gets two full items and one short one.
  auto phase_b = [&](int q0) {
    const int nsub = min(NSUBQ, (qend - q0 + 15) >> 4);
    const int nitems = 3 * nsub;
#pragma unroll 1
    for (int it = wave; it < nitems; it += NW) {
      const bool full = it < 2 * nsub;
      const int j = full ? it >> 1 : it - 2 * nsub, n = full ? it & 1 : 2;
      const int col = full ? 16 * n + c : 32 + (c & 3);
      const float *kp = Kb + 4 * g * HD + col;                 // K[key 16s + 4g + t][col]
      const float *dp_ = DS + (16 * j + c) * DSS + 4 * g;      // dS[query c][key 16s + 4g + t]
      f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      int s = 0;
      if constexpr (AR::is16) {
        for (; s < nks_live; ++s) {
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(dp_ + 16 * s);
          const float *k0 = kp + 16 * s * HD;
          const u64 ka = AR::pk4(k0[0], k0[HD], k0[2 * HD], k0[3 * HD]), kb_ = AR::pk4(b0[0], b0[1], b0[2], b0[3]);
          if (full) { if (s & 1) acc1 = AR::mma16(ka, kb_, acc1); else acc0 = AR::mma16(ka, kb_, acc0); }
          else { if (s & 1) acc1 = AR::mma44(ka, kb_, acc1); else acc0 = AR::mma44(ka, kb_, acc0); }
        }
      } else if (full) {
        for (; s + 1 < nks_live; s += 2) {
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(dp_ + 16 * s);
          const f32x4 b1 = *reinterpret_cast<const f32x4 *>(dp_ + 16 * s + 16);
          const float *k0 = kp + 16 * s * HD, *k1 = k0 + 16 * HD;
          const float a00 = k0[0], a01 = k0[HD], a02 = k0[2 * HD], a03 = k0[3 * HD];
          const float a10 = k1[0], a11 = k1[HD], a12 = k1[2 * HD], a13 = k1[3 * HD];
          acc0 = mfma4(a00, b0[0], acc0); acc1 = mfma4(a10, b1[0], acc1);
          acc0 = mfma4(a01, b0[1], acc0); acc1 = mfma4(a11, b1[1], acc1);
          acc0 = mfma4(a02, b0[2], acc0); acc1 = mfma4(a12, b1[2], acc1);
          acc0 = mfma4(a03, b0[3], acc0); acc1 = mfma4(a13, b1[3], acc1);
        }
        if (s < nks_live) {
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(dp_ + 16 * s);
          const float *k0 = kp + 16 * s * HD;
          acc0 = mfma4(k0[0], b0[0], acc0);
          acc0 = mfma4(k0[HD], b0[1], acc0);
          acc0 = mfma4(k0[2 * HD], b0[2], acc0);
          acc0 = mfma4(k0[3 * HD], b0[3], acc0);
        }
      } else {
        for (; s + 1 < nks_live; s += 2) {
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(dp_ + 16 * s);
          const f32x4 b1 = *reinterpret_cast<const f32x4 *>(dp_ + 16 * s + 16);
          const float *k0 = kp + 16 * s * HD, *k1 = k0 + 16 * HD;
          const float a00 = k0[0], a01 = k0[HD], a02 = k0[2 * HD], a03 = k0[3 * HD];
          const float a10 = k1[0], a11 = k1[HD], a12 = k1[2 * HD], a13 = k1[3 * HD];
          acc0 = mfma44(a00, b0[0], acc0); acc1 = mfma44(a10, b1[0], acc1);
          acc0 = mfma44(a01, b0[1], acc0); acc1 = mfma44(a11, b1[1], acc1);
          acc0 = mfma44(a02, b0[2], acc0); acc1 = mfma44(a12, b1[2], acc1);
          acc0 = mfma44(a03, b0[3], acc0); acc1 = mfma44(a13, b1[3], acc1);
        }
        if (s < nks_live) {
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(dp_ + 16 * s);
          const float *k0 = kp + 16 * s * HD;
          acc0 = mfma44(k0[0], b0[0], acc0);
          acc0 = mfma44(k0[HD], b0[1], acc0);
          acc0 = mfma44(k0[2 * HD], b0[2], acc0);
          acc0 = mfma44(k0[3 * HD], b0[3], acc0);
        }
      }
      f32x4 r = acc0 + acc1;
      if (!full) r = grp_sum4(r);                              // the four lane groups hold partial sums over their key quarters
      const int gq = q0 + 16 * j + c;
      if (gq < qend && (full || g == 0)) {
        const float sc = a.scale;
        const f32x4 val = {r[0] * sc, r[1] * sc, r[2] * sc, r[3] * sc};
        if (a.n_kb > 1 && a.bwd_merge)   // this key block's partial: WRITE-THROUGH, the last-arriving key block of the rows merges (below)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rs_dq, (gq * D + 16 * n + 4 * g) * 4, 0, 16);
        else
          *reinterpret_cast<f32x4 *>(dq_out + (long)gq * dq_sl + 16 * n + 4 * g) = val;
      }
    }
  };

  const int nchunks = qend > qbeg ? (qend - qbeg + QC - 1) / QC : 0;
  // With 16 waves and <= 12 phase-B items the last four waves have no dQ tile: they stage the NEXT chunk (lse, delta,
  // DMA) during phase B instead of everybody doing it in front of phase A.
  constexpr bool STAGE_IN_B = NBUF == 2 && NW == 16 && 3 * NSUBQ <= 12;
  if (nchunks > 0) stage(Qs0, Ds0, lse_s[0], del_s[0], qbeg, 0, NW);
  EDA_SYNC_DMA();
  PSTAMP(0);
  for (int ci = 0; ci < nchunks; ci += NBUF) {
    const int q0 = qbeg + ci * QC;
    if (NBUF == 2) {
      if (!STAGE_IN_B && ci + 1 < nchunks) stage(Qs1, Ds1, lse_s[1], del_s[1], q0 + QC, 0, NW);
      PSTAMP(1);
      phase_a(Qs0, Ds0, lse_s[0], del_s[0], q0);
      PSTAMP(2);
      EDA_SYNC_DMA();
      PSTAMP(3);
      if (STAGE_IN_B && ci + 1 < nchunks && wave >= 12) stage(Qs1, Ds1, lse_s[1], del_s[1], q0 + QC, 12, 4);
      phase_b(q0);
      PSTAMP(4);
      EDA_SYNC_DMA();
      PSTAMP(5);
      if (ci + 1 >= nchunks) break;
      if (!STAGE_IN_B && ci + 2 < nchunks) stage(Qs0, Ds0, lse_s[0], del_s[0], q0 + 2 * QC, 0, NW);
      PSTAMP(1);
      phase_a(Qs1, Ds1, lse_s[1], del_s[1], q0 + QC);
      PSTAMP(2);
      EDA_SYNC_DMA();
      PSTAMP(3);
      if (STAGE_IN_B && ci + 2 < nchunks && wave >= 12) stage(Qs0, Ds0, lse_s[0], del_s[0], q0 + 2 * QC, 12, 4);
      phase_b(q0 + QC);
      PSTAMP(4);
      EDA_SYNC_DMA();
      PSTAMP(5);
    } else {
      phase_a(Qs0, Ds0, lse_s[0], del_s[0], q0);
      PSTAMP(2);
      EDA_SYNC_DMA();
      PSTAMP(3);
      phase_b(q0);
      PSTAMP(4);
      if (ci + 1 < nchunks) stage(Qs0, Ds0, lse_s[0], del_s[0], q0 + QC, 0, NW);      // (Q / dO of this chunk are dead)
      PSTAMP(1);
      EDA_SYNC_DMA();
      PSTAMP(5);
    }
  }

  // dK / dV: merge the QG query groups of a key sub-tile through LDS (the dS tile is dead), then store
  if (QG > 1) {
    for (int r_ = 1; r_ < QG; ++r_) {
      float *slot = DS + (ks * 64 + lane) * 24;
      if (qg == r_ && wave_live) {
        *reinterpret_cast<f32x4 *>(slot) = dk[0]; *reinterpret_cast<f32x4 *>(slot + 4) = dk[1];
        *reinterpret_cast<f32x4 *>(slot + 8) = dk[2]; *reinterpret_cast<f32x4 *>(slot + 12) = dv[0];
        *reinterpret_cast<f32x4 *>(slot + 16) = dv[1]; *reinterpret_cast<f32x4 *>(slot + 20) = dv[2];
      }
      __syncthreads();
      if (qg == 0 && wave_live) {
        dk[0] += *reinterpret_cast<const f32x4 *>(slot); dk[1] += *reinterpret_cast<const f32x4 *>(slot + 4);
        dk[2] += *reinterpret_cast<const f32x4 *>(slot + 8); dv[0] += *reinterpret_cast<const f32x4 *>(slot + 12);
        dv[1] += *reinterpret_cast<const f32x4 *>(slot + 16); dv[2] += *reinterpret_cast<const f32x4 *>(slot + 20);
      }
      __syncthreads();
    }
  }
  if (wave_live && qg == 0) { dk[2] = grp_sum4(dk[2]); dv[2] = grp_sum4(dv[2]); }
  const long kv_per = (long)a.B * a.Lk * D;                 // one tensor of one query split's dense partial
  if (kvalid && wave_live && qg == 0) {
    const float sc = a.scale, ik = dc.inv_keep;
    if (a.n_qs > 1 && a.bwd_merge) {   // dense partial of this query split: [split][dk | dv][B][Lk][D], WRITE-THROUGH (merged below)
      const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
          a.dkv_part + (long)qsp * 2 * kv_per + ((long)b * a.Lk + ki) * D + h * HD, 0, HD * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
          a.dkv_part + (long)qsp * 2 * kv_per + kv_per + ((long)b * a.Lk + ki) * D + h * HD, 0, HD * 4, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dk[0] * sc), rk, 16 * g, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dk[1] * sc), rk, 64 + 16 * g, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dv[0] * ik), rv, 16 * g, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dv[1] * ik), rv, 64 + 16 * g, 0, 16);
      if (g == 0) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dk[2] * sc), rk, 128, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dv[2] * ik), rv, 128, 0, 16);
      }
    } else {
      float *ok = a.dk + (long)b * a.dk_sb + (long)ki * a.dk_sl + h * HD;
      float *ov = a.dv + (long)b * a.dv_sb + (long)ki * a.dv_sl + h * HD;
      if (a.n_qs > 1) {          // (merged by mha2_part_reduce_kernel)
        ok = a.dkv_part + (long)qsp * 2 * kv_per + ((long)b * a.Lk + ki) * D + h * HD;
        ov = ok + kv_per;
      }
      *reinterpret_cast<f32x4 *>(ok + 4 * g) = dk[0] * sc;
      *reinterpret_cast<f32x4 *>(ok + 16 + 4 * g) = dk[1] * sc;
      *reinterpret_cast<f32x4 *>(ov + 4 * g) = dv[0] * ik;
      *reinterpret_cast<f32x4 *>(ov + 16 + 4 * g) = dv[1] * ik;
      if (g == 0) {
        *reinterpret_cast<f32x4 *>(ok + 32) = dk[2] * sc;
        *reinterpret_cast<f32x4 *>(ov + 32) = dv[2] * ik;
      }
    }
  }
  // ---- cross-workgroup merge of the split ranges (round 6; until round 5 a second launch, mha2_part_reduce_kernel, summed the
  // dense partials: 36 launches x 5 us per training step).  Same hand-off as the key-split forward: the partials left with
  // write-through (sc1) stores, every wave drains its stores, one lane takes the range's ticket, and the LAST arriver sums
  // all partials of the range IN SPLIT ORDER from sc1 loads (so the bits do not depend on who is last) and writes the
  // gradient.  dQ rows [qbeg, qend) of head h: ticket (bh, query split), n_kb arrivals.  dK / dV rows of key block kb: ticket
  // (bh, key block), n_qs arrivals.
  if (a.bwd_merge && (a.n_kb > 1 || a.n_qs > 1)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned *flag = reinterpret_cast<unsigned *>(lse_s);
    if (tid == 0) {
      unsigned lq = 0u, lkv = 0u;
      if (a.n_kb > 1) {
        unsigned *tk = a.bwd_tickets + (long)bh * a.n_qs + qsp;
        lq = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.n_kb - 1u;
        if (lq) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // re-armed
      }
      if (a.n_qs > 1) {
        unsigned *tk = a.bwd_tickets + (long)BH * a.n_qs + (long)bh * a.n_kb + kb;
        lkv = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.n_qs - 1u;
        if (lkv) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      flag[0] = lq; flag[1] = lkv;
    }
    __syncthreads();
    // the merging workgroup is ALONE on this range and everybody else may already have left the chip: the sum is latency,
    // so every thread keeps MU items x up to 4 splits = 16 sixteen-byte loads in flight
    constexpr int MU = 4;
    auto merge = [&](const float *p0, long zstride, int nz, int rows, int row0, float *fin, long fin_sl, int span_bytes) {
      const int nitems = rows * 9;
      for (int base = tid; base < nitems; base += NT * MU) {
        f32x4 t[MU];
        int off[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
          const int i = min(base + u * NT, nitems - 1);
          const int row = i / 9, c4 = i - row * 9;
          off[u] = ((row0 + row) * D + 4 * c4) * 4;
          t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 4
        for (int z = 0; z < nz; ++z) {
          const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p0 + z * zstride), 0, span_bytes, 0x00020000);
#pragma unroll
          for (int u = 0; u < MU; ++u) t[u] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off[u], 0, 16));
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
          const int i = base + u * NT;
          if (i < nitems) {
            const int row = i / 9, c4 = i - row * 9;
            *reinterpret_cast<f32x4 *>(fin + (long)(row0 + row) * fin_sl + 4 * c4) = t[u];
          }
        }
      }
    };
    if (flag[0])
      merge(a.dq_part + (long)b * a.Lq * D + h * HD, (long)a.B * a.Lq * D, a.n_kb, qend - qbeg, qbeg,
            a.dq + (long)b * a.dq_sb + h * HD, a.dq_sl, a.Lq * D * 4);
    if (flag[1]) {
      const int nkeys = min(KB, a.Lk - kblock0);
      const float *p0 = a.dkv_part + (long)b * a.Lk * D + h * HD;
      merge(p0, 2 * kv_per, a.n_qs, nkeys, kblock0, a.dk + (long)b * a.dk_sb + h * HD, a.dk_sl, a.Lk * D * 4);
      merge(p0 + kv_per, 2 * kv_per, a.n_qs, nkeys, kblock0, a.dv + (long)b * a.dv_sb + h * HD, a.dv_sl, a.Lk * D * 4);
    }
  }
#ifdef EDA_MHA2_PROFILE
  PSTAMP(6);
  prof_acc[7] = 1;
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&mha2_prof[((blockIdx.x * 16 + wave) & 63) * 8 + i], prof_acc[i]);
#endif
}

// out tensors (1: dq; 2: dk, dv) = sum over the splits of the dense partials [split][tensor][B][L][D], written
// with the outputs' strides, in split order (deterministic).  The second launch of a split backward whose ranges are too
// large for the in-launch merge (bwd_merge_in_launch below).
__global__ __launch_bounds__(256) void mha2_part_reduce_kernel(const float *__restrict__ part, int nsplit, int ntens,
                                                               int B, int L, int D, float *__restrict__ o0, long o0_sb,
                                                               long o0_sl, float *__restrict__ o1, long o1_sb,
                                                               long o1_sl) {
  const long per = (long)B * L * D, n4 = per / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ntens * n4) return;
  const int which = i >= n4;
  const long e = (i - which * n4) * 4;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < nsplit; ++z) {
    const float4 v = *reinterpret_cast<const float4 *>(part + ((long)z * ntens + which) * per + e);
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  const long row = e / D;
  const int col = (int)(e - row * D);
  const long bb = row / L, l = row - bb * L;
  float *o = which ? o1 + bb * o1_sb + l * o1_sl + col : o0 + bb * o0_sb + l * o0_sl + col;
  *reinterpret_cast<float4 *>(o) = t;
}

template <int NQ, int KS, int CHK, int NBUF, bool SPLIT = false>
int launch_fwd(Mha2Args &a, hipStream_t stream) {
  a.n_qs = (a.Lq + 16 * NQ - 1) / (16 * NQ);
  const unsigned grid = (unsigned)(a.B * a.H * a.n_qs * (SPLIT ? a.n_ksplit : 1));
  const dim3 g(grid), b(NQ * KS * 64);
  const bool drop = a.p_drop > 0.f;
#define EDA_FWD(AR)                                                                                               \
  do {                                                                                                            \
    if (drop) hipLaunchKernelGGL((mha2_fwd_kernel<NQ, KS, CHK, NBUF, true, AR, SPLIT>), g, b, 0, stream, a);      \
    else hipLaunchKernelGGL((mha2_fwd_kernel<NQ, KS, CHK, NBUF, false, AR, SPLIT>), g, b, 0, stream, a);          \
  } while (0)
  if (a.dtype == EDA_DTYPE_BF16) EDA_FWD(ArBf16);
  else if (a.dtype == EDA_DTYPE_F16) EDA_FWD(ArFp16);
  else EDA_FWD(ArF32);
#undef EDA_FWD
  EDA_CHECK_LAUNCH();
  return 0;
}

// Key-split plan of a forward shape (ns <= 1: none).  Short query sets against >= 512 keys -- the text -> point cross
// attention, 80 / 130 queries x 1024 keys: 128 / 192 workgroups, each staging a head's whole 295 KB of K | V -- are
// divided over key splits of whole 256-key chunks until every CU holds a workgroup.  Measured (rocprofv3, B = 8,
// profiles/r05_mha_ksplit.txt): 80 x 1024 unsplit 34.7 us, 2 splits 24.7, 4 splits 30.7 (4 splits with release / acquire
// fences instead of write-through stores: 34); one 80-query block of five sub-tiles instead of two 64-query blocks:
// 31.3 / 25.5 us in 2 / 4 splits -- no better, not kept.  256 queries x 1024 keys stays unsplit: 256 workgroups already
// fill the chip and a second round of workgroups costs more than the shorter staging returns (37.9 us unsplit, 49.9 /
// 65.9 in 2 / 4 splits).
// EDA_MHA2_KSPLIT=0 switches it off, =n forces n splits wherever the shape allows one (<= 256 queries, >= 512 keys).
constexpr int FWD_SPLIT_CHK = 256, FWD_SPLIT_NQ = 4;
struct FwdSplit { int ns, nq; };
FwdSplit fwd_ksplit(int B, int H, int Lq, int Lk) {
  FwdSplit p = {1, FWD_SPLIT_NQ};
  const long env = eda_knob(EDA_K_MHA2_KSPLIT);
  if (env == 0 || Lk < 2 * FWD_SPLIT_CHK || Lq > 256) return p;
  const int chunks = (Lk + FWD_SPLIT_CHK - 1) / FWD_SPLIT_CHK;
  const long blocks = (long)B * H * ((Lq + 16 * p.nq - 1) / (16 * p.nq));
  long want = env > 0 ? env : (blocks >= 224 ? 1 : (256 + blocks - 1) / blocks);
  if (want > chunks) want = chunks;
  if (want < 2) return p;
  const int cps = (int)((chunks + want - 1) / want);          // chunks per split
  p.ns = (chunks + cps - 1) / cps;
  return p;
}

// Who sums the split ranges' partials: the last-arriving workgroup of a range inside the backward launch (1), or a second
// launch over the whole chip (0).  The in-launch merge is ONE workgroup reading n partials of its range while the rest of
// the chip has gone idle, behind write-through partial stores that every workgroup has to drain: it wins while a range is a few
// tens of KB (80 x 1024: 43.3 -> 41.1 us and one launch boundary less), loses when it is hundreds (256 x 256: 36.8 -> 46.5 us,
// 1024 x 1024: 288.8 -> 297.7; profiles/r06_mha_bwd_merge.txt -- the same finding as the key-split forward's, whose partials
// are a few KB).  EDA_MHA2_BWD_MERGE = 0 / 1 forces never / always.
struct BwdPlan;
int bwd_merge_in_launch(const BwdPlan &p, int Lq, int Lk, int kb_keys);

// Backward decomposition of a shape: which kernel variant, how many key blocks / query splits.
constexpr long BWD_MERGE_MAX_BYTES = 64 << 10;
struct BwdPlan { int variant, n_kb, n_qs, q_per_wg, merge; };
BwdPlan bwd_plan(int B, int H, int Lq, int Lk) {
  const long BH = (long)B * H;
  BwdPlan p = {0, 1, 1, 0, 0};
  int ksub, qc;
  if (Lk <= 80) { p.variant = 2; ksub = 5; qc = 96; }          // text tokens: 5 key waves x 3 query groups
  else if (Lk <= 144) { p.variant = 1; ksub = 9; qc = 64; }    // detected boxes / long utterances: 9 key waves
  else { p.variant = 0; ksub = 16; qc = 64; }
  p.n_kb = (Lk + 16 * ksub - 1) / (16 * ksub);
  if (p.n_kb < 1) p.n_kb = 1;
  const int chunks = Lq > 0 ? (Lq + qc - 1) / qc : 1;
  // enough workgroups to fill 256 CUs, but no more query splits than chunks
  // (one workgroup per CU and ONE round of them: the short-key variants planned for 512 until round 5 -- 1024 x 80 then ran
  //  384 workgroups of 15 waves in two rounds, 54.3 us; 256: 41.6 us, 1024 x 132 67.8 -> 61.6; profiles/r05_mha_bwd_wgs.txt)
  const long want_wgs = 256;
  long n_qs = (want_wgs + BH * p.n_kb - 1) / (BH * p.n_kb);
  if (n_qs < 1) n_qs = 1;
  if (n_qs > chunks) n_qs = chunks;
  const int cpw = (int)((chunks + n_qs - 1) / n_qs);            // chunks per workgroup
  p.q_per_wg = cpw * qc;
  p.n_qs = (Lq + p.q_per_wg - 1) / p.q_per_wg;
  if (p.n_qs < 1) p.n_qs = 1;
  p.merge = bwd_merge_in_launch(p, Lq, Lk, 16 * ksub);
  return p;
}

int bwd_merge_in_launch(const BwdPlan &p, int Lq, int Lk, int kb_keys) {
  if (p.n_kb <= 1 && p.n_qs <= 1) return 0;
  const long env = eda_knob(EDA_K_MHA2_BWD_MERGE);
  if (env >= 0) return env != 0;
  // bytes the largest merging workgroup reads: n_kb partials of its query range (dQ), n_qs partials of its key block (dK | dV)
  const long dq_bytes = p.n_kb > 1 ? (long)p.n_kb * (p.q_per_wg < Lq ? p.q_per_wg : Lq) * HD * 4 : 0;
  const long kv_bytes = p.n_qs > 1 ? (long)p.n_qs * 2 * (kb_keys < Lk ? kb_keys : Lk) * HD * 4 : 0;
  return (dq_bytes > kv_bytes ? dq_bytes : kv_bytes) <= BWD_MERGE_MAX_BYTES;
}

size_t bwd_workspace_floats(const BwdPlan &p, int B, int H, int Lq, int Lk) {
  size_t n = 0;
  if (p.n_kb > 1) n += (size_t)p.n_kb * B * Lq * H * HD;
  if (p.n_qs > 1) n += (size_t)p.n_qs * 2 * B * Lk * H * HD;
  return n;
}
// arrival tickets of the split ranges: [B*H][n_qs] for dQ, then [B*H][n_kb] for dK | dV (zero before a launch, left zero)
size_t bwd_ticket_words(const BwdPlan &p, int B, int H) {
  return p.merge ? (size_t)B * H * (p.n_qs + p.n_kb) : 0;
}

template <int KSUB, int QG, int QC, int NBUF>
int launch_bwd(Mha2Args &a, hipStream_t stream) {
  const unsigned grid = (unsigned)(a.B * a.H * a.n_kb * a.n_qs);
  const dim3 g(grid), b(KSUB * QG * 64);
  const bool drop = a.p_drop > 0.f;
#define EDA_BWD(AR)                                                                                        \
  do {                                                                                                     \
    if (drop) hipLaunchKernelGGL((mha2_bwd_kernel<KSUB, QG, QC, NBUF, true, AR>), g, b, 0, stream, a);     \
    else hipLaunchKernelGGL((mha2_bwd_kernel<KSUB, QG, QC, NBUF, false, AR>), g, b, 0, stream, a);         \
  } while (0)
  if (a.dtype == EDA_DTYPE_BF16) EDA_BWD(ArBf16);
  else if (a.dtype == EDA_DTYPE_F16) EDA_BWD(ArFp16);
  else EDA_BWD(ArF32);
#undef EDA_BWD
  EDA_CHECK_LAUNCH();
  return 0;
}

}  // namespace

size_t eda_mha2_fwd_workspace_bytes(int B, int H, int Lq, int Lk) {
  if (B <= 0 || Lq <= 0 || Lk <= 0) return 0;
  if (eda_mha4_takes(EDA_DTYPE_F32, Lq, Lk)) return eda_mha4_fwd_workspace_bytes(B, H, Lq, Lk);
  const FwdSplit sp = fwd_ksplit(B, H, Lq, Lk);
  if (sp.ns <= 1) return 0;
  const size_t blocks = (size_t)B * H * ((Lq + 16 * sp.nq - 1) / (16 * sp.nq));
  const size_t tick = (blocks * sizeof(unsigned) + 255) / 256 * 256;
  return tick + blocks * sp.ns * sp.nq * 64 * 16 * sizeof(float);
}

// Shape -> configuration.  B*H = 64 in EDA; the chip has 256 CUs.
int eda_mha2_fwd_launch(Mha2Args &a, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (a.B == 0 || a.Lq == 0) return 0;
  const int BH = a.B * a.H;
  if (ws && eda_mha4_takes(a.dtype, a.Lq, a.Lk)) return eda_mha4_fwd_launch(a, ws, ws_bytes, stream);     // keys per wave (mha4.hip)
  {
    const FwdSplit sp = ws ? fwd_ksplit(a.B, a.H, a.Lq, a.Lk) : FwdSplit{1, 4};
    if (sp.ns > 1) {
      const size_t need = eda_mha2_fwd_workspace_bytes(a.B, a.H, a.Lq, a.Lk);
      EDA_CHECK_ARG(ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0,
                    "workspace of eda_mha_fwd_workspace_bytes() required (16-byte aligned, its ticket words zero)");
      const size_t blocks = (size_t)BH * ((a.Lq + 16 * sp.nq - 1) / (16 * sp.nq));
      const size_t tick = (blocks * sizeof(unsigned) + 255) / 256 * 256;
      a.fwd_tickets = reinterpret_cast<unsigned *>(ws);
      a.fwd_part = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(ws) + tick);
      a.n_ksplit = sp.ns;
      const int chunks = (a.Lk + FWD_SPLIT_CHK - 1) / FWD_SPLIT_CHK;
      a.keys_per_split = (chunks + sp.ns - 1) / sp.ns * FWD_SPLIT_CHK;
      const bool one = a.keys_per_split == FWD_SPLIT_CHK;
      return one ? launch_fwd<4, 4, FWD_SPLIT_CHK, 1, true>(a, stream) : launch_fwd<4, 4, FWD_SPLIT_CHK, 2, true>(a, stream);
    }
  }
  {
    const int rc3 = eda_mha3_fwd_launch(a, stream);                  // >= 512 keys, >= 192 queries: the bf16-pipe kernel
    if (rc3 >= 0) return rc3;
  }
  const bool short_q = (long)BH * ((a.Lq + 255) / 256) < 192;        // 16-query waves alone would leave CUs idle
  if (a.Lk <= 144) {
    if (short_q) return launch_fwd<4, 2, 192, 1>(a, stream);
    return launch_fwd<16, 1, 192, 1>(a, stream);
  }
  if (a.Lk <= 256) {
    if (short_q) return launch_fwd<4, 4, 256, 1>(a, stream);
    return launch_fwd<16, 1, 256, 1>(a, stream);
  }
  if (short_q) return launch_fwd<4, 4, 256, 2>(a, stream);
  return launch_fwd<16, 1, 256, 2>(a, stream);
}

int eda_mha2_qproj_fwd_launch(Mha2Args &a, hipStream_t stream) {
  if (a.B == 0 || a.Lq == 0) return 0;
  const dim3 g((unsigned)(a.B * a.H * ((a.Lq + 63) / 64))), b(256);
  const bool drop = a.p_drop > 0.f;
#define EDA_QP(AR)                                                                                   \
  do {                                                                                               \
    if (drop) hipLaunchKernelGGL((mha2_qproj_fwd_kernel<true, AR>), g, b, 0, stream, a);             \
    else hipLaunchKernelGGL((mha2_qproj_fwd_kernel<false, AR>), g, b, 0, stream, a);                 \
  } while (0)
  if (a.dtype == EDA_DTYPE_BF16) EDA_QP(ArBf16);
  else if (a.dtype == EDA_DTYPE_F16) EDA_QP(ArFp16);
  else EDA_QP(ArF32);
#undef EDA_QP
  EDA_CHECK_LAUNCH();
  return 0;
}

#ifdef EDA_MHA2_PROFILE
extern "C" int eda_mha2_profile_read(unsigned long long *out16) {
  static unsigned long long all[64 * 8], zero[64 * 8];
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(mha2_prof), sizeof(all)) != hipSuccess) return 1;
  for (int i = 0; i < 16; ++i) out16[i] = 0;
  for (int s = 0; s < 64; ++s)
    for (int i = 0; i < 8; ++i) out16[i] += all[s * 8 + i];
  return hipMemcpyToSymbol(HIP_SYMBOL(mha2_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif

// The scratch of a backward: [tickets (eda_mha2_bwd_ticket_bytes, 256-byte rounded) | dense partials].  The ticket words must
// be ZERO before the launch and are left zero; a caller with a PERSISTENT zeroed ticket buffer passes it separately
// (eda_mha2_bwd_launch tickets != nullptr) and the kernel never needs a fill launch; with tickets == nullptr the leading
// ticket area of `ws` is zeroed by a fill kernel first.
size_t eda_mha2_bwd_ticket_bytes(int B, int H, int Lq, int Lk) {
  if (B <= 0 || Lq <= 0 || Lk <= 0) return 0;
  return (sizeof(unsigned) * bwd_ticket_words(bwd_plan(B, H, Lq, Lk), B, H) + 255) / 256 * 256;
}
size_t eda_mha2_bwd_workspace_bytes(int B, int H, int Lq, int Lk) {
  if (B <= 0 || Lq <= 0 || Lk <= 0) return 0;
  return eda_mha2_bwd_ticket_bytes(B, H, Lq, Lk) + sizeof(float) * bwd_workspace_floats(bwd_plan(B, H, Lq, Lk), B, H, Lq, Lk);
}

int eda_mha2_bwd_launch(Mha2Args &a, void *ws, size_t ws_bytes, unsigned *tickets, size_t tickets_bytes, hipStream_t stream) {
  if (a.B == 0) return 0;
  a.prio_mode = (int)eda_knob(EDA_K_MHA2_PRIO);
  const BwdPlan p = bwd_plan(a.B, a.H, a.Lq, a.Lk);
  a.n_kb = p.n_kb; a.n_qs = p.n_qs; a.q_per_wg = p.q_per_wg;
  const size_t tk = eda_mha2_bwd_ticket_bytes(a.B, a.H, a.Lq, a.Lk);
  const size_t need = tk + sizeof(float) * bwd_workspace_floats(p, a.B, a.H, a.Lq, a.Lk);
  EDA_CHECK_ARG(need == 0 || (ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0),
                "workspace of eda_mha_bwd_workspace_bytes() required (16-byte aligned)");
  EDA_CHECK_ARG(!tickets || tickets_bytes >= tk, "ticket buffer of eda_mha_bwd_ticket_bytes() required");
  float *w = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(ws) + tk);
  a.dq_part = nullptr; a.dkv_part = nullptr; a.bwd_tickets = nullptr;
  if (p.n_kb > 1) { a.dq_part = w; w += (size_t)p.n_kb * a.B * a.Lq * a.H * HD; }
  if (p.n_qs > 1) a.dkv_part = w;
  if (tk) {
    a.bwd_tickets = tickets ? tickets : reinterpret_cast<unsigned *>(ws);
    if (!tickets)
      if (int rc = eda_zero_async(ws, tk, stream)) return rc;
  }
  a.bwd_merge = p.merge;
  // the next chunk's Q / dO staged underneath this chunk's phases -- also for the short-key variants from round 6 on (1024 x 80:
  // 41.4 -> 40.3 us, 1024 x 132: 61.8 -> 61.0, one-chunk shapes unchanged; profiles/r06_mha_bwd_merge.txt)
  const bool dbuf = eda_knob(EDA_K_MHA2_BWD_DBUF) != 0;
  int rc;
  if (p.variant == 0) rc = launch_bwd<16, 1, 64, 2>(a, stream);
  else if (p.variant == 1) rc = dbuf ? launch_bwd<9, 1, 64, 2>(a, stream) : launch_bwd<9, 1, 64, 1>(a, stream);
  else rc = dbuf ? launch_bwd<5, 3, 96, 2>(a, stream) : launch_bwd<5, 3, 96, 1>(a, stream);
  if (rc || p.merge) return rc;
  const int D = a.H * HD;
  if (p.n_kb > 1) {
    const long items = (long)a.B * a.Lq * D / 4;
    hipLaunchKernelGGL(mha2_part_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                       a.dq_part, p.n_kb, 1, a.B, a.Lq, D, a.dq, a.dq_sb, a.dq_sl, a.dq, a.dq_sb, a.dq_sl);
    EDA_CHECK_LAUNCH();
  }
  if (p.n_qs > 1) {
    const long items = 2L * a.B * a.Lk * D / 4;
    hipLaunchKernelGGL(mha2_part_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                       a.dkv_part, p.n_qs, 2, a.B, a.Lk, D, a.dk, a.dk_sb, a.dk_sl, a.dv, a.dv_sb, a.dv_sl);
    EDA_CHECK_LAUNCH();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------- C entry points ----
namespace {

bool mult4(long v) { return (v & 3) == 0; }
bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// rows x D floats = value (degenerate shapes: no keys / no queries)
__global__ __launch_bounds__(256) void mha2_fill_rows_kernel(float *__restrict__ o, long sb, long sl, int B, int L, int D,
                                                             float value) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * L * D) return;
  const long row = i / D;
  const int col = (int)(i - row * D);
  const long bb = row / L, l = row - bb * L;
  o[bb * sb + l * sl + col] = value;
}
int fill_rows(float *o, long sb, long sl, int B, int L, int D, float value, hipStream_t stream) {
  const long n = (long)B * L * D;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(mha2_fill_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, o, sb, sl, B, L, D, value);
  EDA_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" size_t eda_mha_fwd_workspace_bytes(int B, int H, int Lq, int Lk) { return eda_mha2_fwd_workspace_bytes(B, H, Lq, Lk); }

extern "C" int eda_mha_fwd(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                           long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                           int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                           float *out, float *lse, int dtype, void *stream_) {
  return eda_mha_fwd_ws(q, k, v, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, key_padding_mask, B, H, Lq, Lk, head_dim, scale, p_drop,
                        seed_ptr, salt, out, lse, dtype, nullptr, 0, stream_);
}

extern "C" int eda_mha_fwd_ws(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                              long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                              int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                              float *out, float *lse, int dtype, void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(dtype == EDA_DTYPE_F32 || dtype == EDA_DTYPE_BF16 || dtype == EDA_DTYPE_F16,
                "dtype must be EDA_DTYPE_F32 / BF16 / F16");
  EDA_CHECK_ARG(head_dim == HD, "only head_dim 36 (d_model 288 / 8 heads) is built");
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "bad dimension");
  if (B == 0 || Lq == 0) return 0;
  EDA_CHECK_ARG(q && k && v && out && lse, "null pointer");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "bad dropout arguments");
  EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                    al16(q) && al16(k) && al16(v) && al16(out),
                "rows must be 16-byte aligned");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  if (Lk == 0) {            // no keys: P V is an empty sum (torch: bmm over a zero-length dimension = 0)
    if (int rc = fill_rows(out, (long)Lq * H * HD, (long)H * HD, B, Lq, H * HD, 0.f, stream)) return rc;
    return fill_rows(lse, (long)H * Lq, (long)Lq, B, H, Lq, -INFINITY, stream);
  }
  Mha2Args m = {};
  m.q = q; m.k = k; m.v = v; m.q_sb = q_sb; m.q_sl = q_sl; m.k_sb = k_sb; m.k_sl = k_sl;
  m.v_sb = v_sb; m.v_sl = v_sl; m.o = out; m.o_sb = (long)Lq * H * HD; m.o_sl = (long)H * HD;
  m.lse = lse; m.mask = key_padding_mask; m.B = B; m.H = H; m.Lq = Lq; m.Lk = Lk; m.scale = scale;
  m.p_drop = p_drop; m.seed_ptr = seed_ptr; m.salt = salt; m.dtype = dtype;
  return eda_mha2_fwd_launch(m, ws, ws_bytes, stream);
}

extern "C" int eda_mha_qproj_supported(int H, int head_dim, int Lk) { return H * head_dim == QP_D && head_dim == HD && Lk >= 1 && Lk <= 192; }

extern "C" int eda_mha_qproj_fwd(const float *x, long x_sb, long x_sl, const float *wq, long ldwq, const float *bq,
                                 const float *k, const float *v, long k_sb, long k_sl, long v_sb, long v_sl,
                                 const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk, int head_dim, float scale,
                                 float p_drop, const unsigned long long *seed_ptr, unsigned salt, float *q_out, long q_sb,
                                 long q_sl, float *out, float *lse, int dtype, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(dtype == EDA_DTYPE_F32 || dtype == EDA_DTYPE_BF16 || dtype == EDA_DTYPE_F16,
                "dtype must be EDA_DTYPE_F32 / BF16 / F16");
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0, "bad dimension");
  EDA_CHECK_ARG(eda_mha_qproj_supported(H, head_dim, Lk), "the fused launch exists for 8 heads x 36 and 1..192 keys");
  if (B == 0 || Lq == 0) return 0;
  EDA_CHECK_ARG(x && wq && k && v && q_out && out && lse, "null pointer");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "bad dropout arguments");
  EDA_CHECK_ARG(mult4(x_sb) && mult4(x_sl) && mult4(ldwq) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                    mult4(q_sb) && mult4(q_sl) && al16(x) && al16(wq) && al16(k) && al16(v) && al16(q_out) && al16(out) &&
                    (!bq || al16(bq)),
                "rows must be 16-byte aligned");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  Mha2Args m = {};
  m.q = q_out; m.q_sb = q_sb; m.q_sl = q_sl;
  m.k = k; m.v = v; m.k_sb = k_sb; m.k_sl = k_sl; m.v_sb = v_sb; m.v_sl = v_sl;
  m.o = out; m.o_sb = (long)Lq * H * HD; m.o_sl = (long)H * HD; m.lse = lse; m.mask = key_padding_mask;
  m.B = B; m.H = H; m.Lq = Lq; m.Lk = Lk; m.scale = scale; m.p_drop = p_drop; m.seed_ptr = seed_ptr; m.salt = salt;
  m.dtype = dtype;
  m.xq = x; m.xq_sb = x_sb; m.xq_sl = x_sl; m.wq = wq; m.ldwq = ldwq; m.bq = bq;
  m.q_out = q_out; m.qo_sb = q_sb; m.qo_sl = q_sl;
  return eda_mha2_qproj_fwd_launch(m, stream);
}

extern "C" int eda_mha_fwd_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl,
                               long k_sb, long k_sl, long v_sb, long v_sl,
                               const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                               int head_dim, float scale, float p_drop,
                               const unsigned long long *seed_ptr, unsigned salt, float *out,
                               float *lse, void *stream_) {
  return eda_mha_fwd(q, k, v, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, key_padding_mask, B, H, Lq, Lk, head_dim, scale,
                     p_drop, seed_ptr, salt, out, lse, EDA_DTYPE_F32, stream_);
}

extern "C" size_t eda_mha_bwd_workspace_bytes(int B, int H, int Lq, int Lk) {
  return eda_mha2_bwd_workspace_bytes(B, H, Lq, Lk);
}

// delta_ws: unused since round 3 (delta = rowsum(dO o O) is formed while a chunk is staged); may be NULL.
extern "C" size_t eda_mha_bwd_ticket_bytes(int B, int H, int Lq, int Lk) { return eda_mha2_bwd_ticket_bytes(B, H, Lq, Lk); }

extern "C" int eda_mha_bwd(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                           long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                           int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                           const float *out, const float *lse, const float *dout, long do_sb, long do_sl,
                           float *delta_ws, float *dq, float *dk, float *dv, long dq_sb, long dq_sl, long dk_sb,
                           long dk_sl, long dv_sb, long dv_sl, void *ws, size_t ws_bytes, int dtype, void *stream_) {
  return eda_mha_bwd_tk(q, k, v, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, key_padding_mask, B, H, Lq, Lk, head_dim, scale, p_drop,
                        seed_ptr, salt, out, lse, dout, do_sb, do_sl, delta_ws, dq, dk, dv, dq_sb, dq_sl, dk_sb, dk_sl, dv_sb,
                        dv_sl, ws, ws_bytes, nullptr, 0, dtype, stream_);
}

// eda_mha_bwd with the arrival tickets of the split ranges in a PERSISTENT buffer of the caller (zero before its first use;
// every launch leaves it zero): no fill launch in front of the kernel.  tickets == NULL: the leading
// eda_mha_bwd_ticket_bytes() of `ws` are zeroed by a fill kernel first (what eda_mha_bwd does).
extern "C" int eda_mha_bwd_tk(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                              long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                              int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                              const float *out, const float *lse, const float *dout, long do_sb, long do_sl,
                              float *delta_ws, float *dq, float *dk, float *dv, long dq_sb, long dq_sl, long dk_sb,
                              long dk_sl, long dv_sb, long dv_sl, void *ws, size_t ws_bytes, void *tickets,
                              size_t tickets_bytes, int dtype, void *stream_) {
  (void)delta_ws;
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(dtype == EDA_DTYPE_F32 || dtype == EDA_DTYPE_BF16 || dtype == EDA_DTYPE_F16,
                "dtype must be EDA_DTYPE_F32 / BF16 / F16");
  EDA_CHECK_ARG(head_dim == HD, "only head_dim 36 (d_model 288 / 8 heads) is built");
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "bad dimension");
  if (B == 0) return 0;
  EDA_CHECK_ARG(q && k && v && out && lse && dout && dq && dk && dv, "null pointer");
  EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                    mult4(do_sb) && mult4(do_sl) && mult4(dq_sb) && mult4(dq_sl) && mult4(dk_sb) &&
                    mult4(dk_sl) && mult4(dv_sb) && mult4(dv_sl) && al16(q) && al16(k) && al16(v) && al16(out) &&
                    al16(dout) && al16(dq) && al16(dk) && al16(dv),
                "rows must be 16-byte aligned");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  if (Lq == 0 || Lk == 0) {      // empty sums: every gradient that has rows is zero
    if (int rc = fill_rows(dq, dq_sb, dq_sl, B, Lq, H * HD, 0.f, stream)) return rc;
    if (int rc = fill_rows(dk, dk_sb, dk_sl, B, Lk, H * HD, 0.f, stream)) return rc;
    return fill_rows(dv, dv_sb, dv_sl, B, Lk, H * HD, 0.f, stream);
  }
  Mha2Args m = {};
  m.q = q; m.k = k; m.v = v; m.q_sb = q_sb; m.q_sl = q_sl; m.k_sb = k_sb; m.k_sl = k_sl;
  m.v_sb = v_sb; m.v_sl = v_sl; m.o = const_cast<float *>(out); m.o_sb = (long)Lq * H * HD; m.o_sl = (long)H * HD;
  m.lse = const_cast<float *>(lse); m.mask = key_padding_mask; m.B = B; m.H = H; m.Lq = Lq; m.Lk = Lk;
  m.scale = scale; m.p_drop = p_drop; m.seed_ptr = seed_ptr; m.salt = salt;
  m.dout = dout; m.do_sb = do_sb; m.do_sl = do_sl; m.dq = dq; m.dk = dk; m.dv = dv;
  m.dq_sb = dq_sb; m.dq_sl = dq_sl; m.dk_sb = dk_sb; m.dk_sl = dk_sl; m.dv_sb = dv_sb; m.dv_sl = dv_sl;
  m.dtype = dtype;
  return eda_mha2_bwd_launch(m, ws, ws_bytes, reinterpret_cast<unsigned *>(tickets), tickets_bytes, stream);
}

extern "C" int eda_mha_bwd_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl,
                               long k_sb, long k_sl, long v_sb, long v_sl,
                               const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                               int head_dim, float scale, float p_drop,
                               const unsigned long long *seed_ptr, unsigned salt, const float *out,
                               const float *lse, const float *dout, long do_sb, long do_sl,
                               float *delta_ws, float *dq, float *dk, float *dv, long dq_sb,
                               long dq_sl, long dk_sb, long dk_sl, long dv_sb, long dv_sl,
                               void *ws, size_t ws_bytes, void *stream_) {
  return eda_mha_bwd(q, k, v, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, key_padding_mask, B, H, Lq, Lk, head_dim, scale, p_drop,
                     seed_ptr, salt, out, lse, dout, do_sb, do_sl, delta_ws, dq, dk, dv, dq_sb, dq_sl, dk_sb, dk_sl, dv_sb,
                     dv_sl, ws, ws_bytes, EDA_DTYPE_F32, stream_);
}
