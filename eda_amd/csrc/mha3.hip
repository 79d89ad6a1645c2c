// mha3.hip -- attention forward on the gfx950 bf16 matrix instructions (v_mfma_f32_16x16x32_bf16) for the long-key sites
// (1024 point keys: the encoder's point self-attention and the decoder's query -> point cross-attention,
// models/encoder_decoder_layers.py:179-183, 393-399 -> F.multi_head_attention_forward), in two arithmetic modes:
//
//   NPL = 3  "bf16 x 3": every fp32 operand is split EXACTLY into three bf16 planes v = h + m + l (8 + 8 + 8 mantissa
//            bits) and a product a.b is formed from the six plane products whose weight is >= 2^-24 (hh, hm, mh, hl,
//            lh, mm), accumulated in fp32 -- fp32 accuracy (the same scheme as csrc/wgrad.hip's grouped weight gradient,
//            profiles/r04_experiments.md) on a pipe that runs 16 x the fp32 MFMA rate: the headline / parity path.
//   NPL = 1  plain bf16 operands, fp32 accumulation and softmax (BASELINE.json configs[2]): a separate bench line.
//
// What differs from mha2.hip's forward (whose decomposition, transposed formulation S^T = K Q^T, online softmax in the
// log2 domain, dropout hash and epilogue this kernel keeps):
//   * K and V of a 128-key chunk arrive as fp32 rows by LDS-DMA (double-buffered) and are converted ONCE per workgroup
//     into bf16 operand planes in LDS -- K row-major [key][40] (16-byte operand reads, conflict-free at the 80-byte row),
//     V TRANSPOSED [dim][136] so that the PV operand (8 keys of one head dim) is two 8-byte reads; the 16 waves of a
//     workgroup would otherwise each split the same rows again (16 x the VALU work);
//   * a 16-key x 16-query score tile is 6 (NPL) / 1 MFMAs over head dims 0..31 plus ONE more for dims 32..35: the
//     remainders of all six plane products are laid side by side along the 32-deep contraction of one instruction
//     (lane group g = 0: qh.kh + qh.km, 1: qm.kh + qh.kl, 2: ql.kh + qm.km) -- 7 instructions of ~17 cycles instead of nine
//     32-cycle fp32 ones;
//   * P V over 32 keys: the probabilities of a lane (keys 4g..4g+3 of two 16-key tiles) are split into planes in
//     registers and used as the B operand as they are (the contraction slot <-> key map is a permutation both operands
//     share); head dims 32..35 of all V planes form one stacked 16-row tile (rows 0-3 Vh, 4-7 Vm, 8-11 Vl), three more
//     instructions whose lane-group partials are added once at the end (as mha2's 4x4 tiles).
#include "eda_common.h"
#include "mha2.h"

namespace {

constexpr int HD = 36;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int CHK = 128;                 // keys per chunk
constexpr int KROW = 40;                 // bf16 per K plane row (80 bytes)
constexpr int KREM = 24;                 // bf16 per remainder row (48 bytes): [kh | km | kh | kl | kh | km] x 4 dims
constexpr int VROW = CHK + 8;            // bf16 per V^T row (272 bytes)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
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

// two fp32 values -> their bf16 planes, packed pairwise (one v_cvt_pk_bf16_f32 per plane): v = h + m + l exactly
template <int NPL>
__device__ __forceinline__ void split2(float a, float b, unsigned (&pl)[3]) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  pl[0] = __builtin_bit_cast(unsigned, h);
  pl[1] = 0u; pl[2] = 0u;
  if (NPL == 3) {
    const float ra = a - (float)h[0], rb = b - (float)h[1];
    const bf16x2 m = __builtin_convertvector(f32x2{ra, rb}, bf16x2);
    pl[1] = __builtin_bit_cast(unsigned, m);
    const bf16x2 l = __builtin_convertvector(f32x2{ra - (float)m[0], rb - (float)m[1]}, bf16x2);
    pl[2] = __builtin_bit_cast(unsigned, l);
  }
}

__device__ __forceinline__ f32x4 mma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct DropCfg { unsigned seed, thresh; float inv_keep; };

// LDS-DMA of `rows` x 36 floats into a LINEAR [rows][36] tile (mha2.hip's dma_rows: piece = 64 granules of 16 bytes)
template <int ROWS>
__device__ __forceinline__ void dma_rows(float *lds, const float *base, long row_stride, int row0, int nrows,
                                         int wave, int nwaves, int lane) {
  constexpr int G = ROWS * 9, P = (G + 63) / 64;
  for (int p = wave; p < P; p += nwaves) {
    const int i = 64 * p + lane;
    if (G % 64 == 0 || i < G) {
      const int row = i / 9, c4 = i - row * 9;
      const int grow = min(row0 + row, nrows - 1);
      const float *src = base + (long)grow * row_stride + 4 * c4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(lds + 256 * p), 16, 0, 0);
    }
  }
}

// LDS carve (bytes)
constexpr int STAGE_B = 2 * CHK * HD * 4;                         // K | V fp32 rows of one chunk
constexpr int KP_B = CHK * KROW * 2;                              // one K plane
constexpr int KR_B = CHK * KREM * 2;
constexpr int VT_B = 32 * VROW * 2;                               // one V^T plane (dims 0..31)
constexpr int VR_B = 16 * VROW * 2;                               // stacked remainder tile
template <int NPL> constexpr int lds_bytes() { return 2 * STAGE_B + NPL * KP_B + KR_B + NPL * VT_B + VR_B + 2 * (CHK / 4) * 4 + 64; }

template <int NPL, int NQ, int KS, bool DROP>
__global__ __launch_bounds__(NQ * KS * 64) void mha3_fwd_kernel(const Mha2Args a) {
  constexpr int NW = NQ * KS, NT = NW * 64;
#ifdef EDA_MHA3_ABLATE       // timing experiments (results are then wrong): bits 1 convert once, 2 no softmax, 4 no PV, 8 no QK^T, 16 no dropout
  const int DBG3 = a.dbg3;
#else
  constexpr int DBG3 = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *stage0 = reinterpret_cast<float *>(smem);
  float *stage1 = reinterpret_cast<float *>(smem + STAGE_B);
  unsigned char *kp = smem + 2 * STAGE_B;                           // [NPL][CHK][KROW] bf16
  unsigned char *kr = kp + NPL * KP_B;                              // [CHK][KREM]
  unsigned char *vt = kr + KR_B;                                    // [NPL][32][VROW]
  unsigned char *vr = vt + NPL * VT_B;                              // [16][VROW]
  unsigned *dead0 = reinterpret_cast<unsigned *>(vr + VR_B);        // [2][CHK / 4] dead-key flags (a byte per key)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int BH = a.B * a.H;
  const int bh = (int)(blockIdx.x % (unsigned)BH), qb = (int)(blockIdx.x / (unsigned)BH);
  const int b = bh / a.H, h = bh - b * a.H;
  const int qs = wave / KS, ks = wave - qs * KS;
  const int qi = qb * (16 * NQ) + 16 * qs + c;
  const bool qvalid = qi < a.Lq;
  const bool wave_live = qb * (16 * NQ) + 16 * qs < a.Lq;

  const float *kbase = a.k + (long)b * a.k_sb + h * HD;
  const float *vbase = a.v + (long)b * a.v_sb + h * HD;
  const unsigned char *mrow = a.mask ? a.mask + (long)b * a.Lk : nullptr;

  // ---- the query operand: scaled (scale * log2 e) in fp32, then split into planes
  u32x4 qp[NPL];                 // dims 8g .. 8g+7
  u32x4 qr = {0u, 0u, 0u, 0u};   // remainder slots (see the header)
  {
    const float *qrow = a.q + (long)b * a.q_sb + (long)(qvalid ? qi : 0) * a.q_sl + h * HD;
    const float sc = qvalid ? a.scale * LOG2E : 0.f;
    const float4 x = *reinterpret_cast<const float4 *>(qrow + 8 * g);
    const float4 y = *reinterpret_cast<const float4 *>(qrow + 8 * g + 4);
    const float4 z = *reinterpret_cast<const float4 *>(qrow + 32);
    const float v8[8] = {x.x * sc, x.y * sc, x.z * sc, x.w * sc, y.x * sc, y.y * sc, y.z * sc, y.w * sc};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned pl[3];
      split2<NPL>(v8[2 * i], v8[2 * i + 1], pl);
#pragma unroll
      for (int p = 0; p < NPL; ++p) qp[p][i] = pl[p];
    }
    unsigned r01[3], r23[3];
    split2<NPL>(z.x * sc, z.y * sc, r01);
    split2<NPL>(z.z * sc, z.w * sc, r23);
    // lane group g: q-side remainder pair [first four | second four]: 0: qh qh, 1: qm qh, 2: ql qm, 3: nothing
    const int pa = g == 0 ? 0 : (g == 1 ? 1 : 2), pb = g == 2 ? 1 : 0;
    if (g < 3 && (NPL == 3 || g == 0)) {
      qr[0] = r01[pa]; qr[1] = r23[pa];
      if (NPL == 3) { qr[2] = r01[pb]; qr[3] = r23[pb]; }
    }
  }
  DropCfg dc = {0u, 0u, 1.f};
  if (DROP && a.p_drop > 0.f) {
    dc.seed = hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt);
    dc.thresh = (unsigned)((double)a.p_drop * 65536.0 + 0.5);
    dc.inv_keep = 1.f / (1.f - a.p_drop);
  }
  const unsigned rowbase = ((unsigned)bh * (unsigned)a.Lq + (unsigned)qi) * (unsigned)a.Lk;

  // rows 4 NPL .. 15 of the stacked remainder tile are never written by the conversion: zero once
  for (int i = tid; i < (16 - 4 * NPL) * VROW / 2; i += NT) reinterpret_cast<unsigned *>(vr + 4 * NPL * VROW * 2)[i] = 0u;

  auto stage = [&](float *st, unsigned *dd, int k0) {
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
    dma_rows<CHK>(st, kbase, a.k_sl, k0, a.Lk, wave, NW, lane);
    dma_rows<CHK>(st + CHK * HD, vbase, a.v_sl, k0, a.Lk, wave, NW, lane);
  };

  // fp32 chunk -> bf16 operand planes (every element split once per workgroup)
  auto convert = [&](const float *st) {
    const float *Kf = st, *Vf = st + CHK * HD;
    for (int i = tid; i < CHK * 18; i += NT) {                       // K: (key, pair of dims)
      const int key = i / 18, dp = i - key * 18;
      const float2 x = *reinterpret_cast<const float2 *>(Kf + key * HD + 2 * dp);
      unsigned pl[3];
      split2<NPL>(x.x, x.y, pl);
      if (dp < 16) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<unsigned *>(kp + p * KP_B + (key * KROW + 2 * dp) * 2) = pl[p];
      } else {
        unsigned *row = reinterpret_cast<unsigned *>(kr + key * KREM * 2) + (dp - 16);        // [kh km | kh kl | kh km], 2 words each
        row[0] = pl[0]; row[2] = pl[1]; row[4] = pl[0]; row[6] = pl[2]; row[8] = pl[0]; row[10] = pl[1];
      }
    }
    for (int i = tid; i < HD * (CHK / 2); i += NT) {                  // V^T: (pair of keys, dim), dim fastest: conflict-free reads
      const int kq = i / HD, dim = i - kq * HD;
      unsigned pl[3];
      split2<NPL>(Vf[(2 * kq) * HD + dim], Vf[(2 * kq + 1) * HD + dim], pl);
      if (dim < 32) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<unsigned *>(vt + p * VT_B + (dim * VROW + 2 * kq) * 2) = pl[p];
      } else {
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<unsigned *>(vr + ((4 * p + dim - 32) * VROW + 2 * kq) * 2) = pl[p];
      }
    }
  };

  float m = -INFINITY, lsum = 0.f;
  f32x4 o[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

  // One 32-key block (two 16-key score tiles) of the converted chunk for this wave's 16 queries, in two halves so that
  // the score MFMAs of the NEXT block are in the pipe while the VALU works through this block's softmax (a wave issues
  // in order: without this the four waves of a SIMD, released together by the chunk barrier, sit in the same phase).
  // Scores: two accumulators per tile (small terms / large terms): four independent MFMA chains of 4 + 3 instead of two of 7.
  auto scores = [&](int kb, f32x4 (&st)[2]) {
    f32x4 sa[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, sb[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    u32x4 ka[2][NPL], kx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = kb + 16 * j + c;
#pragma unroll
      for (int p = 0; p < NPL; ++p) ka[j][p] = *reinterpret_cast<const u32x4 *>(kp + p * KP_B + key * (KROW * 2) + 16 * g);
      kx[j] = u32x4{0u, 0u, 0u, 0u};
      if (g < 3 && (NPL == 3 || g == 0)) kx[j] = *reinterpret_cast<const u32x4 *>(kr + key * (KREM * 2) + 16 * g);
    }
    if (DBG3 & 8) { st[0] = st[1] = f32x4{__uint_as_float(ka[0][0][0] & 0x3f800000u), 0, 0, 0}; return; }
    if (NPL == 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { sa[j] = mma32(ka[j][1], qp[1], sa[j]); sb[j] = mma32(ka[j][0], qp[1], sb[j]); }
#pragma unroll
      for (int j = 0; j < 2; ++j) { sa[j] = mma32(ka[j][0], qp[2], sa[j]); sb[j] = mma32(ka[j][1], qp[0], sb[j]); }
#pragma unroll
      for (int j = 0; j < 2; ++j) sa[j] = mma32(ka[j][2], qp[0], sa[j]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { sa[j] = mma32(kx[j], qr, sa[j]); sb[j] = mma32(ka[j][0], qp[0], sb[j]); }
#pragma unroll
    for (int j = 0; j < 2; ++j) st[j] = sa[j] + sb[j];
  };

  auto finish = [&](int kb, int key0, const unsigned *dd, bool need_mask, f32x4 (&st)[2]) {
    if (need_mask) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned dw = dd[(kb >> 2) + 4 * j + g];
#pragma unroll
        for (int r = 0; r < 4; ++r) st[j][r] = ((dw >> (8 * r)) & 0xffu) != 0u ? -INFINITY : st[j][r];
      }
    }
    // the V operands are requested now: their LDS latency runs under the softmax arithmetic
    u32x4 va[2][NPL], vx;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const unsigned char *row = vt + p * VT_B + ((16 * t + c) * VROW + kb + 4 * g) * 2;
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(row), hi = *reinterpret_cast<const u32x2 *>(row + 32);
        va[t][p] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
    {
      const unsigned char *row = vr + (c * VROW + kb + 4 * g) * 2;       // stacked rows: lane c = (plane, dim 32 + c % 4)
      const u32x2 lo = *reinterpret_cast<const u32x2 *>(row), hi = *reinterpret_cast<const u32x2 *>(row + 32);
      vx = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    float tmax = -INFINITY;
    if (DBG3 & 2) tmax = 0.f;
    else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, st[j][r]);
      tmax = grp_max(tmax);
    }
    const float m_new = fmaxf(m, tmax);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (DBG3 & 2) ? st[j][r] : __builtin_amdgcn_exp2f(st[j][r] - m_safe);
        st[j][r] = p;
        psum += p;
      }
    lsum = lsum * alpha + psum;
    m = m_new;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) o[nt] *= alpha;
    if (DROP && !(DBG3 & 16)) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r2 = 0; r2 < 4; r2 += 2) {
          const unsigned hh = hash32(dc.seed ^ (rowbase + (unsigned)(key0 + 16 * j + 4 * g + r2)));
          st[j][r2] = (hh & 0xffffu) >= dc.thresh ? st[j][r2] : 0.f;
          st[j][r2 + 1] = (hh >> 16) >= dc.thresh ? st[j][r2 + 1] : 0.f;
        }
    }
    // P^T planes: contraction slot 4 j + r of lane group g <-> key kb + 16 j + 4 g + r
    u32x4 pp[NPL];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        unsigned pl[3];
        split2<NPL>(st[j][2 * r2], st[j][2 * r2 + 1], pl);
#pragma unroll
        for (int p = 0; p < NPL; ++p) pp[p][2 * j + r2] = pl[p];
      }
    if (DBG3 & 4) { o[0][0] += __uint_as_float(pp[0][0] ^ va[0][0][0] ^ vx[0]); return; }
    // O^T[dim][query] += V^T P^T: the three accumulator chains (dims 0-15, 16-31, the stacked 32-35 tile) interleaved
    if (NPL == 3) {
      o[0] = mma32(va[0][1], pp[1], o[0]); o[1] = mma32(va[1][1], pp[1], o[1]); o[2] = mma32(vx, pp[2], o[2]);
      o[0] = mma32(va[0][0], pp[2], o[0]); o[1] = mma32(va[1][0], pp[2], o[1]);
      o[0] = mma32(va[0][2], pp[0], o[0]); o[1] = mma32(va[1][2], pp[0], o[1]); o[2] = mma32(vx, pp[1], o[2]);
      o[0] = mma32(va[0][0], pp[1], o[0]); o[1] = mma32(va[1][0], pp[1], o[1]);
      o[0] = mma32(va[0][1], pp[0], o[0]); o[1] = mma32(va[1][1], pp[0], o[1]);
    }
    o[0] = mma32(va[0][0], pp[0], o[0]); o[1] = mma32(va[1][0], pp[0], o[1]); o[2] = mma32(vx, pp[0], o[2]);
  };

  auto compute = [&](const unsigned *dd, int k0) {
    if (!wave_live) return;
    constexpr int NB = (CHK / 32) / KS;                 // blocks of this wave per chunk: t = ks + KS * i
    auto live = [&](int i) { return i < NB && k0 + 32 * (ks + KS * i) < a.Lk; };
    auto fin = [&](int i, f32x4 (&st)[2]) {
      const int t = ks + KS * i, key0 = k0 + 32 * t;
      finish(32 * t, key0, dd, (mrow != nullptr) || (key0 + 32 > a.Lk), st);
    };
    f32x4 sA[2], sB[2];
    if (!live(0)) return;
    scores(32 * ks, sA);
#pragma unroll
    for (int i = 0; i < NB; i += 2) {
      if (live(i + 1)) scores(32 * (ks + KS * (i + 1)), sB);
      fin(i, sA);
      if (!live(i + 1)) break;
      if (live(i + 2)) scores(32 * (ks + KS * (i + 2)), sA);
      fin(i + 1, sB);
      if (!live(i + 2)) break;
    }
  };

  const int nchunks = (a.Lk + CHK - 1) / CHK;
  if (nchunks > 0) stage(stage0, dead0, 0);
  for (int ci = 0; ci < nchunks; ++ci) {
    float *cur = (ci & 1) ? stage1 : stage0, *nxt = (ci & 1) ? stage0 : stage1;
    unsigned *dcur = dead0 + (ci & 1) * (CHK / 4), *dnxt = dead0 + ((ci + 1) & 1) * (CHK / 4);
    EDA_SYNC_DMA();                   // chunk ci has landed (every wave waited for its own pieces), the planes are free
    if (!(DBG3 & 1) || ci == 0) convert(cur);
    __syncthreads();                  // planes ready; nothing in flight: the next chunk's DMA may start
    if (ci + 1 < nchunks) stage(nxt, dnxt, (ci + 1) * CHK);
    compute(dcur, ci * CHK);
  }
  __syncthreads();

  // ---- epilogue (mha2.hip's): merge the KS key shares of a query sub-tile through LDS, normalise, store
  lsum = grp_sum(lsum);
  // (the stacked remainder tile: rows 0-3 = Vh P, 4-7 = Vm P, 8-11 = Vl P of dims 32..35 -> lane group g holds plane g's part
  //  in o[2][0..3]; their sum over the lane groups is the value, as for mha2's 4x4 partials)
  if (KS > 1) {
    float *scr = reinterpret_cast<float *>(smem);                    // one slot = 64 lanes x 16 floats
    auto slot = [&](int q_, int k_) -> float * { return scr + (q_ * (KS - 1) + (k_ - 1)) * 1024; };
    static_assert((KS - 1) * NQ * 4096 <= 2 * STAGE_B, "merge scratch must fit the stages");
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
  if (qvalid && (KS == 1 || ks == 0)) {
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

template <int NPL, int NQ, int KS>
int launch3(Mha2Args &a, hipStream_t stream) {
  a.n_qs = (a.Lq + 16 * NQ - 1) / (16 * NQ);
  const dim3 grid((unsigned)(a.B * a.H * a.n_qs)), block(NQ * KS * 64);
  constexpr size_t lds = lds_bytes<NPL>();
  const bool drop = a.p_drop > 0.f;
  hipError_t e;
  if (drop) {
    auto kern = mha3_fwd_kernel<NPL, NQ, KS, true>;
    e = eda_set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e == hipSuccess) hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else {
    auto kern = mha3_fwd_kernel<NPL, NQ, KS, false>;
    e = eda_set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e == hipSuccess) hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  if (e != hipSuccess) { eda_set_error("mha3: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  EDA_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// Returns -1 when the shape / arithmetic is not this kernel's (the caller then launches mha2.hip's forward).
// Measured (rocprofv3, B = 8, dropout 0.1, profiles/r05_mha3.txt): 1024 x 1024 fp32-accurate 119.9 -> 97-98 us; 256 x 1024
// 40.1 against mha2's 37.9 (64-query workgroups: the conversion pass is paid per 64 queries) -> stays on mha2; plain bf16
// (NPL = 1) 58.1 / 26.9 us against 57.5 / 22.8 for mha2's packed-quad kernels -> BF16 stays on mha2 as well.  Ablations
// (same file): the matrix pipe is ~25 % busy; conversion 10 us, P split + V operand reads + PV 28, dropout hash 9, scores 12
// -- VALU / LDS / barrier time, not MFMA time, is what is left.
// EDA_MHA3=0 switches the kernel off; =2 sends every shape with >= 512 keys and >= 192 queries here, BF16 included.
int eda_mha3_fwd_launch(Mha2Args &a, hipStream_t stream) {
  const long mode = eda_knob(EDA_K_MHA3);
  if (mode == 0) return -1;
  a.dbg3 = (int)eda_knob(EDA_K_MHA3_DBG);
  if (a.dtype != EDA_DTYPE_F32 && a.dtype != EDA_DTYPE_BF16) return -1;
  if (mode == 2) { if (a.Lk < 512 || a.Lq < 192) return -1; }
  else if (a.dtype != EDA_DTYPE_F32 || a.Lk < 512 || a.Lq < 512) return -1;
  const long BH = (long)a.B * a.H;
  const bool short_q = BH * ((a.Lq + 255) / 256) < 192;            // 256-query workgroups alone would leave CUs idle
  if (a.dtype == EDA_DTYPE_F32) return short_q ? launch3<3, 4, 4>(a, stream) : launch3<3, 16, 1>(a, stream);
  return short_q ? launch3<1, 4, 4>(a, stream) : launch3<1, 16, 1>(a, stream);
}
