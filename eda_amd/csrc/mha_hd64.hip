// mha_hd64.hip -- forward-only fused attention for head_dim 64 on the fp32 MFMA pipe: the self-attention of the FROZEN
// RoBERTa-base text encoder (12 layers x 12 heads x 64; reference: models/bdetr.py:77-80, 170-175 ->
// transformers RobertaSelfAttention: q k^T / 8 + additive padding mask, softmax, P v).  Inference only (the encoder
// has no gradient), utterances of up to 256 tokens: K and V of one (sentence, head) sit in LDS whole.
//
// Same transposed formulation as mha2.hip (lane owns query l & 15, S^T = K Q^T, P^T feeds O^T += V^T P^T from the
// accumulator registers); head_dim 64 = 16 k-steps in the first product and exactly four 16-row output tiles in the
// second (no padding).  K / V rows are staged with a 68-float stride: the 16 rows of a sub-tile then start in 16
// different 4-bank groups and a lane's 16 contraction values (dims 16g .. 16g+15 of its lane group g) are four
// conflict-free ds_read_b128.
#include "eda_common.h"

namespace {

constexpr int HD64 = 64;
constexpr int LDK = 68;
constexpr int MAXLK = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
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

struct Hd64Args {
  const float *q, *k, *v;
  long q_sb, q_sl, k_sb, k_sl, v_sb, v_sl;
  float *o; long o_sb, o_sl;
  const unsigned char *mask;          // (B, Lk) 1 = ignore, or null
  int B, H, Lq, Lk;
  float scale;
  float p_drop;                       // dropout on the probabilities (transformers RobertaSelfAttention, train mode)
  const unsigned long long *seed_ptr;
  unsigned salt;
};

__device__ __forceinline__ unsigned hd64_hash32(unsigned x) {       // (the mask hash of eda_mha_* / csrc/mha2.hip)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// the lane's 16 contraction values of one 64-float row: dims 16g .. 16g+15 (MFMA k-step s of lane group g <-> dim 16g + s)
__device__ __forceinline__ void load_row16(float (&r)[16], const float *row, int g) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 x = *reinterpret_cast<const float4 *>(row + 16 * g + 4 * i);
    r[4 * i] = x.x; r[4 * i + 1] = x.y; r[4 * i + 2] = x.z; r[4 * i + 3] = x.w;
  }
}

template <bool DROP>
__global__ __launch_bounds__(256) void mha_hd64_fwd_kernel(const Hd64Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lk = a.Lk;
  float *Ks = smem, *Vs = smem + (long)Lk * LDK;
  unsigned char *dead = reinterpret_cast<unsigned char *>(Vs + (long)Lk * LDK);      // Lk (padded to 16) flags
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int qi = blockIdx.y * 64 + wave * 16 + c;
  const bool qvalid = qi < a.Lq;

  const float *kbase = a.k + (long)b * a.k_sb + h * HD64;
  const float *vbase = a.v + (long)b * a.v_sb + h * HD64;
  for (int i = tid; i < Lk * 16; i += 256) {
    const int row = i >> 4, c4 = i & 15;
    *reinterpret_cast<float4 *>(Ks + row * LDK + 4 * c4) = *reinterpret_cast<const float4 *>(kbase + (long)row * a.k_sl + 4 * c4);
    *reinterpret_cast<float4 *>(Vs + row * LDK + 4 * c4) = *reinterpret_cast<const float4 *>(vbase + (long)row * a.v_sl + 4 * c4);
  }
  const int Lk16 = (Lk + 15) & ~15;
  for (int i = tid; i < Lk16; i += 256) dead[i] = (i >= Lk || (a.mask && a.mask[(long)b * Lk + i])) ? 1 : 0;

  float qreg[16];
  {
    const float *qrow = a.q + (long)b * a.q_sb + (long)(qvalid ? qi : 0) * a.q_sl + h * HD64;
    load_row16(qreg, qrow, g);
    const float sc = a.scale * 1.4426950408889634f;         // scores in the log2 domain
#pragma unroll
    for (int s = 0; s < 16; ++s) qreg[s] = qvalid ? qreg[s] * sc : 0.f;
  }
  __syncthreads();

  unsigned dseed = 0u, dthresh = 0u;
  float inv_keep = 1.f;
  if (DROP) {                                   // 16-bit thresholds, two probabilities per hash: as csrc/mha2.hip
    dseed = hd64_hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt);
    dthresh = (unsigned)((double)a.p_drop * 65536.0 + 0.5);
    inv_keep = 1.f / (1.f - a.p_drop);
  }
  const unsigned rowbase = ((unsigned)bh * (unsigned)a.Lq + (unsigned)qi) * (unsigned)Lk;
  float m = -INFINITY, lsum = 0.f;
  f32x4 o[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const int nsub = Lk16 >> 4;
  for (int j0 = 0; j0 < nsub; j0 += 4) {                      // 64-key tiles: online softmax between them
    const int nj = min(4, nsub - j0);
    f32x4 st[4];
    float tmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      st[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (jj < nj) {
        const int key = 16 * (j0 + jj) + c;                  // rows >= Lk: the last valid row (finite; masked dead below)
        float kr[16];
        load_row16(kr, Ks + min(key, Lk - 1) * LDK, g);
#pragma unroll
        for (int s = 0; s < 16; ++s) st[jj] = mfma4(kr[s], qreg[s], st[jj]);
        const unsigned dw = *reinterpret_cast<const unsigned *>(dead + 16 * (j0 + jj) + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[jj][r] = ((dw >> (8 * r)) & 0xffu) ? -INFINITY : st[jj][r];
          tmax = fmaxf(tmax, st[jj][r]);
        }
      }
    }
    tmax = grp_max(tmax);
    const float m_new = fmaxf(m, tmax);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (jj < nj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[jj][r] - m_safe);
          st[jj][r] = p;
          psum += p;
        }
      }
    lsum = lsum * alpha + psum;
    m = m_new;
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] *= alpha;
    if (DROP) {                                 // (the softmax sum keeps every probability; only P v sees the mask)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (jj < nj) {
#pragma unroll
          for (int r2 = 0; r2 < 4; r2 += 2) {
            const unsigned hh = hd64_hash32(dseed ^ (rowbase + (unsigned)(16 * (j0 + jj) + 4 * g + r2)));
            st[jj][r2] = (hh & 0xffffu) >= dthresh ? st[jj][r2] : 0.f;
            st[jj][r2 + 1] = (hh >> 16) >= dthresh ? st[jj][r2 + 1] : 0.f;
          }
        }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (jj < nj) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int key = min(16 * (j0 + jj) + 4 * g + t, Lk - 1);
          const float *vr = Vs + key * LDK + c;
          const float pb = st[jj][t];
          o[0] = mfma4(vr[0], pb, o[0]);
          o[1] = mfma4(vr[16], pb, o[1]);
          o[2] = mfma4(vr[32], pb, o[2]);
          o[3] = mfma4(vr[48], pb, o[3]);
        }
      }
  }
  lsum = grp_sum(lsum);
  if (qvalid) {
    const float inv = inv_keep / lsum;         // every key masked -> NaN, like the reference
    float *orow = a.o + (long)b * a.o_sb + (long)qi * a.o_sl + h * HD64;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      *reinterpret_cast<float4 *>(orow + 16 * n + 4 * g) = make_float4(o[n][0] * inv, o[n][1] * inv, o[n][2] * inv, o[n][3] * inv);
  }
}

bool mult4(long v) { return (v & 3) == 0; }
bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int eda_mha_fwd_hd64_drop_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb,
                                         long k_sl, long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H,
                                         int Lq, int Lk, float scale, float p_drop, const unsigned long long *seed_ptr,
                                         unsigned salt, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "bad dimension");
  if (B == 0 || Lq == 0) return 0;
  EDA_CHECK_ARG(Lk >= 1 && Lk <= MAXLK, "1 <= Lk <= 256 (K and V of a (sentence, head) are kept in LDS whole)");
  EDA_CHECK_ARG(q && k && v && out, "null pointer");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "0 <= p < 1, and a counter when p > 0");
  EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) && al16(q) &&
                    al16(k) && al16(v) && al16(out), "rows must be 16-byte aligned");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  Hd64Args a = {};
  a.q = q; a.k = k; a.v = v; a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.v_sb = v_sb; a.v_sl = v_sl;
  a.o = out; a.o_sb = (long)Lq * H * HD64; a.o_sl = (long)H * HD64; a.mask = key_padding_mask;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
  a.p_drop = p_drop; a.seed_ptr = seed_ptr; a.salt = salt;
  const size_t lds = sizeof(float) * 2 * (size_t)Lk * LDK + (size_t)((Lk + 15) & ~15);
  auto kern = p_drop > 0.f ? mha_hd64_fwd_kernel<true> : mha_hd64_fwd_kernel<false>;
  {           // dynamic LDS above 64 KB needs the opt-in; the attribute is per DEVICE and cheap: set on every launch
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(float) * 2 * MAXLK * LDK + 256));
    if (e != hipSuccess) { eda_set_error("eda_mha_fwd_hd64_f32: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * H), (unsigned)((Lq + 63) / 64)), dim3(256), lds, stream, a);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_mha_fwd_hd64_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb,
                                    long k_sl, long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H,
                                    int Lq, int Lk, float scale, float *out, void *stream_) {
  return eda_mha_fwd_hd64_drop_f32(q, k, v, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, key_padding_mask, B, H, Lq, Lk, scale, 0.f, nullptr,
                                   0u, out, stream_);
}
