// mha16.hip -- fused multi-head attention with 16-bit MFMA contractions (bf16 or fp16 inputs, fp32
// accumulate, fp32 softmax), forward + backward, for EDA's shapes (8 heads x 36, key-padding masks,
// attention-probability dropout).  BASELINE.json configs[2] ("bf16 attention on MFMA") and configs[4]
// ("fp16 MFMA cross-attn"); the fp32 kernels of mha.hip remain the parity path (the reference computes
// nn.MultiheadAttention in fp32, models/encoder_decoder_layers.py:87-117, and the 1e-4 bound of the
// north star applies to that).
//
// The tensors at the boundary stay fp32 (the projections around the core are fp32 GEMMs): q/k/v/dO
// are converted to 16 bits while they are staged into LDS or loaded into operand registers, so the
// HBM traffic is the fp32 kernel's and what changes is the contraction rate:
// v_mfma_f32_16x16x16_{bf16,f16} performs 8192 FLOP per instruction against 2048 of
// v_mfma_f32_16x16x4_f32.  head_dim 36 is padded to 48 (three 16-deep k steps; the 12 padded
// columns are zeros and are NOT counted as useful FLOPs).
//
// Same transposed formulation as mha.hip: lane l owns query (l & 15) in the forward and dQ kernels and
// key (l & 15) in the dK/dV kernel,
//     S^T = K Q^T                 D[key = 4g + r][query = l & 15],  g = l >> 4
// so the row statistics of a query live in the 16 lanes of a column (+ a reduction over the four lane
// groups) and P^T is already in the B-operand layout (4 consecutive keys per lane) of O^T += V^T P^T.
// Operands that are read along the "other" axis (V^T, K^T, Q^T, dO^T) are staged TRANSPOSED in LDS.
// One barrier pair per 64-row tile, no software pipelining: these variants exist for the 16-bit
// configurations and their roofline report, not for the fp32 headline.
#include "eda_common.h"
#include "mha2.h"
#include <string.h>

namespace {

constexpr int HD = 36, HDP = 48;    // head dim, padded
constexpr int T16 = 64;             // rows per LDS tile
constexpr int RS = 56;              // row stride (elements) of [row][d] tiles: 112 B -> conflict-free b64 operand reads
constexpr int TS = 72;              // row stride of transposed [d][row] tiles: 144 B

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

struct Bf16 {
  static __device__ __forceinline__ unsigned short cvt(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);                     // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
  }
  static __device__ __forceinline__ f32x4 mfma(unsigned long long a, unsigned long long b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};
struct Fp16 {
  static __device__ __forceinline__ unsigned short cvt(float x) {
    return __builtin_bit_cast(unsigned short, (_Float16)x);
  }
  static __device__ __forceinline__ f32x4 mfma(unsigned long long a, unsigned long long b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
  }
};

template <class T>
__device__ __forceinline__ unsigned long long pack4(float a, float b, float c, float d) {
  return (unsigned long long)T::cvt(a) | ((unsigned long long)T::cvt(b) << 16) | ((unsigned long long)T::cvt(c) << 32) |
         ((unsigned long long)T::cvt(d) << 48);
}

struct Args16 {
  const float *q, *k, *v; long q_sb, q_sl, k_sb, k_sl, v_sb, v_sl;
  float *o; long o_sb, o_sl; float *lse;
  const unsigned char *mask;
  int B, H, Lq, Lk; float scale, p_drop;
  const unsigned long long *seed_ptr; unsigned salt;
  const float *dout; long do_sb, do_sl; float *delta;
  float *dq, *dk, *dv; long dq_sb, dq_sl, dk_sb, dk_sl, dv_sb, dv_sl;
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct Drop { bool on; unsigned seed, thresh; float inv_keep; };
__device__ __forceinline__ Drop make_drop(const Args16 &a, int bh) {
  Drop d; d.on = a.p_drop > 0.f && a.seed_ptr != nullptr; d.seed = 0; d.thresh = 0; d.inv_keep = 1.f;
  if (d.on) {
    d.seed = hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt * 0x85EBCA6Bu + (unsigned)bh);
    d.thresh = (unsigned)((double)a.p_drop * 4294967296.0);
    d.inv_keep = 1.f / (1.f - a.p_drop);
  }
  return d;
}
// keep factor of probability (query, key): the same in forward and backward
__device__ __forceinline__ float keep(const Drop &d, int query, int key, int Lk) {
  return hash32(d.seed ^ (unsigned)(query * Lk + key)) >= d.thresh ? d.inv_keep : 0.f;
}

// 64 rows x 36 floats of a strided fp32 matrix -> LDS [row][RS] 16-bit (columns 36..47 zero) and/or
// transposed [d][TS].  Rows >= nrows are zero.
template <class T, bool ROWS, bool TRANS>
__device__ __forceinline__ void stage_tile(unsigned short *rows_lds, unsigned short *trans_lds, const float *base,
                                           long row_stride, int r0, int nrows, float mul) {
  for (int s = threadIdx.x; s < T16 * 12; s += blockDim.x) {
    const int row = s / 12, c = s - row * 12;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < 9 && r0 + row < nrows) v = *reinterpret_cast<const float4 *>(base + (long)(r0 + row) * row_stride + 4 * c);
    v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    if (ROWS) *reinterpret_cast<unsigned long long *>(rows_lds + row * RS + 4 * c) = pack4<T>(v.x, v.y, v.z, v.w);
    if (TRANS) {
      trans_lds[(4 * c + 0) * TS + row] = T::cvt(v.x);
      trans_lds[(4 * c + 1) * TS + row] = T::cvt(v.y);
      trans_lds[(4 * c + 2) * TS + row] = T::cvt(v.z);
      trans_lds[(4 * c + 3) * TS + row] = T::cvt(v.w);
    }
  }
}

// B-operand fragments of one row (query / key owned by lane l & 15): d = 16*step + 4g .. +3
template <class T>
__device__ __forceinline__ void load_frag(unsigned long long (&f)[3], const float *row, bool valid, int g, float mul) {
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const int d0 = 16 * st + 4 * g;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && d0 < HD) v = *reinterpret_cast<const float4 *>(row + d0);
    f[st] = pack4<T>(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
  }
}

__device__ __forceinline__ float gmax(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float gsum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

// ---------------------------------------------------------------------------------------------- forward
template <class T>
__global__ __launch_bounds__(256) void mha16_fwd_kernel(const Args16 a) {
  __shared__ __attribute__((aligned(16))) unsigned short Kl[T16 * RS];
  __shared__ __attribute__((aligned(16))) unsigned short Vt[HDP * TS];
  __shared__ unsigned char dead[T16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int q0 = blockIdx.x * T16 + wave * 16;
  const int query = q0 + c;
  const bool qok = query < a.Lq;
  const float *qrow = a.q + (long)b * a.q_sb + (long)(qok ? query : 0) * a.q_sl + h * HD;
  unsigned long long qf[3];
  load_frag<T>(qf, qrow, qok, g, a.scale);
  const Drop dr = make_drop(a, bh);
  f32x4 o[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;
  const float *kb = a.k + (long)b * a.k_sb + h * HD, *vb = a.v + (long)b * a.v_sb + h * HD;
  for (int k0 = 0; k0 < a.Lk; k0 += T16) {
    __syncthreads();
    stage_tile<T, true, false>(Kl, nullptr, kb, a.k_sl, k0, a.Lk, 1.f);
    stage_tile<T, false, true>(nullptr, Vt, vb, a.v_sl, k0, a.Lk, 1.f);
    if (threadIdx.x < T16) {
      const int key = k0 + threadIdx.x;
      dead[threadIdx.x] = (key >= a.Lk) || (a.mask && a.mask[(long)b * a.Lk + key]);
    }
    __syncthreads();
    f32x4 s[4];
    float mx = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      s[sub] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 3; ++st) {
        const unsigned long long kf = *reinterpret_cast<const unsigned long long *>(Kl + (16 * sub + c) * RS + 16 * st + 4 * g);
        s[sub] = T::mfma(kf, qf[st], s[sub]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (dead[16 * sub + 4 * g + r]) s[sub][r] = -INFINITY;
        mx = fmaxf(mx, s[sub][r]);
      }
    }
    mx = gmax(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = m_new == -INFINITY ? 1.f : __expf(m_run - m_new);
#pragma unroll
    for (int t = 0; t < 3; ++t) o[t] *= alpha;
    l_part *= alpha;
    m_run = m_new;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = m_new == -INFINITY ? 0.f : __expf(s[sub][r] - m_new);
        l_part += p[r];
        if (dr.on) p[r] *= keep(dr, query, k0 + 16 * sub + 4 * g + r, a.Lk);
      }
      const unsigned long long pf = pack4<T>(p[0], p[1], p[2], p[3]);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned long long vf = *reinterpret_cast<const unsigned long long *>(Vt + (16 * t + c) * TS + 16 * sub + 4 * g);
        o[t] = T::mfma(vf, pf, o[t]);
      }
    }
  }
  const float l_tot = gsum(l_part);
  const float inv = 1.f / l_tot;                  // (a fully masked row gives NaN like the reference)
  if (qok) {
    float *orow = a.o + (long)b * a.o_sb + (long)query * a.o_sl + h * HD;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int d0 = 16 * t + 4 * g;
      if (d0 < HD) *reinterpret_cast<float4 *>(orow + d0) = make_float4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
    }
    if (g == 0) a.lse[((long)b * a.H + h) * a.Lq + query] = m_run + __logf(l_tot);
  }
}

// ---------------------------------------------------------------------------------------------- dQ (+ delta)
template <class T>
__global__ __launch_bounds__(256) void mha16_dq_kernel(const Args16 a) {
  __shared__ __attribute__((aligned(16))) unsigned short Kl[T16 * RS];
  __shared__ __attribute__((aligned(16))) unsigned short Vl[T16 * RS];
  __shared__ __attribute__((aligned(16))) unsigned short Kt[HDP * TS];
  __shared__ unsigned char dead[T16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int query = blockIdx.x * T16 + wave * 16 + c;
  const bool qok = query < a.Lq;
  const long qi = qok ? query : 0;
  unsigned long long qf[3], df[3];
  load_frag<T>(qf, a.q + (long)b * a.q_sb + qi * a.q_sl + h * HD, qok, g, a.scale);
  const float *dorow = a.dout + (long)b * a.do_sb + qi * a.do_sl + h * HD;
  load_frag<T>(df, dorow, qok, g, 1.f);
  // delta = rowsum(dO * O) in fp32
  float dl = 0.f;
  {
    const float *orow = a.o + (long)b * a.o_sb + qi * a.o_sl + h * HD;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const int d0 = 16 * st + 4 * g;
      if (qok && d0 < HD) {
        const float4 x = *reinterpret_cast<const float4 *>(dorow + d0), y = *reinterpret_cast<const float4 *>(orow + d0);
        dl += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
    }
    dl = gsum(dl);
    if (qok && g == 0) a.delta[((long)b * a.H + h) * a.Lq + query] = dl;
  }
  const float lse = qok ? a.lse[((long)b * a.H + h) * a.Lq + query] : 0.f;
  const Drop dr = make_drop(a, bh);
  f32x4 dq[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) dq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *kb = a.k + (long)b * a.k_sb + h * HD, *vb = a.v + (long)b * a.v_sb + h * HD;
  for (int k0 = 0; k0 < a.Lk; k0 += T16) {
    __syncthreads();
    stage_tile<T, true, true>(Kl, Kt, kb, a.k_sl, k0, a.Lk, 1.f);
    stage_tile<T, true, false>(Vl, nullptr, vb, a.v_sl, k0, a.Lk, 1.f);
    if (threadIdx.x < T16) {
      const int key = k0 + threadIdx.x;
      dead[threadIdx.x] = (key >= a.Lk) || (a.mask && a.mask[(long)b * a.Lk + key]);
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 3; ++st) {
        const unsigned long long kf = *reinterpret_cast<const unsigned long long *>(Kl + (16 * sub + c) * RS + 16 * st + 4 * g);
        const unsigned long long vf = *reinterpret_cast<const unsigned long long *>(Vl + (16 * sub + c) * RS + 16 * st + 4 * g);
        s = T::mfma(kf, qf[st], s);
        dp = T::mfma(vf, df[st], dp);
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kl = 16 * sub + 4 * g + r;
        const float p = dead[kl] ? 0.f : __expf(s[r] - lse);
        float dpr = dp[r];
        if (dr.on) dpr *= keep(dr, query, k0 + kl, a.Lk);
        ds[r] = p * (dpr - dl) * a.scale;
      }
      const unsigned long long dsf = pack4<T>(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned long long ktf = *reinterpret_cast<const unsigned long long *>(Kt + (16 * t + c) * TS + 16 * sub + 4 * g);
        dq[t] = T::mfma(ktf, dsf, dq[t]);
      }
    }
  }
  if (qok) {
    float *row = a.dq + (long)b * a.dq_sb + (long)query * a.dq_sl + h * HD;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int d0 = 16 * t + 4 * g;
      if (d0 < HD) *reinterpret_cast<float4 *>(row + d0) = make_float4(dq[t][0], dq[t][1], dq[t][2], dq[t][3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- dK, dV
template <class T>
__global__ __launch_bounds__(256) void mha16_dkv_kernel(const Args16 a) {
  __shared__ __attribute__((aligned(16))) unsigned short Ql[T16 * RS];
  __shared__ __attribute__((aligned(16))) unsigned short Dl[T16 * RS];
  __shared__ __attribute__((aligned(16))) unsigned short Qt[HDP * TS];
  __shared__ __attribute__((aligned(16))) unsigned short Dt[HDP * TS];
  __shared__ float lse_l[T16], del_l[T16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int key = blockIdx.x * T16 + wave * 16 + c;
  const bool kok = key < a.Lk;
  const long ki = kok ? key : 0;
  const bool kdead = !kok || (a.mask && a.mask[(long)b * a.Lk + ki]);
  unsigned long long kf[3], vf[3];
  load_frag<T>(kf, a.k + (long)b * a.k_sb + ki * a.k_sl + h * HD, kok, g, 1.f);
  load_frag<T>(vf, a.v + (long)b * a.v_sb + ki * a.v_sl + h * HD, kok, g, 1.f);
  const Drop dr = make_drop(a, bh);
  f32x4 dk[3], dv[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) { dk[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float *qb = a.q + (long)b * a.q_sb + h * HD, *db = a.dout + (long)b * a.do_sb + h * HD;
  for (int q0 = 0; q0 < a.Lq; q0 += T16) {
    __syncthreads();
    stage_tile<T, true, true>(Ql, Qt, qb, a.q_sl, q0, a.Lq, a.scale);        // scaled Q: S = (scale Q) K^T, dK = dS^T (scale Q)
    stage_tile<T, true, true>(Dl, Dt, db, a.do_sl, q0, a.Lq, 1.f);
    if (threadIdx.x < T16) {
      const int qq = q0 + threadIdx.x;
      lse_l[threadIdx.x] = qq < a.Lq ? a.lse[((long)b * a.H + h) * a.Lq + qq] : INFINITY;     // exp(s - inf) = 0
      del_l[threadIdx.x] = qq < a.Lq ? a.delta[((long)b * a.H + h) * a.Lq + qq] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      // S[query = 4g + r][key = c], dP likewise
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 3; ++st) {
        const unsigned long long qfr = *reinterpret_cast<const unsigned long long *>(Ql + (16 * sub + c) * RS + 16 * st + 4 * g);
        const unsigned long long dfr = *reinterpret_cast<const unsigned long long *>(Dl + (16 * sub + c) * RS + 16 * st + 4 * g);
        s = T::mfma(qfr, kf[st], s);
        dp = T::mfma(dfr, vf[st], dp);
      }
      float pd[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * sub + 4 * g + r;
        const float p = kdead ? 0.f : __expf(s[r] - lse_l[ql]);
        const float kp = dr.on ? keep(dr, q0 + ql, key, a.Lk) : 1.f;
        pd[r] = p * kp;
        ds[r] = p * (dp[r] * kp - del_l[ql]);
      }
      const unsigned long long pf = pack4<T>(pd[0], pd[1], pd[2], pd[3]);
      const unsigned long long dsf = pack4<T>(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned long long dtf = *reinterpret_cast<const unsigned long long *>(Dt + (16 * t + c) * TS + 16 * sub + 4 * g);
        const unsigned long long qtf = *reinterpret_cast<const unsigned long long *>(Qt + (16 * t + c) * TS + 16 * sub + 4 * g);
        dv[t] = T::mfma(dtf, pf, dv[t]);
        dk[t] = T::mfma(qtf, dsf, dk[t]);
      }
    }
  }
  if (kok) {
    float *rk = a.dk + (long)b * a.dk_sb + (long)key * a.dk_sl + h * HD;
    float *rv = a.dv + (long)b * a.dv_sb + (long)key * a.dv_sl + h * HD;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int d0 = 16 * t + 4 * g;
      if (d0 < HD) {
        *reinterpret_cast<float4 *>(rk + d0) = make_float4(dk[t][0], dk[t][1], dk[t][2], dk[t][3]);
        *reinterpret_cast<float4 *>(rv + d0) = make_float4(dv[t][0], dv[t][1], dv[t][2], dv[t][3]);
      }
    }
  }
}

bool mult4(long v) { return (v & 3) == 0; }

}  // namespace

// dtype: EDA_DTYPE_F32 forwards to the fp32 kernels of mha.hip; BF16 / F16 run the kernels above.
extern "C" int eda_mha_fwd(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                           long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                           int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                           float *out, float *lse, int dtype, void *stream_) {
  if (dtype == EDA_DTYPE_F32)
    return eda_mha_fwd_f32(q, k, v, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, key_padding_mask, B, H, Lq, Lk, head_dim, scale,
                           p_drop, seed_ptr, salt, out, lse, stream_);
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(dtype == EDA_DTYPE_BF16 || dtype == EDA_DTYPE_F16, "dtype must be EDA_DTYPE_F32 / BF16 / F16");
  if (eda_mha_impl() != 1 && Lk > 0 && head_dim == HD && B > 0 && Lq > 0) {
    // the round-3 kernels (mha2.hip) with 16-bit contraction quads; this file's kernels stay behind EDA_MHA_IMPL=1
    EDA_CHECK_ARG(q && k && v && out && lse, "null pointer");
    EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl),
                  "strides must be multiples of 4 floats");
    EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
    Mha2Args m = {};
    m.q = q; m.k = k; m.v = v; m.q_sb = q_sb; m.q_sl = q_sl; m.k_sb = k_sb; m.k_sl = k_sl; m.v_sb = v_sb; m.v_sl = v_sl;
    m.o = out; m.o_sb = (long)Lq * H * HD; m.o_sl = (long)H * HD; m.lse = lse; m.mask = key_padding_mask;
    m.B = B; m.H = H; m.Lq = Lq; m.Lk = Lk; m.scale = scale; m.p_drop = p_drop; m.seed_ptr = seed_ptr; m.salt = salt;
    m.dtype = dtype;
    return eda_mha2_fwd_launch(m, (hipStream_t)stream_);
  }
  EDA_CHECK_ARG(head_dim == HD, "only head_dim 36 (d_model 288 / 8 heads) is built");
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "bad dimension");
  if (B == 0 || Lq == 0) return 0;
  EDA_CHECK_ARG(q && k && v && out && lse, "null pointer");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "bad dropout arguments");
  EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                    ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                    ((uintptr_t)out % 16 == 0), "rows must be 16-byte aligned");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  Args16 a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.v_sb = v_sb; a.v_sl = v_sl;
  a.o = out; a.o_sb = (long)Lq * H * HD; a.o_sl = (long)H * HD; a.lse = lse; a.mask = key_padding_mask;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.p_drop = p_drop; a.seed_ptr = seed_ptr; a.salt = salt;
  const dim3 grid((Lq + T16 - 1) / T16, B * H);
  if (dtype == EDA_DTYPE_BF16) hipLaunchKernelGGL(mha16_fwd_kernel<Bf16>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(mha16_fwd_kernel<Fp16>, grid, dim3(256), 0, stream, a);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_mha_bwd(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                           long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                           int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                           const float *out, const float *lse, const float *dout, long do_sb, long do_sl,
                           float *delta_ws, float *dq, float *dk, float *dv, long dq_sb, long dq_sl, long dk_sb,
                           long dk_sl, long dv_sb, long dv_sl, void *ws, size_t ws_bytes, int dtype, void *stream_) {
  if (dtype == EDA_DTYPE_F32)
    return eda_mha_bwd_f32(q, k, v, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, key_padding_mask, B, H, Lq, Lk, head_dim, scale,
                           p_drop, seed_ptr, salt, out, lse, dout, do_sb, do_sl, delta_ws, dq, dk, dv, dq_sb, dq_sl, dk_sb,
                           dk_sl, dv_sb, dv_sl, ws, ws_bytes, stream_);
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(dtype == EDA_DTYPE_BF16 || dtype == EDA_DTYPE_F16, "dtype must be EDA_DTYPE_F32 / BF16 / F16");
  if (eda_mha_impl() != 1 && Lq > 0 && Lk > 0 && head_dim == HD && B > 0) {
    EDA_CHECK_ARG(q && k && v && out && lse && dout && dq && dk && dv, "null pointer");
    EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                      mult4(do_sb) && mult4(do_sl) && mult4(dq_sb) && mult4(dq_sl) && mult4(dk_sb) && mult4(dk_sl) &&
                      mult4(dv_sb) && mult4(dv_sl), "strides must be multiples of 4 floats");
    EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
    Mha2Args m = {};
    m.q = q; m.k = k; m.v = v; m.q_sb = q_sb; m.q_sl = q_sl; m.k_sb = k_sb; m.k_sl = k_sl; m.v_sb = v_sb; m.v_sl = v_sl;
    m.o = const_cast<float *>(out); m.o_sb = (long)Lq * H * HD; m.o_sl = (long)H * HD; m.lse = const_cast<float *>(lse);
    m.mask = key_padding_mask; m.B = B; m.H = H; m.Lq = Lq; m.Lk = Lk; m.scale = scale; m.p_drop = p_drop;
    m.seed_ptr = seed_ptr; m.salt = salt; m.dout = dout; m.do_sb = do_sb; m.do_sl = do_sl; m.dq = dq; m.dk = dk; m.dv = dv;
    m.dq_sb = dq_sb; m.dq_sl = dq_sl; m.dk_sb = dk_sb; m.dk_sl = dk_sl; m.dv_sb = dv_sb; m.dv_sl = dv_sl;
    m.dtype = dtype;
    return eda_mha2_bwd_launch(m, ws, ws_bytes, (hipStream_t)stream_);
  }
  EDA_CHECK_ARG(head_dim == HD, "only head_dim 36 (d_model 288 / 8 heads) is built");
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "bad dimension");
  if (B == 0) return 0;
  EDA_CHECK_ARG(q && k && v && out && lse && dout && delta_ws && dq && dk && dv, "null pointer");
  EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                    mult4(do_sb) && mult4(do_sl) && mult4(dq_sb) && mult4(dq_sl) && mult4(dk_sb) && mult4(dk_sl) &&
                    mult4(dv_sb) && mult4(dv_sl), "strides must be multiples of 4 floats");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  Args16 a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.v_sb = v_sb; a.v_sl = v_sl;
  a.o = const_cast<float *>(out); a.o_sb = (long)Lq * H * HD; a.o_sl = (long)H * HD; a.lse = const_cast<float *>(lse);
  a.mask = key_padding_mask; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.p_drop = p_drop;
  a.seed_ptr = seed_ptr; a.salt = salt; a.dout = dout; a.do_sb = do_sb; a.do_sl = do_sl; a.delta = delta_ws;
  a.dq = dq; a.dk = dk; a.dv = dv; a.dq_sb = dq_sb; a.dq_sl = dq_sl; a.dk_sb = dk_sb; a.dk_sl = dk_sl; a.dv_sb = dv_sb; a.dv_sl = dv_sl;
  if (Lq > 0) {
    const dim3 grid((Lq + T16 - 1) / T16, B * H);
    if (dtype == EDA_DTYPE_BF16) hipLaunchKernelGGL(mha16_dq_kernel<Bf16>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(mha16_dq_kernel<Fp16>, grid, dim3(256), 0, stream, a);
    EDA_CHECK_LAUNCH();
  }
  if (Lk > 0) {
    const dim3 grid((Lk + T16 - 1) / T16, B * H);
    if (dtype == EDA_DTYPE_BF16) hipLaunchKernelGGL(mha16_dkv_kernel<Bf16>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(mha16_dkv_kernel<Fp16>, grid, dim3(256), 0, stream, a);
    EDA_CHECK_LAUNCH();
  }
  return 0;
}
