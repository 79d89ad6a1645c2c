// sa_eval.hip -- the set-abstraction level of the point backbone in ONE pass for inference (BatchNorm on running
// statistics): ball-query rows gathered -> conv1x1 + BN + ReLU x 3 -> max over the neighbourhood, nothing but the pooled
// (centres x C3) output written.  Reference chain: pointnet2/pointnet2_utils.py:317-376 (QueryAndGroup) ->
// pointnet2/pytorch_utils.py:67-120 (SharedMLP in eval mode) -> pointnet2/pointnet2_modules.py:251-257 (max_pool2d).
//
// This is the form SURVEY.md section 7 step 4 / section 8d prices ("fused SA layer fwd: grouped tensor never written"):
// in TRAINING every pre-activation has to be kept for the BatchNorm backward (csrc/sa_cl.hip: one launch per layer, 1 GB
// of z_l for SA1), with folded BatchNorm nothing has to leave the chip.  Built for SA1 of the backbone (3 feature
// channels; 6 -> 64 -> 64 -> 128, 64 neighbours: 26.6 GFLOP against 22.4 MB of algorithmic bytes at B = 8).
//
// Arithmetic: layer 1 (contraction 3 + 3) on the fp32 MFMA as gemm_gather3_kernel does; layers 2 and 3 as bf16 x 3 on
// v_mfma_f32_16x16x32_bf16 (operands split exactly into three bf16 planes, the six plane products of weight >= 2^-24:
// fp32 accuracy, csrc/mha3.hip / wgrad.hip).  The weight planes are built once per workgroup in LDS; the ACTIVATIONS never
// leave the registers: with the weight as the A operand, a lane's accumulators are channels 16 t + 4 g + r of data row c,
// which is exactly a B-operand fragment of the next layer once the contraction slots are numbered accordingly (slot
// 8 g + 4 t' + r of 32-channel block u <-> channel 16 (2 u + t') + 4 g + r; the weight planes are stored in that order).
// A wave owns a centre: two passes of 32 rows (two 16-row blocks share every weight operand read), running max in
// registers, one cross-lane reduction per centre.
#include "eda_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SaEvalArgs {
  const float *xyz, *new_xyz, *feats; const int *idx;
  int b, n, m, ns; float inv_radius;
  const float *w[3], *gamma[3], *beta[3], *rmean[3], *rvar[3]; float eps;
  float *out;
};

__device__ __forceinline__ void split2(float a, float b, unsigned (&pl)[3]) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  pl[0] = __builtin_bit_cast(unsigned, h);
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  const bf16x2 m = __builtin_convertvector(f32x2{ra, rb}, bf16x2);
  pl[1] = __builtin_bit_cast(unsigned, m);
  const bf16x2 l = __builtin_convertvector(f32x2{ra - (float)m[0], rb - (float)m[1]}, bf16x2);
  pl[2] = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f32x4 mma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float row_max16(float v) {           // max over the 16 lanes of a DPP row
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR1>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR2>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(4)>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(8)>(__float_as_int(v))));
  return v;
}

constexpr int SE_WAVES = 8;             // (12 waves = 3 per SIMD: 175.8 us against 166.3, profiles/r05_sa_eval.txt)

// six plane products, smallest first: acc += sum over the planes (A = weight planes wa[3], B = activation planes xb[3])
__device__ __forceinline__ f32x4 dot6(const u32x4 (&wa)[3], const u32x4 (&xb)[3], f32x4 acc) {
  acc = mma32(wa[1], xb[1], acc);
  acc = mma32(wa[0], xb[2], acc);
  acc = mma32(wa[2], xb[0], acc);
  acc = mma32(wa[0], xb[1], acc);
  acc = mma32(wa[1], xb[0], acc);
  return mma32(wa[0], xb[0], acc);
}

template <int C1, int C2, int C3>
__global__ __launch_bounds__(64 * SE_WAVES) void sa_eval_kernel(const SaEvalArgs a) {
  static_assert(C1 % 32 == 0 && C2 % 32 == 0 && C3 % 16 == 0, "channel counts");
  constexpr int T1 = C1 / 16, T3 = C3 / 16, U1 = C1 / 32, U2 = C2 / 32;
  constexpr int ROW2 = C1 + 8, ROW3 = C2 + 8;                        // bf16 per weight-plane row (16-byte reads, conflict-free)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *w2p = smem;                                         // [3][C2][ROW2] bf16
  unsigned char *w3p = w2p + 3 * C2 * ROW2 * 2;                      // [3][C3][ROW3]
  float *cst = reinterpret_cast<float *>(w3p + 3 * C3 * ROW3 * 2);   // scale | shift of the three layers
  float *sc1 = cst, *sh1 = sc1 + C1, *sc2 = sh1 + C1, *sh2 = sc2 + C2, *sc3 = sh2 + C2, *sh3 = sc3 + C3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;

  // ---- once per workgroup: BatchNorm constants from the running statistics, the weight planes of layers 2 and 3
  for (int i = tid; i < C1 + C2 + C3; i += 64 * SE_WAVES) {
    const int l = i < C1 ? 0 : (i < C1 + C2 ? 1 : 2), ch = i - (l == 0 ? 0 : (l == 1 ? C1 : C1 + C2));
    const float rstd = 1.f / sqrtf(a.rvar[l][ch] + a.eps);            // (csrc/sa_cl.hip bn_eval_affine_kernel)
    const float s = a.gamma[l][ch] * rstd;
    float *scp = l == 0 ? sc1 : (l == 1 ? sc2 : sc3), *shp = l == 0 ? sh1 : (l == 1 ? sh2 : sh3);
    scp[ch] = s;
    shp[ch] = a.beta[l][ch] - a.rmean[l][ch] * s;
  }
  auto planes = [&](const float *w, int N, int K, int ROW, unsigned char *dst) {
    // element pair (n, slot pair): slot 8 g' + 4 t' + r of block u <-> channel 16 (2 u + t') + 4 g' + r, r in {0, 2}
    for (int i = tid; i < N * (K / 2); i += 64 * SE_WAVES) {
      const int n = i / (K / 2), sp = i - n * (K / 2);
      const int slot = 2 * sp, u = slot >> 5, g_ = (slot >> 3) & 3, tq = (slot >> 2) & 1, r = slot & 3;
      const int ch = 16 * (2 * u + tq) + 4 * g_ + r;
      unsigned pl[3];
      split2(w[(long)n * K + ch], w[(long)n * K + ch + 1], pl);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned *>(dst + ((p * N + n) * ROW + slot) * 2) = pl[p];
    }
  };
  planes(a.w[1], C2, C1, ROW2, w2p);
  planes(a.w[2], C3, C2, ROW3, w3p);
  // layer 1's weight columns as fp32 A operands (gemm_gather3_kernel's form: contraction steps (x, y, z) | (three features))
  float av0[T1], av1[T1];
#pragma unroll
  for (int j = 0; j < T1; ++j) {
    const float *wp = a.w[0] + (long)(16 * j + c) * 6;
    av0[j] = g < 3 ? wp[g] : 0.f;
    av1[j] = g < 3 ? wp[3 + g] : 0.f;
  }
  __syncthreads();

  const long ncentres = (long)a.b * a.m;
  for (long ctr = (long)blockIdx.x * SE_WAVES + wave; ctr < ncentres; ctr += (long)gridDim.x * SE_WAVES) {
    const long scene = ctr / a.m;
    float cx = 0.f;
    if (g < 3) cx = a.new_xyz[ctr * 3 + g];
    f32x4 best[T3];
#pragma unroll
    for (int t = 0; t < T3; ++t) best[t] = f32x4{0.f, 0.f, 0.f, 0.f};       // (max of ReLU outputs: >= 0)
#pragma unroll 1
    for (int pass = 0; pass < a.ns; pass += 32) {
      // ---- gather + layer 1 (fp32 MFMA), BatchNorm + ReLU, planes
      u32x4 b1[2][U1][3];
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        const int k = min(pass + 16 * bi + c, a.ns - 1);                      // (ns % 32 != 0: the tail repeats a row -- max is idempotent)
        const long gp = scene * a.n + a.idx[ctr * a.ns + k];
        float bx = 0.f, bf = 0.f;
        if (g < 3) {
          bx = (a.xyz[gp * 3 + g] - cx) * a.inv_radius;
          bf = a.feats[gp * 3 + g];
        }
        float act[T1][4];
#pragma unroll
        for (int j = 0; j < T1; ++j) {
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          z = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bx, z, 0, 0, 0);
          z = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bf, z, 0, 0, 0);
          const f32x4 s = *reinterpret_cast<const f32x4 *>(sc1 + 16 * j + 4 * g), h = *reinterpret_cast<const f32x4 *>(sh1 + 16 * j + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) act[j][r] = fmaxf(z[r] * s[r] + h[r], 0.f);
        }
#pragma unroll
        for (int u = 0; u < U1; ++u)
#pragma unroll
          for (int tq = 0; tq < 2; ++tq)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
              unsigned pl[3];
              split2(act[2 * u + tq][2 * r2], act[2 * u + tq][2 * r2 + 1], pl);
#pragma unroll
              for (int p = 0; p < 3; ++p) b1[bi][u][p][2 * tq + r2] = pl[p];
            }
      }
      // ---- layer 2: output tiles in pairs (2 u, 2 u + 1) = one 32-channel contraction block of layer 3, split at once
      // (sched_barrier: the compiler otherwise hoists the weight-plane reads of every tile to the top -- 300 spilled registers)
      u32x4 b2[2][U2][3];
#pragma unroll
      for (int u2 = 0; u2 < U2; ++u2) {
        float act[2][2][4];
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
          const int n = 2 * u2 + tq;
          f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < U1; ++u) {
            u32x4 wa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[p] = *reinterpret_cast<const u32x4 *>(w2p + ((p * C2 + 16 * n + c) * ROW2 + 32 * u + 8 * g) * 2);
            z0 = dot6(wa, b1[0][u], z0);
            z1 = dot6(wa, b1[1][u], z1);
          }
          const f32x4 s = *reinterpret_cast<const f32x4 *>(sc2 + 16 * n + 4 * g), h = *reinterpret_cast<const f32x4 *>(sh2 + 16 * n + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) { act[0][tq][r] = fmaxf(z0[r] * s[r] + h[r], 0.f); act[1][tq][r] = fmaxf(z1[r] * s[r] + h[r], 0.f); }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
          for (int tq = 0; tq < 2; ++tq)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
              unsigned pl[3];
              split2(act[bi][tq][2 * r2], act[bi][tq][2 * r2 + 1], pl);
#pragma unroll
              for (int p = 0; p < 3; ++p) b2[bi][u2][p][2 * tq + r2] = pl[p];
            }
      }
      // ---- layer 3, BatchNorm + ReLU, running max over the rows of the centre
#pragma unroll
      for (int n = 0; n < T3; ++n) {
        f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U2; ++u) {
          u32x4 wa[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) wa[p] = *reinterpret_cast<const u32x4 *>(w3p + ((p * C3 + 16 * n + c) * ROW3 + 32 * u + 8 * g) * 2);
          z0 = dot6(wa, b2[0][u], z0);
          z1 = dot6(wa, b2[1][u], z1);
        }
        const f32x4 s = *reinterpret_cast<const f32x4 *>(sc3 + 16 * n + 4 * g), h = *reinterpret_cast<const f32x4 *>(sh3 + 16 * n + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          best[n][r] = fmaxf(best[n][r], fmaxf(fmaxf(z0[r] * s[r] + h[r], 0.f), fmaxf(z1[r] * s[r] + h[r], 0.f)));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- max over the 16 rows a lane group saw, one 16-byte store per (tile, lane group)
#pragma unroll
    for (int n = 0; n < T3; ++n) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = row_max16(best[n][r]);
      if (c == 0) *reinterpret_cast<f32x4 *>(a.out + ctr * C3 + 16 * n + 4 * g) = v;
    }
  }
}

template <int C1, int C2, int C3>
constexpr size_t sa_eval_lds() { return (size_t)3 * C2 * (C1 + 8) * 2 + (size_t)3 * C3 * (C2 + 8) * 2 + (size_t)2 * (C1 + C2 + C3) * 4; }

}  // namespace

extern "C" int eda_sa_fused_eval_supported(int c_feat, int nlayers, const int *channels, int ns) {
  return c_feat == 3 && nlayers == 3 && channels && channels[0] == 6 && channels[1] == 64 && channels[2] == 64 &&
         channels[3] == 128 && ns >= 1 && ns <= 255;
}

extern "C" int eda_sa_fused_eval_f32(const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx, int b, int n,
                                     int m, int ns, int c_feat, float radius, int normalize_xyz, int nlayers,
                                     const int *channels, const float *const *weight, const float *const *gamma,
                                     const float *const *beta, const float *const *running_mean,
                                     const float *const *running_var, float eps, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && n > 0 && m >= 0 && ns >= 1, "bad dimension");
  if (!eda_sa_fused_eval_supported(c_feat, nlayers, channels, ns)) {
    eda_set_error("eda_sa_fused_eval_f32: built for 3 feature channels and the 6 -> 64 -> 64 -> 128 stack (SA1)");
    return EDA_ERR_UNSUPPORTED;
  }
  if (b == 0 || m == 0) return 0;
  EDA_CHECK_ARG(xyz && new_xyz && feats_cl && idx && weight && gamma && beta && running_mean && running_var && out, "null pointer");
  SaEvalArgs a;
  a.xyz = xyz; a.new_xyz = new_xyz; a.feats = feats_cl; a.idx = idx;
  a.b = b; a.n = n; a.m = m; a.ns = ns; a.inv_radius = normalize_xyz ? 1.0f / radius : 1.0f;
  for (int l = 0; l < 3; ++l) {
    EDA_CHECK_ARG(weight[l] && gamma[l] && beta[l] && running_mean[l] && running_var[l], "null pointer");
    a.w[l] = weight[l]; a.gamma[l] = gamma[l]; a.beta[l] = beta[l]; a.rmean[l] = running_mean[l]; a.rvar[l] = running_var[l];
  }
  a.eps = eps; a.out = out;
  constexpr size_t lds = sa_eval_lds<64, 64, 128>();
  auto kern = sa_eval_kernel<64, 64, 128>;
  EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
  const long centres = (long)b * m;
  long grid = (centres + SE_WAVES - 1) / SE_WAVES;
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * SE_WAVES), lds, stream, a);
  EDA_CHECK_LAUNCH();
  return 0;
}
