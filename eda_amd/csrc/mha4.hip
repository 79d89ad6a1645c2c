// mha4.hip -- attention forward for a SHORT query set against the 1024 point keys, keys per WAVE (round 6).
//
// The site: the encoder's text -> point cross-attention (models/encoder_decoder_layers.py:87-93: q = the 80 / 130 text
// tokens, k = v = the 1024 seed points; F.multi_head_attention_forward: q * scale, QK^T, -inf key-padding fill, softmax,
// dropout, PV) -- and, on request, the decoder's 256 queries against the points (:393-399).
//
// Why another decomposition.  mha2.hip's forward gives a wave 16 QUERIES and streams the keys through LDS in 256-key
// chunks; with 80 queries that is 5 query waves per (scene, head), so it adds key shares (4 waves per query tile, merged
// through LDS) and key splits (2 workgroups, merged through HBM) and still leaves half the chip idle behind a staging pass
// nobody overlaps: 24.7 us for 0.755 GFLOP (19 % of the fp32 matrix peak, profiles/r05_mha_ksplit.txt: "~10 us of fixed
// cost per workgroup").  Here a wave owns 32 KEYS: it loads their K and V rows ONCE, straight from global memory into
// MFMA operand registers (no LDS staging, no chunk barrier, every load of the kernel in flight at the same time), and walks
// over ALL query tiles, whose rows sit in LDS (80 x 36 floats).  For one query tile it sees all of its keys at once, so the
// softmax needs no running rescale: S^T = K Q^T (18 MFMA), m = max, P = exp2(S - m), l = sum, O^T = V^T P^T (18 + 2 x 4 small).
// 8 waves = 256 keys per workgroup, Lk / 256 workgroups per (scene, head): 256 workgroups at B = 8, one per CU, one round.
//   merge 1 (LDS)   the 8 waves' (O, m, l) of a query tile, by wave j for tile j, in wave order;
//   merge 2 (HBM)   the key splits' (O, m, l): write-through partials + ticket, the last-arriving workgroup of a (scene, head)
//                   merges in split order -- the protocol of mha2.hip's SPLIT forward and of the split-K GEMMs
//                   (cdna_hip_programming.md G16 form R1); results do not depend on who arrives last.
// MEASURED (profiles/r06_mha4_keys_per_wave.md) and NOT DISPATCHED by default (EDA_MHA4=1 / 2 select it): 80 x 1024 24.9 us
// against mha2's 24.7, 130 x 1024 37.0 against 34.3, 256 x 1024 56.9 against 38.4.  The phase timeline of the 80-query launch
// (wall-clock stamps per workgroup, tools/mha4_phase_profile.py) says why no decomposition of this family gets near the
// matrix roof at this size: 5.0 us until the operands have landed and the first barrier is passed (kernel arguments, the
// dependent address chain, one HBM round trip), 6.6 us for a wave's five query tiles and 11.8 us until BOTH waves of a SIMD are
// through them (per tile 42 MFMAs = 1300 cycles next to ~150 dependent VALU instructions -- max, exp2, hash, lane-group
// reductions -- that the in-order wave does not overlap with its own MFMAs; v_mfma_f32_16x16x4_f32 issues every 36 cycles at
// 2.17 GHz with the chip busy, tools/probe/mfma_f32_rate.hip: 126 TFLOP/s is the real roof), then 2.3 us draining the
// write-through partials, 0.5 us for the ticket and 2.9 us for the last arriver's merge.  5 + 6 us of latency that no tiling
// removes bracket 6 us of matrix work.
// Arithmetic: fp32 MFMA (v_mfma_f32_16x16x4_f32), the parity path; the dropout hash, the log2-domain scores, lse and the
// all-keys-masked behaviour (NaN, as the reference) are mha2.hip's, so mha2's backward consumes this forward's (out, lse).
#include "eda_common.h"
#include "mha2.h"

namespace {

constexpr int HD = 36;
constexpr int KSTEPS = 9;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
#ifndef EDA_MHA4_NW
#define EDA_MHA4_NW 8
#endif
#ifndef EDA_MHA4_KT
#define EDA_MHA4_KT 2
#endif
#ifndef EDA_MHA4_QB
#define EDA_MHA4_QB 5
#endif
constexpr int NW4 = EDA_MHA4_NW;         // waves per workgroup
constexpr int KT4 = EDA_MHA4_KT;         // 16-key tiles per wave
constexpr int WGK = NW4 * KT4 * 16;      // keys per workgroup (256)
constexpr int QB4 = EDA_MHA4_QB;         // query tiles per batch (one LDS merge per batch)
constexpr int MAXQT = 16;                // query tiles (Lq <= 256)
constexpr int SLOT_F = 64 * 8 + 16 * 8;  // floats per (wave, tile) merge slot: o0 | o1 of 64 lanes, then o2 | m | l | - | - of 16

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// 16 blocks of 4x4x1: D[reg i][lane 4b+j] += A[lane 4b+i] * B[lane 4b+j] (mha2.hip mfma44: head dims 32..35, per-lane-group partials)
__device__ __forceinline__ f32x4 mfma44(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

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
// a lane's 9 contraction values of one 36-float row (k-step s of lane group g <-> head dim 8g+s, s < 8; 32+g for s = 8)
__device__ __forceinline__ void load_row_operand(float (&r)[KSTEPS], const float *row, int g) {
  const float4 x = *reinterpret_cast<const float4 *>(row + 8 * g);
  const float4 y = *reinterpret_cast<const float4 *>(row + 8 * g + 4);
  r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w;
  r[4] = y.x; r[5] = y.y; r[6] = y.z; r[7] = y.w;
  r[8] = row[32 + g];
}

// online-softmax merge of a partial (p0, p1, p2, mo, lo) into (o, m, l)
__device__ __forceinline__ void merge_in(f32x4 (&o)[3], float &m, float &l, f32x4 p0, f32x4 p1, f32x4 p2, float mo, float lo) {
  const float mn = fmaxf(m, mo);
  const float ms = (mn == -INFINITY) ? 0.f : mn;
  const float fa = __builtin_amdgcn_exp2f(m - ms), fb = __builtin_amdgcn_exp2f(mo - ms);
  o[0] = o[0] * fa + p0 * fb; o[1] = o[1] * fa + p1 * fb; o[2] = o[2] * fa + p2 * fb;
  l = l * fa + lo * fb;
  m = mn;
}

#ifdef EDA_MHA4_PROFILE
// phase timeline (experiments only, tools/mha4_phase_profile.py): wall-clock stamps (100 MHz) of thread 0 of every workgroup
__device__ unsigned long long mha4_prof[1024 * 8];
#define P4(slot) do { if (threadIdx.x == 0) mha4_prof[(blockIdx.x & 1023) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define P4(slot) do { } while (0)
#endif

template <bool DROP>
__global__ __launch_bounds__(NW4 * 64) void mha4_fwd_kernel(const Mha2Args a) {
  P4(0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nqt = (a.Lq + 15) >> 4;
  float *Qs = smem;                                   // [nqt * 16][36], scaled by scale * log2 e, zero rows beyond Lq
  float *slots = smem + nqt * 16 * HD;                // [NW4][QB4][SLOT_F]
  __shared__ unsigned flag_s;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int BH = a.B * a.H;
  const int bh = (int)(blockIdx.x % (unsigned)BH), sp = (int)(blockIdx.x / (unsigned)BH);
  const int b = bh / a.H, h = bh - b * a.H;
  const int k0 = sp * WGK + wave * (16 * KT4);        // first key of this wave

  // ---- this wave's K and V operands, straight from global memory (issued before anything else: one round trip) ----
  const float *kbase = a.k + (long)b * a.k_sb + h * HD;
  const float *vbase = a.v + (long)b * a.v_sb + h * HD;
  float kreg[KT4][KSTEPS];
  float vreg[KT4][4][3];                              // v[t][r][n] = V[key 16t + 4g + r][dim c + 16n] (n = 2: dim 32 + (c & 3))
#pragma unroll
  for (int t = 0; t < KT4; ++t) {
    const int kr = min(k0 + 16 * t + c, a.Lk - 1);
    load_row_operand(kreg[t], kbase + (long)kr * a.k_sl, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int vr = min(k0 + 16 * t + 4 * g + r, a.Lk - 1);
      const float *vrow = vbase + (long)vr * a.v_sl;
      vreg[t][r][0] = vrow[c];
      vreg[t][r][1] = vrow[16 + c];
      vreg[t][r][2] = vrow[32 + (c & 3)];
    }
  }
  // dead keys of this lane's score registers: key 16t + 4g + r beyond Lk or padded
  bool dead[KT4][4];
  const unsigned char *mrow = a.mask ? a.mask + (long)b * a.Lk : nullptr;
#pragma unroll
  for (int t = 0; t < KT4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = k0 + 16 * t + 4 * g + r;
      dead[t][r] = key >= a.Lk || (mrow && mrow[min(key, a.Lk - 1)]);
    }
  // ---- the queries of this (scene, head) -> LDS ----
  {
    const float *qbase = a.q + (long)b * a.q_sb + h * HD;
    const float sc = a.scale * LOG2E;
    for (int i = tid; i < nqt * 16 * 9; i += NW4 * 64) {
      const int row = i / 9, c4 = i - row * 9;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < a.Lq) {
        x = *reinterpret_cast<const float4 *>(qbase + (long)row * a.q_sl + 4 * c4);
        x.x *= sc; x.y *= sc; x.z *= sc; x.w *= sc;
      }
      *reinterpret_cast<float4 *>(Qs + row * HD + 4 * c4) = x;
    }
  }
  unsigned seed = 0u, thresh = 0u;
  float inv_keep = 1.f;
  if (DROP) {
    seed = hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt);
    thresh = (unsigned)((double)a.p_drop * 65536.0 + 0.5);
    inv_keep = 1.f / (1.f - a.p_drop);
  }
  P4(1);
  __syncthreads();
  P4(2);

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      a.fwd_part + (long)bh * a.n_ksplit * nqt * (64 * 16), 0, a.n_ksplit * nqt * 64 * 16 * 4, 0x00020000);

  // out / lse of one query tile from the merged (o, m, l) in this wave's lanes
  auto write_out = [&](int qt, const f32x4 (&o)[3], float m, float l) {
    const int qi = 16 * qt + c;
    if (qi >= a.Lq) return;
    const float inv = inv_keep / l;                    // all keys masked -> NaN, like the reference
    float *orow = a.o + (long)b * a.o_sb + (long)qi * a.o_sl + h * HD;
    *reinterpret_cast<f32x4 *>(orow + 4 * g) = o[0] * inv;
    *reinterpret_cast<f32x4 *>(orow + 16 + 4 * g) = o[1] * inv;
    if (g == 0) {
      *reinterpret_cast<f32x4 *>(orow + 32) = o[2] * inv;
      a.lse[(long)bh * a.Lq + qi] = (m + __builtin_amdgcn_logf(l)) * LN2;       // v_log_f32 = log2
    }
  };

#pragma unroll 1
  for (int bt = 0; bt < nqt; bt += QB4) {
    const int nb = min(QB4, nqt - bt);
    // ---- every wave: its 32 keys against the batch's query tiles ----
#pragma unroll 1
    for (int jj = 0; jj < nb; ++jj) {
      const int qt = bt + jj;
      float qreg[KSTEPS];
      load_row_operand(qreg, Qs + (16 * qt + c) * HD, g);
      f32x4 st[KT4];
#pragma unroll
      for (int t = 0; t < KT4; ++t) st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int t = 0; t < KT4; ++t) st[t] = mfma4(kreg[t][s], qreg[s], st[t]);      // S^T[key 4g+r][query c], log2 domain
      float tmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < KT4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[t][r] = dead[t][r] ? -INFINITY : st[t][r];
          tmax = fmaxf(tmax, st[t][r]);
        }
      const float m = grp_max(tmax);
      const float m_safe = (m == -INFINITY) ? 0.f : m;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < KT4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[t][r] - m_safe);
          st[t][r] = p;
          psum += p;
        }
      const float l = grp_sum(psum);
      if (DROP) {
        const unsigned rowbase = ((unsigned)bh * (unsigned)a.Lq + (unsigned)(16 * qt + c)) * (unsigned)a.Lk;
#pragma unroll
        for (int t = 0; t < KT4; ++t)
#pragma unroll
          for (int r2 = 0; r2 < 4; r2 += 2) {
            const unsigned hsh = hash32(seed ^ (rowbase + (unsigned)(k0 + 16 * t + 4 * g + r2)));
            st[t][r2] = (hsh & 0xffffu) >= thresh ? st[t][r2] : 0.f;
            st[t][r2 + 1] = (hsh >> 16) >= thresh ? st[t][r2 + 1] : 0.f;
          }
      }
      f32x4 o[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int t = 0; t < KT4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pb = st[t][r];
          o[0] = mfma4(vreg[t][r][0], pb, o[0]);       // O^T[dim 4g+r'][query c]
          o[1] = mfma4(vreg[t][r][1], pb, o[1]);
          o[2] = mfma44(vreg[t][r][2], pb, o[2]);      // dims 32..35: per-lane-group partials
        }
      o[2] = grp_sum4(o[2]);
      float *sl = slots + (wave * QB4 + jj) * SLOT_F;
      *reinterpret_cast<f32x4 *>(sl + lane * 8) = o[0];
      *reinterpret_cast<f32x4 *>(sl + lane * 8 + 4) = o[1];
      if (g == 0) {
        *reinterpret_cast<f32x4 *>(sl + 512 + c * 8) = o[2];
        sl[512 + c * 8 + 4] = m;
        sl[512 + c * 8 + 5] = l;
      }
    }
    P4(7);
    __syncthreads();
    // ---- merge 1: wave j sums the 8 waves' partials of the batch's tile j, in wave order ----
    if (wave < nb) {
      const int qt = bt + wave;
      float m = -INFINITY, l = 0.f;
      f32x4 o[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 1
      for (int w = 0; w < NW4; ++w) {
        const float *sl = slots + (w * QB4 + wave) * SLOT_F;
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(sl + lane * 8);
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(sl + lane * 8 + 4);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(sl + 512 + c * 8);
        merge_in(o, m, l, p0, p1, p2, sl[512 + c * 8 + 4], sl[512 + c * 8 + 5]);
      }
      if (a.n_ksplit > 1) {
        // this key split's (O, m, l) of the tile: write-through, merged by the last-arriving split (below)
        const int off = ((sp * nqt + qt) * 64 + lane) * 64;           // bytes
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[0]), rs, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[1]), rs, off + 16, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[2]), rs, off + 32, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{m, l, 0.f, 0.f}), rs, off + 48, 0, 16);
      } else {
        write_out(qt, o, m, l);
      }
    }
    __syncthreads();                                   // (the slots are rewritten by the next batch)
  }
  P4(3);
  if (a.n_ksplit <= 1) return;

  // ---- merge 2: the key splits, by the last workgroup of this (scene, head) to arrive ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every publishing wave drains its write-through stores
  __syncthreads();
  P4(4);
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(a.fwd_tickets + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == (unsigned)a.n_ksplit - 1u;
    if (last) __hip_atomic_store(a.fwd_tickets + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    flag_s = last ? 1u : 0u;
  }
  __syncthreads();
  P4(5);
  if (flag_s == 0u) return;
#pragma unroll 1
  for (int qt = wave; qt < nqt; qt += NW4) {
    float m = -INFINITY, l = 0.f;
    f32x4 o[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 4
    for (int z = 0; z < a.n_ksplit; ++z) {             // split order: the result does not depend on who is last
      const int off = ((z * nqt + qt) * 64 + lane) * 64;
      const f32x4 p0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
      const f32x4 p1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16));
      const f32x4 p2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 16));
      const f32x4 ml = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 16));
      merge_in(o, m, l, p0, p1, p2, ml[0], ml[1]);
    }
    write_out(qt, o, m, l);
  }
  P4(6);
}

size_t lds_bytes(int Lq) { return sizeof(float) * ((size_t)((Lq + 15) / 16) * 16 * HD + (size_t)NW4 * QB4 * SLOT_F); }

}  // namespace

// Does this kernel take the shape?  EDA_MHA4: 0 never; 1 (default) <= 144 queries against >= 512 keys (the text -> point
// cross-attention, 80 / 130 tokens); 2 also up to 256 queries (the decoder's query -> point cross-attention).
bool eda_mha4_takes(int dtype, int Lq, int Lk) {
  const long mode = eda_knob(EDA_K_MHA4);
  if (mode == 0 || dtype != EDA_DTYPE_F32 || Lk < 512 || Lq < 1) return false;
  return Lq <= (mode >= 2 ? 16 * MAXQT : 144);
}

size_t eda_mha4_fwd_workspace_bytes(int B, int H, int Lq, int Lk) {
  const size_t blocks = (size_t)B * H;
  const size_t tick = (blocks * sizeof(unsigned) + 255) / 256 * 256;
  const size_t ns = (size_t)(Lk + WGK - 1) / WGK;
  return tick + blocks * ns * ((Lq + 15) / 16) * 64 * 16 * sizeof(float);
}

// ws: [tickets (zero before the first call, left zero) | partials], eda_mha4_fwd_workspace_bytes()
int eda_mha4_fwd_launch(Mha2Args &a, void *ws, size_t ws_bytes, hipStream_t stream) {
  const size_t need = eda_mha4_fwd_workspace_bytes(a.B, a.H, a.Lq, a.Lk);
  EDA_CHECK_ARG(ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0,
                "workspace of eda_mha_fwd_workspace_bytes() required (16-byte aligned, its ticket words zero)");
  const size_t blocks = (size_t)a.B * a.H;
  const size_t tick = (blocks * sizeof(unsigned) + 255) / 256 * 256;
  a.fwd_tickets = reinterpret_cast<unsigned *>(ws);
  a.fwd_part = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(ws) + tick);
  a.n_ksplit = (a.Lk + WGK - 1) / WGK;
  a.keys_per_split = WGK;
  EDA_CHECK_ARG(a.Lq <= 16 * MAXQT, "at most 256 queries");
  const size_t lds = lds_bytes(a.Lq), lds_max = lds_bytes(16 * MAXQT);     // (the attribute is set once per kernel: the largest)
  const dim3 grid((unsigned)(blocks * a.n_ksplit)), block(NW4 * 64);
  if (a.p_drop > 0.f) {
    EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(&mha4_fwd_kernel<true>), lds_max));
    hipLaunchKernelGGL(mha4_fwd_kernel<true>, grid, block, lds, stream, a);
  } else {
    EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(&mha4_fwd_kernel<false>), lds_max));
    hipLaunchKernelGGL(mha4_fwd_kernel<false>, grid, block, lds, stream, a);
  }
  EDA_CHECK_LAUNCH();
  return 0;
}

#ifdef EDA_MHA4_PROFILE
extern "C" __attribute__((visibility("default"))) int eda_mha4_profile_read(unsigned long long *out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mha4_prof), sizeof(unsigned long long) * (size_t)n) != hipSuccess;
}
#endif
