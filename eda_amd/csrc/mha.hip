// mha.hip -- fused multi-head attention (forward + backward) on the fp32 MFMA
// pipe of gfx950, for the shapes EDA uses: d_model 288 = 8 heads x 36, sequence
// lengths 80..1024, boolean key-padding masks, attention-probability dropout.
//
// Replaces the unfused path inside torch.nn.MultiheadAttention that the reference
// runs 39 times per forward (models/encoder_decoder_layers.py:47,62,69,133,298,
// 306,314,319 -> F.multi_head_attention_forward: q*scale, bmm QK^T, -inf key
// padding fill, softmax, dropout(0.1), bmm PV) -- the (B*8, Lq, Lk) probability
// tensor (268 MB for the 1024x1024 self-attention at B=8) is never written.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; bit-for-bit an fmaf
// chain), so results match an fp32 reference to rounding; head_dim 36 = 9 x 4 means
// the QK^T contraction needs no padding, only PV pads 36 -> 48 output columns.
//
// Formulation.  Everything is computed TRANSPOSED so that one index stays
// lane-local: in the forward and dQ kernels lane l owns query (l & 15) of its
// wave's 16-query tile, in the dK/dV kernel it owns key (l & 15):
//     S^T = K Q^T      A = K[key = l&15][dim = 4s + (l>>4)],  B = Q[query = l&15][dim = 4s + (l>>4)]
//     D   = S^T[key = 4(l>>4) + r][query = l&15]               r = 0..3 (accumulator regs)
// so the softmax max/sum of a query are 16 lane-local values plus a reduction over
// the 4 lane groups, and P^T sits in exactly the B-operand layout of the second
// product  O^T += V^T P^T  (keys taken in the permuted order 4(l>>4)+t, which a sum
// does not care about).  O^T[dim = 4(l>>4)+r][query = l&15] makes the 1/l rescale
// lane-local and the epilogue a 16-byte store of 4 consecutive head dims.
//
// Dropout: counter-based, identical in forward and backward: the two keys 2m, 2m+1 of a query share
// one 32-bit hash of (seed, linear index of the even key); key 2m keeps iff its low 16 bits >=
// p * 2^16, key 2m+1 uses the high 16 bits (half the integer multiplies of one hash per element;
// p is quantised to 1/65536).  seed = *seed_ptr (device counter, so a replayed HIP graph sees a new
// mask every step) mixed with a per-call-site salt.
#include "eda_common.h"
#include "mha2.h"

#include <stdlib.h>

namespace {

constexpr int HD = 36;            // head dim
constexpr int KSTEPS = HD / 4;    // 9 MFMA k-steps for a 36-deep contraction
constexpr int TILE = 64;          // rows of the streamed operand staged in LDS per iteration
// Waves per workgroup: 4 for long sequences (one K/V tile shared by 64 queries), 1 for the
// short decoder / text shapes so that they still spread over >= 1024 workgroups.

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MhaArgs {
  const float *q, *k, *v;   // head h at column h*36; rows strided
  long q_sb, q_sl, k_sb, k_sl, v_sb, v_sl;   // element strides (batch, row)
  float *o; long o_sb, o_sl;                  // forward output (B,Lq,288)
  float *lse;                                 // (B,H,Lq) log-sum-exp of the scaled masked scores
  const unsigned char *mask;                  // (B,Lk) 1 = ignore, or null
  int B, H, Lq, Lk;
  float scale, p_drop;
  const unsigned long long *seed_ptr; unsigned salt;
  // backward
  const float *dout; long do_sb, do_sl;       // (B,Lq,288)
  const float *delta;                         // (B,H,Lq) rowsum(dO * O)
  float *dq, *dk, *dv;                        // (B,L,288) rows, strided like q/k/v
  long dq_sb, dq_sl, dk_sb, dk_sl, dv_sb, dv_sl;
  // dK/dV with the query range split over gridDim.z workgroups (short Lk): partial results
  int q_split_rows;                           // queries per split (multiple of TILE); 0 = no split
  float *dkv_part;                            // [split][dk|dv][B][Lk][H*36] dense
  // dQ with the key range split over gridDim.z workgroups (short Lq): partial results
  int k_split_rows;                           // keys per split (multiple of TILE); 0 = no split
  float *dq_part;                             // [split][B][Lq][H*36] dense
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ float xor_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float xor_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// Stage TILE rows x 36 floats of one head into LDS (row stride 36), zero-filling
// rows >= nrows.  16-byte global loads, 16-byte LDS stores.
template <int NW>
__device__ __forceinline__ void stage_rows(float *lds, const float *base, long row_stride,
                                           int row0, int nrows) {
  for (int i = threadIdx.x; i < TILE * 9; i += NW * 64) {
    const int row = i / 9, c4 = i - row * 9;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < nrows)
      v = *reinterpret_cast<const float4 *>(base + (long)(row0 + row) * row_stride + 4 * c4);
    *reinterpret_cast<float4 *>(lds + row * HD + 4 * c4) = v;
  }
}

// Software-pipelined staging: a tile's 16-byte pieces are first loaded into registers
// (issue_rows, while the previous tile is being multiplied) and written to the other
// LDS buffer afterwards (commit_rows) -- one barrier per tile, global latency hidden.
template <int NW> struct RowStage { float4 r[(TILE * 9 + NW * 64 - 1) / (NW * 64)]; };

template <int NW>
__device__ __forceinline__ void issue_rows(RowStage<NW> &st, const float *base, long row_stride,
                                           int row0, int nrows) {
#pragma unroll
  for (int t = 0; t < (TILE * 9 + NW * 64 - 1) / (NW * 64); ++t) {
    const int i = threadIdx.x + NW * 64 * t;
    const int row = i / 9, c4 = i - row * 9;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < TILE * 9 && row0 + row < nrows)
      v = *reinterpret_cast<const float4 *>(base + (long)(row0 + row) * row_stride + 4 * c4);
    st.r[t] = v;
  }
}

template <int NW>
__device__ __forceinline__ void commit_rows(float *lds, const RowStage<NW> &st) {
#pragma unroll
  for (int t = 0; t < (TILE * 9 + NW * 64 - 1) / (NW * 64); ++t) {
    const int i = threadIdx.x + NW * 64 * t;
    const int row = i / 9, c4 = i - row * 9;
    if (i < TILE * 9) *reinterpret_cast<float4 *>(lds + row * HD + 4 * c4) = st.r[t];
  }
}

// Dead-key flags of one 64-key tile, packed 4 per word: byte r of word w is 1 when
// key k0 + 4w + r is padding (>= Lk) or masked.  Staged through LDS so that the
// masking in the hot loop is branch-free (no predicated byte loads).
__device__ __forceinline__ void stage_dead(unsigned *flags, const unsigned char *mrow, int k0, int Lk) {
  if (threadIdx.x < TILE / 4) {
    unsigned w = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = k0 + 4 * (int)threadIdx.x + r;
      unsigned dead = key >= Lk ? 1u : 0u;
      if (key < Lk && mrow) dead = mrow[key] ? 1u : 0u;
      w |= dead << (8 * r);
    }
    flags[threadIdx.x] = w;
  }
}

// ---- LDS operand layouts ---------------------------------------------------------------------
// Row operand (contracted over the 36 head dims, e.g. K in S^T = K Q^T): plain rows of 36 floats.
// MFMA k-step s of lane group g contracts dim 8g+s for s < 8 and dim 32+g for s = 8 (any
// permutation of the contraction index is fine as long as both operands use it: the register
// operand is loaded as q[8g+s] / q[32+g]).  A lane's first 8 values are then CONTIGUOUS and
// 16-byte aligned: two ds_read_b128 + one ds_read_b32 instead of nine ds_read_b32, and the row
// stride 36 = 4*9 makes the 16 rows of a quarter-wave hit 16 different 4-bank groups.
// Transposed operand (contracted over the tile's 64 rows, e.g. V in O^T += V^T P^T): [dim][row],
// row stride LDT = 68; the 4 rows 16j+4g+t of k-steps t = 0..3 are one ds_read_b128.
constexpr int LDT = 68;

// the lane's 9 contraction values of one row (see above)
__device__ __forceinline__ void load_row_operand(float (&r)[KSTEPS], const float *row, int g) {
  const float4 x = *reinterpret_cast<const float4 *>(row + 8 * g);
  const float4 y = *reinterpret_cast<const float4 *>(row + 8 * g + 4);
  r[0] = x.x; r[1] = x.y; r[2] = x.z; r[3] = x.w;
  r[4] = y.x; r[5] = y.y; r[6] = y.z; r[7] = y.w;
  r[8] = row[32 + g];
}

// Clamped staging loads for the layouts below: rows beyond the end re-read the last valid row
// (their scores are masked dead / their probabilities are zero, the values only have to be
// finite), so the loads need no per-tile predicate; addresses are a uniform base plus a 32-bit
// per-thread offset (no 64-bit address registers held across the tile).
template <int NW>
__device__ __forceinline__ void issue_rows_clamped(RowStage<NW> &st, const float *base, unsigned row_stride,
                                                   int row0, int nrows) {
#pragma unroll
  for (int t = 0; t < (TILE * 9 + NW * 64 - 1) / (NW * 64); ++t) {
    const int i = threadIdx.x + NW * 64 * t;
    const int row = i / 9, c4 = i - row * 9;
    st.r[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < TILE * 9) {
      const unsigned r = (unsigned)min(row0 + row, nrows - 1);
      st.r[t] = *reinterpret_cast<const float4 *>(base + (r * row_stride + 4u * (unsigned)c4));
    }
  }
}

template <int NW>
__device__ __forceinline__ void commit_rows_transposed(float *lds, const RowStage<NW> &st) {
#pragma unroll
  for (int t = 0; t < (TILE * 9 + NW * 64 - 1) / (NW * 64); ++t) {
    const int i = threadIdx.x + NW * 64 * t;
    const int row = i / 9, c4 = i - row * 9;
    if (i < TILE * 9) {
      float *dst = lds + (4 * c4) * LDT + row;
      dst[0] = st.r[t].x;
      dst[LDT] = st.r[t].y;
      dst[2 * LDT] = st.r[t].z;
      dst[3 * LDT] = st.r[t].w;
    }
  }
}

struct DropCfg { bool on; unsigned seed, thresh; float inv_keep; };

#ifdef EDA_MHA_PROFILE
// Phase timing (experiments only): per-wave s_memtime stamps, forced behind the phase's last result.
__device__ unsigned long long mha_prof[8];
#define MHA_STAMP(slot, dep)                                                     \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    const int dep__ = __builtin_amdgcn_readfirstlane(__float_as_int(dep));       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"s"(dep__));                  \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime();               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(now__));                           \
    prof_acc[slot] += now__ - prof_t;                                            \
    prof_t = now__;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
#define MHA_PROF_ARGS , unsigned long long (&prof_acc)[8], unsigned long long &prof_t
#define MHA_PROF_PASS , prof_acc, prof_t
#else
#define MHA_STAMP(slot, dep) do { } while (0)
#define MHA_PROF_ARGS
#define MHA_PROF_PASS
#endif

// One 64-key tile of the forward for one wave (16 queries): NSUB = live 16-key sub-tiles
// (compile-time, so the body is branch-free: the four score chains interleave and every LDS
// operand is requested before the first MFMA that needs it).
template <int NSUB, class StageNext>
__device__ __forceinline__ void fwd_tile(const float *__restrict__ Kl, const float *__restrict__ Vl,
                                         const unsigned *__restrict__ deadl, const float (&qreg)[KSTEPS],
                                         int c, int g, int k0, unsigned rowbase, const DropCfg &dc,
                                         float &m, float &lsum, f32x4 (&o)[3], StageNext &&stage_next MHA_PROF_ARGS) {
  float kreg[NSUB][KSTEPS];
#pragma unroll
  for (int j = 0; j < NSUB; ++j) load_row_operand(kreg[j], Kl + (16 * j + c) * HD, g);
  f32x4 st[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int j = 0; j < NSUB; ++j) st[j] = mfma4(kreg[j][s], qreg[s], st[j]);
  __builtin_amdgcn_sched_barrier(0);     // (keeps the V operand's 48 registers from overlapping the K operand's 36)
  MHA_STAMP(1, st[NSUB - 1][3]);
  // V^T operand: dims c, 16+c, 32+(c&3) (output rows >= 36 of the third tile are never stored,
  // so lanes c >= 4 may read any in-bounds row), keys 16j + 4g + (0..3); sub-tile 0 is requested
  // now (it lands under the softmax arithmetic), sub-tile j+1 under the MFMAs of sub-tile j.
  const float *vp = Vl + c * LDT + 4 * g;
  const float *vp2 = Vl + (32 + (c & 3)) * LDT + 4 * g;
  f32x4 va[3], vb[3];
  va[0] = *reinterpret_cast<const f32x4 *>(vp);
  va[1] = *reinterpret_cast<const f32x4 *>(vp + 16 * LDT);
  va[2] = *reinterpret_cast<const f32x4 *>(vp2);
  float tmax = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned dw = deadl[4 * j + g];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool dead = ((dw >> (8 * r)) & 0xffu) != 0u;
      st[j][r] = dead ? -INFINITY : st[j][r];
      tmax = fmaxf(tmax, st[j][r]);
    }
  }
  tmax = xor_max(tmax);
  const float m_new = fmaxf(m, tmax);
  const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
  const float alpha = __expf(m - m_safe);       // m = -inf -> 0
  float psum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __expf(st[j][r] - m_safe);
      st[j][r] = p;
      psum += p;
    }
  psum = xor_sum(psum);
  lsum = lsum * alpha + psum;
  m = m_new;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) o[nt] *= alpha;
  if (dc.on) {
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
      for (int r2 = 0; r2 < 4; r2 += 2) {
        const unsigned h = hash32(dc.seed ^ (rowbase + (unsigned)(k0 + 16 * j + 4 * g + r2)));
        st[j][r2] = (h & 0xffffu) >= dc.thresh ? st[j][r2] * dc.inv_keep : 0.f;
        st[j][r2 + 1] = (h >> 16) >= dc.thresh ? st[j][r2 + 1] * dc.inv_keep : 0.f;
      }
  }
  MHA_STAMP(2, st[0][0] + lsum + va[2][3]);
  // the next tile's global loads are issued HERE (their registers are only live across the PV
  // MFMAs, which cover the L2 latency) rather than at the top of the tile
  __builtin_amdgcn_sched_barrier(0);
  stage_next();
  __builtin_amdgcn_sched_barrier(0);
  MHA_STAMP(0, lsum);
  // O^T[dim][query] += V^T P^T
#pragma unroll
  for (int j = 0; j < NSUB; ++j) {
    if (j + 1 < NSUB) {
      vb[0] = *reinterpret_cast<const f32x4 *>(vp + 16 * (j + 1));
      vb[1] = *reinterpret_cast<const f32x4 *>(vp + 16 * LDT + 16 * (j + 1));
      vb[2] = *reinterpret_cast<const f32x4 *>(vp2 + 16 * (j + 1));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float pb = st[j][t];
      o[0] = mfma4(va[0][t], pb, o[0]);
      o[1] = mfma4(va[1][t], pb, o[1]);
      o[2] = mfma4(va[2][t], pb, o[2]);
    }
    if (j + 1 < NSUB) {
      __builtin_amdgcn_sched_barrier(0);
      va[0] = vb[0]; va[1] = vb[1]; va[2] = vb[2];
    }
  }
  MHA_STAMP(3, o[0][0] + o[1][0] + o[2][0]);
}

// ============================================================== forward ======
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(3, 4))) void mha_fwd_kernel(MhaArgs a) {
  __shared__ __attribute__((aligned(16))) float Kbuf[2][TILE * HD];
  __shared__ __attribute__((aligned(16))) float Vbuf[2][HD * LDT];
  __shared__ unsigned deadbuf[2][TILE / 4];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  // grid = (B*H, row tiles): consecutive workgroup ids (round-robin over the 8 XCDs) are
  // different heads, so every row tile of one (scene, head) lands on ONE XCD and its K/V (Q/dO)
  // rows are fetched into one L2 instead of all eight.
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int qi = blockIdx.y * (NW * 16) + wave * 16 + c;
  const bool qvalid = qi < a.Lq;

  const float *qrow = a.q + (long)b * a.q_sb + (long)(qvalid ? qi : 0) * a.q_sl + h * HD;
  float qreg[KSTEPS];
  load_row_operand(qreg, qrow, g);
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) qreg[s] = qvalid ? qreg[s] * a.scale : 0.f;

  const float *kbase = a.k + (long)b * a.k_sb + h * HD;
  const float *vbase = a.v + (long)b * a.v_sb + h * HD;
  const unsigned char *mrow = a.mask ? a.mask + (long)b * a.Lk : nullptr;

  DropCfg dc = {a.p_drop > 0.f, 0u, 0u, 1.f};
  if (dc.on) {
    dc.seed = hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt);
    dc.thresh = (unsigned)((double)a.p_drop * 65536.0 + 0.5);
    dc.inv_keep = 1.f / (1.f - a.p_drop);
  }
  const unsigned rowbase = ((unsigned)bh * (unsigned)a.Lq + (unsigned)qi) * (unsigned)a.Lk;

  float m = -INFINITY, lsum = 0.f;
  f32x4 o[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

  if (a.Lk > 0) {
    RowStage<NW> ks, vs;
    issue_rows_clamped<NW>(ks, kbase, (unsigned)a.k_sl, 0, a.Lk);
    issue_rows_clamped<NW>(vs, vbase, (unsigned)a.v_sl, 0, a.Lk);
    commit_rows<NW>(Kbuf[0], ks);
    commit_rows_transposed<NW>(Vbuf[0], vs);
    stage_dead(deadbuf[0], mrow, 0, a.Lk);
  }
  __syncthreads();
#ifdef EDA_MHA_PROFILE
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prof_t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(prof_t));
#endif
  int cur = 0;
  for (int k0 = 0; k0 < a.Lk; k0 += TILE, cur ^= 1) {
    const float *Kl = Kbuf[cur], *Vl = Vbuf[cur];
    const unsigned *deadl = deadbuf[cur];
    const bool more = k0 + TILE < a.Lk;
    RowStage<NW> ks, vs;
    auto stage_next = [&]() {
      if (more) {
        issue_rows_clamped<NW>(ks, kbase, (unsigned)a.k_sl, k0 + TILE, a.Lk);
        issue_rows_clamped<NW>(vs, vbase, (unsigned)a.v_sl, k0 + TILE, a.Lk);
      }
    };
    // 16-key sub-tiles that lie entirely beyond Lk (e.g. 3 of 8 when Lk = 80) are skipped:
    // their probabilities are zero anyway (workgroup-uniform condition).
    const int nsub = min(4, (a.Lk - k0 + 15) / 16);
    if (nsub == 4) fwd_tile<4>(Kl, Vl, deadl, qreg, c, g, k0, rowbase, dc, m, lsum, o, stage_next MHA_PROF_PASS);
    else if (nsub == 3) fwd_tile<3>(Kl, Vl, deadl, qreg, c, g, k0, rowbase, dc, m, lsum, o, stage_next MHA_PROF_PASS);
    else if (nsub == 2) fwd_tile<2>(Kl, Vl, deadl, qreg, c, g, k0, rowbase, dc, m, lsum, o, stage_next MHA_PROF_PASS);
    else fwd_tile<1>(Kl, Vl, deadl, qreg, c, g, k0, rowbase, dc, m, lsum, o, stage_next MHA_PROF_PASS);
    if (more) {                       // registers -> the other LDS buffer (its readers finished last iteration)
      commit_rows<NW>(Kbuf[cur ^ 1], ks);
      commit_rows_transposed<NW>(Vbuf[cur ^ 1], vs);
      stage_dead(deadbuf[cur ^ 1], mrow, k0 + TILE, a.Lk);
    }
    MHA_STAMP(4, lsum);
    __syncthreads();
    MHA_STAMP(5, lsum);
#ifdef EDA_MHA_PROFILE
    prof_acc[6] += 1;
#endif
  }
#ifdef EDA_MHA_PROFILE
  if (lane == 0)
    for (int i = 0; i < 7; ++i) atomicAdd(&mha_prof[i], prof_acc[i]);
#endif

  if (qvalid) {
    const float inv = 1.f / lsum;                  // all keys masked -> NaN, like the reference
    float *orow = a.o + (long)b * a.o_sb + (long)qi * a.o_sl + h * HD;
    *reinterpret_cast<float4 *>(orow + 4 * g) = make_float4(o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv);
    *reinterpret_cast<float4 *>(orow + 16 + 4 * g) = make_float4(o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv);
    if (g == 0) {
      *reinterpret_cast<float4 *>(orow + 32) = make_float4(o[2][0] * inv, o[2][1] * inv, o[2][2] * inv, o[2][3] * inv);
      a.lse[(long)bh * a.Lq + qi] = m + __logf(lsum);
    }
  }
}

// The lane's transposed-operand values of one 16-row sub-tile, read from a ROW-major tile:
// element [t][n] = X[row 16j + 4g + t][dim c + 16n] (third tile: dim 32 + (c & 3); its output rows
// >= 36 are never stored).  12 conflict-free ds_read_b32 (the 16 lanes of a group read 16
// consecutive floats); requested one sub-tile ahead of the MFMAs that consume them.
struct ColOperand { float v[4][3]; };
__device__ __forceinline__ void load_col_operand(ColOperand &o, const float *tile, int j, int c, int g) {
  const float *p = tile + (16 * j + 4 * g) * HD;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    o.v[t][0] = p[t * HD + c];
    o.v[t][1] = p[t * HD + 16 + c];
    o.v[t][2] = p[t * HD + 32 + (c & 3)];
  }
}

// ======================================================== backward: dQ =======
// lane owns query (l&15); streams K/V tiles.
//   S^T = K Q^T, P^T = exp(S^T - lse);  dP^T = V dO^T;  dS^T = P^T o (dP^T_eff - delta)
//   dQ^T[dim][query] += K^T dS^T    (then * scale)
// One 64-key tile for one wave; NSUB = live 16-key sub-tiles (compile time: branch-free body).
template <int NSUB, class StageNext>
__device__ __forceinline__ void dq_tile(const float *__restrict__ Kl, const float *__restrict__ Vl,
                                        const unsigned *__restrict__ deadl, const float (&qreg)[KSTEPS],
                                        const float (&dreg)[KSTEPS], int c, int g, int k0, unsigned rowbase,
                                        const DropCfg &dc, bool qvalid, float lse, float delta,
                                        f32x4 (&dq)[3], StageNext &&stage_next) {
  f32x4 ds[4];
  // scores and dP two sub-tiles at a time: four independent accumulator chains
#pragma unroll
  for (int j0 = 0; j0 < NSUB; j0 += 2) {
    float ka[2][KSTEPS], va[2][KSTEPS];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (j0 + jj < NSUB) {
        load_row_operand(ka[jj], Kl + (16 * (j0 + jj) + c) * HD, g);
        load_row_operand(va[jj], Vl + (16 * (j0 + jj) + c) * HD, g);
      }
    f32x4 sacc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, pacc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (j0 + jj < NSUB) {
          sacc[jj] = mfma4(ka[jj][s], qreg[s], sacc[jj]);
          pacc[jj] = mfma4(va[jj][s], dreg[s], pacc[jj]);
        }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (j0 + jj < NSUB) {
        const int j = j0 + jj;
        const unsigned dw = deadl[4 * j + g];
        unsigned hpair[2] = {0u, 0u};
        if (dc.on) {
          hpair[0] = hash32(dc.seed ^ (rowbase + (unsigned)(k0 + 16 * j + 4 * g)));
          hpair[1] = hash32(dc.seed ^ (rowbase + (unsigned)(k0 + 16 * j + 4 * g + 2)));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = (((dw >> (8 * r)) & 0xffu) != 0u) || !qvalid;
          const float p = dead ? 0.f : __expf(sacc[jj][r] - lse);
          float dp = pacc[jj][r];
          if (dc.on) {
            const unsigned r16 = (r & 1) ? hpair[r >> 1] >> 16 : hpair[r >> 1] & 0xffffu;
            dp = r16 >= dc.thresh ? dp * dc.inv_keep : 0.f;
          }
          ds[j][r] = p * (dp - delta);
        }
      }
  }
  ColOperand ca, cb;
  __builtin_amdgcn_sched_barrier(0);
  stage_next();                      // next tile's global loads: live only across the MFMAs below
  __builtin_amdgcn_sched_barrier(0);
  load_col_operand(ca, Kl, 0, c, g);
#pragma unroll
  for (int j = 0; j < NSUB; ++j) {
    if (j + 1 < NSUB) load_col_operand(cb, Kl, j + 1, c, g);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float sb = ds[j][t];
      dq[0] = mfma4(ca.v[t][0], sb, dq[0]);
      dq[1] = mfma4(ca.v[t][1], sb, dq[1]);
      dq[2] = mfma4(ca.v[t][2], sb, dq[2]);
    }
    if (j + 1 < NSUB) {
      __builtin_amdgcn_sched_barrier(0);
      ca = cb;
    }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(3, 4))) void mha_bwd_dq_kernel(MhaArgs a) {
  __shared__ __attribute__((aligned(16))) float Kbuf[2][TILE * HD];
  __shared__ __attribute__((aligned(16))) float Vbuf[2][TILE * HD];
  __shared__ unsigned deadbuf[2][TILE / 4];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  // grid = (B*H, row tiles): consecutive workgroup ids (round-robin over the 8 XCDs) are
  // different heads, so every row tile of one (scene, head) lands on ONE XCD and its K/V (Q/dO)
  // rows are fetched into one L2 instead of all eight.
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int qi = blockIdx.y * (NW * 16) + wave * 16 + c;
  const bool qvalid = qi < a.Lq;

  const float *qrow = a.q + (long)b * a.q_sb + (long)(qvalid ? qi : 0) * a.q_sl + h * HD;
  const float *drow = a.dout + (long)b * a.do_sb + (long)(qvalid ? qi : 0) * a.do_sl + h * HD;
  float qreg[KSTEPS], dreg[KSTEPS];
  load_row_operand(qreg, qrow, g);
  load_row_operand(dreg, drow, g);
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    qreg[s] = qvalid ? qreg[s] * a.scale : 0.f;
    dreg[s] = qvalid ? dreg[s] : 0.f;
  }
  const float lse = qvalid ? a.lse[(long)bh * a.Lq + qi] : 0.f;
  // delta = rowsum(dO * O): each of the 4 lane groups holds 9 of the 36 head dims of its query;
  // computed here (and published for the dK/dV kernel, which runs after this one) instead of
  // in a separate pre-pass launch.
  float delta = 0.f;
  {
    const float *orow = a.o + (long)b * a.o_sb + (long)(qvalid ? qi : 0) * a.o_sl + h * HD;
    float oreg[KSTEPS];
    load_row_operand(oreg, orow, g);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) delta += dreg[s] * oreg[s];
    delta = xor_sum(delta);
    if (qvalid && g == 0) const_cast<float *>(a.delta)[(long)bh * a.Lq + qi] = delta;
  }

  const float *kbase = a.k + (long)b * a.k_sb + h * HD;
  const float *vbase = a.v + (long)b * a.v_sb + h * HD;
  const unsigned char *mrow = a.mask ? a.mask + (long)b * a.Lk : nullptr;
  DropCfg dc = {a.p_drop > 0.f, 0u, 0u, 1.f};
  if (dc.on) {
    dc.seed = hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt);
    dc.thresh = (unsigned)((double)a.p_drop * 65536.0 + 0.5);
    dc.inv_keep = 1.f / (1.f - a.p_drop);
  }
  const unsigned rowbase = ((unsigned)bh * (unsigned)a.Lq + (unsigned)qi) * (unsigned)a.Lk;

  f32x4 dq[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  // key range of this workgroup: everything, or split blockIdx.z of the range
  const int kbeg = a.k_split_rows ? (int)blockIdx.z * a.k_split_rows : 0;
  const int kend = a.k_split_rows ? min(a.Lk, kbeg + a.k_split_rows) : a.Lk;
  if (kbeg < kend) {
    RowStage<NW> ks, vs;
    issue_rows_clamped<NW>(ks, kbase, (unsigned)a.k_sl, kbeg, a.Lk);
    issue_rows_clamped<NW>(vs, vbase, (unsigned)a.v_sl, kbeg, a.Lk);
    commit_rows<NW>(Kbuf[0], ks);
    commit_rows<NW>(Vbuf[0], vs);
    stage_dead(deadbuf[0], mrow, kbeg, a.Lk);
  }
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += TILE, cur ^= 1) {
    const float *Kl = Kbuf[cur], *Vl = Vbuf[cur];
    const unsigned *deadl = deadbuf[cur];
    const bool more = k0 + TILE < kend;
    RowStage<NW> ks, vs;
    auto stage_next = [&]() {
      if (more) {
        issue_rows_clamped<NW>(ks, kbase, (unsigned)a.k_sl, k0 + TILE, a.Lk);
        issue_rows_clamped<NW>(vs, vbase, (unsigned)a.v_sl, k0 + TILE, a.Lk);
      }
    };
    const int nsub = min(4, (a.Lk - k0 + 15) / 16);
    if (nsub == 4) dq_tile<4>(Kl, Vl, deadl, qreg, dreg, c, g, k0, rowbase, dc, qvalid, lse, delta, dq, stage_next);
    else if (nsub == 3) dq_tile<3>(Kl, Vl, deadl, qreg, dreg, c, g, k0, rowbase, dc, qvalid, lse, delta, dq, stage_next);
    else if (nsub == 2) dq_tile<2>(Kl, Vl, deadl, qreg, dreg, c, g, k0, rowbase, dc, qvalid, lse, delta, dq, stage_next);
    else dq_tile<1>(Kl, Vl, deadl, qreg, dreg, c, g, k0, rowbase, dc, qvalid, lse, delta, dq, stage_next);
    if (more) {
      commit_rows<NW>(Kbuf[cur ^ 1], ks);
      commit_rows<NW>(Vbuf[cur ^ 1], vs);
      stage_dead(deadbuf[cur ^ 1], mrow, k0 + TILE, a.Lk);
    }
    __syncthreads();
  }
  if (qvalid) {
    float *out = a.dq + (long)b * a.dq_sb + (long)qi * a.dq_sl + h * HD;
    if (a.k_split_rows)         // partial of this key split, dense (B,Lq,H*36); summed by mha_part_reduce_kernel
      out = a.dq_part + ((long)blockIdx.z * a.B * a.Lq + (long)b * a.Lq + qi) * (a.H * HD) + h * HD;
    const float sc = a.scale;
    *reinterpret_cast<float4 *>(out + 4 * g) = make_float4(dq[0][0] * sc, dq[0][1] * sc, dq[0][2] * sc, dq[0][3] * sc);
    *reinterpret_cast<float4 *>(out + 16 + 4 * g) = make_float4(dq[1][0] * sc, dq[1][1] * sc, dq[1][2] * sc, dq[1][3] * sc);
    if (g == 0)
      *reinterpret_cast<float4 *>(out + 32) = make_float4(dq[2][0] * sc, dq[2][1] * sc, dq[2][2] * sc, dq[2][3] * sc);
  }
}

// ===================================================== backward: dK, dV ======
// lane owns key (l&15); streams Q/dO tiles (plus their lse/delta).
//   S = Q K^T [query 4g+r][key l&15], P = exp(S - lse[query]);  dP = dO V^T
//   dV^T[dim][key] += dO^T P_drop;  dS = P o (dP_eff - delta[query]);  dK^T[dim][key] += Q^T dS
struct DkvCtx {
  int c, g, q0, ki, Lq, Lk;
  unsigned bh;
  bool kdead;
  float scale;
};

template <int NSUB, class StageNext>
__device__ __forceinline__ void dkv_tile(const float *__restrict__ Ql, const float *__restrict__ Dl,
                                         const float *__restrict__ lse_l, const float *__restrict__ delta_l,
                                         const float (&kreg)[KSTEPS], const float (&vreg)[KSTEPS],
                                         const DkvCtx &x, const DropCfg &dc, f32x4 (&dk)[3], f32x4 (&dv)[3],
                                         StageNext &&stage_next) {
  const int c = x.c, g = x.g;
  f32x4 pd[4], ds[4];     // dropped P (for dV) and dS (for dK), B-operand layout
#pragma unroll
  for (int j0 = 0; j0 < NSUB; j0 += 2) {
    float qa[2][KSTEPS], da[2][KSTEPS];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (j0 + jj < NSUB) {
        load_row_operand(qa[jj], Ql + (16 * (j0 + jj) + c) * HD, g);
        load_row_operand(da[jj], Dl + (16 * (j0 + jj) + c) * HD, g);
      }
    f32x4 sacc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, pacc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (j0 + jj < NSUB) {
          sacc[jj] = mfma4(qa[jj][s], kreg[s], sacc[jj]);
          pacc[jj] = mfma4(da[jj][s], vreg[s], pacc[jj]);
        }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (j0 + jj < NSUB) {
        const int j = j0 + jj;
        const float4 lse4 = *reinterpret_cast<const float4 *>(lse_l + 16 * j + 4 * g);
        const float4 del4 = *reinterpret_cast<const float4 *>(delta_l + 16 * j + 4 * g);
        const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
        const float del_r[4] = {del4.x, del4.y, del4.z, del4.w};
        // dropout bits: lanes c and c^1 (keys 2m, 2m+1) need the same four pair hashes (one per
        // query r); each computes two of them and they swap through a quad-permute DPP move
        unsigned h16[4] = {0u, 0u, 0u, 0u};
        if (dc.on) {
          const int odd = c & 1;
          const unsigned qa0 = (unsigned)(x.q0 + 16 * j + 4 * g + 2 * odd);
          const unsigned kev = (unsigned)(x.ki & ~1);
          const unsigned hx = hash32(dc.seed ^ ((x.bh * (unsigned)x.Lq + qa0) * (unsigned)x.Lk + kev));
          const unsigned hy = hash32(dc.seed ^ ((x.bh * (unsigned)x.Lq + qa0 + 1u) * (unsigned)x.Lk + kev));
          const unsigned nx = (unsigned)__builtin_amdgcn_mov_dpp((int)hx, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
          const unsigned ny = (unsigned)__builtin_amdgcn_mov_dpp((int)hy, 0xB1, 0xf, 0xf, true);
          const unsigned h0 = odd ? nx : hx, h1 = odd ? ny : hy, h2 = odd ? hx : nx, h3 = odd ? hy : ny;
          const int sh = 16 * odd;            // this lane's key is the odd one of its pair -> high half
          h16[0] = (h0 >> sh) & 0xffffu; h16[1] = (h1 >> sh) & 0xffffu;
          h16[2] = (h2 >> sh) & 0xffffu; h16[3] = (h3 >> sh) & 0xffffu;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = x.q0 + 16 * j + 4 * g + r;
          const bool dead = x.kdead || qq >= x.Lq;
          const float p = dead ? 0.f : __expf(sacc[jj][r] * x.scale - lse_r[r]);
          float dp = pacc[jj][r];
          float pdrop = p;
          if (dc.on) {
            const bool keep = h16[r] >= dc.thresh;
            dp = keep ? dp * dc.inv_keep : 0.f;
            pdrop = keep ? p * dc.inv_keep : 0.f;
          }
          pd[j][r] = pdrop;
          ds[j][r] = p * (dp - del_r[r]);
        }
      }
  }
  // second products as a pipeline of 2*NSUB steps (step 2j: dV += dO^T P with sub-tile j of dO,
  // step 2j+1: dK += Q^T dS with sub-tile j of Q); three rotating operand buffers, the operand of
  // step k+2 is requested while step k multiplies
  ColOperand buf[3];
  load_col_operand(buf[0], Dl, 0, c, g);
  load_col_operand(buf[1], Ql, 0, c, g);
  __builtin_amdgcn_sched_barrier(0);
  stage_next();                      // next tile's global loads: live only across the MFMAs below
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 2 * NSUB; ++k) {
    const int j = k >> 1;
    if (k + 2 < 2 * NSUB) load_col_operand(buf[(k + 2) % 3], (k & 1) ? Ql : Dl, j + 1, c, g);
    const ColOperand &cur = buf[k % 3];
    if ((k & 1) == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float pb = pd[j][t];
        dv[0] = mfma4(cur.v[t][0], pb, dv[0]);
        dv[1] = mfma4(cur.v[t][1], pb, dv[1]);
        dv[2] = mfma4(cur.v[t][2], pb, dv[2]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float sb = ds[j][t];
        dk[0] = mfma4(cur.v[t][0], sb, dk[0]);
        dk[1] = mfma4(cur.v[t][1], sb, dk[1]);
        dk[2] = mfma4(cur.v[t][2], sb, dk[2]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 4))) void mha_bwd_dkv_kernel(MhaArgs a) {
  __shared__ __attribute__((aligned(16))) float Qbuf[2][TILE * HD];
  __shared__ __attribute__((aligned(16))) float Dbuf[2][TILE * HD];
  __shared__ __attribute__((aligned(16))) float lsebuf[2][TILE];
  __shared__ __attribute__((aligned(16))) float deltabuf[2][TILE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;   // (B*H, key tiles): see mha_fwd_kernel
  const int ki = blockIdx.y * (NW * 16) + wave * 16 + c;
  const bool kvalid = ki < a.Lk;
  const bool kdead = !kvalid || (a.mask && a.mask[(long)b * a.Lk + (kvalid ? ki : 0)]);

  const float *krow = a.k + (long)b * a.k_sb + (long)(kvalid ? ki : 0) * a.k_sl + h * HD;
  const float *vrow = a.v + (long)b * a.v_sb + (long)(kvalid ? ki : 0) * a.v_sl + h * HD;
  float kreg[KSTEPS], vreg[KSTEPS];
  load_row_operand(kreg, krow, g);
  load_row_operand(vreg, vrow, g);
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    kreg[s] = kvalid ? kreg[s] : 0.f;
    vreg[s] = kvalid ? vreg[s] : 0.f;
  }
  const float *qbase = a.q + (long)b * a.q_sb + h * HD;
  const float *dbase = a.dout + (long)b * a.do_sb + h * HD;
  DropCfg dc = {a.p_drop > 0.f, 0u, 0u, 1.f};
  if (dc.on) {
    dc.seed = hash32((unsigned)(*a.seed_ptr) * 0x9E3779B1u + a.salt);
    dc.thresh = (unsigned)((double)a.p_drop * 65536.0 + 0.5);
    dc.inv_keep = 1.f / (1.f - a.p_drop);
  }

  f32x4 dk[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 dv[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  auto stage_stats = [&](int buf, int q0) {
    if (threadIdx.x < TILE) {
      const int qq = q0 + threadIdx.x;
      lsebuf[buf][threadIdx.x] = qq < a.Lq ? a.lse[(long)bh * a.Lq + qq] : 0.f;
      deltabuf[buf][threadIdx.x] = qq < a.Lq ? a.delta[(long)bh * a.Lq + qq] : 0.f;
    }
  };
  // query range of this workgroup: everything, or split blockIdx.z of the range
  const int qbeg = a.q_split_rows ? (int)blockIdx.z * a.q_split_rows : 0;
  const int qend = a.q_split_rows ? min(a.Lq, qbeg + a.q_split_rows) : a.Lq;
  if (qbeg < qend) {
    RowStage<NW> qs, dsg;
    issue_rows_clamped<NW>(qs, qbase, (unsigned)a.q_sl, qbeg, a.Lq);
    issue_rows_clamped<NW>(dsg, dbase, (unsigned)a.do_sl, qbeg, a.Lq);
    commit_rows<NW>(Qbuf[0], qs);
    commit_rows<NW>(Dbuf[0], dsg);
    stage_stats(0, qbeg);
  }
  __syncthreads();
  DkvCtx x = {c, g, 0, ki, a.Lq, a.Lk, (unsigned)bh, kdead, a.scale};
  int cur = 0;
  for (int q0 = qbeg; q0 < qend; q0 += TILE, cur ^= 1) {
    const float *Ql = Qbuf[cur], *Dl = Dbuf[cur];
    const float *lse_l = lsebuf[cur], *delta_l = deltabuf[cur];
    const bool more = q0 + TILE < qend;
    RowStage<NW> qs, dsg;
    auto stage_next = [&]() {
      if (more) {
        issue_rows_clamped<NW>(qs, qbase, (unsigned)a.q_sl, q0 + TILE, a.Lq);
        issue_rows_clamped<NW>(dsg, dbase, (unsigned)a.do_sl, q0 + TILE, a.Lq);
      }
    };
    x.q0 = q0;
    const int nsub = min(4, (a.Lq - q0 + 15) / 16);
    if (nsub == 4) dkv_tile<4>(Ql, Dl, lse_l, delta_l, kreg, vreg, x, dc, dk, dv, stage_next);
    else if (nsub == 3) dkv_tile<3>(Ql, Dl, lse_l, delta_l, kreg, vreg, x, dc, dk, dv, stage_next);
    else if (nsub == 2) dkv_tile<2>(Ql, Dl, lse_l, delta_l, kreg, vreg, x, dc, dk, dv, stage_next);
    else dkv_tile<1>(Ql, Dl, lse_l, delta_l, kreg, vreg, x, dc, dk, dv, stage_next);
    if (more) {
      commit_rows<NW>(Qbuf[cur ^ 1], qs);
      commit_rows<NW>(Dbuf[cur ^ 1], dsg);
      stage_stats(cur ^ 1, q0 + TILE);
    }
    __syncthreads();
  }
  if (kvalid) {
    const float sc = a.scale;
    float *ok = a.dk + (long)b * a.dk_sb + (long)ki * a.dk_sl + h * HD;
    float *ov = a.dv + (long)b * a.dv_sb + (long)ki * a.dv_sl + h * HD;
    if (a.q_split_rows) {       // partial of this query split, dense (B,Lk,H*36); summed by mha_dkv_reduce_kernel
      const long per = (long)a.B * a.Lk * (a.H * HD);
      float *pbase = a.dkv_part + (long)blockIdx.z * 2 * per + ((long)b * a.Lk + ki) * (a.H * HD) + h * HD;
      ok = pbase; ov = pbase + per;
    }
    *reinterpret_cast<float4 *>(ok + 4 * g) = make_float4(dk[0][0] * sc, dk[0][1] * sc, dk[0][2] * sc, dk[0][3] * sc);
    *reinterpret_cast<float4 *>(ok + 16 + 4 * g) = make_float4(dk[1][0] * sc, dk[1][1] * sc, dk[1][2] * sc, dk[1][3] * sc);
    *reinterpret_cast<float4 *>(ov + 4 * g) = make_float4(dv[0][0], dv[0][1], dv[0][2], dv[0][3]);
    *reinterpret_cast<float4 *>(ov + 16 + 4 * g) = make_float4(dv[1][0], dv[1][1], dv[1][2], dv[1][3]);
    if (g == 0) {
      *reinterpret_cast<float4 *>(ok + 32) = make_float4(dk[2][0] * sc, dk[2][1] * sc, dk[2][2] * sc, dk[2][3] * sc);
      *reinterpret_cast<float4 *>(ov + 32) = make_float4(dv[2][0], dv[2][1], dv[2][2], dv[2][3]);
    }
  }
}

// out tensors (1: dq; 2: dk, dv) = sum over the splits of the dense partials [split][tensor][B][L][D],
// written with the outputs' strides, in split order (deterministic).
__global__ __launch_bounds__(256) void mha_part_reduce_kernel(const float *__restrict__ part, int nsplit,
                                                              int ntens, int B, int Lk, int D,
                                                              float *__restrict__ dk, long dk_sb, long dk_sl,
                                                              float *__restrict__ dv, long dv_sb, long dv_sl) {
  const long per = (long)B * Lk * D, n4 = per / 4;
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
  const long b = row / Lk, l = row - b * Lk;
  float *o = which ? dv + b * dv_sb + l * dv_sl + col : dk + b * dk_sb + l * dk_sl + col;
  *reinterpret_cast<float4 *>(o) = t;
}

// 4 waves per workgroup share one staged tile.  A 1-wave variant (4x more workgroups for the
// short decoder / text shapes) was measured slower on every EDA shape (rocprofv3 kernel
// durations, e.g. 256x1024 dK/dV 90 -> 152 us, 256x80 fwd 13.0 -> 16.9 us: each workgroup then
// stages all of K/V alone) and is no longer instantiated.
bool mult4(long v) { return (v & 3) == 0; }
// EDA_MHA_IMPL=1 selects the round-1/2 kernels of this file (64-row tiles, two-kernel backward) instead of
// csrc/mha2.hip (read once; for A/B measurements and as a cross-check in the tests)
int mha_impl() {
  static const int impl = [] { const char *e = getenv("EDA_MHA_IMPL"); return e ? atoi(e) : 2; }();
  return impl;
}
bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

int eda_mha_impl() { return mha_impl(); }

#ifdef EDA_MHA_PROFILE
extern "C" int eda_mha_profile_read(unsigned long long *out8) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mha_prof), sizeof(zero)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(mha_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif

extern "C" int eda_mha_fwd_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl,
                               long k_sb, long k_sl, long v_sb, long v_sl,
                               const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                               int head_dim, float scale, float p_drop,
                               const unsigned long long *seed_ptr, unsigned salt, float *out,
                               float *lse, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(head_dim == HD, "only head_dim 36 (d_model 288 / 8 heads) is built");
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "bad dimension");
  if (B == 0 || Lq == 0) return 0;
  EDA_CHECK_ARG(q && k && v && out && lse, "null pointer");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "bad dropout arguments");
  EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                    al16(q) && al16(k) && al16(v) && al16(out),
                "rows must be 16-byte aligned");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  if (mha_impl() != 1 && Lk > 0) {
    Mha2Args m = {};
    m.q = q; m.k = k; m.v = v; m.q_sb = q_sb; m.q_sl = q_sl; m.k_sb = k_sb; m.k_sl = k_sl;
    m.v_sb = v_sb; m.v_sl = v_sl; m.o = out; m.o_sb = (long)Lq * H * HD; m.o_sl = (long)H * HD;
    m.lse = lse; m.mask = key_padding_mask; m.B = B; m.H = H; m.Lq = Lq; m.Lk = Lk; m.scale = scale;
    m.p_drop = p_drop; m.seed_ptr = seed_ptr; m.salt = salt;
    return eda_mha2_fwd_launch(m, stream);
  }
  MhaArgs a = {};
  a.q = q; a.k = k; a.v = v; a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl;
  a.v_sb = v_sb; a.v_sl = v_sl; a.o = out; a.o_sb = (long)Lq * H * HD; a.o_sl = (long)H * HD;
  a.lse = lse; a.mask = key_padding_mask; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
  a.p_drop = p_drop; a.seed_ptr = seed_ptr; a.salt = salt;
  hipLaunchKernelGGL(mha_fwd_kernel<4>, dim3((unsigned)(B * H), (unsigned)((Lq + 63) / 64)), dim3(256), 0,
                     stream, a);
  EDA_CHECK_LAUNCH();
  return 0;
}

// Query splits for dK/dV: a short key dimension gives only B*H*ceil(Lk/64) workgroups that each
// walk ALL Lq queries (Lk = 80/132, Lq = 1024: 128-192 workgroups x 16 tiles, 108 us); splitting
// the query range over up to 4 workgroups fills the chip (measured: see DESIGN.md).  The dQ
// kernel uses the same rule with the roles swapped (short Lq, e.g. 80 text tokens over 1024
// points: the KEY range is split).
static int dkv_splits(int B, int H, int Lq, int Lk) {
  const long wgs = (long)B * H * ((Lk + TILE - 1) / TILE);
  if (wgs >= 256 || Lq < 4 * TILE) return 1;
  int s = (int)((512 + wgs - 1) / wgs);
  if (s > 4) s = 4;
  if (s > Lq / (2 * TILE)) s = Lq / (2 * TILE);      // at least two query tiles per split
  return s < 1 ? 1 : s;
}

extern "C" size_t eda_mha_bwd_workspace_bytes(int B, int H, int Lq, int Lk) {
  if (mha_impl() != 1) return eda_mha2_bwd_workspace_bytes(B, H, Lq, Lk);
  // one scratch area, used first by the dQ key split, then by the dK/dV query split
  const int s = dkv_splits(B, H, Lq, Lk), sq = dkv_splits(B, H, Lk, Lq);
  size_t need = 0;
  if (s > 1) need = sizeof(float) * (size_t)s * 2 * (size_t)B * Lk * (H * HD);
  if (sq > 1) {
    const size_t nq = sizeof(float) * (size_t)sq * (size_t)B * Lq * (H * HD);
    if (nq > need) need = nq;
  }
  return need;
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
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(head_dim == HD, "only head_dim 36 (d_model 288 / 8 heads) is built");
  EDA_CHECK_ARG(B >= 0 && H > 0 && Lq >= 0 && Lk >= 0, "bad dimension");
  if (B == 0) return 0;
  EDA_CHECK_ARG(q && k && v && out && lse && dout && delta_ws && dq && dk && dv, "null pointer");
  EDA_CHECK_ARG(mult4(q_sb) && mult4(q_sl) && mult4(k_sb) && mult4(k_sl) && mult4(v_sb) && mult4(v_sl) &&
                    mult4(do_sb) && mult4(do_sl) && mult4(dq_sb) && mult4(dq_sl) && mult4(dk_sb) &&
                    mult4(dk_sl) && mult4(dv_sb) && mult4(dv_sl) && al16(q) && al16(k) && al16(v) && al16(out) &&
                    al16(dout) && al16(dq) && al16(dk) && al16(dv),
                "rows must be 16-byte aligned");
  EDA_CHECK_ARG((long)B * H <= 65535, "B*H too large");
  if (mha_impl() != 1 && Lq > 0 && Lk > 0) {
    Mha2Args m = {};
    m.q = q; m.k = k; m.v = v; m.q_sb = q_sb; m.q_sl = q_sl; m.k_sb = k_sb; m.k_sl = k_sl;
    m.v_sb = v_sb; m.v_sl = v_sl; m.o = const_cast<float *>(out); m.o_sb = (long)Lq * H * HD; m.o_sl = (long)H * HD;
    m.lse = const_cast<float *>(lse); m.mask = key_padding_mask; m.B = B; m.H = H; m.Lq = Lq; m.Lk = Lk;
    m.scale = scale; m.p_drop = p_drop; m.seed_ptr = seed_ptr; m.salt = salt;
    m.dout = dout; m.do_sb = do_sb; m.do_sl = do_sl; m.dq = dq; m.dk = dk; m.dv = dv;
    m.dq_sb = dq_sb; m.dq_sl = dq_sl; m.dk_sb = dk_sb; m.dk_sl = dk_sl; m.dv_sb = dv_sb; m.dv_sl = dv_sl;
    return eda_mha2_bwd_launch(m, ws, ws_bytes, stream);
  }
  MhaArgs a = {};
  a.q = q; a.k = k; a.v = v; a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl;
  a.v_sb = v_sb; a.v_sl = v_sl; a.lse = const_cast<float *>(lse); a.mask = key_padding_mask;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.p_drop = p_drop; a.seed_ptr = seed_ptr;
  a.salt = salt; a.dout = dout; a.do_sb = do_sb; a.do_sl = do_sl; a.delta = delta_ws; a.dq = dq;
  a.dk = dk; a.dv = dv;
  a.dq_sb = dq_sb; a.dq_sl = dq_sl; a.dk_sb = dk_sb; a.dk_sl = dk_sl; a.dv_sb = dv_sb; a.dv_sl = dv_sl;
  a.o = const_cast<float *>(out); a.o_sb = (long)Lq * H * HD; a.o_sl = (long)H * HD;
  const size_t ws_need = eda_mha_bwd_workspace_bytes(B, H, Lq, Lk);
  const bool ws_ok = ws && ws_bytes >= ws_need && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0;
  if (Lq > 0) {
    // (delta = rowsum(dO * O) is computed inside the dQ kernel and published for dK/dV)
    const int nsplit = dkv_splits(B, H, Lk, Lq);          // same rule with the roles of Lq / Lk swapped
    const bool split = nsplit > 1 && ws_ok;
    if (split) {
      a.k_split_rows = ((Lk + nsplit - 1) / nsplit + TILE - 1) / TILE * TILE;
      a.dq_part = reinterpret_cast<float *>(ws);
    }
    const unsigned gz = split ? (unsigned)((Lk + a.k_split_rows - 1) / a.k_split_rows) : 1u;
    hipLaunchKernelGGL(mha_bwd_dq_kernel<4>, dim3((unsigned)(B * H), (unsigned)((Lq + 63) / 64), gz), dim3(256),
                         0, stream, a);
    EDA_CHECK_LAUNCH();
    if (split) {
      const long items = (long)B * Lq * (H * HD) / 4;
      hipLaunchKernelGGL(mha_part_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                         a.dq_part, (int)gz, 1, B, Lq, H * HD, dq, dq_sb, dq_sl, dq, dq_sb, dq_sl);
      EDA_CHECK_LAUNCH();
    }
  }
  if (Lk > 0) {
    const int nsplit = dkv_splits(B, H, Lq, Lk);
    const bool split = nsplit > 1 && ws_ok;
    if (split) {
      a.q_split_rows = ((Lq + nsplit - 1) / nsplit + TILE - 1) / TILE * TILE;
      a.dkv_part = reinterpret_cast<float *>(ws);
    }
    const unsigned gz = split ? (unsigned)((Lq + a.q_split_rows - 1) / a.q_split_rows) : 1u;
    hipLaunchKernelGGL(mha_bwd_dkv_kernel<4>, dim3((unsigned)(B * H), (unsigned)((Lk + 63) / 64), gz), dim3(256),
                         0, stream, a);
    EDA_CHECK_LAUNCH();
    if (split) {
      const long items = 2L * B * Lk * (H * HD) / 4;
      hipLaunchKernelGGL(mha_part_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                         a.dkv_part, (int)gz, 2, B, Lk, H * HD, dk, dk_sb, dk_sl, dv, dv_sb, dv_sl);
      EDA_CHECK_LAUNCH();
    }
  }
  return 0;
}
