// gemm_frozen.hip -- row products against a FROZEN weight on the bf16 matrix pipe at fp32 accuracy (round 6).
//
// The site: the text encoder.  The reference builds RobertaModel.from_pretrained(...), sets requires_grad = False
// (models/bdetr.py:77-80) and reads last_hidden_state once per step (:170-175): 48 linear layers of 640 rows (8 utterances
// x 80 tokens) against 768 x 768 / 2304 / 3072 and 3072 x 768 weights that NEVER change.  On the fp32 matrix instruction
// (csrc/gemm.hip gemm_dma_kernel) they take 25-38 us each (hipBLASLt: 21-30), 1.5 ms per step on the second stream -- and
// cost the main stream ~0.85 ms of its time (profiles/r05_side_stream_interference.md: "the lever that is left is cheaper
// arithmetic").
//
// bf16 x 3: every fp32 value v = h + m + l exactly (three bf16 planes, 8 + 8 + 8 mantissa bits); a product is formed from
// the six plane products whose weight is >= 2^-24 (hh, hm, mh, hl, lh, mm) on v_mfma_f32_16x16x32_bf16, accumulated in fp32:
// fp32 accuracy (tests/test_gemm_frozen_gpu.py: the fp32-MFMA kernel's error bound against fp64) at 6 x 17 cycles per
// 16 x 16 x 32 block instead of 8 x 36.  VERDICT r05 item 2's lesson ("the in-kernel split costs what the MFMAs save -- the
// operand must arrive pre-split") is free here: the weight is frozen, its planes are written ONCE (eda_bf16x3_split_f32,
// eda_amd/roberta_fast.py at packing time) in the layout the kernel consumes; only the 640 activation rows are split in the
// launch, by the wave that owns them, once per 64 output columns.
//
// Kernel: workgroup = 4 waves x 16 rows = 64 rows x 64 output columns.  The weight planes of a 64-deep contraction chunk
// (3 x 64 x 64 bf16 = 24 KB) arrive by LDS-DMA into a double buffer (16-byte granules XOR-swizzled by row so that the
// 16-row fragment reads spread over the banks: the DMA's LDS side is linear, its global side is per lane); a wave's own 16
// activation rows come straight from global memory into the B-operand layout (lane (g, i): k = 8 g .. 8 g + 7 of row i, two
// 16-byte loads per 32-deep step, a ring of steps in flight), are split in registers and meet the four column tiles'
// weight fragments: 24 MFMAs per step and wave.  D = W-fragment (rows = output columns) x row fragment: a lane ends with
// four consecutive columns of one row (16-byte stores).  Epilogue: bias, ReLU / GELU (erf).  Block id -> (column tile, row
// block) keeps the row blocks of a column tile on ONE XCD (block % 8): its weight tile is fetched from HBM once per step.
#include "eda_common.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int FB_BN = 64, FB_KC = 64, FB_PF = 4;
#ifndef EDA_FROZEN_NSTG
#define EDA_FROZEN_NSTG 2
#endif
constexpr int FB_NSTG = EDA_FROZEN_NSTG;      // LDS ring slots of the weight chunks (2; 3 measured slower: 72 KB per workgroup = one workgroup less per CU)

// two fp32 values -> their three bf16 planes, packed pairwise (first value in the low half)
__device__ __forceinline__ void split_pk(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  f32x2 v = {a, b};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  v = v - f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  v = v - f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x4 mma(const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// w (N, K) fp32 rows -> planes [3][N][K] bf16 (h | m | l), K contiguous
__global__ __launch_bounds__(256) void bf16x3_split_kernel(const float *__restrict__ w, long ldw, int N, int K,
                                                           unsigned short *__restrict__ planes) {
  const long pairs = (long)N * (K / 2);
  const long plane = (long)N * K;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (long)gridDim.x * 256) {
    const long n = e / (K / 2);
    const int kp = (int)(e - n * (K / 2));
    const float2 v = *reinterpret_cast<const float2 *>(w + n * ldw + 2 * kp);
    unsigned h, m, l;
    split_pk(v.x, v.y, h, m, l);
    unsigned *o = reinterpret_cast<unsigned *>(planes) + e;
    o[0] = h; o[plane / 2] = m; o[plane] = l;
  }
}

struct FrozenArgs {
  const float *x; long ldx; long R; int K, N;
  const unsigned short *wp;          // [3][N][K]
  const float *bias; int act;        // 0 none, 1 ReLU, 2 GELU (erf)
  float *y; long ldy;
  int col_tiles, row_blocks;
};

template <int FB_NW>
__global__ __launch_bounds__(64 * FB_NW) void linear_frozen_b3_kernel(const FrozenArgs a) {
  constexpr int FB_BM = 16 * FB_NW;
  // LDS (dynamic): [FB_NSTG ring slots][3 planes][64 columns][64 k] bf16, rows of 128 bytes = 8 granules of 16 bytes, granule q of
  // row r at slot q ^ (r & 7).  Two slots.  (PMC, tools/frozen_pmc.py: waves sit in s_waitcnt / barriers for 46 % of their
  // cycles, the matrix pipe is 21 % busy; a THIRD slot -- chunk c + 2 requested while chunk c is multiplied -- made it slower,
  // 27.8 -> 29.0 us and 33 -> 46 us at 1040 rows: what hides the latency here is a third workgroup per CU, which 72 KB forbid.)
  extern __shared__ __attribute__((aligned(1024))) unsigned short Ws[];
  constexpr int STG = 3 * FB_BN * FB_KC, PW = 24 / FB_NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // block -> (column tile, row block): the row blocks of a column tile share block % 8 (= the XCD whose L2 holds its weight tile)
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ct_per = (a.col_tiles + 7) >> 3;                  // column tiles per XCD
  const int rb = q % a.row_blocks, cti = q / a.row_blocks;
  const int ct = cti * 8 + xcd;
  if (cti >= ct_per || ct >= a.col_tiles) return;
  const int n0 = ct * FB_BN;
  const long row0 = (long)rb * FB_BM + 16 * wave;
  const long row = row0 + r16;
  const float *xp = a.x + (row < a.R ? row : a.R - 1) * a.ldx + 8 * g;
  const long plane = (long)a.N * a.K;
  const int nchunks = a.K / FB_KC, nsteps = a.K / 32;

  // weight chunk c -> buffer c & 1: 3 planes x 64 rows x 8 granules = 1536 granules = 24 pieces of 64; 24 / NW per wave
  auto stage_w = [&](int c, int slot) {
    unsigned short *dst = Ws + slot * STG;
#pragma unroll
    for (int i = 0; i < 24 / FB_NW; ++i) {
      const int p = wave + FB_NW * i;                          // piece: 8 rows of one plane
      const int pl = p >> 3, r = 8 * (p & 7) + (lane >> 3);
      const int qg = (lane & 7) ^ (r & 7);                     // the global granule that belongs in this lane's LDS slot
      const unsigned short *src = a.wp + pl * plane + (long)(n0 + r) * a.K + c * FB_KC + 8 * qg;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(dst + 512 * p), 16, 0, 0);
    }
  };
  // the wave's row fragment of step s (8 fp32 per lane) -> ring slot
  float4 xr[FB_PF][2];
  auto load_x = [&](int s, int slot) {
    xr[slot][0] = *reinterpret_cast<const float4 *>(xp + 32 * s);
    xr[slot][1] = *reinterpret_cast<const float4 *>(xp + 32 * s + 4);
  };

  stage_w(0, 0);
#pragma unroll
  for (int s = 0; s < FB_PF - 1; ++s)
    if (s < nsteps) load_x(s, s);
  float4 bias_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bias_v[j] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + n0 + 16 * j + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  bool pre1 = false;
  if (FB_NSTG == 3 && nchunks > 1) { stage_w(1, 1); pre1 = true; }     // chunk 1 is requested LAST: the wait below leaves only it in flight
  // acc: the running 128-deep partial on the matrix pipe; tot: the sum of the partials (one fp32 addition per 128 of
  // contraction keeps the rounding of a 3072-deep sum at the fp32 kernels' level: blocked summation)
  f32x4 acc[4], tot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (pre1) { if (PW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // one chunk = two 32-deep steps; two chunks per trip so that the ring slots (step % 4) are compile-time registers
  int buf = 0;                                                 // ring slot of the chunk being multiplied
  auto chunk = [&](int c, auto parity) {
    constexpr int PAR = decltype(parity)::value;
    const bool ahead = c + FB_NSTG - 1 < nchunks;
    if (ahead) stage_w(c + FB_NSTG - 1, (buf + FB_NSTG - 1) % FB_NSTG);
    const unsigned short *wb = Ws + buf * STG;
    int young = 0;                                             // row loads issued behind this chunk's DMA (uniform)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = 2 * c + u;
      const int slot = 2 * PAR + u;                            // = s % 4
      if (s + FB_PF - 1 < nsteps) { load_x(s + FB_PF - 1, (slot + FB_PF - 1) % FB_PF); young += 2; }
      const float4 v0 = xr[slot][0], v1 = xr[slot][1];
      uint4 bh, bm, bl;
      split_pk(v0.x, v0.y, bh.x, bm.x, bl.x);
      split_pk(v0.z, v0.w, bh.y, bm.y, bl.y);
      split_pk(v1.x, v1.y, bh.z, bm.z, bl.z);
      split_pk(v1.z, v1.w, bh.w, bm.w, bl.w);
      uint4 ah[4], am[4], al[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 16 * j + r16;
        const int off = r * FB_KC + 8 * ((4 * u + g) ^ (r & 7));        // granule 4 u + g of row r, swizzled
        ah[j] = *reinterpret_cast<const uint4 *>(wb + off);
        am[j] = *reinterpret_cast<const uint4 *>(wb + FB_BN * FB_KC + off);
        al[j] = *reinterpret_cast<const uint4 *>(wb + 2 * FB_BN * FB_KC + off);
      }
      // six of the nine plane products, smallest first; product-major so that no MFMA waits for the one before it
#define FB_ROUND(A_, B_) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = mma(A_[j], B_, acc[j]);
      FB_ROUND(al, bh) FB_ROUND(ah, bl) FB_ROUND(am, bm) FB_ROUND(am, bh) FB_ROUND(ah, bm) FB_ROUND(ah, bh)
#undef FB_ROUND
    }
    // chunk c + 1 must have landed.  vmcnt counts in order: what this wave issued AFTER chunk c + 1's pieces may stay in
    // flight -- the pieces of the chunk requested at the top of this one (three-slot ring: PW of them) and `young` row loads.
    // Then the barrier: everybody has its pieces of chunk c + 1, and is done reading the slot the next request overwrites.
    const int keep = ((FB_NSTG == 3 && ahead) ? PW : 0) + young;
    switch (keep) {
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __syncthreads();
    buf = buf + 1 == FB_NSTG ? 0 : buf + 1;
  };
#pragma unroll 1
  for (int c = 0; c < nchunks; c += 2) {
    chunk(c, std::integral_constant<int, 0>{});
    if (c + 1 < nchunks) chunk(c + 1, std::integral_constant<int, 1>{});
#pragma unroll
    for (int j = 0; j < 4; ++j) { tot[j] += acc[j]; acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  if (row >= a.R) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + 16 * j + 4 * g;
    const float b4[4] = {bias_v[j].x, bias_v[j].y, bias_v[j].z, bias_v[j].w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = tot[j][e] + b4[e];
      if (a.act == 1) o[e] = fmaxf(o[e], 0.f);
      else if (a.act == 2) o[e] = 0.5f * o[e] * (1.f + erff(o[e] * 0.70710678118654752f));
    }
    *reinterpret_cast<float4 *>(a.y + row * a.ldy + col) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

extern "C" size_t eda_bf16x3_planes_bytes(int N, int K) { return (size_t)3 * N * K * sizeof(unsigned short); }

// w (N, K) fp32 rows (row stride ldw) -> planes: [3][N][K] bf16, v = h + m + l exactly
extern "C" int eda_bf16x3_split_f32(const float *w, long ldw, int N, int K, void *planes, void *stream_) {
  EDA_CHECK_ARG(N > 0 && K > 0 && K % 2 == 0 && ldw >= K && ldw % 2 == 0, "even contraction length");
  EDA_CHECK_ARG(w && planes && (reinterpret_cast<uintptr_t>(w) & 7u) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15u) == 0,
                "aligned pointers");
  const long pairs = (long)N * (K / 2);
  long blocks = (pairs + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bf16x3_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, w, ldw, N, K,
                     reinterpret_cast<unsigned short *>(planes));
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_linear_frozen_b3_supported(long R, int K, int N) { return R > 0 && K >= 64 && K % 64 == 0 && N >= 64 && N % 64 == 0; }

// y = act(x W^T + bias) with W given as the planes of eda_bf16x3_split_f32; act: 0 none, 1 ReLU, 2 GELU (erf)
extern "C" int eda_linear_frozen_b3_f32(const float *x, long ldx, long R, int K, const void *wplanes, int N, const float *bias,
                                        int act, float *y, long ldy, void *stream_) {
  EDA_CHECK_ARG(eda_linear_frozen_b3_supported(R, K, N), "K and N must be multiples of 64");
  EDA_CHECK_ARG(x && wplanes && y, "null pointer");
  EDA_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= K && ldy >= N && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 &&
                    (reinterpret_cast<uintptr_t>(y) & 15u) == 0 && (reinterpret_cast<uintptr_t>(wplanes) & 15u) == 0 &&
                    (!bias || (reinterpret_cast<uintptr_t>(bias) & 15u) == 0),
                "rows must be 16-byte aligned");
  EDA_CHECK_ARG(act >= 0 && act <= 2, "activation 0 / 1 / 2");
  FrozenArgs a = {};
  a.x = x; a.ldx = ldx; a.R = R; a.K = K; a.N = N; a.wp = reinterpret_cast<const unsigned short *>(wplanes);
  a.bias = bias; a.act = act; a.y = y; a.ldy = ldy;
  a.col_tiles = N / FB_BN;
  // 4 waves x 16 rows per workgroup; 8 waves sharing a weight chunk measured no better (profiles/r06_gemm_frozen.txt:
  // the launch is bound by what a 64-column tile has to ingest per chunk, not by the waves' overlap); EDA_FROZEN_NW=8 selects it
  const bool w8 = eda_knob(EDA_K_FROZEN_NW) == 8;
  const int bm = w8 ? 128 : 64;
  a.row_blocks = (int)((R + bm - 1) / bm);
  const int ct_per = (a.col_tiles + 7) / 8;
  const long grid = (long)8 * ct_per * a.row_blocks;
  EDA_CHECK_ARG(grid <= 0x7fffffffL, "too many workgroups");
  constexpr size_t lds = (size_t)FB_NSTG * 3 * FB_BN * FB_KC * sizeof(unsigned short);
  if (w8) {
    EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(&linear_frozen_b3_kernel<8>), lds));
    hipLaunchKernelGGL(linear_frozen_b3_kernel<8>, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream_, a);
  } else {
    EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(&linear_frozen_b3_kernel<4>), lds));
    hipLaunchKernelGGL(linear_frozen_b3_kernel<4>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream_, a);
  }
  EDA_CHECK_LAUNCH();
  return 0;
}
