// wgrad.hip -- weight gradient of a pointwise linear layer, dW = dY^T X (+ d(bias) = column
// sums of dY), for the shapes of this path: SMALL output (M x N = 64..864 x 128..288), LONG
// reduction (K = rows = 640..8192).  The library GEMM picks 32x32 macro-tiles with no K split
// for these and reaches 21-29 TFLOP/s fp32 (14 us at K=2048, 47 us at K=8192 for 288x288,
// tools/bench_dw.py); the ~135 of them per training step are a third of all GEMM time.
//
// Formulation: both operands are K-major exactly as they lie in memory (dY is (K,M) rows, X is
// (K,N) rows), which is the native operand order of v_mfma_f32_16x16x4_f32:
//   A: lane l holds dY[k + (l>>4)][m + (l&15)],  B: lane l holds X[k + (l>>4)][n + (l&15)],
// so a K-chunk is staged into LDS as a plain row copy (16-byte loads, no transposes anywhere).
//
// Decomposition: a workgroup of 12 waves (3 per SIMD) owns a 96x96 output tile (288, 576 and
// 864 are multiples of 96) for one K split; it walks its split in 64-row chunks with the next
// chunk's global loads in flight during the MFMAs.  The splits' partial tiles go to a
// workspace and a second small kernel adds them in split order (deterministic); with one
// split the first kernel writes dW directly.  Workgroups of the first column tile also sum
// their dY chunk's columns from LDS: d(bias) costs no extra pass over dY.
//
// Four kernel families live here: the one-at-a-time split-K form (wgrad_partial_kernel + wgrad_reduce_kernel), the GROUPED
// form that computes every queued weight gradient of a backward pass in one launch -- on fp32 MFMA (wgrad_grouped_kernel) or,
// by default since round 4, on the bf16 matrix pipe with exactly split operands and fp32-level accuracy
// (wgrad_grouped_bf16x3_kernel: see the comment in front of it) --, the prologue form of the fused SA / FP layers
// (wgrad_x_kernel), and the whole backward of an SA layer in one launch where its dW is a single tile (sa_layer_bwd_kernel:
// weight gradient + masked input gradient + BatchNorm-backward sums; sa_gather_layer_bwd_kernel + rows_scatter_add_kernel for
// the first layer of SA2).
#include "eda_common.h"
#include "gemm.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WG_T 96            // output tile edge
#define WG_KC 64           // K rows per LDS chunk
#define WG_THREADS 768     // 12 waves: 3 per SIMD
#define WG_LD 2            // float4 loads per thread per operand per chunk (64 rows * 24 float4 / 768)
#define WG_STRIDE 112      // LDS row stride in floats: 96 + 16 so that the four K rows of one MFMA
                           // operand read (lanes 0-15, 16-31, ...) fall on disjoint banks

__global__ __launch_bounds__(WG_THREADS) void wgrad_partial_kernel(
    const float *__restrict__ dy, long ld_dy, const float *__restrict__ x, long ld_x, long K, int M,
    int N, int chunks_per_split, int tiles_n, int ntiles, int nsplits, float *__restrict__ out,
    float *__restrict__ db_out, float *__restrict__ partial) {
  __shared__ float As[WG_KC][WG_STRIDE];
  __shared__ float Bs[WG_KC][WG_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // wave w owns a 16 x 48 strip of the tile: rows 16*wm.., columns 48*wh + {0,16,32}.  One A
  // operand feeds three MFMAs whose accumulators are independent (no dependent-issue stalls).
  const int wm = w >> 1, wh = w & 1;
  // XCD-aware placement: consecutive workgroup ids go round-robin to the 8 XCDs (each with its
  // own L2), so id%8 picks the XCD and ALL output tiles of a K split are given to one XCD:
  // the split's dY / X rows are then re-read from that L2 by the other tiles (3-9x re-use).
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = xcd + 8 * (j / ntiles);
  if (s >= nsplits) return;
  const int tile = j - (j / ntiles) * ntiles;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * WG_T, n0 = tn * WG_T;
  const long kbeg = (long)s * chunks_per_split * WG_KC;
  long kend = kbeg + (long)chunks_per_split * WG_KC;
  if (kend > K) kend = K;

  float4 ra[WG_LD], rb[WG_LD];
  int lrow[WG_LD], lcol[WG_LD];
#pragma unroll
  for (int i = 0; i < WG_LD; ++i) {
    const int idx = tid + WG_THREADS * i;
    lrow[i] = idx / (WG_T / 4);
    lcol[i] = (idx - lrow[i] * (WG_T / 4)) * 4;
  }
  auto fetch = [&](long k0) {
#pragma unroll
    for (int i = 0; i < WG_LD; ++i) {
      const long k = k0 + lrow[i];
      const bool rowok = k < kend;
      ra[i] = (rowok && m0 + lcol[i] < M) ? *reinterpret_cast<const float4 *>(dy + k * ld_dy + m0 + lcol[i])
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[i] = (rowok && n0 + lcol[i] < N) ? *reinterpret_cast<const float4 *>(x + k * ld_x + n0 + lcol[i])
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool active = (m0 + 16 * wm < M) && (n0 + 48 * wh < N);
  const bool do_db = db_out != nullptr && tn == 0 && tid < WG_T;
  float dbsum = 0.f;
  const float *ap = &As[lane >> 4][16 * wm + (lane & 15)];
  const float *bp = &Bs[lane >> 4][48 * wh + (lane & 15)];

  if (kbeg < kend) fetch(kbeg);
  for (long k0 = kbeg; k0 < kend; k0 += WG_KC) {
#pragma unroll
    for (int i = 0; i < WG_LD; ++i) {
      *reinterpret_cast<float4 *>(&As[lrow[i]][lcol[i]]) = ra[i];
      *reinterpret_cast<float4 *>(&Bs[lrow[i]][lcol[i]]) = rb[i];
    }
    __syncthreads();
    if (k0 + WG_KC < kend) fetch(k0 + WG_KC);
    if (active) {
      // all 16 K steps' operands into registers first (one LDS latency per chunk), then MFMAs
      float av[WG_KC / 4], bv[3][WG_KC / 4];
#pragma unroll
      for (int kk = 0; kk < WG_KC / 4; ++kk) {
        av[kk] = ap[kk * 4 * WG_STRIDE];
#pragma unroll
        for (int t = 0; t < 3; ++t) bv[t][kk] = bp[kk * 4 * WG_STRIDE + 16 * t];
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the loads ahead of the MFMAs (the scheduler sinks them)
#pragma unroll
      for (int kk = 0; kk < WG_KC / 4; ++kk) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[t][kk], acc[t], 0, 0, 0);
      }
    }
    if (do_db) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int r = 0; r < WG_KC; r += 4) {
        t0 += As[r][tid]; t1 += As[r + 1][tid]; t2 += As[r + 2][tid]; t3 += As[r + 3][tid];
      }
      dbsum += (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
  }
  // One split: this workgroup owns the final tile.  Otherwise its partial tile goes to the
  // workspace slab of split s ((M*N + M) floats: tile data, then the d(bias) partials).
  const long MN = (long)M * N;
  float *o = nsplits == 1 ? out : partial + (long)s * (MN + M);
  float *od = nsplits == 1 ? db_out : o + MN;
  if (active) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int col = n0 + 48 * wh + 16 * t + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * wm + 4 * (lane >> 4) + r;
        if (row < M && col < N) o[(long)row * N + col] = acc[t][r];
      }
    }
  }
  if (do_db && m0 + tid < M) od[m0 + tid] = dbsum;
}

// dW[e] = sum_s slab[s][e] (float4 per thread) and db[m] = sum_s slab[s][M*N + m], in split
// order (deterministic).  (Folding this into the first kernel -- last workgroup per tile, ticket
// counter -- was measured 2x SLOWER: one workgroup walking 16-32 slabs is latency-bound.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int S,
                                                           long MN, int M, float *__restrict__ dW,
                                                           float *__restrict__ db) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n4 = MN / 4, slab = MN + M;
  if (i < n4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int s = 0; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4 *>(partial + (long)s * slab + 4 * i);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4 *>(dW + 4 * i) = t;
  } else if (db != nullptr && i < n4 + M) {
    const int m = (int)(i - n4);
    float t = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) t += partial[(long)s * slab + MN + m];
    db[m] = t;
  }
}


// ---------------------------------------------------------------------------------------------
// Grouped form: ALL the weight gradients of a backward pass in one launch.
//
// A weight gradient is not needed until the optimizer step, so the host queues (dY, X) pairs
// during the backward (eda_amd/wgrad_queue.py) and hands them over together.  Work unit = one
// 96x96 tile of one TARGET (a weight matrix); a target may have several JOBS (a module applied
// at several places: their dY^T X are summed in the accumulators, in job order).  With
// ~1700 tiles in flight there is no K split, hence no partial tiles and no second pass, every
// element has ONE writer (deterministic), and the launch/latency floor that dominates the
// one-at-a-time kernels above (14 us for 0.34 GFLOP) is paid once.
//
// Descriptors (device arrays of 64-bit words, built by the host):
//   task   = {target (or -1: padding), tm, tn, unused}; task id % 8 = the XCD it runs on
//   target = {dW, db or 0, M, N, first job, job count, accumulate (0: store, 1: add to dW/db), unused}
//   job    = {dY, ld_dy, X, ld_x, K, unused x3}
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) void wgrad_grouped_kernel(const long long *__restrict__ tasks,
                                                                   const long long *__restrict__ targets,
                                                                   const long long *__restrict__ jobs) {
  __shared__ float As[WG_KC][WG_STRIDE];
  __shared__ float Bs[WG_KC][WG_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wh = w & 1;
  const long long *tk = tasks + (long)blockIdx.x * 4;
  if (tk[0] < 0) return;                   // padding task (the host pads the 8 per-XCD lists to one depth)
  const long long *tg = targets + tk[0] * 8;
  const int tm = (int)tk[1], tn = (int)tk[2];
  float *dW = reinterpret_cast<float *>(tg[0]);
  float *db = reinterpret_cast<float *>(tg[1]);
  const int M = (int)tg[2], N = (int)tg[3];
  const int job0 = (int)tg[4], njobs = (int)tg[5];
  const bool accumulate = tg[6] != 0;
  const int m0 = tm * WG_T, n0 = tn * WG_T;

  float4 ra[WG_LD], rb[WG_LD];
  int lrow[WG_LD], lcol[WG_LD];
#pragma unroll
  for (int i = 0; i < WG_LD; ++i) {
    const int idx = tid + WG_THREADS * i;
    lrow[i] = idx / (WG_T / 4);
    lcol[i] = (idx - lrow[i] * (WG_T / 4)) * 4;
  }
  f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool active = (m0 + 16 * wm < M) && (n0 + 48 * wh < N);
  const bool do_db = db != nullptr && tn == 0 && tid < WG_T;
  float dbsum = 0.f;
  const float *ap = &As[lane >> 4][16 * wm + (lane & 15)];
  const float *bp = &Bs[lane >> 4][48 * wh + (lane & 15)];

  for (int jb = 0; jb < njobs; ++jb) {
    const long long *jd = jobs + (long)(job0 + jb) * 8;
    const float *dy = reinterpret_cast<const float *>(jd[0]);
    const long ld_dy = (long)jd[1];
    const float *x = reinterpret_cast<const float *>(jd[2]);
    const long ld_x = (long)jd[3];
    const long K = (long)jd[4];
    auto fetch = [&](long k0) {
#pragma unroll
      for (int i = 0; i < WG_LD; ++i) {
        const long k = k0 + lrow[i];
        const bool rowok = k < K;
        ra[i] = (rowok && m0 + lcol[i] < M) ? *reinterpret_cast<const float4 *>(dy + k * ld_dy + m0 + lcol[i])
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[i] = (rowok && n0 + lcol[i] < N) ? *reinterpret_cast<const float4 *>(x + k * ld_x + n0 + lcol[i])
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    if (K > 0) fetch(0);
    for (long k0 = 0; k0 < K; k0 += WG_KC) {
#pragma unroll
      for (int i = 0; i < WG_LD; ++i) {
        *reinterpret_cast<float4 *>(&As[lrow[i]][lcol[i]]) = ra[i];
        *reinterpret_cast<float4 *>(&Bs[lrow[i]][lcol[i]]) = rb[i];
      }
      __syncthreads();
      if (k0 + WG_KC < K) fetch(k0 + WG_KC);
      if (active) {
        // operands of 4 K steps at a time: 16 registers instead of 64, which keeps the kernel at
        // <= 80 VGPRs = 6 waves per SIMD = TWO workgroups per CU (one hides the other's barriers,
        // LDS latency and global fetches; with 128 VGPRs a CU held one workgroup: 32 % of peak)
#pragma unroll
        for (int half = 0; half < 4; ++half) {
          float av[WG_KC / 16], bv[3][WG_KC / 16];
#pragma unroll
          for (int kk = 0; kk < WG_KC / 16; ++kk) {
            av[kk] = ap[(half * (WG_KC / 16) + kk) * 4 * WG_STRIDE];
#pragma unroll
            for (int t = 0; t < 3; ++t) bv[t][kk] = bp[(half * (WG_KC / 16) + kk) * 4 * WG_STRIDE + 16 * t];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kk = 0; kk < WG_KC / 16; ++kk) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[t][kk], acc[t], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (do_db) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll 2
        for (int r = 0; r < WG_KC; r += 4) {
          t0 += As[r][tid]; t1 += As[r + 1][tid]; t2 += As[r + 2][tid]; t3 += As[r + 3][tid];
        }
        dbsum += (t0 + t1) + (t2 + t3);
      }
      __syncthreads();
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int col = n0 + 48 * wh + 16 * t + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * wm + 4 * (lane >> 4) + r;
        if (row < M && col < N) {
          float *o = dW + (long)row * N + col;
          *o = accumulate ? *o + acc[t][r] : acc[t][r];
        }
      }
    }
  }
  if (do_db && m0 + tid < M) db[m0 + tid] = accumulate ? db[m0 + tid] + dbsum : dbsum;
}

// ---------------------------------------------------------------------------------------------
// The grouped kernel on the bf16 matrix pipe with fp32-level accuracy ("bf16 x 3").
//
// gfx950 runs v_mfma_f32_16x16x32_bf16 at 16x the rate of v_mfma_f32_16x16x4_f32 per contraction step, and the fp32
// kernel above is bound by the matrix pipe (MI355X_MICROARCH.md: 157 TFLOP/s fp32 vs 2.5 PFLOP/s bf16).  Every fp32
// operand value is split EXACTLY into three bf16 terms, v = h + m + l (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m):
// 3 x 8 significant bits cover fp32's 24; the subtractions are exact), and a product a b is evaluated as the six
// terms of weight >= 2^-16:  ah bh + ah bm + am bh + am bm + ah bl + al bh  -- every bf16 x bf16 product is exact in
// fp32, the accumulators are fp32, and the three dropped terms (am bl, al bm, al bl) are <= 2^-23 |a b|: the error of
// a contraction is that of an fp32 evaluation (measured in tests/test_wgrad_gpu.py against fp64, same bound as the
// fp32 kernel).  Six 16-cycle instructions per 32 contraction steps instead of eight 32-cycle ones: 2.7x less time
// on the matrix pipe.
//
// What it costs is the split (11 VALU operations per two values) and 16-bit operand tiles: each value is split ONCE,
// by the thread that stages it, and the three planes of a tile lie in LDS contraction-minor ([96 rows of the output
// tile][64 k] bf16, 128-byte rows) because a lane of the bf16 MFMA holds EIGHT consecutive contraction values of one
// row.  The staging thread therefore owns a 4 (k) x 4 (columns) block -- four 16-byte row loads, as coalesced as the
// fp32 kernel's -- and writes, per column and plane, its four k values as one 8-byte store.  16-byte granule g of row
// r is kept in slot g ^ ((r >> 1) & 7): with the lane -> block map below both the 8-byte stores (16-lane groups, 32
// banks) and the operand reads (ds_read_b128 service groups, 64 banks) are conflict-free.
typedef __bf16 wb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wb_f32x2 __attribute__((ext_vector_type(2)));
#define WB_PLANE_BYTES (WG_T * WG_KC * 2)        // one bf16 plane of a 96 x 64 operand tile

__device__ __forceinline__ unsigned wb_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(wb_f32x2{a, b}, wb_bf16x2));
}
// (a, b) -> packed bf16 pairs of the three terms
__device__ __forceinline__ void wb_split(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  h = wb_pk(a, b);
  a -= __uint_as_float(h << 16); b -= __uint_as_float(h & 0xffff0000u);
  m = wb_pk(a, b);
  a -= __uint_as_float(m << 16); b -= __uint_as_float(m & 0xffff0000u);
  l = wb_pk(a, b);
}
// workgroup barrier for LDS hand-overs that leaves the global prefetch in flight (__syncthreads() is also a fence, for
// which the compiler may drain vmcnt to 0, i.e. wait for the very loads the barrier should overlap)
__device__ __forceinline__ void wb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f32x4 wb_mma(const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wb_bf16x8, a), __builtin_bit_cast(wb_bf16x8, b), c, 0, 0, 0);
}

// Workgroup = 4 waves (one per SIMD), wave = 48 x 48 of the 96 x 96 tile (three row strips x three column tiles: 18
// operand reads feed 54 MFMAs per K step), two workgroups per CU by LDS (2 x 72 KB) = two waves per SIMD with up to 256
// registers each: while one workgroup splits and stages its next chunk (VALU, LDS stores) the other multiplies.
// Measured on the way (profiles/r04_experiments.md): the fp32 kernel's 12 waves x 16 x 48 under its 80-register cap
// spilled the address registers (every reload waited for the prefetch: 1.63 ms against 1.38 fp32); 6 waves x 32 x 48
// at 140-168 registers ran ONE workgroup per CU (six waves land 2 + 2 + 1 + 1 on the SIMDs from a varying start and a
// SIMD holding two such waves cannot take two more; SQ_WAVE_CYCLES: 5.3 waves per CU on average): 1.31 ms.  Tried on
// top of this form and measured equal (0.79-0.83 ms on tools/bench_wgrad_grouped.py's 95.6 GFLOP against 0.84): a second
// chunk of register prefetch, and 8-wave workgroups with staging / multiplying ROLES over a double-buffered LDS (one
// workgroup per CU) -- what bounds it is not latency: the staging waves ingest ~10 bytes per clock and CU, the
// beyond-L2 rate of MI355X_MICROARCH.md, with the matrix pipe 34 % and the VALU 40 % busy.
#define WB_THREADS 256
#define WB_ITEMS 3                                 // 4 x 4 blocks per thread and chunk (2 x 384 blocks / 256 threads)
__global__ __launch_bounds__(WB_THREADS, 2) void wgrad_grouped_bf16x3_kernel(
    const long long *__restrict__ tasks, const long long *__restrict__ targets, const long long *__restrict__ jobs) {
  // planes: dY h, m, l | X h, m, l
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * WB_PLANE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wh = w & 1;
  const long long *tk = tasks + (long)blockIdx.x * 4;
  if (tk[0] < 0) return;
  const long long *tg = targets + tk[0] * 8;
  const int tm = (int)tk[1], tn = (int)tk[2];
  float *dW = reinterpret_cast<float *>(tg[0]);
  float *db = reinterpret_cast<float *>(tg[1]);
  const int M = (int)tg[2], N = (int)tg[3];
  const int job0 = (int)tg[4], njobs = (int)tg[5];
  const bool accumulate = tg[6] != 0;
  const int m0 = tm * WG_T, n0 = tn * WG_T;

  // staging: block j = tid + 256 q of the chunk's 768 4 (k) x 4 (columns) blocks -- 0..383 of dY, 384..767 of X (the
  // operand of a (thread, q) is wave-uniform); block (rq, cq) = rows 4 rq.. of the chunk x columns 4 cq.. of the tile,
  // the 16 lanes of a store group span 4 rq x 4 cq
  int rq[WB_ITEMS], gcol[WB_ITEMS], soff[WB_ITEMS][4];
  bool isb[WB_ITEMS], cok[WB_ITEMS];
#pragma unroll
  for (int q = 0; q < WB_ITEMS; ++q) {
    const int item = tid + WB_THREADS * q;
    isb[q] = item >= 384;
    const int it = item - (isb[q] ? 384 : 0), lq = it & 15, blk = it >> 4;
    rq[q] = 4 * (blk / 6) + (lq & 3);
    const int cq = 4 * (blk % 6) + (lq >> 2);
    const int c0 = (isb[q] ? n0 : m0) + 4 * cq, lim = isb[q] ? N : M;
    cok[q] = c0 < lim;
    gcol[q] = cok[q] ? c0 : lim - 4;               // loads are unconditional: clamped here, zeroed when staged
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 4 * cq + u;
      soff[q][u] = (isb[q] ? 3 * WB_PLANE_BYTES : 0) + r * 128 + 16 * ((rq[q] >> 1) ^ ((r >> 1) & 7)) + 8 * (rq[q] & 1);
    }
  }
  const bool edge_tile = (m0 + WG_T > M) || (n0 + WG_T > N);                        // (uniform)
  // operand reads: lane (i, kg) holds row i of a 16-row strip, granule 4 s + kg of K step s (s = 1: offset ^ 64)
  const int li = lane & 15, kg = lane >> 4;
  int aoff[3], boff[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int ra_ = 48 * wm + 16 * t + li, rb_ = 48 * wh + 16 * t + li;
    aoff[t] = ra_ * 128 + 16 * (kg ^ ((ra_ >> 1) & 7));
    boff[t] = 3 * WB_PLANE_BYTES + rb_ * 128 + 16 * (kg ^ ((rb_ >> 1) & 7));
  }

  f32x4 acc[3][3];
#pragma unroll
  for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[a_][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool active = (m0 + 48 * wm < M) && (n0 + 48 * wh < N);
  const bool do_db = db != nullptr && tn == 0;
  float dbp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // (blocks q = 0 and, for threads < 128, q = 1 are dY's)
  f32x4 rv[WB_ITEMS][4];
  typedef const __attribute__((address_space(1))) unsigned char *gbytes;
  typedef const __attribute__((address_space(1))) f32x4 *gf4;

  for (int jb = 0; jb < njobs; ++jb) {
    const long long *jd = jobs + (long)(job0 + jb) * 8;
    const gbytes srca = reinterpret_cast<gbytes>(jd[0]), srcb = reinterpret_cast<gbytes>(jd[2]);
    const unsigned lda = 4u * (unsigned)jd[1], ldbb = 4u * (unsigned)jd[3];          // row strides in bytes (K * ld * 4 < 2^32)
    const int K = (int)jd[4];
    if (K <= 0) continue;
    // unconditional loads at 32-bit byte offsets from the (uniform) operand bases; rows beyond K are clamped to the
    // last one (zeroed when staged), so a prefetch past the end is harmless and no load sits behind a branch
    auto fetch = [&](int q, int k0) {
      const gbytes src = isb[q] ? srcb : srca;
      const unsigned ld = isb[q] ? ldbb : lda;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned r = (unsigned)min(k0 + 4 * rq[q] + i, K - 1);
        rv[q][i] = *reinterpret_cast<gf4>(src + (r * ld + 4u * (unsigned)gcol[q]));
      }
    };
#pragma unroll
    for (int q = 0; q < WB_ITEMS; ++q) fetch(q, 0);
    for (int k0 = 0; k0 < K; k0 += WG_KC) {
#pragma unroll
      for (int q = 0; q < WB_ITEMS; ++q) {
        if (edge_tile || k0 + WG_KC > K) {                              // (uniform branch)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (!cok[q] || k0 + 4 * rq[q] + i >= K) rv[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          unsigned h0, m0_, l0, h1, m1, l1;
          wb_split(rv[q][0][u], rv[q][1][u], h0, m0_, l0);
          wb_split(rv[q][2][u], rv[q][3][u], h1, m1, l1);
          *reinterpret_cast<uint2 *>(lds + soff[q][u]) = make_uint2(h0, h1);
          *reinterpret_cast<uint2 *>(lds + WB_PLANE_BYTES + soff[q][u]) = make_uint2(m0_, m1);
          *reinterpret_cast<uint2 *>(lds + 2 * WB_PLANE_BYTES + soff[q][u]) = make_uint2(l0, l1);
          if (q < 2 && do_db && !isb[q]) dbp[q][u] += (rv[q][0][u] + rv[q][1][u]) + (rv[q][2][u] + rv[q][3][u]);
        }
        // the block's rows of the NEXT chunk are requested as soon as its registers are free: in flight for the rest of
        // the staging and the whole multiply phase
        fetch(q, k0 + WG_KC);
      }
      wb_barrier();
      if (active) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const int x_ = s_ * 64;
          uint4 ah[3], am[3], al[3], b[3];
#pragma unroll
          for (int a_ = 0; a_ < 3; ++a_) {
            ah[a_] = *reinterpret_cast<const uint4 *>(lds + (aoff[a_] ^ x_));
            am[a_] = *reinterpret_cast<const uint4 *>(lds + WB_PLANE_BYTES + (aoff[a_] ^ x_));
            al[a_] = *reinterpret_cast<const uint4 *>(lds + 2 * WB_PLANE_BYTES + (aoff[a_] ^ x_));
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const uint4 *>(lds + (boff[t] ^ x_));                           // X h
#pragma unroll
          for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[a_][t] = wb_mma(al[a_], b[t], acc[a_][t]);
#pragma unroll
          for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[a_][t] = wb_mma(am[a_], b[t], acc[a_][t]);
#pragma unroll
          for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[a_][t] = wb_mma(ah[a_], b[t], acc[a_][t]);
#pragma unroll
          for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const uint4 *>(lds + WB_PLANE_BYTES + (boff[t] ^ x_));          // X m
#pragma unroll
          for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[a_][t] = wb_mma(am[a_], b[t], acc[a_][t]);
#pragma unroll
          for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[a_][t] = wb_mma(ah[a_], b[t], acc[a_][t]);
#pragma unroll
          for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const uint4 *>(lds + 2 * WB_PLANE_BYTES + (boff[t] ^ x_));      // X l
#pragma unroll
          for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[a_][t] = wb_mma(ah[a_], b[t], acc[a_][t]);
        }
      }
      wb_barrier();
    }
  }
  if (active) {
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int col = n0 + 48 * wh + 16 * t + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + 48 * wm + 16 * a_ + 4 * (lane >> 4) + r;
          if (row < M && col < N) {
            float *o = dW + (long)row * N + col;
            *o = accumulate ? *o + acc[a_][t][r] : acc[a_][t][r];
          }
        }
      }
  }
  if (do_db) {
    // column sums: 16 row-quad partials per column, added in row order (the planes are dead: LDS is the scratch)
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (!isb[q]) {
        const int it = tid + WB_THREADS * q, lq = it & 15, blk = it >> 4, cq = 4 * (blk % 6) + (lq >> 2);
        *reinterpret_cast<float4 *>(red + rq[q] * WG_T + 4 * cq) = make_float4(dbp[q][0], dbp[q][1], dbp[q][2], dbp[q][3]);
      }
    __syncthreads();
    if (tid < WG_T && m0 + tid < M) {
      float t_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t_ += red[r * WG_T + tid];
      db[m0 + tid] = accumulate ? db[m0 + tid] + t_ : t_;
    }
  }
}

// arithmetic of eda_wgrad_grouped_f32: 0 = fp32 MFMA (wgrad_grouped_kernel), 1 = bf16 x 3 (wgrad_grouped_bf16x3_kernel);
// default from EDA_WGRAD_BF16X3
static int g_wgrad_arith = -1;          // -1: from EDA_WGRAD_BF16X3 (eda_wgrad_set_arith overrides)
static int wgrad_arith() { return g_wgrad_arith < 0 ? (eda_knob(EDA_K_WGRAD_BF16X3) != 0) : g_wgrad_arith; }
void eda_wgrad_env_reset() { g_wgrad_arith = -1; }
extern "C" int eda_wgrad_set_arith(int mode) {
  if (mode < -1 || mode > 1) { eda_set_error("eda_wgrad_set_arith: mode must be -1 (default), 0 (fp32 MFMA) or 1 (bf16 x 3)"); return EDA_ERR_INVALID_ARG; }
  g_wgrad_arith = mode;
  return 0;
}

extern "C" int eda_wgrad_grouped_f32(const long long *tasks, int ntasks, const long long *targets,
                                     const long long *jobs, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(ntasks >= 0, "bad dimension");
  if (ntasks == 0) return 0;
  EDA_CHECK_ARG(tasks && targets && jobs, "null pointer");
  if (wgrad_arith() == 1)
    hipLaunchKernelGGL(wgrad_grouped_bf16x3_kernel, dim3((unsigned)ntasks), dim3(WB_THREADS), 0, stream, tasks, targets, jobs);
  else
    hipLaunchKernelGGL(wgrad_grouped_kernel, dim3((unsigned)ntasks), dim3(WG_THREADS), 0, stream, tasks, targets, jobs);
  EDA_CHECK_LAUNCH();
  return 0;
}

namespace {
struct WgPlan { int tiles_m, tiles_n, splits, cps; };
WgPlan wg_plan(long K, int M, int N) {
  WgPlan p;
  p.tiles_m = (M + WG_T - 1) / WG_T;
  p.tiles_n = (N + WG_T - 1) / WG_T;
  const long nchunks = (K + WG_KC - 1) / WG_KC;
  long target = eda_knob(EDA_K_WGRAD_WGS);     // 144: measured best for the 288x288 outputs (tools/bench_dw.py)
  if (target < 1) target = 1;
  long want = (target + p.tiles_m * p.tiles_n - 1) / (p.tiles_m * p.tiles_n);   // splits for ~target workgroups
  if (want > nchunks) want = nchunks;
  if (want < 1) want = 1;
  if (want >= 8) want = (want + 7) / 8 * 8;        // whole rounds of the 8 XCDs
  if (want > nchunks) want = nchunks;
  p.cps = (int)((nchunks + want - 1) / want);
  if (p.cps < 1) p.cps = 1;
  p.splits = (int)((nchunks + p.cps - 1) / p.cps);
  if (p.splits < 1) p.splits = 1;
  return p;
}
}  // namespace

extern "C" size_t eda_wgrad_workspace_bytes(long K, int M, int N) {
  if (K <= 0 || M <= 0 || N <= 0) return 0;
  const WgPlan p = wg_plan(K, M, N);
  if (p.splits == 1) return 0;
  return sizeof(float) * (size_t)p.splits * ((size_t)M * N + M);   // (M*N + M) % 4 == 0: rows stay 16-byte aligned
}

extern "C" int eda_wgrad_f32(const float *dy, long ld_dy, const float *x, long ld_x, long K, int M, int N,
                             float *dW, float *db, void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(K >= 0 && M > 0 && N > 0, "bad dimension");
  EDA_CHECK_ARG(M % 4 == 0 && N % 4 == 0 && ld_dy % 4 == 0 && ld_x % 4 == 0 && ld_dy >= M && ld_x >= N,
                "M, N and the row strides must be multiples of 4");
  EDA_CHECK_ARG(dW, "null pointer");
  if (K == 0) {
    const int z1 = eda_zero_async(dW, sizeof(float) * (size_t)M * N, stream);
    if (z1 || !db) return z1;
    return eda_zero_async(db, sizeof(float) * M, stream);
  }
  EDA_CHECK_ARG(dy && x, "null pointer");
  EDA_CHECK_ARG(((uintptr_t)dy % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dW % 16 == 0),
                "operands must be 16-byte aligned");
  const WgPlan p = wg_plan(K, M, N);
  const int ntiles = p.tiles_m * p.tiles_n;
  if (p.splits > 1 && (!ws || ws_bytes < eda_wgrad_workspace_bytes(K, M, N) || (uintptr_t)ws % 16 != 0)) {
    eda_set_error("wgrad: workspace too small or misaligned");
    return EDA_ERR_WORKSPACE;
  }
  const dim3 grid((unsigned)(8 * ntiles * ((p.splits + 7) / 8)));
  hipLaunchKernelGGL(wgrad_partial_kernel, grid, dim3(WG_THREADS), 0, stream, dy, ld_dy, x, ld_x, K, M, N,
                     p.cps, p.tiles_n, ntiles, p.splits, dW, db, reinterpret_cast<float *>(ws));
  EDA_CHECK_LAUNCH();
  if (p.splits > 1) {
    const long items = (long)M * N / 4 + (db ? M : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float *>(ws), p.splits, (long)M * N, M, dW, db);
    EDA_CHECK_LAUNCH();
  }
  return 0;
}


// ---------------------------------------------------------------------------------------------
// The split-K weight gradient with a PROLOGUE on the X operand: the fused set-abstraction / feature-
// propagation pipeline (sa_cl.hip) never writes the activated tensor relu(BN(z)) nor the grouped
// neighbourhood rows to HBM, so the weight gradient recomputes them while staging X into LDS --
// exactly what csrc/gemm.hip's forward kernel does with its row operand.  The K axis here is the
// row axis (up to 10^6 positions), always split; dY is read as stored (16-byte rows).
//
// Tile TM x TN (TM in {64, 96, 128} output rows = layer outputs, TN in {64, 128, 160} output
// columns = layer inputs; 160 holds the gather layer's 4 + 128 columns in one tile): the 96 x 96
// tile of the kernels above wasted 41-56 % of the MFMAs on the 64 / 128 / 256-wide SA layers.
// TM/16 row strips x 2 column halves of waves, a wave owns 16 x TN/2 (TN/32 accumulators fed by ONE
// A operand read each k step).
// DYP: the dY operand is not stored; it is the dz of a pooled last SharedMLP layer, formed from the layer's
// pre-activation, the pooling arg-max and the pooled gradient while staging (gemm.h: WgradXArgs::dy_pool)
// DYM = 2: dY is the masked gradient of a non-pooled layer's output, dz = ka*g + kb*z + kd formed while staging
// (WgradXArgs::dy_bn: the element-wise BatchNorm-backward pass is not run).
template <int TM, int TN, int XMODE, int DYM = 0>
__global__ __launch_bounds__(TM * 8) void wgrad_x_kernel(const WgradXArgs a, int chunks_per_split, int tiles_n,
                                                         int ntiles, int nsplits) {
  constexpr int THREADS = TM * 8;                  // (TM/16) x 2 waves
  constexpr int SA = TM + 16, SB = TN + 16;        // LDS row strides (= 16 mod 32: conflict-free b32 operand reads)
  constexpr int NA = 64 * (TM / 4), NB = 64 * (TN / 4);          // float4 slots per chunk
  constexpr int LDA = (NA + THREADS - 1) / THREADS, LDB = (NB + THREADS - 1) / THREADS;
  constexpr int NT = TN / 32;                      // accumulators per wave
  __shared__ float As[WG_KC][SA];
  __shared__ float Bs[WG_KC][SB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wh = w & 1;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = xcd + 8 * (j / ntiles);
  if (s >= nsplits) return;
  const int tile = j - (j / ntiles) * ntiles;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int M = a.M, N = a.N;
  const long K = a.R;
  const long kbeg = (long)s * chunks_per_split * WG_KC;
  long kend = kbeg + (long)chunks_per_split * WG_KC;
  if (kend > K) kend = K;
  if (kbeg >= kend) return;

  float4 ra[LDA], rb[LDB];
  int arow[LDA], acol[LDA], brow[LDB], bcol[LDB];
#pragma unroll
  for (int i = 0; i < LDA; ++i) {
    const int idx = tid + THREADS * i;
    arow[i] = idx / (TM / 4);
    acol[i] = (idx - arow[i] * (TM / 4)) * 4;
  }
#pragma unroll
  for (int i = 0; i < LDB; ++i) {
    const int idx = tid + THREADS * i;
    brow[i] = idx / (TN / 4);
    bcol[i] = (idx - brow[i] * (TN / 4)) * 4;
  }
  // column constants of the prologue (the column of a thread's float4 never changes)
  float4 bsc[LDB], bsh[LDB];
  if (XMODE == X_BNRELU) {
#pragma unroll
    for (int i = 0; i < LDB; ++i) {
      const int c = n0 + bcol[i] < N ? n0 + bcol[i] : N - 4;
      bsc[i] = *reinterpret_cast<const float4 *>(a.in_scale + c);
      bsh[i] = *reinterpret_cast<const float4 *>(a.in_shift + c);
    }
  }
  constexpr bool DYP = DYM == 1, DYB = DYM == 2;
  float4 qsc[DYP ? LDA : 1], qsh[DYP ? LDA : 1], qka[DYM ? LDA : 1], qkb[DYM ? LDA : 1], qkd[DYM ? LDA : 1];
  unsigned ram[DYP ? LDA : 1];
  float4 rdo[DYM ? LDA : 1];           // DYP: the pooled gradient of the row's group; DYB: the row's pre-activation
  int rrp[DYP ? LDA : 1];
  if (DYB) {
#pragma unroll
    for (int i = 0; i < LDA; ++i) {
      const int mc = m0 + acol[i] < M ? m0 + acol[i] : M - 4;
      const float *c = a.dy_consts + mc;
      qka[i] = *reinterpret_cast<const float4 *>(c + 2 * M);
      qkb[i] = *reinterpret_cast<const float4 *>(c + 3 * M);
      qkd[i] = *reinterpret_cast<const float4 *>(c + 4 * M);
    }
  }
  if (DYP) {
#pragma unroll
    for (int i = 0; i < LDA; ++i) {
      const int mc = m0 + acol[i] < M ? m0 + acol[i] : M - 4;
      const float *c = a.dy_consts + mc;
      qsc[i] = *reinterpret_cast<const float4 *>(c);
      qsh[i] = *reinterpret_cast<const float4 *>(c + M);
      qka[i] = *reinterpret_cast<const float4 *>(c + 2 * M);
      qkb[i] = *reinterpret_cast<const float4 *>(c + 3 * M);
      qkd[i] = *reinterpret_cast<const float4 *>(c + 4 * M);
    }
  }
  const bool cvec = (a.c_feat & 3) == 0;
  // plain rows that are not 16-byte addressable (odd channel counts): element loads
  const bool x_elem = XMODE == X_PLAIN && ((N & 3) != 0 || (a.ld_x & 3) != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15u) != 0);
  const long rows_per_scene = (long)a.m * a.ns;
  int gp[LDB];                         // X_GATHER: b*n_pts + point of the row being fetched NEXT
  auto fetch_idx = [&](long k0) {
#pragma unroll
    for (int i = 0; i < LDB; ++i) {
      long k = k0 + brow[i];
      if (k >= kend) k = kend - 1;
      const int b = (int)(k / rows_per_scene);
      gp[i] = b * a.n_pts + a.idx[k];
    }
  };
  float gx[LDB][3], gc[LDB][3];        // raw point / centre coordinates of the xyz quad (stage-time arithmetic)
  auto fetch = [&](long k0) {
#pragma unroll
    for (int i = 0; i < LDA; ++i) {
      long k = k0 + arow[i];
      if (k >= kend) k = kend - 1;                       // clamped: zeroed when staged
      if (arow[i] >= WG_KC) k = kend - 1;
      const int mc = m0 + acol[i] < M ? m0 + acol[i] : M - 4;
      if (DYP) {
        ra[i] = *reinterpret_cast<const float4 *>(a.dyz + k * M + mc);
        const long grp = k / a.dy_pool;
        rrp[i] = (int)(k - grp * a.dy_pool);
        ram[i] = *reinterpret_cast<const unsigned *>(a.dy_argmax + grp * M + mc);
        rdo[i] = *reinterpret_cast<const float4 *>(a.dy_dout + grp * M + mc);
      } else {
        ra[i] = *reinterpret_cast<const float4 *>(a.dy + k * a.ld_dy + mc);
        if (DYB) rdo[i] = *reinterpret_cast<const float4 *>(a.dyz + k * M + mc);
      }
    }
#pragma unroll
    for (int i = 0; i < LDB; ++i) {
      long k = k0 + brow[i];
      if (k >= kend || brow[i] >= WG_KC) k = kend - 1;
      const int nc = n0 + bcol[i];
      if (XMODE == X_GATHER) {
        const int C = a.c_feat;
        if (nc == 0) {
          const int b = (int)(k / rows_per_scene);
          const int cen = b * a.m + (int)((k - (long)b * rows_per_scene) / a.ns);
#pragma unroll
          for (int u = 0; u < 3; ++u) { gx[i][u] = a.xyz[(long)gp[i] * 3 + u]; gc[i][u] = a.new_xyz[(long)cen * 3 + u]; }
          rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (C == 0) {
          rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (cvec) {
          int f = nc - 4;
          if (f > C - 4) f = C - 4;
          rb[i] = *reinterpret_cast<const float4 *>(a.feats + (long)gp[i] * C + f);
        } else {
          float e[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { int f = nc - 4 + u; if (f > C - 1) f = C - 1; e[u] = a.feats[(long)gp[i] * C + f]; }
          rb[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      } else if (x_elem) {
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = nc + u < N ? nc + u : N - 1; e[u] = a.x[k * a.ld_x + c]; }
        rb[i] = make_float4(e[0], e[1], e[2], e[3]);
      } else {
        const int c = nc < N ? nc : N - 4;
        rb[i] = *reinterpret_cast<const float4 *>(a.x + k * a.ld_x + c);
      }
    }
  };
  auto stage = [&](long k0) {
#pragma unroll
    for (int i = 0; i < LDA; ++i) {
      if (arow[i] < WG_KC) {
        float4 va = ra[i];
        if (DYP) {
          const unsigned rp = (unsigned)rrp[i], am = ram[i];
          const float dx_ = (am & 0xffu) == rp && va.x * qsc[i].x + qsh[i].x > 0.f ? rdo[i].x : 0.f;
          const float dy_ = ((am >> 8) & 0xffu) == rp && va.y * qsc[i].y + qsh[i].y > 0.f ? rdo[i].y : 0.f;
          const float dz_ = ((am >> 16) & 0xffu) == rp && va.z * qsc[i].z + qsh[i].z > 0.f ? rdo[i].z : 0.f;
          const float dw_ = (am >> 24) == rp && va.w * qsc[i].w + qsh[i].w > 0.f ? rdo[i].w : 0.f;
          va.x = qka[i].x * dx_ + qkb[i].x * va.x + qkd[i].x; va.y = qka[i].y * dy_ + qkb[i].y * va.y + qkd[i].y;
          va.z = qka[i].z * dz_ + qkb[i].z * va.z + qkd[i].z; va.w = qka[i].w * dw_ + qkb[i].w * va.w + qkd[i].w;
        }
        if (DYB) {
          va.x = qka[i].x * va.x + qkb[i].x * rdo[i].x + qkd[i].x; va.y = qka[i].y * va.y + qkb[i].y * rdo[i].y + qkd[i].y;
          va.z = qka[i].z * va.z + qkb[i].z * rdo[i].z + qkd[i].z; va.w = qka[i].w * va.w + qkb[i].w * rdo[i].w + qkd[i].w;
        }
        if (k0 + arow[i] >= kend || m0 + acol[i] >= M) va = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(&As[arow[i]][acol[i]]) = va;
      }
    }
#pragma unroll
    for (int i = 0; i < LDB; ++i) {
      if (brow[i] < WG_KC) {
        const bool rowok = k0 + brow[i] < kend;
        float4 vb = rb[i];
        const int nc = n0 + bcol[i];
        if (XMODE == X_BNRELU) {
          vb.x = fmaxf(vb.x * bsc[i].x + bsh[i].x, 0.f); vb.y = fmaxf(vb.y * bsc[i].y + bsh[i].y, 0.f);
          vb.z = fmaxf(vb.z * bsc[i].z + bsh[i].z, 0.f); vb.w = fmaxf(vb.w * bsc[i].w + bsh[i].w, 0.f);
        }
        if (XMODE == X_GATHER) {
          if (nc == 0)
            vb = make_float4((gx[i][0] - gc[i][0]) * a.inv_radius, (gx[i][1] - gc[i][1]) * a.inv_radius,
                             (gx[i][2] - gc[i][2]) * a.inv_radius, 0.f);
          else if (!cvec) {
            const int C = a.c_feat;
            if (nc - 4 + 0 >= C) vb.x = 0.f;
            if (nc - 4 + 1 >= C) vb.y = 0.f;
            if (nc - 4 + 2 >= C) vb.z = 0.f;
            if (nc - 4 + 3 >= C) vb.w = 0.f;
          }
        }
        if (x_elem) {
          if (nc + 1 >= N) vb.y = 0.f;
          if (nc + 2 >= N) vb.z = 0.f;
          if (nc + 3 >= N) vb.w = 0.f;
        }
        if (!rowok || nc >= N) vb = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(&Bs[brow[i]][bcol[i]]) = vb;
      }
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool active = (m0 + 16 * wm < M) && (n0 + (TN / 2) * wh < N);
  const float *ap = &As[lane >> 4][16 * wm + (lane & 15)];
  const float *bp = &Bs[lane >> 4][(TN / 2) * wh + (lane & 15)];

  if (XMODE == X_GATHER) fetch_idx(kbeg);
  fetch(kbeg);
  if (XMODE == X_GATHER && kbeg + WG_KC < kend) fetch_idx(kbeg + WG_KC);
  for (long k0 = kbeg; k0 < kend; k0 += WG_KC) {
    stage(k0);
    __syncthreads();
    if (k0 + WG_KC < kend) {
      fetch(k0 + WG_KC);                                 // uses the indices loaded one chunk earlier
      if (XMODE == X_GATHER && k0 + 2 * WG_KC < kend) fetch_idx(k0 + 2 * WG_KC);
    }
    if (active) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {                      // 4 k steps at a time (16 rows of the chunk)
        float av[4], bv[NT][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          av[kk] = ap[(q * 4 + kk) * 4 * SA];
#pragma unroll
          for (int t = 0; t < NT; ++t) bv[t][kk] = bp[(q * 4 + kk) * 4 * SB + 16 * t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[t][kk], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  // partial tile of split s -> slab s of the workspace (M x N floats, kernel column order)
  float *o = a.ws + (long)s * M * N;
  if (active) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + (TN / 2) * wh + 16 * t + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * wm + 4 * (lane >> 4) + r;
        if (row < M && col < N) o[(long)row * N + col] = acc[t][r];
      }
    }
  }
}

// dW[m][dst(n)] = sum_s slab[s][m][n].  A block owns 32 consecutive elements, 8 lanes of splits per
// element (each sums every 8th slab, then the 8 partial sums are added in lane order: deterministic).
// gather: kernel column n of [dx dy dz 0 | feats] goes to column n (n < 3) / n - 1 (n >= 4) of the
// (M, 3 + c_feat) weight; column 3 is padding.
__global__ __launch_bounds__(256) void wgrad_x_reduce_kernel(const float *__restrict__ partial, int S, int M, int N,
                                                             int gather, float *__restrict__ dW) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long MN = (long)M * N;
  const long e = (long)blockIdx.x * 32 + el;
  float t = 0.f;
  if (e < MN) {
#pragma unroll 4
    for (int s = sl; s < S; s += 8) t += partial[(long)s * MN + e];
  }
  red[sl][el] = t;
  __syncthreads();
  if (sl != 0 || e >= MN) return;
  t = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
  if (!gather) { dW[e] = t; return; }
  const int m = (int)(e / N), n = (int)(e - (long)m * N);
  if (n == 3) return;
  dW[(long)m * (N - 1) + (n < 3 ? n : n - 1)] = t;
}


// ---------------------------------------------------------------------------------------------
// A SharedMLP layer's WHOLE backward in one pass over its dz (the SA layers whose dW is one tile: M, N in {64, 128}).
// wgrad_x_kernel and the streaming input-gradient kernel (gemm.hip) each read dz -- 0.25-0.5 GB per SA1 layer -- and, for a
// non-pooled layer, an element-wise kernel wrote it first.  Here a workgroup of 16 waves owns a range of rows and walks it in
// 64-row chunks: the chunk's dz (formed while staging: DYM as in wgrad_x_kernel) and the layer input's pre-activation z_in
// are staged ONCE;
//   waves 0-7   dW += dz^T relu(bn(z_in))     (rows are the contraction: operand fragments as in wgrad_x_kernel; the
//                                              BatchNorm+ReLU of the input is applied to the B fragments as they are read)
//   waves 8-15  g_in = (dz W) masked by the input's ReLU -> HBM, and the input BatchNorm's backward sums (sum g_in,
//               sum g_in * xhat) -- the epilogue of the streaming kernel's E_MASK.  Transposed product: A = the wave's
//               16 columns of W (from an LDS copy of the weight), B = dz rows read from LDS with ds_read_b128 (one read feeds
//               four k steps: k slot g of step j is output channel 16 kq + 4 g + j), D = four consecutive input channels of a row.
// Each SIMD gets two waves of either kind, both kinds issue the same number of MFMAs (the two products have the same FLOPs).
// LDS row strides are = 4 (mod 32): the dW fragments of a k step take rows r, r + 4, r + 8, r + 12 (conflict-free b32 reads),
// the b128 reads of the transposed product are within one extra cycle per lane group of conflict-free.
// Partial dW tiles go to the slabs of wgrad_x_kernel (wgrad_x_reduce_kernel adds them).
// Measured (SA1, 1 048 576 rows, profiles/r04_sa_layer_bwd.md): 128 <- 64 pooled 442 us where the two kernels took 277 + 254;
// 64 <- 64 239 us (1 GB: the HBM ceiling) where element-wise pass + two kernels took 155 + 168 + 141.  With every global
// access removed the 128 <- 64 launch still takes 389 us: it is bound by the fp32 MFMA pipe (34.4 GFLOP = 218 us at the
// peak) plus 73 us of staging arithmetic that does not hide under it, not by memory.
// NBUF = 1 (128 <- 128: two images and the weight do not fit): one LDS image, two barriers per chunk.
template <int TM, int TN, int DYM, int NBUF>
__global__ __launch_bounds__(1024) void sa_layer_bwd_kernel(const WgradXArgs a, int chunks_per_split, int nsplits) {
  constexpr int THREADS = 1024;
  constexpr int SA = TM + 4, SB = TN + 4;
  constexpr int LDA = 64 * (TM / 4) / THREADS, LDB = 64 * (TN / 4) / THREADS;      // float4 per thread and chunk: 1 or 2
  constexpr int MS = TM / 16, NCT = TN / 16;
  constexpr int WT = MS * NCT / 8;                 // dW tiles of a wave of the first kind: one row strip, WT column tiles
  constexpr int RG = 8 / NCT, DT = 4 / RG;         // second kind: column tile v % NCT, row tiles v / NCT + RG * i
  constexpr bool DYP = DYM == 1, DYB = DYM == 2;
  static_assert(LDA >= 1 && LDB >= 1 && WT >= 1 && RG >= 1, "tile");
  // LDS (dynamic: sa_layer_bwd_lds_bytes): As[NBUF][64][SA] | Bs[NBUF][64][SB] | ctab[4][TN] | qtab[5][TM] | red[2][8][16] | Ws
  extern __shared__ __attribute__((aligned(16))) float slb_smem[];
  float *As = slb_smem, *Bs = As + NBUF * 64 * SA;
  float *ctab = Bs + NBUF * 64 * SB;                  // scale | shift | mean | rstd of the input's BatchNorm
  float *qtab = ctab + 4 * TN;                     // sc | sh | ka | kb | kd of the layer's own BatchNorm backward
  float *red = qtab + 5 * TM;
  float *Ws = red + 256;                           // the layer's weight, [TM][TN + 4]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = xcd + 8 * j;
  if (s >= nsplits) return;
  const long K = a.R;
  const long kbeg = (long)s * chunks_per_split * 64;
  long kend = kbeg + (long)chunks_per_split * 64;
  if (kend > K) kend = K;
  if (kbeg >= kend) return;
  if (tid < TN) {
    ctab[tid] = a.in_scale[tid]; ctab[TN + tid] = a.in_shift[tid];
    ctab[2 * TN + tid] = a.dx_mean[tid]; ctab[3 * TN + tid] = a.dx_rstd[tid];
  }
  if (DYM) for (int e = tid; e < 5 * TM; e += THREADS) qtab[e] = a.dy_consts[e];
  for (int e = tid; e < TM * (TN / 4); e += THREADS) {
    const int r = e / (TN / 4), c = (e % (TN / 4)) * 4;
    *reinterpret_cast<float4 *>(&Ws[r * SB + c]) = *reinterpret_cast<const float4 *>(a.dx_w + (long)r * a.dx_ldw + c);
  }
  __syncthreads();

  // ---- staging maps (a thread's column never changes: its LDA / LDB rows are 32 apart)
  const int acol = (tid % (TM / 4)) * 4, arow0 = tid / (TM / 4);
  const int bcol = (tid % (TN / 4)) * 4, brow0 = tid / (TN / 4);
  constexpr int ARS = THREADS / (TM / 4), BRS = THREADS / (TN / 4);
  float4 ra[LDA], rb[LDB], rdo[DYM ? LDA : 1];
  unsigned ram[DYP ? LDA : 1];
  int rrp[DYP ? LDA : 1];
  // (32-bit row and element offsets against uniform base pointers: the launcher checks R * 128 < 2^31)
  const int kend_i = (int)kend, ld_dy = (int)a.ld_dy, ld_x = (int)a.ld_x;
  auto fetch = [&](long k0_) {
    const int k0 = (int)k0_;
#pragma unroll
    for (int i = 0; i < LDA; ++i) {
      int k = k0 + arow0 + ARS * i;
      if (k >= kend_i) k = kend_i - 1;                   // clamped: zeroed when staged
      if (DYP) {
        ra[i] = *reinterpret_cast<const float4 *>(a.dyz + (unsigned)(k * TM + acol));
        const unsigned grp = (unsigned)k / (unsigned)a.dy_pool;
        rrp[i] = (int)((unsigned)k - grp * (unsigned)a.dy_pool);
        ram[i] = *reinterpret_cast<const unsigned *>(a.dy_argmax + (grp * TM + acol));
        rdo[i] = *reinterpret_cast<const float4 *>(a.dy_dout + (grp * TM + acol));
      } else {
        ra[i] = *reinterpret_cast<const float4 *>(a.dy + (unsigned)(k * ld_dy + acol));
        if (DYB) rdo[i] = *reinterpret_cast<const float4 *>(a.dyz + (unsigned)(k * TM + acol));
      }
    }
#pragma unroll
    for (int i = 0; i < LDB; ++i) {
      int k = k0 + brow0 + BRS * i;
      if (k >= kend_i) k = kend_i - 1;
      rb[i] = *reinterpret_cast<const float4 *>(a.x + (unsigned)(k * ld_x + bcol));
    }
  };
  auto stage = [&](long k0, int buf) {
    float4 qsc, qsh, qka, qkb, qkd;          // (from LDS every chunk: 20 registers less across the MFMA phase)
    if (DYP) { qsc = *reinterpret_cast<const float4 *>(&qtab[acol]); qsh = *reinterpret_cast<const float4 *>(&qtab[TM + acol]); }
    if (DYM) {
      qka = *reinterpret_cast<const float4 *>(&qtab[2 * TM + acol]); qkb = *reinterpret_cast<const float4 *>(&qtab[3 * TM + acol]);
      qkd = *reinterpret_cast<const float4 *>(&qtab[4 * TM + acol]);
    }
#pragma unroll
    for (int i = 0; i < LDA; ++i) {
      float4 va = ra[i];
      if (DYP) {
        const unsigned rp = (unsigned)rrp[i], am = ram[i];
        const float dx_ = (am & 0xffu) == rp && va.x * qsc.x + qsh.x > 0.f ? rdo[i].x : 0.f;
        const float dy_ = ((am >> 8) & 0xffu) == rp && va.y * qsc.y + qsh.y > 0.f ? rdo[i].y : 0.f;
        const float dz_ = ((am >> 16) & 0xffu) == rp && va.z * qsc.z + qsh.z > 0.f ? rdo[i].z : 0.f;
        const float dw_ = (am >> 24) == rp && va.w * qsc.w + qsh.w > 0.f ? rdo[i].w : 0.f;
        va.x = qka.x * dx_ + qkb.x * va.x + qkd.x; va.y = qka.y * dy_ + qkb.y * va.y + qkd.y;
        va.z = qka.z * dz_ + qkb.z * va.z + qkd.z; va.w = qka.w * dw_ + qkb.w * va.w + qkd.w;
      }
      if (DYB) {
        va.x = qka.x * va.x + qkb.x * rdo[i].x + qkd.x; va.y = qka.y * va.y + qkb.y * rdo[i].y + qkd.y;
        va.z = qka.z * va.z + qkb.z * rdo[i].z + qkd.z; va.w = qka.w * va.w + qkb.w * rdo[i].w + qkd.w;
      }
      if (k0 + arow0 + ARS * i >= kend) va = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(&As[(buf * 64 + arow0 + ARS * i) * SA + acol]) = va;
    }
#pragma unroll
    for (int i = 0; i < LDB; ++i)                        // raw (rows past the end: a real row's values against zero dz rows)
      *reinterpret_cast<float4 *>(&Bs[(buf * 64 + brow0 + BRS * i) * SB + bcol]) = rb[i];
  };

  // ---- first kind: dW tiles.  One register file for both kinds: U = the accumulators of the first, the column sums of the second
  const int wm = w % MS, wq = (w & 7) / MS;              // (TM = 64: two waves per row strip, half of the column tiles each)
  constexpr int NU = WT > 2 ? WT : 2;
  f32x4 U[NU];
#define SLB_ACC(t) U[t]
#define SLB_T1 U[0]
#define SLB_T2 U[1]
  // ---- second kind: input-gradient tiles
  const int v = w & 7, ct = v % NCT, rg = v / NCT;
#pragma unroll
  for (int t = 0; t < NU; ++t) U[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *ap = &As[4 * g * SA + 16 * wm + li];
  const float *bp = &Bs[4 * g * SB + 16 * wq * WT + li];
  const float *dp = &As[li * SA + 4 * g];
  const float *zp = &Bs[li * SB + 16 * ct + 4 * g];
  const float *wp = &Ws[4 * g * SB + 16 * ct + li];      // k slot g of step jj of part kq: output channel 16 kq + 4 g + jj

  auto dw_tiles = [&](int buf) {
    const float *ab = ap + buf * 64 * SA, *bb = bp + buf * 64 * SB;
    float bsc[WT], bsh[WT];                              // (from LDS every chunk, like the staging constants)
#pragma unroll
    for (int t = 0; t < WT; ++t) { bsc[t] = ctab[16 * (wq * WT + t) + li]; bsh[t] = ctab[TN + 16 * (wq * WT + t) + li]; }
    constexpr int KB = WT >= 8 ? 1 : 2;                  // k steps whose operands are read together (registers: WT * KB B values)
#pragma unroll
    for (int q = 0; q < 16 / KB; ++q) {                  // k step kk takes rows 16 (r / 4) + (r % 4) + 4 g, r = KB q + kk
      float av[KB], bv[WT][KB];
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int r = 16 * ((KB * q + kk) >> 2) + ((KB * q + kk) & 3);
        av[kk] = ab[r * SA];
#pragma unroll
        for (int t = 0; t < WT; ++t) bv[t][kk] = fmaxf(bb[r * SB + 16 * t] * bsc[t] + bsh[t], 0.f);
      }
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
        for (int t = 0; t < WT; ++t)
          SLB_ACC(t) = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[t][kk], SLB_ACC(t), 0, 0, 0);
      }
    }
  };
  f32x4 og[DT];                                          // the chunk's masked input-gradient tiles, stored one phase later
  auto dx_tiles = [&](int buf) {
    const float *db = dp + buf * 64 * SA, *zb = zp + buf * 64 * SB;
    f32x4 dacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) dacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < MS; ++kq) {
      float wv[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) wv[jj] = wp[(16 * kq + jj) * SB];
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const f32x4 dz = *reinterpret_cast<const f32x4 *>(db + 16 * (rg + RG * i) * SA + 16 * kq);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          dacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[jj], dz[jj], dacc[i], 0, 0, 0);
      }
    }
    const f32x4 csc = *reinterpret_cast<const f32x4 *>(&ctab[16 * ct + 4 * g]);
    const f32x4 csh = *reinterpret_cast<const f32x4 *>(&ctab[TN + 16 * ct + 4 * g]);
    const f32x4 cmu = *reinterpret_cast<const f32x4 *>(&ctab[2 * TN + 16 * ct + 4 * g]);
    const f32x4 crs = *reinterpret_cast<const f32x4 *>(&ctab[3 * TN + 16 * ct + 4 * g]);
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const f32x4 zz = *reinterpret_cast<const f32x4 *>(zb + 16 * (rg + RG * i) * SB);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float o = zz[u] * csc[u] + csh[u] > 0.f ? dacc[i][u] : 0.f;
        og[i][u] = o;
        SLB_T1[u] += o;
        SLB_T2[u] += o * (zz[u] - cmu[u]) * crs[u];
      }
    }
  };
  auto dx_store = [&](long k0) {
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const int row = (int)k0 + 16 * (rg + RG * i) + li;
      if (row < kend_i) *reinterpret_cast<f32x4 *>(a.dx_out + (unsigned)(row * TN + 16 * ct + 4 * g)) = og[i];
    }
  };

  // Two LDS images, one barrier per chunk: chunk k + 1 is staged into the other image in the same phase in which chunk k is
  // multiplied -- the waves of the first kind stage first, those of the second kind multiply first, so that on every SIMD
  // the staging arithmetic of two waves runs under the MFMAs of the other two.  The second kind stores its tiles AFTER it
  // has staged the next chunk and BEFORE it requests the one after: loads and stores share one in-order counter (vmcnt), so
  // a wait for the next chunk's loads also waits for every store issued before them -- this way those stores are a whole
  // phase old by then (stores issued right after the MFMAs exposed their full write latency once per chunk).
  if (NBUF == 2) {
    fetch(kbeg);
    stage(kbeg, 0);
    if (kbeg + 64 < kend) fetch(kbeg + 64);
    __syncthreads();
    int buf = 0;
    for (long k0 = kbeg; k0 < kend; k0 += 64, buf ^= 1) {
      const bool more = k0 + 64 < kend;
      if (w < 8) {
        if (more) { stage(k0 + 64, buf ^ 1); if (k0 + 128 < kend) fetch(k0 + 128); }
        dw_tiles(buf);
      } else {
        dx_tiles(buf);
        if (more) {                                      // (one block: the compiler then knows that nothing but the stores is pending at the fetch)
          stage(k0 + 64, buf ^ 1);
          dx_store(k0);
          if (k0 + 128 < kend) fetch(k0 + 128);
        } else {
          dx_store(k0);
        }
      }
      __syncthreads();
    }
  } else {
    // one image: stage | barrier | request the next chunk, multiply | barrier; a chunk's tiles are stored after the NEXT chunk is staged
    fetch(kbeg);
    long k0 = kbeg;
    for (; k0 < kend; k0 += 64) {
      stage(k0, 0);
      if (w >= 8 && k0 > kbeg) dx_store(k0 - 64);
      __syncthreads();
      if (k0 + 64 < kend) fetch(k0 + 64);
      if (w < 8) dw_tiles(0); else dx_tiles(0);
      __syncthreads();
    }
    if (w >= 8) dx_store(k0 - 64);
  }
  // partial dW tile of split s -> slab s of the workspace (TM x TN floats)
  if (w < 8) {
    float *o = a.ws + (long)s * TM * TN;
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(long)(16 * wm + 4 * g + r) * TN + 16 * (wq * WT + t) + li] = SLB_ACC(t)[r];
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t1 = SLB_T1[u], t2 = SLB_T2[u];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { t1 += __shfl_xor(t1, m, 64); t2 += __shfl_xor(t2, m, 64); }
      if (li == 0) { red[v * 16 + 4 * g + u] = t1; red[128 + v * 16 + 4 * g + u] = t2; }
    }
  }
  __syncthreads();
  if (tid < TN) {
    const int c = tid >> 4, e = tid & 15;
    double c1 = 0.0, c2 = 0.0;
#pragma unroll
    for (int r = 0; r < RG; ++r) { c1 += (double)red[(c + NCT * r) * 16 + e]; c2 += (double)red[128 + (c + NCT * r) * 16 + e]; }
    const double o1 = atomicAdd(a.dx_s1 + tid, c1);
    const double o2 = atomicAdd(a.dx_s2 + tid, c2);
    asm volatile("" ::"v"(o1), "v"(o2));
  }
#undef SLB_ACC
#undef SLB_T1
#undef SLB_T2
}

// The FIRST layer of an SA stack with 128 gathered feature channels and 128 outputs (SA2: 128 <- 3 + 128), same idea: one pass
// over dz (formed while staging from the masked gradient and the pre-activation, WgradXArgs::dy_bn) gives
//   waves 0-7   dW += dz^T [dx dy dz 0 | gathered features]   (wgrad_x_kernel's X_GATHER staging: neighbour indices one chunk
//               ahead of the feature rows they address; nine 16-column tiles per wave, 132 of the 144 columns are real)
//   waves 8-15  dz W[:, 3:] -> dense rows (R x 128), stored one phase later; rows_scatter_add_kernel then adds them to
//               d(features)[point of the row] with one atomic instruction per 64 consecutive channels of a row.  (Scattering
//               straight from the MFMA tiles -- a lane owns four channels of a row, an instruction 16 rows -- was measured
//               first: 2.24 ms for SA2, the atomics want contiguous channels.)
// instead of element-wise pass + weight-gradient kernel + scatter kernel (gemm.hip E_SCATTER), each with its own pass over dz.
// One LDS image (dz 33 KB, rows 37 KB, feature columns of W 66 KB), point indices of three consecutive chunks in LDS.
__global__ __launch_bounds__(1024) void sa_gather_layer_bwd_kernel(const WgradXArgs a, int chunks_per_split, int nsplits) {
  constexpr int THREADS = 1024, TM = 128, CF = 128, TN = 144;
  constexpr int SA = TM + 4, SB = TN + 4, SW = CF + 4;
  constexpr int NCT = TN / 16;                     // 9 column tiles of dW per wave of the first kind (row strip = wave)
  extern __shared__ __attribute__((aligned(16))) float slb_smem[];
  float *As = slb_smem, *Bs = As + 64 * SA, *Ws = Bs + 64 * SB;
  float *qtab = Ws + TM * SW;                      // . | . | ka | kb | kd of the layer's BatchNorm backward
  int *pidx = reinterpret_cast<int *>(qtab + 5 * TM);      // [3][64]: b * n_pts + neighbour index of the rows of chunks k - 1, k, k + 1
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int s = xcd + 8 * j;
  if (s >= nsplits) return;
  const long kbeg_l = (long)s * chunks_per_split * 64;
  long kend_l = kbeg_l + (long)chunks_per_split * 64;
  if (kend_l > a.R) kend_l = a.R;
  if (kbeg_l >= kend_l) return;
  const int kbeg = (int)kbeg_l, kend = (int)kend_l;
  const int rps = a.m * a.ns, ldw = 3 + CF;
  for (int e = tid; e < 5 * TM; e += THREADS) qtab[e] = a.dy_consts[e];
  for (int e = tid; e < TM * CF; e += THREADS) Ws[(e >> 7) * SW + (e & 127)] = a.dx_w[(long)(e >> 7) * ldw + 3 + (e & 127)];
  for (int e = tid; e < 64 * 4; e += THREADS)      // columns 132..147 of the row image stay zero
    *reinterpret_cast<float4 *>(&Bs[(e >> 2) * SB + 132 + 4 * (e & 3)]) = make_float4(0.f, 0.f, 0.f, 0.f);
  auto point_of = [&](int k) {                     // (clamped row: staged against zero dz rows)
    if (k >= kend) k = kend - 1;
    return (k / rps) * a.n_pts + a.idx[k];
  };
  int pnext = 0;                                   // threads 0..63: point of row tid of the chunk after the one being fetched
  if (tid < 64) { pidx[tid] = point_of(kbeg + tid); pnext = point_of(kbeg + 64 + tid); }
  __syncthreads();

  const int acol = (tid & 31) * 4, row0 = tid >> 5;          // rows row0, row0 + 32 of dz and of the feature part
  float4 ra[2], rz[2], rb[2];
  float gx[3], gc[3];
  auto fetch = [&](int k0, int slot) {             // uses pidx[slot] (written before the last barrier)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int k = k0 + row0 + 32 * i;
      if (k >= kend) k = kend - 1;
      ra[i] = *reinterpret_cast<const float4 *>(a.dy + (unsigned)(k * (int)a.ld_dy + acol));
      rz[i] = *reinterpret_cast<const float4 *>(a.dyz + (unsigned)(k * TM + acol));
      rb[i] = *reinterpret_cast<const float4 *>(a.feats + (unsigned)(pidx[64 * slot + row0 + 32 * i] * CF + acol));
    }
    if (tid < 64) {
      int k = k0 + tid;
      if (k >= kend) k = kend - 1;
      const int b = k / rps, cen = b * a.m + (k - b * rps) / a.ns, pt = pidx[64 * slot + tid];
#pragma unroll
      for (int u = 0; u < 3; ++u) { gx[u] = a.xyz[(unsigned)(pt * 3 + u)]; gc[u] = a.new_xyz[(unsigned)(cen * 3 + u)]; }
    }
  };
  auto stage = [&](int k0) {
    const float4 qka = *reinterpret_cast<const float4 *>(&qtab[2 * TM + acol]);
    const float4 qkb = *reinterpret_cast<const float4 *>(&qtab[3 * TM + acol]);
    const float4 qkd = *reinterpret_cast<const float4 *>(&qtab[4 * TM + acol]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 va = ra[i];
      va.x = qka.x * va.x + qkb.x * rz[i].x + qkd.x; va.y = qka.y * va.y + qkb.y * rz[i].y + qkd.y;
      va.z = qka.z * va.z + qkb.z * rz[i].z + qkd.z; va.w = qka.w * va.w + qkb.w * rz[i].w + qkd.w;
      if (k0 + row0 + 32 * i >= kend) va = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(&As[(row0 + 32 * i) * SA + acol]) = va;
      *reinterpret_cast<float4 *>(&Bs[(row0 + 32 * i) * SB + 4 + acol]) = rb[i];
    }
    if (tid < 64)
      *reinterpret_cast<float4 *>(&Bs[tid * SB]) = make_float4((gx[0] - gc[0]) * a.inv_radius, (gx[1] - gc[1]) * a.inv_radius,
                                                               (gx[2] - gc[2]) * a.inv_radius, 0.f);
  };

  f32x4 U[NCT];                                    // first kind: 9 accumulators; second kind: U[0..3] = the tiles held for the scatter
#pragma unroll
  for (int t = 0; t < NCT; ++t) U[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *ap = &As[4 * g * SA + 16 * (w & 7) + li];
  const float *bp = &Bs[4 * g * SB + li];
  const int ct = w & 7;
  const float *dp = &As[li * SA + 4 * g];
  const float *wp = &Ws[4 * g * SW + 16 * ct + li];

  auto dw_tiles = [&]() {
#pragma unroll
    for (int q = 0; q < 16; ++q) {                       // k step q takes rows 16 (q / 4) + (q % 4) + 4 g
      const int r = 16 * (q >> 2) + (q & 3);
      const float av = ap[r * SA];
      float bv[NCT];
#pragma unroll
      for (int t = 0; t < NCT; ++t) bv[t] = bp[r * SB + 16 * t];
#pragma unroll
      for (int t = 0; t < NCT; ++t) U[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], U[t], 0, 0, 0);
    }
  };
  auto dx_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) U[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < TM / 16; ++kq) {
      float wv[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) wv[jj] = wp[(16 * kq + jj) * SW];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 dz = *reinterpret_cast<const f32x4 *>(dp + 16 * i * SA + 16 * kq);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) U[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[jj], dz[jj], U[i], 0, 0, 0);
      }
    }
  };
  auto scatter = [&](int k0, int) {                // chunk k0's tiles: row 16 i + li, channels 16 ct + 4 g .. (dense rows; see the header)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = k0 + 16 * i + li;
      if (row < kend) *reinterpret_cast<f32x4 *>(a.dx_out + (unsigned)(row * CF + 16 * ct + 4 * g)) = U[i];
    }
  };

  // stage | [scatter the previous chunk] | barrier | request the next chunk, multiply | barrier
  fetch(kbeg, 0);
  int slot = 0;                                    // pidx slot of the chunk being staged
  int k0 = kbeg;
  for (; k0 < kend; k0 += 64) {
    const int nslot = slot == 2 ? 0 : slot + 1, pslot = slot == 0 ? 2 : slot - 1;
    stage(k0);
    if (w >= 8 && k0 > kbeg) scatter(k0 - 64, pslot);
    if (tid < 64) pidx[64 * nslot + tid] = pnext;
    __syncthreads();
    if (k0 + 64 < kend) {
      if (tid < 64) pnext = point_of(k0 + 128 + tid);
      fetch(k0 + 64, nslot);
    }
    if (w < 8) dw_tiles(); else dx_tiles();
    __syncthreads();
    slot = nslot;
  }
  if (w >= 8) {
    scatter(k0 - 64, slot == 0 ? 2 : slot - 1);
  } else {
    // partial dW tile of split s -> slab s (128 x 132 floats, kernel column order [dx dy dz 0 | features])
    float *o = a.ws + (long)s * TM * 132;
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
      const int col = 16 * t + li;
      if (col < 132) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(16 * w + 4 * g + r) * 132 + col] = U[t][r];
      }
    }
  }
}
// d(features)[(row / rows_per_scene) * n_pts + idx[row]][:] += g[row][:] for dense rows of C = 64 q channels: a wave = 64
// consecutive channels of one row per atomic instruction.
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const float *__restrict__ g, const int *__restrict__ idx, long R,
                                                               int rps, int n_pts, int C, float *__restrict__ dfeats) {
  const int lane = threadIdx.x & 63;
  const int parts = C >> 6;                        // 64-channel parts of a row
  const long nitems = R * parts;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long it = wave; it < nitems; it += nwaves) {
    const long row = it / parts;
    const int part = (int)(it - row * parts);
    const long p = (row / rps) * n_pts + idx[row];
    atomicAdd(dfeats + p * C + 64 * part + lane, g[row * C + 64 * part + lane]);
  }
}
constexpr size_t sa_gather_layer_bwd_lds_bytes() { return sizeof(float) * (64 * 132 + 64 * 148 + 128 * 132 + 5 * 128 + 3 * 64); }

template <int TM, int TN, int NBUF>
constexpr size_t sa_layer_bwd_lds_bytes() { return sizeof(float) * (NBUF * 64 * (TM + 4 + TN + 4) + 4 * TN + 5 * TM + 256 + TM * (TN + 4)); }

namespace {
struct WgxPlan { int TM, TN, tiles_m, tiles_n, splits, cps; };
WgxPlan wgx_plan(long R, int M, int N, bool gather) {
  WgxPlan p;
  // tile: the candidate with the least padded area (ties: the larger tile)
  const int tms[3] = {128, 96, 64};
  const int tns_plain[2] = {128, 64}, tns_gather[2] = {160, 64};
  const int *tns = gather ? tns_gather : tns_plain;
  long best = -1;
  p.TM = 128; p.TN = tns[0];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 2; ++j) {
      const long area = (long)((M + tms[i] - 1) / tms[i]) * tms[i] * ((N + tns[j] - 1) / tns[j]) * tns[j];
      if (best < 0 || area < best) { best = area; p.TM = tms[i]; p.TN = tns[j]; }
    }
  p.tiles_m = (M + p.TM - 1) / p.TM;
  p.tiles_n = (N + p.TN - 1) / p.TN;
  const long nchunks = (R + WG_KC - 1) / WG_KC;
  // ~512 workgroups, at least 4 chunks per split
  long want = (512 + p.tiles_m * p.tiles_n - 1) / (p.tiles_m * p.tiles_n);
  if (want > nchunks / 4) want = nchunks / 4;
  if (want < 1) want = 1;
  if (want >= 8) want = want / 8 * 8;
  p.cps = (int)((nchunks + want - 1) / want);
  p.splits = (int)((nchunks + p.cps - 1) / p.cps);
  return p;
}

template <int TM, int TN>
void wgx_launch(const WgradXArgs &a, const WgxPlan &p, hipStream_t stream) {
  const int ntiles = p.tiles_m * p.tiles_n;
  const dim3 grid((unsigned)(8 * ntiles * ((p.splits + 7) / 8))), block(TM * 8);
  if (a.xmode == X_PLAIN && a.dy_bn)
    hipLaunchKernelGGL((wgrad_x_kernel<TM, TN, X_PLAIN, 2>), grid, block, 0, stream, a, p.cps, p.tiles_n, ntiles, p.splits);
  else if (a.xmode == X_PLAIN)
    hipLaunchKernelGGL((wgrad_x_kernel<TM, TN, X_PLAIN>), grid, block, 0, stream, a, p.cps, p.tiles_n, ntiles, p.splits);
  else if (a.xmode == X_BNRELU && a.dy_pool > 0)
    hipLaunchKernelGGL((wgrad_x_kernel<TM, TN, X_BNRELU, 1>), grid, block, 0, stream, a, p.cps, p.tiles_n, ntiles, p.splits);
  else if (a.xmode == X_BNRELU && a.dy_bn)
    hipLaunchKernelGGL((wgrad_x_kernel<TM, TN, X_BNRELU, 2>), grid, block, 0, stream, a, p.cps, p.tiles_n, ntiles, p.splits);
  else if (a.xmode == X_BNRELU)
    hipLaunchKernelGGL((wgrad_x_kernel<TM, TN, X_BNRELU>), grid, block, 0, stream, a, p.cps, p.tiles_n, ntiles, p.splits);
  else if (a.dy_bn)
    hipLaunchKernelGGL((wgrad_x_kernel<TM, TN, X_GATHER, 2>), grid, block, 0, stream, a, p.cps, p.tiles_n, ntiles, p.splits);
  else
    hipLaunchKernelGGL((wgrad_x_kernel<TM, TN, X_GATHER>), grid, block, 0, stream, a, p.cps, p.tiles_n, ntiles, p.splits);
}
}  // namespace

bool eda_wgrad_x_fuses_gather(int M, int c_feat) { return M == 128 && c_feat == 128; }
bool eda_wgrad_x_fuses_dx(int M, int N) { return ((M == 128 || M == 64) && N == 64) || (M == 128 && N == 128); }

size_t eda_wgrad_x_workspace_bytes(long R, int M, int N) {
  if (R <= 0 || M <= 0 || N <= 0) return 0;
  // (the split count depends on the tile; take the larger of the plain / gather plans)
  const WgxPlan p = wgx_plan(R, M, N, false), q = wgx_plan(R, M, N, true);
  const int sp = p.splits > q.splits ? p.splits : q.splits;
  return sizeof(float) * (size_t)sp * (size_t)M * N;
}

int eda_wgrad_x_launch(const WgradXArgs &a, hipStream_t stream) {
  if (a.R <= 0) {
    const int ncols = a.xmode == X_GATHER ? a.N - 1 : a.N;
    return eda_zero_async(a.dW, sizeof(float) * (size_t)a.M * ncols, stream);
  }
  if (a.dy_pool > 0) {
    if (a.xmode != X_BNRELU || a.M % 4 != 0 || !a.dyz || !a.dy_argmax || !a.dy_dout || !a.dy_consts ||
        ((reinterpret_cast<uintptr_t>(a.dyz) | reinterpret_cast<uintptr_t>(a.dy_dout) | reinterpret_cast<uintptr_t>(a.dy_consts)) & 15u) ||
        (reinterpret_cast<uintptr_t>(a.dy_argmax) & 3u)) {
      eda_set_error("wgrad_x: bad operands for the pooled BatchNorm-backward dY prologue");
      return EDA_ERR_INVALID_ARG;
    }
  } else if (a.M % 4 != 0 || a.ld_dy % 4 != 0 || (reinterpret_cast<uintptr_t>(a.dy) & 15u) != 0 || a.M < 4) {
    eda_set_error("wgrad_x: dY rows must be 16-byte addressable");
    return EDA_ERR_INVALID_ARG;
  }
  if (a.xmode == X_BNRELU && (a.N % 4 != 0 || a.N < 4 || a.ld_x % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15u) != 0)) {
    eda_set_error("wgrad_x: BN+ReLU rows must be 16-byte addressable");
    return EDA_ERR_INVALID_ARG;
  }
  if (a.dy_bn && (a.dy_pool > 0 || !a.dyz || !a.dy_consts ||
                  ((reinterpret_cast<uintptr_t>(a.dyz) | reinterpret_cast<uintptr_t>(a.dy_consts)) & 15u))) {
    eda_set_error("wgrad_x: bad operands for the BatchNorm-backward dY prologue");
    return EDA_ERR_INVALID_ARG;
  }
  const WgxPlan p = wgx_plan(a.R, a.M, a.N, a.xmode == X_GATHER);
  if (!a.ws || a.ws_bytes < sizeof(float) * (size_t)p.splits * a.M * a.N) {
    eda_set_error("wgrad_x: workspace too small");
    return EDA_ERR_WORKSPACE;
  }
  if (a.dx_out && a.xmode == X_GATHER) {
    // first layer of an SA stack: weight gradient and the scatter of the input gradient in one launch
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (!eda_wgrad_x_fuses_gather(a.M, a.c_feat) || !a.dy_bn || a.R * 128 >= 0x7fffffffL || a.ld_dy != a.M || !a.dx_w || !a.dx_scatter || !al16(a.dx_out) ||
        !al16(a.feats) || !a.idx || !a.xyz || !a.new_xyz || (long)a.n_pts * (a.R / ((long)a.m * a.ns) + 1) * 128 >= 0x7fffffffL ||
        p.tiles_m * p.tiles_n != 1) {
      eda_set_error("wgrad_x: bad operands for the fused gather layer");
      return EDA_ERR_INVALID_ARG;
    }
    const dim3 grid((unsigned)(8 * ((p.splits + 7) / 8))), block(1024);
    const size_t lds = sa_gather_layer_bwd_lds_bytes();
    hipError_t e = eda_set_max_dynamic_lds(reinterpret_cast<const void *>(sa_gather_layer_bwd_kernel), lds);
    if (e != hipSuccess) { eda_set_error("wgrad_x: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(sa_gather_layer_bwd_kernel, grid, block, lds, stream, a, p.cps, p.splits);
    EDA_CHECK_LAUNCH();
    if (eda_deterministic()) {                    // ordered per-point sums instead of fp32 atomics (scatter_det.hip)
      const long rows = (long)a.m * a.ns;
      EdaDetScatter d = {a.idx, nullptr, rows, (int)rows, 1, a.dx_out, rows * a.c_feat, (long)a.c_feat, 1,
                         a.dx_scatter, (long)a.n_pts * a.c_feat, (long)a.c_feat, 1, (int)(a.R / rows), a.n_pts, a.c_feat};
      const int rc = eda_det_scatter_launch(d, stream);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(2048), dim3(256), 0, stream, a.dx_out, a.idx, a.R, a.m * a.ns, a.n_pts,
                         a.c_feat, a.dx_scatter);
      EDA_CHECK_LAUNCH();
    }
    const long MN = (long)a.M * a.N;
    hipLaunchKernelGGL(wgrad_x_reduce_kernel, dim3((unsigned)((MN + 31) / 32)), dim3(256), 0, stream, a.ws, p.splits,
                       a.M, a.N, 1, a.dW);
    EDA_CHECK_LAUNCH();
    return 0;
  }
  if (a.dx_out) {
    // the layer's input gradient in the same launch (sa_layer_bwd_kernel)
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (a.xmode != X_BNRELU || !eda_wgrad_x_fuses_dx(a.M, a.N) || a.ld_x != a.N || a.R * 128 >= 0x7fffffffL || a.ld_dy > 128 || !a.dx_w || !a.dx_mean || !a.dx_rstd ||
        !a.dx_s1 || !a.dx_s2 || !al16(a.dx_out) || !al16(a.in_scale) || !al16(a.in_shift) || p.tiles_m * p.tiles_n != 1) {
      eda_set_error("wgrad_x: bad operands for the fused input gradient");
      return EDA_ERR_INVALID_ARG;
    }
    const dim3 grid((unsigned)(8 * ((p.splits + 7) / 8))), block(1024);
    const int dym = a.dy_pool > 0 ? 1 : a.dy_bn ? 2 : 0;
#define EDA_SLB(TM_, TN_, NB_)                                                                                              \
    do {                                                                                                                    \
      const size_t lds = sa_layer_bwd_lds_bytes<TM_, TN_, NB_>();                                                           \
      void (*kern)(const WgradXArgs, int, int) = dym == 1 ? sa_layer_bwd_kernel<TM_, TN_, 1, NB_>                            \
                                               : dym == 2 ? sa_layer_bwd_kernel<TM_, TN_, 2, NB_> : sa_layer_bwd_kernel<TM_, TN_, 0, NB_>; \
      hipError_t e = eda_set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds); \
      if (e != hipSuccess) { eda_set_error("wgrad_x: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }           \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a, p.cps, p.splits);                                               \
    } while (0)
    if (a.N == 128) EDA_SLB(128, 128, 1); else if (a.M == 128) EDA_SLB(128, 64, 2); else EDA_SLB(64, 64, 2);
#undef EDA_SLB
    EDA_CHECK_LAUNCH();
    const long MN = (long)a.M * a.N;
    hipLaunchKernelGGL(wgrad_x_reduce_kernel, dim3((unsigned)((MN + 31) / 32)), dim3(256), 0, stream, a.ws, p.splits,
                       a.M, a.N, 0, a.dW);
    EDA_CHECK_LAUNCH();
    return 0;
  }
  if (p.TN == 64) {
    if (p.TM == 128) wgx_launch<128, 64>(a, p, stream); else if (p.TM == 96) wgx_launch<96, 64>(a, p, stream); else wgx_launch<64, 64>(a, p, stream);
  } else if (p.TN == 128) {
    if (p.TM == 128) wgx_launch<128, 128>(a, p, stream); else if (p.TM == 96) wgx_launch<96, 128>(a, p, stream); else wgx_launch<64, 128>(a, p, stream);
  } else {
    if (p.TM == 128) wgx_launch<128, 160>(a, p, stream); else if (p.TM == 96) wgx_launch<96, 160>(a, p, stream); else wgx_launch<64, 160>(a, p, stream);
  }
  EDA_CHECK_LAUNCH();
  const long MN = (long)a.M * a.N;
  hipLaunchKernelGGL(wgrad_x_reduce_kernel, dim3((unsigned)((MN + 31) / 32)), dim3(256), 0, stream, a.ws, p.splits,
                     a.M, a.N, a.xmode == X_GATHER ? 1 : 0, a.dW);
  EDA_CHECK_LAUNCH();
  return 0;
}
