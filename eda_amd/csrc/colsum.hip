// colsum.hip -- column sums of a row-major (R, C) fp32 matrix in ONE launch.
//
// The bias gradient of every pointwise linear layer on the path (attention in-projections,
// FFN, 1x1-conv heads: d(bias) = sum over rows of dY) -- ~150 of them per training step, each
// a separate 11-13 us two-pass reduction when left to the framework.  HBM-bound: R*C*4 bytes
// read once.
//
// Grid = (column groups of 64) x (row slabs).  A block sums its slab for 64 columns with 16
// rows in flight per iteration (float4 loads when the rows are 16-byte aligned), writes the 64
// partial sums to the workspace, and takes a ticket on its column group's counter; the block
// that draws the last ticket adds up the slabs' partial rows in slab order (deterministic:
// the same association for every run) and resets the counter to zero, so the caller-owned
// counter array only has to be zero once, at allocation.  Kernels sharing one counter array must
// be ordered (same stream / same graph branch).
#include "eda_common.h"
#include <string.h>

#define CS_THREADS 256
#define CS_MAX_SLABS 32      // the last block walks this many partial rows: keep it short

template <bool VEC>
__global__ __launch_bounds__(CS_THREADS) void colsum_kernel(const float *__restrict__ x, long R, int C,
                                                            long ld, int rows_per_slab,
                                                            float *__restrict__ out,
                                                            float *__restrict__ partial,
                                                            unsigned *__restrict__ counters) {
  __shared__ float red[16][65];
  __shared__ int is_last;
  const int tid = threadIdx.x;
  const int cg = blockIdx.x, slab = blockIdx.y, nslab = gridDim.y;
  const int c0 = cg * 64;
  const long r0 = (long)slab * rows_per_slab;
  long r1 = r0 + rows_per_slab;
  if (r1 > R) r1 = R;
  if (VEC) {
    // 16 column lanes x float4 = 64 columns, 16 row lanes
    const int cl = tid & 15, rl = tid >> 4;
    const int c = c0 + 4 * cl;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
#pragma unroll 4
      for (long r = r0 + rl; r < r1; r += 16) {
        const float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    red[rl][4 * cl + 0] = acc.x; red[rl][4 * cl + 1] = acc.y;
    red[rl][4 * cl + 2] = acc.z; red[rl][4 * cl + 3] = acc.w;
  } else {
    // 64 column lanes, 4 row lanes, 4 independent accumulators each
    const int cl = tid & 63, rl = tid >> 6;
    const int c = c0 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
      long r = r0 + rl;
      for (; r + 12 < r1; r += 16) {
        a0 += x[r * ld + c]; a1 += x[(r + 4) * ld + c];
        a2 += x[(r + 8) * ld + c]; a3 += x[(r + 12) * ld + c];
      }
      for (; r < r1; r += 4) a0 += x[r * ld + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[4 * rl + k][cl] = 0.f;
    red[4 * rl][cl] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (tid < 64 && c0 + tid < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][tid];
    if (nslab == 1) {
      out[c0 + tid] = t;
    } else {
      // published with a RETURNING agent-scope exchange: the wave waits for the old value, i.e. for the
      // write to have been performed at the memory side, before the block may take its ticket (a plain
      // write-through store is only acknowledged by the local L2; a ticket overtaking fire-and-forget
      // atomics was observed in the BN statistics kernel, see sa_cl.hip).  No L2 write-back fence needed.
      const float old = __hip_atomic_exchange(&partial[(long)slab * C + c0 + tid], t, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old));
    }
  }
  if (nslab == 1) return;
  // the exchanges have returned, i.e. the partial row is in memory; then take a ticket.  (A
  // release FENCE at agent scope here costs a whole-L2 write-back per block: measured 3-4x
  // slower than the framework's two-pass reduction.)
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(&counters[cg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = ticket == (unsigned)(nslab - 1);
  }
  __syncthreads();
  if (!is_last) return;
  // 64 columns x 4 slab lanes, slab order fixed per lane
  {
    const int cl = tid & 63, sl = tid >> 6;
    float t = 0.f;
    if (c0 + cl < C) {
      float v[CS_MAX_SLABS / 4];
#pragma unroll
      for (int i = 0; i < CS_MAX_SLABS / 4; ++i) {      // all loads in flight, then a fixed-order sum
        const int s = sl + 4 * i;
        v[i] = s < nslab ? __hip_atomic_load(&partial[(long)s * C + c0 + cl], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT)
                         : 0.f;
      }
#pragma unroll
      for (int i = 0; i < CS_MAX_SLABS / 4; ++i) t += v[i];
    }
    __syncthreads();
    red[sl][cl] = t;
    __syncthreads();
    if (tid < 64 && c0 + tid < C) out[c0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid == 0) __hip_atomic_store(&counters[cg], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}


// Weighted column sums: out[m][c] = sum_r w[r][m] * x[r][c] for MW <= 4 weight columns, and
// wsum[m] = sum_r w[r][m].  This is the weight (and bias) gradient of a pointwise layer with 1-4
// OUTPUT channels (box centre / size residuals, objectness: dW = dY^T X with dY only MW wide),
// which the library GEMM runs as a 41 us 32x16 macro-tile kernel; here x is streamed once,
// like a column sum.  Same grid, ticket and slab-order reduction as colsum_kernel.
template <int MW>
__global__ __launch_bounds__(CS_THREADS) void wcolsum_kernel(const float *__restrict__ x, long R, int C,
                                                             long ld, const float *__restrict__ w, long ldw,
                                                             int rows_per_slab, float *__restrict__ out,
                                                             float *__restrict__ wsum,
                                                             float *__restrict__ partial,
                                                             unsigned *__restrict__ counters) {
  __shared__ float red[MW][16][65];
  __shared__ float wred[MW][16];
  __shared__ int is_last;
  const int tid = threadIdx.x;
  const int cg = blockIdx.x, slab = blockIdx.y, nslab = gridDim.y;
  const int c0 = cg * 64;
  const long r0 = (long)slab * rows_per_slab;
  long r1 = r0 + rows_per_slab;
  if (r1 > R) r1 = R;
  const int cl = tid & 15, rl = tid >> 4;
  const int c = c0 + 4 * cl;
  float4 acc[MW];
  float ws[MW];
#pragma unroll
  for (int m = 0; m < MW; ++m) { acc[m] = make_float4(0.f, 0.f, 0.f, 0.f); ws[m] = 0.f; }
  if (c < C) {
#pragma unroll 2
    for (long r = r0 + rl; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c);
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const float wv = w[r * ldw + m];
        acc[m].x += wv * v.x; acc[m].y += wv * v.y; acc[m].z += wv * v.z; acc[m].w += wv * v.w;
        ws[m] += wv;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    red[m][rl][4 * cl + 0] = acc[m].x; red[m][rl][4 * cl + 1] = acc[m].y;
    red[m][rl][4 * cl + 2] = acc[m].z; red[m][rl][4 * cl + 3] = acc[m].w;
    if (cl == 0) wred[m][rl] = ws[m];
  }
  __syncthreads();
  // slab layout in `partial`: [slab][MW][C] then, after all slabs, [slab][MW] weight sums
  float *psum = partial + (long)nslab * MW * C;
  if (tid < 64 && c0 + tid < C) {
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[m][k][tid];
      if (nslab == 1) {
        out[(long)m * C + c0 + tid] = t;
      } else {       // returning exchange: see colsum_kernel
        const float old = __hip_atomic_exchange(&partial[((long)slab * MW + m) * C + c0 + tid], t, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old));
      }
    }
  }
  if (cg == 0 && tid < MW) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += wred[tid][k];
    if (nslab == 1) {
      if (wsum) wsum[tid] = t;
    } else {
      const float old = __hip_atomic_exchange(&psum[(long)slab * MW + tid], t, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old));
    }
  }
  if (nslab == 1) return;
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(&counters[cg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = ticket == (unsigned)(nslab - 1);
  }
  __syncthreads();
  if (!is_last) return;
  {
    const int col = tid & 63, sl = tid >> 6;
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      float t = 0.f;
      if (c0 + col < C) {
        float v[CS_MAX_SLABS / 4];
#pragma unroll
        for (int i = 0; i < CS_MAX_SLABS / 4; ++i) {
          const int s = sl + 4 * i;
          v[i] = s < nslab ? __hip_atomic_load(&partial[((long)s * MW + m) * C + c0 + col], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)
                           : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CS_MAX_SLABS / 4; ++i) t += v[i];
      }
      __syncthreads();
      red[0][sl][col] = t;
      __syncthreads();
      if (tid < 64 && c0 + tid < C)
        out[(long)m * C + c0 + tid] = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
    }
    // every column group waits for ALL slabs, so the last block of group 0 can also finish wsum
    if (cg == 0 && wsum && tid < MW) {
      float t = 0.f;
      for (int s = 0; s < nslab; ++s)
        t += __hip_atomic_load(&psum[(long)s * MW + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wsum[tid] = t;
    }
    if (tid == 0) __hip_atomic_store(&counters[cg], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static int cs_slabs(long R, int C) {
  const int cgs = (C + 63) / 64;
  long slabs = (R + 63) / 64;                 // >= 64 rows per slab
  const long want = (1024 + cgs - 1) / cgs;   // up to ~1024 blocks over the chip
  if (slabs > want) slabs = want;
  if (slabs > CS_MAX_SLABS) slabs = CS_MAX_SLABS;
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

extern "C" size_t eda_colsum_workspace_bytes(long R, int C) {
  if (R <= 0 || C <= 0) return 0;
  return sizeof(float) * (size_t)cs_slabs(R, C) * (size_t)C;
}

extern "C" int eda_colsum_f32(const float *x, long R, int C, long ld, float *out, void *ws,
                              size_t ws_bytes, unsigned *counters, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && ld >= C, "bad dimension");
  EDA_CHECK_ARG(out, "null pointer");
  if (R == 0) return eda_zero_async(out, sizeof(float) * C, stream);
  EDA_CHECK_ARG(x && counters, "null pointer");
  const int slabs = cs_slabs(R, C);
  if (slabs > 1 && (!ws || ws_bytes < eda_colsum_workspace_bytes(R, C))) {
    eda_set_error("colsum: workspace too small");
    return EDA_ERR_WORKSPACE;
  }
  const int rows_per_slab = (int)((R + slabs - 1) / slabs);
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)slabs);
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && ((uintptr_t)x % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(CS_THREADS), 0, stream, x, R, C, ld, rows_per_slab, out,
                       reinterpret_cast<float *>(ws), counters);
  else
    hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(CS_THREADS), 0, stream, x, R, C, ld, rows_per_slab,
                       out, reinterpret_cast<float *>(ws), counters);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t eda_wcolsum_workspace_bytes(long R, int C, int MW) {
  if (R <= 0 || C <= 0 || MW <= 0) return 0;
  return sizeof(float) * (size_t)cs_slabs(R, C) * (size_t)MW * ((size_t)C + 1);
}

extern "C" int eda_wcolsum_f32(const float *x, long R, int C, long ld, const float *w, long ldw, int MW,
                               float *out, float *wsum, void *ws, size_t ws_bytes, unsigned *counters,
                               void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && ld >= C && MW >= 1 && MW <= 4 && ldw >= MW, "bad dimension (1 <= MW <= 4)");
  EDA_CHECK_ARG(out, "null pointer");
  if (R == 0) {
    const int z1 = eda_zero_async(out, sizeof(float) * (size_t)MW * C, stream);
    if (z1 || !wsum) return z1;
    return eda_zero_async(wsum, sizeof(float) * MW, stream);
  }
  EDA_CHECK_ARG(x && w && counters, "null pointer");
  EDA_CHECK_ARG(C % 4 == 0 && ld % 4 == 0 && (uintptr_t)x % 16 == 0, "x rows must be 16-byte aligned");
  const int slabs = cs_slabs(R, C);
  if (slabs > 1 && (!ws || ws_bytes < eda_wcolsum_workspace_bytes(R, C, MW))) {
    eda_set_error("wcolsum: workspace too small");
    return EDA_ERR_WORKSPACE;
  }
  const int rows_per_slab = (int)((R + slabs - 1) / slabs);
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)slabs);
  float *partial = reinterpret_cast<float *>(ws);
#define WCS_LAUNCH(MWC)                                                                                   \
  hipLaunchKernelGGL(wcolsum_kernel<MWC>, grid, dim3(CS_THREADS), 0, stream, x, R, C, ld, w, ldw, rows_per_slab, \
                     out, wsum, partial, counters)
  switch (MW) {
    case 1: WCS_LAUNCH(1); break;
    case 2: WCS_LAUNCH(2); break;
    case 3: WCS_LAUNCH(3); break;
    default: WCS_LAUNCH(4); break;
  }
#undef WCS_LAUNCH
  EDA_CHECK_LAUNCH();
  return 0;
}


// ---- the whole backward of 1-4-output-channel layers, several layers in one launch ------------------------------
// The last layers of the prediction heads' box stacks (centre / size: 288 -> 3, models/modules.py:66-86, 150-175) have
// outputs too narrow for the GEMM kernels: per layer the input gradient da = dY W ran as an element-wise product launch
// and dW = dY^T a, db = colsum(dY) as a wcolsum launch -- two launches per layer, two such layers per head, seven
// heads whose backward passes are issued together (eda_amd/heads_batched.py).  Here ONE streaming pass per layer reads
// a and dY, writes da and accumulates dW / db, for up to TO_MAXG layers of one shape in one launch (blockIdx.z = layer,
// blockIdx.y = row slab); a second small launch adds the slabs in order (deterministic, no atomics).
constexpr int TO_MAXG = 16;
constexpr int TO_SLABS = 4;
struct TinyOutArgs {
  const float *dy[TO_MAXG], *w[TO_MAXG], *a[TO_MAXG];
  float *da[TO_MAXG], *dW[TO_MAXG], *db[TO_MAXG];
  long lda[TO_MAXG], ldda[TO_MAXG];
};
template <int MW>
__global__ __launch_bounds__(256) void tiny_out_bwd_kernel(const TinyOutArgs A, int R, int C, float *__restrict__ partial) {
  __shared__ float red[MW][16][65];
  __shared__ float wred[MW][16];
  const int tid = threadIdx.x, g = blockIdx.z, slab = blockIdx.y, cg = blockIdx.x;
  const int cl = tid & 15, rl = tid >> 4;
  const int c = cg * 64 + 4 * cl;
  const int rps = (R + TO_SLABS - 1) / TO_SLABS;
  const int r0 = slab * rps, r1 = min(R, r0 + rps);
  const float *dy = A.dy[g], *a = A.a[g];
  float *da = A.da[g];
  const long lda = A.lda[g], ldda = A.ldda[g];
  float4 wv[MW], acc[MW];
  float ws[MW];
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    wv[m] = c < C ? *reinterpret_cast<const float4 *>(A.w[g] + (long)m * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    ws[m] = 0.f;
  }
  if (c < C) {
#pragma unroll 2
    for (int r = r0 + rl; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(a + (long)r * lda + c);
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const float d = dy[(long)r * MW + m];
        o.x = fmaf(d, wv[m].x, o.x); o.y = fmaf(d, wv[m].y, o.y); o.z = fmaf(d, wv[m].z, o.z); o.w = fmaf(d, wv[m].w, o.w);
        acc[m].x = fmaf(d, v.x, acc[m].x); acc[m].y = fmaf(d, v.y, acc[m].y);
        acc[m].z = fmaf(d, v.z, acc[m].z); acc[m].w = fmaf(d, v.w, acc[m].w);
        ws[m] += d;
      }
      *reinterpret_cast<float4 *>(da + (long)r * ldda + c) = o;
    }
  }
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    red[m][rl][4 * cl + 0] = acc[m].x; red[m][rl][4 * cl + 1] = acc[m].y;
    red[m][rl][4 * cl + 2] = acc[m].z; red[m][rl][4 * cl + 3] = acc[m].w;
    if (cl == 0) wred[m][rl] = ws[m];
  }
  __syncthreads();
  // partial: [layer][slab][MW][C + 1] (the last entry of a row: the slab's sum of dY column m)
  float *p = partial + ((long)g * TO_SLABS + slab) * MW * (C + 1);
  if (tid < 64 && cg * 64 + tid < C) {
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[m][k][tid];
      p[(long)m * (C + 1) + cg * 64 + tid] = t;
    }
  }
  if (cg == 0 && tid < MW) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += wred[tid][k];
    p[(long)tid * (C + 1) + C] = t;
  }
}
template <int MW>
__global__ __launch_bounds__(256) void tiny_out_reduce_kernel(const TinyOutArgs A, int C, const float *__restrict__ partial) {
  const int g = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;            // element of the (MW, C + 1) table
  if (e >= MW * (C + 1)) return;
  float t = 0.f;
#pragma unroll
  for (int s = 0; s < TO_SLABS; ++s) t += partial[((long)g * TO_SLABS + s) * MW * (C + 1) + e];
  const int m = e / (C + 1), c = e - m * (C + 1);
  if (c < C) A.dW[g][(long)m * C + c] = t;
  else if (A.db[g]) A.db[g][m] = t;
}

extern "C" size_t eda_tiny_out_bwd_workspace_bytes(int ngroups, int C, int MW) {
  if (ngroups <= 0 || C <= 0 || MW <= 0) return 0;
  return sizeof(float) * (size_t)ngroups * TO_SLABS * (size_t)MW * ((size_t)C + 1);
}

extern "C" int eda_tiny_out_bwd_multi_f32(int ngroups, long R, int C, int MW, const float *const *dy, const float *const *w,
                                          const float *const *a, const long *lda, float *const *da, const long *ldda,
                                          float *const *dW, float *const *db, void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(ngroups >= 1 && ngroups <= TO_MAXG && R > 0 && R < 0x7fffffffL && C > 0 && C % 4 == 0 && MW >= 1 && MW <= 4,
                "1..16 layers, 1..4 output channels, C a multiple of 4");
  EDA_CHECK_ARG(dy && w && a && lda && da && ldda && dW && db, "null pointer");
  if (!ws || ws_bytes < eda_tiny_out_bwd_workspace_bytes(ngroups, C, MW)) {
    eda_set_error("tiny_out_bwd: workspace too small");
    return EDA_ERR_WORKSPACE;
  }
  TinyOutArgs A;
  memset(&A, 0, sizeof(A));
  for (int g = 0; g < ngroups; ++g) {
    EDA_CHECK_ARG(dy[g] && w[g] && a[g] && da[g] && dW[g] && lda[g] >= C && ldda[g] >= C && lda[g] % 4 == 0 && ldda[g] % 4 == 0 &&
                      ((uintptr_t)a[g] % 16 == 0) && ((uintptr_t)da[g] % 16 == 0) && ((uintptr_t)w[g] % 16 == 0),
                  "bad group (16-byte addressable rows)");
    A.dy[g] = dy[g]; A.w[g] = w[g]; A.a[g] = a[g]; A.da[g] = da[g]; A.dW[g] = dW[g]; A.db[g] = db[g];
    A.lda[g] = lda[g]; A.ldda[g] = ldda[g];
  }
  float *partial = reinterpret_cast<float *>(ws);
  const dim3 grid((unsigned)((C + 63) / 64), TO_SLABS, (unsigned)ngroups);
  const dim3 rgrid((unsigned)((MW * (C + 1) + 255) / 256), (unsigned)ngroups);
#define TO_LAUNCH(MWC)                                                                                     \
  hipLaunchKernelGGL(tiny_out_bwd_kernel<MWC>, grid, dim3(256), 0, stream, A, (int)R, C, partial);        \
  hipLaunchKernelGGL(tiny_out_reduce_kernel<MWC>, rgrid, dim3(256), 0, stream, A, C, (const float *)partial)
  switch (MW) {
    case 1: TO_LAUNCH(1); break;
    case 2: TO_LAUNCH(2); break;
    case 3: TO_LAUNCH(3); break;
    default: TO_LAUNCH(4); break;
  }
#undef TO_LAUNCH
  EDA_CHECK_LAUNCH();
  return 0;
}
