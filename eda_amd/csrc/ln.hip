// ln.hip -- fused residual + dropout + LayerNorm for gfx950.
//
// The reference's post-norm blocks (models/encoder_decoder_layers.py:94-96,106-122,154-156,
// 184-186,371-405) all have the shape   x = norm(x + dropout(y))   with y coming out of an
// attention out-projection or the second FFN linear: three launches forward (dropout, add,
// layer_norm) and about five backward in stock PyTorch.  Here: one forward kernel (one wave
// per row of d_model = 288 features, statistics by wave reductions, two-pass variance like
// torch) and one backward kernel that produces dx, dy (dropout mask regenerated from the same
// counter-based hash as the attention kernels) and accumulates d(gamma), d(beta).
#include "eda_common.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_MAXI = 16;            // features per lane: C <= 1024

__device__ __forceinline__ unsigned ln_hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float ln_wave_sum(float v) { return eda_wave_sum_f32(v); }

struct LnDrop {
  bool on; unsigned seed, thresh; float inv_keep;
};
__device__ __forceinline__ LnDrop ln_drop(float p, const unsigned long long *seed_ptr, unsigned salt) {
  LnDrop d; d.on = p > 0.f; d.seed = 0; d.thresh = 0; d.inv_keep = 1.f;
  if (d.on) {
    d.seed = ln_hash32((unsigned)(*seed_ptr) * 0x9E3779B1u + salt);
    d.thresh = (unsigned)((double)p * 4294967296.0);
    d.inv_keep = 1.f / (1.f - p);
  }
  return d;
}

template <int NI, bool EXTRA>
__global__ __launch_bounds__(LN_THREADS) void add_dropout_ln_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ ybias,
    const float *__restrict__ gamma, const float *__restrict__ beta, long R, int C, float eps, float p,
    const unsigned long long *seed_ptr, unsigned salt, float *__restrict__ out,
    float *__restrict__ mean_out, float *__restrict__ rstd_out, const float *__restrict__ pos,
    float *__restrict__ out_pos) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= R) return;
  const LnDrop d = ln_drop(p, seed_ptr, salt);
  float v[NI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    float t = 0.f;
    if (c < C) {
      float yy = y[row * C + c] + (ybias ? ybias[c] : 0.f);
      if (d.on) yy = ln_hash32(d.seed ^ (unsigned)(row * C + c)) >= d.thresh ? yy * d.inv_keep : 0.f;
      t = x[row * C + c] + yy;
    }
    v[i] = t;
    s += t;
  }
  const float mean = ln_wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    const float dlt = c < C ? v[i] - mean : 0.f;
    q += dlt * dlt;
  }
  const float rstd = 1.f / sqrtf(ln_wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    if (c < C) {
      const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
      out[row * C + c] = o;
      if (EXTRA) out_pos[row * C + c] = o + pos[row * C + c];        // the next block's query = out + pos
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Backward.  One wave per row, two rows in flight per wave; the per-feature sums for
// d(gamma)/d(beta) are accumulated in registers over the wave's rows, merged over the block's
// 4 waves in LDS and written as ONE partial row per block (no atomics); ln_reduce_partials
// then adds the <= 1024 partial rows.
template <int NI, bool EXTRA>
__global__ __launch_bounds__(LN_THREADS) void add_dropout_ln_bwd_kernel(
    const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ y,
    const float *__restrict__ ybias, const float *__restrict__ gamma,
    const float *__restrict__ mean_in, const float *__restrict__ rstd_in, long R, int C, float p,
    const unsigned long long *seed_ptr, unsigned salt, float *__restrict__ dx,
    float *__restrict__ dy, float *__restrict__ partial, const float *__restrict__ dout2) {
  __shared__ float red[3][LN_THREADS / 64][64 * NI];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const LnDrop d = ln_drop(p, seed_ptr, salt);
  float g[NI], yb[NI], dg[NI], db[NI], dyb[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    g[i] = c < C ? gamma[c] : 0.f;
    yb[i] = (ybias && c < C) ? ybias[c] : 0.f;
    dg[i] = 0.f; db[i] = 0.f; dyb[i] = 0.f;
  }
  const long nwaves = (long)gridDim.x * (LN_THREADS / 64);
  for (long row0 = (long)blockIdx.x * (LN_THREADS / 64) + wave; row0 < R; row0 += 2 * nwaves) {
    // two independent rows (row0, row0 + nwaves) so that both rows' loads are in flight together
    float xv[2][NI], yv[2][NI], gv[2][NI];
    float mean[2], rstd[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long row = row0 + u * nwaves;
      live[u] = row < R;
      mean[u] = live[u] ? mean_in[row] : 0.f;
      rstd[u] = live[u] ? rstd_in[row] : 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        const bool ok = live[u] && c < C;
        xv[u][i] = ok ? x[row * C + c] : 0.f;
        yv[u][i] = (ok && y) ? y[row * C + c] : 0.f;
        gv[u][i] = ok ? (EXTRA ? dout[row * C + c] + dout2[row * C + c] : dout[row * C + c]) : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!live[u]) continue;                  // wave-uniform
      const long row = row0 + u * nwaves;
      float xh[NI], gd[NI];
      bool keep[NI];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        keep[i] = true;
        float yy = yv[u][i] + yb[i];
        if (d.on) {
          keep[i] = ln_hash32(d.seed ^ (unsigned)(row * C + c)) >= d.thresh;
          yy = keep[i] ? yy * d.inv_keep : 0.f;
        }
        if (!y) yy = 0.f;                      // x already is the pre-norm sum (fused linear + LayerNorm forward)
        const float xhat = c < C ? (xv[u][i] + yy - mean[u]) * rstd[u] : 0.f;
        const float go = gv[u][i];
        dg[i] += go * xhat;
        db[i] += go;
        xh[i] = xhat;
        gd[i] = go * g[i];
        s1 += gd[i];
        s2 += gd[i] * xhat;
      }
      s1 = ln_wave_sum(s1) / (float)C;
      s2 = ln_wave_sum(s2) / (float)C;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
          const float dv = rstd[u] * (gd[i] - s1 - xh[i] * s2);
          const float dyv = d.on ? (keep[i] ? dv * d.inv_keep : 0.f) : dv;
          dx[row * C + c] = dv;
          dy[row * C + c] = dyv;
          dyb[i] += dyv;            // gradient of the bias that was added to y (if any)
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    red[0][wave][lane + 64 * i] = dg[i]; red[1][wave][lane + 64 * i] = db[i]; red[2][wave][lane + 64 * i] = dyb[i];
  }
  __syncthreads();
  float *prow = partial + (long)blockIdx.x * 3 * C;
  for (int c = threadIdx.x; c < C; c += LN_THREADS) {
    float a = 0.f, b = 0.f, e = 0.f;
#pragma unroll
    for (int w = 0; w < LN_THREADS / 64; ++w) { a += red[0][w][c]; b += red[1][w][c]; e += red[2][w][c]; }
    prow[c] = a;
    prow[C + c] = b;
    prow[2 * C + c] = e;
  }
}

// out3[0..C) = d(gamma), [C..2C) = d(beta), [2C..3C) = d(y-bias): sums of the per-block partial rows.
// A 1024-thread block owns 64 of the 3C columns with 16 row lanes each (many loads in flight).
__global__ __launch_bounds__(1024) void ln_reduce_partials_kernel(const float *__restrict__ partial,
                                                                  int nblocks, int C,
                                                                  float *__restrict__ out3) {
  __shared__ float red[16][65];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;               // over 3*C
  float s = 0.f;
  if (i < 3 * C) {
#pragma unroll 8
    for (int b = rl; b < nblocks; b += 16) s += partial[(long)b * 3 * C + i];
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl == 0 && i < 3 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][col];
    out3[i] = t;
  }
}

// The same reduction for MANY LayerNorm sites in one launch (deferred by eda_amd/wgrad_queue.py:
// the gamma / beta / bias gradients are only needed by the optimizer step).  desc[site] =
// {partial, nblocks, C, dgamma, dbeta, dbias or 0, 0, 0}; grid = (column groups of 64 over 3*maxC, sites).
__global__ __launch_bounds__(1024) void ln_reduce_grouped_kernel(const long long *__restrict__ desc) {
  __shared__ float red[16][65];
  const long long *d = desc + (long)blockIdx.y * 8;
  const float *partial = reinterpret_cast<const float *>(d[0]);
  const int nblocks = (int)d[1], C = (int)d[2];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;               // over 3*C
  if (blockIdx.x * 64 >= 3 * C) return;
  float s = 0.f;
  if (i < 3 * C) {
#pragma unroll 8
    for (int b = rl; b < nblocks; b += 16) s += partial[(long)b * 3 * C + i];
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl == 0 && i < 3 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][col];
    float *out = reinterpret_cast<float *>(i < C ? d[3] : (i < 2 * C ? d[4] : d[5]));
    if (out) out[i < C ? i : (i < 2 * C ? i - C : i - 2 * C)] = t;
  }
}

// out[i] = keep(i) ? x[i] / (1 - p) : 0 with the library's counter-based mask (the hash of ln_drop over the flat element index):
// the frozen text encoder's embedding dropout (transformers RobertaEmbeddings: dropout(LayerNorm(.)), active because the reference
// trains with the whole model in train mode, main_utils.py:459)
__global__ __launch_bounds__(256) void dropout_flat_kernel(const float *__restrict__ x, long n, float p,
                                                           const unsigned long long *__restrict__ seed_ptr, unsigned salt,
                                                           float *__restrict__ out) {
  const LnDrop d = ln_drop(p, seed_ptr, salt);
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 v = *reinterpret_cast<const float4 *>(x + i);
    v.x = ln_hash32(d.seed ^ (unsigned)i) >= d.thresh ? v.x * d.inv_keep : 0.f;
    v.y = ln_hash32(d.seed ^ (unsigned)(i + 1)) >= d.thresh ? v.y * d.inv_keep : 0.f;
    v.z = ln_hash32(d.seed ^ (unsigned)(i + 2)) >= d.thresh ? v.z * d.inv_keep : 0.f;
    v.w = ln_hash32(d.seed ^ (unsigned)(i + 3)) >= d.thresh ? v.w * d.inv_keep : 0.f;
    *reinterpret_cast<float4 *>(out + i) = v;
  } else {
    for (long j = i; j < n; ++j) out[j] = ln_hash32(d.seed ^ (unsigned)j) >= d.thresh ? x[j] * d.inv_keep : 0.f;
  }
}

}  // namespace

// (EXTRA = the variant with the second output / second gradient input: a template flag, because a
// run-time `ptr ? ... : 0` in the load loop slowed EVERY call of the backward from 8.7 to 11.6 us)
#define LN_DISPATCH_X(NI_EXPR, KERNEL, EXTRA, GRID, ...)                                                      \
  do {                                                                                                        \
    const int ni__ = (NI_EXPR);                                                                               \
    if (ni__ <= 5) hipLaunchKernelGGL((KERNEL<5, EXTRA>), GRID, dim3(LN_THREADS), 0, stream, __VA_ARGS__);    \
    else if (ni__ <= 8) hipLaunchKernelGGL((KERNEL<8, EXTRA>), GRID, dim3(LN_THREADS), 0, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<LN_MAXI, EXTRA>), GRID, dim3(LN_THREADS), 0, stream, __VA_ARGS__);        \
  } while (0)
#define LN_DISPATCH(NI_EXPR, KERNEL, EXTRA_COND, GRID, ...)                  \
  do {                                                                       \
    if (EXTRA_COND) LN_DISPATCH_X(NI_EXPR, KERNEL, true, GRID, __VA_ARGS__);   \
    else LN_DISPATCH_X(NI_EXPR, KERNEL, false, GRID, __VA_ARGS__);             \
  } while (0)

extern "C" int eda_add_dropout_ln_fwd_f32(const float *x, const float *y, const float *y_bias,
                                          const float *gamma, const float *beta, long R, int C,
                                          float eps, float p_drop,
                                          const unsigned long long *seed_ptr, unsigned salt,
                                          float *out, float *mean, float *rstd, const float *pos,
                                          float *out_pos, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && C <= 64 * LN_MAXI, "bad dimension (C <= 1024)");
  if (R == 0) return 0;
  EDA_CHECK_ARG(x && y && gamma && beta && out && mean && rstd, "null pointer");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "bad dropout arguments");
  EDA_CHECK_ARG(R * C < (1ll << 32), "R*C must fit the 32-bit dropout counter");
  const dim3 grid((unsigned)((R + LN_THREADS / 64 - 1) / (LN_THREADS / 64)));
  EDA_CHECK_ARG((pos == nullptr) == (out_pos == nullptr), "pos and out_pos go together");
  LN_DISPATCH((C + 63) / 64, add_dropout_ln_fwd_kernel, pos != nullptr, grid, x, y, y_bias, gamma, beta, R, C, eps, p_drop,
              seed_ptr, salt, out, mean, rstd, pos, out_pos);
  EDA_CHECK_LAUNCH();
  return 0;
}

#define LN_BWD_MAX_BLOCKS 1024

extern "C" size_t eda_add_dropout_ln_bwd_workspace_bytes(long R, int C) {
  (void)R;
  return sizeof(float) * (size_t)LN_BWD_MAX_BLOCKS * 3 * (size_t)(C > 0 ? C : 0);
}

// grads3: 3*C floats out = [d(gamma) | d(beta) | d(y_bias)].
// ws: eda_add_dropout_ln_bwd_workspace_bytes(R, C) bytes of scratch (per-block partial sums).
extern "C" int eda_add_dropout_ln_bwd_f32(const float *dout, const float *x, const float *y,
                                          const float *y_bias, const float *gamma, const float *mean,
                                          const float *rstd, long R, int C, float p_drop,
                                          const unsigned long long *seed_ptr, unsigned salt,
                                          float *dx, float *dy, float *grads3, void *ws,
                                          size_t ws_bytes, const float *dout2, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && C <= 64 * LN_MAXI, "bad dimension (C <= 1024)");
  // grads3 == NULL: leave the per-block partial sums in `ws` (eda_add_dropout_ln_bwd_blocks(R) rows
  // of 3*C floats) for a later eda_ln_reduce_grouped_f32 over many sites
  if (R == 0) return grads3 ? eda_zero_async(grads3, sizeof(float) * 3 * C, stream) : 0;
  EDA_CHECK_ARG(dout && x && gamma && mean && rstd && dx && dy && ws, "null pointer");   // y == NULL: x is the pre-norm sum
  if (ws_bytes < eda_add_dropout_ln_bwd_workspace_bytes(R, C)) {
    eda_set_error("add_dropout_ln_bwd: workspace too small");
    return EDA_ERR_WORKSPACE;
  }
  // 8 rows per block (2 per wave, both in flight) up to 1024 blocks, then grid-stride
  long blocks = (R + 7) / 8;
  if (blocks > LN_BWD_MAX_BLOCKS) blocks = LN_BWD_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks);
  float *partial = reinterpret_cast<float *>(ws);
  LN_DISPATCH((C + 63) / 64, add_dropout_ln_bwd_kernel, dout2 != nullptr, grid, dout, x, y, y_bias, gamma, mean, rstd, R, C,
              p_drop, seed_ptr, salt, dx, dy, partial, dout2);
  EDA_CHECK_LAUNCH();
  if (grads3) {
    hipLaunchKernelGGL(ln_reduce_partials_kernel, dim3((3 * C + 63) / 64), dim3(1024), 0, stream, partial,
                       (int)blocks, C, grads3);
    EDA_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int eda_add_dropout_ln_bwd_blocks(long R) {
  long blocks = (R + 7) / 8;
  if (blocks > LN_BWD_MAX_BLOCKS) blocks = LN_BWD_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

extern "C" int eda_ln_reduce_grouped_f32(const long long *desc, int nsites, int max_c, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(nsites >= 0 && max_c > 0 && max_c <= 64 * LN_MAXI, "bad dimension");
  if (nsites == 0) return 0;
  EDA_CHECK_ARG(desc, "null pointer");
  hipLaunchKernelGGL(ln_reduce_grouped_kernel, dim3((unsigned)((3 * max_c + 63) / 64), (unsigned)nsites), dim3(1024),
                     0, stream, desc);
  EDA_CHECK_LAUNCH();
  return 0;
}

// ---- row-wise L2 normalisation (torch.nn.functional.normalize(x, p=2, dim=-1)) -----------------
// The reference normalises the 64-channel contrastive projections of the queries (7 heads) and of the
// text tokens (models/bdetr.py:224-226, 262-264, 316-320): unfused that is 3 launches forward and ~12
// backward per call.  One wave per row, lane l owns columns l, l+64, ...
namespace {
constexpr int L2_MAXC = 1024;

__device__ __forceinline__ float l2_wave_sum(float v) { return eda_wave_sum_f32(v); }

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float *__restrict__ x, long R, int C, float eps,
                                                         float *__restrict__ y, float *__restrict__ norm) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long r = wave; r < R; r += nwaves) {
    const float *xr = x + r * C;
    float v[L2_MAXC / 64];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < L2_MAXC / 64; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < C ? xr[c] : 0.f;
      ss += v[i] * v[i];
    }
    const float n = sqrtf(l2_wave_sum(ss));
    const float inv = 1.f / fmaxf(n, eps);
#pragma unroll
    for (int i = 0; i < L2_MAXC / 64; ++i) {
      const int c = lane + 64 * i;
      if (c < C) y[r * C + c] = v[i] * inv;
    }
    if (lane == 0) norm[r] = n;
  }
}

// dx = (dy - y <dy, y>) / n where n > eps (y = x / n); rows clamped at eps: dx = dy / eps
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                         const float *__restrict__ norm, long R, int C, float eps,
                                                         float *__restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long r = wave; r < R; r += nwaves) {
    float g[L2_MAXC / 64], yy[L2_MAXC / 64];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < L2_MAXC / 64; ++i) {
      const int c = lane + 64 * i;
      g[i] = c < C ? dy[r * C + c] : 0.f;
      yy[i] = c < C ? y[r * C + c] : 0.f;
      dot += g[i] * yy[i];
    }
    dot = l2_wave_sum(dot);
    const float n = norm[r];
    const bool clamped = !(n > eps);
    const float inv = 1.f / fmaxf(n, eps);
    if (clamped) dot = 0.f;
#pragma unroll
    for (int i = 0; i < L2_MAXC / 64; ++i) {
      const int c = lane + 64 * i;
      if (c < C) dx[r * C + c] = (g[i] - yy[i] * dot) * inv;
    }
  }
}
}  // namespace

extern "C" int eda_l2norm_rows_fwd_f32(const float *x, long R, int C, float eps, float *y, float *norm,
                                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && C <= L2_MAXC, "rows of 1..1024 floats");
  if (R == 0) return 0;
  EDA_CHECK_ARG(x && y && norm, "null pointer");
  long blocks = (R + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, R, C, eps, y, norm);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_l2norm_rows_bwd_f32(const float *dy, const float *y, const float *norm, long R, int C,
                                       float eps, float *dx, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && C <= L2_MAXC, "rows of 1..1024 floats");
  if (R == 0) return 0;
  EDA_CHECK_ARG(dy && y && norm && dx, "null pointer");
  long blocks = (R + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dy, y, norm, R, C, eps, dx);
  EDA_CHECK_LAUNCH();
  return 0;
}

// x, out: n floats, 16-byte aligned (out may be x); 0 < p < 1; the mask of element i is the hash of (counter, salt, i)
extern "C" int eda_dropout_f32(const float *x, long n, float p_drop, const unsigned long long *seed_ptr, unsigned salt, float *out,
                               void *stream_) {
  EDA_CHECK_ARG(n >= 0 && p_drop > 0.f && p_drop < 1.f, "0 < p < 1");
  if (n == 0) return 0;
  EDA_CHECK_ARG(x && out && seed_ptr, "null pointer");
  EDA_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0, "16-byte aligned buffers");
  hipLaunchKernelGGL(dropout_flat_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream_, x, n, p_drop,
                     seed_ptr, salt, out);
  EDA_CHECK_LAUNCH();
  return 0;
}
