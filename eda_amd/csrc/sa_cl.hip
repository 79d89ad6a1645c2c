// sa_cl.hip -- the set-abstraction "grouped MLP" pipeline in channels-last layout.
//
// Reference path (pointnet2/pointnet2_utils.py:317-376 QueryAndGroup, then
// pointnet2/pytorch_utils.py SharedMLP = 3 x [conv1x1 -> BatchNorm2d -> ReLU], then
// F.max_pool2d over the nsample axis, pointnet2_modules.py:243-257): every step
// round-trips a (B, C, npoint, nsample) tensor through HBM -- group xyz, subtract
// centre, divide, group features, concat, conv out, BN out, ReLU, pool.
//
// Here a position (scene, centre j, neighbour k) is a ROW of a (B*m*ns, C) matrix.  On the training path
// (eda_sa_fused_fwd/bwd_f32 below) one native call per direction runs the whole module body on this repo's own
// MFMA kernels (csrc/gemm.hip, csrc/wgrad.hip -- no library GEMM):
//  * layer 0 gathers [ (xyz[idx]-centre)*(1/r) | feats[idx] ] rows straight into the product's operand staging
//    (gemm_gather3_kernel / the gather prologue of gemm_stream_kernel): the grouped tensor never exists;
//  * layers >= 1 apply the previous layer's BatchNorm + ReLU while staging; every layer's BatchNorm statistics
//    (column sums of y, y^2) come out of its accumulators (fp64 atomics once per workgroup, last one finalises);
//  * BN + ReLU + max-pool over the ns consecutive rows of a centre (arg-max kept for the backward) in one pass
//    over the last layer's pre-activations (bn_relu_pool_kernel);
//  * backward: weight gradients with the forward's prologue recomputed while staging (wgrad_x_kernel), input
//    gradients as products against W^T with the ReLU mask + BN-backward reductions in the epilogue, one
//    element-wise pass dz = A*gy + B*z + D (bn_relu_bwd_apply_kernel), the pooled last layer's dz formed while
//    staging; layer 0's input gradient is scattered straight into d(features).
// The stand-alone kernels of this file (group_concat_cl, bn_stats / bn_relu_apply, bn_relu_bwd_*) serve the
// generic op API and the EDA_SA_FUSED=0 path.
// Results equal the reference's to fp32 rounding (tested against goldens generated
// by the reference's own modules, tests/golden/model_sa_*).
#include "eda_common.h"
#include "peer.h"
#include "gemm.h"
#include <string.h>

namespace {

constexpr int CL_THREADS = 256;

// ---------------------------------------------------------------- group + concat
// X[b, j*ns+k, :] = [ (xyz[b, idx, :] - new_xyz[b, j, :]) * inv_radius , feats[b, idx, :] ]
__global__ __launch_bounds__(CL_THREADS) void group_concat_cl_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats, const int *__restrict__ idx, int n, int m, int ns, int c,
    float inv_radius, float *__restrict__ out) {
  const int cw = 3 + c;
  const long rows_per_scene = (long)m * ns;
  const int b = blockIdx.y;
  const long total = rows_per_scene * cw;
  const float *xyz_b = xyz + (long)b * n * 3;
  const float *ctr_b = new_xyz + (long)b * m * 3;
  const float *f_b = feats ? feats + (long)b * n * c : nullptr;
  const int *idx_b = idx + (long)b * rows_per_scene;
  float *out_b = out + (long)b * total;
  for (long e = (long)blockIdx.x * CL_THREADS + threadIdx.x; e < total;
       e += (long)gridDim.x * CL_THREADS) {
    const long row = e / cw;
    const int ch = (int)(e - row * cw);
    const int src = idx_b[row];
    float v;
    if (ch < 3) {
      const int j = (int)(row / ns);
      v = (xyz_b[(long)src * 3 + ch] - ctr_b[(long)j * 3 + ch]) * inv_radius;
    } else {
      v = f_b[(long)src * c + (ch - 3)];
    }
    out_b[e] = v;
  }
}

// dfeats[b, idx, :] += dX[b, row, 3:]
__global__ __launch_bounds__(CL_THREADS) void group_concat_cl_grad_kernel(
    const float *__restrict__ dx, const int *__restrict__ idx, int n, int m, int ns, int c,
    float *__restrict__ dfeats) {
  const int cw = 3 + c;
  const long rows_per_scene = (long)m * ns;
  const int b = blockIdx.y;
  const long total = rows_per_scene * c;
  const int *idx_b = idx + (long)b * rows_per_scene;
  const float *dx_b = dx + (long)b * rows_per_scene * cw;
  float *df_b = dfeats + (long)b * n * c;
  for (long e = (long)blockIdx.x * CL_THREADS + threadIdx.x; e < total;
       e += (long)gridDim.x * CL_THREADS) {
    const long row = e / c;
    const int ch = (int)(e - row * c);
    atomicAdd(df_b + (long)idx_b[row] * c + ch, dx_b[row * cw + 3 + ch]);
  }
}

// ---------------------------------------------------------------- BN statistics
// Column sums of Z (R, C): every block owns a slab of rows, sums it in fp32
// (<= a few hundred terms per partial), then merges into fp64 accumulators.
// Thread t handles column (t % C4)*4.. +3 of rows (t / C4) + k * rows_per_pass.
// Global-batch statistics inside the kernels (SyncBatchNorm, main_utils.py:336-338, without a collective): world > 0 = the
// per-channel sums are exchanged with the other ranks through peer memory (csrc/peer.h) where they are formed; the row
// count of the statistics is then R * world.
struct BnSync { EdaPeer P; int world; };
eda_bn_sync_fn g_sync_fn = nullptr;
void *g_sync_user = nullptr;
int g_sync_world = 1;
bool g_sync_native = false;      // the hook is eda_peer_bn_hook and the single-launch kernels exchange their sums themselves
bool bn_sync_on() { return g_sync_fn != nullptr; }   // (world == 1 with a hook: the split kernels alone -- tests)
BnSync bn_sync_native_arg() {
  BnSync S;
  memset(&S, 0, sizeof(S));
  const EdaPeer *P = eda_peer_active();
  if (g_sync_native && P) { S.P = *P; S.world = g_sync_world; }
  return S;
}
bool bn_sync_native_on() { return g_sync_native && eda_peer_active() != nullptr; }
struct BnFinalize {          // what the last block of bn_stats_kernel needs to finish the statistics
  const float *gamma, *beta;
  float eps, momentum;
  float *running_mean, *running_var, *mean_out, *rstd_out, *scale, *shift;
  unsigned *ticket;
  BnSync sync;
};

template <int VEC>
__global__ __launch_bounds__(CL_THREADS) void bn_stats_kernel(const float *__restrict__ z, long R,
                                                              int C, long rows_per_block,
                                                              double *__restrict__ sum,
                                                              double *__restrict__ sumsq, BnFinalize fin) {
  __shared__ float red[2][CL_THREADS * VEC];
  __shared__ int is_last;
  const int cgroups = C / VEC;                       // threads per row
  const int rpp = CL_THREADS / cgroups;              // rows per pass (threads beyond rpp*cgroups idle)
  const int tcol = threadIdx.x % cgroups, trow = threadIdx.x / cgroups;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  float s[VEC], q[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { s[v] = 0.f; q[v] = 0.f; }
  if (trow < rpp) {
    long r = r0 + trow;
    if (VEC == 4) {
      // eight rows per iteration: eight independent 16-byte loads in flight per thread (one load
      // per iteration ran at 2.9 TB/s; four at 3.6 TB/s)
      for (; r + 7 * rpp < r1; r += 8 * rpp) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4 *>(z + (r + (long)u * rpp) * C + tcol * VEC);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s[0] += x[u].x; s[1] += x[u].y; s[2] += x[u].z; s[3] += x[u].w;
          q[0] += x[u].x * x[u].x; q[1] += x[u].y * x[u].y; q[2] += x[u].z * x[u].z; q[3] += x[u].w * x[u].w;
        }
      }
    }
    for (; r < r1; r += rpp) {
      const float *p = z + r * C + tcol * VEC;
      if (VEC == 4) {
        const float4 x = *reinterpret_cast<const float4 *>(p);
        s[0] += x.x; s[1] += x.y; s[2] += x.z; s[3] += x.w;
        q[0] += x.x * x.x; q[1] += x.y * x.y; q[2] += x.z * x.z; q[3] += x.w * x.w;
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) { const float x = p[v]; s[v] += x; q[v] += x * x; }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    red[0][threadIdx.x * VEC + v] = s[v];
    red[1][threadIdx.x * VEC + v] = q[v];
  }
  __syncthreads();
  // column c = tcol*VEC+v lives at threads with the same tcol: reduce over trow
  for (int col = threadIdx.x; col < C; col += CL_THREADS) {
    const int tc = col / VEC, v = col - tc * VEC;
    double a = 0.0, bq = 0.0;
    for (int tr = 0; tr < rpp; ++tr) {
      const int t = tr * cgroups + tc;
      a += (double)red[0][t * VEC + v];
      bq += (double)red[1][t * VEC + v];
    }
    // RETURNING atomics: the wave waits for the old values, i.e. for the read-modify-writes to have
    // been performed at the memory side, before this block may take its ticket.  (With fire-and-forget
    // atomics the ticket could overtake them: the last block then finalised incomplete sums -- observed
    // as step-to-step noise in the loss.)
    const double o1 = atomicAdd(sum + col, a);
    const double o2 = atomicAdd(sumsq + col, bq);
    asm volatile("" ::"v"(o1), "v"(o2));
  }
  // The block that draws the last ticket turns the sums into mean / rstd / scale / shift and the
  // running statistics (torch.nn.BatchNorm: biased variance normalises, unbiased variance is
  // tracked) and leaves sums and ticket at ZERO for the next call: no zero-fill launch before and
  // no finalize launch after this kernel.
  __syncthreads();                           // this block's atomics have completed
  if (threadIdx.x == 0)
    is_last = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!is_last) return;
  const unsigned long long pseq = fin.sync.world > 0 ? eda_peer_seq(fin.sync.P) : 0ull;
  const double Rt = fin.sync.world > 0 ? (double)R * fin.sync.world : (double)R;
  for (int c = threadIdx.x; c < C; c += CL_THREADS) {
    double su = __hip_atomic_load(sum + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double sq = __hip_atomic_load(sumsq + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sum + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sumsq + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fin.sync.world > 0) eda_peer_exchange2(fin.sync.P, pseq, c, su, sq);
    const double mean = su / Rt;
    double var = sq / Rt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)fin.eps));
    const float meanf = (float)mean;
    fin.mean_out[c] = meanf;
    fin.rstd_out[c] = rstd;
    const float sc = fin.gamma[c] * rstd;
    fin.scale[c] = sc;
    fin.shift[c] = fin.beta[c] - meanf * sc;
    if (fin.running_mean) {
      const double unbiased = Rt > 1.0 ? var * Rt / (Rt - 1.0) : var;
      fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * meanf;
      fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
    }
  }
  if (fin.sync.world > 0) {                  // (only this block exchanges: a one-workgroup launch for the peer protocol)
    __syncthreads();
    if (threadIdx.x == 0) eda_peer_done(fin.sync.P, 1u);
  }
  if (threadIdx.x == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (kept for reference / eval-mode callers that already hold the sums)
__global__ void bn_finalize_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq,
                                   long R, int C, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, float eps, float momentum,
                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                   float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                   float *__restrict__ scale, float *__restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = sum[c] / (double)R;
  double var = sumsq[c] / (double)R - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  mean_out[c] = meanf;
  rstd_out[c] = rstd;
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - meanf * sc;
  if (running_mean) {
    const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_affine_kernel(const float *__restrict__ running_mean,
                                      const float *__restrict__ running_var,
                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                      float eps, int C, float *__restrict__ mean_out,
                                      float *__restrict__ rstd_out, float *__restrict__ scale,
                                      float *__restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(running_var[c] + eps);
  mean_out[c] = running_mean[c];
  rstd_out[c] = rstd;
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
}

// ---------------------------------------------------------------- BN + ReLU apply
// A = relu(Z*scale + shift), elementwise over (R, C); C % 4 == 0.
__global__ __launch_bounds__(CL_THREADS) void bn_relu_apply_kernel(const float *__restrict__ z,
                                                                   long total4, int C,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift,
                                                                   float *__restrict__ a) {
  for (long i = (long)blockIdx.x * CL_THREADS + threadIdx.x; i < total4;
       i += (long)gridDim.x * CL_THREADS) {
    const int col = (int)((i * 4) % C);
    const float4 x = reinterpret_cast<const float4 *>(z)[i];
    const float4 sc = *reinterpret_cast<const float4 *>(scale + col);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + col);
    float4 y;
    y.x = fmaxf(x.x * sc.x + sh.x, 0.f);
    y.y = fmaxf(x.y * sc.y + sh.y, 0.f);
    y.z = fmaxf(x.z * sc.z + sh.z, 0.f);
    y.w = fmaxf(x.w * sc.w + sh.w, 0.f);
    reinterpret_cast<float4 *>(a)[i] = y;
  }
}

// pooled[g, c] = max_k relu(Z[g*ns+k, c]*scale+shift); argmax[g, c] = first k attaining it.
// Thread = four consecutive channels of one group (16-byte loads), four rows in flight.
__global__ __launch_bounds__(CL_THREADS) void bn_relu_pool_kernel(const float *__restrict__ z,
                                                                  long G, int ns, int C,
                                                                  const float *__restrict__ scale,
                                                                  const float *__restrict__ shift,
                                                                  float *__restrict__ pooled,
                                                                  unsigned char *__restrict__ argmax) {
  const int c4n = C / 4;
  const long total = G * c4n;
  for (long i = (long)blockIdx.x * CL_THREADS + threadIdx.x; i < total;
       i += (long)gridDim.x * CL_THREADS) {
    const long g = i / c4n;
    const int c = (int)(i - g * c4n) * 4;
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
    const float *p = z + (g * ns) * C + c;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    auto take = [&](const float4 &x, int k) {
      const float y[4] = {fmaxf(x.x * sc.x + sh.x, 0.f), fmaxf(x.y * sc.y + sh.y, 0.f),
                          fmaxf(x.z * sc.z + sh.z, 0.f), fmaxf(x.w * sc.w + sh.w, 0.f)};
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (y[v] > best[v]) { best[v] = y[v]; bi[v] = k; }
    };
    int k = 0;
    for (; k + 3 < ns; k += 4) {
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4 *>(p + (long)(k + u) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) take(x[u], k + u);
    }
    for (; k < ns; ++k) take(*reinterpret_cast<const float4 *>(p + (long)k * C), k);
    *reinterpret_cast<float4 *>(pooled + g * C + c) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<uchar4 *>(argmax + g * C + c) =
        make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
  }
}

// ---------------------------------------------------------------- backward
// dY = dA * (Y > 0);  s1 = sum dY, s2 = sum dY * xhat  (per channel, fp64 merge)
// POOL: dA is given per (group, channel) and lands on row argmax only.
template <bool POOL>
__global__ __launch_bounds__(CL_THREADS) void bn_relu_bwd_stats_kernel(
    const float *__restrict__ da, const unsigned char *__restrict__ argmax,
    const float *__restrict__ z, long R, int C, int ns, long rows_per_block,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ scale, const float *__restrict__ shift, double *__restrict__ s1,
    double *__restrict__ s2) {
  __shared__ float red[2][CL_THREADS];
  // one thread per column, CL_THREADS / C row lanes (C <= 256 assumed by the launcher, else loops)
  const int lanes = CL_THREADS / C > 0 ? CL_THREADS / C : 1;
  for (int cbase = 0; cbase < C; cbase += CL_THREADS) {
    const int col = cbase + (int)(threadIdx.x % (C < CL_THREADS ? C : CL_THREADS));
    const int rl = threadIdx.x / (C < CL_THREADS ? C : CL_THREADS);
    float a1 = 0.f, a2 = 0.f;
    if (col < C && rl < lanes) {
      const float mu = mean[col], rs = rstd[col], sc = scale[col], sh = shift[col];
      const long r0 = (long)blockIdx.x * rows_per_block;
      const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
      if (POOL) {
        // rows_per_block is a multiple of ns: iterate groups
        for (long g = r0 / ns + rl; g < r1 / ns; g += lanes) {
          const int k = argmax[g * C + col];
          const float x = z[(g * ns + k) * C + col];
          const float y = x * sc + sh;
          const float dy = y > 0.f ? da[g * C + col] : 0.f;
          a1 += dy;
          a2 += dy * (x - mu) * rs;
        }
      } else {
        for (long r = r0 + rl; r < r1; r += lanes) {
          const float x = z[r * C + col];
          const float y = x * sc + sh;
          const float dy = y > 0.f ? da[r * C + col] : 0.f;
          a1 += dy;
          a2 += dy * (x - mu) * rs;
        }
      }
    }
    red[0][threadIdx.x] = a1;
    red[1][threadIdx.x] = a2;
    __syncthreads();
    if (rl == 0 && col < C) {
      double t1 = 0.0, t2 = 0.0;
      const int stride = C < CL_THREADS ? C : CL_THREADS;
      for (int l = 0; l < lanes; ++l) {
        t1 += (double)red[0][l * stride + (col - cbase)];
        t2 += (double)red[1][l * stride + (col - cbase)];
      }
      atomicAdd(s1 + col, t1);
      atomicAdd(s2 + col, t2);
    }
    __syncthreads();
  }
}

// dZ = gamma*rstd * (dY - s1/R - xhat * s2/R).  Thread = 4 consecutive channels of a row
// (16-byte loads/stores); per-channel constants are folded into  dz = A*dy + B*x + D.
template <bool POOL>
__global__ __launch_bounds__(CL_THREADS) void bn_relu_bwd_apply_kernel(
    const float *__restrict__ da, const unsigned char *__restrict__ argmax,
    const float *__restrict__ z, long R, int C, int ns, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ gamma,
    const double *__restrict__ s1, const double *__restrict__ s2, int train,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dz, float invR, float gscale) {
  // invR = 1 / (rows the statistics were taken over): 1/R, or 1/(R * ranks) with global-batch statistics, where s1 / s2
  // are the all-reduced sums and gscale = 1/ranks turns them into what the gradient all-reduce's mean expects
  const int cgroups = C / 4;
  const int rpp = CL_THREADS / cgroups;
  const int tcol = threadIdx.x % cgroups, trow = threadIdx.x / cgroups;
  if (trow >= rpp) return;
  const int c0 = tcol * 4;
  float sc[4], sh[4], ka[4], kb[4], kd[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int c = c0 + v;
    if (blockIdx.x == 0 && trow == 0) { dbeta[c] = (float)s1[c] * gscale; dgamma[c] = (float)s2[c] * gscale; }
    sc[v] = scale[c]; sh[v] = shift[c];
    const float gr = gamma[c] * rstd[c];
    ka[v] = gr;
    if (train) {
      // gr*(dy - s1/R - (x-mu)*rs*s2/R) = gr*dy + x*(-gr*rs*s2/R) + gr*(mu*rs*s2/R - s1/R)
      const float t2 = rstd[c] * (float)s2[c] * invR;
      kb[v] = -gr * t2;
      kd[v] = gr * (mean[c] * t2 - (float)s1[c] * invR);
    } else {
      kb[v] = 0.f; kd[v] = 0.f;
    }
  }
  for (long r = (long)blockIdx.x * rpp + trow; r < R; r += (long)gridDim.x * rpp) {
    const float4 x4 = *reinterpret_cast<const float4 *>(z + r * C + c0);
    const float x[4] = {x4.x, x4.y, x4.z, x4.w};
    float dy[4];
    if (POOL) {
      const long g = r / ns;
      const int k = (int)(r - g * ns);
      const uchar4 am = *reinterpret_cast<const uchar4 *>(argmax + g * C + c0);
      const float4 d4 = *reinterpret_cast<const float4 *>(da + g * C + c0);
      dy[0] = am.x == k ? d4.x : 0.f; dy[1] = am.y == k ? d4.y : 0.f;
      dy[2] = am.z == k ? d4.z : 0.f; dy[3] = am.w == k ? d4.w : 0.f;
    } else {
      const float4 d4 = *reinterpret_cast<const float4 *>(da + r * C + c0);
      dy[0] = d4.x; dy[1] = d4.y; dy[2] = d4.z; dy[3] = d4.w;
    }
    float o[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float y = x[v] * sc[v] + sh[v];
      const float d = y > 0.f ? dy[v] : 0.f;
      o[v] = ka[v] * d + kb[v] * x[v] + kd[v];
    }
    *reinterpret_cast<float4 *>(dz + r * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Non-pooled backward statistics with 16-byte loads (same mapping as bn_stats_kernel).
__global__ __launch_bounds__(CL_THREADS) void bn_relu_bwd_stats_vec_kernel(
    const float *__restrict__ da, const float *__restrict__ z, long R, int C, long rows_per_block,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ scale, const float *__restrict__ shift, double *__restrict__ s1,
    double *__restrict__ s2) {
  __shared__ float red[2][CL_THREADS * 4];
  const int cgroups = C / 4;
  const int rpp = CL_THREADS / cgroups;
  const int tcol = threadIdx.x % cgroups, trow = threadIdx.x / cgroups;
  const int c0 = tcol * 4;
  float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
  if (trow < rpp) {
    float sc[4], sh[4], mu[4], rs[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { sc[v] = scale[c0 + v]; sh[v] = shift[c0 + v]; mu[v] = mean[c0 + v]; rs[v] = rstd[c0 + v]; }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    long r = r0 + trow;
    for (; r < r1; r += rpp) {
      const float4 x4 = *reinterpret_cast<const float4 *>(z + r * C + c0);
      const float4 d4 = *reinterpret_cast<const float4 *>(da + r * C + c0);
      const float x[4] = {x4.x, x4.y, x4.z, x4.w};
      const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float y = x[v] * sc[v] + sh[v];
        const float dy = y > 0.f ? d[v] : 0.f;
        a1[v] += dy;
        a2[v] += dy * (x[v] - mu[v]) * rs[v];
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    red[0][threadIdx.x * 4 + v] = a1[v];
    red[1][threadIdx.x * 4 + v] = a2[v];
  }
  __syncthreads();
  for (int col = threadIdx.x; col < C; col += CL_THREADS) {
    const int tc = col / 4, v = col - tc * 4;
    double t1 = 0.0, t2 = 0.0;
    for (int tr = 0; tr < rpp; ++tr) {
      const int t = tr * cgroups + tc;
      t1 += (double)red[0][t * 4 + v];
      t2 += (double)red[1][t * 4 + v];
    }
    atomicAdd(s1 + col, t1);
    atomicAdd(s2 + col, t2);
  }
}

// ---------------------------------------------------------------- small inputs
// For the heads / positional embeddings (R = B*Q ~ 2048 rows) the multi-launch scheme
// above is launch-bound.  Here ONE launch does everything: a 1024-thread workgroup owns one
// float4 column group (4 channels) for ALL rows -- C/4 workgroups, every thread only 2-4 rows,
// so all loads are in flight at once; column sums by wave shuffles + a 16-entry LDS merge in
// fp64; the second sweep re-reads the same rows from L2.
constexpr int SM_THREADS = 1024;
constexpr int SM_WAVES = SM_THREADS / 64;

__device__ __forceinline__ float wave_sum64(float v) { return eda_wave_sum_f32(v); }

// Element dropout fused behind the ReLU (the heads' Conv-BN-ReLU-Dropout): the same counter-based
// hash as csrc/ln.hip (step counter x call-site salt, element index), regenerated in the backward.
struct BnDrop { bool on; unsigned seed, thresh; float inv_keep; };
// per-channel-group parameters of the single-launch kernels: channel c belongs to group c / cpg (sibling
// BatchNorm modules applied to column blocks of one row matrix, e.g. the three MLPs of a prediction head)
struct BnGrp {
  const float *gamma[4], *beta[4];
  float *running_mean[4], *running_var[4];
  unsigned salt[4];
  int cpg;
};
__device__ __forceinline__ unsigned bn_hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ BnDrop bn_drop(float p, const unsigned long long *seed_ptr, unsigned salt) {
  BnDrop d; d.on = p > 0.f && seed_ptr != nullptr; d.seed = 0; d.thresh = 0; d.inv_keep = 1.f;
  if (d.on) {
    d.seed = bn_hash32((unsigned)(*seed_ptr) * 0x9E3779B1u + salt);
    d.thresh = (unsigned)((double)p * 4294967296.0);
    d.inv_keep = 1.f / (1.f - p);
  }
  return d;
}
__device__ __forceinline__ float bn_keep(const BnDrop &d, unsigned idx) {
  return bn_hash32(d.seed ^ idx) >= d.thresh ? d.inv_keep : 0.f;
}

// CQ = float4 column groups per workgroup: CQ adjacent threads read 16*CQ contiguous bytes of a
// row (CQ = 4: 64-byte segments instead of 16-byte ones), rows are strided by SM_THREADS / CQ.
template <int CQ>
__global__ __launch_bounds__(SM_THREADS) void bn_relu_small_fwd_kernel(
    const float *__restrict__ z, int R, int C, const BnGrp G, float eps, float momentum, int training,
    float *__restrict__ mean_out,
    float *__restrict__ rstd_out, float *__restrict__ scale_out, float *__restrict__ shift_out,
    float *__restrict__ out, float p_drop, const unsigned long long *seed_ptr, const BnSync S) {
  constexpr int RL = SM_THREADS / CQ;              // row lanes
  __shared__ float red[8][CQ][SM_WAVES];
  __shared__ float sc_l[4 * CQ], sh_l[4 * CQ];
  const unsigned long long pseq = (S.world > 0 && training) ? eda_peer_seq(S.P) : 0ull;
  const int grp = (blockIdx.x * 4 * CQ) / G.cpg, gc0 = grp * G.cpg;     // (a block's channels lie in one group)
  const float *gamma = G.gamma[grp] - gc0, *beta = G.beta[grp] - gc0;
  float *running_mean = G.running_mean[grp] ? G.running_mean[grp] - gc0 : nullptr;
  float *running_var = G.running_var[grp] ? G.running_var[grp] - gc0 : nullptr;
  const BnDrop dr = bn_drop(p_drop, seed_ptr, G.salt[grp]);
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const int c0 = blockIdx.x * 4 * CQ + 4 * cq;     // this thread's four channels
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (training) {
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int r = rl; r < R; r += RL) {
      const float4 x = *reinterpret_cast<const float4 *>(z + (long)r * C + c0);
      s[0] += x.x; s[1] += x.y; s[2] += x.z; s[3] += x.w;
      q[0] += x.x * x.x; q[1] += x.y * x.y; q[2] += x.z * x.z; q[3] += x.w * x.w;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float a = s[v], b = q[v];
#pragma unroll
      for (int o = 32; o >= CQ; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }   // lanes of equal cq
      if (lane < CQ) { red[v][lane][wave] = a; red[4 + v][lane][wave] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 4 * CQ) {
      const int jq = threadIdx.x >> 2, v = threadIdx.x & 3;
      const int c = blockIdx.x * 4 * CQ + threadIdx.x;
      double a = 0.0, b = 0.0;
      for (int w = 0; w < SM_WAVES; ++w) { a += (double)red[v][jq][w]; b += (double)red[4 + v][jq][w]; }
      double Rt = (double)R;
      if (S.world > 0) { eda_peer_exchange2(S.P, pseq, c, a, b); Rt *= S.world; }     // global-batch sums (csrc/peer.h)
      const double mean = a / Rt;
      double var = b / Rt - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      const float meanf = (float)mean;
      const float sc = gamma[c] * rstd, sh = beta[c] - meanf * sc;
      mean_out[c] = meanf; rstd_out[c] = rstd; scale_out[c] = sc; shift_out[c] = sh;
      sc_l[threadIdx.x] = sc; sh_l[threadIdx.x] = sh;
      if (running_mean) {
        const double unbiased = Rt > 1.0 ? var * Rt / (Rt - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
  } else if (threadIdx.x < 4 * CQ) {
    const int c = blockIdx.x * 4 * CQ + threadIdx.x;
    const float rstd = 1.f / sqrtf(running_var[c] + eps);
    const float sc = gamma[c] * rstd, sh = beta[c] - running_mean[c] * sc;
    mean_out[c] = running_mean[c]; rstd_out[c] = rstd; scale_out[c] = sc; shift_out[c] = sh;
    sc_l[threadIdx.x] = sc; sh_l[threadIdx.x] = sh;
  }
  __syncthreads();
  if (S.world > 0 && training && threadIdx.x == 0) eda_peer_done(S.P, gridDim.x * gridDim.y);
  const float sc0 = sc_l[4 * cq], sc1 = sc_l[4 * cq + 1], sc2 = sc_l[4 * cq + 2], sc3 = sc_l[4 * cq + 3];
  const float sh0 = sh_l[4 * cq], sh1 = sh_l[4 * cq + 1], sh2 = sh_l[4 * cq + 2], sh3 = sh_l[4 * cq + 3];
#pragma unroll 4
  for (int r = rl; r < R; r += RL) {
    const float4 x = *reinterpret_cast<const float4 *>(z + (long)r * C + c0);
    float4 y;
    y.x = fmaxf(x.x * sc0 + sh0, 0.f); y.y = fmaxf(x.y * sc1 + sh1, 0.f);
    y.z = fmaxf(x.z * sc2 + sh2, 0.f); y.w = fmaxf(x.w * sc3 + sh3, 0.f);
    if (dr.on) {
      const unsigned e = (unsigned)(r * C + c0);
      y.x *= bn_keep(dr, e); y.y *= bn_keep(dr, e + 1); y.z *= bn_keep(dr, e + 2); y.w *= bn_keep(dr, e + 3);
    }
    *reinterpret_cast<float4 *>(out + (long)r * C + c0) = y;
  }
}

template <int CQ>
__device__ __forceinline__ void bn_relu_small_bwd_body(
    const float *__restrict__ da, const float *__restrict__ z, int R, int C,
    const BnGrp &G, const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ scale, const float *__restrict__ shift, int train,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dz, float p_drop,
    const unsigned long long *seed_ptr, const BnSync &S, int gbase);

template <int CQ>
__global__ __launch_bounds__(SM_THREADS) void bn_relu_small_bwd_kernel(
    const float *__restrict__ da, const float *__restrict__ z, int R, int C,
    const BnGrp G, const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ scale, const float *__restrict__ shift, int train,
    double *__restrict__ s1_out, double *__restrict__ s2_out, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ dz, float p_drop, const unsigned long long *seed_ptr, const BnSync S) {
  bn_relu_small_bwd_body<CQ>(da, z, R, C, G, mean, rstd, scale, shift, train, dgamma, dbeta, dz, p_drop, seed_ptr, S, 0);
}

// The same backward for up to BN_MAXMAT packed matrices of one shape in ONE launch (blockIdx.y = matrix): the seven
// prediction heads' BatchNorm+ReLU+Dropout backward passes depend on the loss alone and are issued together
// (eda_amd/heads_batched.py).
constexpr int BN_MAXMAT = 8;
struct BnMulti {
  const float *da[BN_MAXMAT], *z[BN_MAXMAT], *stats[BN_MAXMAT];   // stats: (4, C) rows mean | rstd | scale | shift
  float *dgb[BN_MAXMAT], *dz[BN_MAXMAT];                           // dgb: (2, C) rows dgamma | dbeta
  BnGrp grp[BN_MAXMAT];
};
template <int CQ>
__global__ __launch_bounds__(SM_THREADS) void bn_relu_small_bwd_multi_kernel(const BnMulti M, int R, int C, int train, float p_drop,
                                                                             const unsigned long long *seed_ptr, const BnSync S) {
  const int m = blockIdx.y;
  const float *st = M.stats[m];
  bn_relu_small_bwd_body<CQ>(M.da[m], M.z[m], R, C, M.grp[m], st, st + C, st + 2 * C, st + 3 * C, train, M.dgb[m], M.dgb[m] + C,
                             M.dz[m], p_drop, seed_ptr, S, m * C);
}

template <int CQ>
__device__ __forceinline__ void bn_relu_small_bwd_body(
    const float *__restrict__ da, const float *__restrict__ z, int R, int C,
    const BnGrp &G, const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ scale, const float *__restrict__ shift, int train,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dz, float p_drop,
    const unsigned long long *seed_ptr, const BnSync &S, int gbase) {
  constexpr int RL = SM_THREADS / CQ;              // row lanes (see the forward kernel)
  __shared__ float red[8][CQ][SM_WAVES];
  __shared__ float ka_l[4 * CQ], kb_l[4 * CQ], kd_l[4 * CQ];
  const bool psync = S.world > 0 && train;
  const unsigned long long pseq = psync ? eda_peer_seq(S.P) : 0ull;
  const int grp = (blockIdx.x * 4 * CQ) / G.cpg;
  const float *gamma = G.gamma[grp] - grp * G.cpg;
  const BnDrop dr = bn_drop(p_drop, seed_ptr, G.salt[grp]);
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const int c0 = blockIdx.x * 4 * CQ + 4 * cq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sc[4], sh[4], mu[4], rs[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) { sc[v] = scale[c0 + v]; sh[v] = shift[c0 + v]; mu[v] = mean[c0 + v]; rs[v] = rstd[c0 + v]; }
  float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
#pragma unroll 4
  for (int r = rl; r < R; r += RL) {
    const float4 x4 = *reinterpret_cast<const float4 *>(z + (long)r * C + c0);
    const float4 d4 = *reinterpret_cast<const float4 *>(da + (long)r * C + c0);
    const float x[4] = {x4.x, x4.y, x4.z, x4.w};
    float d[4] = {d4.x, d4.y, d4.z, d4.w};
    if (dr.on) {
#pragma unroll
      for (int v = 0; v < 4; ++v) d[v] *= bn_keep(dr, (unsigned)(r * C + c0 + v));
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float dy = (x[v] * sc[v] + sh[v]) > 0.f ? d[v] : 0.f;
      a1[v] += dy;
      a2[v] += dy * (x[v] - mu[v]) * rs[v];
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float a = a1[v], b = a2[v];
#pragma unroll
    for (int o = 32; o >= CQ; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }   // lanes of equal cq
    if (lane < CQ) { red[v][lane][wave] = a; red[4 + v][lane][wave] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 4 * CQ) {
    const int jq = threadIdx.x >> 2, v = threadIdx.x & 3;
    const int c = blockIdx.x * 4 * CQ + threadIdx.x;
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < SM_WAVES; ++w) { t1 += (double)red[v][jq][w]; t2 += (double)red[4 + v][jq][w]; }
    dbeta[c] = (float)t1; dgamma[c] = (float)t2;   // (the shared workspace stays untouched: it must remain zero)
    float invR = 1.f / (float)R;
    if (psync) {     // the mean terms of dz use the GLOBAL sums; d(gamma), d(beta) stay local (the gradient all-reduce adds them)
      eda_peer_exchange2(S.P, pseq, gbase + c, t1, t2);
      invR = 1.f / ((float)R * (float)S.world);
    }
    const float gr = gamma[c] * rstd[c];
    ka_l[threadIdx.x] = gr;
    if (train) {
      const float k2 = rstd[c] * (float)t2 * invR;
      kb_l[threadIdx.x] = -gr * k2;
      kd_l[threadIdx.x] = gr * (mean[c] * k2 - (float)t1 * invR);
    } else {
      kb_l[threadIdx.x] = 0.f; kd_l[threadIdx.x] = 0.f;
    }
  }
  __syncthreads();
  if (psync && threadIdx.x == 0) eda_peer_done(S.P, gridDim.x * gridDim.y);
  float ka[4], kb[4], kd[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) { ka[v] = ka_l[4 * cq + v]; kb[v] = kb_l[4 * cq + v]; kd[v] = kd_l[4 * cq + v]; }
#pragma unroll 4
  for (int r = rl; r < R; r += RL) {
    const float4 x4 = *reinterpret_cast<const float4 *>(z + (long)r * C + c0);
    const float4 d4 = *reinterpret_cast<const float4 *>(da + (long)r * C + c0);
    const float x[4] = {x4.x, x4.y, x4.z, x4.w};
    float d[4] = {d4.x, d4.y, d4.z, d4.w};
    if (dr.on) {
#pragma unroll
      for (int v = 0; v < 4; ++v) d[v] *= bn_keep(dr, (unsigned)(r * C + c0 + v));
    }
    float o[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float dy = (x[v] * sc[v] + sh[v]) > 0.f ? d[v] : 0.f;
      o[v] = ka[v] * dy + kb[v] * x[v] + kd[v];
    }
    *reinterpret_cast<float4 *>(dz + (long)r * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Column quads per workgroup of the single-launch kernels: 4 (64-byte row segments per 4 threads)
// measured ~8 % faster than 1 at 2048 x 288 (EDA_BN_SMALL_CQ=1/2/4 to compare).
int small_cq() { return (int)eda_knob(EDA_K_BN_SMALL_CQ); }

constexpr long SMALL_ROWS = 4096;    // below this the single-launch kernels win (at 8192 rows the strided sweep loses)

int grid_for(long work_items) {
  long g = (work_items + CL_THREADS - 1) / CL_THREADS;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int eda_group_concat_cl_f32(const float *xyz, const float *new_xyz, const float *feats_cl,
                                       const int *idx, int b, int n, int m, int ns, int c,
                                       float radius, int normalize_xyz, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && ns >= 0 && c >= 0, "negative dimension");
  if (b == 0 || m == 0 || ns == 0) return 0;
  EDA_CHECK_ARG(xyz && new_xyz && idx && out && (feats_cl || c == 0), "null pointer");
  EDA_CHECK_ARG(b <= 65535, "batch too large");
  // torch divides a tensor by a host scalar as x * (1/r) on the GPU; mirror that.
  const float inv = normalize_xyz ? 1.0f / radius : 1.0f;
  const long total = (long)m * ns * (3 + c);
  hipLaunchKernelGGL(group_concat_cl_kernel, dim3(grid_for(total), b), dim3(CL_THREADS), 0, stream,
                     xyz, new_xyz, feats_cl, idx, n, m, ns, c, inv, out);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_group_concat_cl_grad_f32(const float *dx, const int *idx, int b, int n, int m,
                                            int ns, int c, float *dfeats_cl, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && ns >= 0 && c >= 0, "negative dimension");
  if (b == 0 || n == 0 || c == 0) return 0;
  EDA_CHECK_ARG(dfeats_cl, "null pointer");
  if (eda_deterministic() && m > 0 && ns > 0 && (long)m * ns < 0x7fffffffL && b <= 65535) {
    EDA_CHECK_ARG(dx && idx, "bad arguments");
    const long rows = (long)m * ns;
    EdaDetScatter d = {idx, nullptr, rows, (int)rows, 1, dx + 3, rows * (3 + c), (long)(3 + c), 1,
                       dfeats_cl, (long)n * c, (long)c, 1, b, n, c};
    return eda_det_scatter_launch(d, stream);
  }
  { const int zrc__ = eda_zero_async(dfeats_cl, sizeof(float) * (size_t)b * n * c, stream); if (zrc__) return zrc__; }
  if (m == 0 || ns == 0) return 0;
  EDA_CHECK_ARG(dx && idx && b <= 65535, "bad arguments");
  const long total = (long)m * ns * c;
  hipLaunchKernelGGL(group_concat_cl_grad_kernel, dim3(grid_for(total), b), dim3(CL_THREADS), 0,
                     stream, dx, idx, n, m, ns, c, dfeats_cl);
  EDA_CHECK_LAUNCH();
  return 0;
}

// Forward of BatchNorm (batch or running statistics) + ReLU (+ max-pool over `pool`
// consecutive rows when pool > 1).  ws: 2*C doubles (zeroed here).  Outputs: mean,
// rstd, scale, shift (C floats each, kept for the backward), and either a (R,C) or
// pooled (R/pool,C) + argmax (R/pool,C bytes).
extern "C" long eda_bn_relu_dropout_max_rows(void) { return SMALL_ROWS; }

extern "C" int eda_bn_relu_fwd_f32(const float *z, long R, int C, const float *gamma,
                                   const float *beta, float eps, float momentum, int training,
                                   float *running_mean, float *running_var, int pool,
                                   double *ws, float *mean, float *rstd, float *scale, float *shift,
                                   float *out, unsigned char *argmax, float p_drop,
                                   const unsigned long long *seed_ptr, unsigned salt, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && pool >= 1, "bad dimension");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "bad dropout arguments");
  if (p_drop > 0.f && !(pool == 1 && R <= SMALL_ROWS)) {
    eda_set_error("bn_relu: fused dropout is built for pool == 1 and R <= eda_bn_relu_dropout_max_rows() only");
    return EDA_ERR_UNSUPPORTED;
  }
  EDA_CHECK_ARG(C % 4 == 0 && C <= 1024, "channel count must be a multiple of 4 (<= 1024)");
  EDA_CHECK_ARG(pool <= 255 && R % pool == 0, "rows must be a multiple of the pooling width");
  if (R == 0) return 0;
  EDA_CHECK_ARG(z && gamma && beta && mean && rstd && scale && shift && out, "null pointer");
  if (pool == 1 && R <= SMALL_ROWS) {
    EDA_CHECK_ARG(training || (running_mean && running_var), "eval mode needs running statistics");
    BnGrp G;
    memset(&G, 0, sizeof(G));
    G.gamma[0] = gamma; G.beta[0] = beta; G.running_mean[0] = running_mean; G.running_var[0] = running_var;
    G.salt[0] = salt; G.cpg = C;
    const int cq_env = small_cq();
    if (cq_env == 4 && C % 16 == 0)
      hipLaunchKernelGGL(bn_relu_small_fwd_kernel<4>, dim3(C / 16), dim3(SM_THREADS), 0, stream, z, (int)R, C,
                         G, eps, momentum, training, mean, rstd, scale,
                         shift, out, p_drop, seed_ptr, bn_sync_native_arg());
    else if (cq_env == 2 && C % 8 == 0)
      hipLaunchKernelGGL(bn_relu_small_fwd_kernel<2>, dim3(C / 8), dim3(SM_THREADS), 0, stream, z, (int)R, C,
                         G, eps, momentum, training, mean, rstd, scale,
                         shift, out, p_drop, seed_ptr, bn_sync_native_arg());
    else
      hipLaunchKernelGGL(bn_relu_small_fwd_kernel<1>, dim3(C / 4), dim3(SM_THREADS), 0, stream, z, (int)R, C,
                         G, eps, momentum, training, mean, rstd, scale,
                         shift, out, p_drop, seed_ptr, bn_sync_native_arg());
    EDA_CHECK_LAUNCH();
    return 0;
  }
  if (training) {
    EDA_CHECK_ARG(ws, "workspace required");
    // ws (2*C + 1 doubles) is ZERO on entry and is left zero (see eda_hip.h): the statistics
    // kernel's last block finalises and cleans up
    int nblocks = 2048;
    long rpb = (R + nblocks - 1) / nblocks;
    if (rpb < 64) rpb = 64;
    nblocks = (int)((R + rpb - 1) / rpb);
    BnFinalize fin = {gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift,
                      reinterpret_cast<unsigned *>(ws + 2 * C), bn_sync_native_arg()};
    hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(nblocks), dim3(CL_THREADS), 0, stream, z, R, C, rpb, ws,
                       ws + C, fin);
  } else {
    EDA_CHECK_ARG(running_mean && running_var, "eval mode needs running statistics");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, stream,
                       running_mean, running_var, gamma, beta, eps, C, mean, rstd, scale, shift);
  }
  EDA_CHECK_LAUNCH();
  if (pool > 1) {
    EDA_CHECK_ARG(argmax, "argmax buffer required when pooling");
    hipLaunchKernelGGL(bn_relu_pool_kernel, dim3(grid_for(R / pool * (C / 4))), dim3(CL_THREADS), 0, stream,
                       z, R / pool, pool, C, scale, shift, out, argmax);
  } else {
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(grid_for(R * C / 4)), dim3(CL_THREADS), 0, stream, z,
                       R * C / 4, C, scale, shift, out);
  }
  EDA_CHECK_LAUNCH();
  return 0;
}

// Backward of the above.  dout: (R,C), or (R/pool,C) when pool > 1.  Outputs dz (R,C),
// dgamma, dbeta (C).  ws: 2*C doubles (zeroed here).
extern "C" int eda_bn_relu_bwd_f32(const float *dout, const unsigned char *argmax, const float *z,
                                   long R, int C, int pool, const float *gamma, const float *mean,
                                   const float *rstd, const float *scale, const float *shift,
                                   int training, double *ws, float *dgamma, float *dbeta, float *dz,
                                   float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                                   void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(R >= 0 && C > 0 && pool >= 1 && C % 4 == 0 && C <= 1024, "bad dimension");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_ptr), "bad dropout arguments");
  if (p_drop > 0.f && !(pool == 1 && R <= SMALL_ROWS)) {
    eda_set_error("bn_relu: fused dropout is built for pool == 1 and R <= eda_bn_relu_dropout_max_rows() only");
    return EDA_ERR_UNSUPPORTED;
  }
  EDA_CHECK_ARG(dgamma && dbeta, "null pointer");
  if (R == 0) {
    const int z1 = eda_zero_async(dgamma, sizeof(float) * C, stream);
    return z1 ? z1 : eda_zero_async(dbeta, sizeof(float) * C, stream);
  }
  EDA_CHECK_ARG(dout && z && gamma && mean && rstd && scale && shift && ws && dz, "null pointer");
  EDA_CHECK_ARG(pool == 1 || argmax, "argmax required when pooling");
  if (pool == 1 && R <= SMALL_ROWS) {
    BnGrp G;
    memset(&G, 0, sizeof(G));
    G.gamma[0] = gamma; G.salt[0] = salt; G.cpg = C;
    const int cq_env = small_cq();
    if (cq_env == 4 && C % 16 == 0)
      hipLaunchKernelGGL(bn_relu_small_bwd_kernel<4>, dim3(C / 16), dim3(SM_THREADS), 0, stream, dout, z, (int)R,
                         C, G, mean, rstd, scale, shift, training, ws, ws + C, dgamma, dbeta, dz, p_drop,
                         seed_ptr, bn_sync_native_arg());
    else if (cq_env == 2 && C % 8 == 0)
      hipLaunchKernelGGL(bn_relu_small_bwd_kernel<2>, dim3(C / 8), dim3(SM_THREADS), 0, stream, dout, z, (int)R,
                         C, G, mean, rstd, scale, shift, training, ws, ws + C, dgamma, dbeta, dz, p_drop,
                         seed_ptr, bn_sync_native_arg());
    else
      hipLaunchKernelGGL(bn_relu_small_bwd_kernel<1>, dim3(C / 4), dim3(SM_THREADS), 0, stream, dout, z, (int)R,
                         C, G, mean, rstd, scale, shift, training, ws, ws + C, dgamma, dbeta, dz, p_drop,
                         seed_ptr, bn_sync_native_arg());
    EDA_CHECK_LAUNCH();
    return 0;
  }
  // (backward: ws is zeroed here per call.  A self-cleaning scheme like the forward's was measured
  // SLOWER: the apply kernel has ~8000 workgroups and their tickets serialise on one address, +16 us)
  { const int zrc__ = eda_zero_async(ws, sizeof(double) * 2 * C, stream); if (zrc__) return zrc__; }
  int nblocks = 1024;
  long rpb = (R + nblocks - 1) / nblocks;
  if (rpb < 64) rpb = 64;
  rpb = (rpb + pool - 1) / pool * pool;           // whole groups per block
  nblocks = (int)((R + rpb - 1) / rpb);
  if (pool > 1)
    hipLaunchKernelGGL(bn_relu_bwd_stats_kernel<true>, dim3(nblocks), dim3(CL_THREADS), 0, stream, dout,
                       argmax, z, R, C, pool, rpb, mean, rstd, scale, shift, ws, ws + C);
  else
    hipLaunchKernelGGL(bn_relu_bwd_stats_vec_kernel, dim3(nblocks), dim3(CL_THREADS), 0, stream, dout, z,
                       R, C, rpb, mean, rstd, scale, shift, ws, ws + C);
  EDA_CHECK_LAUNCH();
  float inv_n = 1.f / (float)R, gscale = 1.f;
  if (training && g_sync_native && eda_peer_active()) {     // global-batch sums; d(gamma), d(beta) = global / world
    const int rc = eda_peer_bn_hook(nullptr, ws, 2L * C, stream_);
    if (rc) return rc;
    inv_n = 1.f / ((float)R * g_sync_world); gscale = 1.f / g_sync_world;
  }
  const int apply_grid = grid_for(R * (C / 4));
  if (pool > 1)
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<true>, dim3(apply_grid), dim3(CL_THREADS), 0, stream,
                       dout, argmax, z, R, C, pool, mean, rstd, scale, shift, gamma, ws, ws + C, training,
                       dgamma, dbeta, dz, inv_n, gscale);
  else
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<false>, dim3(apply_grid), dim3(CL_THREADS), 0, stream,
                       dout, argmax, z, R, C, pool, mean, rstd, scale, shift, gamma, ws, ws + C, training,
                       dgamma, dbeta, dz, inv_n, gscale);
  EDA_CHECK_LAUNCH();
  return 0;
}


// =================================================================================================
// Fused SharedMLP pipeline: QueryAndGroup (or plain rows) -> L x [conv1x1 -> BatchNorm -> ReLU] ->
// max-pool over the nsample rows of a centre (or plain rows out).  Reference chain:
// pointnet2/pointnet2_utils.py:317-376 -> pointnet2/pytorch_utils.py:67-120 ->
// pointnet2/pointnet2_modules.py:251-257 (SA) / :407-414 (FP).
//
// Per layer ONE launch of csrc/gemm.hip's row GEMM: the neighbourhood gather (layer 0) or the
// previous layer's BatchNorm+ReLU (layers >= 1) happens while the row operand is staged into LDS,
// the BatchNorm statistics of the layer's own output come out of the accumulators (E_STATS, last
// workgroup finalises).  Neither the grouped tensor nor any activated tensor is written to HBM;
// what is kept is the pre-activation z_l of every layer (the backward needs it densely: the
// train-mode BatchNorm backward has a mean and a variance term on every row).
// Backward per layer: weight gradient with the same prologue (wgrad.hip), input gradient with the
// ReLU mask of the layer below and that layer's two BatchNorm reductions in the epilogue (E_MASK),
// then one element-wise pass dz = A*gy + B*z + D; the first layer's input gradient is scattered
// straight into d(features) (E_SCATTER).
namespace {
// wt (cols, rows) = w (rows, ld) [:, c0 : c0 + cols]^T -- the layers' weights are a few hundred KB: transposing
// them lets the input-gradient GEMMs run in the K-minor ("NT") form, 25-45 % faster than reading the
// weight along its rows (profiles/r02d_sa_timeline.md)
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float *__restrict__ w, int rows, int ld, int c0,
                                                               int cols, float *__restrict__ wt) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;      // bx: column block of w, by: row block
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int row = by + r, col = bx + threadIdx.x;
    tile[r][threadIdx.x] = (row < rows && col < cols) ? w[(long)row * ld + c0 + col] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int col = bx + r, row = by + threadIdx.x;
    if (col < cols && row < rows) wt[(long)col * rows + row] = tile[threadIdx.x][r];
  }
}

struct MlpGeom {
  const float *x; long ldx;
  const float *xyz, *new_xyz, *feats; const int *idx;
  int b, n, m, ns, c_feat; float inv_radius;
  bool gather;
};

void mlp_row_operand(GemmArgs &a, const MlpGeom &g, int l, int c_in, const float *z_prev, const float *stats_prev) {
  if (l == 0) {
    if (g.gather) {
      a.xmode = X_GATHER;
      a.xyz = g.xyz; a.new_xyz = g.new_xyz; a.feats = g.feats; a.idx = g.idx;
      a.n_pts = g.n; a.m = g.m; a.ns = g.ns; a.c_feat = g.c_feat; a.inv_radius = g.inv_radius;
      a.K = 4 + g.c_feat;
    } else {
      a.xmode = X_PLAIN; a.x = g.x; a.ldx = g.ldx; a.K = c_in;
    }
  } else {
    a.xmode = X_BNRELU; a.x = z_prev; a.ldx = c_in; a.K = c_in;
    a.in_scale = stats_prev + 2 * c_in; a.in_shift = stats_prev + 3 * c_in;
  }
}

int check_mlp(int nlayers, const int *channels, long R, int pool, const MlpGeom &g) {
  EDA_CHECK_ARG(nlayers >= 1 && nlayers <= 8 && channels, "1..8 layers");
  EDA_CHECK_ARG(R >= 0 && pool >= 1 && pool <= 255 && R % pool == 0, "rows must be a multiple of the pooling width");
  for (int l = 1; l <= nlayers; ++l)
    EDA_CHECK_ARG(channels[l] > 0 && channels[l] % 4 == 0 && channels[l] <= 1024, "layer widths must be multiples of 4 (<= 1024)");
  if (g.gather) {
    EDA_CHECK_ARG(g.xyz && g.new_xyz && g.idx && (g.feats || g.c_feat == 0), "null pointer");
    EDA_CHECK_ARG(channels[0] == 3 + g.c_feat, "channels[0] must be 3 + c_feat");
    EDA_CHECK_ARG(R == (long)g.b * g.m * g.ns, "row count must be b*m*ns");
    EDA_CHECK_ARG(pool == 1 || pool == g.ns, "pooling width must be nsample");
  } else {
    EDA_CHECK_ARG(g.x && g.ldx >= channels[0] && channels[0] > 0, "bad plain-row input");
  }
  return 0;
}
}  // namespace

namespace {
// Column constants of the pooled last layer's BatchNorm+ReLU backward (the arithmetic of bn_relu_bwd_apply_kernel,
// train mode) for the kernels that form dz while staging instead of reading it: consts = {sc, sh, ka, kb, kd} (5 x C),
// and the layer's d(gamma), d(beta).
__global__ void bn_bwd_consts_kernel(const float *__restrict__ mean, const float *__restrict__ rstd,
                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                     const float *__restrict__ gamma, const double *__restrict__ s1,
                                     const double *__restrict__ s2, float invR, float gscale, int C,
                                     float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ consts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)s1[c] * gscale;
  dgamma[c] = (float)s2[c] * gscale;
  const float gr = gamma[c] * rstd[c];
  const float t2 = rstd[c] * (float)s2[c] * invR;
  consts[c] = scale[c];
  consts[C + c] = shift[c];
  consts[2 * C + c] = gr;
  consts[3 * C + c] = -gr * t2;
  consts[4 * C + c] = gr * (mean[c] * t2 - (float)s1[c] * invR);
}
}  // namespace

// ---- global-batch BatchNorm statistics (the reference's SyncBatchNorm, main_utils.py:336-338) inside the fused calls ----
// A fused SA / FP call produces a layer's column sums in a kernel epilogue and consumes them in the next kernel's
// prologue; with more than one rank the sums have to be added over the ranks in between.  The host registers a hook
// (eda_set_bn_sync): the fused calls then leave the sums un-finalised, call the hook on the packed fp64 vector
// [sum | sum of squares] (forward) / [sum gy | sum gy (z - mean)] (backward) -- ONE collective per layer and direction,
// enqueued on the same stream, no host synchronisation -- and finalise with the GLOBAL row count R * world (every rank
// holds the same number of rows: scenes x positions are fixed per GPU).  d(gamma), d(beta) are the global sums / world,
// i.e. what the gradient all-reduce's mean makes of the ranks' local sums.
namespace {

__global__ void bn_sync_finalize_kernel(double *__restrict__ sum, double *__restrict__ sumsq, double count, int C,
                                        const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                        float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                                        float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                        float *__restrict__ scale, float *__restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = sum[c] / count;
  double var = sumsq[c] / count - mean * mean;
  sum[c] = 0.0; sumsq[c] = 0.0;                      // re-armed for the next statistics launch
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  mean_out[c] = meanf;
  rstd_out[c] = rstd;
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - meanf * sc;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}
}  // namespace

extern "C" int eda_set_bn_sync(eda_bn_sync_fn fn, void *user, int world) {
  EDA_CHECK_ARG(world >= 1, "world size must be >= 1");
  g_sync_fn = fn; g_sync_user = user; g_sync_world = fn ? world : 1;
  g_sync_native = false;
  return 0;
}

extern "C" int eda_peer_bn_hook(void *user, double *buf, long n, void *stream);     // peer.hip
extern "C" int eda_set_bn_sync_native(int world) {
  EDA_CHECK_ARG(world >= 0 && world <= PEER_MAXW, "0 (off) .. 8 ranks");
  if (world == 0) { g_sync_fn = nullptr; g_sync_user = nullptr; g_sync_world = 1; g_sync_native = false; return 0; }
  const EdaPeer *P = eda_peer_active();
  EDA_CHECK_ARG(P && P->world == world, "eda_peer_connect() with the same world size first");
  g_sync_fn = eda_peer_bn_hook; g_sync_user = nullptr; g_sync_world = world; g_sync_native = true;
  return 0;
}

extern "C" int eda_sa_fused_fwd_f32(const float *x, long ldx, const float *xyz, const float *new_xyz,
                                    const float *feats_cl, const int *idx, int b, int n, int m, int ns, int c_feat,
                                    float radius, int normalize_xyz, long R, int nlayers, const int *channels,
                                    const float *const *weight, const float *const *gamma, const float *const *beta,
                                    float *const *running_mean, float *const *running_var, float eps, float momentum,
                                    int training, int pool, float *const *z, float *const *stats, double *ws,
                                    float *out, unsigned char *argmax, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MlpGeom g = {x, ldx, xyz, new_xyz, feats_cl, idx, b, n, m, ns, c_feat, normalize_xyz ? 1.0f / radius : 1.0f, idx != nullptr};
  { const int rc = check_mlp(nlayers, channels, R, pool, g); if (rc) return rc; }
  if (R == 0) return 0;
  EDA_CHECK_ARG(weight && gamma && beta && z && stats && out, "null pointer");
  EDA_CHECK_ARG(!training || ws, "workspace required in training mode");
  EDA_CHECK_ARG(pool == 1 || argmax, "argmax buffer required when pooling");
  for (int l = 0; l < nlayers; ++l) {
    const int cin = channels[l], cout = channels[l + 1];
    float *st = stats[l];
    EDA_CHECK_ARG(weight[l] && gamma[l] && beta[l] && z[l] && st, "null pointer");
    if (!training) {
      EDA_CHECK_ARG(running_mean && running_var && running_mean[l] && running_var[l], "eval mode needs running statistics");
      hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((cout + 255) / 256), dim3(256), 0, stream, running_mean[l],
                         running_var[l], gamma[l], beta[l], eps, cout, st, st + cout, st + 2 * cout, st + 3 * cout);
      EDA_CHECK_LAUNCH();
    }
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.R = R;
    mlp_row_operand(a, g, l, cin, l ? z[l - 1] : nullptr, l ? stats[l - 1] : nullptr);
    a.w = weight[l]; a.ldw = cin; a.N = cout;
    a.y = z[l]; a.ldy = cout;
    if (training) {
      a.epi = E_STATS;
      a.sum = ws; a.sumsq = ws + cout; a.ticket = reinterpret_cast<unsigned *>(ws + 2 * cout);
      a.gamma = gamma[l]; a.beta = beta[l]; a.eps = eps; a.momentum = momentum;
      a.running_mean = running_mean ? running_mean[l] : nullptr;
      a.running_var = running_var ? running_var[l] : nullptr;
      a.mean_out = st; a.rstd_out = st + cout; a.scale_out = st + 2 * cout; a.shift_out = st + 3 * cout;
      a.defer_finalize = bn_sync_on() ? 1 : 0;
    } else {
      a.epi = E_PLAIN;
    }
    const int rc = eda_gemm_launch(a, W_NT, stream);
    if (rc) return rc;
    if (training && bn_sync_on()) {
      const int src = g_sync_fn(g_sync_user, ws, 2L * cout, stream_);
      if (src) { eda_set_error("eda_sa_fused_fwd_f32: the BatchNorm statistics hook failed (%d)", src); return EDA_ERR_UNSUPPORTED; }
      hipLaunchKernelGGL(bn_sync_finalize_kernel, dim3((cout + 255) / 256), dim3(256), 0, stream, ws, ws + cout,
                         (double)R * g_sync_world, cout, gamma[l], beta[l], eps, momentum,
                         running_mean ? running_mean[l] : nullptr, running_var ? running_var[l] : nullptr, st, st + cout,
                         st + 2 * cout, st + 3 * cout);
      EDA_CHECK_LAUNCH();
    }
  }
  const int C = channels[nlayers];
  const float *st = stats[nlayers - 1];
  if (pool > 1)
    hipLaunchKernelGGL(bn_relu_pool_kernel, dim3(grid_for(R / pool * (C / 4))), dim3(CL_THREADS), 0, stream,
                       z[nlayers - 1], R / pool, pool, C, st + 2 * C, st + 3 * C, out, argmax);
  else
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(grid_for(R * C / 4)), dim3(CL_THREADS), 0, stream, z[nlayers - 1],
                       R * C / 4, C, st + 2 * C, st + 3 * C, out);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t eda_sa_fused_bwd_workspace_bytes(long R, int nlayers, const int *channels, int gather) {
  if (R <= 0 || nlayers < 1 || !channels) return 0;
  size_t slabs = 0;
  int cmax = 0;
  for (int l = 0; l < nlayers; ++l) {
    const int ncols = (l == 0 && gather) ? channels[0] + 1 : channels[l];
    const size_t sbytes = eda_wgrad_x_workspace_bytes(R, channels[l + 1], ncols);
    if (sbytes > slabs) slabs = sbytes;
    if (channels[l + 1] > cmax) cmax = channels[l + 1];
  }
  size_t wt = 0;                                   // transposed weight of one layer (floats), 16-byte rounded
  for (int l = 0; l < nlayers; ++l) {
    const size_t n = (size_t)channels[l] * channels[l + 1];
    if (n > wt) wt = n;
  }
  // red (one region of 2 x cmax doubles per layer, all zeroed by ONE launch) | transposed weight | per layer: constants of a BatchNorm backward formed in its consumers (5 x cmax floats) | slabs
  return sizeof(double) * 2 * (size_t)cmax * nlayers + sizeof(float) * ((wt + 3) / 4 * 4) + sizeof(float) * 5 * (size_t)((cmax + 3) / 4 * 4) * nlayers + slabs;
}

// weight_t: optional per-layer pointers to W^T ((cin, cout) row-major, contiguous), e.g. from the caller's W^T shadow
// (eda_amd/wt_shadow.py); a NULL array or entry makes the call transpose that layer's weight itself.
static int sa_fused_bwd_impl(const float *dout, const unsigned char *argmax, const float *x, long ldx,
                             const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx,
                             int b, int n, int m, int ns, int c_feat, float radius, int normalize_xyz, long R,
                             int nlayers, const int *channels, const float *const *weight,
                             const float *const *weight_t,
                             const float *const *gamma, const float *const *z, const float *const *stats,
                             int training, int pool, float *scratch_a, float *scratch_b, void *ws_,
                             size_t ws_bytes, float *const *dW, float *const *dgamma, float *const *dbeta,
                             float *dx, long lddx, float *dfeats_cl, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MlpGeom g = {x, ldx, xyz, new_xyz, feats_cl, idx, b, n, m, ns, c_feat, normalize_xyz ? 1.0f / radius : 1.0f, idx != nullptr};
  { const int rc = check_mlp(nlayers, channels, R, pool, g); if (rc) return rc; }
  EDA_CHECK_ARG(weight && gamma && z && stats && dW && dgamma && dbeta, "null pointer");
  if (g.gather && dfeats_cl) {
    const int zrc = eda_zero_async(dfeats_cl, sizeof(float) * (size_t)b * n * c_feat, stream);
    if (zrc) return zrc;
  }
  if (R == 0) {
    for (int l = 0; l < nlayers; ++l) {
      int rc = eda_zero_async(dW[l], sizeof(float) * (size_t)channels[l] * channels[l + 1], stream);
      if (!rc) rc = eda_zero_async(dgamma[l], sizeof(float) * channels[l + 1], stream);
      if (!rc) rc = eda_zero_async(dbeta[l], sizeof(float) * channels[l + 1], stream);
      if (rc) return rc;
    }
    return 0;
  }
  EDA_CHECK_ARG(dout && scratch_a && scratch_b && ws_, "null pointer");
  EDA_CHECK_ARG(ws_bytes >= eda_sa_fused_bwd_workspace_bytes(R, nlayers, channels, g.gather) &&
                    (reinterpret_cast<uintptr_t>(ws_) & 15u) == 0, "workspace too small or misaligned");
  EDA_CHECK_ARG(pool == 1 || argmax, "argmax required when pooling");
  // global-batch statistics (eda_set_bn_sync): the reductions are all-reduced before they are used
  const bool sync = training && bn_sync_on();
  const float inv_n = 1.f / ((float)R * (sync ? g_sync_world : 1));
  const float gscale = sync ? 1.f / g_sync_world : 1.f;
  int cmax = 0;
  for (int l = 1; l <= nlayers; ++l) if (channels[l] > cmax) cmax = channels[l];
  double *red_all = reinterpret_cast<double *>(ws_);      // region l: the two BatchNorm-backward reductions of layer l
  size_t wt_floats = 0;
  for (int l = 0; l < nlayers; ++l) {
    const size_t nn = (size_t)channels[l] * channels[l + 1];
    if (nn > wt_floats) wt_floats = nn;
  }
  wt_floats = (wt_floats + 3) / 4 * 4;
  { const int zrc = eda_zero_async(red_all, sizeof(double) * 2 * (size_t)cmax * nlayers, stream); if (zrc) return zrc; }
  float *wt = reinterpret_cast<float *>(red_all + 2 * (size_t)cmax * nlayers);
  const size_t const_floats = 5 * (size_t)((cmax + 3) / 4 * 4);
  float *consts_all = wt + wt_floats;                     // layer l: consts_all + l * const_floats
  float *pool_consts = consts_all + const_floats * (size_t)(nlayers - 1);
  float *slabs = consts_all + const_floats * (size_t)nlayers;
  const size_t slab_bytes = ws_bytes - sizeof(double) * 2 * (size_t)cmax * nlayers - sizeof(float) * (wt_floats + const_floats * (size_t)nlayers);
  // One launch per layer (wgrad.hip, sa_layer_bwd_kernel: weight gradient + masked input gradient + its BatchNorm-backward sums
  // from one pass over dz) where the layer's dW is one tile; and a non-pooled layer's dz = ka*g + kb*z + kd is formed by
  // its consumer(s) while staging instead of by bn_relu_bwd_apply_kernel whenever all of them can (`pending`).
  // (the layer launches address rows with 32-bit element offsets: beyond 2^31 / 128 rows per call the separate kernels run)
  bool layer_fuse = training != 0 && R * 128 < 0x7fffffffL;
  if (eda_knob(EDA_K_SA_LAYER_FUSE) == 0) layer_fuse = false;
  const bool need_dx0 = g.gather ? (dfeats_cl && c_feat > 0) : dx != nullptr;
  // first layer of a gathering stack with 128 feature channels (SA2): weight gradient + scatter of d(features) in one launch
  const bool gl0 = layer_fuse && g.gather && need_dx0 && nlayers >= 2 && eda_wgrad_x_fuses_gather(channels[1], c_feat) &&
                   (reinterpret_cast<uintptr_t>(feats_cl) & 15u) == 0 && (long)(b + 1) * n * 128 < 0x7fffffffL;
  bool pending = false;
  // Pooled last layer in training mode: its dz = ka*d + kb*z + kd need not be written and read back -- the two kernels
  // that consume it (weight gradient, input gradient) form it while staging z, when the input gradient is a launch
  // of the streaming kernels (gemm.hip X_BNBWDPOOL); otherwise bn_relu_bwd_apply_kernel<true> materialises it.
  bool fuse_pool = false;
  if (training && pool > 1 && pool % 16 == 0 && nlayers >= 2) {
    const int l = nlayers - 1, cin = channels[l], cout = channels[l + 1];
    GemmArgs t;
    memset(&t, 0, sizeof(t));
    t.xmode = X_BNBWDPOOL; t.epi = E_MASK; t.x = z[l]; t.ldx = cout; t.R = R; t.K = cout;
    t.w = wt; t.ldw = cout; t.N = cin; t.y = scratch_b; t.ldy = cin; t.zm = z[l - 1]; t.ldzm = cin;
    t.bb_argmax = argmax; t.bb_dout = dout; t.bb_consts = pool_consts; t.bb_pool = pool;
    fuse_pool = eda_knob(EDA_K_SA_BNBWD_FUSE) != 0 && cout % 64 == 0 && eda_gemm_stream_takes(t, W_NT);
  }

  // ---- last layer: BatchNorm+ReLU(+pool) backward from d(out) -> dz in scratch_a
  {
    const int l = nlayers - 1, C = channels[nlayers];
    const float *st = stats[l];
    double *red = red_all + 2 * (size_t)cmax * l;
    int nblocks = 1024;
    long rpb = (R + nblocks - 1) / nblocks;
    if (rpb < 64) rpb = 64;
    rpb = (rpb + pool - 1) / pool * pool;
    nblocks = (int)((R + rpb - 1) / rpb);
    if (pool > 1)
      hipLaunchKernelGGL(bn_relu_bwd_stats_kernel<true>, dim3(nblocks), dim3(CL_THREADS), 0, stream, dout, argmax, z[l],
                         R, C, pool, rpb, st, st + C, st + 2 * C, st + 3 * C, red, red + C);
    else
      hipLaunchKernelGGL(bn_relu_bwd_stats_vec_kernel, dim3(nblocks), dim3(CL_THREADS), 0, stream, dout, z[l], R, C, rpb,
                         st, st + C, st + 2 * C, st + 3 * C, red, red + C);
    EDA_CHECK_LAUNCH();
    if (sync) {
      const int src = g_sync_fn(g_sync_user, red, 2L * C, stream_);
      if (src) { eda_set_error("eda_sa_fused_bwd_f32: the BatchNorm statistics hook failed (%d)", src); return EDA_ERR_UNSUPPORTED; }
    }
    const int apply_grid = grid_for(R * (C / 4));
    if (fuse_pool)
      hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, st, st + C, st + 2 * C,
                         st + 3 * C, gamma[l], red, red + C, inv_n, gscale, C, dgamma[l], dbeta[l], pool_consts);
    else if (pool > 1)
      hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<true>, dim3(apply_grid), dim3(CL_THREADS), 0, stream, dout, argmax,
                         z[l], R, C, pool, st, st + C, st + 2 * C, st + 3 * C, gamma[l], red, red + C, training,
                         dgamma[l], dbeta[l], scratch_a, inv_n, gscale);
    else
      hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<false>, dim3(apply_grid), dim3(CL_THREADS), 0, stream, dout, argmax,
                         z[l], R, C, pool, st, st + C, st + 2 * C, st + 3 * C, gamma[l], red, red + C, training,
                         dgamma[l], dbeta[l], scratch_a, inv_n, gscale);
    EDA_CHECK_LAUNCH();
  }
  float *cur = scratch_a, *other = scratch_b;
  for (int l = nlayers - 1; l >= 0; --l) {
    const int cin = channels[l], cout = channels[l + 1];
    const bool fl = layer_fuse && l >= 1 && eda_wgrad_x_fuses_dx(cout, cin);
    // ---- weight gradient: dW_l = dz_l^T (row operand of the forward GEMM, recomputed while staging)
    {
      WgradXArgs wa;
      memset(&wa, 0, sizeof(wa));
      wa.dy = cur; wa.ld_dy = cout; wa.R = R; wa.M = cout;
      if (l == 0 && g.gather) {
        wa.xmode = X_GATHER;
        wa.xyz = xyz; wa.new_xyz = new_xyz; wa.feats = feats_cl; wa.idx = idx;
        wa.n_pts = n; wa.m = m; wa.ns = ns; wa.c_feat = c_feat; wa.inv_radius = g.inv_radius;
        wa.N = 4 + c_feat;
      } else if (l == 0) {
        wa.xmode = X_PLAIN; wa.x = x; wa.ld_x = ldx; wa.N = cin;
      } else {
        wa.xmode = X_BNRELU; wa.x = z[l - 1]; wa.ld_x = cin; wa.N = cin;
        wa.in_scale = stats[l - 1] + 2 * cin; wa.in_shift = stats[l - 1] + 3 * cin;
      }
      wa.dW = dW[l]; wa.ws = slabs; wa.ws_bytes = slab_bytes;
      if (fuse_pool && l == nlayers - 1) {
        wa.dy = nullptr; wa.dy_pool = pool; wa.dyz = z[l]; wa.dy_argmax = argmax; wa.dy_dout = dout; wa.dy_consts = pool_consts;
      } else if (pending) {
        wa.dy_bn = 1; wa.dyz = z[l]; wa.dy_consts = consts_all + const_floats * (size_t)l;
      }
      if (l == 0 && gl0 && pending) {
        wa.dx_w = weight[0]; wa.dx_ldw = 3 + c_feat; wa.dx_out = other; wa.dx_scatter = dfeats_cl;
      }
      if (fl) {
        const float *st = stats[l - 1];
        double *red = red_all + 2 * (size_t)cmax * (l - 1);
        wa.dx_w = weight[l]; wa.dx_ldw = cin; wa.dx_out = other;
        wa.dx_mean = st; wa.dx_rstd = st + cin; wa.dx_s1 = red; wa.dx_s2 = red + cin;
      }
      const int rc = eda_wgrad_x_launch(wa, stream);
      if (rc) return rc;
    }
    // ---- input gradient
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.xmode = X_PLAIN; a.x = cur; a.ldx = cout; a.R = R; a.K = cout;
    if (fuse_pool && l == nlayers - 1) {
      a.xmode = X_BNBWDPOOL; a.x = z[l];
      a.bb_argmax = argmax; a.bb_dout = dout; a.bb_consts = pool_consts; a.bb_pool = pool;
    }
    // dX = dz W as an NT product with W^T (output columns x cout), transposed into the workspace
    const int wcols = (l == 0 && g.gather) ? c_feat : cin, wc0 = (l == 0 && g.gather) ? 3 : 0;
    const bool need_dx = !fl && !(l == 0 && gl0 && pending) && (l > 0 || need_dx0);   // (the input gradient rides in the weight-gradient launch)
    const float *wt_l = wt;
    if (need_dx && weight_t && weight_t[l] && (reinterpret_cast<uintptr_t>(weight_t[l]) & 15u) == 0 && cout % 4 == 0) {
      wt_l = weight_t[l] + (size_t)wc0 * cout;              // rows wc0.. of W^T: the feature columns of a gather layer
    } else if (need_dx) {
      hipLaunchKernelGGL(weight_transpose_kernel, dim3((wcols + 31) / 32, (cout + 31) / 32), dim3(32, 8), 0, stream,
                         weight[l], cout, cin, wc0, wcols, wt);
      EDA_CHECK_LAUNCH();
    }
    a.w = wt_l; a.ldw = cout;
    if (l > 0) {
      const float *st = stats[l - 1];
      double *red = red_all + 2 * (size_t)cmax * (l - 1);
      if (!fl) {
        a.N = cin; a.y = other; a.ldy = cin;
        a.epi = E_MASK;
        a.zm = z[l - 1]; a.ldzm = cin;
        a.m_mean = st; a.m_rstd = st + cin; a.m_scale = st + 2 * cin; a.m_shift = st + 3 * cin;
        a.s1 = red; a.s2 = red + cin;
        const int rc = eda_gemm_launch(a, W_NT, stream);
        if (rc) return rc;
      }
      if (sync) {
        const int src = g_sync_fn(g_sync_user, red, 2L * cin, stream_);
        if (src) { eda_set_error("eda_sa_fused_bwd_f32: the BatchNorm statistics hook failed (%d)", src); return EDA_ERR_UNSUPPORTED; }
      }
      // dz_{l-1} = A*gy + B*z + D: in place (gy is already masked: the kernel's mask is idempotent), or -- when every
      // consumer of dz_{l-1} forms it while staging -- only the five column constants (and d(gamma), d(beta))
      const bool next_fl = layer_fuse && l - 1 >= 1 && eda_wgrad_x_fuses_dx(cin, channels[l - 1]);
      pending = layer_fuse && cin % 4 == 0 && (next_fl || (l - 1 == 0 && (!need_dx0 || gl0)));
      if (pending)
        hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((cin + 255) / 256), dim3(256), 0, stream, st, st + cin, st + 2 * cin,
                           st + 3 * cin, gamma[l - 1], red, red + cin, inv_n, gscale, cin, dgamma[l - 1], dbeta[l - 1],
                           consts_all + const_floats * (size_t)(l - 1));
      else
        hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<false>, dim3(grid_for(R * (cin / 4))), dim3(CL_THREADS), 0, stream,
                           other, nullptr, z[l - 1], R, cin, 1, st, st + cin, st + 2 * cin, st + 3 * cin, gamma[l - 1],
                           red, red + cin, training, dgamma[l - 1], dbeta[l - 1], other, inv_n, gscale);
      EDA_CHECK_LAUNCH();
      float *t = cur; cur = other; other = t;
    } else if (g.gather) {
      if (dfeats_cl && c_feat > 0 && need_dx && eda_deterministic() && (long)m * ns < 0x7fffffffL) {
        // ordered form: the input-gradient rows densely into the free scratch, then per-point sums in row order
        a.N = c_feat; a.epi = E_PLAIN; a.y = other; a.ldy = c_feat;
        const int rc = eda_gemm_launch(a, W_NT, stream);
        if (rc) return rc;
        const long rows = (long)m * ns;
        EdaDetScatter d = {idx, nullptr, rows, (int)rows, 1, other, rows * c_feat, (long)c_feat, 1,
                           dfeats_cl, (long)n * c_feat, (long)c_feat, 1, b, n, c_feat};
        const int rc2 = eda_det_scatter_launch(d, stream);
        if (rc2) return rc2;
      } else if (dfeats_cl && c_feat > 0 && need_dx) {
        a.N = c_feat; a.epi = E_SCATTER;                       // (wt holds the feature columns of the (C1, 3 + c_feat) weight)
        a.idx = idx; a.n_pts = n; a.m = m; a.ns = ns; a.c_feat = c_feat; a.dfeats = dfeats_cl;
        const int rc = eda_gemm_launch(a, W_NT, stream);
        if (rc) return rc;
      }
    } else if (dx) {
      a.N = cin; a.y = dx; a.ldy = lddx; a.epi = E_PLAIN;
      const int rc = eda_gemm_launch(a, W_NT, stream);
      if (rc) return rc;
    }
  }
  return 0;
}


extern "C" int eda_sa_fused_bwd_f32(const float *dout, const unsigned char *argmax, const float *x, long ldx,
                                    const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx,
                                    int b, int n, int m, int ns, int c_feat, float radius, int normalize_xyz, long R,
                                    int nlayers, const int *channels, const float *const *weight,
                                    const float *const *gamma, const float *const *z, const float *const *stats,
                                    int training, int pool, float *scratch_a, float *scratch_b, void *ws_,
                                    size_t ws_bytes, float *const *dW, float *const *dgamma, float *const *dbeta,
                                    float *dx, long lddx, float *dfeats_cl, void *stream_) {
  return sa_fused_bwd_impl(dout, argmax, x, ldx, xyz, new_xyz, feats_cl, idx, b, n, m, ns, c_feat, radius, normalize_xyz, R,
                           nlayers, channels, weight, nullptr, gamma, z, stats, training, pool, scratch_a, scratch_b, ws_,
                           ws_bytes, dW, dgamma, dbeta, dx, lddx, dfeats_cl, stream_);
}

extern "C" int eda_sa_fused_bwd_wt_f32(const float *dout, const unsigned char *argmax, const float *x, long ldx,
                                       const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx,
                                       int b, int n, int m, int ns, int c_feat, float radius, int normalize_xyz, long R,
                                       int nlayers, const int *channels, const float *const *weight,
                                       const float *const *weight_t,
                                       const float *const *gamma, const float *const *z, const float *const *stats,
                                       int training, int pool, float *scratch_a, float *scratch_b, void *ws_,
                                       size_t ws_bytes, float *const *dW, float *const *dgamma, float *const *dbeta,
                                       float *dx, long lddx, float *dfeats_cl, void *stream_) {
  return sa_fused_bwd_impl(dout, argmax, x, ldx, xyz, new_xyz, feats_cl, idx, b, n, m, ns, c_feat, radius, normalize_xyz, R,
                           nlayers, channels, weight, weight_t, gamma, z, stats, training, pool, scratch_a, scratch_b, ws_,
                           ws_bytes, dW, dgamma, dbeta, dx, lddx, dfeats_cl, stream_);
}

// ---- sibling BatchNorm+ReLU(+Dropout) modules on column blocks of one (R, ngroups*cpg) matrix ------
// (the three ThreeLayerMLPs of a ClsAgnosticPredictHead, models/modules.py:111-178, run side by
// side: one launch instead of one per module).  Single-launch kernels only: R <= 4096, cpg % 16 == 0.
extern "C" int eda_bn_relu_grouped_fwd_f32(const float *z, long R, int ngroups, int cpg, const float *const *gamma,
                                           const float *const *beta, float *const *running_mean,
                                           float *const *running_var, float eps, float momentum, int training,
                                           float *mean, float *rstd, float *scale, float *shift, float *out,
                                           float p_drop, const unsigned long long *seed_ptr, const unsigned *salts,
                                           void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(ngroups >= 1 && ngroups <= 4 && cpg > 0 && cpg % 16 == 0, "1..4 groups of a multiple of 16 channels");
  EDA_CHECK_ARG(R >= 0 && R <= SMALL_ROWS, "row count beyond the single-launch kernels");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || (seed_ptr && salts)), "bad dropout arguments");
  if (R == 0) return 0;
  EDA_CHECK_ARG(z && gamma && beta && mean && rstd && scale && shift && out, "null pointer");
  BnGrp G;
  memset(&G, 0, sizeof(G));
  G.cpg = cpg;
  for (int g = 0; g < ngroups; ++g) {
    EDA_CHECK_ARG(gamma[g] && beta[g], "null pointer");
    G.gamma[g] = gamma[g]; G.beta[g] = beta[g];
    G.running_mean[g] = running_mean ? running_mean[g] : nullptr;
    G.running_var[g] = running_var ? running_var[g] : nullptr;
    EDA_CHECK_ARG(training || (G.running_mean[g] && G.running_var[g]), "eval mode needs running statistics");
    G.salt[g] = salts ? salts[g] : 0u;
  }
  const int C = ngroups * cpg;
  EDA_CHECK_ARG(!bn_sync_native_on() || C <= PEER_MAXG, "more channels than the peer slab has granules (PEER_MAXG)");
  hipLaunchKernelGGL(bn_relu_small_fwd_kernel<4>, dim3(C / 16), dim3(SM_THREADS), 0, stream, z, (int)R, C, G, eps,
                     momentum, training, mean, rstd, scale, shift, out, p_drop, seed_ptr, bn_sync_native_arg());
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_bn_relu_grouped_bwd_f32(const float *dout, const float *z, long R, int ngroups, int cpg,
                                           const float *const *gamma, const float *mean, const float *rstd,
                                           const float *scale, const float *shift, int training, float *dgamma,
                                           float *dbeta, float *dz, float p_drop, const unsigned long long *seed_ptr,
                                           const unsigned *salts, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(ngroups >= 1 && ngroups <= 4 && cpg > 0 && cpg % 16 == 0, "1..4 groups of a multiple of 16 channels");
  EDA_CHECK_ARG(R >= 0 && R <= SMALL_ROWS, "row count beyond the single-launch kernels");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || (seed_ptr && salts)), "bad dropout arguments");
  EDA_CHECK_ARG(dgamma && dbeta, "null pointer");
  const int C = ngroups * cpg;
  if (R == 0) {
    const int z1 = eda_zero_async(dgamma, sizeof(float) * C, stream);
    return z1 ? z1 : eda_zero_async(dbeta, sizeof(float) * C, stream);
  }
  EDA_CHECK_ARG(dout && z && gamma && mean && rstd && scale && shift && dz, "null pointer");
  BnGrp G;
  memset(&G, 0, sizeof(G));
  G.cpg = cpg;
  for (int g = 0; g < ngroups; ++g) { EDA_CHECK_ARG(gamma[g], "null pointer"); G.gamma[g] = gamma[g]; G.salt[g] = salts ? salts[g] : 0u; }
  EDA_CHECK_ARG(!bn_sync_native_on() || C <= PEER_MAXG, "more channels than the peer slab has granules (PEER_MAXG)");
  hipLaunchKernelGGL(bn_relu_small_bwd_kernel<4>, dim3(C / 16), dim3(SM_THREADS), 0, stream, dout, z, (int)R, C, G, mean,
                     rstd, scale, shift, training, nullptr, nullptr, dgamma, dbeta, dz, p_drop, seed_ptr, bn_sync_native_arg());
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_bn_relu_grouped_bwd_multi_f32(int nmat, const float *const *dout, const float *const *z, long R, int ngroups,
                                                 int cpg, const float *const *gamma, const float *const *stats, int training,
                                                 float *const *dgb, float *const *dz, float p_drop,
                                                 const unsigned long long *seed_ptr, const unsigned *salts, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(nmat >= 1 && nmat <= BN_MAXMAT, "1..8 matrices");
  EDA_CHECK_ARG(ngroups >= 1 && ngroups <= 4 && cpg > 0 && cpg % 16 == 0, "1..4 groups of a multiple of 16 channels");
  EDA_CHECK_ARG(R > 0 && R <= SMALL_ROWS, "row count beyond the single-launch kernels");
  EDA_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || (seed_ptr && salts)), "bad dropout arguments");
  EDA_CHECK_ARG(dout && z && gamma && stats && dgb && dz, "null pointer");
  const int C = ngroups * cpg;
  BnMulti M;
  memset(&M, 0, sizeof(M));
  for (int m = 0; m < nmat; ++m) {
    EDA_CHECK_ARG(dout[m] && z[m] && stats[m] && dgb[m] && dz[m], "null pointer");
    M.da[m] = dout[m]; M.z[m] = z[m]; M.stats[m] = stats[m]; M.dgb[m] = dgb[m]; M.dz[m] = dz[m];
    M.grp[m].cpg = cpg;
    for (int g = 0; g < ngroups; ++g) {
      EDA_CHECK_ARG(gamma[m * ngroups + g], "null pointer");
      M.grp[m].gamma[g] = gamma[m * ngroups + g];
      M.grp[m].salt[g] = salts ? salts[m * ngroups + g] : 0u;
    }
  }
  // granule index of the in-kernel statistics exchange = matrix * C + channel: it must stay inside one parity's region of
  // the slab (csrc/peer.h), or the stores land in the other parity / past the end of every peer's mapped slab
  EDA_CHECK_ARG(!bn_sync_native_on() || (long)nmat * C <= PEER_MAXG, "nmat * channels beyond the peer slab's granules (PEER_MAXG)");
  hipLaunchKernelGGL(bn_relu_small_bwd_multi_kernel<4>, dim3(C / 16, nmat), dim3(SM_THREADS), 0, stream, M, (int)R, C, training,
                     p_drop, seed_ptr, bn_sync_native_arg());
  EDA_CHECK_LAUNCH();
  return 0;
}
