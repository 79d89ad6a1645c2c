// optim.hip -- the tail of a training step on the FLAT parameter / gradient buffers (eda_amd/parallel.py FlatParams): the global-norm
// clip of main_utils.py:483-486 (torch.nn.utils.clip_grad_norm_) and the AdamW update of main_utils.py:277-305 (torch.optim.AdamW,
// three learning-rate groups) as TWO launches:
//   eda_grad_sumsq_f32    sum of squares of the flat gradient (fixed-order partials, the last workgroup adds them in slice order:
//                         bit-reproducible) -> its square root; the same last workgroup bumps the groups' step counters and copies the
//                         learning rates from a pinned host array (so a scheduler can change them between replays of a captured graph)
//   eda_adamw_flat_f32    one pass over the buffer: g * clip coefficient (never written back), decoupled weight decay, both moments,
//                         bias corrections from the device-side step counters -- the arithmetic of torch's fused AdamW kernel
//                         (FusedAdamMathFunctor, ADAM_MODE = ADAMW, amsgrad off)
// torch ran: vector_norm, add, reciprocal, mul, clamp, mul (over the whole gradient), 2 x _foreach_add (steps), 2 fused Adam launches.
#include "eda_common.h"

namespace {

constexpr int OPT_MAXSEG = 4;
struct OptSegs {
  long lo[OPT_MAXSEG], hi[OPT_MAXSEG];          // element ranges of the flat buffer (lo: a multiple of 4)
  float *exp_avg[OPT_MAXSEG], *exp_avg_sq[OPT_MAXSEG];
  float *step[OPT_MAXSEG];                      // one float each (torch's capturable step tensors)
  double bd1[OPT_MAXSEG], bd2[OPT_MAXSEG];      // the betas as the caller's doubles (bias corrections)
  float beta1[OPT_MAXSEG], beta2[OPT_MAXSEG], omb1[OPT_MAXSEG], omb2[OPT_MAXSEG], eps[OPT_MAXSEG], wd[OPT_MAXSEG];
  int n;
};

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float *__restrict__ g, long n, double *__restrict__ partial,
                                                         unsigned *__restrict__ ticket, float *__restrict__ norm_out, OptSegs S,
                                                         const float *__restrict__ lr_host, float *__restrict__ lr_dev) {
  __shared__ double red[4];
  __shared__ bool last;
  const long per = ((n + gridDim.x - 1) / gridDim.x + 3) & ~3L;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (long i = lo + 4L * threadIdx.x; i < hi; i += 1024) {
    if (i + 4 <= hi) {
      const float4 v = *reinterpret_cast<const float4 *>(g + i);
      a0 += v.x * v.x; a1 += v.y * v.y; a2 += v.z * v.z; a3 += v.w * v.w;
    } else {
      for (long j = i; j < hi; ++j) a0 += g[j] * g[j];
    }
  }
  double s = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partial[blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double t = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)                       // (fixed assignment of slices to threads)
    t += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    norm_out[0] = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
    *ticket = 0u;                                                               // (left zero for the next launch)
    for (int k = 0; k < S.n; ++k) {
      const float st = S.step[k][0] + 1.f;
      S.step[k][0] = st;
      if (lr_host) lr_dev[k] = lr_host[k];
      // bias corrections in double, like torch (1 - 0.999^t in fp32 loses five digits to cancellation)
      lr_dev[OPT_MAXSEG + 2 * k] = (float)(1.0 - pow(S.bd1[k], (double)st));
      lr_dev[OPT_MAXSEG + 2 * k + 1] = (float)sqrt(1.0 - pow(S.bd2[k], (double)st));
    }
  }
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float *__restrict__ p, const float *__restrict__ g, long n, const OptSegs S,
                                                         const float *__restrict__ lr_dev, const float *__restrict__ norm,
                                                         float max_norm, float pre_scale) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  int k = -1;
#pragma unroll
  for (int s = 0; s < OPT_MAXSEG; ++s)
    if (s < S.n && i >= S.lo[s] && i < S.hi[s]) k = s;
  if (k < 0) return;                                                           // (padding between groups)
  // clip_grad_norm_: the buffer may still hold the SUM over ranks; the mean's norm is pre_scale * |sum| and both factors go into
  // the one multiplication (eda_amd/parallel.py FlatParams.clip_grad_norm_)
  float coef = pre_scale;
  if (max_norm > 0.f) coef *= fminf(max_norm / (norm[0] * pre_scale + 1e-6f), 1.f);
  // (1 - beta formed in double by the entry point, like torch: 1 - 0.999f is 1.3e-5 off 0.001)
  const float lr = lr_dev[k], b2 = S.beta2[k], omb1 = S.omb1[k], omb2 = S.omb2[k], eps = S.eps[k], wd = S.wd[k];
  const float bc1 = lr_dev[OPT_MAXSEG + 2 * k], bc2_sqrt = lr_dev[OPT_MAXSEG + 2 * k + 1];      // (eda_grad_sumsq_f32 wrote them)
  const float step_size = lr / bc1;
  const long j = i - S.lo[k];
  const int cnt = S.hi[k] - i >= 4 ? 4 : (int)(S.hi[k] - i);                  // (a group may end inside a 16-byte granule)
  float pp[4], gg[4], mm[4], vq[4];
  if (cnt == 4) {
    const float4 pv = *reinterpret_cast<const float4 *>(p + i), gv = *reinterpret_cast<const float4 *>(g + i);
    const float4 mv = *reinterpret_cast<const float4 *>(S.exp_avg[k] + j), vv = *reinterpret_cast<const float4 *>(S.exp_avg_sq[k] + j);
    pp[0] = pv.x; pp[1] = pv.y; pp[2] = pv.z; pp[3] = pv.w; gg[0] = gv.x; gg[1] = gv.y; gg[2] = gv.z; gg[3] = gv.w;
    mm[0] = mv.x; mm[1] = mv.y; mm[2] = mv.z; mm[3] = mv.w; vq[0] = vv.x; vq[1] = vv.y; vq[2] = vv.z; vq[3] = vv.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool on = e < cnt;
      pp[e] = on ? p[i + e] : 0.f; gg[e] = on ? g[i + e] : 0.f;
      mm[e] = on ? S.exp_avg[k][j + e] : 0.f; vq[e] = on ? S.exp_avg_sq[k][j + e] : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gr = gg[e] * coef;
    float x = pp[e];
    x -= lr * wd * x;
    const float m = mm[e] + (gr - mm[e]) * omb1;                               // lerp(exp_avg, grad, 1 - beta1)
    const float v = b2 * vq[e] + omb2 * gr * gr;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    x -= step_size * m / denom;
    pp[e] = x; mm[e] = m; vq[e] = v;
  }
  if (cnt == 4) {
    *reinterpret_cast<float4 *>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *reinterpret_cast<float4 *>(S.exp_avg[k] + j) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *reinterpret_cast<float4 *>(S.exp_avg_sq[k] + j) = make_float4(vq[0], vq[1], vq[2], vq[3]);
  } else {
    for (int e = 0; e < cnt; ++e) { p[i + e] = pp[e]; S.exp_avg[k][j + e] = mm[e]; S.exp_avg_sq[k][j + e] = vq[e]; }
  }
}

int fill_segs(OptSegs &S, int nseg, const long *lo, const long *hi, float *const *exp_avg, float *const *exp_avg_sq, float *const *step,
              const double *beta1, const double *beta2, const float *eps, const float *wd, long n) {
  EDA_CHECK_ARG(nseg >= 1 && nseg <= OPT_MAXSEG, "1..4 learning-rate groups");
  EDA_CHECK_ARG(lo && hi && step, "null pointer");
  S.n = nseg;
  for (int k = 0; k < nseg; ++k) {
    EDA_CHECK_ARG(lo[k] >= 0 && hi[k] >= lo[k] && hi[k] <= n && lo[k] % 4 == 0, "group ranges: inside the buffer, starting on a multiple of 4");
    EDA_CHECK_ARG(step[k], "null step counter");
    S.lo[k] = lo[k]; S.hi[k] = hi[k]; S.step[k] = step[k];
    S.exp_avg[k] = exp_avg ? exp_avg[k] : nullptr; S.exp_avg_sq[k] = exp_avg_sq ? exp_avg_sq[k] : nullptr;
    S.beta1[k] = beta1 ? (float)beta1[k] : 0.f; S.beta2[k] = beta2 ? (float)beta2[k] : 0.f;
    S.omb1[k] = beta1 ? (float)(1.0 - beta1[k]) : 0.f; S.omb2[k] = beta2 ? (float)(1.0 - beta2[k]) : 0.f;
    S.bd1[k] = beta1 ? beta1[k] : 0.0; S.bd2[k] = beta2 ? beta2[k] : 0.0;
    S.eps[k] = eps ? eps[k] : 0.f; S.wd[k] = wd ? wd[k] : 0.f;
  }
  return 0;
}

}  // namespace

extern "C" size_t eda_grad_sumsq_workspace_bytes(void) { return 1024 * sizeof(double) + 16; }

// grad: n floats (16-byte aligned); ws: eda_grad_sumsq_workspace_bytes() bytes, ZERO before the first call (left zero); norm_out: one
// float = |grad|_2; steps[nseg]: the groups' step counters (one device float each), incremented by one; lr_dev: 12 device floats =
// [4 learning rates | per group: 1 - beta1^step, sqrt(1 - beta2^step)], the rates copied from lr_host (pinned host memory, nseg
// floats; NULL: lr_dev[0..nseg) is the caller's) and the bias corrections formed in double by the same launch.
extern "C" int eda_grad_sumsq_f32(const float *grad, long n, void *ws, float *norm_out, int nseg, float *const *steps, const double *beta1,
                                  const double *beta2, const float *lr_host, float *lr_dev, void *stream_) {
  EDA_CHECK_ARG(n >= 0 && grad && ws && norm_out && beta1 && beta2 && lr_dev, "null pointer");
  EDA_CHECK_ARG((reinterpret_cast<uintptr_t>(grad) & 15u) == 0, "16-byte aligned gradient buffer");
  OptSegs S;
  long z[OPT_MAXSEG] = {0, 0, 0, 0};
  if (int rc = fill_segs(S, nseg, z, z, nullptr, nullptr, steps, beta1, beta2, nullptr, nullptr, n)) return rc;
  long blocks = (n + 16383) / 16384;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  double *partial = static_cast<double *>(ws);
  unsigned *ticket = reinterpret_cast<unsigned *>(partial + 1024);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, grad, n, partial, ticket, norm_out, S,
                     lr_host, lr_dev);
  EDA_CHECK_LAUNCH();
  return 0;
}

// param / grad: the flat buffers (n floats); groups k = 0..nseg-1 own [lo[k], hi[k]) with moment buffers exp_avg[k] / exp_avg_sq[k]
// (hi - lo floats each), step counter steps[k] (already incremented: eda_grad_sumsq_f32), learning rate lr_dev[k] and their own
// beta1 / beta2 / eps / weight_decay; norm: |grad|_2 on the device; max_norm <= 0: no clipping; pre_scale: 1 / world when the buffer
// holds the sum over ranks.
extern "C" int eda_adamw_flat_f32(float *param, const float *grad, long n, int nseg, const long *lo, const long *hi,
                                  float *const *exp_avg, float *const *exp_avg_sq, float *const *steps, const float *lr_dev,
                                  const double *beta1, const double *beta2, const float *eps, const float *weight_decay, const float *norm,
                                  float max_norm, float pre_scale, void *stream_) {
  EDA_CHECK_ARG(n >= 0 && param && grad && exp_avg && exp_avg_sq && lr_dev && beta1 && beta2 && eps && weight_decay && norm, "null pointer");
  EDA_CHECK_ARG(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad)) & 15u) == 0, "16-byte aligned buffers");
  if (n == 0) return 0;
  OptSegs S;
  if (int rc = fill_segs(S, nseg, lo, hi, exp_avg, exp_avg_sq, steps, beta1, beta2, eps, weight_decay, n)) return rc;
  for (int k = 0; k < nseg; ++k)
    EDA_CHECK_ARG(S.hi[k] == S.lo[k] || (S.exp_avg[k] && S.exp_avg_sq[k] && ((reinterpret_cast<uintptr_t>(S.exp_avg[k]) | reinterpret_cast<uintptr_t>(S.exp_avg_sq[k])) & 15u) == 0),
                  "moment buffers: non-null, 16-byte aligned");
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, param, grad, n, S, lr_dev,
                     norm, max_norm, pre_scale);
  EDA_CHECK_LAUNCH();
  return 0;
}
