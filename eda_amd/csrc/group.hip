// group.hip -- index gathers and their scatter-add gradients for gfx950.
//
// Replaces (reference pointnet2/_ext_src/src/):
//   gather_points_kernel        sampling_gpu.cu:13-25     (host sampling.cpp:20-43)
//   gather_points_grad_kernel   sampling_gpu.cu:39-52     (host sampling.cpp:45-69)
//   group_points_kernel         group_points_gpu.cu:13-33 (host group_points.cpp:17-40)
//   group_points_grad_kernel    group_points_gpu.cu:48-69 (host group_points.cpp:42-65)
//
// These are pure HBM-bound copies.  The reference launches one block per scene;
// here the grid covers (scene, channel tile, position tile) so the whole chip
// streams, every thread owns 4 consecutive output positions (one 16-byte
// store per channel) and keeps its 4 indices in registers across the channel
// tile, so the index tensor is read once per CT channels instead of per channel.
#include "eda_common.h"

namespace {

constexpr int GP_THREADS = 256;
constexpr int GP_VEC = 4;   // output positions per thread
constexpr int GP_CT = 8;    // channels per workgroup

// ---- group_points: out[b,l,pos] = points[b,l,idx[b,pos]], pos in [0, J) -----
template <bool VEC_OK>
__global__ __launch_bounds__(GP_THREADS) void group_points_kernel(
    const float *__restrict__ points, const int *__restrict__ idx, int c, int n, int J,
    float *__restrict__ out) {
  const int scene = blockIdx.z;
  const int l0 = blockIdx.y * GP_CT;
  const int pos0 = (blockIdx.x * GP_THREADS + threadIdx.x) * GP_VEC;
  if (pos0 >= J) return;
  const int *ix = idx + (size_t)scene * J + pos0;
  int i0, i1 = 0, i2 = 0, i3 = 0;
  const int rem = J - pos0;
  if (VEC_OK) {
    const int4 v = *reinterpret_cast<const int4 *>(ix);
    i0 = v.x; i1 = v.y; i2 = v.z; i3 = v.w;
  } else {
    i0 = ix[0];
    if (rem > 1) i1 = ix[1];
    if (rem > 2) i2 = ix[2];
    if (rem > 3) i3 = ix[3];
  }
  const int lend = min(l0 + GP_CT, c);
#pragma unroll 4
  for (int l = l0; l < lend; ++l) {
    const float *p = points + ((size_t)scene * c + l) * n;
    float *o = out + ((size_t)scene * c + l) * J + pos0;
    const float a0 = p[i0], a1 = p[i1], a2 = p[i2], a3 = p[i3];
    if (VEC_OK) {
      *reinterpret_cast<float4 *>(o) = make_float4(a0, a1, a2, a3);
    } else {
      o[0] = a0;
      if (rem > 1) o[1] = a1;
      if (rem > 2) o[2] = a2;
      if (rem > 3) o[3] = a3;
    }
  }
}

// ---- group_points_grad: grad_points[b,l,idx[b,pos]] += grad_out[b,l,pos] ----
template <bool VEC_OK>
__global__ __launch_bounds__(GP_THREADS) void group_points_grad_kernel(
    const float *__restrict__ grad_out, const int *__restrict__ idx, int c, int n, int J,
    float *__restrict__ grad_points) {
  const int scene = blockIdx.z;
  const int l0 = blockIdx.y * GP_CT;
  const int pos0 = (blockIdx.x * GP_THREADS + threadIdx.x) * GP_VEC;
  if (pos0 >= J) return;
  const int *ix = idx + (size_t)scene * J + pos0;
  int i0, i1 = 0, i2 = 0, i3 = 0;
  const int rem = J - pos0;
  if (VEC_OK) {
    const int4 v = *reinterpret_cast<const int4 *>(ix);
    i0 = v.x; i1 = v.y; i2 = v.z; i3 = v.w;
  } else {
    i0 = ix[0];
    if (rem > 1) i1 = ix[1];
    if (rem > 2) i2 = ix[2];
    if (rem > 3) i3 = ix[3];
  }
  // Ball-query rows are padded with repeats of the first hit, so neighbouring
  // positions often share a target: fold equal neighbours into slot 0 and issue
  // one atomic for them (same address, same channel row, so the sum is the same).
  const bool live1 = rem > 1, live2 = rem > 2, live3 = rem > 3;
  const bool m1 = live1 && i1 == i0, m2 = live2 && i2 == i0, m3 = live3 && i3 == i0;
  const int lend = min(l0 + GP_CT, c);
  for (int l = l0; l < lend; ++l) {
    float *gp = grad_points + ((size_t)scene * c + l) * n;
    const float *g = grad_out + ((size_t)scene * c + l) * J + pos0;
    float a0, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (VEC_OK) {
      const float4 v = *reinterpret_cast<const float4 *>(g);
      a0 = v.x; a1 = v.y; a2 = v.z; a3 = v.w;
    } else {
      a0 = g[0];
      if (live1) a1 = g[1];
      if (live2) a2 = g[2];
      if (live3) a3 = g[3];
    }
    if (m1) a0 += a1;
    if (m2) a0 += a2;
    if (m3) a0 += a3;
    atomicAdd(gp + i0, a0);
    if (live1 && !m1) atomicAdd(gp + i1, a1);
    if (live2 && !m2) atomicAdd(gp + i2, a2);
    if (live3 && !m3) atomicAdd(gp + i3, a3);
  }
}

// ---- gather_points: out[b,l,j] = points[b,l,idx[b,j]] -----------------------
__global__ __launch_bounds__(256) void gather_points_kernel(const float *__restrict__ points,
                                                            const int *__restrict__ idx, int c,
                                                            int n, int m,
                                                            float *__restrict__ out) {
  const int scene = blockIdx.z;
  const int l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)scene * m + j];
  out[((size_t)scene * c + l) * m + j] = points[((size_t)scene * c + l) * n + a];
}

__global__ __launch_bounds__(256) void gather_points_grad_kernel(
    const float *__restrict__ grad_out, const int *__restrict__ idx, int c, int n, int m,
    float *__restrict__ grad_points) {
  const int scene = blockIdx.z;
  const int l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)scene * m + j];
  atomicAdd(grad_points + ((size_t)scene * c + l) * n + a,
            grad_out[((size_t)scene * c + l) * m + j]);
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int eda_group_points_f32(const float *points, const int *idx, int b, int c, int n,
                                    int npoints, int nsample, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "negative dimension");
  const long long J = (long long)npoints * nsample;
  if (b == 0 || c == 0 || J == 0) return 0;
  EDA_CHECK_ARG(points && idx && out, "null pointer");
  EDA_CHECK_ARG(J < (1ll << 31) && b <= 65535 && (c + GP_CT - 1) / GP_CT <= 65535, "shape too large");
  const dim3 grid((unsigned)((J + GP_THREADS * GP_VEC - 1) / (GP_THREADS * GP_VEC)),
                  (unsigned)((c + GP_CT - 1) / GP_CT), (unsigned)b);
  const bool vec = (J % 4 == 0) && aligned16(idx) && aligned16(out);
  if (vec)
    hipLaunchKernelGGL(group_points_kernel<true>, grid, dim3(GP_THREADS), 0, stream, points, idx, c,
                       n, (int)J, out);
  else
    hipLaunchKernelGGL(group_points_kernel<false>, grid, dim3(GP_THREADS), 0, stream, points, idx,
                       c, n, (int)J, out);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_group_points_grad_f32(const float *grad_out, const int *idx, int b, int c, int n,
                                         int npoints, int nsample, float *grad_points,
                                         void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "negative dimension");
  const long long J = (long long)npoints * nsample;
  if (b == 0 || c == 0 || n == 0) return 0;
  EDA_CHECK_ARG(grad_points, "null pointer");
  if (eda_deterministic() && J > 0 && J < (1ll << 31) && b <= 65535) {     // ordered per-point sums instead of atomics
    EDA_CHECK_ARG(grad_out && idx, "null pointer");
    EdaDetScatter d = {idx, nullptr, (long)J, (int)J, 1, grad_out, (long)c * J, 1, (long)J,
                       grad_points, (long)c * n, 1, (long)n, b, n, c};
    return eda_det_scatter_launch(d, stream);
  }
  { const int zrc__ = eda_zero_async(grad_points, sizeof(float) * (size_t)b * c * n, stream); if (zrc__) return zrc__; }
  if (J == 0) return 0;
  EDA_CHECK_ARG(grad_out && idx, "null pointer");
  EDA_CHECK_ARG(J < (1ll << 31) && b <= 65535 && (c + GP_CT - 1) / GP_CT <= 65535, "shape too large");
  const dim3 grid((unsigned)((J + GP_THREADS * GP_VEC - 1) / (GP_THREADS * GP_VEC)),
                  (unsigned)((c + GP_CT - 1) / GP_CT), (unsigned)b);
  const bool vec = (J % 4 == 0) && aligned16(idx) && aligned16(grad_out);
  if (vec)
    hipLaunchKernelGGL(group_points_grad_kernel<true>, grid, dim3(GP_THREADS), 0, stream, grad_out,
                       idx, c, n, (int)J, grad_points);
  else
    hipLaunchKernelGGL(group_points_grad_kernel<false>, grid, dim3(GP_THREADS), 0, stream, grad_out,
                       idx, c, n, (int)J, grad_points);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_gather_points_f32(const float *points, const int *idx, int b, int c, int n,
                                     int m, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && m >= 0, "negative dimension");
  if (b == 0 || c == 0 || m == 0) return 0;
  EDA_CHECK_ARG(points && idx && out, "null pointer");
  EDA_CHECK_ARG(b <= 65535 && c <= 65535, "shape too large");
  const dim3 grid((unsigned)((m + 255) / 256), (unsigned)c, (unsigned)b);
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, stream, points, idx, c, n, m, out);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" int eda_gather_points_grad_f32(const float *grad_out, const int *idx, int b, int c,
                                          int n, int m, float *grad_points, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && m >= 0, "negative dimension");
  if (b == 0 || c == 0 || n == 0) return 0;
  EDA_CHECK_ARG(grad_points, "null pointer");
  if (eda_deterministic() && m > 0 && b <= 65535) {
    EDA_CHECK_ARG(grad_out && idx, "null pointer");
    EdaDetScatter d = {idx, nullptr, (long)m, m, 1, grad_out, (long)c * m, 1, (long)m,
                       grad_points, (long)c * n, 1, (long)n, b, n, c};
    return eda_det_scatter_launch(d, stream);
  }
  { const int zrc__ = eda_zero_async(grad_points, sizeof(float) * (size_t)b * c * n, stream); if (zrc__) return zrc__; }
  if (m == 0) return 0;
  EDA_CHECK_ARG(grad_out && idx, "null pointer");
  EDA_CHECK_ARG(b <= 65535 && c <= 65535, "shape too large");
  const dim3 grid((unsigned)((m + 255) / 256), (unsigned)c, (unsigned)b);
  hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, stream, grad_out, idx, c, n, m,
                     grad_points);
  EDA_CHECK_LAUNCH();
  return 0;
}
