// ball_query.hip -- radius neighbour query for gfx950.
//
// Replaces query_ball_point_kernel (reference
// pointnet2/_ext_src/src/ball_query_gpu.cu:14-49; host ball_query.cpp:13-37).
// Result contract (bit-exact): for centre j the first `nsample` indices k, in
// ascending k, with d2(new_xyz[j], xyz[k]) < radius^2 (fp32, strict); the row
// is padded with the first hit; a centre with no hit keeps an all-zero row.
//
// Machine mapping.  The reference uses one THREAD per centre and one block per
// scene (8 of 256 CUs busy at B = 8).  Here one WAVE owns CPW centres and the
// 64 lanes test 64 consecutive points at once: hits are appended in index order
// with ballot + prefix-popcount, so "first nsample by ascending index" holds by
// construction.  A workgroup (4 waves, 16 centres) stages each 1024-point chunk
// of the scene once in LDS (AoS; a stride-3 dword read is bank-conflict free on
// the 32-bank ds_read_b32 path) so HBM/L2 traffic per scene is n*12 bytes per
// 16 centres instead of per centre.  Grid = b * ceil(m / 16) workgroups.
#include "eda_common.h"

namespace {

constexpr int BQ_THREADS = 256;
constexpr int BQ_WAVES = BQ_THREADS / 64;
constexpr int BQ_CPW = 4;                       // centres per wave
constexpr int BQ_CPB = BQ_WAVES * BQ_CPW;       // centres per workgroup
constexpr int BQ_CHUNK = 1024;                  // points staged per iteration

template <int MODE>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all, int n, int m,
    float radius2, int nsample, int *__restrict__ idx_all, int blocks_per_scene) {
  __shared__ float pts[BQ_CHUNK * 3];

  const int scene = blockIdx.x / blocks_per_scene;
  const int cblk = blockIdx.x % blocks_per_scene;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const float *xyz = xyz_all + (size_t)scene * n * 3;
  const float *new_xyz = new_xyz_all + (size_t)scene * m * 3;
  int *idx = idx_all + (size_t)scene * m * nsample;

  const int j0 = cblk * BQ_CPB + wave * BQ_CPW;   // first centre of this wave

  float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
  int cnt[BQ_CPW], first[BQ_CPW];
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    const int j = j0 + c;
    const bool live = j < m;
    cx[c] = live ? new_xyz[j * 3 + 0] : 0.f;
    cy[c] = live ? new_xyz[j * 3 + 1] : 0.f;
    cz[c] = live ? new_xyz[j * 3 + 2] : 0.f;
    cnt[c] = live ? 0 : nsample;                  // dead centres are "done"
    first[c] = 0;
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  for (int base = 0; base < n; base += BQ_CHUNK) {
    // Workgroup-uniform early exit: every centre of every wave already full.
    bool wave_done = true;
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) wave_done = wave_done && (cnt[c] >= nsample);
    if (__syncthreads_and(wave_done)) break;      // also orders LDS reuse

    const int npts = min(BQ_CHUNK, n - base);
    const float *src = xyz + (size_t)base * 3;
    for (int f = tid; f < npts * 3; f += BQ_THREADS) pts[f] = src[f];   // coalesced dwords
    __syncthreads();

    if (!wave_done) {
      for (int g = 0; g < npts; g += 64) {
        const int p = g + lane;
        const bool in = p < npts;
        const float x = in ? pts[p * 3 + 0] : 0.f;
        const float y = in ? pts[p * 3 + 1] : 0.f;
        const float z = in ? pts[p * 3 + 2] : 0.f;
        const int k = base + p;
#pragma unroll
        for (int c = 0; c < BQ_CPW; ++c) {
          if (cnt[c] >= nsample) continue;         // wave-uniform
          // centre minus point (ball_query_gpu.cu:36-37)
          const float d2 = eda_sumsq3<MODE>(cx[c] - x, cy[c] - y, cz[c] - z);
          const bool hit = in && (d2 < radius2);
          const unsigned long long mask = __ballot(hit);
          if (mask == 0ull) continue;
          if (cnt[c] == 0) first[c] = base + g + (__ffsll((long long)mask) - 1);
          const int rank = cnt[c] + __popcll(mask & lt_mask);
          if (hit && rank < nsample) idx[(size_t)(j0 + c) * nsample + rank] = k;
          cnt[c] += __popcll(mask);
        }
      }
    }
  }

  // pad with the first hit (ball_query_gpu.cu:39-43); empty ball -> zeros
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    const int j = j0 + c;
    if (j >= m) continue;
    const int have = min(cnt[c], nsample);
    for (int s = have + lane; s < nsample; s += 64) idx[(size_t)j * nsample + s] = first[c];
  }
}

}  // namespace

extern "C" int eda_ball_query_f32(const float *new_xyz, const float *xyz, int b, int n, int m,
                                  float radius, int nsample, int *idx, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative dimension");
  if (b == 0 || m == 0 || nsample == 0) return 0;
  EDA_CHECK_ARG(new_xyz && idx && (xyz || n == 0), "null pointer");
  const float radius2 = radius * radius;          // ball_query_gpu.cu:26, fp32
  const int bps = (m + BQ_CPB - 1) / BQ_CPB;
  const dim3 grid((unsigned)((size_t)b * bps));
  if (g_eda_fma_mode == 0)
    hipLaunchKernelGGL(ball_query_kernel<0>, grid, dim3(BQ_THREADS), 0, stream, new_xyz, xyz, n, m,
                       radius2, nsample, idx, bps);
  else
    hipLaunchKernelGGL(ball_query_kernel<1>, grid, dim3(BQ_THREADS), 0, stream, new_xyz, xyz, n, m,
                       radius2, nsample, idx, bps);
  EDA_CHECK_LAUNCH();
  return 0;
}
