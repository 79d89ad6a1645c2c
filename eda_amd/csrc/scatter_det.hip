// scatter_det.hip -- the ORDERED form of every scatter-add of the path (eda_set_deterministic / EDA_DETERMINISTIC=1).
//
// The backward passes of the gathering ops add gradient rows into the points they were gathered from:
//   group_points_grad, gather_points_grad      pointnet2/_ext_src/src/group_points_gpu.cu:46-69, sampling_gpu.cu:37-62
//   three_interpolate_grad                     pointnet2/_ext_src/src/interpolate_gpu.cu:112-148
//   the fused set-abstraction backward's d(features) (csrc/sa_cl.hip, csrc/wgrad.hip, csrc/gemm.hip E_SCATTER)
// The reference does it with fp32 atomicAdd in whatever order the hardware retires them, and so do the default kernels here:
// two runs of one step differ in the last bits (and a top-k / arg-max tie downstream can then go either way).  The
// reference's answer is `cudnn.deterministic = True` (train_dist_mod.py:342-344), which does not even cover its own
// atomics; this file is the stronger counterpart: out[b][p][:] = sum over the entries r of scene b with idx[b][r] == p, IN
// ASCENDING r, of w[b][r] * src[b][r / rdiv][:] -- per-owner sums, no atomics, bit-identical from run to run (and the order
// the CPU oracle uses, oracle/eda_oracle.c).
//
// One wave owns one point: it scans the scene's index list 64 entries at a time (staged through LDS in 4096-entry
// chunks, shared by the 64 points of the workgroup), `ballot(idx == p)` gives the matches in ascending order, and for each
// match the 64 lanes add 64 channels (x NACC per pass) of the source row.  Cost: B x P x R / 64 wave-iterations of a
// three-instruction loop -- 8.4 M for SA2's 8 x 2048 points x 32 768 rows (~35 us) -- plus the row reads the atomic form
// does too.
#include "eda_common.h"

namespace {

constexpr int SD_WAVES = 16, SD_PPW = 4, SD_CHUNK = 4096, SD_NACC = 4, SD_PEND = 128, SD_BATCH = 8;

template <bool WGT>
__global__ __launch_bounds__(64 * SD_WAVES) void det_scatter_kernel(const EdaDetScatter a) {
  __shared__ int s_idx[SD_CHUNK];
  __shared__ float s_w[WGT ? SD_CHUNK : 4];
  __shared__ int pend[SD_WAVES][SD_PEND];                   // per wave: chunk-relative entries of the point being summed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * (SD_WAVES * SD_PPW) + wave * SD_PPW;
  const int *idx_b = a.idx + (long)b * a.idx_sb;
  const float *wgt_b = WGT ? a.wgt + (long)b * a.idx_sb : nullptr;
  const float *src_b = a.src + (long)b * a.src_sb;
  float *out_b = a.out + (long)b * a.out_sb;
  for (int c0 = 0; c0 < a.C; c0 += 64 * SD_NACC) {
    float acc[SD_PPW][SD_NACC];
#pragma unroll
    for (int q = 0; q < SD_PPW; ++q)
#pragma unroll
      for (int k = 0; k < SD_NACC; ++k) acc[q][k] = 0.f;
    for (int r0 = 0; r0 < a.R; r0 += SD_CHUNK) {
      __syncthreads();
      const int n = min(SD_CHUNK, a.R - r0);
      for (int i = tid; i < SD_CHUNK; i += 64 * SD_WAVES) {
        s_idx[i] = i < n ? idx_b[r0 + i] : -1;
        if (WGT) s_w[i] = i < n ? wgt_b[r0 + i] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < SD_PPW; ++q) {
        const int p = p0 + q;
        if (p >= a.P) break;                                  // (wave-uniform)
        // matches are collected (ascending) into a wave-private list and consumed SD_BATCH at a time: the row loads of a
        // batch are all in flight before the first add, the adds keep the entry order (one dependent load per match made
        // this kernel latency-bound: SA2's scatter 0.7 ms)
        int npend = 0;
        auto flush = [&](int count) {
          __builtin_amdgcn_wave_barrier();                    // (the list was written by other lanes of this wave)
          for (int j0 = 0; j0 < count; j0 += SD_BATCH) {
            float v[SD_BATCH][SD_NACC];
#pragma unroll
            for (int u = 0; u < SD_BATCH; ++u) {
              const int e = min(j0 + u, count - 1);
              const int r = r0 + pend[wave][e];
              const float *row = src_b + (long)(r / a.rdiv) * a.src_sr;
#pragma unroll
              for (int k = 0; k < SD_NACC; ++k) {
                const int c = c0 + 64 * k + lane;
                v[u][k] = c < a.C ? row[(long)c * a.src_sc] : 0.f;
              }
            }
#pragma unroll
            for (int u = 0; u < SD_BATCH; ++u) {
              if (j0 + u < count) {
                const float wv = WGT ? s_w[pend[wave][j0 + u]] : 1.f;
#pragma unroll
                for (int k = 0; k < SD_NACC; ++k) acc[q][k] += WGT ? v[u][k] * wv : v[u][k];
              }
            }
          }
        };
        for (int i0 = 0; i0 < n; i0 += 64) {
          unsigned long long mt = __ballot(s_idx[i0 + lane] == p);
          const int cnt = __builtin_popcountll(mt);
          if (cnt == 0) continue;
          if (npend + cnt > SD_PEND) { flush(npend); npend = 0; }
          // lane with the k-th set bit writes entry k (ascending)
          if ((mt >> lane) & 1ull) pend[wave][npend + __builtin_popcountll(mt & ((1ull << lane) - 1ull))] = i0 + lane;
          npend += cnt;
        }
        flush(npend);
      }
    }
#pragma unroll
    for (int q = 0; q < SD_PPW; ++q) {
      const int p = p0 + q;
      if (p >= a.P) break;
#pragma unroll
      for (int k = 0; k < SD_NACC; ++k) {
        const int c = c0 + 64 * k + lane;
        if (c < a.C) out_b[(long)p * a.out_sp + (long)c * a.out_sc] = acc[q][k];
      }
    }
  }
}

}  // namespace

int eda_det_scatter_launch(const EdaDetScatter &a, hipStream_t stream) {
  if (a.B <= 0 || a.P <= 0 || a.C <= 0) return 0;
  if (a.B > 65535 || a.rdiv < 1) { eda_set_error("ordered scatter: bad shape"); return EDA_ERR_INVALID_ARG; }
  const dim3 grid((unsigned)((a.P + SD_WAVES * SD_PPW - 1) / (SD_WAVES * SD_PPW)), (unsigned)a.B);
  if (a.wgt) hipLaunchKernelGGL(det_scatter_kernel<true>, grid, dim3(64 * SD_WAVES), 0, stream, a);
  else hipLaunchKernelGGL(det_scatter_kernel<false>, grid, dim3(64 * SD_WAVES), 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("ordered scatter: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// C ABI: the weight gradient of an embedding lookup as ordered per-row sums (eda_amd/nn_utils.py _EmbeddingRows in the
// deterministic mode; torch's index_add_ is an fp32-atomic scatter)
extern "C" int eda_index_add_rows_ordered_f32(const float *src, const int *idx, long R, int C, int P, float *out, void *stream_) {
  EDA_CHECK_ARG(R >= 0 && C >= 0 && P >= 0 && R < 0x7fffffffL, "bad dimension");
  if (P == 0 || C == 0) return 0;
  EDA_CHECK_ARG(out && (R == 0 || (src && idx)), "null pointer");
  EdaDetScatter d = {idx, nullptr, R, (int)R, 1, src, R * C, (long)C, 1, out, (long)P * C, (long)C, 1, 1, P, C};
  return eda_det_scatter_launch(d, (hipStream_t)stream_);
}
