// lsa.hip -- minimum-cost assignment of a scene's ground-truth boxes to queries, on the device.
//
// The reference's HungarianMatcher (models/losses.py:262-336) builds a (B, Q, sum T_b) cost
// matrix on the GPU, copies it to the host and calls scipy.optimize.linear_sum_assignment once
// per scene and per prediction head: 14 host synchronisations per training step
// (2 matchers x 7 heads, losses.py:319-329, 617-628).  This kernel solves all scenes of a head in
// one launch and leaves the result on the device, so the loss can stay inside the captured step.
//
// Problem per scene b: cost[b] is (Q queries) x (G target slots), of which the first n_b columns
// are real targets (n_b <= Q); find distinct queries q(t), t < n_b, minimising sum_t cost[q(t)][t].
// Algorithm: shortest augmenting paths with dual potentials (Jonker-Volgenant / "Hungarian"),
// the same family scipy 1.15 implements (scipy/optimize/rectangular_lsap, Crouse 2016); the
// optimum is unique unless costs tie, and potentials / path lengths are kept in fp64 like there.
// One workgroup per scene: thread j owns query columns j, j+256, ...; every step of a path search
// is one parallel relaxation of the unvisited columns plus a workgroup arg-min (lowest column on
// ties).  Work O(n_b^2 Q): n_b is 1-3 on ScanRefer-style batches (the referred object and its
// context), tens of microseconds; a full 132-target scene costs ~17k steps (tens of ms) and is
// still correct.
#include "eda_common.h"

#define LSA_THREADS 256
#define LSA_MAXQ 1024
#define LSA_MAXT 256

namespace {

struct LsaBest { double v; int j; };

__device__ __forceinline__ LsaBest lsa_min(LsaBest a, LsaBest b) {
  return (b.v < a.v || (b.v == a.v && b.j < a.j)) ? b : a;
}

__global__ __launch_bounds__(LSA_THREADS) void lsa_kernel(const float *__restrict__ cost, long sb, long sq,
                                                          long st, int Q, int G,
                                                          const int *__restrict__ ntargets,
                                                          int *__restrict__ assign) {
  __shared__ double u[LSA_MAXT + 1], v[LSA_MAXQ + 1], minv[LSA_MAXQ + 1];
  __shared__ int p[LSA_MAXQ + 1], way[LSA_MAXQ + 1];
  __shared__ unsigned char used[LSA_MAXQ + 1];
  __shared__ double wbest_v[LSA_THREADS / 64];
  __shared__ int wbest_j[LSA_THREADS / 64];
  __shared__ int sh_j0, sh_done;
  __shared__ double sh_delta;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *c = cost + (long)b * sb;
  int n = ntargets[b];
  if (n > G) n = G;
  if (n > Q) n = Q;
  int *out = assign + (long)b * G;
  for (int g = tid; g < G; g += LSA_THREADS) out[g] = -1;
  for (int j = tid; j <= Q; j += LSA_THREADS) { v[j] = 0.0; p[j] = 0; }
  for (int i = tid; i <= n; i += LSA_THREADS) u[i] = 0.0;
  __syncthreads();
  const double INF = 1e300;
  for (int i = 1; i <= n; ++i) {
    for (int j = tid; j <= Q; j += LSA_THREADS) { minv[j] = INF; used[j] = 0; }
    if (tid == 0) { p[0] = i; sh_j0 = 0; }
    __syncthreads();
    for (;;) {
      const int j0 = sh_j0;
      const int i0 = p[j0];
      const double ui0 = u[i0];
      __syncthreads();                       // everyone has read sh_j0 / p[j0] before they change
      if (tid == 0) used[j0] = 1;
      // relax the unvisited columns with row i0, keep the smallest label
      LsaBest best = {INF, 0x7fffffff};
      for (int j = tid + 1; j <= Q; j += LSA_THREADS) {
        if (j != j0 && !used[j]) {
          const double cur = (double)c[(long)(j - 1) * sq + (long)(i0 - 1) * st] - ui0 - v[j];
          double m = minv[j];
          if (cur < m) { m = cur; minv[j] = cur; way[j] = j0; }
          if (m < best.v || (m == best.v && j < best.j)) { best.v = m; best.j = j; }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        LsaBest other = {__shfl_xor(best.v, o), __shfl_xor(best.j, o)};
        best = lsa_min(best, other);
      }
      if (lane == 0) { wbest_v[wave] = best.v; wbest_j[wave] = best.j; }
      __syncthreads();
      if (tid == 0) {
        LsaBest t = {wbest_v[0], wbest_j[0]};
        for (int w = 1; w < LSA_THREADS / 64; ++w) t = lsa_min(t, LsaBest{wbest_v[w], wbest_j[w]});
        sh_delta = t.v;
        sh_j0 = t.j;
        sh_done = p[t.j] == 0;               // p[] is only rewritten by the augmentation below
      }
      __syncthreads();
      const double delta = sh_delta;
      // dual update: visited columns (incl. the virtual column 0) move with their rows
      for (int j = tid; j <= Q; j += LSA_THREADS) {
        if (used[j] || j == j0) { u[p[j]] += delta; v[j] -= delta; }
        else minv[j] -= delta;
      }
      __syncthreads();
      // Free column reached?  Decided by thread 0 BEFORE the barriers above and read from a flag:
      // testing p[j1] here would race with thread 0's augmentation (its first write is p[j1] = ... != 0),
      // a wave arriving late would stay in the loop while thread 0 moves on to the next row.
      if (sh_done) break;
    }
    if (tid == 0) {                          // augment along the stored path
      int j0 = sh_j0;
      do {
        const int j1 = way[j0];
        p[j0] = p[j1];
        j0 = j1;
      } while (j0 != 0);
    }
    __syncthreads();
  }
  for (int j = tid + 1; j <= Q; j += LSA_THREADS)
    if (p[j] != 0) out[p[j] - 1] = j - 1;
}

}  // namespace

extern "C" int eda_lsa_f32(const float *cost, long sb, long sq, long st, int B, int Q, int G,
                           const int *ntargets, int *assign, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EDA_CHECK_ARG(B >= 0 && Q >= 0 && G >= 0, "negative dimension");
  EDA_CHECK_ARG(Q <= LSA_MAXQ && G <= LSA_MAXT, "at most 1024 queries and 256 target slots are built");
  if (B == 0 || G == 0) return 0;
  EDA_CHECK_ARG(ntargets && assign && (cost || Q == 0), "null pointer");
  hipLaunchKernelGGL(lsa_kernel, dim3((unsigned)B), dim3(LSA_THREADS), 0, stream, cost, sb, sq, st, Q, G,
                     ntargets, assign);
  EDA_CHECK_LAUNCH();
  return 0;
}
