// loss.hip -- the training loss's small-tensor arithmetic as a handful of launches (round 6).
//
// The reference's SetCriterion (models/losses.py:339-647) is ~300 element-wise torch operations on tensors of a few
// thousand elements -- (7 heads x 8 scenes) x 132 target slots x 6 box coordinates, 256 queries x 80 tokens -- and its
// autograd backward is twice that: 690 launches, 3.1 ms of a 20.4 ms training step (tools/loss_census.py), all of it
// launch floor.  Here each loss part is ONE forward launch that also forms its gradient analytically (saved for the
// backward), and ONE backward launch that scales it by the upstream gradient and routes it to the matched query:
//   box loss      losses.py:462-497 (loss_boxes): L1 on centre + 0.2 x L1 on size, 1 - generalised 3-D IoU, on the matched pairs;
//   labels        losses.py:396-460 (loss_pos_align): soft-token cross entropy of every query against its target's token map;
//   alignment     losses.py:499-608 (loss_sem_align): the two-directional masked log-sum-exp contrastive loss of queries x tokens;
//   matching cost losses.py:262-318 (HungarianMatcher): softmax . token map, L1, GIoU for the REAL target slots only (csrc/lsa.hip
//                 reads nothing else), and the slot -> query inverse of the assignment.
// Same numbers as eda_amd/losses.py's torch form (tests/test_losses_fused_gpu.py: value and gradient against autograd of
// the torch form; tests/test_losses.py: the reference's own goldens through either form).
#include "eda_common.h"

namespace {

struct Box { float lo[3], hi[3], w[3]; };

// corner form of a (centre, size) box, sizes clamped at 1e-6 (losses.py:33-43)
__device__ __forceinline__ Box corners(const float (&b)[6]) {
  Box o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.w[i] = fmaxf(b[3 + i], 1e-6f);
    o.lo[i] = b[i] - 0.5f * o.w[i];
    o.hi[i] = b[i] + 0.5f * o.w[i];
  }
  return o;
}

// 1 - GIoU(a, t) and its gradient with respect to a's (centre, size); t is a constant (losses.py:46-97)
__device__ __forceinline__ float giou_loss(const float (&a6)[6], const float (&t6)[6], float (&grad)[6]) {
  const Box a = corners(a6), t = corners(t6);
  float e[3], hl[3], wa[3], wt[3];
  float inter = 1.f, hull = 1.f, va = 1.f, vt = 1.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    e[i] = fmaxf(fminf(a.hi[i], t.hi[i]) - fmaxf(a.lo[i], t.lo[i]), 0.f);
    hl[i] = fmaxf(fmaxf(a.hi[i], t.hi[i]) - fminf(a.lo[i], t.lo[i]), 0.f);
    wa[i] = a.hi[i] - a.lo[i];
    wt[i] = t.hi[i] - t.lo[i];
    inter *= e[i]; hull *= hl[i]; va *= wa[i]; vt *= wt[i];
  }
  // (torch evaluates (x1 - x0) * (y1 - y0) * (z1 - z0) and e0 * e1 * e2 left to right: the same order)
  const float uni = va + vt - inter;
  const float loss = 1.f - (inter / uni - (hull - uni) / hull);
  // d loss / d inter, d uni, d hull with uni treated as an independent variable first
  const float d_inter0 = -1.f / uni;
  const float d_uni = inter / (uni * uni) - 1.f / hull;
  const float d_hull = uni / (hull * hull);
  const float d_inter = d_inter0 - d_uni;          // uni = va + vt - inter
  const float d_va = d_uni;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const float de = d_inter * e[j] * e[k];                                  // d / d e_i
    const float dh = d_hull * hl[j] * hl[k];
    const float dw = d_va * wa[j] * wa[k];
    float g_hi = dw, g_lo = -dw;
    // clamp(x, min = 0) passes the gradient where x >= 0; minimum / maximum pass it to the selected operand (both halves on a tie)
    const float xi = fminf(a.hi[i], t.hi[i]) - fmaxf(a.lo[i], t.lo[i]);
    if (xi >= 0.f) {
      g_hi += de * (a.hi[i] < t.hi[i] ? 1.f : (a.hi[i] == t.hi[i] ? 0.5f : 0.f));
      g_lo -= de * (a.lo[i] > t.lo[i] ? 1.f : (a.lo[i] == t.lo[i] ? 0.5f : 0.f));
    }
    const float xh = fmaxf(a.hi[i], t.hi[i]) - fminf(a.lo[i], t.lo[i]);
    if (xh >= 0.f) {
      g_hi += dh * (a.hi[i] > t.hi[i] ? 1.f : (a.hi[i] == t.hi[i] ? 0.5f : 0.f));
      g_lo -= dh * (a.lo[i] < t.lo[i] ? 1.f : (a.lo[i] == t.lo[i] ? 0.5f : 0.f));
    }
    grad[i] = g_lo + g_hi;                                                   // centre
    grad[3 + i] = a6[3 + i] >= 1e-6f ? 0.5f * (g_hi - g_lo) : 0.f;           // size (through the clamp at 1e-6)
  }
  return loss;
}

// workgroup sum of two values (NW waves; red holds 2 * NW floats)
template <int NW = 4>
__device__ __forceinline__ void block_sum2(float &x, float &y, float *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * w] = x; red[2 * w + 1] = y; }
  __syncthreads();
  x = 0.f; y = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) { x += red[2 * i]; y += red[2 * i + 1]; }
  __syncthreads();
}

// one workgroup per scene-of-a-head: value (summed over the scene's target slots, divided by the box count) and per-pair
// gradient of both box losses
__global__ __launch_bounds__(256) void box_loss_fwd_kernel(const float *__restrict__ pred, long p_sb, long p_sq,
                                                           const float *__restrict__ tgt, const int *__restrict__ assign,
                                                           const unsigned char *__restrict__ valid,
                                                           const float *__restrict__ num_boxes, int Bt, int Q, int G,
                                                           float *__restrict__ loss_l1, float *__restrict__ loss_giou,
                                                           float *__restrict__ g_l1, float *__restrict__ g_giou) {
  __shared__ float red[8];
  const int b = blockIdx.x, bt = b % Bt;               // (targets are per scene, predictions per head x scene)
  float s_l1 = 0.f, s_gi = 0.f;
  for (int g = threadIdx.x; g < G; g += 256) {
    const long i = (long)b * G + g, it = (long)bt * G + g;
    float gl[6] = {0, 0, 0, 0, 0, 0}, gg[6] = {0, 0, 0, 0, 0, 0};
    if (valid[it]) {
      int q = assign[i];
      q = q < 0 ? 0 : (q >= Q ? Q - 1 : q);
      const float *pp = pred + (long)b * p_sb + (long)q * p_sq;
      float a6[6], t6[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) { a6[c] = pp[c]; t6[c] = tgt[it * 6 + c]; }
      float l1 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = a6[c] - t6[c], ds = a6[3 + c] - t6[3 + c];
        l1 += fabsf(d) + 0.2f * fabsf(ds);
        gl[c] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        gl[3 + c] = 0.2f * (ds > 0.f ? 1.f : (ds < 0.f ? -1.f : 0.f));
      }
      s_l1 += l1;
      s_gi += giou_loss(a6, t6, gg);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) { g_l1[i * 6 + c] = gl[c]; g_giou[i * 6 + c] = gg[c]; }
  }
  block_sum2(s_l1, s_gi, red);
  if (threadIdx.x == 0) {
    const float nb = num_boxes[0];
    loss_l1[b] = s_l1 / nb;
    loss_giou[b] = s_gi / nb;
  }
}

// d(pred)[b][q][:] = (w_l1[b] * g_l1[b][tq] + w_giou[b] * g_giou[b][tq]) / num_boxes for the slot tq matched to query q, else 0
__global__ __launch_bounds__(256) void box_loss_bwd_kernel(const float *__restrict__ g_l1, const float *__restrict__ g_giou,
                                                           const long *__restrict__ tq, const float *__restrict__ w_l1,
                                                           const float *__restrict__ w_giou, const float *__restrict__ num_boxes,
                                                           int B, int Q, int G, float *__restrict__ dpred) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Q) return;
  const int b = i / Q;
  const long g = tq[i];
  float o[6] = {0, 0, 0, 0, 0, 0};
  if (g >= 0 && g < G) {
    const float inv = 1.f / num_boxes[0];
    const float a = w_l1[b] * inv, c = w_giou[b] * inv;
    const long base = ((long)b * G + g) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = a * g_l1[base + k] + c * g_giou[base + k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) dpred[(long)i * 6 + k] = o[k];
}

}  // namespace

// pred (PB, Q, 6) boxes with element strides (p_sb, p_sq, 1), PB = heads x Bt; tgt (Bt, G, 6) dense and valid (Bt, G) bytes: the
// scenes' targets (row pb uses scene pb % Bt); assign (PB, G) int32; num_boxes: one float on the device (the normaliser,
// losses.py:630-636).  Out: loss_l1 / loss_giou (PB) = the sums of |dc| + 0.2 |ds| and of 1 - GIoU over the valid matched pairs,
// divided by num_boxes; g_l1 / g_giou (PB, G, 6) dense = the pairs' gradients with respect to the matched prediction (zero where
// invalid), for eda_box_loss_bwd_f32.
extern "C" int eda_box_loss_fwd_f32(const float *pred, long p_sb, long p_sq, const float *tgt, const int *assign,
                                    const unsigned char *valid, const float *num_boxes, int PB, int Bt, int Q, int G,
                                    float *loss_l1, float *loss_giou, float *g_l1, float *g_giou, void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && Bt > 0 && PB % Bt == 0 && Q > 0 && G >= 0, "bad dimension");
  if (PB == 0) return 0;
  EDA_CHECK_ARG(pred && tgt && assign && valid && num_boxes && loss_l1 && loss_giou && g_l1 && g_giou, "null pointer");
  hipLaunchKernelGGL(box_loss_fwd_kernel, dim3((unsigned)PB), dim3(256), 0, (hipStream_t)stream_, pred, p_sb, p_sq, tgt, assign,
                     valid, num_boxes, Bt, Q, G, loss_l1, loss_giou, g_l1, g_giou);
  EDA_CHECK_LAUNCH();
  return 0;
}

// tq (B, Q) int64: the target slot matched to each query or -1; w_l1 / w_giou (B): upstream gradients of loss_l1 / loss_giou;
// dpred (B, Q, 6) dense, every element written.
extern "C" int eda_box_loss_bwd_f32(const float *g_l1, const float *g_giou, const long *tq, const float *w_l1,
                                    const float *w_giou, const float *num_boxes, int B, int Q, int G, float *dpred,
                                    void *stream_) {
  EDA_CHECK_ARG(B >= 0 && Q >= 0 && G > 0, "bad dimension");
  if (B == 0 || Q == 0) return 0;
  EDA_CHECK_ARG(g_l1 && g_giou && tq && w_l1 && w_giou && num_boxes && dpred, "null pointer");
  hipLaunchKernelGGL(box_loss_bwd_kernel, dim3((unsigned)((B * Q + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, g_l1,
                     g_giou, tq, w_l1, w_giou, num_boxes, B, Q, G, dpred);
  EDA_CHECK_LAUNCH();
  return 0;
}

// =====================================================================================================================
// The matched target slot of every query, the matching cost, the position-aligned cross entropy and the semantic-alignment
// contrastive loss.  Heads are stacked into the batch dimension (PB = heads x scenes, eda_amd/losses.py); the targets are
// NOT repeated per head: scene b of the targets serves rows pb with pb % B == b.
namespace {

__device__ __forceinline__ float wave_sum(float v) { return eda_wave_sum_f32(v); }
__device__ __forceinline__ float wave_max(float v) { return eda_wave_max_f32(v); }

// tq[pb][q] = the target slot g < ntargets whose assignment is q, else -1 (one workgroup per pb)
__global__ __launch_bounds__(256) void match_slots_kernel(const int *__restrict__ assign, const int *__restrict__ ntargets,
                                                          int B, int Q, int G, long *__restrict__ tq) {
  const int pb = blockIdx.x, nt = min(ntargets[pb % B], G);
  for (int q = threadIdx.x; q < Q; q += 256) tq[(long)pb * Q + q] = -1;
  __syncthreads();
  for (int g = threadIdx.x; g < nt; g += 256) {
    const int q = assign[(long)pb * G + g];
    if (q >= 0 && q < Q) tq[(long)pb * Q + q] = g;
  }
}

constexpr int CE_MAXK = 8;          // classes per lane: C <= 512

// softmax statistics of a wave's row: x[k] = logits[c = lane + 64 k]; returns (max, sum of exp(x - max))
__device__ __forceinline__ void row_softmax(const float *row, int C, int lane, float (&x)[CE_MAXK], float &m, float &s) {
  m = -INFINITY;
#pragma unroll
  for (int k = 0; k < CE_MAXK; ++k) {
    const int c = lane + 64 * k;
    x[k] = c < C ? row[c] : -INFINITY;
    m = fmaxf(m, x[k]);
  }
  m = wave_max(m);
  s = 0.f;
#pragma unroll
  for (int k = 0; k < CE_MAXK; ++k) s += lane + 64 * k < C ? __expf(x[k] - m) : 0.f;
  s = wave_sum(s);
}

// cost[pb][q][g] = w_bbox * L1 + w_class * (-sum_c softmax[c] pmap[b][g][c]  |  -softmax[label[b][g]]) + w_giou * (-GIoU) for the real
// target slots g < ntargets[b], 0 beyond (HungarianMatcher.cost_matrix, losses.py:262-318); one wave per (pb, q)
__global__ __launch_bounds__(256) void match_cost_kernel(const float *__restrict__ logits, const float *__restrict__ pred,
                                                         const float *__restrict__ tgt_boxes, const float *__restrict__ pmap,
                                                         long pm_sg, const long *__restrict__ labels,
                                                         const int *__restrict__ ntargets, int PB, int B, int Q, int G, int C,
                                                         float w_class, float w_bbox, float w_giou, float *__restrict__ cost) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)PB * Q) return;
  const int pb = (int)(row / Q), b = pb % B;
  const int nt = min(ntargets[b], G);
  float x[CE_MAXK], m, s;
  row_softmax(logits + row * C, C, lane, x, m, s);
  const float inv = 1.f / s;
  float a6[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) a6[c] = pred[row * 6 + c];
  float *out = cost + row * G;
  for (int g = 0; g < nt; ++g) {
    float cls;
    if (pmap) {
      const float *pm = pmap + ((long)b * G + g) * pm_sg;
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < CE_MAXK; ++k) {
        const int c = lane + 64 * k;
        if (c < C) d += __expf(x[k] - m) * inv * pm[c];
      }
      cls = -wave_sum(d);
    } else {
      const long lab = labels[(long)b * G + g];
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < CE_MAXK; ++k) d += (lane + 64 * k == lab) ? __expf(x[k] - m) * inv : 0.f;
      cls = -wave_sum(d);
    }
    if (lane == 0) {
      float t6[6], gr[6];
      float l1 = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) { t6[c] = tgt_boxes[((long)b * G + g) * 6 + c]; l1 += fabsf(a6[c] - t6[c]); }
      const float giou = 1.f - giou_loss(a6, t6, gr);
      float v = w_bbox * l1 + w_class * cls + w_giou * (-giou);
      if (v != v) v = 0.f;                                   // nan_to_num(nan = 0, +-inf = +-1e30) of the torch form
      v = fminf(fmaxf(v, -1e30f), 1e30f);
      out[g] = v;
    }
  }
  for (int g = nt + lane; g < G; g += 64) out[g] = 0.f;
}

struct PosAlignMaps { const float *m[4]; float w[4]; long sb, sg; };     // (B, G, >= C) maps, element strides

// position-aligned soft-token cross entropy (loss_pos_align, losses.py:396-460): workgroup (chunk, pb) takes the query rows
// [chunk * QC, (chunk + 1) * QC) of scene pb, a wave per row; loss[pb][chunk] = sum over its rows of ce / num_boxes (the S chunk
// sums of a scene add up to the reference's value), grad0 (PB, Q, C) = d(sum_q ce) / d(logits)
__global__ __launch_bounds__(256) void pos_align_fwd_kernel(const float *__restrict__ logits, const long *__restrict__ tq,
                                                            const PosAlignMaps M, const float *__restrict__ num_boxes,
                                                            int B, int Q, int G, int C, int QC, float eos,
                                                            float *__restrict__ loss, float *__restrict__ grad0) {
  __shared__ float red[8];
  const int pb = blockIdx.y, b = pb % B, S = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q_end = min(Q, ((int)blockIdx.x + 1) * QC);
  float acc = 0.f;
  for (int q = blockIdx.x * QC + wave; q < q_end; q += 4) {
    const long row = (long)pb * Q + q;
    float x[CE_MAXK], m, s;
    row_softmax(logits + row * C, C, lane, x, m, s);
    const float lse = m + __logf(s);
    const long g = tq[row];
    const bool matched = g >= 0 && g < G;
    float sim[CE_MAXK], S_ = 0.f, ce = 0.f;
#pragma unroll
    for (int k = 0; k < CE_MAXK; ++k) {
      const int c = lane + 64 * k;
      float v = 0.f;
      if (c < C) {
        if (matched) {
          const long o = (long)b * M.sb + g * M.sg + c;
          v = M.m[0][o] * M.w[0] + M.m[1][o] * M.w[1] + M.m[2][o] * M.w[2] + M.m[3][o] * M.w[3];
        } else {
          v = c == C - 1 ? 1.f : 0.f;
        }
        ce += __logf(v + 1e-6f) * v - (x[k] - lse) * v;
      }
      sim[k] = v;
      S_ += v;
    }
    S_ = wave_sum(S_);
    ce = wave_sum(ce);
    const float qw = matched ? 1.f : eos;
    acc += ce * qw;
#pragma unroll
    for (int k = 0; k < CE_MAXK; ++k) {
      const int c = lane + 64 * k;
      if (c < C) grad0[row * C + c] = qw * (__expf(x[k] - lse) * S_ - sim[k]);
    }
  }
  float dummy = 0.f;
  if (lane != 0) acc = 0.f;                 // (every lane of a wave holds the same sum)
  block_sum2(acc, dummy, red);
  if (threadIdx.x == 0) loss[(long)pb * S + blockIdx.x] = acc / num_boxes[0];
}

// out[pb][i] = g0[pb][i] * w[pb][part of i] / num_boxes  (the backward of every per-scene loss whose gradient was formed in the
// forward); a scene's `per` elements are S parts of per_part elements (the last one may be short)
__global__ __launch_bounds__(256) void scale_by_scene_kernel(const float *__restrict__ g0, const float *__restrict__ w,
                                                             const float *__restrict__ num_boxes, long per, long per_part, int S,
                                                             long total, float *__restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  const long pb = i / per;                              // (per % 4 == 0 and per_part % 4 == 0: the four elements share a part)
  const float sc = w[pb * S + (i - pb * per) / per_part] / num_boxes[0];
  const float4 v = *reinterpret_cast<const float4 *>(g0 + i);
  *reinterpret_cast<float4 *>(out + i) = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
}

struct SemMaps { const float *pos, *modi, *pron, *other, *rel; long sb, sg; };

// semantic-alignment contrastive loss (loss_sem_align, losses.py:499-608): one workgroup of 16 waves per pb with the scene's (Q, L)
// logits in LDS (row stride odd: a thread per row reads without bank conflicts).
//   phase 1  object -> text.  Unmatched rows (nearly all): a thread per row, two passes over its tokens in LDS.  Matched rows (a few
//            per scene): a wave per row, lanes over the tokens, the five maps read once; the lane also keeps the column sums of
//            phase 2b over the wave's rows.
//   phase 2a text -> object: the log-sum-exp of every token column over all queries, a wave per 8 columns x 8 row parts;
//   phase 2b the map sums of every column over the matched rows (the waves' sums meet in LDS, in wave order) + the closed form of the
//            unmatched ones;
//   phase 3  the gradient, a wave per query row (coalesced stores); every divisor was inverted once per row / column.
// loss[pb] = (b2t + t2b) / 2 / num_boxes; grad0 (PB, Q, L) = d(b2t + t2b) / 2 / d(logits)
constexpr int SEM_NW = 16;
#ifdef EDA_LOSS_PROFILE
// phase timeline (experiments only, tools/loss_phase_profile.py): wall-clock stamps (100 MHz) of thread 0 of every workgroup
__device__ unsigned long long loss_prof[256 * 8];
#define PL(slot) do { if (threadIdx.x == 0) loss_prof[(blockIdx.x & 255) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PL(slot) do { } while (0)
#endif
constexpr int SEM_MAXK = 4;                 // token columns per lane: L <= 256
constexpr size_t SEM_LDS_MAX = 160 * 1024 - 1024;
__host__ __device__ constexpr int sem_row_stride(int L) { return L + 1 + (L & 1); }

__global__ __launch_bounds__(64 * SEM_NW) void sem_align_fwd_kernel(const float *__restrict__ logits, const long *__restrict__ tq,
                                                                   const SemMaps M, const long *__restrict__ attn_mask,
                                                                   const float *__restrict__ num_boxes, int B, int Q, int G, int L,
                                                                   int R, float eos, float *__restrict__ loss,
                                                                   float *__restrict__ grad0) {
  PL(0);
  extern __shared__ __attribute__((aligned(16))) float sem_smem[];
  constexpr int NT = 64 * SEM_NW;
  const int LS = sem_row_stride(L);
  float *X = sem_smem;                      // [Q][LS]
  float *rowst = X + (size_t)Q * LS;        // [Q][8]: 1/(sp+e), .2/(sm+e), .2/(sr+e), .1/(srel+e), m1, 1/s1, gq, -
  float *colst = rowst + (size_t)Q * 8;     // [4][L]: 1/nb (phase 2a: sum of the unmatched rows' logits), gl, mcol, 1/scol (2a: scol)
  int *slot = reinterpret_cast<int *>(colst + (size_t)L * 4);       // [Q]: matched target slot or -1
  int *mlist = slot + Q;                    // [Q]: the matched rows, ascending
  float *part_sums = reinterpret_cast<float *>(mlist + Q);          // [R][8][L]: R waves' column sums at a time
  __shared__ float red[2 * SEM_NW];
  __shared__ int shared_i[3];               // last, prev, number of matched rows
  const int pb = blockIdx.x, b = pb % B, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  {
    const float *src = logits + (long)pb * Q * L;
    const int n = Q * L;
    if ((n & 3) == 0) {                     // four 16-byte loads in flight per thread
      const int n4 = n >> 2;
      for (int i0 = tid; i0 < n4; i0 += NT * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + u * NT < n4) v[u] = reinterpret_cast<const float4 *>(src)[i0 + u * NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u * NT >= n4) continue;
          const int i = (i0 + u * NT) * 4;
          int q = i / L, l = i - q * L;
          const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            X[q * LS + l] = e[j];
            if (++l == L) { l = 0; ++q; }
          }
        }
      }
    } else {
      for (int i = tid; i < n; i += NT) {
        const int q = i / L, l = i - q * L;
        X[q * LS + l] = src[i];
      }
    }
  }
  for (int q = tid; q < Q; q += NT) {
    const long g = tq[(long)pb * Q + q];
    slot[q] = (g >= 0 && g < G) ? (int)g : -1;
  }
  __syncthreads();
  if (wv == 0) {                            // number of real tokens -> the "not mentioned" token and the one before it
    long n = 0;
    for (int l = lane; l < L; l += 64) n += attn_mask[(long)b * L + l];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) {
      shared_i[0] = (int)((((n - 1) % L) + L) % L);
      shared_i[1] = (int)((((n - 2) % L) + L) % L);
    }
  } else if (wv == 1) {                     // ordered list of the matched rows
    int nm = 0;
    for (int q0 = 0; q0 < Q; q0 += 64) {
      const int q = q0 + lane;
      const bool mt = q < Q && slot[q] >= 0;
      const unsigned long long bal = __ballot(mt);
      if (mt) mlist[nm + __popcll(bal & ((1ull << lane) - 1))] = q;
      nm += __popcll(bal);
    }
    if (lane == 0) shared_i[2] = nm;
  }
  __syncthreads();
  PL(1);
  const int last = shared_i[0], prev = shared_i[1], nmatched = shared_i[2];
  const float *mp = M.pos + (long)b * M.sb, *mm = M.modi + (long)b * M.sb, *mr = M.pron + (long)b * M.sb;
  const float *mo = M.other + (long)b * M.sb, *ml = M.rel + (long)b * M.sb;
  // ---- phase 1, unmatched rows: the two "not mentioned" tokens are the positives, nothing else is set; four threads per row ----
  float b2t_acc = 0.f;
  for (int q = tid >> 2; q < ((Q + 255) & ~255); q += NT / 4) {
    const bool on = q < Q && slot[q] < 0;
    const float *x = X + (on ? q : 0) * LS;
    float m1 = -INFINITY, s1 = 0.f;
    if (on) {
#pragma unroll 4
      for (int l = tid & 3; l < L; l += 4) m1 = fmaxf(m1, x[l]);
    }
    m1 = fmaxf(m1, __shfl_xor(m1, 1)); m1 = fmaxf(m1, __shfl_xor(m1, 2));
    if (on) {
#pragma unroll 4
      for (int l = tid & 3; l < L; l += 4) s1 += __expf(x[l] - m1);
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    if (on && (tid & 3) == 0) {
      const float sp = last == prev ? 1.f : 2.f;
      const float dp = last == prev ? x[last] : x[last] + x[prev];
      b2t_acc += (-dp / (sp + 1e-6f) + (m1 + __logf(s1))) * eos;
      float *rs = rowst + q * 8;
      rs[0] = 1.f / (sp + 1e-6f); rs[1] = 0.f; rs[2] = 0.f; rs[3] = 0.f; rs[4] = m1; rs[5] = 1.f / s1; rs[6] = eos;
    }
  }
  // ---- phase 1, matched rows: a wave per row; lane owns the token columns lane + 64 k and keeps, over the wave's rows, the column
  //      sums phase 2b needs: counts of the four maps' set entries, the three maps' values, -sum(logit x sets) ----
  float pc[SEM_MAXK][8];
  int keep[SEM_MAXK];                       // the map bits of the wave's first matched row, for phase 3
#pragma unroll
  for (int k = 0; k < SEM_MAXK; ++k) {
    keep[k] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) pc[k][i] = 0.f;
  }
  for (int mi = wv; mi < nmatched; mi += SEM_NW) {
    const int q = mlist[mi], g = slot[q];
    const float *x = X + q * LS;
    float sp = 0.f, sm = 0.f, sr = 0.f, sl = 0.f, dp = 0.f, dm = 0.f, dr = 0.f, dl = 0.f, m1 = -INFINITY, s1 = 0.f;
    float vo[SEM_MAXK];                     // logit x (1 + other-entity bit)
#pragma unroll
    for (int k = 0; k < SEM_MAXK; ++k) {
      const int l = lane + 64 * k;
      vo[k] = -INFINITY;
      if (l < L) {
        const long o = (long)g * M.sg + l;
        const float c = mm[o], d = mr[o], e = ml[o];
        const float pm = mp[o] > 0.f ? 1.f : 0.f, mb = c > 0.f ? 1.f : 0.f, rb = d > 0.f ? 1.f : 0.f;
        const float lb = e > 0.f ? 1.f : 0.f, ob = mo[o] > 0.f ? 1.f : 0.f;
        const float v = x[l];
        sp += pm; sm += mb; sr += rb; sl += lb;
        dp += v * pm; dm += v * mb; dr += v * rb; dl += v * lb;
        vo[k] = v + v * ob;
        m1 = fmaxf(m1, vo[k]);
        pc[k][0] += pm; pc[k][1] += mb; pc[k][2] += rb; pc[k][3] += lb;
        pc[k][4] += c; pc[k][5] += d; pc[k][6] += e;
        pc[k][7] -= v * (pm + mb + rb + lb);
        if (mi == wv) keep[k] = (int)pm | (int)mb << 1 | (int)rb << 2 | (int)lb << 3 | (int)ob << 4;
      }
    }
    m1 = wave_max(m1);
#pragma unroll
    for (int k = 0; k < SEM_MAXK; ++k) s1 += lane + 64 * k < L ? __expf(vo[k] - m1) : 0.f;
    sp = wave_sum(sp); sm = wave_sum(sm); sr = wave_sum(sr); sl = wave_sum(sl);
    dp = wave_sum(dp); dm = wave_sum(dm); dr = wave_sum(dr); dl = wave_sum(dl);
    s1 = wave_sum(s1);
    if (lane == 0) {
      const float gq = sp > 0.f ? 1.f : 0.f;
      const float v = -dp / (sp + 1e-6f) - 0.2f * dm / (sm + 1e-6f) - 0.2f * dr / (sr + 1e-6f) - 0.1f * dl / (sl + 1e-6f) +
                      (m1 + __logf(s1));
      b2t_acc += v * gq;
      float *rs = rowst + q * 8;
      rs[0] = 1.f / (sp + 1e-6f); rs[1] = 0.2f / (sm + 1e-6f); rs[2] = 0.2f / (sr + 1e-6f); rs[3] = 0.1f / (sl + 1e-6f);
      rs[4] = m1; rs[5] = 1.f / s1; rs[6] = gq;
    }
  }
  PL(2);
  // ---- phase 2a: column log-sum-exp over all queries; lane = part * 8 + column of the wave's group of 8 ----
  for (int l0 = wv * 8; l0 < L; l0 += SEM_NW * 8) {
    const int l = l0 + (lane & 7), part = lane >> 3;
    const bool on = l < L;
    float mc = -INFINITY;
    if (on) for (int q = part; q < Q; q += 8) mc = fmaxf(mc, X[q * LS + l]);
    mc = fmaxf(mc, __shfl_xor(mc, 8)); mc = fmaxf(mc, __shfl_xor(mc, 16)); mc = fmaxf(mc, __shfl_xor(mc, 32));
    float sc = 0.f, us = 0.f;
    if (on) for (int q = part; q < Q; q += 8) {
      const float v = X[q * LS + l];
      sc += __expf(v - mc);
      us += slot[q] < 0 ? v : 0.f;
    }
    sc += __shfl_xor(sc, 8); sc += __shfl_xor(sc, 16); sc += __shfl_xor(sc, 32);
    us += __shfl_xor(us, 8); us += __shfl_xor(us, 16); us += __shfl_xor(us, 32);
    if (on && part == 0) { colst[l] = us; colst[2 * L + l] = mc; colst[3 * L + l] = sc; }
  }
  PL(3);
  // ---- phase 2b: text -> object; the waves' column sums meet in LDS, R waves at a time, in wave order; thread l owns column l ----
  float col[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nwaves_used = min(nmatched, SEM_NW);               // waves beyond the matched rows hold zeros
  for (int r0 = 0; r0 < nwaves_used; r0 += R) {
    if (wv >= r0 && wv < r0 + R) {
#pragma unroll
      for (int k = 0; k < SEM_MAXK; ++k) {
        const int l = lane + 64 * k;
        if (l < L) {
#pragma unroll
          for (int i = 0; i < 8; ++i) part_sums[((size_t)(wv - r0) * 8 + i) * L + l] = pc[k][i];
        }
      }
    }
    __syncthreads();
    if (tid < L) {
      for (int w = 0; w < R && r0 + w < nwaves_used; ++w) {
#pragma unroll
        for (int i = 0; i < 8; ++i) col[i] += part_sums[((size_t)w * 8 + i) * L + tid];
      }
    }
    __syncthreads();
  }
  if (nwaves_used == 0) __syncthreads();                       // (phase 2a's column statistics)
  float t2b_acc = 0.f;
  if (tid < L) {
    const int l = tid;
    float cp = col[0], post = col[7];
    bool anyp = col[0] > 0.f;
    const bool anym = col[1] > 0.f, anyr = col[2] > 0.f, anyl = col[3] > 0.f;
    if ((l == last || l == prev) && nmatched < Q) {     // every unmatched row counts this token as its positive
      cp += (float)(Q - nmatched);
      anyp = true;
      post -= colst[l];
    }
    const float mc = colst[2 * L + l], sc = colst[3 * L + l];
    const float nb = cp + col[4] + col[5] + col[6] + 1e-6f;
    float tm = eos;                                        // overwritten in this order (losses.py:550-556)
    if (l == last) tm = 1.f;
    if (anyp) tm = 1.f;
    if (anym) tm = 0.2f;
    if (anyr) tm = 0.2f;
    if (anyl) tm = 0.1f;
    if (l == prev) tm = 0.1f;
    const float gl = (anyp || anym || anyr || anyl) ? tm : 0.f;
    t2b_acc = (-__logf(nb + 1e-6f) / nb + post / nb + (mc + __logf(sc))) * gl;
    colst[l] = 1.f / nb; colst[L + l] = gl; colst[3 * L + l] = 1.f / sc;
  }
  __syncthreads();
  PL(4);
  // ---- phase 3: gradient, a wave per query row: the unmatched rows, then the wave's matched rows (the first one's map bits are still
  //      in registers from phase 1) ----
  for (int q = wv; q < Q; q += SEM_NW) {
    if (slot[q] >= 0) continue;
    const float *x = X + q * LS;
    const float *rs = rowst + q * 8;
    const float a0 = rs[0], m1 = rs[4], r1 = rs[5], gq = rs[6];
    float *out = grad0 + ((long)pb * Q + q) * L;
    for (int l = lane; l < L; l += 64) {
      const float v = x[l];
      const float pm = (l == last || l == prev) ? 1.f : 0.f;
      const float row_t = -pm * a0 + __expf(v - m1) * r1;
      const float col_t = -pm * colst[l] + __expf(v - colst[2 * L + l]) * colst[3 * L + l];
      out[l] = 0.5f * (gq * row_t + colst[L + l] * col_t);
    }
  }
  for (int mi = wv; mi < nmatched; mi += SEM_NW) {
    const int q = mlist[mi], g = slot[q];
    const float *x = X + q * LS;
    const float *rs = rowst + q * 8;
    const float a0 = rs[0], a1 = rs[1], a2 = rs[2], a3 = rs[3], m1 = rs[4], r1 = rs[5], gq = rs[6];
    float *out = grad0 + ((long)pb * Q + q) * L;
#pragma unroll
    for (int k = 0; k < SEM_MAXK; ++k) {
      const int l = lane + 64 * k;
      if (l >= L) continue;
      int bits = keep[k];
      if (mi != wv) {
        const long o = (long)g * M.sg + l;
        bits = (mp[o] > 0.f ? 1 : 0) | (mm[o] > 0.f ? 2 : 0) | (mr[o] > 0.f ? 4 : 0) | (ml[o] > 0.f ? 8 : 0) | (mo[o] > 0.f ? 16 : 0);
      }
      const float pm = (float)(bits & 1), mb = (float)(bits >> 1 & 1), rb = (float)(bits >> 2 & 1), lb = (float)(bits >> 3 & 1);
      const float ob = (float)(bits >> 4 & 1);
      const float v = x[l];
      const float row_t = -pm * a0 - mb * a1 - rb * a2 - lb * a3 + __expf(v + v * ob - m1) * r1 * (1.f + ob);
      const float col_t = -(pm + mb + rb + lb) * colst[l] + __expf(v - colst[2 * L + l]) * colst[3 * L + l];
      out[l] = 0.5f * (gq * row_t + colst[L + l] * col_t);
    }
  }
  PL(5);
  block_sum2<SEM_NW>(b2t_acc, t2b_acc, red);
  if (tid == 0) loss[pb] = (b2t_acc + t2b_acc) * 0.5f / num_boxes[0];
  PL(6);
}

// ---- seed objectness (losses.py:166-228, compute_points_obj_cls_loss_hard_topk) ------------------------------------------------
// One workgroup per scene.  Positives = for every real target slot g the `topk` seeds with the smallest value of
//   owned(k, g) ? sqrt(sum(((xyz_k - centre_g) / (size_g + 1e-6))^2) + 1e-6) : 100        (owner = instance id, background -> G - 1)
// that are not background; the reference takes them with torch.topk, whose choice among EQUAL values (an instance with fewer than
// `topk` seeds: the rest come from the 100s) is implementation-defined -- here the lowest seed index wins.  Then the sigmoid focal
// loss (alpha 0.25, gamma 2; losses.py:100-164) of all K seeds with weight 1/K, divided by the number of scenes, and its gradient.
constexpr int OBJ_MAXTOP = 8;

__global__ __launch_bounds__(256) void seed_objectness_fwd_kernel(const float *__restrict__ logits, const float *__restrict__ seed_xyz,
                                                                  const int *__restrict__ seed_inds,
                                                                  const long *__restrict__ instance_label, long npoints,
                                                                  const float *__restrict__ centre, long c_sg,
                                                                  const float *__restrict__ size, long s_sg,
                                                                  const float *__restrict__ mask, int B, int K, int G, int topk,
                                                                  float *__restrict__ loss, float *__restrict__ grad0) {
  extern __shared__ float obj_lds[];
  float *sx = obj_lds;                                   // K x 3
  int *owner = reinterpret_cast<int *>(obj_lds + 3 * (long)K);   // K: instance id, G - 1 for background, -1 - id kept in `fg`
  int *label = owner + K;                                // K
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int k = tid; k < K; k += 256) {
    const long inst = instance_label[(long)b * npoints + seed_inds[(long)b * K + k]];
    // owner: background seeds belong to slot G - 1; ids beyond the slots match nothing.  Bit 30 marks "not background".
    const int own = inst < 0 ? G - 1 : (inst < G ? (int)inst : G);
    owner[k] = own | (inst >= 0 ? (1 << 30) : 0);
    label[k] = 0;
    const float *x = seed_xyz + ((long)b * K + k) * 3;
    sx[3 * k] = x[0]; sx[3 * k + 1] = x[1]; sx[3 * k + 2] = x[2];
  }
  __syncthreads();
  for (int g = wv; g < G; g += 4) {
    if (mask[(long)b * G + g] == 0.f) continue;
    const float *c = centre + ((long)b * G + g) * c_sg, *z = size + ((long)b * G + g) * s_sg;
    const float c0 = c[0], c1 = c[1], c2 = c[2], z0 = z[0] + 1e-6f, z1 = z[1] + 1e-6f, z2 = z[2] + 1e-6f;
    // each lane keeps its own `topk` smallest (value, index) keys in ascending order; value bits of a positive float order as integers
    unsigned long long best[OBJ_MAXTOP];
#pragma unroll
    for (int i = 0; i < OBJ_MAXTOP; ++i) best[i] = ~0ull;
    for (int k = lane; k < K; k += 64) {
      float v = 100.f;
      if ((owner[k] & ~(1 << 30)) == g) {
        const float d0 = (sx[3 * k] - c0) / z0, d1 = (sx[3 * k + 1] - c1) / z1, d2 = (sx[3 * k + 2] - c2) / z2;
        v = sqrtf(d0 * d0 + d1 * d1 + d2 * d2 + 1e-6f);
      }
      unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)k;
#pragma unroll
      for (int i = 0; i < OBJ_MAXTOP; ++i) {
        if (i < topk && key < best[i]) { const unsigned long long t = best[i]; best[i] = key; key = t; }
      }
    }
    for (int r = 0; r < topk; ++r) {
      unsigned long long m = best[0];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(m, o);
        m = other < m ? other : m;
      }
      if (m == ~0ull) break;                             // fewer than topk seeds in the scene
      if (best[0] == m) {                                // exactly one lane holds this key (the index is part of it)
        label[(int)(m & 0xffffffffu)] = 1;
#pragma unroll
        for (int i = 0; i + 1 < OBJ_MAXTOP; ++i) best[i] = best[i + 1];
        best[OBJ_MAXTOP - 1] = ~0ull;
      }
    }
  }
  __syncthreads();
  const float scale = 1.f / (float)(K > 1 ? K : 1) / (float)B;
  float acc = 0.f, dummy = 0.f;
  for (int k = tid; k < K; k += 256) {
    const bool pos = label[k] != 0 && (owner[k] & (1 << 30));
    const float x = logits[(long)b * K + k];
    const float p = 1.f / (1.f + expf(-x));
    const float sp = fmaxf(pos ? -x : x, 0.f) + log1pf(expf(-fabsf(x)));      // softplus(-x) for a positive, softplus(x) otherwise
    float f, df;
    if (pos) {
      const float q = 1.f - p;
      f = 0.25f * q * q * sp;
      df = -0.25f * q * q * (2.f * p * sp + q);
    } else {
      f = 0.75f * p * p * sp;
      df = 0.75f * p * p * (2.f * (1.f - p) * sp + p);
    }
    acc += f;
    grad0[(long)b * K + k] = df * scale;
  }
  block_sum2(acc, dummy, red);
  if (tid == 0) loss[b] = acc * scale;
}

// ---- the padded targets, valid slots first (eda_amd/losses.py compact_targets: what the reference's boolean indexing per scene does,
// losses.py:660-690), as one launch: workgroup (part, b) orders scene b's slots (valid ones first, order kept) and moves its share of
// the rows of up to 8 tensors; rows beyond the scene's count are zeroed (nothing reads them).  Also the per-scene counts, the valid
// mask of the compacted layout and the batch's box count (the losses' normaliser before any all-reduce).
constexpr int CT_MAXT = 8;
struct CompactDesc { const unsigned *src; unsigned *dst; long src_sb, src_sg, dst_sg, dst_off; int words; };
struct CompactArgs { CompactDesc d[CT_MAXT]; int n; };

__global__ __launch_bounds__(256) void compact_targets_kernel(const float *__restrict__ mask, const CompactArgs A, int B, int G,
                                                              int *__restrict__ ntargets, unsigned char *__restrict__ valid,
                                                              float *__restrict__ num_boxes) {
  extern __shared__ int ct_order[];                      // [G]
  __shared__ int s_nt, s_all[4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (wv == 0) {
    int nt = 0;
    for (int g0 = 0; g0 < G; g0 += 64) {
      const int g = g0 + lane;
      const bool v = g < G && mask[(long)b * G + g] > 0.f;
      const unsigned long long bal = __ballot(v);
      if (v) ct_order[nt + __popcll(bal & ((1ull << lane) - 1))] = g;
      nt += __popcll(bal);
    }
    if (lane == 0) s_nt = nt;
  }
  if (blockIdx.x == 0 && b == 0) {                        // box count of the whole batch
    int c = 0;
    for (long i = tid; i < (long)B * G; i += 256) c += mask[i] > 0.f ? 1 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) s_all[wv] = c;
  }
  __syncthreads();
  const int nt = s_nt;
  if (blockIdx.x == 0) {
    if (tid == 0) ntargets[b] = nt;
    for (int g = tid; g < G; g += 256) valid[(long)b * G + g] = g < nt ? 1 : 0;
    if (b == 0 && tid == 0) num_boxes[0] = (float)(s_all[0] + s_all[1] + s_all[2] + s_all[3]);
  }
  const int rows_per = (G + gridDim.x - 1) / gridDim.x;
  const int j0 = blockIdx.x * rows_per, j1 = min(G, j0 + rows_per);
  for (int t = 0; t < A.n; ++t) {
    const CompactDesc d = A.d[t];
    const long n = (long)(j1 - j0) * d.words;
    for (long i = tid; i < n; i += 256) {
      const int j = j0 + (int)(i / d.words), w = (int)(i % d.words);
      d.dst[((long)b * G + j) * d.dst_sg + d.dst_off + w] = j < nt ? d.src[(long)b * d.src_sb + (long)ct_order[j] * d.src_sg + w] : 0u;
    }
  }
}

// ---- the final weighted sum (losses.py:716-738): per-head and total values of the four criterion losses, the objectness sum, and
//   loss = w_obj * objectness + inv * (w[0] ce + w[1] bbox + w[2] giou + w[3] sem)
struct CombineArgs { const float *rows[4]; int parts[4]; float w[4]; float inv, w_obj; };

__global__ __launch_bounds__(256) void loss_combine_fwd_kernel(const CombineArgs A, const float *__restrict__ obj, int P, int B,
                                                               float *__restrict__ per_head, float *__restrict__ totals,
                                                               float *__restrict__ loss) {
  __shared__ float ph[4 * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int pair = wv; pair < 4 * P; pair += 4) {
    const int j = pair / P, p = pair - j * P;
    float s = 0.f;
    if (A.rows[j]) {
      const int n = B * A.parts[j];
      for (int i = lane; i < n; i += 64) s += A.rows[j][(long)p * n + i];
      s = wave_sum(s);
    }
    if (lane == 0) { ph[pair] = s; per_head[pair] = s; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
    for (int j = 0; j < 4; ++j) {
      t[j] = 0.f;
      for (int p = 0; p < P; ++p) t[j] += ph[j * P + p];
      totals[j] = t[j];
    }
    float qp = 0.f;
    if (obj) for (int b = 0; b < B; ++b) qp += obj[b];
    totals[4] = qp;
    loss[0] = A.w_obj * qp + A.inv * (A.w[0] * t[0] + A.w[1] * t[1] + A.w[2] * t[2] + A.w[3] * t[3]);
  }
}

struct CombineGrads { float *rows[4]; int n[4]; float *obj; int B; };

__global__ __launch_bounds__(256) void loss_combine_bwd_kernel(const float *__restrict__ g, const CombineArgs A, const CombineGrads D) {
  const float up = g[0];
  const int j = blockIdx.x;
  if (j < 4) {
    if (!D.rows[j]) return;
    const float v = up * (A.inv * A.w[j]);
    for (int i = threadIdx.x; i < D.n[j]; i += 256) D.rows[j][i] = v;
  } else if (D.obj) {
    const float v = up * A.w_obj;
    for (int i = threadIdx.x; i < D.B; i += 256) D.obj[i] = v;
  }
}

}  // namespace

extern "C" int eda_match_slots_i64(const int *assign, const int *ntargets, int PB, int B, int Q, int G, long *tq, void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && B > 0 && PB % B == 0 && Q > 0 && G > 0, "bad dimension");
  if (PB == 0) return 0;
  EDA_CHECK_ARG(assign && ntargets && tq, "null pointer");
  hipLaunchKernelGGL(match_slots_kernel, dim3((unsigned)PB), dim3(256), 0, (hipStream_t)stream_, assign, ntargets, B, Q, G, tq);
  EDA_CHECK_LAUNCH();
  return 0;
}

// logits (PB, Q, C) dense, pred (PB, Q, 6) dense, tgt_boxes (B, G, 6) dense, pmap (B, G, >= C) with slot stride pm_sg (soft-token
// matching) or NULL with labels (B, G) int64; cost (PB, Q, G) dense, every element written
extern "C" int eda_match_cost_f32(const float *logits, const float *pred, const float *tgt_boxes, const float *pmap, long pm_sg,
                                  const long *labels, const int *ntargets, int PB, int B, int Q, int G, int C, float w_class,
                                  float w_bbox, float w_giou, float *cost, void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && B > 0 && PB % B == 0 && Q > 0 && G > 0 && C > 0 && C <= 64 * CE_MAXK, "bad dimension (at most 512 classes)");
  if (PB == 0) return 0;
  EDA_CHECK_ARG(logits && pred && tgt_boxes && (pmap || labels) && ntargets && cost, "null pointer");
  const long rows = (long)PB * Q;
  hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, logits, pred,
                     tgt_boxes, pmap, pm_sg, labels, ntargets, PB, B, Q, G, C, w_class, w_bbox, w_giou, cost);
  EDA_CHECK_LAUNCH();
  return 0;
}

// maps: the four (B, G, >= C) token maps (positive, modify, pronoun, relation) with element strides (map_sb, map_sg, 1) and their
// weights w[4]; logits (PB, Q, C) dense; loss (PB, S): S partial sums per scene, one per chunk of eda_pos_align_chunk(Q, S) query
// rows (S >= 1 spreads a scene over S workgroups; the chunks of a scene add up to its loss); grad0 (PB, Q, C)
extern "C" int eda_pos_align_chunk(int Q, int S) { return S > 0 ? ((Q + S - 1) / S + 3) / 4 * 4 : 0; }
extern "C" int eda_pos_align_fwd_f32(const float *logits, const long *tq, const float *const *maps, const float *w, long map_sb,
                                     long map_sg, const float *num_boxes, int PB, int B, int Q, int G, int C, int S, float eos,
                                     float *loss, float *grad0, void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && B > 0 && PB % B == 0 && Q > 0 && G > 0 && C > 0 && C <= 64 * CE_MAXK, "bad dimension (at most 512 classes)");
  EDA_CHECK_ARG(S >= 1 && S <= 64, "1..64 chunks per scene");
  if (PB == 0) return 0;
  EDA_CHECK_ARG(logits && tq && maps && w && num_boxes && loss && grad0, "null pointer");
  PosAlignMaps M;
  for (int i = 0; i < 4; ++i) { EDA_CHECK_ARG(maps[i], "null pointer"); M.m[i] = maps[i]; M.w[i] = w[i]; }
  M.sb = map_sb; M.sg = map_sg;
  hipLaunchKernelGGL(pos_align_fwd_kernel, dim3((unsigned)S, (unsigned)PB), dim3(256), 0, (hipStream_t)stream_, logits, tq, M, num_boxes,
                     B, Q, G, C, eda_pos_align_chunk(Q, S), eos, loss, grad0);
  EDA_CHECK_LAUNCH();
  return 0;
}

// out[pb][i] = g0[pb][i] * w[pb][min(i / per_part, S - 1)] / num_boxes[0]; per = elements per scene, per_part = elements per part
// (both multiples of 4; S = 1, per_part = per: one weight per scene), 16-byte aligned buffers
extern "C" int eda_scale_by_scene_f32(const float *g0, const float *w, const float *num_boxes, int PB, long per, long per_part, int S,
                                      float *out, void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && per >= 0 && per % 4 == 0, "elements per scene must be a multiple of 4");
  if (PB == 0 || per == 0) return 0;
  EDA_CHECK_ARG(S >= 1 && per_part > 0 && per_part % 4 == 0 && per_part * S >= per, "bad parts");
  EDA_CHECK_ARG(g0 && w && num_boxes && out, "null pointer");
  const long total = (long)PB * per;
  hipLaunchKernelGGL(scale_by_scene_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, g0, w,
                     num_boxes, per, per_part, S, total, out);
  EDA_CHECK_LAUNCH();
  return 0;
}

// LDS of the alignment kernel with room for R waves' column sums (R = 1: the least it runs with)
static size_t sem_align_lds(int Q, int L, int R) {
  return sizeof(float) * ((size_t)Q * sem_row_stride(L) + (size_t)Q * 8 + (size_t)L * 4 + 2 * (size_t)Q + (size_t)R * L * 8);
}
extern "C" size_t eda_sem_align_lds_bytes(int Q, int L) { return sem_align_lds(Q, L, 1); }
extern "C" int eda_sem_align_supported(int Q, int L) { return Q > 0 && L > 0 && L <= 64 * SEM_MAXK && sem_align_lds(Q, L, 1) <= SEM_LDS_MAX; }

// maps: positive, modify, pronoun, other-entity, relation (B, G, >= L) with element strides (map_sb, map_sg, 1); logits (PB, Q, L)
// dense = proj_queries . proj_tokens^T / temperature; attn_mask (B, L) int64 (1 = token); loss (PB); grad0 (PB, Q, L)
extern "C" int eda_sem_align_fwd_f32(const float *logits, const long *tq, const float *const *maps, long map_sb, long map_sg,
                                     const long *attn_mask, const float *num_boxes, int PB, int B, int Q, int G, int L, float eos,
                                     float *loss, float *grad0, void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && B > 0 && PB % B == 0 && G > 0, "bad dimension");
  EDA_CHECK_ARG(eda_sem_align_supported(Q, L), "Q x L beyond the LDS tile (eda_sem_align_supported)");
  if (PB == 0) return 0;
  EDA_CHECK_ARG(logits && tq && maps && attn_mask && num_boxes && loss && grad0, "null pointer");
  SemMaps M;
  for (int i = 0; i < 5; ++i) EDA_CHECK_ARG(maps[i], "null pointer");
  M.pos = maps[0]; M.modi = maps[1]; M.pron = maps[2]; M.other = maps[3]; M.rel = maps[4];
  M.sb = map_sb; M.sg = map_sg;
  int R = SEM_NW;
  while (R > 1 && sem_align_lds(Q, L, R) > SEM_LDS_MAX) --R;
  const size_t lds = sem_align_lds(Q, L, R);
  EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(&sem_align_fwd_kernel), SEM_LDS_MAX));
  hipLaunchKernelGGL(sem_align_fwd_kernel, dim3((unsigned)PB), dim3(64 * SEM_NW), lds, (hipStream_t)stream_, logits, tq, M, attn_mask,
                     num_boxes, B, Q, G, L, R, eos, loss, grad0);
  EDA_CHECK_LAUNCH();
  return 0;
}

// logits (B, K) dense; seed_xyz (B, K, 3) dense; seed_inds (B, K) int32 into instance_label (B, npoints) int64 (< 0 = background);
// centre / size (B, G, >= 3) with slot strides c_sg / s_sg (elements); mask (B, G) float (0 = padded slot); loss (B): the scene's
// share of the reference's scalar (their sum IS compute_points_obj_cls_loss_hard_topk); grad0 (B, K) = d sum(loss) / d logits
extern "C" size_t eda_seed_objectness_lds_bytes(int K) { return (size_t)K * 20; }
extern "C" int eda_seed_objectness_fwd_f32(const float *logits, const float *seed_xyz, const int *seed_inds, const long *instance_label,
                                           long npoints, const float *centre, long c_sg, const float *size, long s_sg, const float *mask,
                                           int B, int K, int G, int topk, float *loss, float *grad0, void *stream_) {
  EDA_CHECK_ARG(B >= 0 && K > 0 && G > 0 && npoints > 0, "bad dimension");
  EDA_CHECK_ARG(topk >= 1 && topk <= OBJ_MAXTOP, "topk must be 1..8");
  EDA_CHECK_ARG(eda_seed_objectness_lds_bytes(K) <= 150 * 1024, "too many seeds for the LDS tile (at most 7680)");
  if (B == 0) return 0;
  EDA_CHECK_ARG(logits && seed_xyz && seed_inds && instance_label && centre && size && mask && loss && grad0, "null pointer");
  EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(&seed_objectness_fwd_kernel), 150 * 1024));
  hipLaunchKernelGGL(seed_objectness_fwd_kernel, dim3((unsigned)B), dim3(256), eda_seed_objectness_lds_bytes(K), (hipStream_t)stream_,
                     logits, seed_xyz, seed_inds, instance_label, npoints, centre, c_sg, size, s_sg, mask, B, K, G, topk, loss, grad0);
  EDA_CHECK_LAUNCH();
  return 0;
}

// mask (B, G) float (> 0 = a real target); n <= 8 tensors of (B, G, words) 4-byte words: src[t] with element strides (src_sb[t],
// src_sg[t], 1) in WORDS, moved to dst[t] (B, G, dst_sg[t]) at word offset dst_off[t] of each row (two sources can fill one row:
// centre | size -> a box); ntargets (B) int32, valid (B, G) bytes (slot < count), num_boxes: one float = all real targets of the batch
extern "C" int eda_compact_targets(const float *mask, int n, const void *const *src, const long *src_sb, const long *src_sg,
                                   void *const *dst, const long *dst_sg, const long *dst_off, const int *words, int B, int G,
                                   int *ntargets, unsigned char *valid, float *num_boxes, void *stream_) {
  EDA_CHECK_ARG(B >= 0 && G > 0 && G <= 8192 && n >= 0 && n <= CT_MAXT, "bad dimension (at most 8 tensors, 8192 slots)");
  if (B == 0) return 0;
  EDA_CHECK_ARG(mask && ntargets && valid && num_boxes && (n == 0 || (src && src_sb && src_sg && dst && dst_sg && dst_off && words)),
                "null pointer");
  CompactArgs A;
  A.n = n;
  for (int t = 0; t < n; ++t) {
    EDA_CHECK_ARG(src[t] && dst[t] && words[t] > 0 && dst_off[t] >= 0 && dst_off[t] + words[t] <= dst_sg[t], "bad tensor description");
    A.d[t] = CompactDesc{static_cast<const unsigned *>(src[t]), static_cast<unsigned *>(dst[t]), src_sb[t], src_sg[t], dst_sg[t],
                         dst_off[t], words[t]};
  }
  hipLaunchKernelGGL(compact_targets_kernel, dim3(16, (unsigned)B), dim3(256), sizeof(int) * (size_t)G, (hipStream_t)stream_, mask, A, B,
                     G, ntargets, valid, num_boxes);
  EDA_CHECK_LAUNCH();
  return 0;
}

// rows[j] (P * B, parts[j]) per-scene (partial) sums of loss j = ce, bbox, giou, sem-align, heads-major, or NULL (that loss is off);
// obj (B) per-scene objectness shares or NULL; per_head (4, P), totals (5: the four + objectness), loss (1):
//   loss = w_obj * sum(obj) + inv * (w[0] ce + w[1] bbox + w[2] giou + w[3] sem)
extern "C" int eda_loss_combine_fwd_f32(const float *const *rows, const int *parts, const float *obj, const float *w, float inv,
                                        float w_obj, int P, int B, float *per_head, float *totals, float *loss, void *stream_) {
  EDA_CHECK_ARG(P > 0 && P <= 64 && B > 0, "bad dimension (at most 64 heads)");
  EDA_CHECK_ARG(rows && parts && w && per_head && totals && loss, "null pointer");
  CombineArgs A;
  for (int j = 0; j < 4; ++j) { A.rows[j] = rows[j]; A.parts[j] = parts[j]; A.w[j] = w[j]; EDA_CHECK_ARG(!rows[j] || parts[j] > 0, "bad parts"); }
  A.inv = inv; A.w_obj = w_obj;
  hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, A, obj, P, B, per_head, totals, loss);
  EDA_CHECK_LAUNCH();
  return 0;
}

// the gradient of eda_loss_combine's `loss` given its upstream gradient g (one float on the device): d_rows[j] (n[j] elements, or
// NULL) filled with g * inv * w[j], d_obj (B, or NULL) with g * w_obj
extern "C" int eda_loss_combine_bwd_f32(const float *g, const float *w, float inv, float w_obj, float *const *d_rows, const int *n,
                                        float *d_obj, int B, void *stream_) {
  EDA_CHECK_ARG(g && w && d_rows && n && B >= 0, "null pointer");
  CombineArgs A;
  CombineGrads D;
  for (int j = 0; j < 4; ++j) { A.rows[j] = nullptr; A.parts[j] = 0; A.w[j] = w[j]; D.rows[j] = d_rows[j]; D.n[j] = n[j]; }
  A.inv = inv; A.w_obj = w_obj;
  D.obj = d_obj; D.B = B;
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(5), dim3(256), 0, (hipStream_t)stream_, g, A, D);
  EDA_CHECK_LAUNCH();
  return 0;
}

#ifdef EDA_LOSS_PROFILE
extern "C" __attribute__((visibility("default"))) int eda_loss_profile_read(unsigned long long *out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(loss_prof), sizeof(unsigned long long) * (size_t)n) != hipSuccess;
}
#endif
