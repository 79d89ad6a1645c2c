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

// workgroup sum of two values (256 threads)
__device__ __forceinline__ void block_sum2(float &x, float &y, float *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * w] = x; red[2 * w + 1] = y; }
  __syncthreads();
  x = red[0] + red[2] + red[4] + red[6];
  y = red[1] + red[3] + red[5] + red[7];
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

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

// position-aligned soft-token cross entropy (loss_pos_align, losses.py:396-460): one workgroup per pb, a wave per query row;
// loss[pb] = sum_q ce / num_boxes, grad0 (PB, Q, C) = d(sum_q ce) / d(logits)
__global__ __launch_bounds__(256) void pos_align_fwd_kernel(const float *__restrict__ logits, const long *__restrict__ tq,
                                                            const PosAlignMaps M, const float *__restrict__ num_boxes,
                                                            int B, int Q, int G, int C, float eos, float *__restrict__ loss,
                                                            float *__restrict__ grad0) {
  __shared__ float red[8];
  const int pb = blockIdx.x, b = pb % B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int q = wave; q < Q; q += 4) {
    const long row = (long)pb * Q + q;
    float x[CE_MAXK], m, s;
    row_softmax(logits + row * C, C, lane, x, m, s);
    const float lse = m + __logf(s);
    const long g = tq[row];
    const bool matched = g >= 0 && g < G;
    float sim[CE_MAXK], S = 0.f, ce = 0.f;
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
      S += v;
    }
    S = wave_sum(S);
    ce = wave_sum(ce);
    const float qw = matched ? 1.f : eos;
    acc += ce * qw;
#pragma unroll
    for (int k = 0; k < CE_MAXK; ++k) {
      const int c = lane + 64 * k;
      if (c < C) grad0[row * C + c] = qw * (__expf(x[k] - lse) * S - sim[k]);
    }
  }
  float dummy = 0.f;
  if (lane != 0) acc = 0.f;                 // (every lane of a wave holds the same sum)
  block_sum2(acc, dummy, red);
  if (threadIdx.x == 0) loss[pb] = acc / num_boxes[0];
}

// out[pb][i] = g0[pb][i] * w[pb] / num_boxes  (the backward of every per-scene loss whose gradient was formed in the forward)
__global__ __launch_bounds__(256) void scale_by_scene_kernel(const float *__restrict__ g0, const float *__restrict__ w,
                                                             const float *__restrict__ num_boxes, long per, long total,
                                                             float *__restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  const float sc = w[i / per] / num_boxes[0];           // (per % 4 == 0: the four elements share a scene)
  const float4 v = *reinterpret_cast<const float4 *>(g0 + i);
  *reinterpret_cast<float4 *>(out + i) = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
}

struct SemMaps { const float *pos, *modi, *pron, *other, *rel; long sb, sg; };

// semantic-alignment contrastive loss (loss_sem_align, losses.py:499-608): one workgroup per pb with the scene's (Q, L) logits in
// LDS; phase 1 per query row (object -> text), phase 2 per token column (text -> object), phase 3 the gradient.
// loss[pb] = (b2t + t2b) / 2 / num_boxes; grad0 (PB, Q, L) = d(b2t + t2b) / 2 / d(logits)
__global__ __launch_bounds__(256) void sem_align_fwd_kernel(const float *__restrict__ logits, const long *__restrict__ tq,
                                                            const SemMaps M, const long *__restrict__ attn_mask,
                                                            const float *__restrict__ num_boxes, int B, int Q, int G, int L,
                                                            float eos, float *__restrict__ loss, float *__restrict__ grad0) {
  extern __shared__ __attribute__((aligned(16))) float sem_smem[];
  const int LS = L + 1;
  float *X = sem_smem;                      // [Q][LS]
  float *rowst = X + (size_t)Q * LS;        // [Q][8]: sp, sm, sr, srel, m1, s1, gq, -
  float *colst = rowst + (size_t)Q * 8;     // [L][4]: nb, gl, mcol, scol
  int *slot = reinterpret_cast<int *>(colst + (size_t)L * 4);       // [Q]: matched target slot or -1
  __shared__ float red[8];
  __shared__ int lastprev[2];
  const int pb = blockIdx.x, b = pb % B, tid = threadIdx.x;
  for (int i = tid; i < Q * L; i += 256) {
    const int q = i / L, l = i - q * L;
    X[q * LS + l] = logits[((long)pb * Q + q) * L + l];
  }
  for (int q = tid; q < Q; q += 256) {
    const long g = tq[(long)pb * Q + q];
    slot[q] = (g >= 0 && g < G) ? (int)g : -1;
  }
  if (tid < 64) {                           // number of real tokens -> the "not mentioned" token and the one before it
    long n = 0;
    for (int l = tid; l < L; l += 64) n += attn_mask[(long)b * L + l];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if (tid == 0) {
      lastprev[0] = (int)((((n - 1) % L) + L) % L);
      lastprev[1] = (int)((((n - 2) % L) + L) % L);
    }
  }
  __syncthreads();
  const int last = lastprev[0], prev = lastprev[1];
  const float *mp = M.pos + (long)b * M.sb, *mm = M.modi + (long)b * M.sb, *mr = M.pron + (long)b * M.sb;
  const float *mo = M.other + (long)b * M.sb, *ml = M.rel + (long)b * M.sb;
  // ---- phase 1: object -> text, one thread per query row ----
  float b2t_acc = 0.f;
  for (int q = tid; q < Q; q += 256) {
    const int g = slot[q];
    const float *x = X + q * LS;
    float sp = 0.f, sm = 0.f, sr = 0.f, sl = 0.f, dp = 0.f, dm = 0.f, dr = 0.f, dl = 0.f, m1 = -INFINITY;
    for (int l = 0; l < L; ++l) {
      float pm, mb = 0.f, rb = 0.f, lb = 0.f, ob = 0.f;
      if (g >= 0) {
        const long o = (long)g * M.sg + l;
        pm = mp[o] > 0.f ? 1.f : 0.f; mb = mm[o] > 0.f ? 1.f : 0.f; rb = mr[o] > 0.f ? 1.f : 0.f;
        lb = ml[o] > 0.f ? 1.f : 0.f; ob = mo[o] > 0.f ? 1.f : 0.f;
      } else {
        pm = (l == last || l == prev) ? 1.f : 0.f;
      }
      const float v = x[l];
      sp += pm; sm += mb; sr += rb; sl += lb;
      dp += v * pm; dm += v * mb; dr += v * rb; dl += v * lb;
      m1 = fmaxf(m1, v + v * ob);
    }
    float s1 = 0.f;
    for (int l = 0; l < L; ++l) {
      const float ob = (g >= 0 && mo[(long)g * M.sg + l] > 0.f) ? 1.f : 0.f;
      s1 += __expf(x[l] + x[l] * ob - m1);
    }
    const float gq = (sp > 0.f ? 1.f : 0.f) * (g >= 0 ? 1.f : eos);
    const float v = -dp / (sp + 1e-6f) - 0.2f * dm / (sm + 1e-6f) - 0.2f * dr / (sr + 1e-6f) - 0.1f * dl / (sl + 1e-6f) +
                    (m1 + __logf(s1));
    b2t_acc += v * gq;
    float *rs = rowst + q * 8;
    rs[0] = sp; rs[1] = sm; rs[2] = sr; rs[3] = sl; rs[4] = m1; rs[5] = s1; rs[6] = gq;
  }
  __syncthreads();
  // ---- phase 2: text -> object, one thread per token column ----
  float t2b_acc = 0.f;
  for (int l = tid; l < L; l += 256) {
    float cp = 0.f, vm = 0.f, vr = 0.f, vl = 0.f, post = 0.f, mc = -INFINITY;
    bool anyp = false, anym = false, anyr = false, anyl = false;
    for (int q = 0; q < Q; ++q) {
      const int g = slot[q];
      const float v = X[q * LS + l];
      float pm, mb = 0.f, rb = 0.f, lb = 0.f;
      if (g >= 0) {
        const long o = (long)g * M.sg + l;
        const float a = mp[o], c = mm[o], d = mr[o], e = ml[o];
        pm = a > 0.f ? 1.f : 0.f; mb = c > 0.f ? 1.f : 0.f; rb = d > 0.f ? 1.f : 0.f; lb = e > 0.f ? 1.f : 0.f;
        vm += c; vr += d; vl += e;
      } else {
        pm = (l == last || l == prev) ? 1.f : 0.f;
      }
      cp += pm;
      anyp |= pm > 0.f; anym |= mb > 0.f; anyr |= rb > 0.f; anyl |= lb > 0.f;
      post -= v * (pm + mb + rb + lb);
      mc = fmaxf(mc, v);
    }
    float sc = 0.f;
    for (int q = 0; q < Q; ++q) sc += __expf(X[q * LS + l] - mc);
    const float nb = cp + vm + vr + vl + 1e-6f;
    float tm = eos;                                        // overwritten in this order (losses.py:550-556)
    if (l == last) tm = 1.f;
    if (anyp) tm = 1.f;
    if (anym) tm = 0.2f;
    if (anyr) tm = 0.2f;
    if (anyl) tm = 0.1f;
    if (l == prev) tm = 0.1f;
    const float gl = (anyp || anym || anyr || anyl) ? tm : 0.f;
    t2b_acc += (-__logf(nb + 1e-6f) / nb + post / nb + (mc + __logf(sc))) * gl;
    float *cs = colst + l * 4;
    cs[0] = nb; cs[1] = gl; cs[2] = mc; cs[3] = sc;
  }
  __syncthreads();
  // ---- phase 3: gradient, one thread per query row ----
  for (int q = tid; q < Q; q += 256) {
    const int g = slot[q];
    const float *x = X + q * LS;
    const float *rs = rowst + q * 8;
    const float sp = rs[0], sm = rs[1], sr = rs[2], sl = rs[3], m1 = rs[4], s1 = rs[5], gq = rs[6];
    float *out = grad0 + ((long)pb * Q + q) * L;
    for (int l = 0; l < L; ++l) {
      float pm, mb = 0.f, rb = 0.f, lb = 0.f, ob = 0.f;
      if (g >= 0) {
        const long o = (long)g * M.sg + l;
        pm = mp[o] > 0.f ? 1.f : 0.f; mb = mm[o] > 0.f ? 1.f : 0.f; rb = mr[o] > 0.f ? 1.f : 0.f;
        lb = ml[o] > 0.f ? 1.f : 0.f; ob = mo[o] > 0.f ? 1.f : 0.f;
      } else {
        pm = (l == last || l == prev) ? 1.f : 0.f;
      }
      const float v = x[l];
      const float *cs = colst + l * 4;
      const float row_t = -pm / (sp + 1e-6f) - 0.2f * mb / (sm + 1e-6f) - 0.2f * rb / (sr + 1e-6f) - 0.1f * lb / (sl + 1e-6f) +
                          __expf(v + v * ob - m1) / s1 * (1.f + ob);
      const float col_t = -(pm + mb + rb + lb) / cs[0] + __expf(v - cs[2]) / cs[3];
      out[l] = 0.5f * (gq * row_t + cs[1] * col_t);
    }
  }
  block_sum2(b2t_acc, t2b_acc, red);
  if (tid == 0) loss[pb] = (b2t_acc + t2b_acc) * 0.5f / num_boxes[0];
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
// weights w[4]; logits (PB, Q, C) dense; loss (PB); grad0 (PB, Q, C)
extern "C" int eda_pos_align_fwd_f32(const float *logits, const long *tq, const float *const *maps, const float *w, long map_sb,
                                     long map_sg, const float *num_boxes, int PB, int B, int Q, int G, int C, float eos,
                                     float *loss, float *grad0, void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && B > 0 && PB % B == 0 && Q > 0 && G > 0 && C > 0 && C <= 64 * CE_MAXK, "bad dimension (at most 512 classes)");
  if (PB == 0) return 0;
  EDA_CHECK_ARG(logits && tq && maps && w && num_boxes && loss && grad0, "null pointer");
  PosAlignMaps M;
  for (int i = 0; i < 4; ++i) { EDA_CHECK_ARG(maps[i], "null pointer"); M.m[i] = maps[i]; M.w[i] = w[i]; }
  M.sb = map_sb; M.sg = map_sg;
  hipLaunchKernelGGL(pos_align_fwd_kernel, dim3((unsigned)PB), dim3(256), 0, (hipStream_t)stream_, logits, tq, M, num_boxes, B, Q,
                     G, C, eos, loss, grad0);
  EDA_CHECK_LAUNCH();
  return 0;
}

// out[pb][:] = g0[pb][:] * w[pb] / num_boxes[0]; per = elements per scene (a multiple of 4), 16-byte aligned buffers
extern "C" int eda_scale_by_scene_f32(const float *g0, const float *w, const float *num_boxes, int PB, long per, float *out,
                                      void *stream_) {
  EDA_CHECK_ARG(PB >= 0 && per >= 0 && per % 4 == 0, "elements per scene must be a multiple of 4");
  if (PB == 0 || per == 0) return 0;
  EDA_CHECK_ARG(g0 && w && num_boxes && out, "null pointer");
  const long total = (long)PB * per;
  hipLaunchKernelGGL(scale_by_scene_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, g0, w,
                     num_boxes, per, total, out);
  EDA_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t eda_sem_align_lds_bytes(int Q, int L) { return sizeof(float) * ((size_t)Q * (L + 1) + (size_t)Q * 8 + (size_t)L * 4 + Q); }
extern "C" int eda_sem_align_supported(int Q, int L) { return Q > 0 && L > 0 && eda_sem_align_lds_bytes(Q, L) <= 150 * 1024; }

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
  const size_t lds = eda_sem_align_lds_bytes(Q, L);
  EDA_CHECK_HIP(eda_set_max_dynamic_lds(reinterpret_cast<const void *>(&sem_align_fwd_kernel), 150 * 1024));
  hipLaunchKernelGGL(sem_align_fwd_kernel, dim3((unsigned)PB), dim3(256), lds, (hipStream_t)stream_, logits, tq, M, attn_mask,
                     num_boxes, B, Q, G, L, eos, loss, grad0);
  EDA_CHECK_LAUNCH();
  return 0;
}
