// gemm.hip -- the row-GEMM family of the path: Y = f(X) W^T (forward of every 1x1 convolution /
// linear layer) and dX = dY W (their input gradients), fp32 in / fp32 accumulate on
// v_mfma_f32_16x16x4_f32, with the element-wise work of the neighbouring layers folded into the
// operand staging (prologue) and the accumulator write-out (epilogue):
//
//   prologues of the ROW operand X (R rows, contraction length K)
//     X_PLAIN    X as stored
//     X_BNRELU   relu(X * scale[k] + shift[k])        BatchNorm+ReLU of the previous SharedMLP layer
//                                                     (pointnet2/pytorch_utils.py:67-120): the
//                                                     activated tensor is never written to HBM
//     X_GATHER   [ feats[idx] | (xyz[idx]-centre)/r ] QueryAndGroup (pointnet2/pointnet2_utils.py:
//                                                     317-376): neighbourhood rows are gathered
//                                                     straight into LDS, the grouped tensor never
//                                                     exists (columns are permuted: features first,
//                                                     so that feature rows stay 16-byte aligned; the
//                                                     weight's columns are permuted to match)
//   epilogues
//     E_PLAIN    + bias, optional ReLU                                  linear / conv1x1 forward, dX
//     E_STATS    column sums  (sum y, sum y^2) in fp64 accumulators; the LAST workgroup turns them
//                into mean / rstd / scale / shift and the running statistics (train-mode BatchNorm)
//     E_MASK     gy = acc * (z*scale+shift > 0)  and  (sum gy, sum gy*xhat): ReLU backward of the
//                PREVIOUS layer and the two reductions its BatchNorm backward needs
//     E_SCATTER  atomicAdd into dfeats[b, idx[row], col]                backward of the gather
//
// Tiling.  MFMA operand roles are swapped with respect to the usual picture: the WEIGHT tile is the
// A operand (m = output column) and the ROW tile the B operand (n = row), so a lane's four
// accumulator registers are four CONSECUTIVE output columns of one row: 16-byte stores, column
// reductions = DPP reductions over the 16 lanes of a row.  A workgroup of 4 waves owns a
// (64*WR) x (16*WC) output tile, wave w the rows [16*WR*w, 16*WR*(w+1)) and ALL columns, i.e.
// WR + WC ds_read_b128 per 4*WR*WC MFMAs (2+6 per 48 for the 128x96 tile).  Both operands are
// staged K-minor exactly as they lie in memory ([row][k], 32-float chunks, row stride 40 floats:
// the lane pattern (l&15)*40 + 4*(l>>4) is conflict-free for ds_read_b128), the four k of a lane's
// b128 feed four consecutive MFMA steps (the contraction order inside a 16-chunk is
// 4*(l>>4)+step for both operands).  The dX form reads the weight [c][n] row-major with
// ds_read_b32 (stride = 4 mod 8 floats: conflict-free).  The next chunk's global loads are in
// flight during the MFMAs (register prefetch), LDS is single-buffered: 36 KB per workgroup at
// 128x96, 4 workgroups = 4 waves per SIMD per CU.
//
// Grid: block id -> (row block, column tile) with id % 8 = XCD (MI355X_MICROARCH.md: block b runs
// on XCD b % 8): all column tiles of a row block run on ONE XCD back to back, so the row tile is
// fetched from HBM once and re-read from that XCD's L2.
#include "eda_common.h"
#include "gemm.h"
#include <string.h>
#include <stdio.h>
#include <map>
#include <mutex>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned gemm_hash32(unsigned x) {          // = ln_hash32 (ln.hip)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ float row_sum16(float v) {
  // sum over the 16 lanes of a DPP row (result in every lane of the row)
  int i = __float_as_int(v);
  v += __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR1>(i)); i = __float_as_int(v);
  v += __int_as_float(eda_dpp<EDA_DPP_QUAD_XOR2>(i)); i = __float_as_int(v);
  v += __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(4)>(i)); i = __float_as_int(v);
  v += __int_as_float(eda_dpp<EDA_DPP_ROW_ROR(8)>(i));
  return v;
}

// E_STATS, last workgroup: column sums -> mean / rstd / scale / shift and the running statistics
// (train-mode BatchNorm, pointnet2/pytorch_utils.py:67-120); re-arms the accumulators and the ticket.
__device__ __forceinline__ void bn_finalize(const GemmArgs &a, int tid, int nthreads) {
  if (a.defer_finalize) {          // global-batch statistics: the sums are all-reduced over the ranks first
    if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const long R = a.R;
  for (int c = tid; c < a.N; c += nthreads) {
    const double su = __hip_atomic_load(a.sum + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double sq = __hip_atomic_load(a.sumsq + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.sum + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.sumsq + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double mean = su / (double)R;
    double var = sq / (double)R - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean;
    a.mean_out[c] = meanf;
    a.rstd_out[c] = rstd;
    const float scv = a.gamma[c] * rstd;
    a.scale_out[c] = scv;
    a.shift_out[c] = a.beta[c] - meanf * scv;
    if (a.running_mean) {
      const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
      a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * meanf;
      a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
  }
  if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// VEC: every operand row is 16-byte addressable (K % 4 == 0 for the row operand, aligned strides and
// pointers; the gather's feature rows when c_feat % 4 == 0).  The staging loads are then
// UNCONDITIONAL: rows / columns beyond the matrix are clamped to the last valid one (their results
// are neither stored nor counted) and the contraction tail is clamped too and zeroed when the
// chunk is written to LDS -- no arithmetic and no select touches a loaded value before the MFMAs
// of the current chunk have been issued, so the loads stay in flight underneath them.
// !VEC: element-wise guarded loads (odd K / strides: the 3- and 6-channel inputs).
//
// X_GATHER column order: [dx dy dz 0 | feats 0..C) ], K = 4 + C; the weight (N, 3+C) is read with
// the matching map (k < 3 -> k, k == 3 -> zero, k >= 4 -> k - 1).
// WN = waves along the columns (1: four waves stacked along the rows; 2: a 2 x 2 arrangement, i.e.
// 32*WR-row tiles for the small launches -- plain epilogue only).
// PF = chunks in flight ahead of the one being multiplied (register ring of PF staging sets).
// KC = contraction chunk (32 or 64 floats): a chunk costs two barriers and a staging pass whatever its
// width, which is what bounds the small launches (9 chunks of 32 for K = 288).
// DB = LDS double buffering: chunk c+1 is written to the other half of LDS while chunk c is multiplied,
// ONE barrier per chunk instead of two (PF must be 1: the register set holds chunk c+2 meanwhile).
template <int WR, int WC, int WMODE, int XMODE, bool VEC, int WN = 1, int PF = 1, int KC = G_KC, bool DB = false>
__global__ __launch_bounds__(G_THREADS) void gemm_rows_kernel(const GemmArgs a_in) {
  constexpr int BM = 64 * WR / WN, BN = 16 * WC * WN;
  constexpr int XS = KC + 8;                                     // LDS row stride of [row][k] tiles
  constexpr int KQ = KC / 4;                                     // float4 per tile row
  constexpr int RPP = G_THREADS / KQ;                            // tile rows covered by one pass of the workgroup
  static_assert(BM % 32 == 0, "row tile must be a multiple of 32");
  constexpr int WSN = BN + 4;                                    // W_NN LDS row stride
  constexpr int WS_FLOATS = WMODE == W_NT ? BN * XS : KC * WSN;
  constexpr int XLD = BM / RPP;                                  // float4 per thread per chunk, row operand
  constexpr int WLD = BN * KQ / G_THREADS;                       // float4 per thread per chunk, weight
  static_assert(BN * KQ % G_THREADS == 0 && BM % RPP == 0, "tiles must divide evenly");
  constexpr int TILE_FLOATS = BM * XS + WS_FLOATS;
  static_assert(!DB || PF == 1, "double buffering uses one register set");
  __shared__ __attribute__((aligned(16))) float smem[TILE_FLOATS * (DB ? 2 : 1)];
  __shared__ int is_last;
  float *Xs = smem, *Ws = smem + BM * XS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int xcd = blockIdx.x & 7;
  const long q = blockIdx.x >> 3;
  GemmArgs a = a_in;                      // (uniform: scalar registers; a grouped launch patches the operands.
                                          //  The group table is only ever indexed in the kernel-argument
                                          //  segment: a dynamic index into the local copy would put it in scratch)
  a.ngroups = 0;
  int ct = (int)(q % a_in.col_tiles);
  if (a_in.ngroups > 1) {
    int g = 0;
#pragma unroll
    for (int i = 1; i < G_MAXGROUPS; ++i)
      if (i < a_in.ngroups && ct >= a_in.grp[i].ct0) g = i;
    const GemmGroup *gp = &a_in.grp[g];
    a.x = gp->x; a.ldx = gp->ldx; a.w = gp->w; a.ldw = gp->ldw; a.bias = gp->bias;
    a.y = gp->y; a.ldy = gp->ldy; a.K = gp->K; a.N = gp->N;
    ct -= gp->ct0;
  }
  // persistent over row blocks: this workgroup owns row blocks rb_first, rb_first + rb_step, ...
  // (one pass unless the launcher capped the grid: the statistics epilogues do, so that a column
  // costs one fp64 atomic per WORKGROUP, not per row block)
  const long rb_first = (q / a.col_tiles) * 8 + xcd;
  if (rb_first >= a.row_blocks) return;
  const long rb_step = (long)a.row_slots * 8;
  const int n0 = ct * BN;
  const long R = a.R;
  const int K = a.K, N = a.N;             // X_GATHER: K = 4 + c_feat
  double cs1 = 0.0, cs2 = 0.0;            // E_STATS / E_MASK: column tid's sums over this workgroup's row blocks
  for (long rb = rb_first; rb < a.row_blocks; rb += rb_step) {
  const long row0 = rb * BM;

  // ---- staging maps -------------------------------------------------------------------------
  const int kq4 = 4 * (tid % KQ);         // first k of this thread's float4 inside a chunk
  const int lr = tid / KQ;                // local row (+RPP*i)
  const int rows_here = (int)(R - row0 < BM ? R - row0 : BM);
  int xoff[XLD];                          // element offset of the row from the operand base
  float4 gxyz[XLD];                       // X_GATHER, threads with kq4 == 0: (dx, dy, dz, 0)
  const float *xbase = XMODE == X_GATHER ? a.feats : a.x + row0 * a.ldx;
  if (XMODE == X_GATHER) {
    const long rows_per_scene = (long)a.m * a.ns;
#pragma unroll
    for (int i = 0; i < XLD; ++i) {
      int l = lr + RPP * i;
      if (l >= rows_here) l = rows_here - 1;
      const long row = row0 + l;
      const int b = (int)(row / rows_per_scene);
      const int gc = b * a.m + (int)((row - (long)b * rows_per_scene) / a.ns);
      const int gp = b * a.n_pts + a.idx[row];
      xoff[i] = gp * a.c_feat;
      gxyz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kq4 == 0) {
        const float *p = a.xyz + (long)gp * 3, *c = a.new_xyz + (long)gc * 3;
        gxyz[i].x = (p[0] - c[0]) * a.inv_radius;
        gxyz[i].y = (p[1] - c[1]) * a.inv_radius;
        gxyz[i].z = (p[2] - c[2]) * a.inv_radius;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < XLD; ++i) {
      int l = lr + RPP * i;
      if (l >= rows_here) l = rows_here - 1;
      xoff[i] = l * (int)a.ldx;
    }
  }
  int woff[WLD];
#pragma unroll
  for (int i = 0; i < WLD; ++i) {
    const int s = tid + G_THREADS * i;
    if (WMODE == W_NT) {
      int n = n0 + s / KQ;
      if (n >= N) n = N - 1;
      woff[i] = n * (int)a.ldw;
    } else {
      int nn = n0 + 4 * (s % (BN / 4));
      if (VEC && nn > N - 4) nn = N - 4;
      woff[i] = nn;
    }
  }

  float4 xr[PF][XLD], wr[PF][WLD], bsc[PF], bsh[PF];
  auto fetch = [&](const int set, int kc) {
    const int k = kc + kq4;
    if (VEC) {
      // ---- row operand: unconditional 16-byte loads
      if (XMODE == X_GATHER) {
        int f = k - 4;                                  // feature index of this float4
        if (f > a.c_feat - 4) f = a.c_feat - 4;
        if (f < 0) f = 0;
#pragma unroll
        for (int i = 0; i < XLD; ++i) xr[set][i] = *reinterpret_cast<const float4 *>(xbase + xoff[i] + f);
      } else {
        const int kx = k > K - 4 ? K - 4 : k;
#pragma unroll
        for (int i = 0; i < XLD; ++i) xr[set][i] = *reinterpret_cast<const float4 *>(xbase + xoff[i] + kx);
        if (XMODE == X_BNRELU) {
          bsc[set] = *reinterpret_cast<const float4 *>(a.in_scale + kx);
          bsh[set] = *reinterpret_cast<const float4 *>(a.in_shift + kx);
        }
      }
      // ---- weight
#pragma unroll
      for (int i = 0; i < WLD; ++i) {
        const int s = tid + G_THREADS * i;
        if (WMODE == W_NT) {
          if (XMODE == X_GATHER) {
            // (N, 3 + C) rows are not 16-byte aligned: four element loads, clamped
            const int kmax = K - 2;                     // last source column = 3 + C - 1 = K - 2
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              int sidx = k + u < 4 ? k + u : k + u - 1;
              if (sidx > kmax) sidx = kmax;
              e[u] = a.w[woff[i] + sidx];
            }
            wr[set][i] = make_float4(e[0], e[1], e[2], e[3]);
          } else {
            const int kx = k > K - 4 ? K - 4 : k;
            wr[set][i] = *reinterpret_cast<const float4 *>(a.w + woff[i] + kx);
          }
        } else {
          int c = kc + s / (BN / 4);
          if (c > K - 1) c = K - 1;
          if (a.w_elem) {
            const float *p = a.w + (long)c * a.ldw + woff[i];
            wr[set][i] = make_float4(p[0], p[1], p[2], p[3]);
          } else {
            wr[set][i] = *reinterpret_cast<const float4 *>(a.w + (long)c * a.ldw + woff[i]);
          }
        }
      }
    } else {
      // ---- element-wise path
#pragma unroll
      for (int i = 0; i < XLD; ++i) {
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = k + u;
          e[u] = 0.f;
          if (XMODE == X_GATHER) {
            if (kk >= 4 && kk < K) e[u] = xbase[xoff[i] + kk - 4];
          } else if (kk < K) {
            e[u] = xbase[xoff[i] + kk];
            if (XMODE == X_BNRELU) e[u] = fmaxf(e[u] * a.in_scale[kk] + a.in_shift[kk], 0.f);
          }
        }
        xr[set][i] = make_float4(e[0], e[1], e[2], e[3]);
      }
#pragma unroll
      for (int i = 0; i < WLD; ++i) {
        const int s = tid + G_THREADS * i;
        float e[4];
        if (WMODE == W_NT) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int kk = k + u;
            e[u] = 0.f;
            if (XMODE == X_GATHER) {
              if (kk < K && kk != 3) e[u] = a.w[woff[i] + (kk < 4 ? kk : kk - 1)];
            } else if (kk < K) e[u] = a.w[woff[i] + kk];
          }
        } else {
          const int c = kc + s / (BN / 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            e[u] = 0.f;
            if (c < K && woff[i] + u < N) e[u] = a.w[(long)c * a.ldw + woff[i] + u];
          }
        }
        wr[set][i] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  };
  // write the fetched chunk kc to LDS (VEC: zero the contraction tail, apply the prologue here)
  auto stage = [&](const int set, int kc, const int boff = 0) {
    float *Xs = smem + boff, *Ws = smem + boff + BM * XS;
    const int k = kc + kq4;
    const bool kok = k < K;
#pragma unroll
    for (int i = 0; i < XLD; ++i) {
      float4 v = xr[set][i];
      if (VEC) {
        if (XMODE == X_BNRELU) {
          v.x = fmaxf(v.x * bsc[set].x + bsh[set].x, 0.f); v.y = fmaxf(v.y * bsc[set].y + bsh[set].y, 0.f);
          v.z = fmaxf(v.z * bsc[set].z + bsh[set].z, 0.f); v.w = fmaxf(v.w * bsc[set].w + bsh[set].w, 0.f);
        }
        if (XMODE == X_GATHER && k == 0) v = gxyz[i];
        if (!kok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (XMODE == X_GATHER && k == 0) {
        v = gxyz[i];
      }
      *reinterpret_cast<float4 *>(&Xs[(lr + RPP * i) * XS + kq4]) = v;
    }
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int s = tid + G_THREADS * i;
      float4 v = wr[set][i];
      if (WMODE == W_NT) {
        if (VEC) {
          if (XMODE == X_GATHER && k == 0) v.w = 0.f;          // the padding column between xyz and features
          if (!kok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *reinterpret_cast<float4 *>(&Ws[(s / KQ) * XS + kq4]) = v;
      } else {
        if (VEC && kc + s / (BN / 4) >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(&Ws[(s / (BN / 4)) * WSN + 4 * (s % (BN / 4))]) = v;
      }
    }
  };

  f32x4 acc[WC][WR];
#pragma unroll
  for (int j = 0; j < WC; ++j)
#pragma unroll
    for (int i = 0; i < WR; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float *xb0 = Xs + (wave_m * 16 * WR + (lane & 15)) * XS + 4 * (lane >> 4);
  const float *wb0 = WMODE == W_NT ? Ws + (wave_n * 16 * WC + (lane & 15)) * XS + 4 * (lane >> 4)
                                  : Ws + (4 * (lane >> 4)) * WSN + wave_n * 16 * WC + (lane & 15);

  // E_MASK: the previous layer's pre-activation tile, requested while the LAST chunk is being
  // multiplied (the staging registers are dead by then, so this costs no registers; requested in
  // the epilogue the loads were a fully exposed HBM round trip per row block)
  const bool yvec = (N % 4 == 0) && (a.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15u) == 0);
  const bool zvec = yvec && (a.ldzm % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.zm) & 15u) == 0);
  // plain epilogue: the bias is requested before the main loop (loaded in the epilogue it is an exposed memory round
  // trip at the end of every workgroup -- the small launches are a few microseconds long)
  float bias_r[WC][4];
#pragma unroll
  for (int j = 0; j < WC; ++j) {
    const int col = n0 + 16 * j + 16 * WC * wave_n + 4 * (lane >> 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) bias_r[j][u] = (a.bias && col + u < N) ? a.bias[col + u] : 0.f;
  }
  float4 zt[WC][WR];
  auto load_z = [&]() {
    if (a.dbg & 8) return;
#pragma unroll
    for (int j = 0; j < WC; ++j) {
      const int col = n0 + 16 * j + 16 * WC * wave_n + 4 * (lane >> 4);
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const long row = row0 + 16 * WR * wave_m + 16 * i + (lane & 15);
        zt[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R && col < N) {
          const float *zp = a.zm + row * a.ldzm + col;
          if (zvec) zt[j][i] = *reinterpret_cast<const float4 *>(zp);
          else {
            zt[j][i].x = zp[0];
            if (col + 1 < N) zt[j][i].y = zp[1];
            if (col + 2 < N) zt[j][i].z = zp[2];
            if (col + 3 < N) zt[j][i].w = zp[3];
          }
        }
      }
    }
  };
  auto multiply = [&](const int boff = 0) {
    const float *xb = xb0 + boff, *wb = wb0 + boff;
#pragma unroll
    for (int ks = 0; ks < KC; ks += 16) {
        f32x4 bv[WR];
#pragma unroll
        for (int i = 0; i < WR; ++i) bv[i] = *reinterpret_cast<const f32x4 *>(xb + 16 * i * XS + ks);
#pragma unroll
        for (int j = 0; j < WC; ++j) {
          f32x4 av;
          if (WMODE == W_NT) av = *reinterpret_cast<const f32x4 *>(wb + 16 * j * XS + ks);
          else {
            av[0] = wb[(ks + 0) * WSN + 16 * j]; av[1] = wb[(ks + 1) * WSN + 16 * j];
            av[2] = wb[(ks + 2) * WSN + 16 * j]; av[3] = wb[(ks + 3) * WSN + 16 * j];
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < WR; ++i)
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[i][s], acc[j][i], 0, 0, 0);
        }
      }
  };
  if (DB) {
    // chunk 0 -> buffer 0; then per chunk: write c+1 into the other buffer, request c+2, multiply c, barrier
    fetch(0, 0);
    stage(0, 0, 0);
    if (KC < K) fetch(0, KC);
    __syncthreads();
    int cur = 0;
    for (int kc = 0; kc < K; kc += KC) {
      if (kc + KC < K) {
        stage(0, kc + KC, (cur ^ 1) * TILE_FLOATS);
        if (kc + 2 * KC < K) fetch(0, kc + 2 * KC);
        else if (WN == 1 && a.epi == E_MASK) load_z();
      } else if (WN == 1 && a.epi == E_MASK && K <= KC) load_z();
      multiply(cur * TILE_FLOATS);
      __syncthreads();
      cur ^= 1;
    }
  } else {
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p * KC < K) fetch(p, p * KC);
  for (int kc0 = 0; kc0 < K; kc0 += PF * KC) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int kc = kc0 + p * KC;
      if (kc < K) {
        stage(p, kc);
        __syncthreads();
        if (kc + PF * KC < K) fetch(p, kc + PF * KC);
        else if (WN == 1 && a.epi == E_MASK && kc + KC >= K) load_z();
        multiply();
        __syncthreads();
      }
    }
  }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // lane: row = row0 + 16*WR*wave_m + 16*i + (lane&15), columns n0 + 16*WC*wave_n + 16*j + 4*(lane>>4) + {0..3}
  const int cq = 16 * WC * wave_n + 4 * (lane >> 4);
  // column partials of this wave (E_STATS / E_MASK) are parked in Xs, which is free after the main
  // loop's last barrier: [w*BN + c] first statistic of wave w, [4*BN + w*BN + c] second
  float *red = Xs;
  auto park = [&](int j, int u, float s1v, float s2v) {
    if (a.dbg & 1) return;
    s1v = row_sum16(s1v);
    s2v = row_sum16(s2v);
    if ((lane & 15) == 0) {
      red[wave * BN + 16 * j + cq + u] = s1v;
      red[4 * BN + wave * BN + 16 * j + cq + u] = s2v;
    }
  };
  if (a.epi == E_PLAIN || (WN == 1 && a.epi == E_STATS)) {
    // Dropout of the plain epilogue: same counter-based hash as the LayerNorm / attention kernels
    unsigned dseed = 0, dthresh = 0;
    float dinv = 1.f;
    const bool drop = a.epi == E_PLAIN && a.drop_p > 0.f;
    if (drop) {
      dseed = gemm_hash32((unsigned)(*a.drop_seed) * 0x9E3779B1u + a.drop_salt);
      dthresh = (unsigned)((double)a.drop_p * 4294967296.0);
      dinv = 1.f / (1.f - a.drop_p);
    }
    const bool gated = a.epi == E_PLAIN && a.gate != nullptr;
#pragma unroll
    for (int j = 0; j < WC; ++j) {
      const int col = n0 + 16 * j + cq;
      const float bb[4] = {bias_r[j][0], bias_r[j][1], bias_r[j][2], bias_r[j][3]};
      float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const long row = row0 + 16 * WR * wave_m + 16 * i + (lane & 15);
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          o[u] = acc[j][i][u] + bb[u];
          if (a.relu == 1) o[u] = fmaxf(o[u], 0.f);
          else if (a.relu == 2) o[u] = 0.5f * o[u] * (1.f + erff(o[u] * 0.70710678118654752f));   // GELU (erf form)
        }
        if (row < R && col < N) {
          if (drop) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              o[u] = gemm_hash32(dseed ^ (unsigned)(row * N + col + u)) >= dthresh ? o[u] * dinv : 0.f;
          }
          if (gated) {
            const float *gp = a.gate + row * a.ldgate + col;
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (col + u < N) o[u] = a.gate_mode == 1 ? o[u] + gp[u] : (gp[u] > 0.f ? o[u] * a.gate_scale : 0.f);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) { t1[u] += o[u]; t2[u] += o[u] * o[u]; }
          float *p = a.y + row * a.ldy + col;
          if (yvec) *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
          else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (col + u < N) p[u] = o[u];
          }
        }
      }
      if (WN == 1 && a.epi == E_STATS) {
#pragma unroll
        for (int u = 0; u < 4; ++u) park(j, u, t1[u], t2[u]);
      }
    }
  } else if (WN == 1 && a.epi == E_MASK) {
#pragma unroll
    for (int j = 0; j < WC; ++j) {
      const int col = n0 + 16 * j + cq;
      float sc[4], sh[4], mu[4], rs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = col + u < N;
        sc[u] = ok ? a.m_scale[col + u] : 0.f; sh[u] = ok ? a.m_shift[col + u] : 0.f;
        mu[u] = ok ? a.m_mean[col + u] : 0.f; rs[u] = ok ? a.m_rstd[col + u] : 0.f;
      }
      float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const long row = row0 + 16 * WR * wave_m + 16 * i + (lane & 15);
        if (row < R && col < N) {
          const float zz[4] = {zt[j][i].x, zt[j][i].y, zt[j][i].z, zt[j][i].w};
          float o[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float yv = zz[u] * sc[u] + sh[u];
            o[u] = yv > 0.f ? acc[j][i][u] : 0.f;
            t1[u] += o[u];
            t2[u] += o[u] * (zz[u] - mu[u]) * rs[u];
          }
          float *p = a.y + row * a.ldy + col;
          if (yvec) *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
          else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (col + u < N) p[u] = o[u];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) park(j, u, t1[u], t2[u]);
    }
  } else if (WN == 1 && a.epi == E_SCATTER) {
    // d(features)[b, idx[row], col] += acc: the tile goes through LDS so that one atomic
    // instruction covers 64 CONSECUTIVE channels of ONE row (lane = column): with the
    // accumulator layout (lane = row) the same 33 M atomics of SA2 ran 25x slower.
    const long rows_per_scene = (long)a.m * a.ns;
    const int C = a.c_feat;
    constexpr int SST = 68;                                 // staging row stride (64 columns + 4)
    float *stg = smem + wave * 16 * SST;                    // 16 rows x 64 columns per wave: 17 KB in all
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const long row = row0 + 16 * WR * wave_m + 16 * i + (lane & 15);
      int mybase = -1;                                      // element offset of this lane's row in dfeats
      if (row < R) mybase = ((int)(row / rows_per_scene) * a.n_pts + a.idx[row]) * C;
#pragma unroll
      for (int jc = 0; jc < WC; jc += 4) {                  // 64 columns at a time
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if (jc + jj < WC)
            *reinterpret_cast<float4 *>(&stg[(lane & 15) * SST + 16 * jj + cq]) =
                make_float4(acc[jc + jj][i][0], acc[jc + jj][i][1], acc[jc + jj][i][2], acc[jc + jj][i][3]);
        }
        const int col = n0 + 16 * jc + lane;
        const bool colok = col < C && 16 * jc + lane < BN;
        for (int r = 0; r < 16; ++r) {
          const int base = __shfl(mybase, r);               // lanes 0..15 hold rows 0..15 of this row tile
          if (base >= 0 && colok) atomicAdd(a.dfeats + base + col, stg[r * SST + lane]);
        }
      }
    }
  }
  if (WN == 1 && (a.epi == E_STATS || a.epi == E_MASK)) {
    __syncthreads();
    if (tid < BN) {
      cs1 += ((double)red[tid] + (double)red[BN + tid]) + ((double)red[2 * BN + tid] + (double)red[3 * BN + tid]);
      cs2 += ((double)red[4 * BN + tid] + (double)red[5 * BN + tid]) + ((double)red[6 * BN + tid] + (double)red[7 * BN + tid]);
    }
  }
  if (rb + rb_step < a.row_blocks) __syncthreads();        // the epilogue's LDS scratch vs the next block's staging
  }  // row blocks

  if (WN == 1 && (a.epi == E_STATS || a.epi == E_MASK)) {
    double *d1 = a.epi == E_STATS ? a.sum : a.s1;
    double *d2 = a.epi == E_STATS ? a.sumsq : a.s2;
    if (tid < BN && n0 + tid < N && !(a.dbg & 4)) {
      // returning atomics: the ticket below must not overtake them (see sa_cl.hip bn_stats_kernel)
      const double o1 = atomicAdd(d1 + n0 + tid, cs1);
      const double o2 = atomicAdd(d2 + n0 + tid, cs2);
      asm volatile("" ::"v"(o1), "v"(o2));
    }
    if (a.epi == E_STATS) {
      __syncthreads();
      if (tid == 0)
        is_last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ticket_target - 1;
      __syncthreads();
      if (is_last) bn_finalize(a, tid, G_THREADS);
    }
  }
}

// ---- streaming variants: the many-row layers of the set-abstraction modules ------------------------
// 10^5..10^6 rows against a weight of 64..256 channels a side: a row costs 256-1024 bytes of HBM traffic
// each way and 64-512 MFMA steps, and the (tiny) weight is the only thing worth keeping.  Here every WAVE
// runs its own pipeline over 16-row tiles (persistent, tiles interleaved over all waves of the launch): the
// weight (all of it, or a 64/128-column tile of it) sits in LDS for the lifetime of the workgroup --
// loaded once, not once per row block --, a row tile is requested with coalesced 16-byte loads one tile
// ahead, goes through a wave-private LDS strip into MFMA fragment order (an in-wave LDS write -> read
// needs no barrier: the LDS pipeline is in order per wave), and after the one barrier that publishes the
// weight no wave ever waits for another.  Column statistics stay in registers over all tiles of a wave
// (a few dozen fp32 additions each, then fp64) and cost one fp64 atomic per column per WORKGROUP.
// Measured (B = 8, rocprofv3): 262144 x 128 -> 128 with BatchNorm prologue + statistics 93 us = 92 TFLOP/s
// (the tiled kernel above: 150 us, hipBLASLt's plain GEMM: 102 us); 1048576 x 64 -> 64: 126 us = 4.3 TB/s.
//   KT  contraction length (64 / 128 / 256; X_GATHER: the feature channels, the 3 coordinates ride on one
//       extra MFMA step whose row operand the lanes compute themselves)
//   NT  column tile (64 / 128); block id % col_tiles selects it, the row operand is then read col_tiles
//       times (the repeats come from L2)
template <int KT, int NT, int XMODE, int EPI, int NW, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void gemm_stream_kernel(const GemmArgs a) {
  constexpr int XS = 64 + 8;                 // wave strip: 16 rows x 64 k (longer contractions go through it in 64-wide parts)
  constexpr int WSS = KT + 8;                // weight row stride ([n][k], = 8 mod 16: conflict-free b128 fragments)
  constexpr int KH = KT / 64, NJ = NT / 16;
  constexpr bool mask = EPI == E_MASK, stats = EPI == E_STATS || EPI == E_MASK, gather = XMODE == X_GATHER;
  static_assert(KT % 64 == 0 && NT % 64 == 0 && 2 * NT <= 16 * XS, "shape");
  __shared__ __attribute__((aligned(16))) float smem[NT * WSS + NW * 16 * XS];
  __shared__ __attribute__((aligned(16))) float mtab[mask ? 4 * NT : 4];   // E_MASK: scale | shift | mean | rstd
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  float *Ws = smem, *strip = smem + NT * WSS + wave * 16 * XS;
  const long R = a.R;
  // block id % 8 = XCD: the column tiles of a slot (they stream the same rows) run on ONE XCD, so the repeats
  // of a row tile come from that XCD's L2 (the launcher makes the slot count a multiple of 8)
  const unsigned xq = blockIdx.x >> 3;
  const int ct = (int)(xq % (unsigned)a.col_tiles), n0 = ct * NT;
  const long slot = (long)(xq / (unsigned)a.col_tiles) * 8 + (blockIdx.x & 7), nslots = gridDim.x / (unsigned)a.col_tiles;
  {
    if (mask && tid < NT) {
      mtab[tid] = a.m_scale[n0 + tid]; mtab[NT + tid] = a.m_shift[n0 + tid];
      mtab[2 * NT + tid] = a.m_mean[n0 + tid]; mtab[3 * NT + tid] = a.m_rstd[n0 + tid];
    }
    if (gather) {
      // (N, 3 + C) weight: the feature columns start at element 3 of a row (not 16-byte aligned)
      for (int e = tid; e < NT * KT; e += 64 * NW) {
        const int n = e / KT, k = e % KT;
        Ws[n * WSS + k] = a.w[(long)(n0 + n) * a.ldw + 3 + k];
      }
    } else {
      constexpr int KQ = KT / 4;
      for (int e = tid; e < NT * KQ; e += 64 * NW) {
        const int n = e / KQ, kq = e % KQ;
        *reinterpret_cast<float4 *>(&Ws[n * WSS + 4 * kq]) =
            *reinterpret_cast<const float4 *>(a.w + (long)(n0 + n) * a.ldw + 4 * kq);
      }
    }
  }
  __syncthreads();

  const long ntiles = (R + 15) >> 4;
  const long tstride = nslots * NW;
  long t = slot * NW + wave;
  // staging map of a 64-wide part: lane -> rows (lane>>4) + 4i, k = 4*(lane&15)
  const int sk = 4 * (lane & 15), srow = lane >> 4;
  float4 bsc[KH], bsh[KH];
  if (XMODE == X_BNRELU) {
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      bsc[h] = *reinterpret_cast<const float4 *>(a.in_scale + 64 * h + sk);
      bsh[h] = *reinterpret_cast<const float4 *>(a.in_shift + 64 * h + sk);
    }
  }
  // X_BNBWDPOOL: five column constants per part; per tile (16 rows of ONE pooling group: pool % 16 == 0) the group's
  // arg-max bytes and pooled gradient of the lane's four columns
  constexpr bool bnbwd = XMODE == X_BNBWDPOOL;
  float4 qsc[bnbwd ? KH : 1], qsh[bnbwd ? KH : 1], qka[bnbwd ? KH : 1], qkb[bnbwd ? KH : 1], qkd[bnbwd ? KH : 1];
  if (bnbwd) {
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      const float *c = a.bb_consts + 64 * h + sk;
      qsc[h] = *reinterpret_cast<const float4 *>(c);
      qsh[h] = *reinterpret_cast<const float4 *>(c + KT);
      qka[h] = *reinterpret_cast<const float4 *>(c + 2 * KT);
      qkb[h] = *reinterpret_cast<const float4 *>(c + 3 * KT);
      qkd[h] = *reinterpret_cast<const float4 *>(c + 4 * KT);
    }
  }
  unsigned pam[bnbwd ? KH : 1];
  float4 pdo[bnbwd ? KH : 1];
  int prp = 0;
  // X_GATHER: the neighbour indices of a tile's rows are requested TWO tiles ahead (the feature rows they
  // address one tile ahead), the coordinate operand of the next tile during the current one
  const unsigned rps = (gather || EPI == E_SCATTER) ? (unsigned)a.m * (unsigned)a.ns : 1u;
  int gidx[4], gidx16 = 0;                   // idx of the staging rows / of row r16, tile t + tstride
  float gp_[2] = {0.f, 0.f};                 // raw (point, centre) coordinate g of row r16, tile t (g < 3)
  float avx[gather ? NJ : 1];                // weight fragment of the coordinate step
  auto scene_of = [&](unsigned row) { return row / rps; };
  auto load_idx = [&](long tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long row = tile * 16 + srow + 4 * i;
      if (row > R - 1) row = R - 1;
      gidx[i] = a.idx[row];
    }
    long row = tile * 16 + r16;
    if (row > R - 1) row = R - 1;
    gidx16 = a.idx[row];
  };
  auto load_xyz = [&](long tile) {           // uses gidx16 of that tile
    long row = tile * 16 + r16;
    if (row > R - 1) row = R - 1;
    const unsigned b = scene_of((unsigned)row);
    const unsigned gc = b * (unsigned)a.m + ((unsigned)row - b * rps) / (unsigned)a.ns;
    const long gp = (long)b * a.n_pts + gidx16;
    if (g < 3) { gp_[0] = a.xyz[gp * 3 + g]; gp_[1] = a.new_xyz[(long)gc * 3 + g]; }
  };
  float4 xr[KH][4];
  int goff[4];                               // X_GATHER: element offsets of the staging rows' feature rows
  auto row_offsets = [&](long tile) {        // uses gidx of that tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long row = tile * 16 + srow + 4 * i;
      if (row > R - 1) row = R - 1;
      goff[i] = ((int)scene_of((unsigned)row) * a.n_pts + gidx[i]) * KT;
    }
  };
  auto fetch = [&](const int h, long tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (gather) {
        xr[h][i] = *reinterpret_cast<const float4 *>(a.feats + goff[i] + 64 * h + sk);
      } else {
        long row = tile * 16 + srow + 4 * i;
        if (row > R - 1) row = R - 1;
        xr[h][i] = *reinterpret_cast<const float4 *>(a.x + row * a.ldx + 64 * h + sk);
      }
    }
  };
  auto stage = [&](const int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = xr[h][i];
      if (XMODE == X_BNRELU) {
        v.x = fmaxf(v.x * bsc[h].x + bsh[h].x, 0.f); v.y = fmaxf(v.y * bsc[h].y + bsh[h].y, 0.f);
        v.z = fmaxf(v.z * bsc[h].z + bsh[h].z, 0.f); v.w = fmaxf(v.w * bsc[h].w + bsh[h].w, 0.f);
      }
      if (bnbwd) {
        const unsigned rp = (unsigned)(prp + srow + 4 * i);       // position of this row inside its pooling group
        const unsigned am = pam[h];
        const float dx_ = (am & 0xffu) == rp && v.x * qsc[h].x + qsh[h].x > 0.f ? pdo[h].x : 0.f;
        const float dy_ = ((am >> 8) & 0xffu) == rp && v.y * qsc[h].y + qsh[h].y > 0.f ? pdo[h].y : 0.f;
        const float dz_ = ((am >> 16) & 0xffu) == rp && v.z * qsc[h].z + qsh[h].z > 0.f ? pdo[h].z : 0.f;
        const float dw_ = (am >> 24) == rp && v.w * qsc[h].w + qsh[h].w > 0.f ? pdo[h].w : 0.f;
        v.x = qka[h].x * dx_ + qkb[h].x * v.x + qkd[h].x; v.y = qka[h].y * dy_ + qkb[h].y * v.y + qkd[h].y;
        v.z = qka[h].z * dz_ + qkb[h].z * v.z + qkd[h].z; v.w = qka[h].w * dw_ + qkb[h].w * v.w + qkd[h].w;
      }
      *reinterpret_cast<float4 *>(&strip[(srow + 4 * i) * XS + sk]) = v;
    }
  };
  float t1[stats ? NJ : 1][4], t2[stats ? NJ : 1][4];
  if (stats) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) { t1[j][u] = 0.f; t2[j][u] = 0.f; }
  }
  if (gather) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) avx[j] = g < 3 ? a.w[(long)(n0 + 16 * j + r16) * a.ldw + g] : 0.f;
  }

  if (t < ntiles) {
    if (gather) {
      load_idx(t);
      row_offsets(t);
      load_xyz(t);
      if (t + tstride < ntiles) load_idx(t + tstride);
    }
#pragma unroll
    for (int h = 0; h < KH; ++h) fetch(h, t);
  }
  const float *xf = strip + r16 * XS + 4 * g;
  const float *wf = Ws + r16 * WSS + 4 * g;
  for (; t < ntiles; t += tstride) {
    const long row = t * 16 + r16;
    const bool rok = row < R;
    const bool more = t + tstride < ntiles;
    float4 zt[mask ? NJ : 1];
    if (mask) {
      const long zr = rok ? row : R - 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) zt[j] = *reinterpret_cast<const float4 *>(a.zm + zr * a.ldzm + n0 + 16 * j + 4 * g);
    }
    if (bnbwd) {
      const long r0t = t * 16;
      const long grp = r0t / a.bb_pool;
      prp = (int)(r0t - grp * a.bb_pool);
#pragma unroll
      for (int h = 0; h < KH; ++h) {
        pam[h] = *reinterpret_cast<const unsigned *>(a.bb_argmax + grp * KT + 64 * h + sk);
        pdo[h] = *reinterpret_cast<const float4 *>(a.bb_dout + grp * KT + 64 * h + sk);
      }
    }
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int mybase = -1;                         // E_SCATTER: element offset of row r16's point in dfeats
    if (gather) {
      const float bx = g < 3 ? (gp_[0] - gp_[1]) * a.inv_radius : 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(avx[j], bx, acc[j], 0, 0, 0);
      if (more) { row_offsets(t + tstride); load_xyz(t + tstride); }
    }
    if (EPI == E_SCATTER && rok)
      mybase = ((int)scene_of((unsigned)row) * a.n_pts + a.idx[row]) * a.c_feat;
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      stage(h);
      if (more) fetch(h, t + tstride);
#pragma unroll
      for (int ks = 0; ks < 64; ks += 16) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(xf + ks);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x4 av = *reinterpret_cast<const f32x4 *>(wf + 16 * j * WSS + 64 * h + ks);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc[j], 0, 0, 0);
        }
      }
    }
    if (gather && t + 2 * tstride < ntiles) load_idx(t + 2 * tstride);
    if (EPI == E_SCATTER) {
      // d(features)[point of the row, column] += acc: through the strip, so that one atomic instruction
      // covers 64 consecutive channels of ONE row (lane = column)
#pragma unroll
      for (int jc = 0; jc < NJ; jc += 4) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          *reinterpret_cast<float4 *>(&strip[r16 * XS + 16 * jj + 4 * g]) =
              make_float4(acc[jc + jj][0], acc[jc + jj][1], acc[jc + jj][2], acc[jc + jj][3]);
        for (int r = 0; r < 16; ++r) {
          const int base = __shfl(mybase, r);
          if (base >= 0) atomicAdd(a.dfeats + base + n0 + 16 * jc + lane, strip[r * XS + lane]);
        }
      }
    } else if (rok) {
      float *yp = a.y + row * a.ldy + n0 + 4 * g;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float o[4];
        if (mask) {
          const float zz[4] = {zt[j].x, zt[j].y, zt[j].z, zt[j].w};
          const float4 c0 = *reinterpret_cast<const float4 *>(&mtab[16 * j + 4 * g]);
          const float4 c1 = *reinterpret_cast<const float4 *>(&mtab[NT + 16 * j + 4 * g]);
          const float4 c2 = *reinterpret_cast<const float4 *>(&mtab[2 * NT + 16 * j + 4 * g]);
          const float4 c3 = *reinterpret_cast<const float4 *>(&mtab[3 * NT + 16 * j + 4 * g]);
          const float sc[4] = {c0.x, c0.y, c0.z, c0.w}, sh[4] = {c1.x, c1.y, c1.z, c1.w};
          const float mu[4] = {c2.x, c2.y, c2.z, c2.w}, rs[4] = {c3.x, c3.y, c3.z, c3.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[u] = zz[u] * sc[u] + sh[u] > 0.f ? acc[j][u] : 0.f;
            t1[j][u] += o[u];
            t2[j][u] += o[u] * (zz[u] - mu[u]) * rs[u];
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[u] = acc[j][u];
            if (stats) { t1[j][u] += o[u]; t2[j][u] += o[u] * o[u]; }
          }
        }
        *reinterpret_cast<float4 *>(yp + 16 * j) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  if (!stats) return;
  // column sums of this wave -> its strip, then one fp64 atomic per column per workgroup
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s1v = row_sum16(t1[j][u]), s2v = row_sum16(t2[j][u]);
      if (r16 == 0) { strip[16 * j + 4 * g + u] = s1v; strip[NT + 16 * j + 4 * g + u] = s2v; }
    }
  __syncthreads();
  double *d1 = mask ? a.s1 : a.sum, *d2 = mask ? a.s2 : a.sumsq;
  if (tid < NT) {
    double c1 = 0.0, c2 = 0.0;
    const float *st = smem + NT * WSS;
#pragma unroll
    for (int w = 0; w < NW; ++w) { c1 += (double)st[w * 16 * XS + tid]; c2 += (double)st[w * 16 * XS + NT + tid]; }
    const double o1 = atomicAdd(d1 + n0 + tid, c1);
    const double o2 = atomicAdd(d2 + n0 + tid, c2);
    asm volatile("" ::"v"(o1), "v"(o2));
  }
  if (EPI == E_STATS) {
    __syncthreads();
    if (tid == 0)
      is_last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ticket_target - 1;
    __syncthreads();
    if (is_last) bn_finalize(a, tid, 64 * NW);
  }
}

// ---- bf16 x 3: the streaming kernel on the bf16 matrix pipe at fp32 accuracy ----------------------------------------------
// The SA layers' streaming products are bound by the fp32 matrix pipe, not by memory (profiles/r04_sa_layer_bwd.md).  Both
// operands are K-major as stored (weight rows, activation rows), so every value is split exactly into three bf16 terms
// (v = h + m + l up to 2^-24 |v|: the weight tile once per workgroup, a wave's 16 x 64 strip once per row tile -- it is used
// for all NT/16 column tiles) and a 32-deep contraction step is six v_mfma_f32_16x16x32_bf16 (96 cycles) instead of eight
// v_mfma_f32_16x16x4_f32 (256 cycles), accumulated in fp32, smallest terms first.  Same prologues, epilogues and tile ->
// wave / XCD maps as gemm_stream_kernel (this is that kernel with the staging stores, the weight image and the MFMA core
// replaced); LDS holds three bf16 planes of the weight tile and of every wave's strip (1.5 x the fp32 images: dynamic).
typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float b3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned b3_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(b3_f32x2{a, b}, b3_bf16x2));
}
__device__ __forceinline__ void b3_split(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  h = b3_pk(a, b);
  a -= __uint_as_float(h << 16); b -= __uint_as_float(h & 0xffff0000u);
  m = b3_pk(a, b);
  a -= __uint_as_float(m << 16); b -= __uint_as_float(m & 0xffff0000u);
  l = b3_pk(a, b);
}
// the same split with the two subtractions of a pair as ONE packed fp32 instruction (v_pk_add_f32)
__device__ __forceinline__ void b3_split_pk(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  b3_f32x2 v = {a, b};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b3_bf16x2));
  v = v - b3_f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b3_bf16x2));
  v = v - b3_f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b3_bf16x2));
}
__device__ __forceinline__ f32x4 b3_mma(const uint4 &a, const uint4 &b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b3_bf16x8, a), __builtin_bit_cast(b3_bf16x8, b), c, 0, 0, 0);
}
template <int KT, int NT, int XMODE, int EPI, int NW, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void gemm_stream_b3_kernel(const GemmArgs a) {
  constexpr int XS = 64 + 8;                 // wave strip: 16 rows x 64 k, three bf16 planes (row stride 144 bytes)
  constexpr int WSS = KT + 8;                // weight row stride in bf16 ([n][k] per plane)
  constexpr int KH = KT / 64, NJ = NT / 16;
  constexpr bool mask = EPI == E_MASK, stats = EPI == E_STATS || EPI == E_MASK, gather = XMODE == X_GATHER;
  static_assert(KT % 64 == 0 && NT % 64 == 0, "shape");
  constexpr int WPL = NT * WSS, SPL = 16 * XS;     // bf16 elements of a weight plane / of a strip plane
  static_assert(16 * XS * 4 <= 3 * SPL * 2, "the scatter epilogue's 16 x 72 float image must fit the strip planes");
  // LDS (dynamic, gemm_stream_b3_lds_bytes): weight planes h | m | l, then per wave its strip planes h | m | l; the column
  // partials of the epilogue are parked in the wave's strip (3 * 16 * 72 * 2 = 6912 bytes >= 2 * NT floats)
  extern __shared__ __attribute__((aligned(16))) unsigned short b3_smem[];
  static_assert(2 * NT * 4 <= 3 * SPL * 2, "partials must fit the strip");
  __shared__ __attribute__((aligned(16))) float mtab[mask ? 4 * NT : 4];   // E_MASK: scale | shift | mean | rstd
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  unsigned short *Wp = b3_smem, *sp = b3_smem + 3 * WPL + wave * 3 * SPL;
  float *strip = reinterpret_cast<float *>(sp);      // (epilogue: the planes are dead)
  const long R = a.R;
  // block id % 8 = XCD: the column tiles of a slot (they stream the same rows) run on ONE XCD, so the repeats
  // of a row tile come from that XCD's L2 (the launcher makes the slot count a multiple of 8)
  const unsigned xq = blockIdx.x >> 3;
  const int ct = (int)(xq % (unsigned)a.col_tiles), n0 = ct * NT;
  const long slot = (long)(xq / (unsigned)a.col_tiles) * 8 + (blockIdx.x & 7), nslots = gridDim.x / (unsigned)a.col_tiles;
  {
    if (mask && tid < NT) {
      mtab[tid] = a.m_scale[n0 + tid]; mtab[NT + tid] = a.m_shift[n0 + tid];
      mtab[2 * NT + tid] = a.m_mean[n0 + tid]; mtab[3 * NT + tid] = a.m_rstd[n0 + tid];
    }
    {
      // the weight tile, every value split ONCE into three bf16 terms (w = h + m + l exactly up to 2^-24 relative)
      constexpr int KQ = KT / 4;
      for (int e = tid; e < NT * KQ; e += 64 * NW) {
        const int n = e / KQ, kq = e % KQ;
        float4 wv;
        if (gather) {                        // (N, 3 + C) weight: the feature columns start at element 3 of a row
          const float *wr = a.w + (long)(n0 + n) * a.ldw + 3 + 4 * kq;
          wv = make_float4(wr[0], wr[1], wr[2], wr[3]);
        } else {
          wv = *reinterpret_cast<const float4 *>(a.w + (long)(n0 + n) * a.ldw + 4 * kq);
        }
        unsigned h0, m0, l0, h1, m1, l1;
        b3_split(wv.x, wv.y, h0, m0, l0);
        b3_split(wv.z, wv.w, h1, m1, l1);
        *reinterpret_cast<uint2 *>(&Wp[n * WSS + 4 * kq]) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(&Wp[WPL + n * WSS + 4 * kq]) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(&Wp[2 * WPL + n * WSS + 4 * kq]) = make_uint2(l0, l1);
      }
    }
  }
  __syncthreads();

  const long ntiles = (R + 15) >> 4;
  const long tstride = nslots * NW;
  long t = slot * NW + wave;
  // staging map of a 64-wide part: lane -> rows (lane>>4) + 4i, k = 4*(lane&15)
  const int sk = 4 * (lane & 15), srow = lane >> 4;
  float4 bsc[KH], bsh[KH];
  if (XMODE == X_BNRELU) {
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      bsc[h] = *reinterpret_cast<const float4 *>(a.in_scale + 64 * h + sk);
      bsh[h] = *reinterpret_cast<const float4 *>(a.in_shift + 64 * h + sk);
    }
  }
  // X_BNBWDPOOL: five column constants per part; per tile (16 rows of ONE pooling group: pool % 16 == 0) the group's
  // arg-max bytes and pooled gradient of the lane's four columns
  constexpr bool bnbwd = XMODE == X_BNBWDPOOL;
  // (the five column constants of the pooled BatchNorm backward live in LDS, behind the strips, and are read per part at
  // stage time: as 5 * KH float4 per lane they pushed the prefetch registers of the 256-deep variant to scratch)
  float *qt = reinterpret_cast<float *>(b3_smem + 3 * WPL + NW * 3 * SPL);
  if (bnbwd) {
    for (int e = tid; e < 5 * KT; e += 64 * NW) qt[e] = a.bb_consts[e];
    __syncthreads();
  }
  unsigned pam[bnbwd ? KH : 1];
  float4 pdo[bnbwd ? KH : 1];
  int prp = 0;
  // X_GATHER: the neighbour indices of a tile's rows are requested TWO tiles ahead (the feature rows they
  // address one tile ahead), the coordinate operand of the next tile during the current one
  const unsigned rps = (gather || EPI == E_SCATTER) ? (unsigned)a.m * (unsigned)a.ns : 1u;
  int gidx[4], gidx16 = 0;                   // idx of the staging rows / of row r16, tile t + tstride
  float gp_[2] = {0.f, 0.f};                 // raw (point, centre) coordinate g of row r16, tile t (g < 3)
  float avx[gather ? NJ : 1];                // weight fragment of the coordinate step
  auto scene_of = [&](unsigned row) { return row / rps; };
  auto load_idx = [&](long tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long row = tile * 16 + srow + 4 * i;
      if (row > R - 1) row = R - 1;
      gidx[i] = a.idx[row];
    }
    long row = tile * 16 + r16;
    if (row > R - 1) row = R - 1;
    gidx16 = a.idx[row];
  };
  auto load_xyz = [&](long tile) {           // uses gidx16 of that tile
    long row = tile * 16 + r16;
    if (row > R - 1) row = R - 1;
    const unsigned b = scene_of((unsigned)row);
    const unsigned gc = b * (unsigned)a.m + ((unsigned)row - b * rps) / (unsigned)a.ns;
    const long gp = (long)b * a.n_pts + gidx16;
    if (g < 3) { gp_[0] = a.xyz[gp * 3 + g]; gp_[1] = a.new_xyz[(long)gc * 3 + g]; }
  };
  float4 xr[KH][4];
  int goff[4];                               // X_GATHER: element offsets of the staging rows' feature rows
  auto row_offsets = [&](long tile) {        // uses gidx of that tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long row = tile * 16 + srow + 4 * i;
      if (row > R - 1) row = R - 1;
      goff[i] = ((int)scene_of((unsigned)row) * a.n_pts + gidx[i]) * KT;
    }
  };
  auto fetch = [&](const int h, long tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (gather) {
        xr[h][i] = *reinterpret_cast<const float4 *>(a.feats + goff[i] + 64 * h + sk);
      } else {
        long row = tile * 16 + srow + 4 * i;
        if (row > R - 1) row = R - 1;
        xr[h][i] = *reinterpret_cast<const float4 *>(a.x + row * a.ldx + 64 * h + sk);
      }
    }
  };
  auto stage = [&](const int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = xr[h][i];
      if (XMODE == X_BNRELU) {
        v.x = fmaxf(v.x * bsc[h].x + bsh[h].x, 0.f); v.y = fmaxf(v.y * bsc[h].y + bsh[h].y, 0.f);
        v.z = fmaxf(v.z * bsc[h].z + bsh[h].z, 0.f); v.w = fmaxf(v.w * bsc[h].w + bsh[h].w, 0.f);
      }
      if (bnbwd) {
        const float4 qsc_ = *reinterpret_cast<const float4 *>(qt + 64 * h + sk), qsh_ = *reinterpret_cast<const float4 *>(qt + KT + 64 * h + sk);
        const float4 qka_ = *reinterpret_cast<const float4 *>(qt + 2 * KT + 64 * h + sk), qkb_ = *reinterpret_cast<const float4 *>(qt + 3 * KT + 64 * h + sk);
        const float4 qkd_ = *reinterpret_cast<const float4 *>(qt + 4 * KT + 64 * h + sk);
        const unsigned rp = (unsigned)(prp + srow + 4 * i);       // position of this row inside its pooling group
        const unsigned am = pam[h];
        const float dx_ = (am & 0xffu) == rp && v.x * qsc_.x + qsh_.x > 0.f ? pdo[h].x : 0.f;
        const float dy_ = ((am >> 8) & 0xffu) == rp && v.y * qsc_.y + qsh_.y > 0.f ? pdo[h].y : 0.f;
        const float dz_ = ((am >> 16) & 0xffu) == rp && v.z * qsc_.z + qsh_.z > 0.f ? pdo[h].z : 0.f;
        const float dw_ = (am >> 24) == rp && v.w * qsc_.w + qsh_.w > 0.f ? pdo[h].w : 0.f;
        v.x = qka_.x * dx_ + qkb_.x * v.x + qkd_.x; v.y = qka_.y * dy_ + qkb_.y * v.y + qkd_.y;
        v.z = qka_.z * dz_ + qkb_.z * v.z + qkd_.z; v.w = qka_.w * dw_ + qkb_.w * v.w + qkd_.w;
      }
      unsigned h0, m0, l0, h1, m1, l1;
      b3_split(v.x, v.y, h0, m0, l0);
      b3_split(v.z, v.w, h1, m1, l1);
      *reinterpret_cast<uint2 *>(&sp[(srow + 4 * i) * XS + sk]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(&sp[SPL + (srow + 4 * i) * XS + sk]) = make_uint2(m0, m1);
      *reinterpret_cast<uint2 *>(&sp[2 * SPL + (srow + 4 * i) * XS + sk]) = make_uint2(l0, l1);
    }
  };
  float t1[stats ? NJ : 1][4], t2[stats ? NJ : 1][4];
  if (stats) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) { t1[j][u] = 0.f; t2[j][u] = 0.f; }
  }
  if (gather) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) avx[j] = g < 3 ? a.w[(long)(n0 + 16 * j + r16) * a.ldw + g] : 0.f;
  }

  if (t < ntiles) {
    if (gather) {
      load_idx(t);
      row_offsets(t);
      load_xyz(t);
      if (t + tstride < ntiles) load_idx(t + tstride);
    }
#pragma unroll
    for (int h = 0; h < KH; ++h) fetch(h, t);
  }
  const unsigned short *xf = sp + r16 * XS + 8 * g;        // 16x16x32: lane (g, i) holds k = 8 g .. 8 g + 7 of row / column i
  const unsigned short *wf = Wp + r16 * WSS + 8 * g;
  for (; t < ntiles; t += tstride) {
    const long row = t * 16 + r16;
    const bool rok = row < R;
    const bool more = t + tstride < ntiles;
    float4 zt[mask ? NJ : 1];
    if (mask) {
      const long zr = rok ? row : R - 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) zt[j] = *reinterpret_cast<const float4 *>(a.zm + zr * a.ldzm + n0 + 16 * j + 4 * g);
    }
    if (bnbwd) {
      const long r0t = t * 16;
      const long grp = r0t / a.bb_pool;
      prp = (int)(r0t - grp * a.bb_pool);
#pragma unroll
      for (int h = 0; h < KH; ++h) {
        pam[h] = *reinterpret_cast<const unsigned *>(a.bb_argmax + grp * KT + 64 * h + sk);
        pdo[h] = *reinterpret_cast<const float4 *>(a.bb_dout + grp * KT + 64 * h + sk);
      }
    }
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int mybase = -1;                         // E_SCATTER: element offset of row r16's point in dfeats
    if (gather) {
      const float bx = g < 3 ? (gp_[0] - gp_[1]) * a.inv_radius : 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(avx[j], bx, acc[j], 0, 0, 0);
      if (more) { row_offsets(t + tstride); load_xyz(t + tstride); }
    }
    if (EPI == E_SCATTER && rok)
      mybase = ((int)scene_of((unsigned)row) * a.n_pts + a.idx[row]) * a.c_feat;
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      stage(h);
      if (more) fetch(h, t + tstride);
#pragma unroll
      for (int ks = 0; ks < 64; ks += 32) {
        const uint4 bh = *reinterpret_cast<const uint4 *>(xf + ks);
        const uint4 bm = *reinterpret_cast<const uint4 *>(xf + SPL + ks);
        const uint4 bl = *reinterpret_cast<const uint4 *>(xf + 2 * SPL + ks);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const uint4 ah = *reinterpret_cast<const uint4 *>(wf + 16 * j * WSS + 64 * h + ks);
          const uint4 am = *reinterpret_cast<const uint4 *>(wf + WPL + 16 * j * WSS + 64 * h + ks);
          const uint4 al = *reinterpret_cast<const uint4 *>(wf + 2 * WPL + 16 * j * WSS + 64 * h + ks);
          // six of the nine products: what is dropped (m l, l m, l l) is below 2^-32 of the full product
          acc[j] = b3_mma(al, bh, acc[j]); acc[j] = b3_mma(ah, bl, acc[j]); acc[j] = b3_mma(am, bm, acc[j]);
          acc[j] = b3_mma(am, bh, acc[j]); acc[j] = b3_mma(ah, bm, acc[j]); acc[j] = b3_mma(ah, bh, acc[j]);
        }
      }
    }
    if (gather && t + 2 * tstride < ntiles) load_idx(t + 2 * tstride);
    if (EPI == E_SCATTER) {
      // d(features)[point of the row, column] += acc: through the strip, so that one atomic instruction
      // covers 64 consecutive channels of ONE row (lane = column)
#pragma unroll
      for (int jc = 0; jc < NJ; jc += 4) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          *reinterpret_cast<float4 *>(&strip[r16 * XS + 16 * jj + 4 * g]) =
              make_float4(acc[jc + jj][0], acc[jc + jj][1], acc[jc + jj][2], acc[jc + jj][3]);
        for (int r = 0; r < 16; ++r) {
          const int base = __shfl(mybase, r);
          if (base >= 0) atomicAdd(a.dfeats + base + n0 + 16 * jc + lane, strip[r * XS + lane]);
        }
      }
    } else if (rok) {
      float *yp = a.y + row * a.ldy + n0 + 4 * g;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float o[4];
        if (mask) {
          const float zz[4] = {zt[j].x, zt[j].y, zt[j].z, zt[j].w};
          const float4 c0 = *reinterpret_cast<const float4 *>(&mtab[16 * j + 4 * g]);
          const float4 c1 = *reinterpret_cast<const float4 *>(&mtab[NT + 16 * j + 4 * g]);
          const float4 c2 = *reinterpret_cast<const float4 *>(&mtab[2 * NT + 16 * j + 4 * g]);
          const float4 c3 = *reinterpret_cast<const float4 *>(&mtab[3 * NT + 16 * j + 4 * g]);
          const float sc[4] = {c0.x, c0.y, c0.z, c0.w}, sh[4] = {c1.x, c1.y, c1.z, c1.w};
          const float mu[4] = {c2.x, c2.y, c2.z, c2.w}, rs[4] = {c3.x, c3.y, c3.z, c3.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[u] = zz[u] * sc[u] + sh[u] > 0.f ? acc[j][u] : 0.f;
            t1[j][u] += o[u];
            t2[j][u] += o[u] * (zz[u] - mu[u]) * rs[u];
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[u] = acc[j][u];
            if (stats) { t1[j][u] += o[u]; t2[j][u] += o[u] * o[u]; }
          }
        }
        *reinterpret_cast<float4 *>(yp + 16 * j) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  if (!stats) return;
  // column sums of this wave -> its strip, then one fp64 atomic per column per workgroup
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s1v = row_sum16(t1[j][u]), s2v = row_sum16(t2[j][u]);
      if (r16 == 0) { strip[16 * j + 4 * g + u] = s1v; strip[NT + 16 * j + 4 * g + u] = s2v; }
    }
  __syncthreads();
  double *d1 = mask ? a.s1 : a.sum, *d2 = mask ? a.s2 : a.sumsq;
  if (tid < NT) {
    double c1 = 0.0, c2 = 0.0;
    const float *st = reinterpret_cast<const float *>(b3_smem + 3 * WPL);
#pragma unroll
    for (int w = 0; w < NW; ++w) { c1 += (double)st[w * (3 * SPL / 2) + tid]; c2 += (double)st[w * (3 * SPL / 2) + NT + tid]; }
    const double o1 = atomicAdd(d1 + n0 + tid, c1);
    const double o2 = atomicAdd(d2 + n0 + tid, c2);
    asm volatile("" ::"v"(o1), "v"(o2));
  }
  if (EPI == E_STATS) {
    __syncthreads();
    if (tid == 0)
      is_last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ticket_target - 1;
    __syncthreads();
    if (is_last) bn_finalize(a, tid, 64 * NW);
  }
}

// SA1's first layer: QueryAndGroup of 3 feature channels + the 6 -> NT convolution + BatchNorm statistics.
// A row is [dx dy dz 0 | f0 f1 f2 0]: two MFMA steps whose B operand the lanes compute themselves (lane
// (row, g) supplies coordinate g and feature g), the weight's two fragments per column tile live in
// registers: no LDS, no barrier; the launch is bound by the 256 bytes written per row.
template <int NT>
__global__ __launch_bounds__(256) void gemm_gather3_kernel(const GemmArgs a) {
  constexpr int NJ = NT / 16, NW = 4;
  __shared__ float red[NW][2 * NT];
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const long R = a.R;
  float av0[NJ], av1[NJ];
  float4 bb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float *wp = a.w + (long)(16 * j + r16) * a.ldw;
    av0[j] = g < 3 ? wp[g] : 0.f;
    av1[j] = g < 3 ? wp[3 + g] : 0.f;
    bb[j] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + 16 * j + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float t1[NJ][4], t2[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int u = 0; u < 4; ++u) { t1[j][u] = 0.f; t2[j][u] = 0.f; }
  const unsigned rps = (unsigned)a.m * (unsigned)a.ns;
  const long ntiles = (R + 15) >> 4;
  const long tstride = (long)gridDim.x * NW;
  for (long t = (long)blockIdx.x * NW + wave; t < ntiles; t += tstride) {
    const long row = t * 16 + r16;
    const bool rok = row < R;
    const unsigned rr = (unsigned)(rok ? row : R - 1);
    const unsigned b = rr / rps;
    const unsigned gc = b * (unsigned)a.m + (rr - b * rps) / (unsigned)a.ns;
    const long gp = (long)b * a.n_pts + a.idx[rr];
    float bx = 0.f, bf = 0.f;
    if (g < 3) {
      bx = (a.xyz[gp * 3 + g] - a.new_xyz[(long)gc * 3 + g]) * a.inv_radius;
      bf = a.feats[gp * 3 + g];
    }
    if (rok) {
      float *yp = a.y + row * a.ldy + 4 * g;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[j], bx, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[j], bf, acc, 0, 0, 0);
        const float b4[4] = {bb[j].x, bb[j].y, bb[j].z, bb[j].w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          o[u] = acc[u] + b4[u];
          t1[j][u] += o[u];
          t2[j][u] += o[u] * o[u];
        }
        *reinterpret_cast<float4 *>(yp + 16 * j) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s1v = row_sum16(t1[j][u]), s2v = row_sum16(t2[j][u]);
      if (r16 == 0) { red[wave][16 * j + 4 * g + u] = s1v; red[wave][NT + 16 * j + 4 * g + u] = s2v; }
    }
  __syncthreads();
  if (tid < NT) {
    double c1 = 0.0, c2 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { c1 += (double)red[w][tid]; c2 += (double)red[w][NT + tid]; }
    const double o1 = atomicAdd(a.sum + tid, c1);
    const double o2 = atomicAdd(a.sumsq + tid, c2);
    asm volatile("" ::"v"(o1), "v"(o2));
  }
  __syncthreads();
  if (tid == 0)
    is_last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ticket_target - 1;
  __syncthreads();
  if (is_last) bn_finalize(a, tid, 256);
}

#ifdef EDA_GEMM_PROFILE
// Section timing of gemm_dma_kernel (experiments only; tools/gemm_dma_profile.py): per-wave s_memtime deltas summed over
// all waves of all launches since the last read.
__device__ unsigned long long gemm_prof[64 * 8];
#define GSTAMP(slot)                                                   \
  do {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime();     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(now__));                 \
    prof_acc[slot] += now__ - prof_t;                                  \
    prof_t = now__;                                                    \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)
#else
#define GSTAMP(slot) do { } while (0)
#endif

// ---- plain products with DMA staging -------------------------------------------------------------------
// Y = X W^T (W_NT) / dX = dY W (W_NN) with the plain epilogue, for 32-multiples of the contraction length and 16-byte
// addressable operands: the linear layers of the encoder / decoder / heads and of the frozen text encoder.  What is
// different from gemm_rows_kernel: the chunk tiles go from memory to LDS directly (`global_load_lds`, 16 bytes a lane,
// no staging registers, no ds_write), NST LDS stages in a ring with the DMA NST - 1 chunks ahead and ONE barrier per
// 32-wide chunk; NWM x NWN waves with (BM/NWM) x (BN/NWN) wave tiles.  96-column tiles divide every width of the path
// (288, 576, 864, 768, 2304, 3072).  Same products in the same order as gemm_rows_kernel: results are bitwise equal
// (tools/check_gemm_dma.py).
// LDS layout: a tile row is the 32 floats of the chunk (8 granules of 16 bytes) with granule g of tile row r stored in
// slot g ^ ((r >> 1) & 7): the DMA writes whole 1 KB pieces (8 rows) in lane order, so padding the rows is not an
// option, and with the XOR the MFMA operand reads (16 rows x one granule) touch all 64 banks once.  The W_NN weight
// tile is [k][BN columns] as it lies in memory, read with ds_read_b32.
// Measured (profiles/r03c_gemm_dma.md): a global_load_lds costs the issuing wave ~90 cycles wherever it is issued (all
// at once after the barrier, one between MFMA steps, or from a producer wave -- which then is the bottleneck), so what
// pays is waves per SIMD, i.e. SMALL tiles (32 x 96, 32 KB of LDS, up to 5 workgroups per CU); 64- and 128-row tiles,
// 8- and 16-wave workgroups and 3-4 stages all lost on the shapes of the path.
// LN: the tile spans all N = BN columns of its rows and the epilogue is the residual + Dropout + LayerNorm of the
// post-norm blocks (csrc/ln.hip's forward kernel folded in: the product itself is never written, only z = the
// pre-norm sum, which the LayerNorm backward needs) -- see eda_linear_add_dropout_ln_fwd_f32.
// SK (split contraction): block id = (slice, tile) -- the tile's chunks [slice * sk_cps, (slice + 1) * sk_cps) only; the
// partial accumulators are published as write-through (sc1) 16-byte stores, every wave drains them, one lane takes the
// tile's ticket, and the last workgroup to arrive re-reads ALL slices with sc1 loads and adds them IN SLICE ORDER (so the
// result does not depend on who arrives last: bit-reproducible), then runs the plain epilogue.  No fences
// (cdna_hip_programming.md G16 form R1; MI355X_MICROARCH.md rows publish-large / splitk-seam).
// KCT: the chunk length.  96 (three of the 32-wide chunks per barrier phase; every contraction length of the decoder is a
// multiple: 288, 576, 864) is for the launches with so few tiles that the per-chunk barrier + DMA round trip is what they
// spend their time on.  The MFMA sequence (16-wide k blocks in ascending order) does not depend on KCT: bitwise the same
// results.  A row is KCT / 4 granules; the XOR stays inside a group of 8 granules (32 floats = 32 banks), rows alternate
// between the two halves of the banks when KCT = 96 (96 floats = 1.5 x 64 banks): 16 rows x one granule still hit every
// bank once.
template <int BM, int BN, int NWM, int NWN, int WMODE, int NST, bool LN = false, bool SK = false, int KCT = 32>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_dma_kernel(const GemmArgs a) {
  static_assert(KCT % 32 == 0 && (BM * KCT) % 256 == 0 && (BN * KCT) % 256 == 0, "whole 1 KB pieces");
  static_assert(!SK || KCT == 32, "the slice plan counts 32-wide chunks");
  constexpr int KC = KCT, GR = KC / 4, NW = NWM * NWN;
  constexpr int WR = BM / NWM / 16, WC = BN / NWN / 16;
  static_assert(WR * 16 * NWM == BM && WC * 16 * NWN == BN, "wave tiles must be multiples of 16");
  constexpr int XF = BM * KC, WF = BN * KC, STAGE = XF + WF;      // floats
  constexpr int XP = BM * KC / 256, WP = BN * KC / 256, NP = XP + WP;   // 1 KB pieces of the row / weight tile
  constexpr int NPW = (NP + NW - 1) / NW, REM = NP % NW;          // pieces per wave (waves >= REM: one less if REM)
  __shared__ __attribute__((aligned(1024))) float smem[NST * STAGE];      // (static LDS may exceed 64 KB on gfx950: up to 160 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / NWN, wave_n = wave % NWN;
  const int xcd = blockIdx.x & 7;
  long q = blockIdx.x >> 3;
  int slice = 0;
  if (SK) { slice = (int)(q / a.sk_per); q -= slice * a.sk_per; }      // (all slices of a tile on one XCD)
  // tile -> XCD (block b runs on XCD b % 8, each XCD has its own L2).  row_slots == 0: an XCD owns row blocks (all column
  // tiles of a row block back to back: the row tile is fetched once, every XCD reads the whole weight) -- many rows, small
  // weight.  row_slots == 1: an XCD owns column tiles (its slice of the weight stays in its L2, every XCD reads all rows)
  // -- few rows against a large weight (the text encoder's 640 x 768 -> 2304: 60 -> 23 MB fetched, 37.6 -> 28.1 us)
  int ct;
  long rb;
  if (a.row_slots == 0) {
    ct = (int)(q % a.col_tiles);
    rb = (q / a.col_tiles) * 8 + xcd;
  } else {
    const int ctg = (a.col_tiles + 7) / 8;
    ct = (int)(q % ctg) * 8 + xcd;
    rb = q / ctg;
  }
  if (rb >= a.row_blocks || ct >= a.col_tiles) return;
  const long row0 = rb * BM, R = a.R;
  const int n0 = ct * BN, K = a.K, N = a.N;
#ifdef EDA_GEMM_PROFILE
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prof_t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(prof_t));
#endif

  // ---- DMA maps: piece p of a stage = 8 tile rows (W_NN weight: 64 granules of the [32][BN/4] granule grid)
  const float *src[NPW];
  int dst[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = wave + NW * i;
    dst[i] = p * 256;
    src[i] = a.x;
    if (p < XP) {
      const int gi = 64 * p + lane, r = gi / GR, gs = gi - GR * r, g = (gs & ~7) | ((gs & 7) ^ ((r >> 1) & 7));
      long row = row0 + r;
      if (row > R - 1) row = R - 1;
      src[i] = a.x + row * a.ldx + 4 * g;
    } else if (p < NP) {
      if (WMODE == W_NT) {
        const int gi = 64 * (p - XP) + lane, r = gi / GR, gs = gi - GR * r, g = (gs & ~7) | ((gs & 7) ^ ((r >> 1) & 7));
        int n = n0 + r;
        if (n > N - 1) n = N - 1;
        src[i] = a.w + (long)n * a.ldw + 4 * g;
      } else {
        const int gi = 64 * (p - XP) + lane, k = gi / (BN / 4), sl = gi - (BN / 4) * k;
        int n = n0 + 4 * sl;
        if (n > N - 4) n = N - 4;
        src[i] = a.w + (long)k * a.ldw + n;
      }
    }
  }
  auto issue = [&](int kc, int stage) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = wave + NW * i;
      if (REM != 0 && i == NPW - 1 && p >= NP) break;            // (wave-uniform)
      const float *g = (WMODE == W_NN && p >= XP) ? src[i] + (long)kc * a.ldw : src[i] + kc;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                       (__attribute__((address_space(3))) void *)(smem + stage * STAGE + dst[i]), 16, 0, 0);
    }
  };

  f32x4 acc[WC][WR];
#pragma unroll
  for (int j = 0; j < WC; ++j)
#pragma unroll
    for (int i = 0; i < WR; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int c = lane & 15, qq = lane >> 4;
  // the bias is requested now: loaded in the epilogue it is an exposed memory round trip at the end of every workgroup
  float4 bias_v[WC];
#pragma unroll
  for (int j = 0; j < WC; ++j) {
    const int col = n0 + 16 * WC * wave_n + 16 * j + 4 * qq;
    bias_v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && col < N) bias_v[j] = *reinterpret_cast<const float4 *>(a.bias + col);
  }
  float4 res_v[LN ? WC : 1][LN ? WR : 1], pos_v[LN ? WC : 1][LN ? WR : 1];
  if (LN) {
#pragma unroll
    for (int j = 0; j < WC; ++j) {
      const int col = 16 * WC * wave_n + 16 * j + 4 * qq;
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        long row = row0 + 16 * WR * wave_m + 16 * i + c;
        if (row > R - 1) row = R - 1;
        res_v[j][i] = *reinterpret_cast<const float4 *>(a.ln_resid + row * N + col);
        pos_v[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.ln_pos) pos_v[j][i] = *reinterpret_cast<const float4 *>(a.ln_pos + row * N + col);
      }
    }
  }
  // operand read offsets (floats) inside a stage for the two half-chunks
  int xo[WR][2], wo[WC][2];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int r = wave_m * 16 * WR + 16 * i + c, f = (r >> 1) & 7;
    xo[i][0] = r * KC + 4 * (qq ^ f);
    xo[i][1] = r * KC + 4 * ((4 + qq) ^ f);
  }
#pragma unroll
  for (int j = 0; j < WC; ++j) {
    if (WMODE == W_NT) {
      const int r = wave_n * 16 * WC + 16 * j + c, f = (r >> 1) & 7;
      wo[j][0] = XF + r * KC + 4 * (qq ^ f);
      wo[j][1] = XF + r * KC + 4 * ((4 + qq) ^ f);
    } else {
      wo[j][0] = XF + (4 * qq) * BN + wave_n * 16 * WC + 16 * j + c;
      wo[j][1] = wo[j][0] + 16 * BN;
    }
  }
  auto multiply = [&](const float *st) {
#pragma unroll
    for (int h = 0; h < KC / 16; ++h) {               // k block h = granules 4 h .. 4 h + 3: group h >> 1 of 8 granules, half h & 1
      f32x4 xv[WR], wv[WC];
#pragma unroll
      for (int i = 0; i < WR; ++i) xv[i] = *reinterpret_cast<const f32x4 *>(st + xo[i][h & 1] + 32 * (h >> 1));
#pragma unroll
      for (int j = 0; j < WC; ++j) {
        if (WMODE == W_NT) wv[j] = *reinterpret_cast<const f32x4 *>(st + wo[j][h & 1] + 32 * (h >> 1));
        else {
          const float *wp = st + wo[j][h & 1] + 32 * (h >> 1) * BN;
          wv[j][0] = wp[0]; wv[j][1] = wp[BN]; wv[j][2] = wp[2 * BN]; wv[j][3] = wp[3 * BN];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
          for (int i = 0; i < WR; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j][s], xv[i][s], acc[j][i], 0, 0, 0);
    }
  };

  // The barrier at the end of chunk ci needs chunk ci + 1 in LDS (every wave waits for its own pieces: loads complete
  // in order, so "at most the pieces of the NST - 2 youngest chunks still in flight") and releases the stage of chunk
  // ci to the DMA of chunk ci + NST.  A plain s_barrier: __syncthreads() is also a fence, for which the compiler
  // drains vmcnt to 0 -- i.e. waits for the prefetch it was meant to overlap.
  const int c_lo = SK ? slice * a.sk_cps : 0;
  const int nchunks = SK ? max(0, min(K / KC, c_lo + a.sk_cps) - c_lo) : K / KC;
  const int kbase = c_lo * KC;
  constexpr int AHEAD = NST - 1;
  constexpr int KEEP_HI = (NST - 2) * NPW, KEEP_LO = (NST - 2) * (NPW - 1);
  // lgkmcnt(0) rides with every one of these waits: the barrier behind it hands the stage this chunk was READ from to
  // the DMA of chunk ci + NST - 1 (issued right after it), and nothing orders a landing DMA piece behind a ds_read that is
  // issued but not yet executed (guide: "restage a buffer 1 phase after its last ds_read only when an lgkmcnt before the
  // barrier retired those reads").  The scheduler parks the chunk's last ds_read + MFMAs behind the s_barrier, so without
  // the count that read crossed the barrier in flight: on a CU shared with another queue's LDS-heavy workgroups it came
  // back with bytes of the next-but-one chunk's weight piece (round 5: the text encoder's 48-row products on the second
  // stream of the pipelined step off by 1e-1 in 8-column stripes, one in three replays; never on an otherwise idle GPU,
  // never at the 640-row shapes -- tools/dbg_pipeline_gemm.py is the reproducer).  The empty asm keeps the compiler from
  // moving the reads across.
  auto wait_keep = [&]() {
    asm volatile("" ::: "memory");
    if (REM != 0 && wave >= REM) __builtin_amdgcn_s_waitcnt((KEEP_LO & 15) | (7 << 4) | (0 << 8) | ((KEEP_LO >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt((KEEP_HI & 15) | (7 << 4) | (0 << 8) | ((KEEP_HI >> 4) << 14));
  };
  constexpr int WAIT_ALL = (7 << 4) | (0 << 8);
#pragma unroll
  for (int p = 0; p < AHEAD; ++p)
    if (p < nchunks) issue(kbase + p * KC, p);
  if (NST > 2 && nchunks >= AHEAD) wait_keep(); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
  __builtin_amdgcn_s_barrier();
  GSTAMP(0);
  int st_cur = 0, st_nxt = AHEAD % NST;
  for (int ci = 0; ci < nchunks; ++ci) {
    const bool more = ci + AHEAD < nchunks;
    if (more) issue(kbase + (ci + AHEAD) * KC, st_nxt);
    GSTAMP(1);
    multiply(smem + st_cur * STAGE);
    GSTAMP(2);
    if (more && NST > 2) wait_keep(); else { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(WAIT_ALL); }
    GSTAMP(3);
    __builtin_amdgcn_s_barrier();
    GSTAMP(4);
    st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
    st_nxt = st_nxt + 1 == NST ? 0 : st_nxt + 1;
  }

  if constexpr (SK) {
    // ---- the slices of this tile meet
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const long tile = rb * a.col_tiles + ct;
    constexpr int WAVE_B = WC * WR * 64 * 16;                        // bytes of one wave's partial
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        a.sk_part + tile * a.sk_slices * (long)(BM * BN), 0, a.sk_slices * BM * BN * 4, 0x00020000);
    const int woff = wave * WAVE_B + lane * 16;
#pragma unroll
    for (int j = 0; j < WC; ++j)
#pragma unroll
      for (int i = 0; i < WR; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), rs,
                                               slice * (BM * BN * 4) + woff + (j * WR + i) * 1024, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every wave drains its write-through stores
    __syncthreads();
    unsigned *flag = reinterpret_cast<unsigned *>(smem);              // (the stages are free after the last barrier)
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(a.sk_tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = t == (unsigned)a.sk_slices - 1u;
      if (last) __hip_atomic_store(a.sk_tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // re-armed
      flag[0] = last ? 1u : 0u;
    }
    __syncthreads();
    if (flag[0] == 0u) return;
#pragma unroll
    for (int j = 0; j < WC; ++j)
#pragma unroll
      for (int i = 0; i < WR; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int z = 0; z < a.sk_slices; ++z) {
#pragma unroll
      for (int j = 0; j < WC; ++j)
#pragma unroll
        for (int i = 0; i < WR; ++i)
          acc[j][i] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                           rs, z * (BM * BN * 4) + woff + (j * WR + i) * 1024, 0, 16));
    }
  }

  // ---- epilogue (the plain one of gemm_rows_kernel: bias, ReLU / GELU, Dropout, gate) ------------------
  unsigned dseed = 0, dthresh = 0;
  float dinv = 1.f;
  const bool drop = a.drop_p > 0.f;
  if (drop) {
    dseed = gemm_hash32((unsigned)(*a.drop_seed) * 0x9E3779B1u + a.drop_salt);
    dthresh = (unsigned)((double)a.drop_p * 4294967296.0);
    dinv = 1.f / (1.f - a.drop_p);
  }
  if constexpr (LN) {
    // z = resid + Dropout(acc + bias) (same element hash as ln.hip, so its backward regenerates the mask), then the
    // two-pass LayerNorm over the N columns of a row: lane (c, qq) holds 4 * WC values of row c of each of its WR row
    // tiles; row sums = over qq by cross-lane adds, over the NWN waves of the row through LDS
    float *red = smem;                                       // [NWN][BM] (the stages are free after the last barrier)
    float t[WC][WR][4];
    float rs[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) rs[i] = 0.f;
#pragma unroll
    for (int j = 0; j < WC; ++j) {
      const int col = 16 * WC * wave_n + 16 * j + 4 * qq;
      const float b4[4] = {bias_v[j].x, bias_v[j].y, bias_v[j].z, bias_v[j].w};
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const long row = row0 + 16 * WR * wave_m + 16 * i + c;
        const float r4[4] = {res_v[j][i].x, res_v[j][i].y, res_v[j][i].z, res_v[j][i].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float yy = acc[j][i][u] + b4[u];
          if (drop) yy = gemm_hash32(dseed ^ (unsigned)(row * N + col + u)) >= dthresh ? yy * dinv : 0.f;
          t[j][i][u] = r4[u] + yy;
          rs[i] += t[j][i][u];
        }
        if (row < R && a.ln_z)
          *reinterpret_cast<float4 *>(a.ln_z + row * N + col) = make_float4(t[j][i][0], t[j][i][1], t[j][i][2], t[j][i][3]);
      }
    }
    auto row_total = [&](float (&v)[WR]) {
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        v[i] += __shfl_xor(v[i], 16);
        v[i] += __shfl_xor(v[i], 32);
        if (qq == 0) red[wave_n * BM + 16 * WR * wave_m + 16 * i + c] = v[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWN; ++w) s += red[w * BM + 16 * WR * wave_m + 16 * i + c];
        v[i] = s;
      }
      __syncthreads();
    };
    row_total(rs);
    float mean[WR], rstd[WR], qs[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      mean[i] = rs[i] / (float)N;
      qs[i] = 0.f;
#pragma unroll
      for (int j = 0; j < WC; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float dl = t[j][i][u] - mean[i]; qs[i] += dl * dl; }
    }
    row_total(qs);
#pragma unroll
    for (int i = 0; i < WR; ++i) rstd[i] = 1.f / sqrtf(qs[i] / (float)N + a.ln_eps);
#pragma unroll
    for (int j = 0; j < WC; ++j) {
      const int col = 16 * WC * wave_n + 16 * j + 4 * qq;
      const float4 g4 = *reinterpret_cast<const float4 *>(a.ln_gamma + col), be4 = *reinterpret_cast<const float4 *>(a.ln_beta + col);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {be4.x, be4.y, be4.z, be4.w};
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const long row = row0 + 16 * WR * wave_m + 16 * i + c;
        if (row >= R) continue;
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = (t[j][i][u] - mean[i]) * rstd[i] * g[u] + be[u];
        *reinterpret_cast<float4 *>(a.ln_out + row * N + col) = make_float4(o[0], o[1], o[2], o[3]);
        if (a.ln_out_pos) {
          const float p4[4] = {pos_v[j][i].x, pos_v[j][i].y, pos_v[j][i].z, pos_v[j][i].w};
          *reinterpret_cast<float4 *>(a.ln_out_pos + row * N + col) = make_float4(o[0] + p4[0], o[1] + p4[1], o[2] + p4[2], o[3] + p4[3]);
        }
      }
    }
    if (wave_n == 0 && qq == 0) {
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const long row = row0 + 16 * WR * wave_m + 16 * i + c;
        if (row < R) { a.ln_mean[row] = mean[i]; a.ln_rstd[row] = rstd[i]; }
      }
    }
    return;
  }
  const bool gated = a.gate != nullptr;
#pragma unroll
  for (int j = 0; j < WC; ++j) {
    const int col = n0 + 16 * WC * wave_n + 16 * j + 4 * qq;
    if (col >= N) continue;                                 // (N % 4 == 0: a lane's four columns are in or out together)
    const float b4[4] = {bias_v[j].x, bias_v[j].y, bias_v[j].z, bias_v[j].w};
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const long row = row0 + 16 * WR * wave_m + 16 * i + c;
      if (row >= R) continue;
      float o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o[u] = acc[j][i][u] + b4[u];
        if (a.relu == 1) o[u] = fmaxf(o[u], 0.f);
        else if (a.relu == 2) o[u] = 0.5f * o[u] * (1.f + erff(o[u] * 0.70710678118654752f));
      }
      if (drop) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          o[u] = gemm_hash32(dseed ^ (unsigned)(row * N + col + u)) >= dthresh ? o[u] * dinv : 0.f;
      }
      if (gated) {
        const float4 gv = *reinterpret_cast<const float4 *>(a.gate + row * a.ldgate + col);
        const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = a.gate_mode == 1 ? o[u] + g4[u] : (g4[u] > 0.f ? o[u] * a.gate_scale : 0.f);
      }
      *reinterpret_cast<float4 *>(a.y + row * a.ldy + col) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
#ifdef EDA_GEMM_PROFILE
  GSTAMP(5);
  prof_acc[7] = 1;
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&gemm_prof[((blockIdx.x * NW + wave) & 63) * 8 + i], prof_acc[i]);
#endif
}

bool gemm_vec_ok(const GemmArgs &a, int wmode) {
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (a.xmode == X_GATHER) {
    if (a.c_feat % 4 != 0 || a.c_feat < 4 || !al16(a.feats)) return false;
    return wmode == W_NT;
  }
  if (a.K % 4 != 0 || a.K < 4 || a.ldx % 4 != 0 || !al16(a.x)) return false;
  if (a.xmode == X_BNRELU && (!al16(a.in_scale) || !al16(a.in_shift))) return false;
  if (wmode == W_NT) return a.ldw % 4 == 0 && al16(a.w);
  if (a.N < 4 || a.N % 4 != 0) return false;
  // dX form: the kernel reads the (K, N) weight with 16-byte loads along N -- rows must be 16-byte addressable too
  // (a row-strided view with an odd ld or a misaligned base takes the element-wise kernel instead)
  return a.ldw % 4 == 0 && al16(a.w);
}

template <int WR, int WC, bool VEC, int WN = 1, int PF = 1, int KC = G_KC, bool DB = false>
int launch_cfg(GemmArgs &a, int wmode, hipStream_t stream) {
  constexpr int BM = 64 * WR / WN, BN = 16 * WC * WN;
  a.row_blocks = (a.R + BM - 1) / BM;
  a.col_tiles = (a.N + BN - 1) / BN;
  if (a.ngroups > 1) {
    int ct = 0;
    for (int g = 0; g < a.ngroups; ++g) { a.grp[g].ct0 = ct; ct += (a.grp[g].N + BN - 1) / BN; }
    a.col_tiles = ct;
  }
  long groups = (a.row_blocks + 7) / 8;
  a.dbg = (int)eda_knob(EDA_K_GEMM_DBG);
  if ((a.epi == E_STATS || a.epi == E_MASK) && !(a.dbg & 2)) {
    // ~2048 persistent workgroups (every CU full at 5-6 waves per SIMD, 1.5 rounds)
    const long cap = 2048 / (8 * a.col_tiles) > 1 ? 2048 / (8 * a.col_tiles) : 1;
    if (groups > cap) groups = cap;
  }
  a.row_slots = (int)groups;
  {
    // workgroups that own at least one row block take a ticket
    const long owners = a.row_blocks < groups * 8 ? a.row_blocks : groups * 8;
    a.ticket_target = (unsigned)(owners * a.col_tiles);
  }
  const long blocks = groups * 8 * a.col_tiles;
  if (blocks > 0x7fffffffL) { eda_set_error("gemm: grid too large"); return EDA_ERR_INVALID_ARG; }
  const dim3 grid((unsigned)blocks), block(G_THREADS);
  if (wmode == W_NN)
    hipLaunchKernelGGL((gemm_rows_kernel<WR, WC, W_NN, X_PLAIN, VEC, WN, PF, KC, DB>), grid, block, 0, stream, a);
  else if (a.xmode == X_PLAIN)
    hipLaunchKernelGGL((gemm_rows_kernel<WR, WC, W_NT, X_PLAIN, VEC, WN, PF, KC, DB>), grid, block, 0, stream, a);
  else if (a.xmode == X_BNRELU)
    hipLaunchKernelGGL((gemm_rows_kernel<WR, WC, W_NT, X_BNRELU, VEC, WN, PF, KC, DB>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((gemm_rows_kernel<WR, WC, W_NT, X_GATHER, VEC, WN, PF, KC, DB>), grid, block, 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// ---- DMA-staged plain products ---------------------------------------------------------------------------
// EDA_GEMM_DMA=0 switches the kernel off, =<id> forces configuration <id> of the table below for every eligible
// launch (experiments); EDA_GEMM_DMA_MAP=0/1 forces the tile -> XCD mapping
int g_dma_mode_v = -2;                 // -2: from EDA_GEMM_DMA (eda_gemm_set_dma overrides)
int g_dma_mode() { return g_dma_mode_v == -2 ? (int)eda_knob(EDA_K_GEMM_DMA) : g_dma_mode_v; }

bool dma_eligible(const GemmArgs &a) {
  if (a.epi != E_PLAIN || a.xmode != X_PLAIN || a.ngroups > 1) return false;
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (a.K % 32 != 0 || a.N % 4 != 0 || a.N < 4) return false;
  if (a.ldx % 4 != 0 || a.ldw % 4 != 0 || a.ldy % 4 != 0 || !al16(a.x) || !al16(a.w) || !al16(a.y)) return false;
  if (a.bias && !al16(a.bias)) return false;
  if (a.gate && (a.ldgate % 4 != 0 || !al16(a.gate))) return false;
  return true;
}

bool dma_takes(const GemmArgs &a, int wmode) {
  if (g_dma_mode() == 0) return false;
  if (!dma_eligible(a)) return false;
  if (g_dma_mode() > 0) return true;
  // where it wins inside the step (bench.py's HIP-event table, r03c): >= 400 tiles of 32 x 96 (the 8192-row layers, the
  // text encoder's 2304- / 3072-wide ones) or a long contraction; the 2048- and 640-row launches of 64-192 tiles stay
  // with the 32 x 32 tiles of gemm_rows_kernel (more, smaller workgroups)
  const long tiles = ((a.R + 31) / 32) * ((a.N + 95) / 96);
  return tiles >= 400 || (a.K >= 2048 && tiles >= 128);
}

// Split contraction (gemm_dma_kernel SK): few tiles, long contraction.  Slices so that ~2 workgroups land on every CU,
// each with at least 6 chunks of 32; EDA_GEMM_SPLITK=0 off, =n forces n slices wherever a launch is eligible at all
// (workspace given, <= 1024 tiles, DMA-staged plain product).
struct SkCfg { int bm, bn; };
SkCfg sk_cfg() { return {32, 96}; }     // (64 x 96 and 64 x 192 tiles, 3-stage rings: all slower, profiles/r05_gemm_splitk.txt)
int splitk_slices(long R, int K, int N) {
  const long env = eda_knob(EDA_K_GEMM_SPLITK);
  if (env == 0 || K % 32 != 0) return 1;
  const SkCfg c = sk_cfg();
  const long tiles = ((R + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
  const int nchunks = K / 32;
  if (tiles > 1024 || nchunks < 2) return 1;
  long s = env > 0 ? env : (tiles >= 224 ? 1 : (512 + tiles / 2) / tiles);
  if (env <= 0 && s > nchunks / 6) s = nchunks / 6;
  if (s > nchunks) s = nchunks;
  if (s > 16) s = 16;
  if (s < 2) return 1;
  const int cps = (int)((nchunks + s - 1) / s);
  return (nchunks + cps - 1) / cps;
}
constexpr size_t SK_TICKET_BYTES = 4096;          // 1024 tiles
size_t splitk_bytes(long R, int N, int slices) {
  const SkCfg c = sk_cfg();
  const long tiles = ((R + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
  return SK_TICKET_BYTES + (size_t)tiles * slices * c.bm * c.bn * sizeof(float);
}

// 96-wide chunks (gemm_dma_kernel KCT = 96): which tile for which launch.  0 = none.
int kc96_config(const GemmArgs &a) {
  const long env = eda_knob(EDA_K_GEMM_KC96);
  if (env >= 0) return (int)env;
  // measured (tools/bench_gemm_kc96.py, profiles/r05_gemm_kc96.txt; us per launch inside a replayed graph): the 32 x 32 tile
  // with two 96-wide stages wins wherever the contraction is short and the launch small -- 2048 x 288 -> 288: 9.7 -> 7.9,
  // 640 x 288 -> 288: 6.1 -> 4.8, 2048 x 288 -> 576: 13.1 -> 11.5 -- and loses from 8192 rows on (16.0 -> 18.6: there the
  // 32 x 96 tile's smaller L2 -> LDS traffic counts)
  const long tiles32 = ((a.R + 31) / 32) * ((a.N + 31) / 32);
  if (a.K <= 864 && tiles32 <= 1152) return 1;
  return 0;
}

template <int BM, int BN, int NWM, int NWN, int NST>
int launch_dma1_sk(GemmArgs &a, int wmode, int slices, void *ws, hipStream_t stream) {
  a.row_blocks = (a.R + BM - 1) / BM;
  a.col_tiles = (a.N + BN - 1) / BN;
  {
    const double xb = 4.0 * a.R * a.K, wb = 4.0 * a.N * a.K;
    const int force_map = (int)eda_knob(EDA_K_GEMM_DMA_MAP);
    a.row_slots = force_map >= 0 ? force_map : (xb + wb / 8 < xb / 8 + wb ? 1 : 0);
  }
  const long blocks = a.row_slots == 0 ? (a.row_blocks + 7) / 8 * 8 * a.col_tiles
                                       : (long)((a.col_tiles + 7) / 8) * 8 * a.row_blocks;
  a.sk_tickets = reinterpret_cast<unsigned *>(ws);
  a.sk_part = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(ws) + SK_TICKET_BYTES);
  a.sk_slices = slices;
  a.sk_cps = (a.K / 32 + slices - 1) / slices;
  a.sk_per = blocks / 8;
  const dim3 grid((unsigned)(blocks * slices)), block(64 * NWM * NWN);
  if (wmode == W_NN) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, NWM, NWN, W_NN, NST, false, true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, NWM, NWN, W_NT, NST, false, true>), grid, block, 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

template <int BM, int BN, int NWM, int NWN, int NST, int KCT = 32>
int launch_dma1(GemmArgs &a, int wmode, hipStream_t stream) {
  a.row_blocks = (a.R + BM - 1) / BM;
  a.col_tiles = (a.N + BN - 1) / BN;
  const int force_map = (int)eda_knob(EDA_K_GEMM_DMA_MAP);
  static_assert((size_t)NST * (BM + BN) * KCT * 4 <= 160 * 1024, "stages must fit the LDS");
  {
    // bytes an XCD pulls through its L2 under either mapping
    const double xb = 4.0 * a.R * a.K, wb = 4.0 * a.N * a.K;
    a.row_slots = force_map >= 0 ? force_map : (xb + wb / 8 < xb / 8 + wb ? 1 : 0);
  }
  const long blocks = a.row_slots == 0 ? (a.row_blocks + 7) / 8 * 8 * a.col_tiles
                                       : (long)((a.col_tiles + 7) / 8) * 8 * a.row_blocks;
  if (blocks > 0x7fffffffL) { eda_set_error("gemm: grid too large"); return EDA_ERR_INVALID_ARG; }
  const dim3 grid((unsigned)blocks), block(64 * NWM * NWN);
  if (wmode == W_NN) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, NWM, NWN, W_NN, NST, false, false, KCT>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, NWM, NWN, W_NT, NST, false, false, KCT>), grid, block, 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

int launch_dma(GemmArgs &a, int wmode, hipStream_t stream) {
  switch (g_dma_mode()) {
    case 2: return launch_dma1<64, 96, 2, 2, 2>(a, wmode, stream);
    case 3: return launch_dma1<64, 96, 4, 2, 3>(a, wmode, stream);
    case 4: return launch_dma1<32, 96, 2, 2, 4>(a, wmode, stream);
    case 5: return launch_dma1<32, 96, 2, 6, 2>(a, wmode, stream);
    case 6: return launch_dma1<32, 96, 2, 6, 3>(a, wmode, stream);
    case 7: return launch_dma1<32, 96, 2, 3, 2>(a, wmode, stream);
    case 8: return launch_dma1<16, 96, 1, 6, 2>(a, wmode, stream);
    case 9: return launch_dma1<16, 96, 1, 3, 2>(a, wmode, stream);
    case 10: return launch_dma1<32, 96, 2, 3, 3>(a, wmode, stream);
    default: return launch_dma1<32, 96, 2, 2, 2>(a, wmode, stream);
  }
}

// ---- streaming launches (SA1) ---------------------------------------------------------------------
template <typename KernelT>
int stream_grid(KernelT kern, int threads, long ntiles, int nw) {
  // every CU full once (persistent waves).  The occupancy is cached PER KERNEL (all instantiations share this
  // function's type, so a function-local static would be one value for all of them)
  static std::map<const void *, int> cache;
  static std::mutex mu;
  int per_cu;
  {
    std::lock_guard<std::mutex> lock(mu);
    int &slot = cache[reinterpret_cast<const void *>(kern)];
    if (!slot) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, 0) != hipSuccess || nb < 1) nb = 1;
      slot = nb;
    }
    per_cu = slot;
  }
  long wgs = (long)per_cu * 256;
  if (eda_knob(EDA_K_GEMM_STREAM_GRID) > 0) wgs = eda_knob(EDA_K_GEMM_STREAM_GRID);   // (tests: any grid must work)
  const long need = (ntiles + nw - 1) / nw;
  if (wgs > need) wgs = need;
  return (int)(wgs < 1 ? 1 : wgs);
}

template <int KT, int NT, int XMODE, int EPI, int NW, int MINW>
int launch_stream1(GemmArgs &a, hipStream_t stream) {
  const long ntiles = (a.R + 15) / 16;
  a.col_tiles = a.N / NT;
  int slots = stream_grid(gemm_stream_kernel<KT, NT, XMODE, EPI, NW, MINW>, 64 * NW, ntiles, NW) / a.col_tiles;
  slots = (slots + 7) / 8 * 8;
  const int grid = slots * a.col_tiles;
  a.ticket_target = (unsigned)grid;
  hipLaunchKernelGGL((gemm_stream_kernel<KT, NT, XMODE, EPI, NW, MINW>), dim3(grid), dim3(64 * NW), 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

template <int KT, int NT, int NW, int XMODE>
constexpr size_t gemm_stream_b3_lds_bytes() {
  return 2 * (size_t)(3 * NT * (KT + 8) + NW * 3 * 16 * 72) + (XMODE == X_BNBWDPOOL ? 4 * 5 * (size_t)KT : 0);
}

// bf16 x 3 variant: one workgroup per CU (its LDS image is 140-157 KB)
template <int KT, int NT, int XMODE, int EPI, int NW>
int launch_stream_b3_1(GemmArgs &a, hipStream_t stream) {
  constexpr size_t lds = gemm_stream_b3_lds_bytes<KT, NT, NW, XMODE>();
  // (160 KB per workgroup: the kernel's static LDS -- the mask epilogue's column table, 16 NT bytes -- counts as well)
  if (lds + (EPI == E_MASK ? 16 * (size_t)NT : 16) + 16 > 160 * 1024) return -1;
  const long ntiles = (a.R + 15) / 16;
  a.col_tiles = a.N / NT;
  long wgs = 256;
  if (eda_knob(EDA_K_GEMM_STREAM_GRID) > 0) wgs = eda_knob(EDA_K_GEMM_STREAM_GRID);
  const long need = (ntiles + NW - 1) / NW;
  if (wgs > need) wgs = need;
  int slots = (int)(wgs < 1 ? 1 : wgs) / a.col_tiles;
  slots = (slots + 7) / 8 * 8;
  const int grid = slots * a.col_tiles;
  a.ticket_target = (unsigned)grid;
  auto kern = gemm_stream_b3_kernel<KT, NT, XMODE, EPI, NW, 1>;
  hipError_t e = eda_set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
  if (e != hipSuccess) { eda_set_error("gemm: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
  e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}
// which launches: the MFMA-bound shapes of SA2-4 (128 / 256 -> 128 / 256 forward with the gather or the BatchNorm+ReLU prologue
// and the statistics epilogue; input gradients with the mask epilogue).
// EDA_GEMM_STREAM_B3=0 keeps the fp32-MFMA kernel.  Returns -1 when the launch is not one of them.
template <int KT, int NT, int NW>
int launch_stream_b3(GemmArgs &a, hipStream_t stream) {
  if (eda_knob(EDA_K_GEMM_STREAM_B3) == 0) return -1;    // (tests switch it inside one process: eda_reload_env)
  if (a.epi == E_MASK && a.xmode == X_BNBWDPOOL) return launch_stream_b3_1<KT, NT, X_BNBWDPOOL, E_MASK, NW>(a, stream);
  if (a.epi == E_MASK && a.xmode == X_PLAIN) return launch_stream_b3_1<KT, NT, X_PLAIN, E_MASK, NW>(a, stream);
  if (a.epi == E_SCATTER && a.xmode == X_PLAIN) return launch_stream_b3_1<KT, NT, X_PLAIN, E_SCATTER, NW>(a, stream);
  if (a.epi == E_STATS && a.xmode == X_GATHER) return launch_stream_b3_1<KT, NT, X_GATHER, E_STATS, NW>(a, stream);
  if (a.epi == E_STATS && a.xmode == X_BNRELU) return launch_stream_b3_1<KT, NT, X_BNRELU, E_STATS, NW>(a, stream);
  if (a.epi == E_STATS && a.xmode == X_PLAIN) return launch_stream_b3_1<KT, NT, X_PLAIN, E_STATS, NW>(a, stream);
  return -1;
}

// ---- bf16 x 3 for the many-row PLAIN products (linear layers on 8192 rows: the point features of the cross-modal encoder,
// models/encoder_decoder_layers.py:87-105, 231-245) ----------------------------------------------------------------------
// gemm_dma_kernel runs them at 47-56 % of the fp32 matrix pipe and is bound by what a 32 x 96 tile has to ingest (147 KB for
// 1.8 MFLOP).  Here a workgroup keeps a WHOLE-contraction weight tile (NT columns x KT) in LDS as three bf16 planes, split
// once, and every wave streams 32-row strips of the row operand straight from global memory into the B-operand layout of
// v_mfma_f32_16x16x32_bf16 (lane (g, i) needs k = 8 g .. 8 g + 7 of row i: two 16-byte loads, the four lanes of a row cover
// one 128-byte line per 32-deep step) -- no LDS round trip and no barrier after the weight tile stands; the lane splits its
// own eight values (v = h + m + l) and the step is six MFMAs per 16 x 16 pair (b3_mma, smallest terms first), the two
// 16-row halves of the strip sharing every weight fragment read.  fp32 accuracy (tests/test_gemm_gpu.py: same bounds).
// Block id -> (column tile, slot) as in the streaming kernels: the column tiles of a slot share an XCD's L2.
template <int KT>
constexpr int b3r_wss() { return KT + ((4 - (KT / 2) % 32 + 32) % 32) * 2; }   // row stride (bf16): words = 4 (mod 32), 16-byte rows
template <int KT, int NT>
constexpr size_t b3r_lds_bytes() { return (size_t)3 * NT * b3r_wss<KT>() * 2; }

// ST: 16-row halves per strip (2: every weight fragment read feeds two MFMAs); PF: steps of the strip in flight (a ring)
template <int KT, int NT, int NW, int ST, int PF>
__global__ __launch_bounds__(64 * NW, 1) void gemm_b3_rows_kernel(const GemmArgs a) {
  constexpr int WSS = b3r_wss<KT>(), WPL = NT * WSS, NJ = NT / 16, KS = KT / 32, RS = 16 * ST;
  static_assert(KT % 32 == 0 && NT % 16 == 0 && WSS % 8 == 0 && KS % PF == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned short b3r_smem[];
  unsigned short *Wp = b3r_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const long R = a.R;
  const int N = a.N;
  const unsigned xq = blockIdx.x >> 3;
  const int ct = (int)(xq % (unsigned)a.col_tiles), n0 = ct * NT;
  const long slot = (long)(xq / (unsigned)a.col_tiles) * 8 + (blockIdx.x & 7), nslots = gridDim.x / (unsigned)a.col_tiles;
  const long ntiles = (R + RS - 1) / RS;
  const long tstride = nslots * NW;
  const long t0 = slot * NW + wave;
  float4 xr[PF][ST][2];
  auto row_ptrs = [&](long t, const float *&pa, const float *&pb) {
    const long ra = t * RS + r16, rb = ST == 2 ? ra + 16 : ra;
    pa = a.x + (ra < R ? ra : R - 1) * a.ldx + 8 * g;
    pb = a.x + (rb < R ? rb : R - 1) * a.ldx + 8 * g;
  };
  auto prologue = [&](const float *pa, const float *pb) {
#pragma unroll
    for (int s = 0; s < PF - 1; ++s) {
      xr[s][0][0] = *reinterpret_cast<const float4 *>(pa + 32 * s); xr[s][0][1] = *reinterpret_cast<const float4 *>(pa + 32 * s + 4);
      if (ST == 2) { xr[s][ST - 1][0] = *reinterpret_cast<const float4 *>(pb + 32 * s); xr[s][ST - 1][1] = *reinterpret_cast<const float4 *>(pb + 32 * s + 4); }
    }
  };
  {
    // the weight tile: every load of the thread in flight at once (a rolled loop paid one memory round trip per iteration),
    // the first strip's rows requested before they are consumed
    constexpr int KQ = KT / 4, WIT = (NT * KQ + 64 * NW - 1) / (64 * NW);
    float4 wv[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int e = tid + i * 64 * NW;
      const int n = e / KQ, kq = e % KQ;
      wv[i] = e < NT * KQ ? *reinterpret_cast<const float4 *>(a.w + (long)(n0 + n) * a.ldw + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (t0 < ntiles) {
      const float *pa, *pb;
      row_ptrs(t0, pa, pb);
      prologue(pa, pb);
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int e = tid + i * 64 * NW;
      if (e >= NT * KQ) break;
      const int n = e / KQ, kq = e % KQ;
      unsigned h0, m0, l0, h1, m1, l1;
      b3_split_pk(wv[i].x, wv[i].y, h0, m0, l0);
      b3_split_pk(wv[i].z, wv[i].w, h1, m1, l1);
      *reinterpret_cast<uint2 *>(&Wp[n * WSS + 4 * kq]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2 *>(&Wp[WPL + n * WSS + 4 * kq]) = make_uint2(m0, m1);
      *reinterpret_cast<uint2 *>(&Wp[2 * WPL + n * WSS + 4 * kq]) = make_uint2(l0, l1);
    }
  }
  // epilogue constants (the plain epilogue of gemm_dma_kernel: bias, ReLU / GELU, Dropout, gate / addend)
  unsigned dseed = 0, dthresh = 0;
  float dinv = 1.f;
  const bool drop = a.drop_p > 0.f;
  if (drop) {
    dseed = gemm_hash32((unsigned)(*a.drop_seed) * 0x9E3779B1u + a.drop_salt);
    dthresh = (unsigned)((double)a.drop_p * 4294967296.0);
    dinv = 1.f / (1.f - a.drop_p);
  }
  float4 bias_v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    bias_v[j] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + n0 + 16 * j + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  const unsigned short *wf = Wp + r16 * WSS + 8 * g;          // lane (g, i): k = 8 g .. 8 g + 7 of weight row (= output column) i
  for (long t = t0; t < ntiles; t += tstride) {
    const long ra = t * RS + r16, rb = ra + 16;
    const float *pa, *pb;
    row_ptrs(t, pa, pb);
    f32x4 acc[ST][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[ST - 1][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    if (t != t0) prologue(pa, pb);
    // a ring of PF steps, unrolled by PF inside a rolled loop: the loads of step s + PF - 1 are issued while step s is
    // multiplied (a fully unrolled loop let the scheduler hoist every load of the strip: 256 registers + scratch)
#pragma unroll 1
    for (int s0 = 0; s0 < KS; s0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int s = s0 + u, sn = s + PF - 1;
        if (sn < KS) {
          constexpr int dummy = 0; (void)dummy;
          float4 *d = &xr[(u + PF - 1) % PF][0][0];
          d[0] = *reinterpret_cast<const float4 *>(pa + 32 * sn); d[1] = *reinterpret_cast<const float4 *>(pa + 32 * sn + 4);
          if (ST == 2) { d[2] = *reinterpret_cast<const float4 *>(pb + 32 * sn); d[3] = *reinterpret_cast<const float4 *>(pb + 32 * sn + 4); }
        }
        uint4 bh[ST], bm[ST], bl[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const float4 v0 = xr[u][st][0], v1 = xr[u][st][1];
          b3_split_pk(v0.x, v0.y, bh[st].x, bm[st].x, bl[st].x);
          b3_split_pk(v0.z, v0.w, bh[st].y, bm[st].y, bl[st].y);
          b3_split_pk(v1.x, v1.y, bh[st].z, bm[st].z, bl[st].z);
          b3_split_pk(v1.z, v1.w, bh[st].w, bm[st].w, bl[st].w);
        }
        const unsigned short *ws = wf + 32 * s;
        uint4 ah[NJ], am[NJ], al[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          ah[j] = *reinterpret_cast<const uint4 *>(ws + 16 * j * WSS);
          am[j] = *reinterpret_cast<const uint4 *>(ws + WPL + 16 * j * WSS);
          al[j] = *reinterpret_cast<const uint4 *>(ws + 2 * WPL + 16 * j * WSS);
        }
        // six of the nine products (what is dropped -- m l, l m, l l -- is below 2^-24 of the full product), smallest first;
        // product-major: the 2 NJ accumulators take turns, so that no MFMA waits for the one before it (accumulator-major
        // order left up to five dependent MFMAs back to back, each stalling for the full pipeline latency)
#define B3R_ROUND(A_, B_)                                                                    \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                     \
          acc[0][j] = b3_mma(A_[j], B_[0], acc[0][j]);                                       \
          if (ST == 2) acc[ST - 1][j] = b3_mma(A_[j], B_[ST - 1], acc[ST - 1][j]);                 \
        }
        B3R_ROUND(al, bh) B3R_ROUND(ah, bl) B3R_ROUND(am, bm) B3R_ROUND(am, bh) B3R_ROUND(ah, bm) B3R_ROUND(ah, bh)
#undef B3R_ROUND
      }
    }
    // D = W-fragment (rows = output columns) x row fragment: lane (g, i) holds columns 4 g .. 4 g + 3 of row i
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      const long row = st ? rb : ra;
      if (row >= R) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n0 + 16 * j + 4 * g;
        const float b4[4] = {bias_v[j].x, bias_v[j].y, bias_v[j].z, bias_v[j].w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          o[u] = acc[st][j][u] + b4[u];
          if (a.relu == 1) o[u] = fmaxf(o[u], 0.f);
          else if (a.relu == 2) o[u] = 0.5f * o[u] * (1.f + erff(o[u] * 0.70710678118654752f));
        }
        if (drop) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            o[u] = gemm_hash32(dseed ^ (unsigned)(row * N + col + u)) >= dthresh ? o[u] * dinv : 0.f;
        }
        if (a.gate) {
          const float4 gv = *reinterpret_cast<const float4 *>(a.gate + row * a.ldgate + col);
          const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) o[u] = a.gate_mode == 1 ? o[u] + g4[u] : (g4[u] > 0.f ? o[u] * a.gate_scale : 0.f);
        }
        *reinterpret_cast<float4 *>(a.y + row * a.ldy + col) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

template <int KT, int NT, int NW, int ST, int PF>
int launch_b3_rows1(GemmArgs &a, hipStream_t stream) {
  constexpr size_t lds = b3r_lds_bytes<KT, NT>();
  static_assert(lds + 64 <= 160 * 1024, "weight planes must fit LDS");
  a.col_tiles = a.N / NT;
  int slots = 256 / a.col_tiles / 8 * 8;               // one workgroup per CU, a multiple of 8 slots (slot % 8 = XCD)
  if (slots < 8) slots = 8;
  const long need = ((a.R + 16 * ST - 1) / (16 * ST) + NW - 1) / NW;
  while (slots > 8 && slots - 8 >= need) slots -= 8;
  const int grid = slots * a.col_tiles;
  auto kern = gemm_b3_rows_kernel<KT, NT, NW, ST, PF>;
  hipError_t e = eda_set_max_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
  if (e != hipSuccess) { eda_set_error("gemm: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
  e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}
// which launches: plain rows (W_NT), plain epilogue, >= 4096 rows against a 288- or 576-deep contraction.  Returns -1 otherwise.
int try_b3_rows(GemmArgs &a, int wmode, hipStream_t stream) {
  if (wmode != W_NT || a.xmode != X_PLAIN || a.epi != E_PLAIN || a.ngroups > 1 || a.R < 4096) return -1;
  if (eda_knob(EDA_K_GEMM_B3ROWS) == 0) return -1;
  if (!gemm_vec_ok(a, wmode)) return -1;
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (a.N % 4 != 0 || a.ldy % 4 != 0 || !al16(a.y) || (a.bias && !al16(a.bias))) return -1;
  if (a.gate && (a.ldgate % 4 != 0 || !al16(a.gate))) return -1;
  if ((long)a.R * a.ldx >= 0x7fffffffL * 2) return -1;
  // Measured (tools/bench_gemm_b3rows.py, us per launch in a replayed graph, fp32 MFMA -> bf16 x 3; profiles/r05_gemm_b3rows.txt):
  //   8192 x 288 -> 288  17.2 -> 15.4    8192 x 288 -> 576  32.1 -> 24.1    8192 x 288 -> 864  42.1 -> 46.4 (144 workgroups)
  //   8192 x 576 -> 288  27.5 -> 39.1    4096 x 288 -> 288  11.4 -> 13.4   16384 x 288 -> 288  31.5 -> 24.4
  // (16 waves of 16-row strips: with 8 waves of 32-row strips the split arithmetic of a wave -- 176 VALU instructions per
  // strip and step, redone for each of the N / 48 column tiles -- and its MFMAs ran one after the other: 16.4 / 26.2 / 26.4.)
  // ~11 us of a launch are a serial chain -- launch, weight tile (load, split, barrier), first rows, one strip, store -- so
  // the gain is bounded; 576-deep contractions need 32-column tiles (three MFMAs per split value: slower than fp32).
  // Inside the training step the 33 launches of 8192 x 288 -> 288 showed nothing (17.58 vs 17.60 ms, three alternating runs):
  // by default only where the isolated gain is >= 20 %; EDA_GEMM_B3ROWS=2 takes every eligible shape.
  const bool all = eda_knob(EDA_K_GEMM_B3ROWS) == 2;
  if (a.K == 288 && a.N % 48 == 0 && (all || (a.N == 576 && a.R >= 8192) || (a.N <= 576 && a.R >= 16384)))
    return launch_b3_rows1<288, 48, 16, 1, 3>(a, stream);
  if (a.K == 576 && a.N % 32 == 0 && all) return launch_b3_rows1<576, 32, 8, 2, 3>(a, stream);
  return -1;
}

// NW waves per workgroup, MINW waves per SIMD the register allocation aims at (8 x 4: 128 registers, two
// workgroups per CU; 8 x 2: one workgroup per CU with up to 256 registers.  Measured on SA1, B = 8: the
// latter beats two 6-wave workgroups at <= 168 registers: 64 -> 128: 230 vs 269 us, 128 -> 64 with the
// mask epilogue: 226 vs 254 us)
template <int KT, int NT, int NW, int MINW>
int launch_stream(GemmArgs &a, hipStream_t stream) {
  if (a.epi == E_MASK && a.xmode == X_BNBWDPOOL) return launch_stream1<KT, NT, X_BNBWDPOOL, E_MASK, NW, MINW>(a, stream);
  if (a.epi == E_MASK) return launch_stream1<KT, NT, X_PLAIN, E_MASK, NW, MINW>(a, stream);     // dX of a layer: plain rows
  if (a.epi == E_SCATTER) return launch_stream1<KT, NT, X_PLAIN, E_SCATTER, NW, MINW>(a, stream);
  if (a.xmode == X_GATHER) return launch_stream1<KT, NT, X_GATHER, E_STATS, NW, MINW>(a, stream);
  if (a.xmode == X_BNRELU) return launch_stream1<KT, NT, X_BNRELU, E_STATS, NW, MINW>(a, stream);
  return launch_stream1<KT, NT, X_PLAIN, E_STATS, NW, MINW>(a, stream);
}

// the streaming kernels take: forward layers with BatchNorm statistics, dX with the mask epilogue and the
// scatter of the gather's backward; many rows, 64/128/256 channels on both sides, 16-byte addressable
// operands.  EDA_GEMM_STREAM=0 switches them off, EDA_GEMM_STREAM_MINR=<rows> moves the threshold (tests).
// Returns -1 if the launch is not theirs.
int try_stream(GemmArgs &a, int wmode, hipStream_t stream, bool dry = false) {
  if (wmode != W_NT || a.ngroups > 1 || a.epi == E_PLAIN) return -1;
  if (eda_knob(EDA_K_GEMM_STREAM) == 0) return -1;
  const long minr = eda_knob(EDA_K_GEMM_STREAM_MINR);
  if (a.R < minr || a.R >= 0x7fffffffL) return -1;
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (a.bias || a.relu) return -1;
  if (a.epi != E_SCATTER && (a.ldy % 4 != 0 || !al16(a.y))) return -1;
  if (a.xmode == X_GATHER && a.c_feat == 3) {
    if (a.epi != E_STATS || a.N != 64) return -1;
    if (dry) return 0;
    const long ntiles = (a.R + 15) / 16;
    const int grid = stream_grid(gemm_gather3_kernel<64>, 256, ntiles, 4);
    a.ticket_target = (unsigned)grid;
    hipLaunchKernelGGL((gemm_gather3_kernel<64>), dim3(grid), dim3(256), 0, stream, a);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { eda_set_error("gemm: launch failed: %s", hipGetErrorString(err)); return (int)err; }
    return 0;
  }
  int K = a.K;
  if (a.xmode == X_GATHER) {
    if (a.epi != E_STATS || !al16(a.feats)) return -1;
    K = a.c_feat;
  } else {
    if (!gemm_vec_ok(a, wmode)) return -1;
    if (a.epi == E_MASK && ((a.xmode != X_PLAIN && a.xmode != X_BNBWDPOOL) || a.ldzm % 4 != 0 || !al16(a.zm))) return -1;
    if (a.xmode == X_BNBWDPOOL && (a.epi != E_MASK || a.ldx != a.K || a.bb_pool <= 0 || a.bb_pool % 16 != 0 || !a.bb_argmax ||
                                   !al16(a.bb_dout) || !al16(a.bb_consts) || (reinterpret_cast<uintptr_t>(a.bb_argmax) & 3u)))
      return -1;
    if (a.epi == E_SCATTER && (a.xmode != X_PLAIN || a.N != a.c_feat)) return -1;
  }
  const int N = a.N;
  if (N % 64 != 0) return -1;
  if (eda_env().skip_on) {       // EDA_GEMM_STREAM_SKIP, debugging aid: "K,N,epi" leaves one shape to the tiled kernel
    const EdaEnv &ev = eda_env();
    if ((ev.skip_k == 0 || ev.skip_k == K) && (ev.skip_n == 0 || ev.skip_n == N) && (ev.skip_e < 0 || ev.skip_e == a.epi)) return -1;
  }
  if (dry) return ((K == 64 || K == 128) && (N == 64 || N % 128 == 0)) || K == 256 ? 0 : -1;
  if (K == 64 && N == 64) return launch_stream<64, 64, 8, 4>(a, stream);
  if (K == 64 && N % 128 == 0) return launch_stream<64, 128, 8, 2>(a, stream);
  if (K == 128 && N == 64) return launch_stream<128, 64, 8, 2>(a, stream);
  if (K == 128 && N % 128 == 0) {
    const int rc = launch_stream_b3<128, 128, 8>(a, stream);
    return rc >= 0 ? rc : launch_stream<128, 128, 8, 2>(a, stream);
  }
  if (K == 256) {
    const int rc = launch_stream_b3<256, 64, 8>(a, stream);
    return rc >= 0 ? rc : launch_stream<256, 64, 8, 2>(a, stream);
  }
  return -1;
}

bool stream_takes(const GemmArgs &a, int wmode) {
  GemmArgs t = a;
  return try_stream(t, wmode, nullptr, true) == 0;
}

int g_force_cfg() { return (int)eda_knob(EDA_K_GEMM_CFG); }

}  // namespace

// Tile selection: the column tile that pads least (96 for the 288-multiples, 128 / 64 for the
// powers of two), 128-row blocks when that still gives >= 2 workgroups per CU, else 64-row blocks.
// Operands that are not 16-byte addressable take the element-wise 64x64 kernel.
// EDA_GEMM_CFG=<WR><WC> forces a tile (e.g. 26).
bool eda_gemm_stream_takes(const GemmArgs &a, int wmode) { return stream_takes(a, wmode); }

int eda_gemm_launch(GemmArgs &a, int wmode, hipStream_t stream) {
  if (a.R <= 0 || a.N <= 0) return 0;
  if (a.xmode == X_BNBWDPOOL && !stream_takes(a, wmode)) {
    eda_set_error("gemm: the pooled BatchNorm-backward prologue exists in the streaming kernels only");
    return EDA_ERR_INVALID_ARG;
  }
  if (wmode == W_NN && a.xmode != X_PLAIN) { eda_set_error("gemm: dX form takes plain rows"); return EDA_ERR_INVALID_ARG; }
  // 32-bit element offsets: per 64/128-row tile for plain rows, from the tensor base for the gather
  if ((long)(64 * 2) * a.ldx >= 0x7fffffffL || (long)(wmode == W_NT ? a.N : a.K) * a.ldw >= 0x7fffffffL ||
      (a.xmode == X_GATHER && (long)a.n_pts * (a.c_feat > 3 ? a.c_feat : 3) *
                                  ((a.R + (long)a.m * a.ns - 1) / ((long)a.m * a.ns)) >= 0x7fffffffL)) {
    eda_set_error("gemm: operand too large for 32-bit element offsets");
    return EDA_ERR_INVALID_ARG;
  }
  {
    const int rc = try_stream(a, wmode, stream);
    if (rc >= 0) return rc;
  }
  {
    const int rc = try_b3_rows(a, wmode, stream);          // >= 4096 plain rows x 288 / 576: the bf16 pipe at fp32 accuracy
    if (rc >= 0) return rc;
  }
  if (!gemm_vec_ok(a, wmode)) return launch_cfg<1, 4, false>(a, wmode, stream);
  // (a contraction of <= 864 with few tiles: the 96-wide chunks unsplit beat the split 32-wide ones -- 640 x 768 -> 768 14.2 ->
  //  11.3 us, 2048 x 576 -> 288 14.6 -> 12.1, 640 x 576 -> 288 9.4 -> 6.5; from 3072 on the two are level and the split stays)
  const bool kc96_first = g_dma_mode() != 0 && a.K % 96 == 0 && a.K <= 864 && dma_eligible(a) && kc96_config(a) == 1 &&
                          eda_knob(EDA_K_GEMM_SPLITK) <= 0;        // (a forced slice count keeps the split: tests)
  if (!kc96_first && a.sk_ws && g_dma_mode() != 0 && dma_eligible(a)) {
    const int slices = splitk_slices(a.R, a.K, a.N);
    if (slices > 1 && a.sk_ws_bytes >= splitk_bytes(a.R, a.N, slices) && (reinterpret_cast<uintptr_t>(a.sk_ws) & 15u) == 0)
      return launch_dma1_sk<32, 96, 2, 2, 2>(a, wmode, slices, a.sk_ws, stream);
  }
  if (g_dma_mode() != 0 && a.K % 96 == 0 && dma_eligible(a)) {
    const int cfg = kc96_config(a);
    switch (cfg) {
      case 1: return launch_dma1<32, 32, 2, 2, 2, 96>(a, wmode, stream);
      case 2: return launch_dma1<32, 32, 2, 2, 3, 96>(a, wmode, stream);
      case 3: return launch_dma1<32, 96, 2, 2, 2, 96>(a, wmode, stream);
      case 4: return launch_dma1<32, 48, 2, 3, 2, 96>(a, wmode, stream);
      case 5: return launch_dma1<32, 48, 2, 3, 3, 96>(a, wmode, stream);
      case 6: return launch_dma1<64, 96, 2, 2, 2, 96>(a, wmode, stream);
      case 7: return launch_dma1<16, 96, 1, 2, 3, 96>(a, wmode, stream);
      case 8: return launch_dma1<32, 96, 2, 6, 2, 96>(a, wmode, stream);
      default: break;
    }
  }
  if (dma_takes(a, wmode)) return launch_dma(a, wmode, stream);
  // measured on MI355X (tools/bench_gemm.py, profiles/r02a_gemm_tiles.txt): the 64x64 tile at 5-6
  // waves per SIMD beats the 128-row tiles at 2-4 on every shape of the path; 64x128 is a few
  // per cent ahead for the 128-multiples with many rows
  const int N = a.N;
  int wr = 1, wc = 4;
  if (N % 128 == 0 && a.R >= 32768) wc = 8;
  if (a.epi == E_PLAIN && a.xmode == X_PLAIN) {
    // small launches: 32-row tiles (2 x 2 waves) so that every CU holds several workgroups
    const int f2 = g_force_cfg();
    const long wgs64 = ((a.R + 63) / 64) * ((N + 63) / 64);
    if (f2 == 112 || (f2 <= 0 && wgs64 <= 384)) return launch_cfg<1, 1, true, 2, 2>(a, wmode, stream);  // 32 x 32
    if (f2 == 122) return launch_cfg<1, 2, true, 2, 2>(a, wmode, stream); // 32 x 64
    // everything larger: 32 x 64 tiles, LDS double-buffered (profiles/r02b_gemm_cold.txt: best or within
    // noise of the best on every shape when the operands come from HBM)
    if (f2 <= 0) return launch_cfg<1, 2, true, 2, 1, 32, true>(a, wmode, stream);
    if (f2 == 132) return launch_cfg<1, 3, true, 2, 2>(a, wmode, stream);                                // 32 x 96
    if (f2 == 142) return launch_cfg<1, 4, true, 1, 2>(a, wmode, stream);                                // 64 x 64, two chunks ahead
    if (f2 == 514) return launch_cfg<1, 4, true, 1, 1, 32, true>(a, wmode, stream);                      // 64 x 64, LDS double buffer
    if (f2 == 518) return launch_cfg<1, 8, true, 1, 1, 32, true>(a, wmode, stream);                      // 64 x 128
    if (f2 == 511) return launch_cfg<1, 1, true, 2, 1, 32, true>(a, wmode, stream);                      // 32 x 32
    if (f2 == 512) return launch_cfg<1, 2, true, 2, 1, 32, true>(a, wmode, stream);                      // 32 x 64
    if (f2 == 524) return launch_cfg<2, 4, true, 1, 1, 32, true>(a, wmode, stream);                      // 128 x 64
    if (f2 == 182) return launch_cfg<1, 8, true, 1, 2>(a, wmode, stream);                                // 64 x 128, two chunks ahead
    if (f2 == 282) return launch_cfg<1, 8, true, 1, 1, 64>(a, wmode, stream);                            // 64 x 128, 64-wide chunks
    if (f2 == 184) return launch_cfg<2, 4, true, 1, 2>(a, wmode, stream);                                // 128 x 64, two chunks ahead
    if (f2 == 212) return launch_cfg<1, 1, true, 2, 1, 64>(a, wmode, stream);                            // 32 x 32, 64-wide chunks
    if (f2 == 222) return launch_cfg<1, 2, true, 2, 1, 64>(a, wmode, stream);                            // 32 x 64, 64-wide chunks
    if (f2 == 232) return launch_cfg<1, 3, true, 2, 1, 64>(a, wmode, stream);                            // 32 x 96
    if (f2 == 242) return launch_cfg<1, 4, true, 1, 1, 64>(a, wmode, stream);                            // 64 x 64
  }
  const int f = g_force_cfg();
  if (f > 0 && f < 100) { wr = f / 10; wc = f % 10; }
  if (wr == 2) {
    switch (wc) {
      case 4: return launch_cfg<2, 4, true>(a, wmode, stream);
      case 6: return launch_cfg<2, 6, true>(a, wmode, stream);
      default: return launch_cfg<2, 8, true>(a, wmode, stream);
    }
  }
  switch (wc) {
    case 4: return launch_cfg<1, 4, true>(a, wmode, stream);
    case 6: return launch_cfg<1, 6, true>(a, wmode, stream);
    default: return launch_cfg<1, 8, true>(a, wmode, stream);
  }
}

static void gemm_defaults(GemmArgs &a) {
  memset(&a, 0, sizeof(a));
}

// ---- C ABI: linear layer + residual + Dropout + LayerNorm in one launch ----------------------------------
extern "C" int eda_linear_add_dropout_ln_supported(int K, int N) { return N == 288 && K >= 32 && K % 32 == 0; }

// Up to 2048 rows the 16-row blocks are at most 128 workgroups of a kernel that streams the WHOLE weight per block (2048 rows:
// 128 workgroups on 256 CUs): with scratch, two workgroups share a block's contraction (split as in gemm_dma_kernel SK: ordered,
// reproducible) and the last one to arrive runs the LayerNorm epilogue.
static int ln_splitk_slices(long R, int K) {
  const long env = eda_knob(EDA_K_GEMM_SPLITK);
  if (env == 0 || K < 256 || eda_knob(EDA_K_GEMM_LN_VAR) == 1) return 1;      // (EDA_GEMM_LN_VAR=1: this split alone off)
  // measured (tools/time_linear_ln.py, us per launch in a replayed graph, K = 288): 640 rows 14.6 -> 13.0, 1024 14.7 -> 13.3,
  // 2048 15.2 -> 14.0; 3072 rows (384 workgroups of 6 waves, 78 KB each) 15.7 -> 20.7: up to 128 row blocks only
  if ((R + 15) / 16 > 128) return 1;
  return 2;
}
extern "C" size_t eda_linear_add_dropout_ln_workspace_bytes(long R, int K, int N) {
  if (R <= 0 || !eda_linear_add_dropout_ln_supported(K, N)) return 0;
  const int s = ln_splitk_slices(R, K);
  return s > 1 ? SK_TICKET_BYTES + (size_t)((R + 15) / 16) * s * 16 * 288 * sizeof(float) : 0;
}

extern "C" int eda_linear_add_dropout_ln_fwd_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                                 const float *bias, const float *resid, const float *gamma,
                                                 const float *beta, float eps, float drop_p,
                                                 const unsigned long long *drop_seed, unsigned drop_salt, float *z,
                                                 float *out, float *mean, float *rstd, const float *pos, float *out_pos,
                                                 void *stream_) {
  return eda_linear_add_dropout_ln_fwd_ws_f32(x, ldx, R, K, w, ldw, N, bias, resid, gamma, beta, eps, drop_p, drop_seed, drop_salt, z,
                                              out, mean, rstd, pos, out_pos, nullptr, 0, stream_);
}

extern "C" int eda_linear_add_dropout_ln_fwd_ws_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                                    const float *bias, const float *resid, const float *gamma,
                                                    const float *beta, float eps, float drop_p,
                                                    const unsigned long long *drop_seed, unsigned drop_salt, float *z,
                                                    float *out, float *mean, float *rstd, const float *pos, float *out_pos,
                                                    void *ws, size_t ws_bytes, void *stream_) {
  EDA_CHECK_ARG(R >= 0 && ldx >= K && ldw >= K, "bad dimension");
  EDA_CHECK_ARG(eda_linear_add_dropout_ln_supported(K, N), "the fused kernel exists for N = 288 and K a multiple of 32");
  EDA_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || drop_seed), "dropout needs 0 <= p < 1 and a seed");
  if (R == 0) return 0;
  EDA_CHECK_ARG(x && w && resid && gamma && beta && out && mean && rstd, "null pointer");
  EDA_CHECK_ARG((pos == nullptr) == (out_pos == nullptr), "pos and out_pos come together");
  EDA_CHECK_ARG((long long)R * N < 0x100000000LL || drop_p == 0.f, "dropout: more than 2^32 elements");
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  EDA_CHECK_ARG(ldx % 4 == 0 && ldw % 4 == 0 && al16(x) && al16(w) && al16(resid) && al16(gamma) && al16(beta) && al16(out) &&
                (!bias || al16(bias)) && (!z || al16(z)) && (!pos || (al16(pos) && al16(out_pos))), "operands must be 16-byte aligned");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN;
  a.x = x; a.ldx = ldx; a.R = R; a.K = K;
  a.w = w; a.ldw = ldw; a.N = N; a.bias = bias;
  a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_salt = drop_salt;
  a.ln_resid = resid; a.ln_gamma = gamma; a.ln_beta = beta; a.ln_eps = eps; a.ln_pos = pos;
  a.ln_z = z; a.ln_out = out; a.ln_out_pos = out_pos; a.ln_mean = mean; a.ln_rstd = rstd;
  a.col_tiles = 1; a.row_slots = 0;
  // 16-row blocks (6 waves) below 4096 rows, 32-row blocks (12 waves) from there (EDA_GEMM_LN_BM forces one)
  const int force_bm = (int)eda_knob(EDA_K_GEMM_LN_BM);
  const int bm = force_bm == 16 || force_bm == 32 ? force_bm : (R >= 4096 ? 32 : 16);
  hipStream_t stream = (hipStream_t)stream_;
  if (bm == 32) {
    a.row_blocks = (R + 31) / 32;
    const long blocks = (a.row_blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_dma_kernel<32, 288, 2, 6, W_NT, 2, true>), dim3((unsigned)blocks), dim3(768), 0, stream, a);
  } else {
    a.row_blocks = (R + 15) / 16;
    const long blocks = (a.row_blocks + 7) / 8 * 8;
    const int slices = ln_splitk_slices(R, K);
    if (slices > 1 && ws && ws_bytes >= eda_linear_add_dropout_ln_workspace_bytes(R, K, N) && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0) {
      a.sk_tickets = reinterpret_cast<unsigned *>(ws);
      a.sk_part = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(ws) + SK_TICKET_BYTES);
      a.sk_slices = slices;
      a.sk_cps = (K / 32 + slices - 1) / slices;
      a.sk_per = blocks / 8;
      hipLaunchKernelGGL((gemm_dma_kernel<16, 288, 1, 6, W_NT, 2, true, true>), dim3((unsigned)(blocks * slices)), dim3(384), 0, stream, a);
      hipError_t e2 = hipGetLastError();
      if (e2 != hipSuccess) { eda_set_error("linear_add_dropout_ln: launch failed: %s", hipGetErrorString(e2)); return (int)e2; }
      return 0;
    }
    // experiment switch (tools/bench_linear_ln.py): ring depth x waves of the 16-row variant
    const int var = (int)eda_knob(EDA_K_GEMM_LN_VAR);
    if (var == 63) hipLaunchKernelGGL((gemm_dma_kernel<16, 288, 1, 6, W_NT, 3, true>), dim3((unsigned)blocks), dim3(384), 0, stream, a);
    else if (var == 64) hipLaunchKernelGGL((gemm_dma_kernel<16, 288, 1, 6, W_NT, 4, true>), dim3((unsigned)blocks), dim3(384), 0, stream, a);
    else if (var == 92) hipLaunchKernelGGL((gemm_dma_kernel<16, 288, 1, 9, W_NT, 2, true>), dim3((unsigned)blocks), dim3(576), 0, stream, a);
    else if (var == 94) hipLaunchKernelGGL((gemm_dma_kernel<16, 288, 1, 9, W_NT, 4, true>), dim3((unsigned)blocks), dim3(576), 0, stream, a);
    else hipLaunchKernelGGL((gemm_dma_kernel<16, 288, 1, 6, W_NT, 2, true>), dim3((unsigned)blocks), dim3(384), 0, stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { eda_set_error("linear_add_dropout_ln: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

#ifdef EDA_GEMM_PROFILE
extern "C" int eda_gemm_profile_read(unsigned long long *out8) {
  static unsigned long long all[64 * 8], zero[64 * 8];
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(gemm_prof), sizeof(all)) != hipSuccess) return 1;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  for (int s = 0; s < 64; ++s)
    for (int i = 0; i < 8; ++i) out8[i] += all[s * 8 + i];
  return hipMemcpyToSymbol(HIP_SYMBOL(gemm_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif

void eda_gemm_env_reset() { g_dma_mode_v = -2; }

extern "C" int eda_gemm_set_dma(int mode) {
  EDA_CHECK_ARG(mode >= -1 && mode <= 10, "mode: -1 (own selection), 0 (off), 1..10 (configuration for every eligible launch)");
  g_dma_mode_v = mode;
  return 0;
}

// ---- C ABI: plain linear layers ----------------------------------------------------------------

extern "C" int eda_linear_fwd_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                  const float *bias, int relu, float *y, long ldy, void *stream_) {
  EDA_CHECK_ARG(R >= 0 && K > 0 && N > 0 && ldx >= K && ldw >= K && ldy >= N, "bad dimension");
  if (R == 0) return 0;
  EDA_CHECK_ARG(x && w && y, "null pointer");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN;
  a.x = x; a.ldx = ldx; a.R = R; a.K = K;
  a.w = w; a.ldw = ldw; a.N = N; a.bias = bias; a.relu = relu;
  a.y = y; a.ldy = ldy;
  return eda_gemm_launch(a, W_NT, (hipStream_t)stream_);
}

extern "C" int eda_linear_ex_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                 const float *bias, int relu, float drop_p, const unsigned long long *drop_seed,
                                 unsigned drop_salt, const float *gate, long ldgate, float gate_scale, float *y,
                                 long ldy, void *stream_) {
  return eda_linear_ex_ws_f32(x, ldx, R, K, w, ldw, N, bias, relu, drop_p, drop_seed, drop_salt, gate, ldgate, gate_scale, y, ldy,
                              nullptr, 0, stream_);
}

extern "C" int eda_linear_ex_ws_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                    const float *bias, int relu, float drop_p, const unsigned long long *drop_seed,
                                    unsigned drop_salt, const float *gate, long ldgate, float gate_scale, float *y,
                                    long ldy, void *ws, size_t ws_bytes, void *stream_) {
  EDA_CHECK_ARG(R >= 0 && K > 0 && N > 0 && ldx >= K && ldw >= K && ldy >= N, "bad dimension");
  EDA_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || drop_seed), "dropout needs 0 <= p < 1 and a seed");
  EDA_CHECK_ARG(!gate || ldgate >= N, "bad gate stride");
  if (R == 0) return 0;
  EDA_CHECK_ARG(x && w && y, "null pointer");
  EDA_CHECK_ARG((long long)R * N < 0x100000000LL || drop_p == 0.f, "dropout: more than 2^32 elements");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN;
  a.x = x; a.ldx = ldx; a.R = R; a.K = K;
  a.w = w; a.ldw = ldw; a.N = N; a.bias = bias; a.relu = relu;
  a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_salt = drop_salt;
  a.gate = gate; a.ldgate = ldgate; a.gate_scale = gate_scale;
  a.y = y; a.ldy = ldy;
  a.sk_ws = ws; a.sk_ws_bytes = ws_bytes;
  return eda_gemm_launch(a, W_NT, (hipStream_t)stream_);
}

extern "C" int eda_linear_dgrad_f32(const float *dy, long lddy, long R, int N, const float *w, long ldw, int K,
                                    float *dx, long lddx, void *stream_) {
  return eda_linear_dgrad_ws_f32(dy, lddy, R, N, w, ldw, K, dx, lddx, nullptr, 0, stream_);
}

extern "C" int eda_linear_dgrad_ws_f32(const float *dy, long lddy, long R, int N, const float *w, long ldw, int K,
                                       float *dx, long lddx, void *ws, size_t ws_bytes, void *stream_) {
  EDA_CHECK_ARG(R >= 0 && K > 0 && N > 0 && lddy >= N && ldw >= K && lddx >= K, "bad dimension");
  if (R == 0) return 0;
  EDA_CHECK_ARG(dy && w && dx, "null pointer");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN;
  a.x = dy; a.ldx = lddy; a.R = R; a.K = N;      // contraction over the layer's output channels
  a.w = w; a.ldw = ldw; a.N = K;                 // W (N, K) read as (contraction, output column)
  a.y = dx; a.ldy = lddx;
  a.sk_ws = ws; a.sk_ws_bytes = ws_bytes;
  return eda_gemm_launch(a, W_NN, (hipStream_t)stream_);
}

// The same two products with a second TERM in the epilogue: y = x W^T (+ bias) + addend, dx = dy W + addend -- an input
// gradient that has another contribution (the residual branch of a post-norm block, models/encoder_decoder_layers.py:
// 87-105, 231-245) leaves the product's launch complete instead of meeting the other term in an element-wise add.
// addend == y / dx is allowed (a lane reads its four floats before it writes them).  Same order as launch + add:
// acc, + bias, + addend -- bitwise equal to the two-launch form.
extern "C" int eda_linear_addend_ws_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                        const float *bias, const float *addend, long ldadd, float *y, long ldy, void *ws,
                                        size_t ws_bytes, void *stream_) {
  EDA_CHECK_ARG(R >= 0 && K > 0 && N > 0 && ldx >= K && ldw >= K && ldy >= N && ldadd >= N, "bad dimension");
  if (R == 0) return 0;
  EDA_CHECK_ARG(x && w && y && addend, "null pointer");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN;
  a.x = x; a.ldx = ldx; a.R = R; a.K = K;
  a.w = w; a.ldw = ldw; a.N = N; a.bias = bias;
  a.gate = addend; a.ldgate = ldadd; a.gate_scale = 1.f; a.gate_mode = 1;
  a.y = y; a.ldy = ldy;
  a.sk_ws = ws; a.sk_ws_bytes = ws_bytes;
  return eda_gemm_launch(a, W_NT, (hipStream_t)stream_);
}

extern "C" int eda_linear_dgrad_addend_ws_f32(const float *dy, long lddy, long R, int N, const float *w, long ldw, int K,
                                              const float *addend, long ldadd, float *dx, long lddx, void *ws,
                                              size_t ws_bytes, void *stream_) {
  EDA_CHECK_ARG(R >= 0 && K > 0 && N > 0 && lddy >= N && ldw >= K && lddx >= K && ldadd >= K, "bad dimension");
  if (R == 0) return 0;
  EDA_CHECK_ARG(dy && w && dx && addend, "null pointer");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN;
  a.x = dy; a.ldx = lddy; a.R = R; a.K = N;
  a.w = w; a.ldw = ldw; a.N = K;
  a.gate = addend; a.ldgate = ldadd; a.gate_scale = 1.f; a.gate_mode = 1;
  a.y = dx; a.ldy = lddx;
  a.sk_ws = ws; a.sk_ws_bytes = ws_bytes;
  return eda_gemm_launch(a, W_NN, (hipStream_t)stream_);
}

// contraction = K for the forward form (x (R, K), y (R, N)); for the input-gradient form pass (R, N_layer, K_layer)
extern "C" size_t eda_linear_splitk_workspace_bytes(long R, int K, int N) {
  if (R <= 0 || K <= 0 || N <= 0) return 0;
  const int s = splitk_slices(R, K, N);
  return s > 1 ? splitk_bytes(R, N, s) : 0;
}


// ---- C ABI: sibling linear layers in one launch -------------------------------------------------
namespace {
int launch_grouped(GemmArgs &a, int wmode, hipStream_t stream) {
  // every group must satisfy the 16-byte conditions for the fast kernel; otherwise the element-wise one
  bool vec = true;
  int nmax = 0;
  for (int g = 0; g < a.ngroups; ++g) {
    GemmArgs t = a;
    t.x = a.grp[g].x; t.ldx = a.grp[g].ldx; t.w = a.grp[g].w; t.ldw = a.grp[g].ldw; t.K = a.grp[g].K; t.N = a.grp[g].N;
    vec = vec && gemm_vec_ok(t, wmode) && !(wmode == W_NN && (t.ldw % 4 != 0 || (reinterpret_cast<uintptr_t>(t.w) & 15u) != 0));
    if (t.N > nmax) nmax = t.N;
  }
  a.N = nmax; a.K = a.grp[0].K; a.x = a.grp[0].x; a.w = a.grp[0].w; a.y = a.grp[0].y;
  a.ldx = a.grp[0].ldx; a.ldw = a.grp[0].ldw; a.ldy = a.grp[0].ldy;
  if (!vec) return launch_cfg<1, 4, false>(a, wmode, stream);
  return launch_cfg<1, 1, true, 2, 2>(a, wmode, stream);            // 32 x 32 tiles: these are small launches
}
}  // namespace

extern "C" int eda_linear_grouped_fwd_f32(int ngroups, const float *const *x, const long *ldx, long R, const int *K,
                                          const float *const *w, const long *ldw, const int *N,
                                          const float *const *bias, int relu, float *const *y, const long *ldy,
                                          void *stream_) {
  EDA_CHECK_ARG(ngroups >= 1 && ngroups <= G_MAXGROUPS && R >= 0, "1..24 groups");
  if (R == 0) return 0;
  EDA_CHECK_ARG(x && ldx && K && w && ldw && N && y && ldy, "null pointer");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN; a.R = R; a.relu = relu; a.ngroups = ngroups;
  for (int g = 0; g < ngroups; ++g) {
    EDA_CHECK_ARG(x[g] && w[g] && y[g] && K[g] > 0 && N[g] > 0 && ldx[g] >= K[g] && ldw[g] >= K[g] && ldy[g] >= N[g], "bad group");
    a.grp[g] = GemmGroup{x[g], ldx[g], w[g], ldw[g], bias ? bias[g] : nullptr, y[g], ldy[g], K[g], N[g], 0};
  }
  if (ngroups == 1) {
    a.ngroups = 0;
    a.x = x[0]; a.ldx = ldx[0]; a.K = K[0]; a.w = w[0]; a.ldw = ldw[0]; a.N = N[0]; a.bias = bias ? bias[0] : nullptr;
    a.y = y[0]; a.ldy = ldy[0];
    return eda_gemm_launch(a, W_NT, (hipStream_t)stream_);
  }
  return launch_grouped(a, W_NT, (hipStream_t)stream_);
}

extern "C" int eda_linear_grouped_dgrad_f32(int ngroups, const float *const *dy, const long *lddy, long R, const int *N,
                                            const float *const *w, const long *ldw, const int *K, float *const *dx,
                                            const long *lddx, void *stream_) {
  EDA_CHECK_ARG(ngroups >= 1 && ngroups <= G_MAXGROUPS && R >= 0, "1..24 groups");
  if (R == 0) return 0;
  EDA_CHECK_ARG(dy && lddy && K && w && ldw && N && dx && lddx, "null pointer");
  GemmArgs a;
  gemm_defaults(a);
  a.xmode = X_PLAIN; a.epi = E_PLAIN; a.R = R; a.ngroups = ngroups;
  for (int g = 0; g < ngroups; ++g) {
    EDA_CHECK_ARG(dy[g] && w[g] && dx[g] && K[g] > 0 && N[g] > 0 && lddy[g] >= N[g] && ldw[g] >= K[g] && lddx[g] >= K[g], "bad group");
    // contraction over the layer's N outputs, K output columns
    a.grp[g] = GemmGroup{dy[g], lddy[g], w[g], ldw[g], nullptr, dx[g], lddx[g], N[g], K[g], 0};
  }
  if (ngroups == 1) {
    a.ngroups = 0;
    a.x = dy[0]; a.ldx = lddy[0]; a.K = N[0]; a.w = w[0]; a.ldw = ldw[0]; a.N = K[0]; a.y = dx[0]; a.ldy = lddx[0];
    return eda_gemm_launch(a, W_NN, (hipStream_t)stream_);
  }
  return launch_grouped(a, W_NN, (hipStream_t)stream_);
}
