// gemm.h -- internal interface of gemm.hip / wgrad.hip (not part of the C ABI).
#pragma once
#include "eda_common.h"

constexpr int G_THREADS = 256;
constexpr int G_KC = 32;          // contraction chunk (floats)
constexpr int G_XS = G_KC + 8;    // LDS row stride of [row][k] tiles

enum { X_PLAIN = 0, X_BNRELU = 1, X_GATHER = 2, X_BNBWDPOOL = 3 };
enum { W_NT = 0, W_NN = 1 };
enum { E_PLAIN = 0, E_STATS = 1, E_MASK = 2, E_SCATTER = 3 };

// one problem of a grouped launch: same row count, own operands / widths (plain rows, plain epilogue)
struct GemmGroup { const float *x; long ldx; const float *w; long ldw; const float *bias; float *y; long ldy; int K, N, ct0; };
constexpr int G_MAXGROUPS = 24;     // (the seven prediction heads' 3 x 7 sibling stacks in one backward launch: eda_amd/heads_batched.py)

struct GemmArgs {
  int xmode, epi;
  // row operand
  const float *x; long ldx; long R; int K;
  const float *in_scale, *in_shift;
  // X_GATHER / E_SCATTER geometry: rows are (scene, centre, neighbour)
  const float *xyz, *new_xyz, *feats; const int *idx;
  int n_pts, m, ns, c_feat; float inv_radius;
  // weight: W_NT (N, K) row-major; W_NN (K, N) row-major.  w_elem: W_NN rows are not 16-byte
  // addressable (set by the launcher).  X_GATHER: w is the layer's (N, 3 + c_feat) weight.
  const float *w; long ldw; int N; int w_elem;
  const float *bias; int relu;
  float *y; long ldy;
  // E_STATS
  double *sum, *sumsq; unsigned *ticket;
  const float *gamma, *beta; float eps, momentum;
  float *running_mean, *running_var, *mean_out, *rstd_out, *scale_out, *shift_out;
  // E_MASK (previous layer's pre-activation and BatchNorm constants)
  const float *zm; long ldzm; const float *m_scale, *m_shift, *m_mean, *m_rstd; double *s1, *s2;
  // E_SCATTER
  float *dfeats;
  int col_tiles; long row_blocks; int row_slots; unsigned ticket_target;
  int ngroups; GemmGroup grp[G_MAXGROUPS];     // ngroups > 1: block column tile ct belongs to the group with ct0 <= ct
  // X_BNBWDPOOL (streaming kernels only): the row operand is dz of a POOLED last SharedMLP layer, formed while staging
  // from the layer's pre-activation x = z (R x K): with y = z*sc+sh, d = (y > 0 and argmax[row/pool][k] == row%pool) ?
  // dout[row/pool][k] : 0, dz = ka*d + kb*z + kd (train-mode BatchNorm backward; bn_relu_bwd_apply_kernel<true> writes
  // the same values to HBM when this prologue is not used).  bb_consts: 5 x K floats {sc, sh, ka, kb, kd}; pool % 16 == 0
  const unsigned char *bb_argmax; const float *bb_dout; const float *bb_consts; int bb_pool;
  // E_PLAIN extras (eda_linear_ex_f32): Dropout after the bias / ReLU (counter-based hash of (seed, salt, element), the
  // scheme of ln.hip), and a gate: y = gate > 0 ? y * gate_scale : 0 -- the ReLU (+ Dropout) backward of the layer
  // whose activated output `gate` is, applied to the input gradient that flows into it
  float drop_p; const unsigned long long *drop_seed; unsigned drop_salt;
  const float *gate; long ldgate; float gate_scale;
  int gate_mode;        // 0: the gate above; 1: `gate` is an ADDEND, y = acc (+ bias, ...) + gate[row][col] (the second term of
                        // an input gradient: eda_linear_addend_ws_f32; gate == y allowed: a lane reads its four floats first)
  // gemm_dma_kernel<..., LN = true> (eda_linear_add_dropout_ln_fwd_f32): the row block spans all N columns and the
  // epilogue is  z = resid + Dropout(acc + bias),  out = LayerNorm(z) * gamma + beta  (+ out_pos = out + pos); y is unused
  const float *ln_resid, *ln_gamma, *ln_beta, *ln_pos; float ln_eps;
  float *ln_z, *ln_out, *ln_out_pos, *ln_mean, *ln_rstd;
  int defer_finalize;   // E_STATS: the last workgroup only re-arms the ticket; the column sums stay in `sum` / `sumsq`
                        // for a cross-rank all-reduce, a separate kernel finalises (sa_cl.hip, eda_set_bn_sync)
  // gemm_dma_kernel<..., SK = true>: the contraction of a tile is divided over sk_slices WORKGROUPS (few tiles against a
  // long contraction: the text encoder's 640 x 3072 -> 768, the 3456-deep input gradients of the hoisted K | V
  // projections); fp32 partial tiles meet in sk_part, the last arriver of a tile (sk_tickets, left at zero) adds them in
  // slice order and runs the epilogue
  float *sk_part; unsigned *sk_tickets; int sk_slices, sk_cps; long sk_per;
  void *sk_ws; size_t sk_ws_bytes;          // (host side: the caller's scratch for it, eda_linear_splitk_workspace_bytes)
  int dbg;   // EDA_GEMM_DBG timing experiments (results are then wrong): 1 skip park, 2 no grid cap, 4 skip atomics, 8 skip z loads
};


// Launch the row GEMM described by `a` (tile shape chosen from R and N).  wmode: W_NT / W_NN.
int eda_gemm_launch(GemmArgs &a, int wmode, hipStream_t stream);
// would the streaming kernels take this launch (same checks as eda_gemm_launch, nothing is launched)?
bool eda_gemm_stream_takes(const GemmArgs &a, int wmode);

// ---- weight gradient with a row-operand prologue (wgrad.hip) ------------------------------------
// dW (M, N) = dY^T X over R rows, X = plain rows | relu(Z*scale+shift) | gathered neighbourhood rows
// (column order [dx dy dz 0 | feats], written back to the layer's (M, 3 + c_feat) weight layout).
struct WgradXArgs {
  int xmode;
  const float *dy; long ld_dy; long R; int M;
  const float *x; long ld_x; int N;                 // X_GATHER: N = 4 + c_feat
  const float *in_scale, *in_shift;                 // X_BNRELU
  const float *xyz, *new_xyz, *feats; const int *idx;
  int n_pts, m, ns, c_feat; float inv_radius;
  float *dW;                                        // (M, N) or, X_GATHER, (M, 3 + c_feat)
  float *ws; size_t ws_bytes;
  // dy_pool > 0: dy is not stored; it is the dz of GemmArgs::X_BNBWDPOOL, formed from dyz (= z, R x M), dy_argmax,
  // dy_dout (R/pool x M) and dy_consts (5 x M) while staging (xmode must be X_BNRELU)
  const float *dyz; const unsigned char *dy_argmax; const float *dy_dout; const float *dy_consts; int dy_pool;
  // dy_bn: dy is the MASKED gradient g of a non-pooled layer's output and the layer's dz = ka*g + kb*z + kd is formed while
  // staging (z = dyz (R x M), {., ., ka, kb, kd} = dy_consts (5 x M)): bn_relu_bwd_apply_kernel's pass is not run
  int dy_bn;
  // dx_out != NULL (sa_layer_bwd_kernel: xmode X_BNRELU, M and N in {64, 128}, ld_x = N): the layer's input gradient in the
  // same launch -- g_prev = (dz W) where the input's ReLU was open -> dx_out (R x N), its BatchNorm-backward column sums
  // (sum g_prev, sum g_prev * xhat) -> dx_s1 / dx_s2 (fp64 atomics); dx_w = the layer's (M, N) weight
  const float *dx_w; long dx_ldw; float *dx_out; const float *dx_mean, *dx_rstd; double *dx_s1, *dx_s2;
  float *dx_scatter;
};
bool eda_wgrad_x_fuses_dx(int M, int N);
// xmode X_GATHER with dx_out = scratch rows (R x c_feat), dx_scatter = d(features) (B * n_pts x c_feat, zeroed by the caller),
// dx_w = the (M, 3 + c_feat) weight and dy_bn: the first layer's weight gradient and input-gradient rows in one launch
// (sa_gather_layer_bwd_kernel), then their scatter-add (rows_scatter_add_kernel)
bool eda_wgrad_x_fuses_gather(int M, int c_feat);
size_t eda_wgrad_x_workspace_bytes(long R, int M, int N);
int eda_wgrad_x_launch(const WgradXArgs &a, hipStream_t stream);
