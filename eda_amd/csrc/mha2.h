// mha2.h -- launch interface of the attention kernels (csrc/mha2.hip) under the C ABI eda_mha_fwd / eda_mha_bwd.
#pragma once
#include <hip/hip_runtime.h>

struct Mha2Args {
  const float *q, *k, *v;                     // head h at column h*36 of a row; rows strided
  long q_sb, q_sl, k_sb, k_sl, v_sb, v_sl;    // element strides (batch, row)
  float *o; long o_sb, o_sl;                  // forward output / saved output (B,Lq,288)
  float *lse;                                 // (B,H,Lq) natural-log log-sum-exp of the scaled masked scores
  const unsigned char *mask;                  // (B,Lk) 1 = ignore, or null
  int B, H, Lq, Lk;
  float scale, p_drop;
  const unsigned long long *seed_ptr; unsigned salt;
  // backward
  const float *dout; long do_sb, do_sl;
  float *dq, *dk, *dv;
  long dq_sb, dq_sl, dk_sb, dk_sl, dv_sb, dv_sl;
  // decomposition (filled by the launchers)
  int n_kb;          // bwd: key blocks of 16*NW keys        fwd: unused
  int n_qs;          // bwd: query splits                    fwd: query blocks of 16*NQ queries
  int q_per_wg;      // bwd: queries per query split (multiple of the chunk)
  int prio_mode;     // 1 = s_setprio by remaining work in phase A (EDA_MHA2_PRIO)
  int dtype;         // EDA_DTYPE_F32 (exact fp32 MFMA: the parity path) / BF16 / F16 contractions, fp32 accumulate
  float *dq_part;    // bwd: [key block][B][Lq][H*36] dense partials of dQ (n_kb > 1)
  float *dkv_part;   // bwd: [query split][dk | dv][B][Lk][H*36] dense partials (n_qs > 1)
  int bwd_merge;           // bwd: 1 = the split ranges are merged inside the launch (tickets), 0 = by mha2_part_reduce_kernel
  unsigned *bwd_tickets;   // bwd: [B*H][n_qs] arrivals of the key blocks at a dQ range, then [B*H][n_kb] of the query splits at a
                           // key block's dK | dV: zero before the launch, left zero (the last arriver merges and re-arms)
  // q-projection fused in front of the forward (eda_mha_qproj_fwd): q = xq Wq^T + bq is computed per (query block, head)
  // inside the attention launch and WRITTEN to q_out (the backward reads it like any projected q)
  const float *xq; long xq_sb, xq_sl;         // (B, Lq, H*36) input rows of the q-projection
  const float *wq; long ldwq;                 // (H*36, H*36) weight, row-major (row = output column)
  const float *bq;                            // (H*36) bias or null
  float *q_out; long qo_sb, qo_sl;
  // key-split forward (flash-decoding form, eda_mha_fwd_ws): n_ksplit workgroups per (scene, head, query block), each
  // over keys_per_split keys (a multiple of the chunk); their (O, m, l) partials meet in `fwd_part` and the LAST arriver
  // of a block (ticket in `fwd_tickets`, left at zero again) merges them in split order and writes out / lse
  int n_ksplit, keys_per_split;
  int dbg3;                  // mha3.hip ablation bits (EDA_MHA3_DBG; results are then wrong): 1 convert once, 2 no softmax, 4 no PV, 8 no QK^T, 16 no dropout
  float *fwd_part;           // [block][split][NQ][64 lanes][16 floats]
  unsigned *fwd_tickets;     // [block], zero before the first call
};

// 0 = launched, EDA_ERR_* otherwise.  Both enqueue on `stream` only, allocate nothing, never synchronise.
int eda_mha2_fwd_launch(Mha2Args &a, void *ws, size_t ws_bytes, hipStream_t stream);
size_t eda_mha2_fwd_workspace_bytes(int B, int H, int Lq, int Lk);
// mha3.hip: the long-key forward on v_mfma_f32_16x16x32_bf16 (bf16 x 3 for F32, plain for BF16); -1 = not its shape
int eda_mha3_fwd_launch(Mha2Args &a, hipStream_t stream);
// mha4.hip: a short query set (<= 144, on request <= 256) against >= 512 keys, keys per wave, fp32 MFMA; needs the workspace
bool eda_mha4_takes(int dtype, int Lq, int Lk);
size_t eda_mha4_fwd_workspace_bytes(int B, int H, int Lq, int Lk);
int eda_mha4_fwd_launch(Mha2Args &a, void *ws, size_t ws_bytes, hipStream_t stream);
int eda_mha2_qproj_fwd_launch(Mha2Args &a, hipStream_t stream);      // Lk <= 192
int eda_mha2_bwd_launch(Mha2Args &a, void *ws, size_t ws_bytes, unsigned *tickets, size_t tickets_bytes, hipStream_t stream);
size_t eda_mha2_bwd_workspace_bytes(int B, int H, int Lq, int Lk);
size_t eda_mha2_bwd_ticket_bytes(int B, int H, int Lq, int Lk);
