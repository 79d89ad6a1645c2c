/*
 * eda_hip.h -- C ABI of libeda_hip.so, the MI355X (gfx950) implementation of
 * the hot path of yanmin-wu/EDA: the nine ops of pointnet2/_ext_src plus the
 * fused kernels built on them.
 *
 * This is the drop-in boundary.  Each entry point replaces one function the
 * reference binds through pybind11 (pointnet2/_ext_src/src/bindings.cpp:11-24);
 * the citation on each declaration names the reference host dispatcher and
 * CUDA kernel it replaces (paths relative to /root/reference/pointnet2/_ext_src).
 *
 * Conventions
 *  - plain pointers + int32 dims; all pointers are DEVICE pointers (HBM);
 *    fp32 / int32 only, dense row-major ("contiguous") exactly as the
 *    reference requires (include/utils.h:15-30).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every
 *    call only ENQUEUES work on that stream: no allocation, no host sync,
 *    graph-capturable.  Scratch memory comes from the caller (`ws`).
 *  - return value: 0 on success, otherwise a hipError_t (or EDA_ERR_*) value;
 *    never exit()s (the reference does, include/cuda_utils.h:35-44).
 *    eda_last_error_string() describes the last failure on this thread.
 *  - outputs are written completely by the kernels (the reference relies on
 *    torch::zeros + partial writes; results are identical).  The *_grad entry
 *    points zero their output themselves before accumulating.
 *  - re-entrant.  PROCESS-WIDE state (one process drives one GPU, as the
 *    reference's one-process-per-GPU launcher does, train_dist_mod.py; the
 *    values are not per device and not per thread):
 *      eda_set_fma_mode          arithmetic form of the squared distances
 *      eda_fps_set_cu_reserve    CUs the cluster sampler leaves to other streams
 *      eda_fps_set_background    cluster sampler as a prefetch underneath other work
 *      eda_fps_set_policy        cluster / bucket / auto sampler
 *      eda_gemm_set_dma          kernel selection of the plain row products
 *      eda_wgrad_set_arith       fp32 MFMA or bf16 x 3 weight gradients
 *      eda_set_bn_sync           SyncBatchNorm hook + world size
 *      eda_set_deterministic     ordered reductions instead of fp32 atomics
 *    plus the EDA_* environment knobs, read ONCE into one table on first use
 *    (eda_reload_env() re-reads them; tests only).  Nothing else persists
 *    between calls except per-(kernel, device) launch attributes.
 *  - exported symbols are exactly the names declared here (the library is
 *    built with -fvisibility=hidden; tests/test_abi.py compares nm -D).
 */
#ifndef EDA_HIP_H
#define EDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define EDA_HIP_ABI_VERSION 1

#define EDA_ERR_INVALID_ARG   10001
#define EDA_ERR_WORKSPACE     10002
#define EDA_ERR_UNSUPPORTED   10003
#define EDA_ERR_PEER_SELFTEST 10004

int         eda_version(void);
const char *eda_last_error_string(void);

/* Arithmetic of the fp32 squared-distance expression (see DESIGN.md
 * "Canonical arithmetic").  0 = nvcc -fmad=true contraction
 * [t=dy*dy; t=fma(dx,dx,t); t=fma(dz,dz,t)] (default), 1 = no contraction. */
int eda_set_fma_mode(int mode);
int eda_get_fma_mode(void);

/* Re-read every EDA_* environment knob of the library (they are read once, on first use, into one
 * process-wide table; a test that flips a knob inside a process calls this afterwards).  Returns 0. */
int eda_reload_env(void);

/* 1: every floating-point reduction of the backward ops runs in a fixed order (sorted-segment / per-owner
 * sums instead of fp32 atomics: group_points_grad, gather_points_grad, three_interpolate_grad, the fused
 * set-abstraction backward's scatter), so two runs of the same step are bit-identical -- the counterpart of
 * the reference's cudnn.deterministic = True (train_dist_mod.py:342-344).  0 (default): atomics.
 * Process-wide; default from EDA_DETERMINISTIC.  Returns 0, or EDA_ERR_INVALID_ARG. */
int eda_set_deterministic(int on);
int eda_get_deterministic(void);
/* out (P, C) = for every row p the sum, in ascending r, of the rows src[r] (R, C) with idx[r] == p (zero where none):
 * the ordered form of an index_add_ / embedding weight gradient (models/bdetr.py:150-156 keeps the class-embedding table
 * trainable); the same per-owner kernel serves every gradient scatter of the library in the deterministic mode. */
int eda_index_add_rows_ordered_f32(const float *src, const int *idx, long R, int C, int P, float *out, void *stream);

/* ---- furthest point sampling ------------------------------------------
 * replaces furthest_point_sampling()            src/sampling.cpp:70-91
 *          furthest_point_sampling_kernel<bs>   src/sampling_gpu.cu:74-234
 * xyz (b,n,3) f32 -> idx (b,m) i32.  The running min-distance array the
 * reference keeps in a (b,n) global `temp` tensor lives in registers here.
 * ws: eda_fps_workspace_bytes(b,n,m) bytes of device scratch.  Its first four int32 are a STICKY
 * status block that the call never clears: the caller zeroes them once and may keep one workspace
 * per stream for all calls; everything behind them (inter-workgroup mailboxes) is zeroed on `stream`
 * by the call itself.  ws[0] != 0 after the stream has drained reports that a call gave up its
 * bounded inter-workgroup spin (the scene's workgroups were not co-resident, e.g. another kernel
 * held the CUs); the samples it had not yet produced are then index 0 (valid, but not FPS).      */
size_t eda_fps_workspace_bytes(int b, int n, int m);
int eda_furthest_point_sampling_f32(const float *xyz, int b, int n, int m,
                                    int *idx, void *ws, size_t ws_bytes,
                                    void *stream);

/* Same result as eda_furthest_point_sampling_f32 for EVERY input, fast when xyz already is in sampling
 * order (the SA2..SA4 levels sample from the previous level's samples, models/backbone_module.py:122-131):
 * two barrier-free kernels check, under the reference's exact arithmetic and tie order, whether 0..m-1 is
 * the answer; scenes that pass are done (~20 us instead of 0.1-0.6 ms of dependent rounds), the others run
 * the regular kernel.  n > 8192 forwards to the regular entry point.                                  */
size_t eda_fps_prefix_workspace_bytes(int b, int n, int m);
int eda_furthest_point_sampling_prefix_f32(const float *xyz, int b, int n, int m, int *idx, void *ws,
                                           size_t ws_bytes, void *stream);

/* ---- gather -----------------------------------------------------------
 * replaces gather_points()        src/sampling.cpp:20-43, sampling_gpu.cu:13-35
 * points (b,c,n), idx (b,m) -> out (b,c,m)                                 */
int eda_gather_points_f32(const float *points, const int *idx, int b, int c,
                          int n, int m, float *out, void *stream);
/* replaces gather_points_grad()   src/sampling.cpp:45-69, sampling_gpu.cu:39-62
 * grad_out (b,c,m), idx (b,m) -> grad_points (b,c,n) (zeroed here)          */
int eda_gather_points_grad_f32(const float *grad_out, const int *idx, int b,
                               int c, int n, int m, float *grad_points,
                               void *stream);

/* ---- ball query -------------------------------------------------------
 * replaces ball_query()               src/ball_query.cpp:13-37
 *          query_ball_point_kernel    src/ball_query_gpu.cu:14-59
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample): the first `nsample`
 * points with d2 < radius^2 in ascending index order, padded with the first
 * hit; an empty ball yields an all-zero row.

 * ws: optional scratch of eda_ball_query_workspace_bytes(b,n,m) bytes.  With it,
 * scenes of n >= 4096 points use a uniform-grid candidate search (cells >= 1.001 *
 * radius, 3x3x3 neighbourhood, hits ranked by index) instead of the brute-force
 * index-ordered scan; the result is identical.  ws == NULL always scans.       */
size_t eda_ball_query_workspace_bytes(int b, int n, int m);
int eda_ball_query_f32(const float *new_xyz, const float *xyz, int b, int n,
                       int m, float radius, int nsample, int *idx,
                       void *ws, size_t ws_bytes, void *stream);

/* ---- grouping ---------------------------------------------------------
 * replaces group_points()        src/group_points.cpp:17-40, group_points_gpu.cu:13-45
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample)      */
int eda_group_points_f32(const float *points, const int *idx, int b, int c,
                         int n, int npoints, int nsample, float *out,
                         void *stream);
/* replaces group_points_grad()   src/group_points.cpp:42-65, group_points_gpu.cu:48-80
 * grad_out (b,c,npoints,nsample) -> grad_points (b,c,n) (zeroed here)       */
int eda_group_points_grad_f32(const float *grad_out, const int *idx, int b,
                              int c, int n, int npoints, int nsample,
                              float *grad_points, void *stream);

/* ---- 3-NN interpolation -----------------------------------------------
 * replaces three_nn()            src/interpolate.cpp:19-45, interpolate_gpu.cu:14-74
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32 (squared), idx (b,n,3) */
int eda_three_nn_f32(const float *unknown, const float *known, int b, int n,
                     int m, float *dist2, int *idx, void *stream);
/* replaces three_interpolate()   src/interpolate.cpp:47-75, interpolate_gpu.cu:77-116
 * points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n)                */
int eda_three_interpolate_f32(const float *points, const int *idx,
                              const float *weight, int b, int c, int m, int n,
                              float *out, void *stream);
/* replaces three_interpolate_grad() src/interpolate.cpp:76-104, interpolate_gpu.cu:121-159
 * grad_out (b,c,n) -> grad_points (b,c,m) (zeroed here)                     */
int eda_three_interpolate_grad_f32(const float *grad_out, const int *idx,
                                   const float *weight, int b, int c, int n,
                                   int m, float *grad_points, void *stream);

/* ---- fused multi-head attention ---------------------------------------
 * replaces the QK^T -> key-padding mask -> softmax -> dropout -> PV core of
 * torch.nn.MultiheadAttention as the reference calls it
 * (models/encoder_decoder_layers.py:87-93,99-105,111-117,149-153,179-183,366-401;
 * 8 heads x 36, attn_mask=None, boolean key_padding_mask, dropout 0.1).
 * q/k/v: PROJECTED activations, head h in columns [h*36, h*36+36) of each row;
 * element strides (batch, row) given explicitly so slices of a packed QKV GEMM
 * output are consumed in place.  out (B,Lq,H*36) dense; lse (B,H,Lq) scratch kept
 * for the backward.  key_padding_mask (B,Lk) bytes, 1 = ignore, or NULL.
 * Dropout (p_drop > 0): keep mask = hash(*seed_ptr, salt, b,h,q,k), regenerated
 * identically by the backward; seed_ptr is a DEVICE counter so that a replayed
 * HIP graph draws a fresh mask whenever the host bumps the counter.
 * fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32 (csrc/mha2.hip); the forward
 * for >= 512 queries against >= 512 keys runs on v_mfma_f32_16x16x32_bf16 with
 * the operands split into three bf16 planes (h + m + l = the fp32 value exactly,
 * six plane products, fp32 softmax and accumulators: csrc/mha3.hip; same error
 * bounds against fp64, EDA_MHA3=0 keeps the fp32-MFMA kernel).               */
int eda_mha_fwd_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl,
                    long k_sb, long k_sl, long v_sb, long v_sl,
                    const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                    int head_dim, float scale, float p_drop,
                    const unsigned long long *seed_ptr, unsigned salt, float *out,
                    float *lse, void *stream);
/* Backward of the above: dq (B,Lq,H*36), dk, dv (B,Lk,H*36) with their own element strides
 * (batch, row), so that the gradients of a packed q|k|v projection land in one buffer;
 * delta_ws: (B,H,Lq) floats of scratch.                                      */
int eda_mha_bwd_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl,
                    long k_sb, long k_sl, long v_sb, long v_sl,
                    const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                    int head_dim, float scale, float p_drop,
                    const unsigned long long *seed_ptr, unsigned salt, const float *out,
                    const float *lse, const float *dout, long do_sb, long do_sl,
                    float *delta_ws, float *dq, float *dk, float *dv, long dq_sb, long dq_sl,
                    long dk_sb, long dk_sl, long dv_sb, long dv_sl, void *ws, size_t ws_bytes,
                    void *stream);
/* Forward-only attention for head_dim 64 (csrc/mha_hd64.hip): the self-attention of the frozen RoBERTa-base text
 * encoder (models/bdetr.py:77-80, 170-175 -> transformers RobertaSelfAttention: softmax(q k^T * scale + padding mask) v).
 * q (B,Lq,.), k/v (B,Lk,.) with head h at columns [64h, 64h+64), element strides as in eda_mha_fwd_f32;
 * key_padding_mask (B,Lk) bytes, 1 = ignore, or NULL; out (B,Lq,H*64) dense.  1 <= Lk <= 256.  No dropout, no
 * backward: the encoder is frozen (requires_grad = False, bdetr.py:78-80). */
int eda_mha_fwd_hd64_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                         long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                         float scale, float *out, void *stream);
/* The same with dropout on the probabilities (RobertaSelfAttention.dropout) and eda_dropout_f32 for RobertaEmbeddings.dropout:
 * the reference trains with the WHOLE model in train mode (main_utils.py:459 model.train()), so its frozen text encoder runs
 * with its dropout layers active.  Masks: the library's counter-based hash (seed_ptr = the device-side counter, one salt per
 * call site), as in eda_mha_fwd_f32 / eda_add_dropout_ln_fwd_f32.  p_drop = 0: the entry above. */
int eda_mha_fwd_hd64_drop_f32(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                              long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                              float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt, float *out,
                              void *stream);
int eda_dropout_f32(const float *x, long n, float p_drop, const unsigned long long *seed_ptr, unsigned salt, float *out,
                    void *stream);

/* ws: scratch of eda_mha_bwd_workspace_bytes() (0 = none needed: ws may be NULL), required when non-zero:
 * the split ranges' tickets and dense partials ("eda_mha_bwd and the split ranges" below).      */
size_t eda_mha_bwd_workspace_bytes(int B, int H, int Lq, int Lk);

/* The same two entry points with the arithmetic type of the QK^T / PV contractions as an argument
 * (SURVEY.md 8b "dtype enum"; BASELINE.json configs[2] bf16, configs[4] fp16).  Tensors stay fp32 at
 * the boundary; F32 forwards to the fp32 kernels above (v_mfma_f32_16x16x4_f32, the parity path);
 * BF16 / F16 convert the operands on the fly and run v_mfma_f32_16x16x16_{bf16,f16} with fp32
 * accumulation and fp32 softmax (the same kernels of csrc/mha2.hip with packed contraction quads; nothing padded, only
 * useful FLOPs).  The 16-bit kernels ignore ws.                                                 */
#define EDA_DTYPE_F32  0
#define EDA_DTYPE_BF16 1
#define EDA_DTYPE_F16  2
int eda_mha_fwd(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                float *out, float *lse, int dtype, void *stream);
int eda_mha_bwd(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                const float *out, const float *lse, const float *dout, long do_sb, long do_sl,
                float *delta_ws, float *dq, float *dk, float *dv, long dq_sb, long dq_sl, long dk_sb,
                long dk_sl, long dv_sb, long dv_sl, void *ws, size_t ws_bytes, int dtype, void *stream);

/* eda_mha_bwd and the split ranges (round 6).  When a shape's key range (Lk > 256: key blocks of 256) or query range (short
 * key sets: query splits, so that every CU holds a workgroup) is divided, each workgroup leaves a dense partial in `ws` with
 * write-through stores and takes an arrival ticket; the LAST workgroup of a range sums the partials in split order (the bits
 * do not depend on who is last) and writes dq / dk / dv -- inside the same launch (until round 5 a second launch did).
 * Layout of ws: [eda_mha_bwd_ticket_bytes() ticket words | partials]; eda_mha_bwd_workspace_bytes() covers both.  Ticket
 * words must be ZERO before the launch and are left zero: eda_mha_bwd zeroes the leading area with a fill kernel;
 * eda_mha_bwd_tk takes them from a PERSISTENT buffer of the caller (zeroed once; one per stream) and launches nothing else. */
size_t eda_mha_bwd_ticket_bytes(int B, int H, int Lq, int Lk);
int eda_mha_bwd_tk(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                   long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                   int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                   const float *out, const float *lse, const float *dout, long do_sb, long do_sl,
                   float *delta_ws, float *dq, float *dk, float *dv, long dq_sb, long dq_sl, long dk_sb,
                   long dk_sl, long dv_sb, long dv_sl, void *ws, size_t ws_bytes, void *tickets, size_t tickets_bytes,
                   int dtype, void *stream);

/* eda_mha_fwd with scratch for the KEY-SPLIT forward (F32 contractions): for at most 256 queries against >= 512 keys
 * (the text -> point and query -> point cross-attention, models/encoder_decoder_layers.py:87-93, 393-399) the keys of a
 * (scene, head, query block) are divided over several workgroups, whose (O, max, sum) partials meet in `ws`; the last
 * workgroup of a block to arrive merges them in split order (the result does not depend on which one that is) and
 * writes out / lse.  eda_mha_fwd_workspace_bytes() = 0: the shape is not split, ws may be NULL.  The first
 * ceil(4 * blocks / 256) * 256 bytes of ws are ticket words: ZERO before the first call, left zero by every call (one
 * workspace may serve every call of one stream; calls on different streams need their own).  ws == NULL = eda_mha_fwd. */
size_t eda_mha_fwd_workspace_bytes(int B, int H, int Lq, int Lk);
int eda_mha_fwd_ws(const float *q, const float *k, const float *v, long q_sb, long q_sl, long k_sb, long k_sl,
                   long v_sb, long v_sl, const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk,
                   int head_dim, float scale, float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                   float *out, float *lse, int dtype, void *ws, size_t ws_bytes, void *stream);

/* The q-projection fused in front of the forward (csrc/mha2.hip, mha2_qproj_fwd_kernel): one launch computes
 * q = x Wq^T + bq for a site whose key set is short (1 <= Lk <= 192: the text tokens / detected boxes of
 * models/encoder_decoder_layers.py:99-117, 375-391 -> nn.MultiheadAttention's in-projection of the query followed by
 * the attention core) and runs eda_mha_fwd's arithmetic on it.  x (B,Lq,H*36) with element strides (x_sb, x_sl); wq
 * (H*36, H*36) row-major = rows [0, d) of in_proj_weight, leading dimension ldwq; bq (H*36) or NULL; k, v, mask, scale,
 * dropout, out, lse, dtype as in eda_mha_fwd.  q_out (B,Lq,H*36) with strides (q_sb, q_sl) receives the projected,
 * UNSCALED q: eda_mha_bwd takes it as its `q`.  eda_mha_qproj_supported(H, head_dim, Lk) != 0 says whether the launch
 * exists for a shape. */
int eda_mha_qproj_supported(int H, int head_dim, int Lk);
int eda_mha_qproj_fwd(const float *x, long x_sb, long x_sl, const float *wq, long ldwq, const float *bq,
                      const float *k, const float *v, long k_sb, long k_sl, long v_sb, long v_sl,
                      const unsigned char *key_padding_mask, int B, int H, int Lq, int Lk, int head_dim, float scale,
                      float p_drop, const unsigned long long *seed_ptr, unsigned salt, float *q_out, long q_sb, long q_sl,
                      float *out, float *lse, int dtype, void *stream);

/* ---- set-abstraction grouped MLP, channels-last pipeline -----------------
 * Rows are positions (scene, centre j, neighbour k) of a (b*m*ns, C) matrix.
 *
 * eda_group_concat_cl_f32 replaces QueryAndGroup's grouping + centring + concat
 * (pointnet2/pointnet2_utils.py:347-360: 2x group_points, `-= centre`, `/= radius`,
 * torch.cat): out[b, j*ns+k, :] = [ (xyz[b,idx]-new_xyz[b,j]) * (1/radius) | feats_cl[b,idx,:] ]
 * with feats_cl (b,n,c) channels-last (c may be 0).  *_grad scatter-adds the feature
 * columns of d(out) back into dfeats_cl (b,n,c) (zeroed here).                 */
int eda_group_concat_cl_f32(const float *xyz, const float *new_xyz, const float *feats_cl,
                            const int *idx, int b, int n, int m, int ns, int c, float radius,
                            int normalize_xyz, float *out, void *stream);
int eda_group_concat_cl_grad_f32(const float *dx, const int *idx, int b, int n, int m, int ns,
                                 int c, float *dfeats_cl, void *stream);
/* eda_bn_relu_fwd_f32 replaces BatchNorm2d + ReLU (+ F.max_pool2d over nsample) of
 * SharedMLP (pointnet2/pytorch_utils.py:67-120, pointnet2_modules.py:251-257) on a
 * (R,C) pre-activation matrix z: batch statistics (training != 0; running stats are
 * updated with `momentum`, unbiased variance) or running statistics (eval);
 * out = relu(bn(z)) (R,C), or with pool > 1 its max over each `pool` consecutive rows
 * (R/pool,C) plus the arg-max row (bytes).  mean/rstd/scale/shift (C each) are kept
 * for the backward.
 * eda_bn_relu_bwd_f32: dz (R,C), dgamma, dbeta (C floats each).
 * ws of the forward: 2*C + 1 doubles that must be ZERO on entry and are left zero on return (the last
 * workgroup of the statistics kernel finalises and clears them): a caller zeroes the buffer once and
 * re-uses it for every call ordered on the same stream -- no zero-fill or finalize launch per call.
 * ws of the backward: 2*C doubles of plain scratch (zeroed inside).
 * p_drop > 0: element dropout AFTER the ReLU fused into the same pass (the heads'
 * Conv-BN-ReLU-Dropout, models/modules.py:66-86), mask = the counter hash of eda_mha_* /
 * eda_add_dropout_ln_* on (seed_ptr, salt, element index); built for pool == 1 and
 * R <= eda_bn_relu_dropout_max_rows() (else EDA_ERR_UNSUPPORTED; pass 0 and drop separately). */
long eda_bn_relu_dropout_max_rows(void);
int eda_bn_relu_fwd_f32(const float *z, long R, int C, const float *gamma, const float *beta,
                        float eps, float momentum, int training, float *running_mean,
                        float *running_var, int pool, double *ws, float *mean, float *rstd,
                        float *scale, float *shift, float *out, unsigned char *argmax,
                        float p_drop, const unsigned long long *seed_ptr, unsigned salt, void *stream);
int eda_bn_relu_bwd_f32(const float *dout, const unsigned char *argmax, const float *z, long R,
                        int C, int pool, const float *gamma, const float *mean, const float *rstd,
                        const float *scale, const float *shift, int training, double *ws,
                        float *dgamma, float *dbeta, float *dz, float p_drop,
                        const unsigned long long *seed_ptr, unsigned salt, void *stream);

/* ---- fused residual + dropout + LayerNorm ----------------------------------
 * out = LayerNorm(x + dropout(y + y_bias)) over the last dimension of (R,C) rows: the
 * post-norm blocks of models/encoder_decoder_layers.py:94-96,106-122,154-156,184-186,371-405
 * (`x = norm(x + dropout(y))`, y = output of an attention out-projection or FFN linear whose
 * bias y_bias (C floats, may be NULL) is added here so that its gradient comes out of the
 * backward for free).  mean/rstd (R floats each) are kept for the backward, which returns
 * dx, dy (R,C) and grads3 = [d(gamma) | d(beta) | d(y_bias)] (3*C floats) using `ws` scratch
 * for per-block partial sums.  Dropout mask: the same counter-based hash as eda_mha_*.
 * pos / out_pos (both (R,C), or both NULL): also write out_pos = out + pos, the query of the NEXT
 * attention block (`with_pos_embed(out, query_pos)`, encoder_decoder_layers.py:366-401), so that
 * the separate add launch and -- through dout2, the gradient that arrives for out_pos -- the
 * autograd accumulation of the two gradients of `out` disappear.                              */
int eda_add_dropout_ln_fwd_f32(const float *x, const float *y, const float *y_bias,
                               const float *gamma, const float *beta, long R, int C, float eps,
                               float p_drop, const unsigned long long *seed_ptr, unsigned salt,
                               float *out, float *mean, float *rstd, const float *pos, float *out_pos,
                               void *stream);
size_t eda_add_dropout_ln_bwd_workspace_bytes(long R, int C);
/* (y == NULL: x already is the pre-norm sum z of eda_linear_add_dropout_ln_fwd_f32) */
int eda_add_dropout_ln_bwd_f32(const float *dout, const float *x, const float *y,
                               const float *y_bias, const float *gamma, const float *mean,
                               const float *rstd, long R, int C, float p_drop,
                               const unsigned long long *seed_ptr, unsigned salt, float *dx,
                               float *dy, float *grads3, void *ws, size_t ws_bytes,
                               const float *dout2 /* (R,C) or NULL: added to dout */, void *stream);

/* The post-norm block's linear layer and its residual + Dropout + LayerNorm as ONE launch:
 *     z = resid + Dropout(x W^T + bias),   out = LayerNorm(z) * gamma + beta   (+ out_pos = out + pos)
 * for the attention out-projections and the second FFN linear (models/encoder_decoder_layers.py:94-96, 106-122, 154-156,
 * 184-186, 371-405): x (R,K) with row stride ldx, W (N,K) with row stride ldw, resid / z / out / pos / out_pos dense (R,N),
 * mean / rstd (R).  The product x W^T is never written; z (may be NULL in inference) is what the backward needs:
 * eda_add_dropout_ln_bwd_f32 with x = z and y = NULL (dy then is the gradient of the product, dx that of resid), followed
 * by the linear layer's own eda_linear_dgrad_f32 / weight gradient.  Same Dropout hash as eda_add_dropout_ln_fwd_f32.
 * Exists for N = 288 (d_model of the path) and K % 32 == 0 (eda_linear_add_dropout_ln_supported); 16-byte aligned operands. */
int eda_linear_add_dropout_ln_supported(int K, int N);
int eda_linear_add_dropout_ln_fwd_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                      const float *bias, const float *resid, const float *gamma, const float *beta,
                                      float eps, float p_drop, const unsigned long long *seed_ptr, unsigned salt, float *z,
                                      float *out, float *mean, float *rstd, const float *pos, float *out_pos, void *stream);
/* The same launch with scratch for a SPLIT contraction: up to 2048 rows two workgroups share each 16-row block's contraction
 * and the last one to arrive adds the two partial tiles in slice order (bit-reproducible) and runs the LayerNorm epilogue --
 * 2048 rows then occupy 256 CUs instead of 128.  ws: eda_linear_add_dropout_ln_workspace_bytes(R, K, N) bytes (0: the shape is
 * not split, ws may be NULL), 16-byte aligned, zeroed once (every call leaves its ticket words zero), one per stream; it may be
 * the scratch of eda_linear_ex_ws_f32.  ws == NULL = eda_linear_add_dropout_ln_fwd_f32. */
size_t eda_linear_add_dropout_ln_workspace_bytes(long R, int K, int N);
int eda_linear_add_dropout_ln_fwd_ws_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                                      const float *bias, const float *resid, const float *gamma, const float *beta,
                                      float eps, float p_drop, const unsigned long long *seed_ptr, unsigned salt, float *z,
                                      float *out, float *mean, float *rstd, const float *pos, float *out_pos, void *ws, size_t ws_bytes,
                                         void *stream);

/* Deferred form: with grads3 == NULL eda_add_dropout_ln_bwd_f32 leaves its per-block partial sums
 * in `ws` (eda_add_dropout_ln_bwd_blocks(R) rows of 3*C floats); eda_ln_reduce_grouped_f32 then
 * reduces MANY sites in one launch: desc[site][8] = {ws pointer, rows, C, dgamma, dbeta,
 * dbias or 0, 0, 0} (64-bit words on the device), max_c = the largest C among them.          */
int eda_add_dropout_ln_bwd_blocks(long R);
int eda_ln_reduce_grouped_f32(const long long *desc, int nsites, int max_c, void *stream);

/* Grouped form: every queued weight gradient of a backward pass in ONE launch (no K split, one
 * writer per element).  Device arrays of 64-bit words, host-built (eda_amd/wgrad_queue.py):
 *   tasks[ntasks][4]  = {target index, tile row, tile column, 0}           (96x96 tiles)
 *   targets[..][8]    = {dW pointer, db pointer or 0, M, N, first job, job count, accumulate, 0}
 *   jobs[..][8]       = {dY pointer, ld_dy, X pointer, ld_x, K, 0, 0, 0}
 * dW (M,N) contiguous = (accumulate ? dW : 0) + sum over the target's jobs of dY^T X; db likewise
 * the column sums of the jobs' dY.  Same alignment rules as eda_wgrad_f32.                 */
int eda_wgrad_grouped_f32(const long long *tasks, int ntasks, const long long *targets,
                          const long long *jobs, void *stream);

/* Arithmetic of eda_wgrad_grouped_f32's contractions: 0 = v_mfma_f32_16x16x4_f32 on fp32 operands; 1 = "bf16 x 3": every
 * fp32 operand value split exactly into three bf16 terms while it is staged, six v_mfma_f32_16x16x32_bf16 products of
 * weight >= 2^-16 per contraction step, fp32 accumulators -- the dropped terms are <= 2^-23 of a product, the result
 * is an fp32-accurate dW (same test bound against fp64) at 2.7x less time on the matrix pipe (csrc/wgrad.hip); -1 = the
 * default again (EDA_WGRAD_BF16X3, on unless set to 0).  Process-wide.  Replaces nothing in the reference (its weight
 * gradients are cuBLAS calls issued by autograd for nn.Linear / nn.Conv1d, models/encoder_decoder_layers.py). */
int eda_wgrad_set_arith(int mode);

/* ---- column sums (bias gradients) ------------------------------------------
 * out[c] = sum_r x[r*ld + c] for a row-major (R,C) matrix: d(bias) of every pointwise linear
 * layer on the path (what autograd's AddmmBackward / the reference's nn.Linear, nn.Conv1d(k=1)
 * backward compute with a framework reduction, e.g. models/encoder_decoder_layers.py:47-75,
 * models/modules.py:66-86 in training).  One launch; `ws` = eda_colsum_workspace_bytes(R,C)
 * bytes of scratch; `counters` = at least (C+63)/64 unsigned ints owned by the caller, zero
 * before the first use (the kernel leaves them zero); calls sharing `counters` must be ordered
 * on one stream.  The summation order is fixed (run-to-run deterministic).  */
size_t eda_colsum_workspace_bytes(long R, int C);
int eda_colsum_f32(const float *x, long R, int C, long ld, float *out, void *ws, size_t ws_bytes,
                   unsigned *counters, void *stream);

/* ---- weight gradient of a pointwise linear layer ---------------------------
 * dW (M,N) = dY^T X for dY (K,M) and X (K,N) row-major with row strides ld_dy, ld_x, and
 * optionally db (M) = column sums of dY (db may be NULL): what autograd computes for every
 * nn.Linear / nn.Conv1d(k=1) / in-projection of the encoder-decoder in training
 * (models/encoder_decoder_layers.py:47-75, models/modules.py:66-86, models/bdetr.py:96-143).
 * fp32 MFMA with fp32 accumulation; K is split over workgroups and the partial tiles are added
 * in a fixed order (deterministic) by a second small kernel.  M, N, ld_dy, ld_x multiples
 * of 4; dW contiguous.  ws: eda_wgrad_workspace_bytes(K, M, N) bytes of scratch (16-byte
 * aligned).                                                                       */
size_t eda_wgrad_workspace_bytes(long K, int M, int N);
int eda_wgrad_f32(const float *dy, long ld_dy, const float *x, long ld_x, long K, int M, int N,
                  float *dW, float *db, void *ws, size_t ws_bytes, void *stream);

/* Weighted column sums: out[m][c] = sum_r w[r*ldw + m] * x[r*ld + c] (m < MW <= 4) and
 * wsum[m] = sum_r w[r*ldw + m] (wsum may be NULL): the weight and bias gradient of a pointwise
 * layer with 1-4 output channels (box centre / size / objectness heads, models/modules.py:66-86,
 * 150-175), dW = dY^T X with dY = w.  Workspace / counters as for eda_colsum_f32.        */
size_t eda_wcolsum_workspace_bytes(long R, int C, int MW);
int eda_wcolsum_f32(const float *x, long R, int C, long ld, const float *w, long ldw, int MW,
                    float *out, float *wsum, void *ws, size_t ws_bytes, unsigned *counters,
                    void *stream);

/* The whole backward of pointwise layers with 1-4 output channels (MW), for up to 16 layers of one shape in one pass
 * (the box-centre / size stacks' last layers of all seven prediction heads, models/modules.py:66-86, 150-175; their
 * backward passes are issued together, eda_amd/heads_batched.py): per layer g
 *   da[g] (R, C; ldda) = dy[g] (R, MW contiguous) w[g] (MW, C)     dW[g] (MW, C) = dy[g]^T a[g] (R, C; lda)
 *   db[g] (MW) = column sums of dy[g] (db[g] may be NULL)
 * One streaming launch over (64-column groups, 4 row slabs, layers) + one small launch that adds the slabs in order
 * (deterministic).  ws: eda_tiny_out_bwd_workspace_bytes(ngroups, C, MW) bytes of scratch. */
size_t eda_tiny_out_bwd_workspace_bytes(int ngroups, int C, int MW);
int eda_tiny_out_bwd_multi_f32(int ngroups, long R, int C, int MW, const float *const *dy, const float *const *w,
                               const float *const *a, const long *lda, float *const *da, const long *ldda,
                               float *const *dW, float *const *db, void *ws, size_t ws_bytes, void *stream);

/* ---- Hungarian matching on the device -----------------------
 * Replaces `scipy.optimize.linear_sum_assignment(C[b])` per scene in HungarianMatcher.forward
 * (models/losses.py:319-329): cost is (B, Q, G) with element strides sb, sq, st; the first
 * ntargets[b] target columns of scene b are real.  assign (B, G) int32: assign[b][t] = the query
 * matched to target t (distinct per scene, minimum total cost), -1 for t >= ntargets[b].
 * Q <= 1024, G <= 256.  The optimum is unique unless costs tie (then any optimum may differ from
 * scipy's).                                                                           */
int eda_lsa_f32(const float *cost, long sb, long sq, long st, int B, int Q, int G,
                const int *ntargets, int *assign, void *stream);

/* ---- row-wise L2 normalisation -------------------------------------------------------------
 * y[r,:] = x[r,:] / max(||x[r,:]||_2, eps), norm[r] = ||x[r,:]||_2 (kept for the backward):
 * torch.nn.functional.normalize(x, p=2, dim=-1) as the reference applies it to the 64-channel
 * contrastive projections of queries and text tokens (models/bdetr.py:224-226, 262-264, 316-320).
 * Backward: dx = (dy - y <dy,y>) / norm where norm > eps, dy / eps on clamped rows (what autograd
 * derives for the reference's x / norm.clamp_min(eps)).  Contiguous rows, 1 <= C <= 1024.      */
int eda_l2norm_rows_fwd_f32(const float *x, long R, int C, float eps, float *y, float *norm, void *stream);
int eda_l2norm_rows_bwd_f32(const float *dy, const float *y, const float *norm, long R, int C, float eps,
                            float *dx, void *stream);

/* ---- measurement aid ---------------------------------------------------------
 * dst[0..n) = src[0..n) with 16-byte accesses (n a multiple of 4, both pointers 16-byte aligned):
 * the in-repo device-copy kernel SURVEY.md §8(d) asks for -- bench.py times it on a buffer far
 * larger than the 256 MB Infinity Cache and reports read+write bytes / time as the ACHIEVABLE
 * HBM rate next to the 8 TB/s nominal peak.  Replaces nothing in the reference.          */
int eda_device_copy_f32(const float *src, float *dst, size_t n, void *stream);

/* out[0..count) = srcs[0] + srcs[1] + ... + srcs[n-1] (1 <= n <= 8 device pointers in a HOST array; count a multiple
 * of 4, everything 16-byte aligned; out may alias srcs[0]).  The backward of a tensor with several consumers: the
 * reference leaves the n - 1 accumulations to the autograd engine (one at::add launch per incoming gradient, e.g. the
 * residual stream / positional term of models/encoder_decoder_layers.py:366-405); eda_amd.nn_utils.fan_out sums them in
 * one launch. */
int eda_add_n_f32(const float *const *srcs, int n, size_t count, float *out, void *stream);

/* ---- row GEMMs of the pointwise layers (csrc/gemm.hip, fp32 MFMA) ---------------------------
 * eda_linear_fwd_f32 replaces the cuBLAS/cuDNN call behind every nn.Linear / Conv1d(k=1) /
 * Conv2d(1x1) of the path (pointnet2/pytorch_utils.py:88-120; models/encoder_decoder_layers.py:
 * 47-75,324-330; models/modules.py:19-178): y (R,N) = x (R,K) w(N,K)^T + bias, optional activation
 * (relu = 1: ReLU; relu = 2: GELU in its erf form, the `intermediate` activation of the frozen RoBERTa encoder,
 * models/bdetr.py:77-80).
 * Row strides ldx/ldw/ldy in floats; 16-byte loads/stores are used when K, N, the strides and
 * the pointers allow, element accesses otherwise (any K, N >= 1).                             */
int eda_linear_fwd_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N,
                       const float *bias, int relu, float *y, long ldy, void *stream);
/* input gradient of the same layer: dx (R,K) = dy (R,N) w(N,K)                                 */
int eda_linear_dgrad_f32(const float *dy, long lddy, long R, int N, const float *w, long ldw, int K,
                         float *dx, long lddx, void *stream);

/* ---- fused set-abstraction / feature-propagation MLP (csrc/sa_cl.hip + gemm.hip + wgrad.hip) ----
 * eda_sa_fused_fwd_f32 replaces, in one call, QueryAndGroup.forward (pointnet2/pointnet2_utils.py:
 * 317-376: group_points x2, centre subtraction, /radius, cat), SharedMLP.forward = nlayers x
 * [Conv2d 1x1 without bias -> BatchNorm2d -> ReLU] (pointnet2/pytorch_utils.py:11-36,67-120) and
 * F.max_pool2d over nsample (pointnet2/pointnet2_modules.py:251-257); with plain input rows it is
 * the SharedMLP of PointnetFPModule (pointnet2_modules.py:407-414).
 *   input : idx != NULL -> rows (scene, centre, neighbour) gathered from xyz (b,n,3), new_xyz (b,m,3),
 *           feats_cl (b,n,c_feat) channels-last, idx (b,m,ns) (ball_query output); channels[0] = 3+c_feat
 *           idx == NULL -> x (R, channels[0]) rows with row stride ldx
 *   layers: weight[l] (channels[l+1], channels[l]) row-major, gamma/beta/running_* (channels[l+1]);
 *           host arrays of device pointers; channels[1..] multiples of 4
 *   output: out (R/pool, channels[nlayers]) and, pool > 1, argmax (same shape, uint8: row of the maximum)
 *   kept for the backward: z[l] (R, channels[l+1]) pre-activations, stats[l] = mean|rstd|scale|shift
 *   ws: 2*1024+2 doubles, ZERO on entry, left zero (shared with eda_bn_relu_fwd_f32).
 * The grouped tensor and the activated tensors are never written to HBM (see csrc/gemm.hip).   */
int eda_sa_fused_fwd_f32(const float *x, long ldx, const float *xyz, const float *new_xyz,
                         const float *feats_cl, const int *idx, int b, int n, int m, int ns, int c_feat,
                         float radius, int normalize_xyz, long R, int nlayers, const int *channels,
                         const float *const *weight, const float *const *gamma, const float *const *beta,
                         float *const *running_mean, float *const *running_var, float eps, float momentum,
                         int training, int pool, float *const *z, float *const *stats, double *ws,
                         float *out, unsigned char *argmax, void *stream);
/* backward of the above: dW[l], dgamma[l], dbeta[l]; dx (R, channels[0]) for plain rows or
 * dfeats_cl (b,n,c_feat; zeroed here, scatter-added) for gathered rows (either may be NULL).
 * scratch_a / scratch_b: R * max(channels[1..]) floats each; ws: eda_sa_fused_bwd_workspace_bytes. */
size_t eda_sa_fused_bwd_workspace_bytes(long R, int nlayers, const int *channels, int gather);
int eda_sa_fused_bwd_f32(const float *dout, const unsigned char *argmax, const float *x, long ldx,
                         const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx,
                         int b, int n, int m, int ns, int c_feat, float radius, int normalize_xyz, long R,
                         int nlayers, const int *channels, const float *const *weight,
                         const float *const *gamma, const float *const *z, const float *const *stats,
                         int training, int pool, float *scratch_a, float *scratch_b, void *ws,
                         size_t ws_bytes, float *const *dW, float *const *dgamma, float *const *dbeta,
                         float *dx, long lddx, float *dfeats_cl, void *stream);

/* eda_sa_fused_bwd_f32 with one more argument: weight_t[l] = W_l^T ((channels[l], channels[l+1]) row-major, 16-byte
 * aligned) where the caller keeps the transposes at hand (eda_transpose_batch_f32), NULL entries where not: the call then
 * skips its own per-layer transpose launch. */
int eda_sa_fused_bwd_wt_f32(const float *dout, const unsigned char *argmax, const float *x, long ldx,
                            const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx,
                            int b, int n, int m, int ns, int c_feat, float radius, int normalize_xyz, long R,
                            int nlayers, const int *channels, const float *const *weight,
                            const float *const *weight_t,
                            const float *const *gamma, const float *const *z, const float *const *stats,
                            int training, int pool, float *scratch_a, float *scratch_b, void *ws,
                            size_t ws_bytes, float *const *dW, float *const *dgamma, float *const *dbeta,
                            float *dx, long lddx, float *dfeats_cl, void *stream);

/* ---- sibling layers in one launch ---------------------------------------------------------------
 * The three ThreeLayerMLPs of a ClsAgnosticPredictHead (models/modules.py:111-178: centre, size,
 * semantic scores) are independent stacks of the same shape; the reference runs them one layer at
 * a time (15 small cuDNN/cuBLAS launches per head, 7 heads).  Here layer l of all siblings is ONE
 * launch: a grouped GEMM (up to 24 problems with the same row count, own operands and widths; host
 * arrays of device pointers / strides / dims) and a grouped BatchNorm+ReLU+Dropout over column
 * blocks of one (R, ngroups*cpg) matrix (single-launch kernels: R <= eda_bn_relu_dropout_max_rows()).
 * The BACKWARD passes of the seven heads depend on the loss alone (what a head hands to the next decoder layer is
 * detached, models/bdetr.py:262-316), so they are issued together (eda_amd/heads_batched.py): grouped input gradients
 * over 3 x 7 = 21 problems and eda_bn_relu_grouped_bwd_multi_f32 = the grouped BatchNorm backward for up to 8
 * matrices of one shape in one launch (dout / z / dz / stats = (4, C) rows mean | rstd | scale | shift / dgb = (2, C)
 * rows dgamma | dbeta per matrix; gamma, salts: nmat * ngroups entries, matrix-major). */
int eda_linear_grouped_fwd_f32(int ngroups, const float *const *x, const long *ldx, long R, const int *K,
                               const float *const *w, const long *ldw, const int *N,
                               const float *const *bias, int relu, float *const *y, const long *ldy,
                               void *stream);
int eda_linear_grouped_dgrad_f32(int ngroups, const float *const *dy, const long *lddy, long R, const int *N,
                                 const float *const *w, const long *ldw, const int *K, float *const *dx,
                                 const long *lddx, void *stream);
int eda_bn_relu_grouped_fwd_f32(const float *z, long R, int ngroups, int cpg, const float *const *gamma,
                                const float *const *beta, float *const *running_mean,
                                float *const *running_var, float eps, float momentum, int training,
                                float *mean, float *rstd, float *scale, float *shift, float *out,
                                float p_drop, const unsigned long long *seed_ptr, const unsigned *salts,
                                void *stream);
int eda_bn_relu_grouped_bwd_f32(const float *dout, const float *z, long R, int ngroups, int cpg,
                                const float *const *gamma, const float *mean, const float *rstd,
                                const float *scale, const float *shift, int training, float *dgamma,
                                float *dbeta, float *dz, float p_drop, const unsigned long long *seed_ptr,
                                const unsigned *salts, void *stream);
int eda_bn_relu_grouped_bwd_multi_f32(int nmat, const float *const *dout, const float *const *z, long R, int ngroups,
                                      int cpg, const float *const *gamma, const float *const *stats, int training,
                                      float *const *dgb, float *const *dz, float p_drop,
                                      const unsigned long long *seed_ptr, const unsigned *salts, void *stream);

/* eda_linear_fwd_f32 with two more things in its epilogue, for chains Linear -> ReLU -> Dropout -> Linear (the FFNs of
 * models/encoder_decoder_layers.py:88-96,120-122,403-405 and the contrastive projection MLPs of models/bdetr.py:127-139):
 *   drop_p > 0: nn.Dropout(drop_p) in training mode after the bias / ReLU; element (row, col) is kept iff
 *               hash(seed, drop_salt, row * N + col) >= drop_p * 2^32 (*drop_seed is read on the device);
 *   gate != NULL (R x N, row stride ldgate): y = gate > 0 ? y * gate_scale : 0, i.e. the backward of ReLU (+ Dropout,
 *               gate_scale = 1 / (1 - p)) of the layer whose ACTIVATED output `gate` is, applied to the gradient
 *               that flows into it (x = dY of the next layer, w = that layer's W^T). */
int eda_linear_ex_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N, const float *bias,
                      int relu, float drop_p, const unsigned long long *drop_seed, unsigned drop_salt,
                      const float *gate, long ldgate, float gate_scale, float *y, long ldy, void *stream);

/* Transposed copies of `count` row-major fp32 matrices in ONE launch.  desc: device array of count x 5 int64
 * {source pointer, destination pointer, rows, cols, index of the matrix's first 32x32 tile}; total_tiles = sum over
 * the matrices of ceil(rows/32)*ceil(cols/32).  The destination of matrix i is (cols, rows) row-major.  Used for the
 * W^T shadow of the linear weights that turns autograd's input gradient dX = dY W (torch/nn/functional.linear's
 * backward, every nn.Linear, 1x1 convolution and attention projection under models/) into the same row-GEMM form as
 * the forward (eda_linear_fwd_f32 on W^T). */
int eda_transpose_batch_f32(const long long *desc, int count, long long total_tiles, void *stream);

/* CUs the multi-workgroup furthest point sampler must leave to other resident spin-kernels (its workgroups poll each
 * other and have to be co-resident; at N > 1 RCCL's channel workgroups are such kernels): the sampler then plans with
 * (CUs - reserve) / workgroups-per-scene scenes per launch.  Default 0 (or EDA_FPS_CU_RESERVE).  Replaces nothing in
 * the reference (its sampler is one workgroup per scene, sampling_gpu.cu:74-178). */
int eda_fps_set_cu_reserve(int cus);

/* 1: the multi-workgroup sampler runs as a PREFETCH on a second stream underneath other work (eda_amd/pipeline.py): its
 * workgroups poll one granule per hand-off record instead of five, so that ~100 resident polling workgroups do not take
 * memory bandwidth from the step's kernels (-0.2 ms per step; the sampler itself 3.07 -> 3.29 ms).  0 (default, or
 * EDA_FPS_BACKGROUND): the sampler is on the critical path and polls wide.  Same indices either way (a launch parameter:
 * captured graph nodes keep the value they were captured with).  Replaces nothing in the reference (its sampler runs
 * inside the step, pointnet2_modules.py:128). */
int eda_fps_set_background(int on);

/* Which sampler eda_furthest_point_sampling_f32 runs for scenes of 8193..65536 points (csrc/fps.hip, csrc/fps_bucket.hip;
 * identical indices either way; replaces nothing in the reference, whose sampler is one workgroup per scene,
 * sampling_gpu.cu:74-178):
 *   EDA_FPS_AUTO    (default) the cluster kernels (workgroups of a scene hand candidates to each other: fast, but all of
 *                   a launch's workgroups must be co-resident) and, behind them, the single-workgroup bucket sampler
 *                   gated on the call's give-up flag: a launch that was not co-resident is REPAIRED on the device;
 *                   int 2 of the workspace's status block counts such repairs, int 0 stays 0
 *   EDA_FPS_CLUSTER the cluster kernels only; a give-up sets the sticky int 0 of the status block
 *   EDA_FPS_BUCKET  the bucket sampler only: one workgroup per scene, no inter-workgroup communication (use it next to
 *                   other resident spin-kernels, e.g. RCCL's channel kernels at N > 1)
 * Process-wide; the environment variable EDA_FPS_BUCKET=0|1 (read at every call) overrides it with CLUSTER | BUCKET. */
#define EDA_FPS_AUTO    0
#define EDA_FPS_CLUSTER 1
#define EDA_FPS_BUCKET  2
int eda_fps_set_policy(int policy);

/* The plain row products with scratch for a SPLIT CONTRACTION: launches of few 32 x 96 output tiles against a long
 * contraction (the frozen text encoder's 640 x 3072 -> 768, models/bdetr.py:170-175; the 3456-deep input gradients of the
 * hoisted K | V projections, models/encoder_decoder_layers.py:366-401) divide each tile's contraction over several
 * workgroups; the fp32 partial tiles meet in `ws`, the last workgroup of a tile to arrive adds them in slice order
 * (bit-reproducible) and applies the epilogue.  eda_linear_splitk_workspace_bytes(R, contraction, columns) = 0: that shape
 * is not split, ws may be NULL (= eda_linear_ex_f32 / eda_linear_dgrad_f32).  The first 4096 bytes of ws are ticket
 * words: ZERO before the first call, left zero by every call; one workspace may serve every call of one stream. */
size_t eda_linear_splitk_workspace_bytes(long R, int K, int N);
int eda_linear_ex_ws_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N, const float *bias, int relu,
                         float drop_p, const unsigned long long *drop_seed, unsigned drop_salt, const float *gate, long ldgate,
                         float gate_scale, float *y, long ldy, void *ws, size_t ws_bytes, void *stream);
int eda_linear_dgrad_ws_f32(const float *dy, long lddy, long R, int N, const float *w, long ldw, int K, float *dx, long lddx,
                            void *ws, size_t ws_bytes, void *stream);

/* The two products with a second TERM added in the epilogue: y = x W^T (+ bias) + addend, dx = dy W + addend.  The
 * reference's post-norm blocks feed one tensor to an attention / feed-forward branch AND to the residual sum
 * (models/encoder_decoder_layers.py:87-105 cross-attention, :231-245 encoder layer, :340-401 decoder layer), so its
 * backward adds two gradients per block in a separate element-wise kernel (autograd's accumulation); here the residual
 * gradient rides in the epilogue of the branch's input-gradient product.  addend: (R, columns) rows, 16-byte aligned
 * with a stride that is a multiple of 4 for the fast kernels; addend == y / dx allowed.  Same operation order as
 * product + add: bitwise equal to the two launches.  ws as for eda_linear_ex_ws_f32 (NULL: no split contraction). */
int eda_linear_addend_ws_f32(const float *x, long ldx, long R, int K, const float *w, long ldw, int N, const float *bias,
                             const float *addend, long ldadd, float *y, long ldy, void *ws, size_t ws_bytes, void *stream);
int eda_linear_dgrad_addend_ws_f32(const float *dy, long lddy, long R, int N, const float *w, long ldw, int K,
                                   const float *addend, long ldadd, float *dx, long lddx, void *ws, size_t ws_bytes,
                                   void *stream);

/* Kernel selection of the plain row products (eda_linear_fwd_f32 / _ex_f32 / _dgrad_f32): -1 the library's own choice
 * (the DMA-staged kernel of csrc/gemm.hip for launches of >= 400 tiles of 32 x 96, else the register-staged one), 0 the
 * DMA-staged kernel off, 1..4 one of its configurations for every eligible launch (tests, experiments).  Process-wide;
 * default from EDA_GEMM_DMA.  Both kernels form the same products in the same order: the results are bitwise equal.
 * Replaces nothing in the reference (its linear layers are cuBLAS calls, models/encoder_decoder_layers.py). */
int eda_gemm_set_dma(int mode);

/* Global-batch BatchNorm statistics for the fused set-abstraction / feature-propagation calls: the reference converts
 * every BatchNorm to SyncBatchNorm when more than one GPU trains (main_utils.py:336-338).  With a hook registered
 * (fn != NULL, world > 1) eda_sa_fused_fwd_f32 / eda_sa_fused_bwd*_f32 call fn(user, buf, n, stream) once per layer and
 * direction on a device vector of n fp64 sums that has to be summed over the ranks IN PLACE by work enqueued on
 * `stream` (no host synchronisation; return 0 on success), and use R * world as the row count.  Every rank must hold
 * the same number of rows.  fn == NULL restores per-GPU statistics. */
typedef int (*eda_bn_sync_fn)(void *user, double *buf, long n, void *stream);
int eda_set_bn_sync(eda_bn_sync_fn fn, void *user, int world);

/* ---- inference: a whole set-abstraction level in ONE launch (csrc/sa_eval.hip) ----
 * QueryAndGroup rows (pointnet2/pointnet2_utils.py:317-376) -> 3 x [conv1x1, BatchNorm on RUNNING statistics, ReLU]
 * (pointnet2/pytorch_utils.py:67-120 in eval mode) -> max over the nsample rows of a centre
 * (pointnet2/pointnet2_modules.py:251-257); only out (b*m, channels[3]) is written -- SURVEY 8d's fused-layer bytes.
 * Built for the backbone's SA1: c_feat = 3, channels = {6, 64, 64, 128} (eda_sa_fused_eval_supported() says so; other
 * stacks take eda_sa_fused_fwd_f32 with training = 0).  Layers 2 and 3 run as bf16 x 3 on the bf16 matrix pipe (fp32
 * accuracy).  weight[l]: (channels[l+1], channels[l]) row-major, xyz columns first. */
int eda_sa_fused_eval_supported(int c_feat, int nlayers, const int *channels, int ns);
int eda_sa_fused_eval_f32(const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx, int b, int n, int m,
                          int ns, int c_feat, float radius, int normalize_xyz, int nlayers, const int *channels,
                          const float *const *weight, const float *const *gamma, const float *const *beta,
                          const float *const *running_mean, const float *const *running_var, float eps, float *out,
                          void *stream);

/* ---- The training loss's small-tensor arithmetic as a handful of launches (csrc/loss.hip) -----------------------
 * The reference's SetCriterion (models/losses.py:339-647) is ~300 element-wise operations on tensors of a few thousand elements
 * and twice that in its autograd backward: 690 launches per training step.  Each entry below is one launch; the *_fwd entries also
 * form the gradient of their loss with respect to the predictions (grad0 / g_*), which the backward only scales by the upstream
 * gradient.  PB = heads x scenes rows of predictions (heads stacked into the batch dimension), B scenes of targets: row pb uses the
 * targets of scene pb % B.  num_boxes: ONE float on the device (losses.py:630-636).  All tensors dense unless strides are given.
 *   eda_match_cost_f32      HungarianMatcher.cost_matrix (:262-318) for the real target slots g < ntargets[b] (0 beyond): class
 *                           term from the token maps (pmap != NULL: soft-token) or from labels
 *   eda_match_slots_i64     tq (PB, Q) = the target slot matched to each query, -1 for the others (inverse of eda_lsa_f32's result)
 *   eda_box_loss_fwd/bwd    loss_boxes (:462-497): per-row sums of L1 (centre + 0.2 size) and 1 - GIoU over the valid matched pairs
 *   eda_pos_align_fwd       loss_pos_align (:396-460): maps = positive, modify, pronoun, relation with weights w[4]
 *   eda_sem_align_fwd       loss_sem_align (:499-608): maps = positive, modify, pronoun, other-entity, relation; logits = proj_queries
 *                           . proj_tokens^T / temperature; guard eda_sem_align_supported(Q, L) (the scene's logits live in LDS)
 *   eda_scale_by_scene_f32  out[pb][:] = grad0[pb][:] * w[pb][part] / num_boxes: the backward of pos_align (S parts) / sem_align (1) */
int eda_match_cost_f32(const float *logits, const float *pred, const float *tgt_boxes, const float *pmap, long pm_sg,
                       const long *labels, const int *ntargets, int PB, int B, int Q, int G, int C, float w_class, float w_bbox,
                       float w_giou, float *cost, void *stream);
int eda_match_slots_i64(const int *assign, const int *ntargets, int PB, int B, int Q, int G, long *tq, void *stream);
int eda_box_loss_fwd_f32(const float *pred, long p_sb, long p_sq, const float *tgt, const int *assign, const unsigned char *valid,
                         const float *num_boxes, int PB, int Bt, int Q, int G, float *loss_l1, float *loss_giou, float *g_l1,
                         float *g_giou, void *stream);
int eda_box_loss_bwd_f32(const float *g_l1, const float *g_giou, const long *tq, const float *w_l1, const float *w_giou,
                         const float *num_boxes, int B, int Q, int G, float *dpred, void *stream);
int eda_pos_align_chunk(int Q, int S);     /* query rows per chunk when a scene is spread over S workgroups */
int eda_pos_align_fwd_f32(const float *logits, const long *tq, const float *const *maps, const float *w, long map_sb, long map_sg,
                          const float *num_boxes, int PB, int B, int Q, int G, int C, int S, float eos, float *loss /* (PB, S) */,
                          float *grad0, void *stream);
int eda_scale_by_scene_f32(const float *g0, const float *w /* (PB, S) */, const float *num_boxes, int PB, long per, long per_part,
                           int S, float *out, void *stream);
size_t eda_sem_align_lds_bytes(int Q, int L);
int eda_sem_align_supported(int Q, int L);
int eda_sem_align_fwd_f32(const float *logits, const long *tq, const float *const *maps, long map_sb, long map_sg,
                          const long *attn_mask, const float *num_boxes, int PB, int B, int Q, int G, int L, float eos,
                          float *loss, float *grad0, void *stream);
/* compute_points_obj_cls_loss_hard_topk (models/losses.py:166-228): sigmoid focal loss of the K seed-objectness logits of each
 * scene; positives = the topk (<= 8) seeds of each real target's instance nearest its centre in box-normalised distance (equal
 * distances: lowest seed index -- the reference's torch.topk leaves that choice to the library).  loss (B) = per-scene shares of
 * the reference's scalar; grad0 (B, K) = d sum(loss) / d logits. */
/* A prediction head's box update, models/modules.py (center = base_xyz + center_residual) and the next decoder layer's position
 * input cat([center, size]) of models/bdetr.py:300-308, as one launch: center (rows, 3), qpos (rows, 6). */
int eda_center_query_pos_f32(const float *base, const float *res, const float *size, long rows, float *center, float *qpos,
                             void *stream);

/* ---- The tail of a training step on flat buffers (csrc/optim.hip) -------------------------------------------------------------
 * main_utils.py:483-486 clips the global gradient norm (torch.nn.utils.clip_grad_norm_, 0.1) and :277-305 updates three
 * learning-rate groups with torch.optim.AdamW: on one flat parameter / gradient buffer (eda_amd/parallel.py FlatParams) that is
 * two launches.  eda_grad_sumsq_f32: |grad|_2 (bit-reproducible: fixed-order partials), the groups' step counters + 1, the
 * learning rates copied from a pinned host array (a scheduler changes them between replays of a captured graph).
 * eda_adamw_flat_f32: clip coefficient (not written back to grad), decoupled weight decay, moments, bias corrections: the
 * arithmetic of torch's fused AdamW (amsgrad off).  Group ranges are multiples of 4 floats; state tensors are the caller's
 * (torch.optim.AdamW's own state_dict layout: step as one device float, exp_avg, exp_avg_sq). */
size_t eda_grad_sumsq_workspace_bytes(void);
int eda_grad_sumsq_f32(const float *grad, long n, void *ws /* zero before the first call */, float *norm_out, int nseg,
                       float *const *steps, const double *beta1, const double *beta2, const float *lr_host,
                       float *lr_dev /* 12 floats: rates | bias corrections */, void *stream);
int eda_adamw_flat_f32(float *param, const float *grad, long n, int nseg, const long *lo, const long *hi, float *const *exp_avg,
                       float *const *exp_avg_sq, float *const *steps, const float *lr_dev, const double *beta1, const double *beta2,
                       const float *eps, const float *weight_decay, const float *norm, float max_norm, float pre_scale, void *stream);

/* compute_hungarian_loss's two ends (models/losses.py:650-738).  eda_compact_targets: the padded targets with the valid slots first
 * (the reference's per-scene boolean indexing, :660-690) for up to 8 tensors of 4-byte words + counts, valid mask, box count.
 * eda_loss_combine_*: per-head values, totals and loss = w_obj * objectness + inv * (w . [ce, bbox, giou, sem]) (:716-738) and its
 * gradient. */
int eda_compact_targets(const float *mask, int n, const void *const *src, const long *src_sb, const long *src_sg, void *const *dst,
                        const long *dst_sg, const long *dst_off, const int *words, int B, int G, int *ntargets, unsigned char *valid,
                        float *num_boxes, void *stream);
int eda_loss_combine_fwd_f32(const float *const *rows, const int *parts, const float *obj, const float *w, float inv, float w_obj,
                             int P, int B, float *per_head, float *totals, float *loss, void *stream);
int eda_loss_combine_bwd_f32(const float *g, const float *w, float inv, float w_obj, float *const *d_rows, const int *n, float *d_obj,
                             int B, void *stream);
size_t eda_seed_objectness_lds_bytes(int K);
int eda_seed_objectness_fwd_f32(const float *logits, const float *seed_xyz, const int *seed_inds, const long *instance_label,
                                long npoints, const float *centre, long c_sg, const float *size, long s_sg, const float *mask,
                                int B, int K, int G, int topk, float *loss, float *grad0, void *stream);

/* ---- Row products against a FROZEN weight on the bf16 matrix pipe at fp32 accuracy (csrc/gemm_frozen.hip) --------------------
 * The reference freezes its text encoder (models/bdetr.py:77-80: requires_grad = False on every RobertaModel parameter) and runs
 * its 48 linear layers once per step (:170-175).  A frozen weight is split ONCE into three bf16 planes (v = h + m + l exactly);
 * the launch splits only the activation rows and forms every product from the six plane products of weight >= 2^-24 on
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the fp32 kernels' error bound, 2.7 x their matrix-pipe rate.
 *   eda_bf16x3_planes_bytes(N, K)            bytes of the plane buffer ([3][N][K] bf16)
 *   eda_bf16x3_split_f32(w, ldw, N, K, p, s) write the planes of w (N, K) fp32 rows
 *   eda_linear_frozen_b3_supported(R, K, N)  K % 64 == 0 and N % 64 == 0
 *   eda_linear_frozen_b3_f32(...)            y = act(x W^T + bias); act 0 none / 1 ReLU / 2 GELU (erf); 16-byte aligned rows */
size_t eda_bf16x3_planes_bytes(int N, int K);
int eda_bf16x3_split_f32(const float *w, long ldw, int N, int K, void *planes, void *stream);
int eda_linear_frozen_b3_supported(long R, int K, int N);
int eda_linear_frozen_b3_f32(const float *x, long ldx, long R, int K, const void *wplanes, int N, const float *bias, int act,
                             float *y, long ldy, void *stream);

/* ---- SyncBatchNorm without collectives: the statistics exchange through peer-mapped memory (csrc/peer.h, peer.hip) ----
 * The reference converts every BatchNorm to SyncBatchNorm at N > 1 (main_utils.py:336-338).  Here every rank (one process
 * per GPU of ONE node, <= 8) owns a slab of device memory that the other processes map (hipIpc*); a kernel that has a
 * channel's local sums writes them into every rank's slab with system-scope stores, polls its own slab for the others'
 * and adds them in rank order -- inside the BatchNorm kernels themselves (single-launch heads / positional embeddings,
 * the last block of the statistics kernel) or as a one-launch exchange of a vector (the fused set-abstraction calls'
 * hook).  No host work between kernels, nothing a hipGraph capture cannot hold, every rank gets the same bits.  All ranks
 * must issue the same exchanging launches in the same order ON ONE STREAM with the same shapes (equal rows per rank:
 * drop_last).  Spins are bounded (EDA_PEER_SPIN_LOG2): a rank that never arrives costs eda_peer_timeouts() > 0 and garbage
 * statistics, not a hang -- the host MUST read eda_peer_timeouts() (per step or per epoch) and stop on a non-zero value
 * (bench.py, eda_amd/sync_bn.check()).  The slab is FINE-GRAINED device memory (hipExtMallocWithFlags): it is written and
 * polled by other GPUs while this GPU's kernels run, and coarse-grained memory is coherent between agents at kernel
 * boundaries only.
 *   eda_peer_create(handle)            allocate + zero this process's slab (once); handle: 64 bytes out (hipIpcMemHandle_t)
 *   eda_peer_alloc_kind()              0 fine-grained, 1 uncached, 2 plain device memory (one-device use only), -1 none yet
 *   eda_peer_connect(rank, world, hs)  hs: world x 64 bytes, rank order, gathered by the host (any out-of-band channel);
 *                                      a peer whose handle changed since the last connect is re-opened
 *   eda_peer_reset()                   zero the own slab (every rank, nothing in flight, host barrier behind it)
 *   eda_peer_selftest(s, inject)       one exchange of a known vector by EVERY rank; host-checked sums and timeout word:
 *                                      0 or EDA_ERR_PEER_SELFTEST (inject != 0: publish a wrong tag -- test seam)
 *   eda_set_bn_sync_native(world)      BatchNorm statistics over the world (0 = per-GPU again); replaces eda_set_bn_sync
 *   eda_peer_allreduce_f64(buf, n, s)  in-place sum of n <= 16384 doubles over the ranks, one launch on stream s
 *   eda_peer_bn_hook                   the same as an eda_bn_sync_fn
 * Over xGMI the protocol is unchanged but unmeasured in this repository (one-device box: two processes on one GPU);
 * eda_peer_selftest() is what tells a process at start-up whether its slabs really carry data between the GPUs. */
size_t eda_peer_slab_bytes(void);
int eda_peer_create(void *handle_out);
int eda_peer_alloc_kind(void);
int eda_peer_connect(int rank, int world, const void *handles);
int eda_peer_reset(void);
int eda_peer_selftest(void *stream, int inject_wrong_tag);
int eda_peer_disconnect(void);
int eda_peer_connected(void);
long eda_peer_timeouts(void);
int eda_peer_allreduce_f64(double *buf, long n, void *stream);
int eda_peer_bn_hook(void *user, double *buf, long n, void *stream);
int eda_set_bn_sync_native(int world);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* EDA_HIP_H */
