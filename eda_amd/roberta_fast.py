"""The frozen RoBERTa text encoder on this repo's own kernels.

The reference builds `RobertaModel.from_pretrained(...)`, freezes it (`requires_grad = False`, models/bdetr.py:77-80)
and reads `last_hidden_state` once per step (:170-175).  The module and its state_dict stay Hugging Face's (checkpoints
carry its parameter names); what changes on the GPU is how the forward is computed -- inference only, no autograd:

    embeddings   word[ids] + position[cumsum(ids != pad)] + token_type[0]  -> fused residual LayerNorm (csrc/ln.hip)
    per layer    ONE packed q|k|v projection (own fp32-MFMA row GEMM, csrc/gemm.hip; the three weights are concatenated
                 once, the encoder is frozen), head_dim-64 attention (csrc/mha_hd64.hip), output projection without its bias
                 -> LayerNorm(x + y + bias); intermediate projection with GELU in the GEMM epilogue; output projection
                 -> LayerNorm(x + y + bias)

i.e. 7 launches per layer, none of them a library GEMM, an AOTriton attention or a TunableOp-selected kernel (what the
stock module ran: 85 hipBLASLt + 12 Triton `attn_fwd` launches per step).  Numerics: fp32 throughout (exact-f32 MFMA);
tests/test_roberta_fast_gpu.py compares with the Hugging Face forward of the same weights.

Train mode.  The reference trains with `model.train()` on the WHOLE model (main_utils.py:459): its frozen encoder therefore runs
with its dropout layers active (embeddings, attention probabilities, the two hidden dropouts of every layer: 0.1 each in
roberta-base).  When the Hugging Face module is in train mode this forward does the same inside the same launches (the residual
LayerNorm kernel's dropout, eda_mha_fwd_hd64_drop_f32, one eda_dropout_f32 after the embedding LayerNorm) with the library's
counter-based masks -- a counter of its own, bumped once per call, so that the encoder may run on a second stream underneath the
step and stay reproducible.  (Masks cannot equal torch's Philox stream bit for bit; the statistics are tested.)
"""
import torch

from . import _lib, gemm
from .ext import _timed


def _ln_residual(x2, y2, y_bias, gamma, beta, eps, out=None, p=0.0, seed=None, salt=0):
    """LayerNorm(x2 + dropout_p(y2 + y_bias)) over the last dim of (R, C) fp32 matrices (csrc/ln.hip)."""
    R, C = x2.shape
    if out is None:
        out = torch.empty_like(x2)
    stats = torch.empty((2, R), dtype=torch.float32, device=x2.device)
    with torch.cuda.device(x2.device), _timed("add_dropout_ln_fwd", (R, C)):
        rc = _lib.lib().eda_add_dropout_ln_fwd_f32(
            x2.data_ptr(), y2.data_ptr(), y_bias.data_ptr() if y_bias is not None else None, gamma.data_ptr(),
            beta.data_ptr(), R, C, float(eps), float(p), seed.data_ptr() if p > 0 else None, int(salt) & 0xFFFFFFFF, out.data_ptr(),
            stats[0].data_ptr(), stats[1].data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_add_dropout_ln_fwd_f32")
    return out


def attention_hd64(q, k, v, key_padding_mask, num_heads, scale, p=0.0, seed=None, salt=0):
    """dropout_p(softmax(q k^T * scale + mask)) v for head_dim 64, q (B,Lq,D), k / v (B,Lk,D) column views with unit last stride;
    key_padding_mask (B,Lk) bool, True = ignore.  Forward only."""
    B, Lq, D = q.shape
    Lk = k.shape[1]
    assert D == 64 * num_heads
    out = torch.empty((B, Lq, D), dtype=torch.float32, device=q.device)
    m8 = key_padding_mask.contiguous().view(torch.uint8) if key_padding_mask is not None else None
    with torch.cuda.device(q.device), _timed("mha_fwd_hd64", (B, num_heads, Lq, Lk)):
        rc = _lib.lib().eda_mha_fwd_hd64_drop_f32(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0),
            v.stride(1), m8.data_ptr() if m8 is not None else None, B, num_heads, Lq, Lk, float(scale), float(p),
            seed.data_ptr() if p > 0 else None, int(salt) & 0xFFFFFFFF, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "eda_mha_fwd_hd64_drop_f32")
    return out


class FrozenRobertaFast:
    """Forward of a frozen `transformers.RobertaModel` (absolute position embeddings, GELU, no cross-attention) on the
    repo's kernels.  Holds packed copies of the attention projection weights; `stale()` tells when the module's
    parameters were modified in place (load_state_dict) since packing."""

    def __init__(self, hf_model):
        cfg = hf_model.config
        if cfg.hidden_act != "gelu" or cfg.hidden_size % cfg.num_attention_heads or \
                cfg.hidden_size // cfg.num_attention_heads != 64 or getattr(cfg, "is_decoder", False):
            raise NotImplementedError("FrozenRobertaFast: RoBERTa-base-shaped encoders only (GELU, head_dim 64)")
        if getattr(cfg, "position_embedding_type", "absolute") not in (None, "absolute"):
            raise NotImplementedError("FrozenRobertaFast: absolute position embeddings only")
        self.m = hf_model
        self.cfg = cfg
        self.heads = cfg.num_attention_heads
        self.eps = float(cfg.layer_norm_eps)
        self.pad = int(hf_model.embeddings.padding_idx)
        self._versions = self._param_versions()
        self.layers = []
        self.counter = None                   # train mode: this encoder's own dropout counter (device int64), bumped per call
        from .fused_ln import new_salt_base
        self._salt = new_salt_base()
        import os
        # the encoder is FROZEN: its weights are split once into bf16 x 3 planes and every linear layer runs on the bf16
        # matrix pipe at fp32 accuracy (csrc/gemm_frozen.hip; EDA_FROZEN_B3=0: the fp32-MFMA row products of csrc/gemm.hip)
        b3 = os.environ.get("EDA_FROZEN_B3", "1") != "0"
        with torch.no_grad():
            for lyr in hf_model.encoder.layer:
                sa = lyr.attention.self
                ent = {"wqkv": torch.cat([sa.query.weight, sa.key.weight, sa.value.weight], 0).contiguous(),
                       "bqkv": torch.cat([sa.query.bias, sa.key.bias, sa.value.bias], 0).contiguous(),
                       "lyr": lyr, "planes": None}
                ws = [ent["wqkv"], lyr.attention.output.dense.weight, lyr.intermediate.dense.weight, lyr.output.dense.weight]
                if b3 and all(gemm.linear_frozen_supported(64, w.shape[1], w.shape[0]) for w in ws):
                    ent["planes"] = [gemm.frozen_planes(w) for w in ws]
                    ent["w"] = ws
                self.layers.append(ent)

    def _param_versions(self):
        return tuple(p._version for p in self.m.parameters()) + (next(self.m.parameters()).data_ptr(),)

    def stale(self):
        return self._param_versions() != self._versions

    @torch.no_grad()
    def __call__(self, input_ids, attention_mask):
        """last_hidden_state (B, L, hidden) for token ids / attention mask (1 = token) on the GPU."""
        m, d = self.m, self.cfg.hidden_size
        B, L = input_ids.shape
        if L > 256:
            raise NotImplementedError("FrozenRobertaFast: utterances of up to 256 tokens (csrc/mha_hd64.hip keeps K/V in LDS)")
        emb = m.embeddings
        train = bool(m.training)
        ph = float(self.cfg.hidden_dropout_prob) if train else 0.0
        pa = float(self.cfg.attention_probs_dropout_prob) if train else 0.0
        cnt, sb = None, self._salt
        if ph > 0 or pa > 0:
            if self.counter is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("roberta_fast: run one eager train-mode forward before capturing (dropout counter)")
                from .attention import _initial_dropout_counter
                self.counter = torch.full((1,), _initial_dropout_counter() ^ 0x5EED, dtype=torch.int64, device=input_ids.device)
            cnt = self.counter
            cnt.add_(1)                       # (capturable: every replay draws new masks)
        nonpad = input_ids.ne(self.pad)
        position_ids = torch.cumsum(nonpad.long(), dim=1) * nonpad.long() + self.pad    # create_position_ids_from_input_ids
        x = emb.word_embeddings.weight[input_ids].view(B * L, d)
        y = (emb.position_embeddings.weight[position_ids] + emb.token_type_embeddings.weight[0]).view(B * L, d)
        h = _ln_residual(x, y, None, emb.LayerNorm.weight, emb.LayerNorm.bias, self.eps)
        if ph > 0:                                                              # RobertaEmbeddings: dropout(LayerNorm(.))
            with torch.cuda.device(h.device):
                rc = _lib.lib().eda_dropout_f32(h.data_ptr(), h.numel(), ph, cnt.data_ptr(), (sb + 1) & 0xFFFFFFFF, h.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "eda_dropout_f32")
        kpm = attention_mask.eq(0)                                              # True = padding key
        scale = 64 ** -0.5
        for li, ent in enumerate(self.layers):
            lyr = ent["lyr"]
            pl = ent["planes"]
            ao = lyr.attention.output
            s0 = sb + 8 + 4 * li              # salts of this layer's three dropout sites
            if pl is not None:
                W = ent["w"]
                qkv = _frozen_or_fp32(h, pl[0], W[0], ent["bqkv"], 0).view(B, L, 3 * d)
                ctx = attention_hd64(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], kpm, self.heads, scale, pa, cnt, s0)
                y = _frozen_or_fp32(ctx.view(B * L, d), pl[1], W[1], None, 0)   # (bias added by the LayerNorm kernel)
                h = _ln_residual(h, y, ao.dense.bias, ao.LayerNorm.weight, ao.LayerNorm.bias, self.eps, p=ph, seed=cnt, salt=s0 + 1)
                t = _frozen_or_fp32(h, pl[2], W[2], lyr.intermediate.dense.bias, 2)
                y = _frozen_or_fp32(t, pl[3], W[3], None, 0)
                h = _ln_residual(h, y, lyr.output.dense.bias, lyr.output.LayerNorm.weight, lyr.output.LayerNorm.bias, self.eps,
                                 p=ph, seed=cnt, salt=s0 + 2)
                continue
            qkv = gemm.linear_fwd(h, ent["wqkv"], ent["bqkv"]).view(B, L, 3 * d)
            ctx = attention_hd64(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], kpm, self.heads, scale, pa, cnt, s0)
            y = gemm.linear_fwd(ctx.view(B * L, d), ao.dense.weight)            # (bias added by the LayerNorm kernel)
            h = _ln_residual(h, y, ao.dense.bias, ao.LayerNorm.weight, ao.LayerNorm.bias, self.eps, p=ph, seed=cnt, salt=s0 + 1)
            t = _gelu_linear(h, lyr.intermediate.dense.weight, lyr.intermediate.dense.bias)
            y = gemm.linear_fwd(t, lyr.output.dense.weight)
            h = _ln_residual(h, y, lyr.output.dense.bias, lyr.output.LayerNorm.weight, lyr.output.LayerNorm.bias, self.eps,
                             p=ph, seed=cnt, salt=s0 + 2)
        return h.view(B, L, d)


def _frozen_or_fp32(x2, planes, w, bias, act):
    """The row product on the frozen weight's bf16 x 3 planes where that launch is the faster one, else on the fp32 matrix
    instruction.  Measured (tools/bench_gemm_frozen.py, profiles/r06_gemm_frozen.txt, us per launch, planes | fp32 MFMA |
    hipBLASLt): 640 x 768 -> 2304: 25.7 | 27.1 | 27.4; 640 x 768 -> 3072: 27.8 | 36.5 | 26.7; 640 x 768 -> 768: 20.9 | 11.6 |
    10.0; 640 x 3072 -> 768: 71.0 | 35.3 | 27.2; 1040 rows (130 tokens): -> 2304 33.0 | 44.5 | 48.4, 3072 -> 768 73.7 | 85.9
    | 47.5.  A 64-column tile of a 768-wide output gives 120 workgroups: the planes win where the output is wide or the rows
    are many.  EDA_FROZEN_B3=2 takes the planes everywhere."""
    import os
    R = x2.shape[0]
    N, K = planes.shape[1], planes.shape[2]
    mode = os.environ.get("EDA_FROZEN_B3", "1")
    if mode == "2" or N >= 2304 or (R >= 1024 and K >= 2304):
        return gemm.linear_frozen(x2, planes, bias, act=act)
    return gemm.linear_fwd(x2, w, bias, relu=act)


def _gelu_linear(x2, w, bias):
    """gelu(x2 @ w.T + bias), erf form, in the GEMM's epilogue (activation code 2 of the row products)."""
    return gemm.linear_fwd(x2, w, bias, relu=2)


def encode(hf_model, input_ids, attention_mask):
    """last_hidden_state of `hf_model` through the fast path.  The packed weights live ON the module (`_eda_fast`: freed
    with it, no process-wide table keyed by id()); rebuilt when its parameters were written in place since -- but never
    inside a stream capture (the packed copies would come from the graph's private pool)."""
    fast = hf_model.__dict__.get("_eda_fast")
    if fast is None or fast.m is not hf_model or fast.stale():
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("roberta_fast: the packed weights must be built before a HIP graph is captured "
                               "(run one eager encode_text_frozen first)")
        prev = fast
        fast = FrozenRobertaFast(hf_model)
        if prev is not None and prev.m is hf_model:           # repacked after an in-place weight update: the mask stream goes on
            fast.counter, fast._salt = prev.counter, prev._salt
        object.__setattr__(hf_model, "_eda_fast", fast)       # (not a sub-module / parameter: a plain attribute)
    return fast(input_ids, attention_mask)


def get_dropout_counter(hf_model):
    """Value of the train-mode dropout counter of this encoder's fast forward (host int; synchronises), or None before its
    first train-mode call.  eda_amd/checkpoint.py stores it next to the library's other counter."""
    fast = hf_model.__dict__.get("_eda_fast")
    return None if fast is None or fast.counter is None else int(fast.counter.item())


def set_dropout_counter(hf_model, value):
    """Continue a saved mask stream (in place when the counter exists: graphs that captured it keep working)."""
    fast = hf_model.__dict__.get("_eda_fast")
    if fast is None or fast.m is not hf_model:
        fast = FrozenRobertaFast(hf_model)
        object.__setattr__(hf_model, "_eda_fast", fast)
    dev = next(hf_model.parameters()).device
    if fast.counter is None:
        fast.counter = torch.full((1,), int(value), dtype=torch.int64, device=dev)
    else:
        fast.counter.fill_(int(value))


def supported(hf_model, input_ids):
    cfg = getattr(hf_model, "config", None)
    return (input_ids.is_cuda and cfg is not None and getattr(cfg, "hidden_act", None) == "gelu"
            and hasattr(hf_model, "encoder") and hasattr(hf_model, "embeddings")
            and cfg.hidden_size == 64 * cfg.num_attention_heads and input_ids.shape[1] <= 256)
