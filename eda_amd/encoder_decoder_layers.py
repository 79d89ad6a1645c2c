"""Cross-modal encoder / decoder layers (text tokens <-> points / queries / boxes).

Mirrors models/encoder_decoder_layers.py: PositionEmbeddingLearned (:15-30),
CrossAttentionLayer (:33-124), TransformerEncoderLayerNoFFN (:127-156),
PosTransformerEncoderLayerNoFFN (:159-186), BiEncoderLayer (:189-255), BiEncoder
(:258-285), BiDecoderLayer (:288-407) -- same constructor arguments, forward
signatures, sub-module names (state_dict contract) and post-norm residual order.
Internally everything stays batch-first (B, N, F).
"""
from copy import deepcopy

import torch
import torch.nn.functional as F
from torch import nn

from .attention import MultiheadAttention, ResidualLink, residual_links_enabled
from .fused_ln import (_FFNAddDropoutLN, add_dropout_layer_norm, fuses_bias, fuses_linear, linear_add_dropout_layer_norm,
                       new_salt_base)
from .nn_utils import Conv1dK1, Linear, bn_relu_rows, fan_out, linear_rows, mlp_chain, rows_ok


def _get_clones(module, n):
    return nn.ModuleList([deepcopy(module) for _ in range(n)])


def _ffn(d_model, dim_feedforward, dropout):
    return nn.Sequential(Linear(d_model, dim_feedforward), nn.ReLU(), nn.Dropout(dropout),
                         Linear(dim_feedforward, d_model), nn.Dropout(dropout))


def _ffn_residual_norm(x, ffn, norm, training, salt, pos=None):
    """norm(x + ffn(x)) where ffn = Linear, ReLU, Dropout, Linear, Dropout: the last Dropout is
    applied inside the fused residual+LayerNorm kernel.  With `pos`: (out, out + pos)."""
    if fuses_linear(x, norm, ffn[3].weight.shape[1]) and x.dtype == torch.float32:
        # the whole block as one autograd node: Linear + ReLU + Dropout | Linear + residual + Dropout + LayerNorm
        p1 = float(ffn[2].p) if training else 0.0
        p2 = float(ffn[4].p) if training else 0.0
        args = (x, ffn[0].weight, ffn[0].bias, ffn[3].weight, ffn[3].bias, norm.weight, norm.bias,
                norm.eps, p1, salt + 0x5BD1E995, p2, salt)
        return _FFNAddDropoutLN.apply(*args, pos) if pos is not None else _FFNAddDropoutLN.apply(*args)
    fused = fuses_bias(x, norm)  # second linear's bias (and its gradient) ride in the LN kernels
    # Linear + ReLU + Dropout + Linear as one autograd node, the activations in the GEMM epilogues
    y = mlp_chain(x, [(ffn[0].weight, ffn[0].bias, True, ffn[2].p, salt + 0x5BD1E995),
                      (ffn[3].weight, None if fused else ffn[3].bias, False, 0.0, 0)], training)
    return add_dropout_layer_norm(x, y, norm, ffn[4].p, training, salt, y_bias=ffn[3].bias if fused else None, pos=pos)


def _attn_residual_norm(attn, x, q, k, v, mask, norm, p_drop, training, salt, batch_first=True, pos=None, pre_kv=None):
    """norm(x + dropout(attn(q, k, v))): the out-projection bias is deferred to the fused
    residual+LayerNorm kernel whenever that kernel runs.  With `pos`: (out, out + pos).
    pre_kv = (kv, sink, slot): K | V of this module already projected (attention.StackedKV); k, v are then not read."""
    if batch_first and attn.hip_path(q) and fuses_linear(x, norm, attn.embed_dim):
        # the out-projection rides in the residual LayerNorm's launch
        # x feeds the attention branch too (self-attention: value / everything; text <- points: the query): its residual
        # gradient rides in the branch's input-gradient product instead of an accumulation launch (attention.ResidualLink)
        link = None
        if pre_kv is None and training and residual_links_enabled() and (x is q or x is k or x is v):
            link = ResidualLink()
        o, _ = attn(q, k, v, key_padding_mask=mask, batch_first=True, skip_out_proj=True, pre_kv=pre_kv,
                    residual=(link, x) if link is not None else None)
        return linear_add_dropout_layer_norm(o, attn.out_proj.weight, attn.out_proj.bias, x, norm, p_drop, training,
                                             salt, pos=pos, link=link)
    y, y_bias = attn(q, k, v, key_padding_mask=mask, batch_first=batch_first,
                     defer_out_bias=fuses_bias(x, norm))
    return add_dropout_layer_norm(x, y, norm, p_drop, training, salt, y_bias=y_bias, pos=pos)


class PositionEmbeddingLearned(nn.Module):
    """(B, N, 3 or 6) -> (B, F, N): conv1d - BN - ReLU - conv1d."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            Conv1dK1(input_channel, num_pos_feats, kernel_size=1),
            nn.BatchNorm1d(num_pos_feats),
            nn.ReLU(inplace=True),
            Conv1dK1(num_pos_feats, num_pos_feats, kernel_size=1))

    def forward(self, xyz):
        """(B, N, 3 or 6) -> (B, F, N), the reference's layout."""
        if rows_ok(xyz, self.position_embedding_head[0].out_channels):
            return self.rows(xyz).transpose(1, 2)
        return self.position_embedding_head(xyz.transpose(1, 2).contiguous())

    def rows(self, xyz):
        """(B, N, 3 or 6) -> (B, N, F): the input already is channels-last, so the two 1x1
        convolutions are row-major GEMMs around the fused BN+ReLU kernel (no transposes)."""
        head = self.position_embedding_head
        if not rows_ok(xyz, head[0].out_channels):
            return self.forward(xyz).transpose(1, 2)
        B, N, C = xyz.shape
        h = head[0].rows(xyz.reshape(B * N, C))
        h = bn_relu_rows(head[1], h)
        return head[3].rows(h).view(B, N, -1)


class CrossAttentionLayer(nn.Module):
    """text <- points, points <- text, (points <- detected boxes), FFNs."""

    def __init__(self, d_model=256, dropout=0.1, n_heads=8, dim_feedforward=256,
                 use_butd_enc_attn=False):
        super().__init__()
        self.use_butd_enc_attn = use_butd_enc_attn
        self.cross_lv = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout_lv = nn.Dropout(dropout)
        self.norm_lv = nn.LayerNorm(d_model)
        self.ffn_lv = _ffn(d_model, dim_feedforward, dropout)
        self.norm_lv2 = nn.LayerNorm(d_model)
        self.cross_vl = deepcopy(self.cross_lv)       # same initial weights, as the reference (:62)
        self.dropout_vl = nn.Dropout(dropout)
        self.norm_vl = nn.LayerNorm(d_model)
        self.ffn_vl = deepcopy(self.ffn_lv)
        self.norm_vl2 = nn.LayerNorm(d_model)
        if use_butd_enc_attn:
            self.cross_d = MultiheadAttention(d_model, n_heads, dropout=dropout)
            self.dropout_d = nn.Dropout(dropout)
            self.norm_d = nn.LayerNorm(d_model)
        self._salt = new_salt_base()

    def text_branch(self, text_feats, vis_feats, vis_key_padding_mask):
        """text attends to points (no positional term on the keys, :80-93), then its FFN: reads both inputs, writes text only."""
        tr, sb = self.training, self._salt
        text_out = _attn_residual_norm(self.cross_lv, text_feats, text_feats, vis_feats, vis_feats,
                                       vis_key_padding_mask, self.norm_lv, self.dropout_lv.p, tr, sb)
        return _ffn_residual_norm(text_out, self.ffn_lv, self.norm_lv2, tr, sb + 1)

    def vis_branch(self, vis_feats, text_feats, text_key_padding_mask, pos_feats, detected_feats=None, detected_mask=None,
                   vis_query=None, emit_pos=False):
        """points attend to the ORIGINAL text (:99-105, position added to the query only), then to the detected boxes, then
        their FFN: reads both inputs, writes the points only.  emit_pos: (vis, vis + pos_feats)."""
        tr, sb = self.training, self._salt
        vis = _attn_residual_norm(self.cross_vl, vis_feats, vis_query if vis_query is not None else vis_feats + pos_feats,
                                  text_feats, text_feats, text_key_padding_mask, self.norm_vl, self.dropout_vl.p,
                                  tr, sb + 2)
        if detected_feats is not None and self.use_butd_enc_attn:
            vis = _attn_residual_norm(self.cross_d, vis, vis, detected_feats, detected_feats,
                                      detected_mask, self.norm_d, self.dropout_d.p, tr, sb + 3)
        if emit_pos:
            return _ffn_residual_norm(vis, self.ffn_vl, self.norm_vl2, tr, sb + 4, pos=pos_feats)
        return _ffn_residual_norm(vis, self.ffn_vl, self.norm_vl2, tr, sb + 4), None

    def forward(self, vis_feats, vis_key_padding_mask, text_feats, text_key_padding_mask,
                pos_feats, detected_feats=None, detected_mask=None, vis_query=None, emit_pos=False):
        """vis_query: vis_feats + pos_feats when the caller already has it (the producing LayerNorm launch emits it);
        emit_pos: return (vis, text, vis + pos_feats) -- the next encoder layer's self-attention query / key."""
        text_out = self.text_branch(text_feats, vis_feats, vis_key_padding_mask)
        vis, vis_q = self.vis_branch(vis_feats, text_feats, text_key_padding_mask, pos_feats, detected_feats, detected_mask,
                                     vis_query, emit_pos)
        return (vis, text_out, vis_q) if emit_pos else (vis, text_out)


class TransformerEncoderLayerNoFFN(nn.Module):
    """Self-attention + residual + LayerNorm.  (S,B,F) like the reference unless batch_first."""

    def __init__(self, d_model, nhead, dropout):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self._salt = new_salt_base()

    def forward(self, src, src_mask=None, src_key_padding_mask=None, batch_first=False):
        assert src_mask is None, "EDA always passes attn_mask=None"
        return _attn_residual_norm(self.self_attn, src, src, src, src, src_key_padding_mask,
                                   self.norm1, self.dropout1.p, self.training, self._salt, batch_first)


class PosTransformerEncoderLayerNoFFN(TransformerEncoderLayerNoFFN):
    """Same, with the positional embedding added to query and key (not value)."""

    def forward(self, src, pos, src_mask=None, src_key_padding_mask=None, batch_first=False, qk=None, emit_pos=False):
        """qk: src + pos when the caller already has it; emit_pos (batch_first only): return (out, out + pos)."""
        if qk is None:
            qk = src + pos
        assert src_mask is None, "EDA always passes attn_mask=None"
        return _attn_residual_norm(self.self_attn, src, qk, qk, src, src_key_padding_mask,
                                   self.norm1, self.dropout1.p, self.training, self._salt, batch_first,
                                   pos=pos if emit_pos else None)


class BiEncoderLayer(nn.Module):
    def __init__(self, d_model=256, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256,
                 self_attend_lang=True, self_attend_vis=True, use_butd_enc_attn=False):
        super().__init__()
        self.self_attention_lang = (TransformerEncoderLayerNoFFN(d_model, n_heads, dropout)
                                    if self_attend_lang else None)
        self.self_attention_visual = (PosTransformerEncoderLayerNoFFN(d_model, n_heads, dropout)
                                      if self_attend_vis else None)
        self.cross_layer = CrossAttentionLayer(d_model, dropout, n_heads, dim_feedforward,
                                               use_butd_enc_attn)

    def forward(self, vis_feats, pos_feats, padding_mask, text_feats, text_padding_mask,
                end_points={}, detected_feats=None, detected_mask=None, vis_query=None, emit_pos=False):
        """vis_query / emit_pos: `vis_feats + pos_feats` handed from layer to layer -- every residual LayerNorm whose
        output is next used with the positional term added emits that sum in the same launch (BiEncoder.forward)."""
        vis_q = None
        if self.self_attention_visual is not None:
            vis_feats, vis_q = self.self_attention_visual(vis_feats, pos_feats, src_key_padding_mask=padding_mask,
                                                          batch_first=True, qk=vis_query, emit_pos=True)
        elif vis_query is not None:
            vis_q = vis_query
        if self.self_attention_lang is not None:
            text_feats = self.self_attention_lang(text_feats, src_key_padding_mask=text_padding_mask,
                                                  batch_first=True)
        return self.cross_layer(vis_feats=vis_feats, vis_key_padding_mask=padding_mask,
                                text_feats=text_feats, text_key_padding_mask=text_padding_mask,
                                pos_feats=pos_feats, detected_feats=detected_feats,
                                detected_mask=detected_mask, vis_query=vis_q, emit_pos=emit_pos)


class BiEncoder(nn.Module):
    def __init__(self, bi_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(bi_layer, num_layers)
        self.num_layers = num_layers
        # clones share initial weights (as in the reference) but must not share dropout streams
        for layer in self.layers:
            for m in layer.modules():
                if hasattr(m, "_salt") and not isinstance(m, MultiheadAttention):
                    m._salt = new_salt_base()

    def forward(self, vis_feats, pos_feats, padding_mask, text_feats, text_padding_mask,
                end_points={}, detected_feats=None, detected_mask=None):
        vis_q = None
        for i, layer in enumerate(self.layers):
            last = i == len(self.layers) - 1
            out = layer(vis_feats, pos_feats, padding_mask, text_feats, text_padding_mask, end_points,
                        detected_feats=detected_feats, detected_mask=detected_mask, vis_query=vis_q, emit_pos=not last)
            vis_feats, text_feats = out[0], out[1]
            vis_q = out[2] if not last else None
        return vis_feats, text_feats


    # (Round 6 measured the TEXT chain of every layer on a second HIP stream -- inside a layer the modalities only read each
    # other, models/encoder_decoder_layers.py:231-245, so self_lang | self_vis and [cross_lv, ffn_lv] | [cross_vl, cross_d,
    # ffn_vl] can overlap, ~130 us of 640-row launches per layer and direction.  Bit-identical results, but the fork / join
    # edges inside the replayed hipGraph cost far more than the overlap returns on this ROCm: 17.6 -> 21.5 ms per step
    # (profiles/r06_encoder_two_streams.md); not kept.  text_branch / vis_branch of the cross layer remain as the seam.)


class BiDecoderLayer(nn.Module):
    """queries: self -> text -> (boxes) -> points -> FFN, all post-norm."""

    def __init__(self, d_model, n_heads, dim_feedforward=2048, dropout=0.1, activation="relu",
                 self_position_embedding="loc_learned", butd=False):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.cross_l = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout_l = nn.Dropout(dropout)
        self.norm_l = nn.LayerNorm(d_model)
        if butd:
            self.cross_d = deepcopy(self.cross_l)
            self.dropout_d = nn.Dropout(dropout)
            self.norm_d = nn.LayerNorm(d_model)
        self.cross_v = deepcopy(self.cross_l)
        self.dropout_v = nn.Dropout(dropout)
        self.norm_v = nn.LayerNorm(d_model)
        self.ffn = _ffn(d_model, dim_feedforward, dropout)
        self.norm2 = nn.LayerNorm(d_model)
        if self_position_embedding == "xyz_learned":
            self.self_posembed = PositionEmbeddingLearned(3, d_model)
        elif self_position_embedding == "loc_learned":
            self.self_posembed = PositionEmbeddingLearned(6, d_model)
        else:
            self.self_posembed = None
        self._salt = new_salt_base()

    def forward(self, query, vis_feats, lang_feats, query_pos, padding_mask, text_key_padding_mask,
                detected_feats=None, detected_mask=None, pre_kv=None, pos_batch=None):
        """pre_kv: {"l": .., "d": .., "v": ..} -> (kv, sink, slot) per cross-attention whose K | V projection the caller
        hoisted out of the layer loop (BeaUTyDETR._hoisted_kv); only honoured on the fused GPU path."""
        pre_kv = pre_kv or {}
        if self.self_posembed is not None:
            # pos_batch: the embedding's backward is issued with the other layers' at the end of the backward pass
            # (eda_amd/posembed_batched.py)
            pos = pos_batch.add(self.self_posembed, query_pos) if pos_batch is not None else self.self_posembed.rows(query_pos)
        else:
            pos = torch.full_like(query, 0.0)
        # the incoming query feeds the sum below, the residual and the value of the self-attention; the positional term
        # feeds the sum and the three LayerNorm launches that emit out + pos: one alias per consumer, so that the
        # backward adds their gradients in ONE launch each (nn_utils.fan_out) instead of 2 + 3 engine accumulations
        q_add, q_res, q_val = fan_out(query, 3)
        n_pos = 4 if detected_feats is not None else 3
        pa = fan_out(pos, n_pos)
        qp = q_add + pa[0]
        tr, sb = self.training, self._salt
        # every residual LayerNorm below also emits out + pos, the query of the block that follows it
        query, qp = _attn_residual_norm(self.self_attn, q_res, qp, qp, q_val, padding_mask, self.norm1,
                                        self.dropout1.p, tr, sb, pos=pa[1])
        if detected_feats is not None:
            query, qp = _attn_residual_norm(self.cross_l, query, qp, lang_feats, lang_feats,
                                            text_key_padding_mask, self.norm_l, self.dropout_l.p, tr, sb + 1,
                                            pos=pa[2], pre_kv=pre_kv.get("l"))
            query, qp = _attn_residual_norm(self.cross_d, query, qp, detected_feats, detected_feats,
                                            detected_mask, self.norm_d, self.dropout_d.p, tr, sb + 2, pos=pa[3],
                                            pre_kv=pre_kv.get("d"))
        else:
            query, qp = _attn_residual_norm(self.cross_l, query, qp, lang_feats, lang_feats,
                                            text_key_padding_mask, self.norm_l, self.dropout_l.p, tr, sb + 1,
                                            pos=pa[2], pre_kv=pre_kv.get("l"))
        query = _attn_residual_norm(self.cross_v, query, qp, vis_feats, vis_feats, None,
                                    self.norm_v, self.dropout_v.p, tr, sb + 3, pre_kv=pre_kv.get("v"))
        query = _ffn_residual_norm(query, self.ffn, self.norm2, tr, sb + 4)
        return query.contiguous()
