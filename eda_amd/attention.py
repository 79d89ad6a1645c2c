"""Multi-head attention as EDA uses it (d_model 288, 8 heads x 36, key-padding
masks, attention-probability dropout 0.1).

``MultiheadAttention`` keeps the parameter names and initialisation of
``torch.nn.MultiheadAttention`` (in_proj_weight (3d,d), in_proj_bias, out_proj.*)
-- the reference builds 39 of those (models/encoder_decoder_layers.py:47,62,69,133,
298,306,314,319) and checkpoints must load -- but works batch-first: no
(B,N,F)<->(N,B,F) transposes, one packed projection GEMM when q/k/v share their
input, and the QK^T-softmax-PV core goes to ``attention_core``.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F


def attention_core(q, k, v, key_padding_mask=None, dropout_p=0.0):
    """softmax(q k^T / sqrt(hd) + mask) v for q (B,H,Lq,hd), k/v (B,H,Lk,hd).

    key_padding_mask: (B,Lk) bool, True = ignore.  A fully masked row gives NaN,
    as in the reference.
    """
    mask = None
    if key_padding_mask is not None:
        mask = torch.zeros(key_padding_mask.shape, dtype=q.dtype, device=q.device)
        mask = mask.masked_fill(key_padding_mask, float("-inf"))[:, None, None, :]
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=dropout_p)


class _OutProj(nn.Linear):
    pass


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim, embed_dim, bias=True)
        # torch.nn.MultiheadAttention._reset_parameters
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def _split_heads(self, x):
        B, L, _ = x.shape
        return x.view(B, L, self.num_heads, self.head_dim).transpose(1, 2)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False,
                attn_mask=None, batch_first=False):
        """Returns (output, None).  Inputs are (L,B,F) unless batch_first (then (B,L,F))."""
        if attn_mask is not None:
            raise NotImplementedError("EDA always passes attn_mask=None")
        if not batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        d = self.embed_dim
        W, b = self.in_proj_weight, self.in_proj_bias
        if query is key and key is value:
            q, k, v = F.linear(query, W, b).split(d, dim=-1)
        elif key is value:
            q = F.linear(query, W[:d], b[:d])
            k, v = F.linear(key, W[d:], b[d:]).split(d, dim=-1)
        elif query is key:
            q, k = F.linear(query, W[:2 * d], b[:2 * d]).split(d, dim=-1)
            v = F.linear(value, W[2 * d:], b[2 * d:])
        else:
            q = F.linear(query, W[:d], b[:d])
            k = F.linear(key, W[d:2 * d], b[d:2 * d])
            v = F.linear(value, W[2 * d:], b[2 * d:])
        o = attention_core(self._split_heads(q), self._split_heads(k), self._split_heads(v),
                           key_padding_mask, self.dropout if self.training else 0.0)
        B, H, L, hd = o.shape
        o = self.out_proj(o.transpose(1, 2).reshape(B, L, H * hd))
        if not batch_first:
            o = o.transpose(0, 1)
        return o, None
