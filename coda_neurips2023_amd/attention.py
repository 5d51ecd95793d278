"""Multi-head attention with the ``torch.nn.MultiheadAttention`` module surface.

The 3DETR encoder/decoder layers build ``nn.MultiheadAttention(d_model, nhead,
dropout)`` and call it seq-first as ``attn(query, key, value=..., attn_mask=...,
key_padding_mask=...) -> (out (L,B,E), head-averaged weights (B,L,S))``
(models/transformer.py:422,470-471,506-507,566-573).  This module keeps that
call signature and the checkpoint parameter names (``in_proj_weight (3E,E)``,
``in_proj_bias (3E)``, ``out_proj.weight``, ``out_proj.bias``) and routes the
scaled-dot-product core to the fused gfx950 kernels (``attention_core``).

The head-averaged probability tensor is only materialised when
``need_weights=True``; the reference discards it on the hot path
(transformer.py:477-479,578-580).
"""
import math

import torch
import torch.nn as nn

from . import attention_core as _core
from .linear_fn import linear as _linear


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True):
        super().__init__()
        assert embed_dim % num_heads == 0, "embed_dim must be divisible by num_heads"
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        self.batch_first = False
        self.in_proj_weight = nn.Parameter(torch.empty((3 * embed_dim, embed_dim)))
        if bias:
            self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter("in_proj_bias", None)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self._reset_parameters()

    def _reset_parameters(self):
        # same initialisation as torch.nn.MultiheadAttention
        nn.init.xavier_uniform_(self.in_proj_weight)
        if self.in_proj_bias is not None:
            nn.init.constant_(self.in_proj_bias, 0.0)
            nn.init.constant_(self.out_proj.bias, 0.0)

    def _project(self, query, key, value):
        e = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        bq = bk = bv = None
        if b is not None:
            bq, bk, bv = b[:e], b[e:2 * e], b[2 * e:]
        if key is value and query is key:
            return _linear(query, w, b).chunk(3, dim=-1)
        q = _linear(query, w[:e], bq)
        if key is value:
            k, v = _linear(key, w[e:], b[e:] if b is not None else None).chunk(2, dim=-1)
            return q, k, v
        return q, _linear(key, w[e:2 * e], bk), _linear(value, w[2 * e:], bv)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None,
                average_attn_weights=True):
        """query (L,B,E), key/value (S,B,E); attn_mask bool (L,S) or (B*h,L,S) with
        True = masked out; key_padding_mask bool (B,S)."""
        tgt_len, bsz, e = query.shape
        src_len = key.shape[0]
        h, d = self.num_heads, self.head_dim
        q, k, v = self._project(query, key, value)
        mask = _core.merge_masks(attn_mask, key_padding_mask, bsz, h, tgt_len, src_len)
        # (L, B, h, d) views: the core reads the seq-first layout in place
        out, weights = _core.attention(q.reshape(tgt_len, bsz, h, d), k.reshape(src_len, bsz, h, d),
                                       v.reshape(src_len, bsz, h, d), mask,
                                       1.0 / math.sqrt(d), self.dropout if self.training else 0.0,
                                       need_weights)
        out = _linear(out.reshape(tgt_len, bsz, e), self.out_proj.weight, self.out_proj.bias)
        if need_weights and average_attn_weights:
            weights = weights.mean(dim=1)
        return out, weights
