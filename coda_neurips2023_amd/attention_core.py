"""Scaled-dot-product attention core of the 3DETR encoder/decoder.

``attention(q, k, v, mask, scale, dropout_p, need_weights)`` takes the
seq-first projections ``q (L,B,h,d)``, ``k/v (S,B,h,d)`` and returns
``out (L,B,h,d)`` (+ ``probs (B,h,L,S)`` when asked).  Semantics are those of
``torch.nn.functional.multi_head_attention_forward`` as the reference uses it:
softmax(q*scale @ k^T + mask) -> dropout(p) -> @ v.
"""
import torch
import torch.nn.functional as F


def merge_masks(attn_mask, key_padding_mask, bsz, h, tgt_len, src_len):
    """Boolean (B,h,L,S) mask (True = do not attend) or None."""
    mask = None
    if attn_mask is not None:
        if attn_mask.dtype != torch.bool:
            raise RuntimeError("only boolean attention masks are supported (the reference builds "
                               "boolean radius masks, models/transformer.py:154-161)")
        if attn_mask.dim() == 2:
            mask = attn_mask.view(1, 1, tgt_len, src_len).expand(bsz, h, tgt_len, src_len)
        else:
            mask = attn_mask.view(bsz, h, tgt_len, src_len)
    if key_padding_mask is not None:
        kpm = key_padding_mask.view(bsz, 1, 1, src_len).expand(bsz, h, tgt_len, src_len)
        mask = kpm if mask is None else (mask | kpm)
    return mask


def attention(q, k, v, mask, scale, dropout_p, need_weights):
    # (L,B,h,d) -> (B,h,L,d)
    qh = q.permute(1, 2, 0, 3)
    kh = k.permute(1, 2, 0, 3)
    vh = v.permute(1, 2, 0, 3)
    scores = torch.matmul(qh * scale, kh.transpose(-1, -2))
    if mask is not None:
        scores = scores.masked_fill(mask, float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    if dropout_p > 0.0:
        probs = F.dropout(probs, p=dropout_p)
    out = torch.matmul(probs, vh)  # (B,h,L,d)
    return out.permute(2, 0, 1, 3), (probs if need_weights else None)
