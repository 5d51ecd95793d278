"""Whole pre-norm transformer layers as single autograd nodes.

``fused_layers`` provides the three building blocks (``add_ln``, ``mha``, ``ffn_act``) as
autograd functions; chaining them through autograd costs one Python-level node per block in
the forward and again in the backward (8 per decoder layer), and the step had become
launch-bound on the host.  The functions here run the SAME forward / backward bodies of those
blocks back to back inside ONE node per layer (``torch.autograd`` sees a single function with a
hand-written backward that follows the layer's data flow, models/transformer.py:457-494 /
558-594), which removes the per-block graph bookkeeping.  Values and gradients are those of
the chained blocks (the parity tests run both, ``CODA_LAYER_NODES=ops`` selects the chain).
"""
import torch

from .fused_layers import _AddLN, _FfnAct, _MHA
from .linear_fn import tn_gemm


class _Ctx:
    """Stand-in for an autograd context when a block's forward/backward body is called directly."""

    def __init__(self, needs=(True,) * 10):
        self.saved_tensors = ()
        self.needs_input_grad = needs

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


def _add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return a + b


def _ffn_forward(y, w1, b1, w2, p):
    e = y.shape[-1]
    y2 = y.reshape(-1, e)
    h0 = torch.mm(y2, w1.t())
    cf = _Ctx()
    h = _FfnAct.forward(cf, h0, b1, p)
    o = torch.mm(h, w2.t()).view(y.shape[:-1] + (w2.shape[0],))
    return o, (cf, y2, h)


def _ffn_backward(saved, do, w1, w2):
    cf, y2, h = saved
    do2 = do.reshape(-1, do.shape[-1]).contiguous()
    dw2 = tn_gemm(do2, h)
    dh = torch.mm(do2, w2)
    dh0, db1, _ = _FfnAct.backward(cf, dh)
    dw1 = tn_gemm(dh0, y2)
    dy = torch.mm(dh0, w1)
    return dy, dw1, db1, dw2


class _DecoderLayer(torch.autograd.Function):
    """Pending residual stream (res, x, bias, p) + memory -> (s, o): the materialised stream
    after the cross-attention block and the feed-forward product that is still pending
    (``s + dropout3(o + linear2.bias)`` is formed by the next consumer)."""

    @staticmethod
    def forward(ctx, res, x, bias_prev, memory, memory_pos, query_pos, self_mask, cross_mask, cfg,
                g1, b1n, in1, ib1, ow1, ob1, g2, b2n, in2, ib2, ow2, ob2, g3, b3n, w1, fb1, w2):
        p_prev, eps, nheads, p_attn, p1, p2, p_ffn = cfg
        c1 = _Ctx()
        if x is None:
            _, y1, y1p = _AddLN.forward(c1, res, None, None, query_pos, g1, b1n, eps, 0.0)
            s1 = res
        else:
            s1, y1, y1p = _AddLN.forward(c1, x, bias_prev, res, query_pos, g1, b1n, eps, p_prev)
        qk = y1 if query_pos is None else y1p
        c2 = _Ctx()
        a1 = _MHA.forward(c2, qk, qk, y1, in1, ib1, ow1, self_mask, nheads, p_attn)
        c3 = _Ctx()
        s2, y2, y2p = _AddLN.forward(c3, a1, ob1, s1, query_pos, g2, b2n, eps, p1)
        c4 = _Ctx()
        a2 = _MHA.forward(c4, y2 if query_pos is None else y2p, memory_pos, memory, in2, ib2, ow2, cross_mask, nheads,
                          p_attn)
        c5 = _Ctx()
        s3, y3, _ = _AddLN.forward(c5, a2, ob2, s2, None, g3, b3n, eps, p2)
        o, ffn_saved = _ffn_forward(y3, w1, fb1, w2, p_ffn)
        ctx.blocks = (c1, c2, c3, c4, c5, ffn_saved)
        ctx.flags = (x is None, query_pos is not None)
        ctx.save_for_backward(w1, w2)
        return s3, o

    @staticmethod
    def backward(ctx, ds3, do):
        c1, c2, c3, c4, c5, ffn_saved = ctx.blocks
        first, has_pos = ctx.flags
        w1, w2 = ctx.saved_tensors
        dy3, dw1, dfb1, dw2 = _ffn_backward(ffn_saved, do, w1, w2)
        da2, dob2, ds2, _, dg3, db3n, _, _ = _AddLN.backward(c5, ds3, dy3, None)
        dq2, dmem_pos, dmem, din2, dib2, dow2, _, _, _ = _MHA.backward(c4, da2)
        da1, dob1, ds1, dpos2, dg2, db2n, _, _ = _AddLN.backward(c3, ds2, None if has_pos else dq2,
                                                                 dq2 if has_pos else None)
        dqk, _, dv1, din1, dib1, dow1, _, _, _ = _MHA.backward(c2, da1)
        if has_pos:
            g = _AddLN.backward(c1, ds1, dv1, dqk)
        else:
            g = _AddLN.backward(c1, ds1, _add(dv1, dqk), None)
        dx, dbias_prev, dres, dpos1, dg1, db1n = g[0], g[1], g[2], g[3], g[4], g[5]
        if first:  # the block's x WAS the stream (s1 = res): dx already holds d(stream) + d(LayerNorm path)
            dres, dx, dbias_prev = dx, None, None
        dpos = _add(dpos1, dpos2) if has_pos else None
        return (dres, dx, dbias_prev, dmem, dmem_pos, dpos, None, None, None,
                dg1, db1n, din1, dib1, dow1, dob1, dg2, db2n, din2, dib2, dow2, dob2, dg3, db3n, dw1, dfb1, dw2)


class _EncoderLayer(torch.autograd.Function):
    """Pending residual stream (res, x, bias, p) -> (s, o) after the self-attention block, with
    the feed-forward product pending, or (s, a) with the attention product pending when the layer
    has no feed-forward part."""

    @staticmethod
    def forward(ctx, res, x, bias_prev, pos, mask, cfg, g1, b1n, in1, ib1, ow1, ob1, g2, b2n, w1, fb1, w2):
        p_prev, eps, nheads, p_attn, p1, p_ffn, use_ffn = cfg
        c1 = _Ctx()
        if x is None:
            _, y1, y1p = _AddLN.forward(c1, res, None, None, pos, g1, b1n, eps, 0.0)
            s1 = res
        else:
            s1, y1, y1p = _AddLN.forward(c1, x, bias_prev, res, pos, g1, b1n, eps, p_prev)
        qk = y1 if pos is None else y1p
        c2 = _Ctx()
        a1 = _MHA.forward(c2, qk, qk, y1, in1, ib1, ow1, mask, nheads, p_attn)
        ctx.flags = (x is None, pos is not None, use_ffn)
        if not use_ffn:
            ctx.blocks = (c1, c2)
            return s1, a1
        c3 = _Ctx()
        s2, y2, _ = _AddLN.forward(c3, a1, ob1, s1, None, g2, b2n, eps, p1)
        o, ffn_saved = _ffn_forward(y2, w1, fb1, w2, p_ffn)
        ctx.blocks = (c1, c2, c3, ffn_saved)
        ctx.save_for_backward(w1, w2)
        return s2, o

    @staticmethod
    def backward(ctx, ds_out, do):
        first, has_pos, use_ffn = ctx.flags
        dg2 = db2n = dw1 = dfb1 = dw2 = dob1 = None
        if use_ffn:
            c1, c2, c3, ffn_saved = ctx.blocks
            w1, w2 = ctx.saved_tensors
            dy2, dw1, dfb1, dw2 = _ffn_backward(ffn_saved, do, w1, w2)
            da1, dob1, ds1, _, dg2, db2n, _, _ = _AddLN.backward(c3, ds_out, dy2, None)
        else:
            c1, c2 = ctx.blocks
            da1, ds1 = do, ds_out
        dqk, _, dv1, din1, dib1, dow1, _, _, _ = _MHA.backward(c2, da1)
        if has_pos:
            g = _AddLN.backward(c1, ds1, dv1, dqk)
        else:  # q = k = v = y1: the attention block returned the summed gradient as dxq
            g = _AddLN.backward(c1, ds1, _add(dv1, dqk), None)
        dx, dbias_prev, dres, dpos, dg1, db1n = g[0], g[1], g[2], g[3], g[4], g[5]
        if first:
            dres, dx, dbias_prev = dx, None, None
        return (dres, dx, dbias_prev, dpos if has_pos else None, None, None,
                dg1, db1n, din1, dib1, dow1, dob1, dg2, db2n, dw1, dfb1, dw2)


def decoder_layer(layer, pend, memory, memory_pos, query_pos, self_mask, cross_mask):
    """-> (s, o): see ``_DecoderLayer``; ``layer`` is a TransformerDecoderLayer."""
    def p(m):
        return float(m.p) if m.training else 0.0

    sa, ca = layer.self_attn, layer.multihead_attn
    cfg = (float(pend.p), float(layer.norm1.eps), sa.num_heads, float(sa.dropout) if sa.training else 0.0,
           p(layer.dropout1), p(layer.dropout2), p(layer.dropout))
    return _DecoderLayer.apply(pend.res, pend.x, pend.bias, memory, memory_pos, query_pos, self_mask, cross_mask, cfg,
                               layer.norm1.weight, layer.norm1.bias, sa.in_proj_weight, sa.in_proj_bias,
                               sa.out_proj.weight, sa.out_proj.bias, layer.norm2.weight, layer.norm2.bias,
                               ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias,
                               layer.norm3.weight, layer.norm3.bias, layer.linear1.weight, layer.linear1.bias,
                               layer.linear2.weight)


def encoder_layer(layer, pend, pos, mask):
    """-> (s, o): see ``_EncoderLayer``; ``layer`` is a TransformerEncoderLayer."""
    def p(m):
        return float(m.p) if m.training else 0.0

    sa = layer.self_attn
    use_ffn = bool(layer.use_ffn)
    cfg = (float(pend.p), float(layer.norm1.eps), sa.num_heads, float(sa.dropout) if sa.training else 0.0,
           p(layer.dropout1), p(layer.dropout) if use_ffn else 0.0, use_ffn)
    none = None
    return _EncoderLayer.apply(pend.res, pend.x, pend.bias, pos, mask, cfg,
                               layer.norm1.weight, layer.norm1.bias, sa.in_proj_weight, sa.in_proj_bias,
                               sa.out_proj.weight, sa.out_proj.bias,
                               layer.norm2.weight if use_ffn else none, layer.norm2.bias if use_ffn else none,
                               layer.linear1.weight if use_ffn else none, layer.linear1.bias if use_ffn else none,
                               layer.linear2.weight if use_ffn else none)
