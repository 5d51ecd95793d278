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
import os

import torch

from . import gemm
from .fused_layers import _AddLN, _FfnAct, _MHA, _colsum_into
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
    h0 = gemm.linear(y2, w1)
    cf = _Ctx()
    h = _FfnAct.forward(cf, h0, b1, p)
    o = gemm.linear(h, w2).view(y.shape[:-1] + (w2.shape[0],))
    return o, (cf, y2, h)


def _ffn_backward(saved, do, w1, w2, defer=None):
    cf, y2, h = saved
    do2 = do.reshape(-1, do.shape[-1]).contiguous()
    dh = gemm.mm(do2, w2)
    cf.defer = defer
    dh0, db1, _ = _FfnAct.backward(cf, dh)
    if defer is not None:  # weight gradients join the stack's grouped launch (gemm.DeferredWeightGrads)
        dw2, dw1 = torch.empty_like(w2), torch.empty_like(w1)
        defer.add(dw2, do2, h)
        defer.add(dw1, dh0, y2)
    else:
        dw2 = tn_gemm(do2, h)
        dw1 = tn_gemm(dh0, y2)
    dy = gemm.mm(dh0, w1)
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
        # nothing inside this backward reads a weight / bias / LayerNorm gradient: the sums over the row chunks of
        # the six weight-gradient products and the ~5 column-sum reductions of the layer close in two grouped launches
        # at its end (11 launches before)
        defer = gemm.DeferredWeightGrads(sums_only=True) if gemm.DEFER_SUMS else None
        for c in ctx.blocks[:3]:
            c.defer = defer
        if use_ffn:
            c1, c2, c3, ffn_saved = ctx.blocks
            w1, w2 = ctx.saved_tensors
            dy2, dw1, dfb1, dw2 = _ffn_backward(ffn_saved, do, w1, w2, defer)
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
        if defer is not None:
            defer.flush()
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


# ---- the whole decoder as one node ------------------------------------------------------------
#
# Every decoder layer projects the SAME encoder memory to its cross-attention keys and values
# (models/transformer.py:566-573).  With the decoder as a single autograd node these 2 x 8
# projections of the 16 384 memory tokens are two GEMMs over the concatenated weights before the
# layer loop, the attention kernels read layer l's keys / values as column slices of the packed
# (tokens, 8*E) results and write their gradients back into the same slices, and the memory /
# weight gradients of all layers are again two GEMMs each after the loop -- instead of 8 x
# (2 forward GEMMs, 2 input-gradient GEMMs, 2 split-K weight-gradient GEMMs, bias column sums
# and 2 gradient accumulations over the 16 MB memory tensor).  The per-layer bodies are the
# blocks of fused_layers.py; the decoder norm of every layer output (return_intermediate) is the
# LayerNorm of the kernel that materialises the layer's residual stream.
_NP = 18  # parameters per decoder layer, in the order built by `decoder_stack` below


# CODA_DKV_STREAM=1: cross-attention dK/dV of every layer on a side stream (only dQ is on the dependency chain of
# the decoder's backward).  Measured and left OFF: 18.9-19.1 ms/step against 18.5-18.6 in line -- the big kernel
# competes with the chain's launch-sized kernels for CUs and the per-layer event pair costs more than the idle
# slots it fills (same-box A/B).
DKV_SIDE_STREAM = os.environ.get("CODA_DKV_STREAM", "0") == "1"
_SIDE_STREAMS = {}


def _side_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def _colsum_vec(x2):
    out = torch.empty(x2.shape[1], dtype=torch.float32, device=x2.device)
    _colsum_into(out, x2.unsqueeze(0))
    return out


# Row padding of the packed (tokens, nl * E) key / value buffers: with 8 layers of width 256 the rows of a layer's
# column slice are exactly 8 KB apart and land on a fraction of the memory channels (cross-attention dK/dV: 120 us on
# such slices, 96 us with 64 more floats per row, tools/probe_attn_layout.py).  CODA_KV_PAD=0 switches it off (A/B).
_KV_PAD = int(os.environ.get("CODA_KV_PAD", "64"))


def _kv_buffer(rows, cols, dev):
    """(rows, cols) float32 view with a row stride of cols + _KV_PAD."""
    return torch.empty((rows, cols + _KV_PAD), dtype=torch.float32, device=dev)[:, :cols]


class _DecoderStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tgt, memory, pos, query_pos, self_mask, cross_mask, cfg, norm_g, norm_b, *params):
        from . import _lib
        from . import attention_core as _core
        eps, nheads, p_attn, p1, p2, p_ffn, p3 = cfg
        nl = len(params) // _NP
        nq, bsz, e = tgt.shape
        ns = memory.shape[0]
        d = e // nheads
        dev = tgt.device
        lib = _lib.load()
        layers = [params[_NP * l:_NP * (l + 1)] for l in range(nl)]
        mem2 = memory.reshape(-1, e)
        mp2 = mem2 if pos is None else (memory + pos).reshape(-1, e)
        wk_all = torch.cat([lp[8][e:2 * e] for lp in layers])          # in_proj rows of the keys   (nl*E, E)
        wv_all = torch.cat([lp[8][2 * e:] for lp in layers])
        gemm.declare_weight(wk_all)   # (built per step from the layers' parameters: gemm.py, x3 route)
        gemm.declare_weight(wv_all)
        bk_all = torch.cat([lp[9][e:2 * e] for lp in layers])
        bv_all = torch.cat([lp[9][2 * e:] for lp in layers])
        k_all = gemm.linear(mp2, wk_all, bk_all, out=_kv_buffer(mp2.shape[0], nl * e, dev))   # (S*B, nl*E)
        v_all = gemm.linear(mem2, wv_all, bv_all, out=_kv_buffer(mp2.shape[0], nl * e, dev))
        ld_kv = k_all.stride(0)
        scale = 1.0 / (d ** 0.5)
        mask_ptr = cross_mask.data_ptr() if cross_mask is not None else None

        blocks, outs = [], []
        res = tgt
        for l, lp in enumerate(layers):
            g1, b1n, in1, ib1, ow1, ob1, g2, b2n, in2, ib2, ow2, ob2, g3, b3n, w1, fb1, w2, fb2 = lp
            c1 = _Ctx()
            _, y1, y1p = _AddLN.forward(c1, res, None, None, query_pos, g1, b1n, eps, 0.0)
            qk = y1 if query_pos is None else y1p
            c2 = _Ctx()
            a1 = _MHA.forward(c2, qk, qk, y1, in1, ib1, ow1, self_mask, nheads, p_attn)
            c3 = _Ctx()
            s2, y2, y2p = _AddLN.forward(c3, a1, ob1, res, query_pos, g2, b2n, eps, p1)
            # cross attention on the pre-projected memory
            xq2 = (y2 if query_pos is None else y2p).reshape(-1, e)
            q = gemm.linear(xq2, in2[:e], ib2[:e])
            attn = torch.empty((nq * bsz, e), dtype=torch.float32, device=dev)
            lse = torch.empty((bsz, nheads, nq), dtype=torch.float32, device=dev)
            seed, seed_dev = _core._next_seed() if p_attn > 0.0 else (0, None)
            _lib.check(lib.coda_mha_fwd_opt_f32(q.data_ptr(), k_all.data_ptr() + 4 * l * e, v_all.data_ptr() + 4 * l * e,
                                                mask_ptr, attn.data_ptr(), lse.data_ptr(), bsz, nheads, nq, ns, d, e, ld_kv,
                                                ld_kv, scale, float(p_attn), seed,
                                                seed_dev.data_ptr() if seed_dev is not None else None,
                                                _lib.opt("mfma_dtype"), _lib.current_stream_handle()), "mha_fwd")
            a2 = gemm.linear(attn, ow2).view(nq, bsz, e)
            c5 = _Ctx()
            s3, y3, _ = _AddLN.forward(c5, a2, ob2, s2, None, g3, b3n, eps, p2)
            o, ffn_saved = _ffn_forward(y3, w1, fb1, w2, p_ffn)
            cn = _Ctx()  # materialise the layer output and apply the decoder norm to it in the same pass
            s4, yn, _ = _AddLN.forward(cn, o, fb2, s3, None, norm_g, norm_b, eps, p3)
            blocks.append((c1, c2, c3, (xq2, q, attn, lse, seed, seed_dev), c5, ffn_saved, cn))
            outs.append(yn)
            res = s4
        ctx.blocks = blocks
        ctx.dims = (nl, nq, bsz, e, ns, nheads, scale, float(p_attn), query_pos is not None, pos is not None)
        ctx.mfma_dtype = _lib.opt("mfma_dtype")
        ctx.cross_mask = cross_mask
        ctx.save_for_backward(mem2, mp2, k_all, v_all, wk_all, wv_all, *params)
        return torch.stack(outs)

    @staticmethod
    def backward(ctx, dstack):
        from . import _lib
        nl, nq, bsz, e, ns, nheads, scale, p_attn, has_qpos, has_pos = ctx.dims
        saved = ctx.saved_tensors
        mem2, mp2, k_all, v_all, wk_all, wv_all = saved[:6]
        params = saved[6:]
        layers = [params[_NP * l:_NP * (l + 1)] for l in range(nl)]
        dev = dstack.device
        lib = _lib.load()
        d = e // nheads
        ld_kv = k_all.stride(0)
        mask_ptr = ctx.cross_mask.data_ptr() if ctx.cross_mask is not None else None
        dstack = dstack.contiguous()
        dk_all = _kv_buffer(k_all.shape[0], k_all.shape[1], dev)
        dv_all = _kv_buffer(k_all.shape[0], k_all.shape[1], dev)
        din2_all = torch.empty((nl, 3 * e, e), dtype=torch.float32, device=dev)   # cross in_proj weight grads
        dib2_all = torch.empty((nl, 3 * e), dtype=torch.float32, device=dev)
        dnorm_g, dnorm_b = [], []                                                  # decoder norm: per-layer parts
        grads = [None] * (nl * _NP)
        ds_next = None
        dqpos = None
        defer = gemm.DeferredWeightGrads()  # the layers' weight gradients: one grouped launch after the loop
        side = _side_stream(dev) if DKV_SIDE_STREAM else None
        main = torch.cuda.current_stream(dev) if side is not None else None
        held = []
        for l in range(nl - 1, -1, -1):
            g1, b1n, in1, ib1, ow1, ob1, g2, b2n, in2, ib2, ow2, ob2, g3, b3n, w1, fb1, w2, fb2 = layers[l]
            c1, c2, c3, (xq2, q, attn, lse, seed, seed_dev), c5, ffn_saved, cn = ctx.blocks[l]
            cn.defer = c5.defer = c3.defer = c2.defer = c1.defer = defer
            do, dfb2, ds3, _, dgn, dbn, _, _ = _AddLN.backward(cn, ds_next, dstack[l], None)
            dnorm_g.append(dgn)
            dnorm_b.append(dbn)
            dy3, dw1, dfb1, dw2 = _ffn_backward(ffn_saved, do, w1, w2, defer)
            da2, dob2, ds2, _, dg3, db3n, _, _ = _AddLN.backward(c5, ds3, dy3, None)
            # cross attention backward; dK / dV go straight into layer l's column slices
            da2_2 = da2.reshape(-1, e)
            dow2 = torch.empty_like(ow2)
            defer.add(dow2, da2_2.contiguous(), attn)
            dattn = gemm.mm(da2_2, ow2)
            dq = torch.empty((nq * bsz, e), dtype=torch.float32, device=dev)
            delta = torch.empty((bsz, nheads, nq), dtype=torch.float32, device=dev)
            bwd_args = (q.data_ptr(), k_all.data_ptr() + 4 * l * e, v_all.data_ptr() + 4 * l * e, mask_ptr,
                        attn.data_ptr(), lse.data_ptr(), dattn.data_ptr())
            bwd_dims = (bsz, nheads, nq, ns, d, e, ld_kv, ld_kv, 0, ld_kv, ld_kv, scale, p_attn, seed,
                        seed_dev.data_ptr() if seed_dev is not None else None)
            dkv_ptrs = (dk_all.data_ptr() + 4 * l * e, dv_all.data_ptr() + 4 * l * e)
            if side is None:
                _lib.check(lib.coda_mha_bwd_parts_opt_f32(*bwd_args, dq.data_ptr(), *dkv_ptrs, delta.data_ptr(), *bwd_dims,
                                                          7, ctx.mfma_dtype, _lib.current_stream_handle()), "mha_bwd")
            else:
                # only dQ is on the dependency chain of this backward; dK / dV of every layer are consumed after
                # the loop.  delta + dQ run here, the dK/dV kernel on the side stream behind an event, where it
                # fills the CUs that the chain of launch-sized kernels below leaves idle.
                _lib.check(lib.coda_mha_bwd_parts_opt_f32(*bwd_args, dq.data_ptr(), None, None, delta.data_ptr(),
                                                          *bwd_dims, 1 | 4, ctx.mfma_dtype,
                                                          _lib.current_stream_handle()), "mha_bwd dq")
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                _lib.check(lib.coda_mha_bwd_parts_opt_f32(*bwd_args, None, *dkv_ptrs, delta.data_ptr(), *bwd_dims, 2,
                                                          ctx.mfma_dtype, side.cuda_stream), "mha_bwd dkv")
                held.append((dattn, delta))  # read by the side stream: freed only after the join below
            defer.add(din2_all[l, :e], dq, xq2)
            _colsum_into(dib2_all[l, :e], dq.unsqueeze(0), defer)
            dxq = gemm.mm(dq, in2[:e]).view(nq, bsz, e)
            da1, dob1, ds1, dpos2, dg2, db2n, _, _ = _AddLN.backward(c3, ds2, None if has_qpos else dxq,
                                                                     dxq if has_qpos else None)
            dqk, _, dv1, din1, dib1, dow1, _, _, _ = _MHA.backward(c2, da1)
            if has_qpos:
                g = _AddLN.backward(c1, ds1, dv1, dqk)
                dqpos = _add(dqpos, g[3] + dpos2)
            else:
                g = _AddLN.backward(c1, ds1, _add(dv1, dqk), None)
            ds_next = g[0]  # the block's input WAS the stream: d(stream) + d(LayerNorm path)
            grads[_NP * l:_NP * (l + 1)] = [g[4], g[5], din1, dib1, dow1, dob1, dg2, db2n, din2_all[l], dib2_all[l],
                                            dow2, dob2, dg3, db3n, dw1, dfb1, dw2, dfb2]
        defer.flush()
        if side is not None:
            main.wait_stream(side)
            held.clear()
        # memory side of all layers at once
        dmp2 = gemm.mm(dk_all, wk_all)
        dmem2 = gemm.mm(dv_all, wv_all)
        dwk = tn_gemm(dk_all, mp2)   # (nl*E, E)
        dwv = tn_gemm(dv_all, mem2)
        din2_all[:, e:2 * e].copy_(dwk.view(nl, e, e))
        din2_all[:, 2 * e:].copy_(dwv.view(nl, e, e))
        dib2_all[:, e:2 * e].copy_(_colsum_vec(dk_all).view(nl, e))
        dib2_all[:, 2 * e:].copy_(_colsum_vec(dv_all).view(nl, e))
        dnorm_sum = (torch.stack(dnorm_g).sum(0), torch.stack(dnorm_b).sum(0))
        if has_pos:
            dmemory = (dmp2 + dmem2).view(ns, bsz, e)
            dpos = dmp2.view(ns, bsz, e)
        else:
            dmemory = (dmp2 + dmem2).view(ns, bsz, e)
            dpos = None
        return (ds_next, dmemory, dpos, dqpos, None, None, None, dnorm_sum[0], dnorm_sum[1], *grads)


# ---- the same node with its per-layer launch sequences issued from C++ (include/coda_stack.h) -------------------
#
# _DecoderStack above enqueues ~280 launches per training step from Python (~5 ms of host time); here the layer
# loops of forward and backward are one C call each (csrc/decoder_stack.hip runs the identical sequence through the
# same entry points).  What stays in Python is what is not launch-bound: the two large memory projections before
# the loop, the memory-side GEMMs after it, parameter / gradient bookkeeping.  CODA_STACK=python selects the
# per-launch node (A/B, and the fallback for configurations the C driver does not cover).
import ctypes  # noqa: E402

STACK_IN_C = os.environ.get("CODA_STACK", "c") != "python"


class _StackArgs(ctypes.Structure):
    _fields_ = [("nl", ctypes.c_int), ("nq", ctypes.c_int), ("bsz", ctypes.c_int), ("e", ctypes.c_int),
                ("ns", ctypes.c_int), ("nheads", ctypes.c_int), ("ffn", ctypes.c_int),
                ("eps", ctypes.c_float), ("p_attn", ctypes.c_float), ("p1", ctypes.c_float), ("p2", ctypes.c_float),
                ("p_ffn", ctypes.c_float), ("p3", ctypes.c_float), ("seed", ctypes.c_uint64),
                ("tgt", ctypes.c_void_p), ("query_pos", ctypes.c_void_p), ("k_all", ctypes.c_void_p),
                ("v_all", ctypes.c_void_p), ("norm_g", ctypes.c_void_p), ("norm_b", ctypes.c_void_p),
                ("params", ctypes.c_void_p), ("outs", ctypes.c_void_p), ("ws", ctypes.c_void_p),
                ("ld_kv", ctypes.c_int), ("mfma_dtype", ctypes.c_int), ("attn_ws", ctypes.c_void_p),
                ("attn_ws_bytes", ctypes.c_size_t)]


def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class _DecoderStackC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tgt, memory, pos, query_pos, cfg, norm_g, norm_b, *params):
        from . import _lib
        from . import attention_core as _core
        eps, nheads, p_attn, p1, p2, p_ffn, p3 = cfg
        nl = len(params) // _NP
        nq, bsz, e = tgt.shape
        ns = memory.shape[0]
        ffn = params[14].shape[0]
        dev = tgt.device
        lib = _lib.load()
        layers = [params[_NP * l:_NP * (l + 1)] for l in range(nl)]
        mem2 = memory.reshape(-1, e)
        mp2 = mem2 if pos is None else (memory + pos).reshape(-1, e)
        wk_all = torch.cat([lp[8][e:2 * e] for lp in layers])
        wv_all = torch.cat([lp[8][2 * e:] for lp in layers])
        gemm.declare_weight(wk_all)   # (built per step from the layers' parameters: gemm.py, x3 route)
        gemm.declare_weight(wv_all)
        bk_all = torch.cat([lp[9][e:2 * e] for lp in layers])
        bv_all = torch.cat([lp[9][2 * e:] for lp in layers])
        k_all = gemm.linear(mp2, wk_all, bk_all, out=_kv_buffer(mp2.shape[0], nl * e, dev))
        v_all = gemm.linear(mem2, wv_all, bv_all, out=_kv_buffer(mp2.shape[0], nl * e, dev))
        ws = torch.empty(lib.coda_decoder_stack_ws_floats(nl, nq, bsz, e, nheads, ffn), dtype=torch.float32, device=dev)
        outs = torch.empty((nl, nq, bsz, e), dtype=torch.float32, device=dev)
        tgt, query_pos = tgt.contiguous(), query_pos.contiguous()
        seed = _core._next_seed()[0] if max(p_attn, p1, p2, p_ffn, p3) > 0.0 else 0
        table = _ptr_table(params)
        args = _StackArgs(nl, nq, bsz, e, ns, nheads, ffn, eps, p_attn, p1, p2, p_ffn, p3, seed, tgt.data_ptr(),
                          query_pos.data_ptr(), k_all.data_ptr(), v_all.data_ptr(), norm_g.data_ptr(), norm_b.data_ptr(),
                          ctypes.addressof(table), outs.data_ptr(), ws.data_ptr(), k_all.stride(0), _lib.opt("mfma_dtype"),
                          None, 0)
        _lib.check(lib.coda_decoder_stack_fwd_f32(ctypes.byref(args), _lib.current_stream_handle()), "decoder_stack_fwd")
        ctx.args = (nl, nq, bsz, e, ns, nheads, ffn, eps, p_attn, p1, p2, p_ffn, p3, seed, pos is not None)
        ctx.mfma_dtype = args.mfma_dtype
        ctx.save_for_backward(tgt, query_pos, mem2, mp2, k_all, v_all, wk_all, wv_all, ws, norm_g, norm_b, *params)
        return outs

    @staticmethod
    def backward(ctx, dstack):
        from . import _lib
        from . import attention_core as _core
        nl, nq, bsz, e, ns, nheads, ffn, eps, p_attn, p1, p2, p_ffn, p3, seed, has_pos = ctx.args
        saved = ctx.saved_tensors
        tgt, query_pos, mem2, mp2, k_all, v_all, wk_all, wv_all, ws, norm_g, norm_b = saved[:11]
        params = saved[11:]
        dev = dstack.device
        lib = _lib.load()
        f32 = dict(dtype=torch.float32, device=dev)
        dstack = dstack.contiguous()
        dk_all = _kv_buffer(k_all.shape[0], k_all.shape[1], dev)
        dv_all = _kv_buffer(k_all.shape[0], k_all.shape[1], dev)
        d_tgt = torch.empty((nq, bsz, e), **f32)
        d_qpos = torch.empty((nq, bsz, e), **f32)
        sums = torch.empty((nl, 4, 3 * e), **f32)
        bws = torch.empty(lib.coda_decoder_stack_bwd_ws_floats(nl, nq, bsz, e, nheads, ffn), **f32)
        din1 = torch.empty((nl, 3 * e, e), **f32)        # self_attn.in_proj_weight
        dib1 = torch.empty((nl, 3 * e), **f32)
        dow1 = torch.empty((nl, e, e), **f32)
        din2 = torch.empty((nl, 3 * e, e), **f32)        # multihead_attn.in_proj_weight (query rows by the C driver)
        dib2 = torch.empty((nl, 3 * e), **f32)
        dow2 = torch.empty((nl, e, e), **f32)
        dw1 = torch.empty((nl, ffn, e), **f32)
        dfb1 = torch.empty((nl, ffn), **f32)
        dw2 = torch.empty((nl, e, ffn), **f32)
        gptr = []
        for l in range(nl):
            row = [None] * _NP
            row[2], row[3], row[4] = din1[l], dib1[l], dow1[l]
            row[8], row[9], row[10] = din2[l], dib2[l], dow2[l]
            row[14], row[15], row[16] = dw1[l], dfb1[l], dw2[l]
            gptr += row
        gtable = _ptr_table(gptr)
        table = _ptr_table(params)
        # dS workspace of the attention backward (the cross-attention problem is the larger one; the layers share it)
        attn_ws, attn_ws_bytes = _core.backward_workspace(bsz, nheads, nq, max(ns, nq), e // nheads, dev, ctx.mfma_dtype)
        args = _StackArgs(nl, nq, bsz, e, ns, nheads, ffn, eps, p_attn, p1, p2, p_ffn, p3, seed, tgt.data_ptr(),
                          query_pos.data_ptr(), k_all.data_ptr(), v_all.data_ptr(), norm_g.data_ptr(), norm_b.data_ptr(),
                          ctypes.addressof(table), None, ws.data_ptr(), k_all.stride(0), ctx.mfma_dtype,
                          attn_ws.data_ptr() if attn_ws is not None else None, attn_ws_bytes)
        _lib.check(lib.coda_decoder_stack_bwd_f32(ctypes.byref(args), dstack.data_ptr(), d_tgt.data_ptr(), d_qpos.data_ptr(),
                                                  dk_all.data_ptr(), dv_all.data_ptr(), ctypes.addressof(gtable),
                                                  sums.data_ptr(), bws.data_ptr(), _lib.current_stream_handle()),
                   "decoder_stack_bwd")
        # memory side of all layers at once
        dmp2 = gemm.mm(dk_all, wk_all)
        dmem2 = gemm.mm(dv_all, wv_all)
        dwk = tn_gemm(dk_all, mp2)   # (nl*E, E)
        dwv = tn_gemm(dv_all, mem2)
        din2[:, e:2 * e].copy_(dwk.view(nl, e, e))
        din2[:, 2 * e:].copy_(dwv.view(nl, e, e))
        dib2[:, e:2 * e].copy_(_colsum_vec(dk_all).view(nl, e))
        dib2[:, 2 * e:].copy_(_colsum_vec(dv_all).view(nl, e))
        dnorm = sums[:, 3].sum(0)
        grads = []
        for l in range(nl):
            s0, s1, s2, s3 = sums[l, 0], sums[l, 1], sums[l, 2], sums[l, 3]
            grads += [s0[:e], s0[e:2 * e], din1[l], dib1[l], dow1[l], s1[2 * e:], s1[:e], s1[e:2 * e], din2[l], dib2[l],
                      dow2[l], s2[2 * e:], s2[:e], s2[e:2 * e], dw1[l], dfb1[l], dw2[l], s3[2 * e:]]
        dmemory = (dmp2 + dmem2).view(ns, bsz, e)
        dpos = dmp2.view(ns, bsz, e) if has_pos else None
        return (d_tgt, dmemory, dpos, d_qpos, None, dnorm[:e], dnorm[e:2 * e], *grads)


def _stack_in_c_applies(decoder, tgt, memory, query_pos, self_mask, cross_mask, params):
    if not (STACK_IN_C and query_pos is not None and self_mask is None and cross_mask is None and tgt.is_cuda):
        return False
    e = tgt.shape[-1]
    heads = decoder.layers[0].self_attn.num_heads
    if e % heads or e // heads not in (64, 128) or params[14].shape[0] % 4:
        return False
    return all(t.dtype == torch.float32 and t.is_contiguous() for t in (tgt, memory, *params))


def decoder_stack(decoder, tgt, memory, pos, query_pos, self_mask, cross_mask):
    """All layers of a TransformerDecoder (pre-norm, return_intermediate) -> (num_layers, nq, B, E):
    the decoder-normed output of every layer."""
    def p(m):
        return float(m.p) if m.training else 0.0

    first = decoder.layers[0]
    sa = first.self_attn
    cfg = (float(first.norm1.eps), sa.num_heads, float(sa.dropout) if sa.training else 0.0,
           p(first.dropout1), p(first.dropout2), p(first.dropout), p(first.dropout3))
    params = []
    for layer in decoder.layers:
        s, c = layer.self_attn, layer.multihead_attn
        params += [layer.norm1.weight, layer.norm1.bias, s.in_proj_weight, s.in_proj_bias, s.out_proj.weight,
                   s.out_proj.bias, layer.norm2.weight, layer.norm2.bias, c.in_proj_weight, c.in_proj_bias,
                   c.out_proj.weight, c.out_proj.bias, layer.norm3.weight, layer.norm3.bias, layer.linear1.weight,
                   layer.linear1.bias, layer.linear2.weight, layer.linear2.bias]
    if _stack_in_c_applies(decoder, tgt, memory, query_pos, self_mask, cross_mask, params):
        return _DecoderStackC.apply(tgt, memory.contiguous(), pos, query_pos, cfg, decoder.norm.weight,
                                    decoder.norm.bias, *params)
    return _DecoderStack.apply(tgt, memory.contiguous(), pos, query_pos, self_mask, cross_mask, cfg,
                               decoder.norm.weight, decoder.norm.bias, *params)
