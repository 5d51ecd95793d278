"""3DETR transformer encoder / decoder.

Mirror of the LIVE classes of models/transformer.py: ``TransformerEncoder``
(:19-74), ``MaskedTransformerEncoder`` (:146-211), ``TransformerEncoderLayer``
(:412-494), ``TransformerDecoder`` (:77-143), ``TransformerDecoderLayer``
(:497-594).  Constructor keywords, forward signatures, return tuples and
``state_dict`` keys are the reference's; the attention modules are this
package's ``MultiheadAttention`` (same parameter names as
``nn.MultiheadAttention``).  The classes the reference defines but never
constructs (``TransformerCrossEncoder``, ``*SharedAttention*``, ``*Guidence*``)
are out of scope.

The head-averaged attention-weight tensor is only produced when a caller asks
for it (``return_attn_weights=True``); the reference computes it on every call
and drops it (transformer.py:470-479,570-580).
"""
from typing import Optional

import torch
from torch import Tensor, nn

import os

from . import _lib
from . import attention_core as _core
from . import fused_blocks as _fb
from . import fused_layers as _fl
from .attention import MultiheadAttention
from .helpers import ACTIVATION_DICT, NORM_DICT, WEIGHT_INIT_DICT, get_clones
from .linear_fn import linear as _linear


# ---- fused pre-norm layers -----------------------------------------------------------------
# The layer classes below keep the reference's module structure and run it op by op
# (``forward_pre`` / ``forward_post``) for every configuration.  For the configuration the CoDA
# models build -- pre-norm, LayerNorm, ReLU, fp32 on the GPU, head_dim 64/128 -- the containers
# run the same arithmetic through ``fused_layers``: every bias / Dropout / residual /
# LayerNorm / ``with_pos_embed`` chain between two GEMMs is one kernel each way, and the
# residual stream is carried between layers as a PENDING sum ``res + dropout_p(x + bias)``
# that the next LayerNorm kernel resolves.
class _Pending:
    __slots__ = ("res", "x", "bias", "p")

    def __init__(self, res, x=None, bias=None, p=0.0):
        self.res, self.x, self.bias, self.p = res, x, bias, p


def _resolve(pend, norm=None, pos=None):
    """-> (s, y, yp) with s the materialised residual stream."""
    if pend.x is None:
        if norm is None:
            return pend.res, None, None
        return _fl.add_ln(pend.res, norm=norm, pos=pos)
    return _fl.add_ln(pend.x, norm=norm, bias=pend.bias, res=pend.res, pos=pos, p=pend.p)


def _layer_nodes():
    # A/B switch: "ops" chains the fused_layers blocks through autograd (one node per block)
    return os.environ.get("CODA_LAYER_NODES", "layer") != "ops"


def _drop_p(mod):
    return float(mod.p) if mod.training else 0.0


def _attn_ok(attn):
    e = attn.embed_dim
    return isinstance(attn, MultiheadAttention) and attn.head_dim in (64, 128) and attn.in_proj_bias is not None \
        and e % 4 == 0 and e <= 1024


def _ffn_ok(layer):
    f = layer.linear1.out_features
    return isinstance(layer.activation, nn.ReLU) and f % 4 == 0 and f <= 1024 and 256 % (f // 4) == 0


def _fusable(layers, x, norms_of):
    if os.environ.get("CODA_LAYERS", "fused") == "modules":  # A/B switch: module-by-module path
        return False
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32):
        return False
    for layer in layers:
        if not layer.normalize_before or not all(isinstance(n, nn.LayerNorm) and n.elementwise_affine
                                                 for n in norms_of(layer)):
            return False
    return True


def _mask_u8(attn_mask, key_padding_mask, bsz, nhead, tgt_len, src_len):
    mask = _core.merge_masks(attn_mask, key_padding_mask, bsz, nhead, tgt_len, src_len)
    return None if mask is None else mask.contiguous().to(torch.uint8)


def _ffn_fused(layer, y):
    """linear2's product WITHOUT its bias (it joins the pending residual)."""
    h = _linear(y, layer.linear1.weight, None)
    h = _fl.ffn_act(h, layer.linear1.bias, _drop_p(layer.dropout))
    return _linear(h, layer.linear2.weight, None)


def _ffn(layer, x):
    """linear2(dropout(activation(linear1(x)))) with split-K weight gradients (linear_fn.py)."""
    h = layer.activation(_linear(x, layer.linear1.weight, layer.linear1.bias))
    return _linear(layer.dropout(h), layer.linear2.weight, layer.linear2.bias)


def _tile_mask_per_head(mask, nhead):
    # (B, n, n) -> (B*nhead, n, n), head-major within a scene (transformer.py:55-60)
    bsz, n, _ = mask.shape
    return mask.unsqueeze(1).repeat(1, nhead, 1, 1).view(bsz * nhead, n, n)


def _sublayer(x, norm, branch, drop, pre_norm):
    """One residual sub-layer: pre-norm ``x + drop(branch(norm(x)))``, post-norm ``norm(x + drop(branch(x)))``
    (``norm=None``: no normalisation on that side)."""
    if pre_norm:
        return x + drop(branch(x if norm is None else norm(x)))
    y = x + drop(branch(x))
    return y if norm is None else norm(y)


def _plus(t, pos):
    return t if pos is None else t + pos


def _xavier_matrices(module, weight_init_name):
    init = WEIGHT_INIT_DICT[weight_init_name]
    for prm in module.parameters():
        if prm.dim() > 1:  # matrices only: norm scales and biases keep their defaults
            init(prm)


class _FeatureMapIO:
    """``transpose_swap``: the caller holds (B, C, H, W) feature maps instead of (tokens, B, C) sequences."""

    def __init__(self, enabled, ref):
        self.shape = tuple(ref.shape) if enabled else None

    def seq(self, t):
        return t if (self.shape is None or t is None) else t.flatten(2).permute(2, 0, 1)

    def back(self, t):
        return t if self.shape is None else t.permute(1, 2, 0).view(*self.shape).contiguous()


class TransformerEncoder(nn.Module):
    """Stack of encoder layers; ``forward`` returns ``(xyz, features, None)`` like the masked variant (which
    can down-sample and then reports the kept indices)."""

    def __init__(self, encoder_layer, num_layers, norm=None, weight_init_name="xavier_uniform"):
        super().__init__()
        self.layers = get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self._reset_parameters(weight_init_name)

    def _reset_parameters(self, weight_init_name):
        _xavier_matrices(self, weight_init_name)

    def _layer_masks(self, mask):
        """One (B, n, n) mask or a list of them -> per-layer (B*nhead, n, n) masks (None: unmasked)."""
        if mask is None:
            return [None] * len(self.layers)
        per_layer = mask if isinstance(mask, list) else [mask] * len(self.layers)
        if len(per_layer) != len(self.layers):
            raise AssertionError("one mask per encoder layer expected")
        return [_tile_mask_per_head(m, layer.nhead) for m, layer in zip(per_layer, self.layers)]

    @_lib.on_tensor_device()
    def forward(self, src, mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                xyz: Optional[Tensor] = None, transpose_swap: Optional[bool] = False):
        io = _FeatureMapIO(transpose_swap, src)
        x, pos = io.seq(src), io.seq(pos)
        masks = self._layer_masks(mask)
        if TransformerEncoderLayer.fusable(self.layers, x) and (self.norm is None or isinstance(self.norm, nn.LayerNorm)):
            pend = _Pending(x)
            for layer, m in zip(self.layers, masks):
                pend = layer.forward_fused(pend, m, src_key_padding_mask, pos)
            s, y, _ = _resolve(pend, self.norm)
            x = s if self.norm is None else y
        else:
            for layer, m in zip(self.layers, masks):
                x = layer(x, src_mask=m, src_key_padding_mask=src_key_padding_mask, pos=pos)
            if self.norm is not None:
                x = self.norm(x)
        return xyz, io.back(x), None


class TransformerDecoder(nn.Module):
    """Stack of decoder layers with one shared output norm; ``return_intermediate`` stacks the normed output of
    every layer (the detector supervises all of them).  Returns ``(output(s), attention maps or [])``."""

    def __init__(self, decoder_layer, num_layers, norm_fn_name="ln", return_intermediate=False,
                 weight_init_name="xavier_uniform"):
        super().__init__()
        self.layers = get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        width = self.layers[0].linear2.out_features
        self.norm = None if norm_fn_name is None else NORM_DICT[norm_fn_name](width)
        self.return_intermediate = return_intermediate
        self._reset_parameters(weight_init_name)

    def _reset_parameters(self, weight_init_name):
        _xavier_matrices(self, weight_init_name)

    @_lib.on_tensor_device()
    def forward(self, tgt, memory, image_features_clip=None, text_features_clip=None,
                tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                query_pos: Optional[Tensor] = None, transpose_swap: Optional[bool] = False,
                return_attn_weights: Optional[bool] = False):
        io = _FeatureMapIO(transpose_swap, memory)
        memory, pos = io.seq(memory), io.seq(pos)
        if not return_attn_weights and isinstance(self.norm, nn.LayerNorm) \
                and TransformerDecoderLayer.fusable(self.layers, tgt):
            return self._forward_fused(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask,
                                       memory_key_padding_mask, pos, query_pos), []
        x, per_layer, maps = tgt, [], []
        for layer in self.layers:
            x, attn = layer(x, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                            tgt_key_padding_mask=tgt_key_padding_mask,
                            memory_key_padding_mask=memory_key_padding_mask, pos=pos, query_pos=query_pos,
                            return_attn_weights=return_attn_weights)
            maps.append(attn)
            if self.return_intermediate:
                # (the reference's intermediate outputs go through the norm, which it therefore requires here)
                per_layer.append(self.norm(x))
        last = x if self.norm is None else (per_layer[-1] if per_layer else self.norm(x))
        attns = torch.stack(maps) if return_attn_weights else []
        return (torch.stack(per_layer) if self.return_intermediate else last), attns

    def _uniform_layers(self):
        """All layers share dropout rates / head count / eps and have biases (the stack node reads
        them from the first layer)."""
        f = self.layers[0]
        key = lambda l: (l.dropout.p, l.dropout1.p, l.dropout2.p, l.dropout3.p, l.self_attn.num_heads,  # noqa: E731
                         l.self_attn.dropout, l.multihead_attn.dropout, l.norm1.eps, l.training)
        return all(key(l) == key(f) and l.linear1.bias is not None and l.linear2.bias is not None
                   and l.multihead_attn.num_heads == l.self_attn.num_heads for l in self.layers) \
            and f.self_attn.dropout == f.multihead_attn.dropout and self.norm.eps == f.norm1.eps

    def _forward_fused(self, tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask,
                       memory_key_padding_mask, pos, query_pos):
        """Same values as the loop in ``forward``; ``memory + pos`` (recomputed by every layer of
        the reference, transformer.py:566-569) is formed once."""
        nq, bsz, _ = tgt.shape
        nmem = memory.shape[0]
        nhead = self.layers[0].self_attn.num_heads
        memory = memory.contiguous()
        self_mask = _mask_u8(tgt_mask, tgt_key_padding_mask, bsz, nhead, nq, nq)
        cross_mask = _mask_u8(memory_mask, memory_key_padding_mask, bsz, nhead, nq, nmem)
        if _layer_nodes() and os.environ.get("CODA_DECODER_NODE", "stack") == "stack" and self._uniform_layers():
            # the whole decoder as one node, memory projections of all layers batched
            stacked = _fb.decoder_stack(self, tgt, memory, pos, query_pos, self_mask, cross_mask)
            return stacked if self.return_intermediate else stacked[-1]
        memory_pos = memory if pos is None else memory + pos
        pend = _Pending(tgt)
        intermediate = []
        for layer in self.layers:
            pend = layer.forward_fused(pend, memory, memory_pos, query_pos, self_mask, cross_mask)
            # materialise the layer output and apply the decoder norm to it in the same pass
            s, y, _ = _resolve(pend, self.norm)
            pend = _Pending(s)
            intermediate.append(y)
        if self.return_intermediate:
            return torch.stack(intermediate)
        return intermediate[-1]


class MaskedTransformerEncoder(TransformerEncoder):
    """Encoder whose layer ``i`` only attends within ``masking_radius[i]``; an
    optional set-abstraction module down-samples after layer 0
    (transformer.py:146-211).  NB the reference compares the UNSQUARED cdist
    against radii that build_encoder already squared (model_3detr.py:3976) --
    kept as is."""

    def __init__(self, encoder_layer, num_layers, masking_radius, interim_downsampling, norm=None,
                 weight_init_name="xavier_uniform"):
        super().__init__(encoder_layer, num_layers, norm=norm, weight_init_name=weight_init_name)
        assert len(masking_radius) == num_layers
        self.masking_radius = masking_radius
        self.interim_downsampling = interim_downsampling

    def compute_mask(self, xyz, radius, dist=None):
        """-> (bool (B, n, n) with True = farther than ``radius`` = not attended, pairwise distances for reuse)."""
        with torch.no_grad():
            if dist is None or dist.shape[1] != xyz.shape[1]:  # (the cloud was down-sampled since)
                dist = torch.cdist(xyz, xyz, p=2)
            return dist >= radius, dist

    @_lib.on_tensor_device()
    def forward(self, src, mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                xyz: Optional[Tensor] = None, transpose_swap: Optional[bool] = False):
        io = _FeatureMapIO(transpose_swap, src)
        x, pos = io.seq(src), io.seq(pos)
        dist, kept = None, None
        fused = TransformerEncoderLayer.fusable(self.layers, x)
        for depth, (layer, radius) in enumerate(zip(self.layers, self.masking_radius)):
            local = None
            if radius > 0:
                local, dist = self.compute_mask(xyz, radius, dist)
                local = _tile_mask_per_head(local, layer.nhead)
            if fused:
                x = _resolve(layer.forward_fused(_Pending(x), local, src_key_padding_mask, pos))[0]
            else:
                x = layer(x, src_mask=local, src_key_padding_mask=src_key_padding_mask, pos=pos)
            if depth == 0 and self.interim_downsampling:
                # the set-abstraction module works on (batch, channel, points)
                xyz, x, kept = self.interim_downsampling(xyz, x.permute(1, 2, 0))
                x = x.permute(2, 0, 1)
        if self.norm is not None:
            x = self.norm(x)
        return xyz, io.back(x), kept

    def extra_repr(self):
        radius_str = ", ".join(["%.2f" % (x) for x in self.masking_radius])
        return f"masking_radius={radius_str}"


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead=4, dim_feedforward=128, dropout=0.1, dropout_attn=None,
                 activation="relu", normalize_before=True, norm_name="ln", use_ffn=True,
                 ffn_use_bias=True):
        super().__init__()
        if dropout_attn is None:
            dropout_attn = dropout
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout_attn)
        self.use_ffn = use_ffn
        if self.use_ffn:
            self.linear1 = nn.Linear(d_model, dim_feedforward, bias=ffn_use_bias)
            self.dropout = nn.Dropout(dropout, inplace=False)
            self.linear2 = nn.Linear(dim_feedforward, d_model, bias=ffn_use_bias)
            self.norm2 = NORM_DICT[norm_name](d_model)
            self.dropout2 = nn.Dropout(dropout, inplace=False)
        self.norm1 = NORM_DICT[norm_name](d_model)
        self.dropout1 = nn.Dropout(dropout, inplace=False)
        self.activation = ACTIVATION_DICT[activation]()
        self.normalize_before = normalize_before
        self.nhead = nhead

    def with_pos_embed(self, tensor, pos: Optional[Tensor]):
        return _plus(tensor, pos)

    def _blocks(self, src, src_mask, src_key_padding_mask, pos, want_weights):
        """Self-attention (queries / keys carry the positional embedding, values do not) and the feed-forward
        block as two residual sub-layers; pre- or post-norm by ``normalize_before``.  In the post-norm order the
        attention block is only normed when ``use_norm_fn_on_input`` is set (the reference never sets it)."""
        maps = []

        def attend(t):
            qk = _plus(t, pos)
            out, w = self.self_attn(qk, qk, value=t, attn_mask=src_mask, key_padding_mask=src_key_padding_mask,
                                    need_weights=want_weights)
            maps.append(w)
            return out

        pre = self.normalize_before
        norm1 = self.norm1 if (pre or getattr(self, "use_norm_fn_on_input", False)) else None
        x = _sublayer(src, norm1, attend, self.dropout1, pre)
        if self.use_ffn:
            x = _sublayer(x, self.norm2, lambda t: _ffn(self, t), self.dropout2, pre)
        return x, maps[0]

    def forward_post(self, src, src_mask: Optional[Tensor] = None,
                     src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None):
        assert not self.normalize_before
        return self._blocks(src, src_mask, src_key_padding_mask, pos, False)[0]

    def forward_pre(self, src, src_mask: Optional[Tensor] = None,
                    src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                    return_attn_weights: Optional[Tensor] = False):
        assert self.normalize_before
        out, weights = self._blocks(src, src_mask, src_key_padding_mask, pos, bool(return_attn_weights))
        return (out, weights) if return_attn_weights else out

    @staticmethod
    def fusable(layers, x):
        return _fusable(layers, x, lambda l: [l.norm1] + ([l.norm2] if l.use_ffn else [])) and all(
            _attn_ok(l.self_attn) and (not l.use_ffn or _ffn_ok(l)) for l in layers)

    def forward_fused(self, pend, src_mask, src_key_padding_mask, pos):
        """``forward_pre`` on a pending residual stream; returns the new pending stream."""
        tgt_len, bsz, _ = pend.res.shape
        mask = _mask_u8(src_mask, src_key_padding_mask, bsz, self.nhead, tgt_len, tgt_len)
        if _layer_nodes():
            s, o = _fb.encoder_layer(self, pend, pos, mask)
            if not self.use_ffn:
                return _Pending(s, o, self.self_attn.out_proj.bias, _drop_p(self.dropout1))
            return _Pending(s, o, self.linear2.bias, _drop_p(self.dropout2))
        s, y, yp = _resolve(pend, self.norm1, pos)
        qk = y if pos is None else yp
        a = _fl.mha(self.self_attn, qk, qk, y, mask)
        pend = _Pending(s, a, self.self_attn.out_proj.bias, _drop_p(self.dropout1))
        if not self.use_ffn:
            return pend
        s, y, _ = _resolve(pend, self.norm2)
        return _Pending(s, _ffn_fused(self, y), self.linear2.bias, _drop_p(self.dropout2))

    @_lib.on_tensor_device()
    def forward(self, src, src_mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                return_attn_weights: Optional[Tensor] = False):
        out, weights = self._blocks(src, src_mask, src_key_padding_mask, pos,
                                    bool(return_attn_weights) and self.normalize_before)
        return (out, weights) if (return_attn_weights and self.normalize_before) else out

    def extra_repr(self):
        st = ""
        if hasattr(self.self_attn, "dropout"):
            st += f"attn_dr={self.self_attn.dropout}"
        return st


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead=4, dim_feedforward=256, dropout=0.1, dropout_attn=None,
                 activation="relu", normalize_before=True, norm_fn_name="ln"):
        super().__init__()
        if dropout_attn is None:
            dropout_attn = dropout
        # the reference passes `dropout`, not `dropout_attn`, to both (:506-507)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm1 = NORM_DICT[norm_fn_name](d_model)
        self.norm2 = NORM_DICT[norm_fn_name](d_model)
        self.norm3 = NORM_DICT[norm_fn_name](d_model)
        self.dropout1 = nn.Dropout(dropout, inplace=False)
        self.dropout2 = nn.Dropout(dropout, inplace=False)
        self.dropout3 = nn.Dropout(dropout, inplace=False)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout, inplace=False)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.activation = ACTIVATION_DICT[activation]()
        self.normalize_before = normalize_before

    def with_pos_embed(self, tensor, pos: Optional[Tensor]):
        return _plus(tensor, pos)

    def _blocks(self, tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, pos,
                query_pos, want_weights):
        """Self-attention over the queries, cross-attention into the encoder memory (keys carry ``pos``, queries
        ``query_pos``, values neither) and the feed-forward block: three residual sub-layers."""
        maps = []

        def self_attend(t):
            qk = _plus(t, query_pos)
            return self.self_attn(qk, qk, value=t, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask,
                                  need_weights=False)[0]

        def cross_attend(t):
            out, w = self.multihead_attn(query=_plus(t, query_pos), key=_plus(memory, pos), value=memory,
                                         attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask,
                                         need_weights=want_weights)
            maps.append(w)
            return out

        pre = self.normalize_before
        x = _sublayer(tgt, self.norm1, self_attend, self.dropout1, pre)
        x = _sublayer(x, self.norm2, cross_attend, self.dropout2, pre)
        x = _sublayer(x, self.norm3, lambda t: _ffn(self, t), self.dropout3, pre)
        return x, (maps[0] if want_weights else None)

    def forward_post(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                     memory_mask: Optional[Tensor] = None,
                     tgt_key_padding_mask: Optional[Tensor] = None,
                     memory_key_padding_mask: Optional[Tensor] = None,
                     pos: Optional[Tensor] = None, query_pos: Optional[Tensor] = None,
                     return_attn_weights: Optional[bool] = False):
        assert not self.normalize_before
        return self._blocks(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, pos,
                            query_pos, bool(return_attn_weights))

    def forward_pre(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                    memory_mask: Optional[Tensor] = None,
                    tgt_key_padding_mask: Optional[Tensor] = None,
                    memory_key_padding_mask: Optional[Tensor] = None,
                    pos: Optional[Tensor] = None, query_pos: Optional[Tensor] = None,
                    return_attn_weights: Optional[bool] = False):
        assert self.normalize_before
        return self._blocks(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, pos,
                            query_pos, bool(return_attn_weights))

    @staticmethod
    def fusable(layers, x):
        return _fusable(layers, x, lambda l: [l.norm1, l.norm2, l.norm3]) and all(
            _attn_ok(l.self_attn) and _attn_ok(l.multihead_attn) and _ffn_ok(l) for l in layers)

    def forward_fused(self, pend, memory, memory_pos, query_pos, self_mask, cross_mask):
        """``forward_pre`` on a pending residual stream; returns the new pending stream."""
        if _layer_nodes():
            s, o = _fb.decoder_layer(self, pend, memory, memory_pos, query_pos, self_mask, cross_mask)
            return _Pending(s, o, self.linear2.bias, _drop_p(self.dropout3))
        s, y, yp = _resolve(pend, self.norm1, query_pos)
        qk = y if query_pos is None else yp
        a = _fl.mha(self.self_attn, qk, qk, y, self_mask)
        s, y, yp = _fl.add_ln(a, norm=self.norm2, bias=self.self_attn.out_proj.bias, res=s, pos=query_pos,
                              p=_drop_p(self.dropout1))
        a = _fl.mha(self.multihead_attn, y if query_pos is None else yp, memory_pos, memory, cross_mask)
        s, y, _ = _fl.add_ln(a, norm=self.norm3, bias=self.multihead_attn.out_proj.bias, res=s,
                             p=_drop_p(self.dropout2))
        return _Pending(s, _ffn_fused(self, y), self.linear2.bias, _drop_p(self.dropout3))

    @_lib.on_tensor_device()
    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                query_pos: Optional[Tensor] = None, return_attn_weights: Optional[bool] = False):
        return self._blocks(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, pos,
                            query_pos, bool(return_attn_weights))
