"""Fused box decoder (``include/coda_box_ops.h``: coda_box_decode_fwd/bwd_f32).

One autograd node for everything ``get_box_predictions`` derives from the heads' raw outputs
(models/model_3detr.py:1683-1731): centres, sizes, angle, both corner sets, class / objectness
probabilities, for all decoder layers at once.  GPU fp32 only; the module-by-module torch path in
``model_3detr.py`` stays for everything else (CPU port, custom dataset configs) and as the parity
reference (``CODA_BOX_DECODE=torch``).
"""
import ctypes
import os

import torch

from . import _lib

_P = ctypes.c_void_p


def enabled():
    return os.environ.get("CODA_BOX_DECODE", "fused") != "torch"


def eligible(raws, query_xyz, dims, dataset_config):
    if not enabled() or not getattr(dataset_config, "standard_corner_builders", False):
        return False
    ts = list(raws) + [query_xyz, dims[0], dims[1]]
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts) and all(t.stride(-1) == 1 for t in raws)


def _strides(ts):
    arr = (ctypes.c_longlong * 15)()
    for i, t in enumerate(ts):
        arr[3 * i], arr[3 * i + 1], arr[3 * i + 2] = t.stride(0), t.stride(1), t.stride(2)
    return arr


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _BoxDecode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, center_raw, size_raw, angle_logits, angle_res_norm, cls_logits, query_xyz, dims_min, dims_max):
        nl, b, nq = center_raw.shape[:3]
        nbin, ncls1 = angle_logits.shape[-1], cls_logits.shape[-1]
        dev = center_raw.device
        query_xyz, dims_min, dims_max = query_xyz.contiguous(), dims_min.contiguous(), dims_max.contiguous()

        def new(*shape):
            return torch.empty((nl, b, nq) + shape, dtype=torch.float32, device=dev)

        outs = (new(3), new(3), new(3), new(3), new(nbin), new(), new(8, 3), new(8, 3), new(ncls1 - 1), new())
        strides = _strides([center_raw, size_raw, angle_logits, angle_res_norm, cls_logits])
        with torch.cuda.device(dev):
            st = _lib.load().coda_box_decode_fwd_f32(
                _ptr(center_raw), _ptr(size_raw), _ptr(angle_logits), _ptr(angle_res_norm), _ptr(cls_logits), strides,
                _ptr(query_xyz), _ptr(dims_min), _ptr(dims_max), nl, b, nq, nbin, ncls1, *[_ptr(o) for o in outs],
                _lib.current_stream_handle())
        _lib.check(st, "box_decode_fwd")
        ctx.save_for_backward(center_raw, size_raw, angle_logits, angle_res_norm, dims_min, dims_max)
        ctx.mark_non_differentiable(outs[8], outs[9])
        ctx.set_materialize_grads(False)  # unused outputs arrive as None (the kernel takes NULL), not as zero tensors
        return outs

    @staticmethod
    def backward(ctx, g_cn, g_cu, g_sn, g_su, g_ar, g_ang, g_cor, g_cxyz, _g_prob, _g_obj):
        center_raw, size_raw, angle_logits, angle_res_norm, dims_min, dims_max = ctx.saved_tensors
        nl, b, nq = center_raw.shape[:3]
        nbin = angle_logits.shape[-1]
        dev = center_raw.device
        grads = [g.contiguous() if g is not None else None for g in (g_cn, g_cu, g_sn, g_su, g_ar, g_ang, g_cor, g_cxyz)]
        d_c = torch.empty((nl, b, nq, 3), dtype=torch.float32, device=dev)
        d_s = torch.empty_like(d_c)
        d_a = torch.empty((nl, b, nq, nbin), dtype=torch.float32, device=dev)
        strides = _strides([center_raw, size_raw, angle_logits, angle_res_norm, angle_res_norm])
        with torch.cuda.device(dev):
            st = _lib.load().coda_box_decode_bwd_f32(
                _ptr(center_raw), _ptr(size_raw), _ptr(angle_logits), _ptr(angle_res_norm), strides, _ptr(dims_min),
                _ptr(dims_max), nl, b, nq, nbin, *[_ptr(g) for g in grads], _ptr(d_c), _ptr(d_s), _ptr(d_a),
                _lib.current_stream_handle())
        _lib.check(st, "box_decode_bwd")
        return d_c, d_s, None, d_a, None, None, None, None


def decode(center_raw, size_raw, angle_logits, angle_res_norm, cls_logits, query_xyz, dims):
    """(L,B,nq,C) raw head outputs (strided views are read in place) -> dict of the decoder's tensors."""
    (center_norm, center_unnorm, size_norm, size_unnorm, angle_residual, angle_cont, corners, corners_xyz, cls_prob,
     obj_prob) = _BoxDecode.apply(center_raw, size_raw, angle_logits, angle_res_norm, cls_logits, query_xyz, dims[0],
                                  dims[1])
    return {"center_normalized": center_norm, "center_unnormalized": center_unnorm, "size_normalized": size_norm,
            "size_unnormalized": size_unnorm, "angle_residual": angle_residual, "angle_continuous": angle_cont,
            "box_corners": corners, "box_corners_xyz": corners_xyz, "sem_cls_prob": cls_prob,
            "objectness_prob": obj_prob}
