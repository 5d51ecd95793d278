"""Matched box losses of all decoder layers from one HIP pass per direction
(``include/coda_box_ops.h``: coda_box_loss_fwd/bwd_f32; criterion.py:219-246, 834-900, 1015-1104).

``layer_sums`` returns the per-layer sums (L, 5) of [sem-cls CE * has_object, angle-cls CE, angle-residual
Huber, centre L1, size L1] over the proposals (the last four over matched proposals); SetCriterion applies the
reference's normalisers and weights.  GPU fp32 only; the torch formulation in ``criterion.py`` covers
everything else and is the parity reference (``SetCriterion.fused_box_losses = False``)."""
import ctypes

import torch

from . import _lib

_KEYS = ("sem_cls_logits", "angle_logits", "angle_residual_normalized", "center_normalized", "size_normalized")


def eligible(outputs, targets, assignments):
    ts = [outputs[k] for k in _KEYS]
    if not all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.stride(-1) == 1 for t in ts):
        return False
    return assignments["per_prop_gt_inds"].dtype == torch.int64 and targets["gt_box_sem_cls_label"].dtype == torch.int64 \
        and targets["gt_angle_class_label"].dtype == torch.int64


def _strides(ts):
    arr = (ctypes.c_longlong * 15)()
    for i, t in enumerate(ts):
        arr[3 * i], arr[3 * i + 1], arr[3 * i + 2] = t.stride(0), t.stride(1), t.stride(2)
    return arr


class _BoxLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sem, ang, res, cen, siz, inds, matched, gt_sem, gt_ang, gt_res, gt_cen, gt_siz, has_obj, sem_w):
        nl, b, nq = sem.shape[:3]
        ngt = gt_sem.shape[1]
        aux = [t.contiguous() for t in (inds, matched, gt_sem, gt_ang, gt_res, gt_cen, gt_siz, has_obj, sem_w)]
        aux[1], aux[4], aux[7], aux[8] = (aux[i].to(torch.float32) for i in (1, 4, 7, 8))
        partial = torch.empty((nl, b * nq, 5), dtype=torch.float32, device=sem.device)
        ins = (sem, ang, res, cen, siz)
        with torch.cuda.device(sem.device):
            st = _lib.load().coda_box_loss_fwd_f32(*[t.data_ptr() for t in ins], _strides(ins),
                                                   *[t.data_ptr() for t in aux], nl, b, nq, ngt, sem.shape[-1],
                                                   ang.shape[-1], partial.data_ptr(), _lib.current_stream_handle())
        _lib.check(st, "box_loss_fwd")
        ctx.save_for_backward(*ins, *aux)
        return partial.sum(dim=1)

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        ins, aux = saved[:5], saved[5:]
        sem, ang = ins[0], ins[1]
        nl, b, nq = sem.shape[:3]
        ngt = aux[2].shape[1]
        g = g.contiguous().to(torch.float32)
        outs = [torch.empty((nl, b, nq, t.shape[-1]), dtype=torch.float32, device=sem.device) for t in ins]
        with torch.cuda.device(sem.device):
            st = _lib.load().coda_box_loss_bwd_f32(*[t.data_ptr() for t in ins], _strides(ins),
                                                   *[t.data_ptr() for t in aux], nl, b, nq, ngt, sem.shape[-1],
                                                   ang.shape[-1], g.data_ptr(), *[o.data_ptr() for o in outs],
                                                   _lib.current_stream_handle())
        _lib.check(st, "box_loss_bwd")
        return (*outs, None, None, None, None, None, None, None, None, None)


def layer_sums(outputs, targets, assignments, sem_class_weights, num_angle_bin):
    """-> (L, 5) per-layer sums; see the module docstring."""
    import math
    gt_res_norm = targets["gt_angle_residual_label"] / (math.pi / num_angle_bin)  # criterion.py:846-848
    has_object = (targets["gt_box_present"].sum(dim=1) != 0).to(torch.float32)
    return _BoxLoss.apply(*[outputs[k] for k in _KEYS], assignments["per_prop_gt_inds"],
                          assignments["proposal_matched_mask"], targets["gt_box_sem_cls_label"],
                          targets["gt_angle_class_label"], gt_res_norm, targets["gt_box_centers_normalized"],
                          targets["gt_box_sizes_normalized"], has_object, sem_class_weights)
