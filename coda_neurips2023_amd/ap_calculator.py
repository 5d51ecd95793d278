"""Proposal filtering of the evaluation loop on the device.

The reference's ``utils/ap_calculator.py`` turns the model's outputs into per-scene detection lists in
``parse_predictions`` (:777-1018, called by ``APCalculator.step`` :1451-1489) and ``parse_predictions_obb``
(:45-286, the ``step_meter_show`` evaluate loops): for every proposal a scipy Delaunay triangulation decides
whether at least five scene points fall inside the box, then a numpy greedy NMS per scene, then the confidence
threshold.  Here the two filters are HIP kernels (csrc/eval_post.hip: ``coda_box_point_count_f32``,
``coda_nms_f32``) on the tensors where the model left them; one device->host copy of the survivors' rows builds
the same list-of-tuples the reference's ``APCalculator.accumulate`` / ``eval_det`` consume, in the same order.
Function names, arguments and the config dictionary are the reference's.
"""
import numpy as np
import torch

from . import _lib


def get_ap_config_dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=True,
                       per_class_proposal=True, use_cls_confidence_only=False, conf_thresh=0.05, no_nms=False,
                       dataset_config=None):
    """utils/ap_calculator.py:1021-1051: the evaluation settings (VoteNet's defaults)."""
    return {"remove_empty_box": remove_empty_box, "use_3d_nms": use_3d_nms, "nms_iou": nms_iou,
            "use_old_type_nms": use_old_type_nms, "cls_nms": cls_nms, "per_class_proposal": per_class_proposal,
            "use_cls_confidence_only": use_cls_confidence_only, "conf_thresh": conf_thresh, "no_nms": no_nms,
            "dataset_config": dataset_config}


def box_point_counts(predicted_boxes, point_cloud):
    """(B,K,8,3) upright-camera corners, (B,N,3+) depth-frame points -> (B,K) int32 points inside each box."""
    if not predicted_boxes.is_cuda:
        raise RuntimeError("CPU not supported")
    corners = predicted_boxes.detach().to(torch.float32).contiguous()
    pts = point_cloud.detach().to(device=corners.device, dtype=torch.float32).contiguous()
    b, k = corners.shape[:2]
    assert corners.shape[2:] == (8, 3) and pts.dim() == 3 and pts.shape[0] == b and pts.shape[2] >= 3
    counts = torch.empty((b, k), dtype=torch.int32, device=corners.device)
    with torch.cuda.device(corners.device):
        st = _lib.load().coda_box_point_count_f32(corners.data_ptr(), pts.data_ptr(), counts.data_ptr(), b, k,
                                                  pts.shape[1], pts.shape[2], _lib.current_stream_handle())
    _lib.check(st, "coda_box_point_count_f32")
    return counts


def nms_keep_mask(predicted_boxes, objectness_probs, pred_sem_cls, nonempty, config_dict):
    """(B,K) uint8: 1 for the boxes the configured NMS keeps among the ``nonempty`` candidates."""
    corners = predicted_boxes.detach().to(torch.float32).contiguous()
    b, k = corners.shape[:2]
    scores = objectness_probs.detach().to(torch.float32).contiguous()
    mode = 0 if not config_dict["use_3d_nms"] else (2 if config_dict["cls_nms"] else 1)
    cls = pred_sem_cls.to(torch.int32).contiguous() if mode == 2 else None
    ne = nonempty.to(torch.uint8).contiguous() if nonempty is not None else None
    keep = torch.empty((b, k), dtype=torch.uint8, device=corners.device)
    with torch.cuda.device(corners.device):
        st = _lib.load().coda_nms_f32(corners.data_ptr(), scores.data_ptr(), cls.data_ptr() if cls is not None else None,
                                      ne.data_ptr() if ne is not None else None, keep.data_ptr(), b, k, mode,
                                      float(config_dict["nms_iou"]), int(bool(config_dict["use_old_type_nms"])),
                                      _lib.current_stream_handle())
    _lib.check(st, "coda_nms_f32")
    return keep


def prediction_mask(predicted_boxes, sem_cls_probs, objectness_probs, point_cloud, config_dict):
    """(B,K) bool on the device: survives empty-box removal, NMS and the objectness threshold."""
    nonempty = None
    if config_dict["remove_empty_box"]:
        nonempty = box_point_counts(predicted_boxes, point_cloud) >= 5
        # a scene that loses every box keeps its most object-like one (:869-870) -- the NMS kernel applies the
        # same rule to an all-zero candidate row, so only the no_nms route needs it here
        if config_dict.get("no_nms"):
            none = ~nonempty.any(dim=1)
            fallback = torch.zeros_like(nonempty)
            fallback[torch.arange(nonempty.shape[0], device=nonempty.device), objectness_probs.argmax(dim=1)] = True
            nonempty = torch.where(none.unsqueeze(1), fallback, nonempty)
    if config_dict.get("no_nms"):
        mask = nonempty if nonempty is not None else torch.ones_like(objectness_probs, dtype=torch.bool)
    else:
        mask = nms_keep_mask(predicted_boxes, objectness_probs, sem_cls_probs.argmax(dim=-1), nonempty,
                             config_dict).bool()
    return mask & (objectness_probs > config_dict["conf_thresh"])


def _lists(mask, predicted_boxes, sem_cls_probs, objectness_probs, config_dict, obb=None):
    mask = mask.cpu().numpy()
    corners = predicted_boxes.detach().cpu().numpy()
    probs = sem_cls_probs.detach().cpu().numpy()
    obj = objectness_probs.detach().cpu().numpy()
    cls = np.argmax(probs, -1)
    out = []
    for i in range(corners.shape[0]):
        js = np.nonzero(mask[i])[0]
        if config_dict["per_class_proposal"]:
            assert config_dict["use_cls_confidence_only"] is False
            rows = [(ii, j, probs[i, j, ii] * obj[i, j])
                    for ii in range(config_dict["dataset_config"].num_semcls) for j in js]
        elif config_dict["use_cls_confidence_only"]:
            rows = [(cls[i, j].item(), j, probs[i, j, cls[i, j]]) for j in js]
        else:
            rows = [(cls[i, j].item(), j, obj[i, j]) for j in js]
        out.append([(c, corners[i, j], s) + (() if obb is None else (obb[i, j],)) for c, j, s in rows])
    return out


def parse_predictions(predicted_boxes, sem_cls_probs, objectness_probs, point_cloud, config_dict):
    """utils/ap_calculator.py:777-1018.  Returns, per scene, ``[(class, corners (8,3) ndarray, score), ...]``."""
    mask = prediction_mask(predicted_boxes, sem_cls_probs, objectness_probs, point_cloud, config_dict)
    return _lists(mask, predicted_boxes, sem_cls_probs, objectness_probs, config_dict)


def parse_predictions_obb(predicted_boxes, sem_cls_probs, objectness_probs, point_cloud, config_dict,
                          center_unnormalized, size_unnormalized, angle_continuous, reset_nms_iou=None):
    """utils/ap_calculator.py:45-286: as above, every tuple also carries the proposal's oriented-box row
    ``[centre(3), size(3), angle, class probabilities..., objectness]`` (a tensor on the outputs' device, :68).
    Zero-size boxes count as empty (:117-118)."""
    if reset_nms_iou is not None:
        config_dict["nms_iou"] = reset_nms_iou
    obb = torch.cat([center_unnormalized, size_unnormalized, angle_continuous.unsqueeze(-1), sem_cls_probs,
                     objectness_probs.unsqueeze(-1)], dim=-1)
    mask = prediction_mask(predicted_boxes, sem_cls_probs, objectness_probs, point_cloud, config_dict)
    return _lists(mask, predicted_boxes, sem_cls_probs, objectness_probs, config_dict, obb=obb)
