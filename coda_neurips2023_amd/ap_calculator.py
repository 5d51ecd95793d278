"""Proposal filtering of the evaluation loop on the device.

The reference's ``utils/ap_calculator.py`` turns the model's outputs into per-scene detection lists in
``parse_predictions`` (:777-1018, called by ``APCalculator.step`` :1451-1489) and ``parse_predictions_obb``
(:45-286, the ``step_meter_show`` evaluate loops): for every proposal a scipy Delaunay triangulation decides
whether at least five scene points fall inside the box, then a numpy greedy NMS per scene, then the confidence
threshold.  Here the two filters are HIP kernels (csrc/eval_post.hip: ``coda_box_point_count_f32``,
``coda_nms_f32``) on the tensors where the model left them; one device->host copy of the survivors' rows builds
the same list-of-tuples the reference's ``APCalculator.accumulate`` / ``eval_det`` consume, in the same order.
Function names, arguments and the config dictionary are the reference's.

``APCalculator`` (:1054-1808: ``step_meter`` / ``step`` / ``accumulate`` / ``compute_metrics`` / ``metrics_to_str``)
accumulates those lists and turns them into the mAP / AR tables through ``eval_det`` (eval_det.py: every 3-D IoU of
a class in one launch).  ``merge_across_ranks`` is this package's alternative to gathering every rank's inputs
before ``step_meter`` (dist_utils.py).
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


class APCalculator(object):
    """utils/ap_calculator.py:1054-1808, the entry points engine.py's evaluate loops use."""

    def __init__(self, dataset_config, ap_iou_thresh=[0.25, 0.5], class2type_map=None, exact_eval=True, args=None,
                 ap_config_dict=None, reset_nms_iou=None):
        self.ap_iou_thresh = ap_iou_thresh
        if ap_config_dict is None:
            ap_config_dict = get_ap_config_dict(dataset_config=dataset_config, remove_empty_box=exact_eval)
        self.ap_config_dict = ap_config_dict
        self.class2type_map = class2type_map
        self.args = args
        self.dataset_config = dataset_config
        self.reset()
        self.reset_nms_iou = reset_nms_iou

    def reset(self):
        self.gt_map_cls = {}    # {scan id: [(class, corners)]}
        self.pred_map_cls = {}  # {scan id: [(class, corners, score)]}
        self.point_clouds = {}
        self.scan_cnt = 0

    def make_gt_list(self, gt_box_corners, gt_box_sem_cls_labels, gt_box_present):
        return [[(gt_box_sem_cls_labels[i, j].item(), gt_box_corners[i, j]) for j in range(gt_box_corners.shape[1])
                 if gt_box_present[i, j] == 1] for i in range(gt_box_corners.shape[0])]

    def step_meter(self, outputs, targets):
        if "outputs" in outputs:
            outputs = outputs["outputs"]
        self.step(predicted_box_corners=outputs["box_corners"], sem_cls_probs=outputs["sem_cls_prob"],
                  objectness_probs=outputs["objectness_prob"], point_cloud=targets["point_clouds"],
                  gt_box_corners=targets["gt_box_corners"], gt_box_sem_cls_labels=targets["gt_box_sem_cls_label"],
                  gt_box_present=targets["gt_box_present"])

    def step(self, predicted_box_corners, sem_cls_probs, objectness_probs, point_cloud, gt_box_corners,
             gt_box_sem_cls_labels, gt_box_present):
        batch_gt_map_cls = self.make_gt_list(gt_box_corners.cpu().detach().numpy(),
                                             gt_box_sem_cls_labels.cpu().detach().numpy(),
                                             gt_box_present.cpu().detach().numpy())
        batch_pred_map_cls = parse_predictions(predicted_box_corners, sem_cls_probs, objectness_probs, point_cloud,
                                               self.ap_config_dict)
        self.accumulate(batch_pred_map_cls, batch_gt_map_cls)

    def accumulate(self, batch_pred_map_cls, batch_gt_map_cls):
        assert len(batch_pred_map_cls) == len(batch_gt_map_cls)
        for pred, gt in zip(batch_pred_map_cls, batch_gt_map_cls):
            self.gt_map_cls[self.scan_cnt] = gt
            self.pred_map_cls[self.scan_cnt] = pred
            self.scan_cnt += 1

    def merge_across_ranks(self):
        """Every rank ends up with all ranks' accumulated scans (rank-major scan ids).  Call once before
        ``compute_metrics`` when each rank stepped on its own scenes only."""
        from . import dist_utils
        if not dist_utils.is_distributed():
            return
        import torch.distributed as dist
        mine = [(self.pred_map_cls[i], self.gt_map_cls[i]) for i in range(self.scan_cnt)]
        everyone = [None] * dist_utils.get_world_size()
        dist.all_gather_object(everyone, mine)
        self.reset()
        for part in everyone:
            for pred, gt in part:
                self.pred_map_cls[self.scan_cnt], self.gt_map_cls[self.scan_cnt] = pred, gt
                self.scan_cnt += 1

    def _groups(self, count):
        """Index sets of the frequent / common / base / novel class groups (:1577-1590)."""
        scannet = self.args is not None and getattr(self.args, "dataset_name", "").find("scannet") != -1
        if not scannet or count < 21:
            return slice(0, 4), slice(4, 10), slice(0, 10), slice(10, None)
        seen, novel = self.dataset_config.seen_idx_list, self.dataset_config.novel_idx_list
        return seen, seen, seen, novel

    def compute_metrics(self, get_iou_func=None):
        from collections import OrderedDict
        from .eval_det import eval_det
        overall_ret = OrderedDict()
        for ap_iou_thresh in self.ap_iou_thresh:
            ret = OrderedDict()
            rec, prec, ap = eval_det(self.pred_map_cls, self.gt_map_cls, ovthresh=ap_iou_thresh,
                                     get_iou_func=get_iou_func)
            name = (lambda key: self.class2type_map[key]) if self.class2type_map else str
            for key in sorted(ap.keys()):
                ret["%s Average Precision" % name(key)] = ap[key]
            ap_vals = np.array(list(ap.values()), dtype=np.float32)
            ap_vals[np.isnan(ap_vals)] = 0
            many = ap_vals.shape[0] > 2
            fre, common, base, novel = self._groups(ap_vals.shape[0])
            ret["mAP"] = ap_vals.mean()
            if many:
                ret["mAP_fre"], ret["mAP_common"] = ap_vals[fre].mean(), ap_vals[common].mean()
                ret["mAP_base"], ret["mAP_novel"] = ap_vals[base].mean(), ap_vals[novel].mean()
            prec_list, rec_list = [], []
            for key in sorted(prec.keys()):
                last = prec[key][-1] if len(prec[key]) else 0
                ret["%s Prec" % name(key)] = last
                prec_list.append(last)
            for key in sorted(ap.keys()):
                last = rec[key][-1] if len(rec[key]) else 0
                ret["%s Recall" % name(key)] = last
                rec_list.append(last)
            for label, values in (("Prec", np.array(prec_list)), ("AR", np.array(rec_list))):
                if many:
                    ret[label + "_fre"], ret[label + "_common"] = np.mean(values[fre]), np.mean(values[common])
                    ret[label + "_base"], ret[label + "_novel"] = np.mean(values[base]), np.mean(values[novel])
                ret[label] = np.mean(values)
            overall_ret[ap_iou_thresh] = ret
        return overall_ret

    def __str__(self):
        return self.metrics_to_str(self.compute_metrics())

    def metrics_to_str(self, overall_ret, per_class=True):
        blocks = {"mAP": [], "AR": [], "Prec": []}
        per_class_metrics = []
        for t in self.ap_iou_thresh:
            ret = overall_ret[t]
            for label in ("mAP", "AR", "Prec"):
                blocks[label].append(f"{label}{t:.2f}: {ret[label] * 100:.2f}\n")
                if label + "_fre" in ret:
                    for group in ("fre", "common", "base", "novel"):
                        tail = "\n\n" if group == "novel" else "\n"
                        blocks[label].append(f"{label}_{group}{t:.2f}: {ret[f'{label}_{group}'] * 100:.2f}{tail}")
            if per_class:
                per_class_metrics.append("-" * 5)
                per_class_metrics.append(f"IOU Thresh={t}")
                for x in ret.keys():
                    if x in ("mAP", "AR") or x.endswith(("fre", "common", "base", "novel")):
                        continue
                    per_class_metrics.append(f"{x}: {ret[x] * 100:.2f}")
        ap_str = "".join(blocks["mAP"]) + "\n" + "".join(blocks["AR"]) + "\n" + "".join(blocks["Prec"]) + "\n"
        if per_class:
            ap_str += "\n" + "\n".join(per_class_metrics)
        return ap_str

    def metrics_to_dict(self, overall_ret):
        out = {}
        for t in self.ap_iou_thresh:
            out[f"mAP_{t}"] = overall_ret[t]["mAP"] * 100
            out[f"AR_{t}"] = overall_ret[t]["AR"] * 100
        return out
