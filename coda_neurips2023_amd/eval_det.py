"""Average precision of the accumulated detections (second half).

Mirror of utils/eval_det.py (``voc_ap`` :23-54, ``eval_det_cls`` :66-162, ``eval_det`` :171-224): same inputs
({scan id: [(class, corners (8,3), score)]} / {scan id: [(class, corners)]}), same greedy marking in descending
confidence, same precision / recall / AP definitions.  What the reference spends its evaluation time on is the 3-D
IoU of every (detection, ground-truth box of that scan and class) pair -- ``utils/box_util.box3d_iou`` (:156-183): a
Python Sutherland-Hodgman clip + ``scipy.spatial.ConvexHull`` per pair.  Here, per class, ALL pairs are one launch
of the gIoU kernel's intersection pass (``coda_generalized_box3d_iou_f32`` with ``inter_vols_only = 2``: every
pair clipped, include/coda_box_ops.h) on scans padded to a common size; the marking loop then reads the matrix.  Passing
``get_iou_func`` (e.g. the reference's ``get_iou_obb``) selects the reference's per-pair host route instead."""
import numpy as np
import torch


def voc_ap(rec, prec, use_07_metric=False):
    """VOC average precision from a recall / precision curve (11-point rule or the area under the envelope)."""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):          # precision envelope
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]          # recall steps
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def box3d_vol(corners):
    """(..., 8, 3) -> product of the three edge lengths (utils/box_util.py:142-147)."""
    a = np.sqrt(np.sum((corners[..., 0, :] - corners[..., 1, :]) ** 2, -1))
    b = np.sqrt(np.sum((corners[..., 1, :] - corners[..., 2, :]) ** 2, -1))
    c = np.sqrt(np.sum((corners[..., 0, :] - corners[..., 4, :]) ** 2, -1))
    return a * b * c


def scan_ious(dets, gts, device=None):
    """dets: list over scans of (n_i, 8, 3) arrays, gts: list of (g_i, 8, 3) -> list of (n_i, g_i) float64 IoU
    matrices (box3d_iou's definition: ground-plane polygon intersection x height overlap over the union of the
    edge-product volumes), all scans in ONE device launch."""
    from . import box_util
    dev = torch.device(device if device is not None else "cuda")
    nscan = len(dets)
    k1 = max((d.shape[0] for d in dets), default=0)
    k2 = max((g.shape[0] for g in gts), default=0)
    out = [np.zeros((d.shape[0], g.shape[0])) for d, g in zip(dets, gts)]
    if nscan == 0 or k1 == 0 or k2 == 0:
        return out
    unit = np.zeros((8, 3), np.float32)
    c1 = np.broadcast_to(unit, (nscan, k1, 8, 3)).copy()
    c2 = np.broadcast_to(unit, (nscan, k2, 8, 3)).copy()
    for i, (d, g) in enumerate(zip(dets, gts)):
        c1[i, :d.shape[0]] = d
        c2[i, :g.shape[0]] = g
    nums = torch.tensor([g.shape[0] for g in gts], dtype=torch.int64, device=dev)
    inter = box_util.generalized_box3d_iou(torch.from_numpy(c1).to(dev), torch.from_numpy(c2).to(dev), nums,
                                           rotated_boxes=True, return_inter_vols_only="exact").double().cpu().numpy()
    v1, v2 = box3d_vol(c1.astype(np.float64)), box3d_vol(c2.astype(np.float64))
    for i, (d, g) in enumerate(zip(dets, gts)):
        n, m = d.shape[0], g.shape[0]
        iv = inter[i, :n, :m]
        with np.errstate(invalid="ignore", divide="ignore"):
            out[i] = iv / (v1[i, :n, None] + v2[i, None, :m] - iv)
    return out


def eval_det_cls(pred, gt, ovthresh=0.25, use_07_metric=False, get_iou_func=None, device=None):
    """One class.  pred {scan: [(corners, score)]}, gt {scan: [corners]} -> (recall, precision, ap)."""
    class_recs, npos = {}, 0
    for scan in gt.keys():
        bbox = np.array(gt[scan])
        class_recs[scan] = {"bbox": bbox, "det": [False] * len(bbox)}
        npos += len(bbox)
    for scan in pred.keys():
        if scan not in gt:
            class_recs[scan] = {"bbox": np.array([]), "det": []}
    scans, confidence, boxes, local = [], [], [], []
    for scan in pred.keys():
        for j, (box, score) in enumerate(pred[scan]):
            scans.append(scan)
            confidence.append(score)
            boxes.append(box)
            local.append(j)
    confidence = np.array(confidence)
    order = np.argsort(-confidence)
    ious = None
    if get_iou_func is None:  # every (detection, GT box of its scan) pair of this class in one launch
        keys = [s for s in pred.keys() if class_recs[s]["bbox"].size > 0 and len(pred[s]) > 0]
        mats = scan_ious([np.stack([np.asarray(b, np.float32) for b, _ in pred[s]]) for s in keys],
                         [class_recs[s]["bbox"].astype(np.float32) for s in keys], device)
        ious = dict(zip(keys, mats))
    nd = len(scans)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d, src in enumerate(order):
        rec = class_recs[scans[src]]
        gtb = rec["bbox"].astype(float)
        ovmax, jmax = -np.inf, -1
        if gtb.size > 0:
            if ious is not None:
                row = ious[scans[src]][local[src]]
            else:
                bb = np.asarray(boxes[src]).astype(float)
                row = [get_iou_func(bb, gtb[j, ...]) for j in range(gtb.shape[0])]
            for j in range(gtb.shape[0]):
                if row[j] > ovmax:       # first maximum, NaN never wins: utils/eval_det.py:141-143
                    ovmax, jmax = row[j], j
        if ovmax > ovthresh and not rec["det"][jmax]:
            tp[d] = 1.0
            rec["det"][jmax] = 1
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = np.zeros_like(tp) if npos == 0 else tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def eval_det(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, get_iou_func=None, device=None):
    """All classes.  pred_all {scan: [(class, corners, score)]}, gt_all {scan: [(class, corners)]} ->
    ({class: recall}, {class: precision}, {class: ap}) for the classes that have ground truth or detections."""
    pred, gt = {}, {}
    for scan in pred_all.keys():
        for classname, bbox, score in pred_all[scan]:
            pred.setdefault(classname, {}).setdefault(scan, []).append((bbox, score))
            gt.setdefault(classname, {}).setdefault(scan, [])
    for scan in gt_all.keys():
        for classname, bbox in gt_all[scan]:
            gt.setdefault(classname, {}).setdefault(scan, []).append(bbox)
    rec, prec, ap = {}, {}, {}
    for classname in list(gt.keys()):
        rec[classname], prec[classname], ap[classname] = eval_det_cls(
            pred.get(classname, {}), gt[classname], ovthresh, use_07_metric, get_iou_func, device)
    return rec, prec, ap
