"""Set-prediction loss of CoDA: the LIVE loss terms.

Mirror of criterion.py for the terms that carry a non-zero weight in the shipped
CoDA scripts: the CLIP-space alignment losses on the hot path

* ``loss_predicted_region_embed_l1``                               (:924-943)
* ``loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi``     (:598-644)

and the box terms they are summed with (``loss_sem_cls_softmax_skip_none_gt_sample``
:219-246, ``loss_angle`` :834-900, ``loss_center`` :1015-1039, ``loss_size``
:1065-1104, logged ``loss_cardinality`` :169-179), the per-decoder-layer driver
``single_output_forward`` (:1106-1160) and ``forward`` (:1162-1216).  Method
names, dictionary keys and normalisers are the reference's.

The Hungarian ``Matcher`` (:12-86) is restated as is (host-side scipy, like the
reference).  The 3D gIoU that feeds its cost (utils/box_util.py:655-875; host-side
polygon clipping in the reference) is one HIP launch per call
(``box_util.generalized_box3d_iou``, ``include/coda_box_ops.h``).

``build_criterion(args, dataset_config)`` (:1219-1281) assembles matcher + weight dictionary
from the reference's argparse names; a non-zero weight for a term that is not one of the live
terms above raises ``NotImplementedError`` at construction instead of being dropped.
"""
import os

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def huber_loss(error, delta=1.0):
    """utils/misc.py huber: 0.5*q^2 + delta*(|e|-q), q = min(|e|, delta)."""
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


def all_reduce_average(tensor):
    """utils/dist.py:67-87: sum over ranks / world size (identity single-process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    t = tensor.clone()
    dist.all_reduce(t)
    return t / dist.get_world_size()


class Matcher(nn.Module):
    """Hungarian assignment of proposals to ground-truth boxes (criterion.py:12-86)."""

    def __init__(self, cost_class, cost_objectness, cost_giou, cost_center, solver=None):
        """``solver``: "auto" (default; env CODA_MATCHER overrides) solves on the device when the costs live there
        and the shape fits the kernel, "device" insists on it, "scipy" is the reference's host route."""
        super().__init__()
        self.solver = solver or os.environ.get("CODA_MATCHER", "auto")
        if self.solver not in ("auto", "device", "scipy"):
            raise ValueError(f"Matcher solver {self.solver!r}: expected auto, device or scipy")
        self.cost_class = cost_class
        self.cost_objectness = cost_objectness
        self.cost_giou = cost_giou
        self.cost_center = cost_center

    @torch.no_grad()
    @_lib.on_tensor_device(lambda outputs, targets: outputs["sem_cls_prob"])
    def forward(self, outputs, targets):
        pred_cls_prob = outputs["sem_cls_prob"]
        batchsize, nqueries = pred_cls_prob.shape[0], pred_cls_prob.shape[1]
        ngt = targets["gt_box_sem_cls_label"].shape[1]
        nactual_gt = targets["nactual_gt"]
        labels = targets["gt_box_sem_cls_label"].unsqueeze(1).expand(batchsize, nqueries, ngt)
        class_mat = -torch.gather(pred_cls_prob, 2, labels)
        objectness_mat = -outputs["objectness_prob"].unsqueeze(-1)
        center_mat = outputs["center_dist"].detach()
        giou_mat = -outputs["gious"].detach()
        final_cost = (self.cost_class * class_mat + self.cost_objectness * objectness_mat
                      + self.cost_center * center_mat + self.cost_giou * giou_mat).detach()
        return self.solve(final_cost, nactual_gt)

    @torch.no_grad()
    def solve(self, final_cost, nactual_gt):
        """Assignment for a ready cost tensor (nprob, nq, ngt): on the device when it lives there and fits the
        kernel, else the reference's host route."""
        if final_cost.is_cuda and self.solver != "scipy":
            solved = self._solve_on_device(final_cost.float().contiguous(), nactual_gt)
            if solved is not None:
                return solved
            if self.solver == "device":
                raise RuntimeError("Matcher(solver='device'): problem outside the device solver's limits "
                                   "(nq <= 1024, ngt <= 128, nq >= ngt)")
        return self._solve_on_host(final_cost, nactual_gt)

    def _solve_on_device(self, final_cost, nactual_gt):
        """All problems of the batch in one launch of the shortest-augmenting-path kernel (csrc/hungarian.hip);
        no host round trip.  None when the shape is outside the kernel's limits."""
        nprob, nq, ngt = final_cost.shape
        inds = torch.empty((nprob, nq), dtype=torch.int64, device=final_cost.device)
        mask = torch.empty((nprob, nq), dtype=torch.float32, device=final_cost.device)
        nact = nactual_gt.to(device=final_cost.device, dtype=torch.int64).contiguous()
        with torch.cuda.device(final_cost.device):
            st = _lib.load().coda_hungarian_f32(final_cost.data_ptr(), nact.data_ptr(), inds.data_ptr(),
                                                mask.data_ptr(), nprob, nq, ngt, _lib.current_stream_handle())
        if st == _lib.CODA_ENOSPC:
            return None
        _lib.check(st, "coda_hungarian_f32")
        return {"assignments": _LazyAssignments(inds, mask, nact), "per_prop_gt_inds": inds,
                "proposal_matched_mask": mask}

    @staticmethod
    def _solve_on_host(final_cost, nactual_gt):
        """The reference's route (criterion.py:67-79): cost matrix to the host, scipy per scene."""
        from scipy.optimize import linear_sum_assignment

        dev = final_cost.device
        batchsize, nqueries = final_cost.shape[0], final_cost.shape[1]
        final_cost = final_cost.cpu().numpy()
        per_prop_gt_inds = torch.zeros([batchsize, nqueries], dtype=torch.int64, device=dev)
        proposal_matched_mask = torch.zeros([batchsize, nqueries], dtype=torch.float32, device=dev)
        assignments = []
        for b in range(batchsize):
            assign = []
            if nactual_gt[b] > 0:
                assign = linear_sum_assignment(final_cost[b, :, :nactual_gt[b]])
                assign = [torch.from_numpy(x).long().to(device=dev) for x in assign]
                per_prop_gt_inds[b, assign[0]] = assign[1]
                proposal_matched_mask[b, assign[0]] = 1
            assignments.append(assign)
        return {"assignments": assignments, "per_prop_gt_inds": per_prop_gt_inds,
                "proposal_matched_mask": proposal_matched_mask}


class _LazyAssignments:
    """The reference's ``assignments`` entry -- per scene ``[proposal indices, GT indices]`` (criterion.py:70-79)
    -- derived from the device solver's outputs only when something indexes it (none of the live loss terms
    do): building the index lists needs the match count on the host, i.e. a synchronisation."""

    def __init__(self, inds, mask, nactual):
        self._inds, self._mask, self._nactual = inds, mask, nactual
        self._built = None

    def _build(self):
        if self._built is None:
            self._built = []
            for b in range(self._inds.shape[0]):
                rows = torch.nonzero(self._mask[b] > 0).flatten()
                self._built.append([rows, self._inds[b, rows]] if rows.numel() else [])
        return self._built

    def __len__(self):
        return self._inds.shape[0]

    def __getitem__(self, b):
        return self._build()[b]

    def __iter__(self):
        return iter(self._build())


# every key of the reference's loss_weight_dict (criterion.py:1247-1279) -> the argparse attribute
_WEIGHT_ARGS = {
    "loss_giou_weight": "loss_giou_weight",
    "loss_sem_cls_weight": "loss_sem_cls_weight",
    "loss_sem_cls_softmax_weight": "loss_sem_cls_softmax_weight",
    "loss_sem_cls_softmax_skip_none_gt_sample_weight": "loss_sem_cls_softmax_skip_none_gt_sample_weight",
    "loss_sem_cls_softmax_2d_box_iou_supervised_skip_none_gt_sample_weight":
        "loss_sem_cls_softmax_2d_box_iou_supervised_skip_none_gt_sample_weight",
    "loss_sem_cls_softmax_skip_none_gt_sample_en_discovery_objectness_weight":
        "loss_sem_cls_softmax_skip_none_gt_sample_en_discovery_objectness_weight",
    "loss_sem_cls_softmax_skip_none_gt_sample_keep_discovery_objectness_weight":
        "loss_sem_cls_softmax_skip_none_gt_sample_keep_discovery_objectness_weight",
    "loss_sem_cls_softmax_discovery_novel_objectness_weight": "loss_sem_cls_softmax_discovery_novel_objectness_weight",
    "loss_no_object_weight": "loss_no_object_weight",
    "loss_angle_cls_weight": "loss_angle_cls_weight",
    "loss_angle_reg_weight": "loss_angle_reg_weight",
    "loss_center_weight": "loss_center_weight",
    "loss_size_weight": "loss_size_weight",
    "loss_contrastive_weight": "loss_contrastive_weight",
    "loss_sem_focal_cls_weight": "loss_sem_focal_cls_weight",
    "loss_contrast_object_text_weight": "loss_contrast_object_text",  # (sic: no _weight in the flag, :1264)
    "loss_region_embed_weight": "loss_region_embed_weight",
    "loss_predicted_region_embed_l1_weight": "loss_predicted_region_embed_l1_weight",
    "loss_predicted_region_embed_l1_only_last_layer_weight": "loss_predicted_region_embed_l1_only_last_layer_weight",
    "loss_predicted_region_embed_cos_weight": "loss_predicted_region_embed_cos_weight",
    "loss_3d_2d_region_embed_weight": "loss_3d_2d_region_embed_weight",
    "loss_no_object_contrast_weight": "loss_no_object_contrast_weight",
    "loss_image_seen_class_weight": "loss_image_seen_class_weight",
    "loss_batchwise_contrastive_weight": "loss_batchwise_contrastive_weight",
    "loss_feat_seen_sigmoid_loss_weight": "loss_feat_seen_sigmoid_loss_weight",
    "loss_feat_seen_softmax_loss_weight": "loss_feat_seen_softmax_loss_weight",
    "loss_feat_seen_softmax_weakly_loss_weight": "loss_feat_seen_softmax_weakly_loss_weight",
    "loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi_weight":
        "loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi_weight",
    "loss_feat_seen_softmax_iou_match_weakly_loss_with_novel_cate_confi_weight":
        "loss_feat_seen_softmax_iou_match_weakly_loss_with_novel_cate_confi_weight",
    "loss_feat_seen_softmax_loss_with_novel_cate_confi_weight": "loss_feat_seen_softmax_loss_with_novel_cate_confi_weight",
    "loss_feat_seen_sigmoid_with_full_image_loss_weight": "loss_feat_seen_sigmoid_with_full_image_loss_weight",
    "loss_prompt_softmax_weight": "loss_prompt_softmax_weight",
    "loss_prompt_sigmoid_weight": "loss_prompt_sigmoid_weight",
}
# weights whose loss value comes out of a term under another name (loss_angle -> two values)
_PRODUCED_BY = {"loss_angle_cls": "loss_angle", "loss_angle_reg": "loss_angle"}


class SetCriterion(nn.Module):
    """criterion.py:88-168.  Same positional arguments as the reference's constructor."""

    def __init__(self, matcher, dataset_config, loss_weight_dict, train_range_max=37, only_image_class=False,
                 only_prompt_loss=False, args=None, confidence_type="clip-max-prob", giou_fn="default"):
        super().__init__()
        if only_image_class or only_prompt_loss:
            raise NotImplementedError("only_image_class / only_prompt_loss criteria (criterion.py:1163-1179) are not "
                                      "part of the CoDA training recipes and are outside the hot path")
        self.dataset_config = dataset_config
        self.matcher = matcher
        if giou_fn == "default":
            from .box_util import generalized_box3d_iou as giou_fn
        self.giou_fn = giou_fn
        loss_weight_dict = dict(loss_weight_dict)
        semcls_percls_weights = torch.ones(dataset_config.num_semcls + 1)
        semcls_percls_weights[-1] = loss_weight_dict.pop("loss_no_object_weight", 1.0)
        self.register_buffer("semcls_percls_weights", semcls_percls_weights)
        seen = torch.ones(train_range_max + 1)
        seen[-1] = loss_weight_dict.pop("loss_no_object_contrast_weight", 1.0)
        self.register_buffer("seen_semcls_percls_weights", seen)
        self.loss_weight_dict = loss_weight_dict
        self.confidence_type = getattr(args, "confidence_type", confidence_type) if args else confidence_type
        # read like the reference (:104,128-133); they only steer terms outside the live set
        self.if_skip_no_seen_scene_objectness = getattr(args, "if_skip_no_seen_scene_objectness", False)
        self.if_only_seen_in_loss = getattr(args, "if_only_seen_in_loss", False)
        self.confidence_type_in_datalayer = getattr(args, "confidence_type_in_datalayer", "clip-max-prob")
        self.layer_batched = True  # evaluate all decoder layers in one pass when the model hands them stacked
        self.fused_alignment = True  # GPU fp32: both alignment terms from one HIP pass (align_loss.py)
        self.fused_box_losses = True  # GPU fp32: the four matched box terms from one HIP pass (box_loss.py)
        # GPU: keep num_boxes / num_boxes_replica / the rotated-GT flag as device scalars instead of the reference's
        # three .item() read-backs (criterion.py:1147,1164-1170), so the host keeps enqueueing through the criterion
        self.device_scalars = True
        self.fused_matcher_cost = True  # GPU fp32: gIoU + L1 centre distance + weighted cost matrix in one launch
        assert self.confidence_type in ["non-confidence", "objectness", "clip+objectness", "clip-max-prob"]
        self.loss_functions = {
            "loss_sem_cls_softmax_skip_none_gt_sample": self.loss_sem_cls_softmax_skip_none_gt_sample,
            "loss_angle": self.loss_angle,
            "loss_center": self.loss_center,
            "loss_size": self.loss_size,
            "loss_cardinality": self.loss_cardinality,  # logged only, no weight
            "loss_predicted_region_embed_l1": self.loss_predicted_region_embed_l1,
            "loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi":
                self.loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi,
        }
        missing = [k for k, w in loss_weight_dict.items() if w > 1e-32
                   and _PRODUCED_BY.get(k[:-len("_weight")], k[:-len("_weight")]) not in self.loss_functions]
        if missing:
            raise NotImplementedError(
                "non-zero weights for loss terms outside the CoDA recipes' live set: " + ", ".join(sorted(missing)))
        if matcher is not None and getattr(matcher, "cost_giou", 0) > 0 and self.giou_fn is None:
            raise ValueError("matcher.cost_giou > 0 needs a gIoU function (giou_fn=None was passed)")

    def _live(self, name):
        """Is term `name` evaluated?  (criterion.py:1130-1137: weight > 1e-32, or no weight key at all)"""
        key = name + "_weight"
        return key not in self.loss_weight_dict or self.loss_weight_dict[key] > 1e-32

    # ---- loss terms ----------------------------------------------------------------
    # Every term is written once for tensors with a leading decoder-layer axis,
    # outputs (L,B,nq,...) / assignments (L,B,nq) / targets (B,...), and returns one value per
    # layer, shape (L,).  The reference evaluates the terms layer by layer
    # (criterion.py:1205-1215, ~60 small kernels per layer and term group); the public
    # single-layer methods below keep its signatures and call the same code with L = 1.
    @staticmethod
    def _lift(d, keys):
        return {k: (v.unsqueeze(0) if torch.is_tensor(v) and k in keys else v) for k, v in d.items()}

    _OUT_KEYS = ("sem_cls_logits", "angle_logits", "angle_residual_normalized", "center_dist", "size_normalized",
                 "text_correlation_embedding")
    _ASSIGN_KEYS = ("per_prop_gt_inds", "proposal_matched_mask")

    def _single(self, fn, outputs, targets, assignments):
        res = fn(self._lift(outputs, self._OUT_KEYS), targets, self._lift(assignments, self._ASSIGN_KEYS))
        return {k: v[0] for k, v in res.items()}

    @staticmethod
    def _gather_gt(gt, inds):
        """gt (B,ngt[,k]) indexed by inds (L,B,nq) -> (L,B,nq[,k])."""
        nl = inds.shape[0]
        gt = gt.unsqueeze(0).expand(nl, *gt.shape)
        if gt.dim() == 4:
            return torch.gather(gt, 2, inds.unsqueeze(-1).expand(-1, -1, -1, gt.shape[-1]))
        return torch.gather(gt, 2, inds)

    @staticmethod
    def _ce(logits, labels, weight=None):
        """Per-element cross entropy over the last axis: logits (L,B,nq,K), labels (L,B,nq)."""
        k = logits.shape[-1]
        return F.cross_entropy(logits.reshape(-1, k), labels.reshape(-1), weight, reduction="none").view(labels.shape)

    @torch.no_grad()
    def stacked_loss_cardinality(self, outputs, targets, assignments):
        pred_logits = outputs["sem_cls_logits"]
        pred_objects = (pred_logits.argmax(-1) != pred_logits.shape[-1] - 1).sum(-1)
        card_err = (pred_objects.float() - targets["nactual_gt"]).abs().mean(-1)
        return {"loss_cardinality": card_err}

    def _fused_box_terms(self, outputs, targets, assignments):
        """{loss name: per-layer value} of the four matched box terms from the fused pass (box_loss.py), with the
        reference's normalisers, or None when that pass does not apply."""
        from . import box_loss
        live = ["loss_sem_cls_softmax_skip_none_gt_sample", "loss_angle", "loss_center", "loss_size"]
        if not (self.fused_box_losses and all(self._live(k) for k in live)
                and all(k in outputs for k in box_loss._KEYS) and box_loss.eligible(outputs, targets, assignments)):
            return None
        sums = box_loss.layer_sums(outputs, targets, assignments, self.semcls_percls_weights,
                                   self.dataset_config.num_angle_bin)
        has_object = (targets["gt_box_present"].sum(dim=1) != 0).to(sums.dtype)
        nq = outputs["sem_cls_logits"].shape[2]
        res = {"loss_sem_cls_softmax_skip_none_gt_sample": sums[:, 0] / (has_object.sum() * nq + 1e-32)}
        if torch.is_tensor(targets["num_boxes_replica"]):
            # device scalars: num_boxes is clamped to >= 1 and a replica without boxes has an all-zero matched mask,
            # i.e. zero sums -- the same zeros the branch below produces, without reading the count back.  One
            # division for the five columns (a column select per term costs four launches in its backward).
            nb = targets["num_boxes"].to(sums.dtype).reshape(())
            den = torch.stack((has_object.sum() * nq + 1e-32, nb, nb, nb, nb))
            cols = (sums / den).unbind(1)
            return dict(zip(("loss_sem_cls_softmax_skip_none_gt_sample", "loss_angle_cls", "loss_angle_reg", "loss_center",
                             "loss_size"), cols))
        elif targets["num_boxes_replica"] > 0:
            nb = targets["num_boxes"]
            res.update(loss_angle_cls=sums[:, 1] / nb, loss_angle_reg=sums[:, 2] / nb,
                       loss_center=sums[:, 3] / nb if nb > 0 else sums[:, 3], loss_size=sums[:, 4] / nb)
        else:  # no box on this worker: zero terms that keep the heads in the graph (:894-897, 1035-1037, 1100-1102)
            res.update(loss_angle_cls=sums[:, 1] * 0, loss_angle_reg=sums[:, 2] * 0, loss_center=sums[:, 3] * 0,
                       loss_size=sums[:, 4] * 0)
        return res

    def stacked_loss_sem_cls_softmax_skip_none_gt_sample(self, outputs, targets, assignments):
        if outputs.get("_fused_box_terms") is not None:
            return {"loss_sem_cls_softmax_skip_none_gt_sample":
                    outputs["_fused_box_terms"]["loss_sem_cls_softmax_skip_none_gt_sample"]}
        pred_logits = outputs["sem_cls_logits"]
        gt_box_label = self._gather_gt(targets["gt_box_sem_cls_label"], assignments["per_prop_gt_inds"])
        gt_box_label = torch.where(assignments["proposal_matched_mask"].int() == 0,
                                   torch.full_like(gt_box_label, pred_logits.shape[-1] - 1), gt_box_label)
        loss = self._ce(pred_logits, gt_box_label, self.semcls_percls_weights)
        # scenes without any GT box contribute 0 and are not counted (:236-244)
        has_object = (targets["gt_box_present"].sum(dim=1) != 0).to(loss.dtype)
        final_loss = (loss.sum(dim=2) * has_object).sum(dim=1)
        final_loss = final_loss / (has_object.sum() * loss.shape[2] + 1e-32)
        return {"loss_sem_cls_softmax_skip_none_gt_sample": final_loss}

    def stacked_loss_angle(self, outputs, targets, assignments):
        if outputs.get("_fused_box_terms") is not None:
            f = outputs["_fused_box_terms"]
            return {"loss_angle_cls": f["loss_angle_cls"], "loss_angle_reg": f["loss_angle_reg"]}
        angle_logits = outputs["angle_logits"]
        angle_residual = outputs["angle_residual_normalized"]
        if targets["num_boxes_replica"] > 0:
            inds, matched = assignments["per_prop_gt_inds"], assignments["proposal_matched_mask"]
            gt_angle_residual_normalized = targets["gt_angle_residual_label"] / (
                np.pi / self.dataset_config.num_angle_bin)
            gt_angle_label = self._gather_gt(targets["gt_angle_class_label"], inds)
            angle_cls_loss = (self._ce(angle_logits, gt_angle_label) * matched).sum(dim=(1, 2))
            gt_angle_residual_normalized = self._gather_gt(gt_angle_residual_normalized, inds)
            # residual of the GT angle bin (the reference multiplies by a one-hot and sums, :869-876)
            angle_residual_for_gt_class = torch.gather(angle_residual, 3, gt_angle_label.unsqueeze(-1)).squeeze(-1)
            angle_reg_loss = huber_loss(angle_residual_for_gt_class - gt_angle_residual_normalized, delta=1.0)
            angle_reg_loss = (angle_reg_loss * matched).sum(dim=(1, 2))
            angle_cls_loss = angle_cls_loss / targets["num_boxes"]
            angle_reg_loss = angle_reg_loss / targets["num_boxes"]
        else:
            angle_cls_loss = angle_logits.sum(dim=(1, 2, 3)) * 0
            angle_reg_loss = angle_residual.sum(dim=(1, 2, 3)) * 0
        return {"loss_angle_cls": angle_cls_loss, "loss_angle_reg": angle_reg_loss}

    def stacked_loss_center(self, outputs, targets, assignments):
        if outputs.get("_fused_box_terms") is not None:
            return {"loss_center": outputs["_fused_box_terms"]["loss_center"]}
        center_dist = outputs["center_dist"]
        if targets["num_boxes_replica"] > 0:
            center_loss = torch.gather(center_dist, 3, assignments["per_prop_gt_inds"].unsqueeze(-1)).squeeze(-1)
            center_loss = (center_loss * assignments["proposal_matched_mask"]).sum(dim=(1, 2))
            if targets["num_boxes"] > 0:
                center_loss = center_loss / targets["num_boxes"]
        else:
            center_loss = center_dist.sum(dim=(1, 2, 3)) * 0
        return {"loss_center": center_loss}

    def stacked_loss_size(self, outputs, targets, assignments):
        if outputs.get("_fused_box_terms") is not None:
            return {"loss_size": outputs["_fused_box_terms"]["loss_size"]}
        pred_box_sizes = outputs["size_normalized"]
        if targets["num_boxes_replica"] > 0:
            gt_box_sizes = self._gather_gt(targets["gt_box_sizes_normalized"], assignments["per_prop_gt_inds"])
            size_loss = (pred_box_sizes - gt_box_sizes).abs().sum(dim=-1)
            size_loss = (size_loss * assignments["proposal_matched_mask"]).sum(dim=(1, 2))
            size_loss = size_loss / targets["num_boxes"]
        else:
            size_loss = pred_box_sizes.sum(dim=(1, 2, 3)) * 0
        return {"loss_size": size_loss}

    # CLIP-space alignment terms (hot path)
    def _alignment_labels(self, targets, assignments):
        """Label / confidence of every proposal for the CE alignment term: matched proposals take
        the GT box's, the others the CLIP weak label (criterion.py:618-631)."""
        inds = assignments["per_prop_gt_inds"]
        matched = assignments["proposal_matched_mask"].int() > 0
        seen_label = self._gather_gt(targets["gt_box_seen_sem_cls_label"], inds)
        seen_confi = self._gather_gt(targets["gt_box_seen_sem_cls_confi"], inds)
        gt_box_label = torch.where(matched, seen_label, targets["weak_box_cate_label"])
        gt_box_confidence = torch.where(matched, seen_confi, targets["weak_confidence_weight"])
        if self.confidence_type == "non-confidence":
            gt_box_confidence = torch.where(gt_box_confidence > 1e-16, torch.ones_like(gt_box_confidence),
                                            gt_box_confidence)
        return gt_box_label, gt_box_confidence

    def _fused_alignment(self, outputs, targets, assignments):
        """Both alignment terms of all layers from the fused HIP pass (align_loss.py) as
        (l1 per layer, ce per layer), or None when that pass does not apply: tensors other than
        GPU fp32, or a recipe in which only one of the two terms is live (stage 1 of the shipped
        scripts: L1 only, and its datasets carry no `gt_box_seen_sem_cls_confi` for the CE labels)."""
        from . import align_loss
        if not (self.fused_alignment and self._live("loss_predicted_region_embed_l1")
                and self._live("loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi")):
            return None
        emb = outputs["text_correlation_embedding"]
        gt = targets["gt_text_correlation_embedding"]
        text = targets["text_features_clip"]
        if not align_loss.eligible(emb, gt, text, targets["logit_scale"]):
            return None
        label, conf = self._alignment_labels(targets, assignments)
        weight_maps = targets["gt_text_correlation_embedding_mask"]
        l1_sum, ce_sum = align_loss.align_loss_sums(emb, gt, weight_maps, text, targets["logit_scale"], label, conf)
        ave_weight = torch.sum(weight_maps) * emb.shape[-1]
        all_num = torch.sum(conf > 1e-32, dim=(1, 2)) + 1e-32
        return l1_sum / ave_weight, ce_sum / all_num

    def stacked_loss_predicted_region_embed_l1(self, outputs, targets, assignments):
        """Masked L1 between the predicted region embedding and the CLIP image
        embedding of the cropped box, / (sum(mask) * 512)."""
        fused = outputs["_fused_alignment"] if "_fused_alignment" in outputs \
            else self._fused_alignment(outputs, targets, assignments)  # the drivers evaluate it once for both terms
        if fused is not None:
            return {"loss_predicted_region_embed_l1": fused[0]}
        gt = targets["gt_text_correlation_embedding"]
        pred = outputs["text_correlation_embedding"]
        weight_maps = targets["gt_text_correlation_embedding_mask"]
        ave_weight = torch.sum(weight_maps) * pred.shape[-1]
        l1_loss = (pred * weight_maps - gt * weight_maps).abs().sum(dim=(1, 2, 3)) / ave_weight
        return {"loss_predicted_region_embed_l1": l1_loss}

    def stacked_loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi(self, outputs, targets, assignments):
        """CE(normalised embedding @ text^T * scale, label) * confidence, where matched
        proposals take the GT label/confidence and the others the CLIP weak label."""
        fused = outputs["_fused_alignment"] if "_fused_alignment" in outputs \
            else self._fused_alignment(outputs, targets, assignments)
        if fused is not None:
            return {"loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi": fused[1]}
        emb = outputs["text_correlation_embedding"]
        emb = emb / (emb.norm(dim=-1, keepdim=True) + 1e-32)
        text_features_clip = targets["text_features_clip"].to(emb.dtype)  # (fp16 CLIP output -> the embedding's fp32)
        temperature_param = targets["logit_scale"]
        correlation_map = torch.matmul(emb, text_features_clip.permute(0, 2, 1)) * temperature_param
        gt_box_label, gt_box_confidence = self._alignment_labels(targets, assignments)
        loss = self._ce(correlation_map, gt_box_label)
        all_num = torch.sum(gt_box_confidence > 1e-32, dim=(1, 2)) + 1e-32
        final_loss = torch.sum(loss * gt_box_confidence, dim=(1, 2)) / all_num
        return {"loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi": final_loss}

    # single-layer entry points with the reference's names and signatures
    def loss_cardinality(self, outputs, targets, assignments):                       # :169-179
        return self._single(self.stacked_loss_cardinality, outputs, targets, assignments)

    def loss_sem_cls_softmax_skip_none_gt_sample(self, outputs, targets, assignments):  # :219-246
        return self._single(self.stacked_loss_sem_cls_softmax_skip_none_gt_sample, outputs, targets, assignments)

    def loss_angle(self, outputs, targets, assignments):                             # :834-900
        return self._single(self.stacked_loss_angle, outputs, targets, assignments)

    def loss_center(self, outputs, targets, assignments):                            # :1015-1039
        return self._single(self.stacked_loss_center, outputs, targets, assignments)

    def loss_size(self, outputs, targets, assignments):                              # :1065-1104
        return self._single(self.stacked_loss_size, outputs, targets, assignments)

    def loss_predicted_region_embed_l1(self, outputs, targets, assignments):         # :924-943
        return self._single(self.stacked_loss_predicted_region_embed_l1, outputs, targets, assignments)

    def loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi(self, outputs, targets, assignments):  # :598-644
        return self._single(self.stacked_loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi, outputs, targets,
                            assignments)

    # ---- drivers -------------------------------------------------------------------
    def single_output_forward(self, outputs, targets, if_region_embed=False, if_aux=False,
                              if_last_head=False):
        if self.giou_fn is not None and "gt_box_corners" in targets:
            gious = self.giou_fn(outputs["box_corners"], targets["gt_box_corners"], targets["nactual_gt"],
                                 rotated_boxes=self._rotated_flag(targets), needs_grad=False)
        else:  # giou_fn=None was asked for (matcher.cost_giou == 0): zero cost term
            gious = torch.zeros(outputs["center_normalized"].shape[0], outputs["center_normalized"].shape[1],
                                targets["gt_box_centers_normalized"].shape[1],
                                device=outputs["center_normalized"].device)
        outputs["gious"] = gious
        center_dist = torch.cdist(outputs["center_normalized"], targets["gt_box_centers_normalized"], p=1)
        outputs["center_dist"] = center_dist
        assignments = self.matcher(outputs, targets)
        if "text_correlation_embedding" in outputs and "gt_text_correlation_embedding" in targets:
            lifted = (self._lift(outputs, self._OUT_KEYS), self._lift(assignments, self._ASSIGN_KEYS))
            fused = self._fused_alignment(lifted[0], targets, lifted[1])
            outputs["_fused_alignment"] = None if fused is None else tuple(t for t in fused)

        losses = {}
        for k, fn in self.loss_functions.items():
            if self._live(k):
                losses.update(fn(outputs, targets, assignments))
        outputs.pop("_fused_alignment", None)

        final_loss = 0
        for k, w in self.loss_weight_dict.items():
            if w > 1e-32:
                name = k.replace("_weight", "")
                losses[name] *= w
                final_loss += losses[name]
        return final_loss, losses

    def _fused_cost_applies(self, stacked, targets):
        from . import box_util
        c = stacked["center_normalized"]
        return (self.fused_matcher_cost and c.is_cuda and c.dtype == torch.float32 and type(self.matcher) is Matcher
                and self.giou_fn is box_util.generalized_box3d_iou and "gt_box_corners" in targets
                and "box_corners" in stacked)

    def _rotated_flag(self, targets):
        """criterion.py:1147: are any GT boxes rotated?  A device flag for this package's gIoU kernel when
        ``device_scalars`` is on, the reference's host bool otherwise (any other ``giou_fn``)."""
        from . import box_util
        flag = torch.any(targets["gt_box_angles"] > 0)
        if self.device_scalars and flag.is_cuda and self.giou_fn is box_util.generalized_box3d_iou:
            return flag
        return flag.item()

    def stacked_forward(self, stacked, targets):
        """Matching + every live loss term for all decoder layers at once.  ``stacked[k]`` is
        (L,B,nq,...) with the last decoder layer at index L-1 (what ``outputs`` /
        ``aux_outputs`` hold layer by layer).  Layers are extra, independent scenes for the
        matcher (one cost matrix, one host round trip instead of L)."""
        center = stacked["center_normalized"]
        nl, bsz, nq = center.shape[:3]
        ngt = targets["gt_box_centers_normalized"].shape[1]
        gt_centers = targets["gt_box_centers_normalized"]
        flat_tgt = {"gt_box_sem_cls_label": targets["gt_box_sem_cls_label"].repeat(nl, 1),
                    "nactual_gt": targets["nactual_gt"].repeat(nl)}
        center_dist = l1_dist = None
        if self._fused_cost_applies(stacked, targets):
            # gIoU, L1 centre distance and the weighted cost matrix of all layers from one launch; the distances
            # carry no gradient (the matched centre term below has its own pass, or recomputes them)
            from . import box_util
            m = self.matcher
            gious, l1_dist, cost = box_util.matcher_cost(
                stacked["box_corners"].flatten(0, 1), targets["gt_box_corners"].repeat(nl, 1, 1, 1),
                flat_tgt["nactual_gt"], center.flatten(0, 1), gt_centers.repeat(nl, 1, 1),
                stacked["sem_cls_prob"].flatten(0, 1), flat_tgt["gt_box_sem_cls_label"],
                stacked["objectness_prob"].flatten(0, 1),
                (m.cost_class, m.cost_objectness, m.cost_center, m.cost_giou), self._rotated_flag(targets))
            gious = gious.view(nl, bsz, nq, ngt)
            flat_assign = m.solve(cost, flat_tgt["nactual_gt"])
        else:
            if self.giou_fn is not None and "gt_box_corners" in targets:
                # all decoder layers are extra scenes of ONE launch (the reference: one host loop per layer)
                rotated = self._rotated_flag(targets)
                gious = self.giou_fn(stacked["box_corners"].flatten(0, 1), targets["gt_box_corners"].repeat(nl, 1, 1, 1),
                                     targets["nactual_gt"].repeat(nl), rotated_boxes=rotated,
                                     needs_grad=False).view(nl, bsz, nq, ngt)
            else:  # giou_fn=None was asked for (matcher.cost_giou == 0): zero cost term
                gious = torch.zeros(nl, bsz, nq, ngt, device=center.device)
            center_dist = torch.cdist(center.reshape(nl * bsz, nq, -1), gt_centers.repeat(nl, 1, 1), p=1)  # matcher + loss_center
            flat_out = {"sem_cls_prob": stacked["sem_cls_prob"].flatten(0, 1),
                        "objectness_prob": stacked["objectness_prob"].flatten(0, 1),
                        "center_dist": center_dist, "gious": gious.flatten(0, 1)}
            flat_assign = self.matcher(flat_out, flat_tgt)
        assignments = {"per_prop_gt_inds": flat_assign["per_prop_gt_inds"].view(nl, bsz, nq),
                       "proposal_matched_mask": flat_assign["proposal_matched_mask"].view(nl, bsz, nq)}
        outs = dict(stacked, gious=gious)
        outs["_fused_box_terms"] = self._fused_box_terms(outs, targets, assignments)
        if center_dist is None and outs["_fused_box_terms"] is None:  # the torch centre term differentiates these
            center_dist = torch.cdist(center.reshape(nl * bsz, nq, -1), gt_centers.repeat(nl, 1, 1), p=1)
        outs["center_dist"] = (l1_dist if center_dist is None else center_dist).view(nl, bsz, nq, ngt)
        outs["_fused_alignment"] = self._fused_alignment(outs, targets, assignments) \
            if "gt_text_correlation_embedding" in targets else None

        losses = {}
        for k in self.loss_functions:
            if self._live(k):
                losses.update(getattr(self, "stacked_" + k)(outs, targets, assignments))
        # weighted sum of the live terms (criterion.py:1138-1144, per layer): the (terms, layers) values are
        # stacked, scaled by the weight column and summed in three launches instead of two per term (and as many
        # again in the backward); loss_dict keeps the reference's weighted per-term values
        names = [k.replace("_weight", "") for k, w in self.loss_weight_dict.items() if w > 1e-32]
        if names:
            wcol = self._weight_column(names, center.device)
            weighted = torch.stack([losses[n] for n in names]) * wcol
            final = weighted.sum()
            for i, n in enumerate(names):
                losses[n] = weighted[i]
        else:
            final = center.sum() * 0
        loss_dict = {}
        for name, per_layer in losses.items():
            loss_dict[name] = per_layer[nl - 1]
            for l in range(nl - 1):
                loss_dict[f"{name}_{l}"] = per_layer[l]
        return final, loss_dict

    def _weight_column(self, names, device):
        key = (tuple(names), str(device))
        cache = self.__dict__.setdefault("_wcol_cache", {})
        if key not in cache:
            cache[key] = torch.tensor([[float(self.loss_weight_dict[n + "_weight"])] for n in names],
                                      dtype=torch.float32, device=device)
        return cache[key]

    @_lib.on_tensor_device(lambda outputs, targets: targets["gt_box_present"])
    def forward(self, outputs, targets):
        nactual_gt = targets["gt_box_present"].sum(axis=1).long()
        num_boxes = torch.clamp(all_reduce_average(nactual_gt.sum()), min=1)
        targets["nactual_gt"] = nactual_gt
        if self.device_scalars and nactual_gt.is_cuda and self.layer_batched and "stacked_outputs" in outputs:
            targets["num_boxes"] = num_boxes.to(torch.float32)
            targets["num_boxes_replica"] = nactual_gt.sum()
        else:
            targets["num_boxes"] = num_boxes.item()
            targets["num_boxes_replica"] = nactual_gt.sum().item()
        for key in ["text_features_clip", "logit_scale", "gt_text_correlation_embedding",
                    "gt_text_correlation_embedding_mask", "weak_box_cate_label", "weak_confidence_weight"]:
            if key in outputs["outputs"]:
                targets[key] = outputs["outputs"][key]

        if self.layer_batched and "stacked_outputs" in outputs:
            # this package's model also returns its per-layer tensors stacked: same terms, one pass
            return self.stacked_forward(outputs["stacked_outputs"], targets)
        loss, loss_dict = self.single_output_forward(outputs["outputs"], targets, if_last_head=True)
        if "aux_outputs" in outputs:
            for k in range(len(outputs["aux_outputs"])):
                interm_loss, interm_loss_dict = self.single_output_forward(
                    outputs["aux_outputs"][k], targets, if_aux=True, if_last_head=False)
                loss += interm_loss
                for interm_key in interm_loss_dict:
                    loss_dict[f"{interm_key}_{k}"] = interm_loss_dict[interm_key]
        return loss, loss_dict


def build_criterion(args, dataset_config):
    """criterion.py:1219-1281: Hungarian matcher + the loss-weight dictionary from the reference's
    argparse names (main.py:154-205).  Flags this namespace does not carry count as 0."""
    if getattr(args, "only_image_class", False) or getattr(args, "only_prompt_loss", False):
        raise NotImplementedError("only_image_class / only_prompt_loss criteria are outside the hot path")
    matcher = Matcher(cost_class=args.matcher_cls_cost, cost_giou=args.matcher_giou_cost,
                      cost_center=args.matcher_center_cost, cost_objectness=args.matcher_objectness_cost)
    loss_weight_dict = {key: getattr(args, attr, 0) for key, attr in _WEIGHT_ARGS.items()}
    return SetCriterion(matcher, dataset_config, loss_weight_dict, train_range_max=args.train_range_max, args=args)
