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
reference).  The 3D gIoU that feeds its cost (utils/box_util.py:655-875, Cython
polygon clipping) is SURVEY.md 8f "next": ``giou_fn`` is a constructor hook and,
when absent, the gIoU cost term is zero.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


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

    def __init__(self, cost_class, cost_objectness, cost_giou, cost_center):
        super().__init__()
        self.cost_class = cost_class
        self.cost_objectness = cost_objectness
        self.cost_giou = cost_giou
        self.cost_center = cost_center

    @torch.no_grad()
    def forward(self, outputs, targets):
        from scipy.optimize import linear_sum_assignment

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
                      + self.cost_center * center_mat + self.cost_giou * giou_mat)
        final_cost = final_cost.detach().cpu().numpy()  # host round trip, as in the reference

        dev = pred_cls_prob.device
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


class SetCriterion(nn.Module):
    def __init__(self, matcher, dataset_config, loss_weight_dict, train_range_max=37,
                 confidence_type="clip-max-prob", giou_fn=None, args=None):
        super().__init__()
        self.dataset_config = dataset_config
        self.matcher = matcher
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

    # ---- box terms -----------------------------------------------------------------
    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, assignments):
        pred_logits = outputs["sem_cls_logits"]
        pred_objects = (pred_logits.argmax(-1) != pred_logits.shape[-1] - 1).sum(1)
        card_err = F.l1_loss(pred_objects.float(), targets["nactual_gt"])
        return {"loss_cardinality": card_err}

    def loss_sem_cls_softmax_skip_none_gt_sample(self, outputs, targets, assignments):
        pred_logits = outputs["sem_cls_logits"]
        gt_box_label = torch.gather(targets["gt_box_sem_cls_label"], 1, assignments["per_prop_gt_inds"])
        gt_box_label[assignments["proposal_matched_mask"].int() == 0] = pred_logits.shape[-1] - 1
        loss = F.cross_entropy(pred_logits.transpose(2, 1), gt_box_label, self.semcls_percls_weights,
                               reduction="none")
        # scenes without any GT box contribute 0 and are not counted (:236-244)
        has_object = (targets["gt_box_present"].sum(dim=1) != 0).to(loss.dtype)
        final_loss = (loss.sum(dim=1) * has_object).sum()
        final_loss = final_loss / (has_object.sum() * loss.shape[1] + 1e-32)
        return {"loss_sem_cls_softmax_skip_none_gt_sample": final_loss}

    def loss_angle(self, outputs, targets, assignments):
        angle_logits = outputs["angle_logits"]
        angle_residual = outputs["angle_residual_normalized"]
        if targets["num_boxes_replica"] > 0:
            gt_angle_label = targets["gt_angle_class_label"]
            gt_angle_residual = targets["gt_angle_residual_label"]
            gt_angle_residual_normalized = gt_angle_residual / (np.pi / self.dataset_config.num_angle_bin)
            gt_angle_label = torch.gather(gt_angle_label, 1, assignments["per_prop_gt_inds"])
            angle_cls_loss = F.cross_entropy(angle_logits.transpose(2, 1), gt_angle_label,
                                             reduction="none")
            angle_cls_loss = (angle_cls_loss * assignments["proposal_matched_mask"]).sum()
            gt_angle_residual_normalized = torch.gather(gt_angle_residual_normalized, 1,
                                                        assignments["per_prop_gt_inds"])
            one_hot = torch.zeros_like(angle_residual, dtype=torch.float32)
            one_hot.scatter_(2, gt_angle_label.unsqueeze(-1), 1)
            angle_residual_for_gt_class = torch.sum(angle_residual * one_hot, -1)
            angle_reg_loss = huber_loss(angle_residual_for_gt_class - gt_angle_residual_normalized,
                                        delta=1.0)
            angle_reg_loss = (angle_reg_loss * assignments["proposal_matched_mask"]).sum()
            angle_cls_loss /= targets["num_boxes"]
            angle_reg_loss /= targets["num_boxes"]
        else:
            angle_cls_loss = torch.sum(angle_logits) * 0
            angle_reg_loss = torch.sum(angle_residual) * 0
        return {"loss_angle_cls": angle_cls_loss, "loss_angle_reg": angle_reg_loss}

    def loss_center(self, outputs, targets, assignments):
        center_dist = outputs["center_dist"]
        if targets["num_boxes_replica"] > 0:
            center_loss = torch.gather(center_dist, 2,
                                       assignments["per_prop_gt_inds"].unsqueeze(-1)).squeeze(-1)
            center_loss = center_loss * assignments["proposal_matched_mask"]
            center_loss = center_loss.sum()
            if targets["num_boxes"] > 0:
                center_loss /= targets["num_boxes"]
        else:
            center_loss = torch.sum(center_dist) * 0
        return {"loss_center": center_loss}

    def loss_size(self, outputs, targets, assignments):
        gt_box_sizes = targets["gt_box_sizes_normalized"]
        pred_box_sizes = outputs["size_normalized"]
        if targets["num_boxes_replica"] > 0:
            inds = assignments["per_prop_gt_inds"].unsqueeze(-1).expand(-1, -1, gt_box_sizes.shape[-1])
            gt_box_sizes = torch.gather(gt_box_sizes, 1, inds)
            size_loss = F.l1_loss(pred_box_sizes, gt_box_sizes, reduction="none").sum(dim=-1)
            size_loss = size_loss * assignments["proposal_matched_mask"]
            size_loss = size_loss.sum()
            size_loss /= targets["num_boxes"]
        else:
            size_loss = torch.sum(pred_box_sizes) * 0
        return {"loss_size": size_loss}

    # ---- CLIP-space alignment terms (hot path) -----------------
    def loss_predicted_region_embed_l1(self, outputs, targets, assignments):
        """Masked L1 between the predicted region embedding and the CLIP image
        embedding of the cropped box, / (sum(mask) * 512)."""
        gt = targets["gt_text_correlation_embedding"]
        pred = outputs["text_correlation_embedding"]
        weight_maps = targets["gt_text_correlation_embedding_mask"]
        ave_weight = torch.sum(weight_maps) * pred.shape[2]
        l1_loss = F.l1_loss(pred * weight_maps, gt * weight_maps, reduction="sum") / ave_weight
        return {"loss_predicted_region_embed_l1": l1_loss}

    def loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi(self, outputs, targets, assignments):
        """CE(normalised embedding @ text^T * scale, label) * confidence, where matched
        proposals take the GT label/confidence and the others the CLIP weak label."""
        emb = outputs["text_correlation_embedding"]
        emb = emb / (emb.norm(dim=-1, keepdim=True) + 1e-32)
        text_features_clip = targets["text_features_clip"].to(torch.float32)
        temperature_param = targets["logit_scale"]
        correlation_map = torch.bmm(emb, text_features_clip.permute(0, 2, 1)) * temperature_param
        matched = assignments["proposal_matched_mask"].int() > 0
        seen_label = torch.gather(targets["gt_box_seen_sem_cls_label"], 1, assignments["per_prop_gt_inds"])
        seen_confi = torch.gather(targets["gt_box_seen_sem_cls_confi"], 1, assignments["per_prop_gt_inds"])
        gt_box_label = torch.where(matched, seen_label, targets["weak_box_cate_label"])
        gt_box_confidence = torch.where(matched, seen_confi, targets["weak_confidence_weight"])
        if self.confidence_type == "non-confidence":
            gt_box_confidence[gt_box_confidence > 1e-16] = 1
        loss = F.cross_entropy(correlation_map.transpose(2, 1), gt_box_label, reduction="none")
        all_num = torch.sum(gt_box_confidence > 1e-32) + 1e-32
        final_loss = torch.sum(loss * gt_box_confidence) / all_num
        return {"loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi": final_loss}

    # ---- drivers -------------------------------------------------------------------
    def single_output_forward(self, outputs, targets, if_region_embed=False, if_aux=False,
                              if_last_head=False):
        if self.giou_fn is not None:
            gious = self.giou_fn(outputs["box_corners"], targets["gt_box_corners"], targets["nactual_gt"],
                                 rotated_boxes=torch.any(targets["gt_box_angles"] > 0).item(),
                                 needs_grad=False)
        else:  # gIoU is SURVEY.md 8f "next": zero cost term
            gious = torch.zeros(outputs["center_normalized"].shape[0], outputs["center_normalized"].shape[1],
                                targets["gt_box_centers_normalized"].shape[1],
                                device=outputs["center_normalized"].device)
        outputs["gious"] = gious
        center_dist = torch.cdist(outputs["center_normalized"], targets["gt_box_centers_normalized"], p=1)
        outputs["center_dist"] = center_dist
        assignments = self.matcher(outputs, targets)

        losses = {}
        for k, fn in self.loss_functions.items():
            loss_wt_key = k + "_weight"
            if (loss_wt_key in self.loss_weight_dict and self.loss_weight_dict[loss_wt_key] > 1e-32) \
                    or loss_wt_key not in self.loss_weight_dict:
                losses.update(fn(outputs, targets, assignments))

        final_loss = 0
        for k, w in self.loss_weight_dict.items():
            if w > 1e-32:
                name = k.replace("_weight", "")
                losses[name] *= w
                final_loss += losses[name]
        return final_loss, losses

    def forward(self, outputs, targets):
        nactual_gt = targets["gt_box_present"].sum(axis=1).long()
        num_boxes = torch.clamp(all_reduce_average(nactual_gt.sum()), min=1).item()
        targets["nactual_gt"] = nactual_gt
        targets["num_boxes"] = num_boxes
        targets["num_boxes_replica"] = nactual_gt.sum().item()
        for key in ["text_features_clip", "logit_scale", "gt_text_correlation_embedding",
                    "gt_text_correlation_embedding_mask", "weak_box_cate_label", "weak_confidence_weight"]:
            if key in outputs["outputs"]:
                targets[key] = outputs["outputs"][key]

        loss, loss_dict = self.single_output_forward(outputs["outputs"], targets, if_last_head=True)
        if "aux_outputs" in outputs:
            for k in range(len(outputs["aux_outputs"])):
                interm_loss, interm_loss_dict = self.single_output_forward(
                    outputs["aux_outputs"][k], targets, if_aux=True, if_last_head=False)
                loss += interm_loss
                for interm_key in interm_loss_dict:
                    loss_dict[f"{interm_key}_{k}"] = interm_loss_dict[interm_key]
        return loss, loss_dict
