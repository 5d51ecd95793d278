"""Fused matched box losses (csrc/box_loss.hip through criterion.SetCriterion._fused_box_terms) against the torch
formulation of the same criterion (which tests/test_criterion.py pins against the reference's own loss_* methods):
per-layer values of all five terms and the gradients that reach the five prediction tensors."""
import pytest
import torch

from coda_neurips2023_amd.criterion import SetCriterion
from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig

pytestmark = pytest.mark.gpu


def make_case(dev, nl, b, nq, ngt, seed, empty_scene=True):
    g = torch.Generator().manual_seed(seed)

    def heads(c):  # (layer, query, scene, C) buffer viewed as (layer, scene, query, C), like the model's heads
        return torch.randn(nl, nq, b, c, generator=g).to(dev).permute(0, 2, 1, 3).requires_grad_(True)

    outs = {"sem_cls_logits": heads(2), "angle_logits": heads(12), "angle_residual_normalized": heads(12),
            "center_normalized": torch.rand(nl, b, nq, 3, generator=g).to(dev).requires_grad_(True),
            "size_normalized": torch.rand(nl, b, nq, 3, generator=g).to(dev).requires_grad_(True)}
    nactual = torch.randint(1, ngt + 1, (b,), generator=g)
    if empty_scene:
        nactual[0] = 0
    present = (torch.arange(ngt)[None] < nactual[:, None]).float()
    targets = {"gt_box_present": present.to(dev),
               "gt_box_sem_cls_label": torch.zeros(b, ngt, dtype=torch.int64, device=dev),
               "gt_angle_class_label": torch.randint(0, 12, (b, ngt), generator=g).to(dev),
               "gt_angle_residual_label": ((torch.rand(b, ngt, generator=g) - 0.5) * 1.2).to(dev),  # both Huber branches
               "gt_box_centers_normalized": torch.rand(b, ngt, 3, generator=g).to(dev),
               "gt_box_sizes_normalized": torch.rand(b, ngt, 3, generator=g).to(dev),
               "nactual_gt": nactual.to(dev), "num_boxes": float(max(int(nactual.sum()), 1)),
               "num_boxes_replica": int(nactual.sum())}
    matched = (torch.rand(nl, b, nq, generator=g) < 0.3).float()
    matched[:, 0] = 0  # nothing can match in the empty scene
    assign = {"per_prop_gt_inds": torch.randint(0, ngt, (nl, b, nq), generator=g).to(dev),
              "proposal_matched_mask": matched.to(dev)}
    return outs, targets, assign


@pytest.mark.parametrize("nl,b,nq,ngt", [(8, 8, 256, 64), (2, 3, 33, 5)])
def test_fused_box_terms_match_torch_formulation(dev, nl, b, nq, ngt):
    crit = SetCriterion(None, HotPathDatasetConfig(), {"loss_no_object_weight": 0.2}, train_range_max=10).to(dev)
    outs, targets, assign = make_case(dev, nl, b, nq, ngt, seed=nl + nq)
    fused = crit._fused_box_terms(outs, targets, assign)
    assert fused is not None
    c = outs["center_normalized"]
    ref_outs = dict(outs, center_dist=torch.cdist(c.flatten(0, 1), targets["gt_box_centers_normalized"].repeat(nl, 1, 1),
                                                  p=1).view(nl, b, nq, ngt))
    ref = {}
    for fn in (crit.stacked_loss_sem_cls_softmax_skip_none_gt_sample, crit.stacked_loss_angle, crit.stacked_loss_center,
               crit.stacked_loss_size):
        ref.update(fn(ref_outs, targets, assign))
    assert sorted(ref) == sorted(fused)
    w = {k: torch.randn(nl, generator=torch.Generator().manual_seed(1)).to(dev) for k in ref}
    for k in ref:
        err = float((fused[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12))
        assert err < 1e-5, (k, err)
    leaves = [outs[k] for k in ["sem_cls_logits", "angle_logits", "angle_residual_normalized", "center_normalized",
                                "size_normalized"]]
    g_f = torch.autograd.grad(sum((fused[k] * w[k]).sum() for k in ref), leaves)
    g_r = torch.autograd.grad(sum((ref[k] * w[k]).sum() for k in ref), leaves)
    for a, r in zip(g_f, g_r):
        assert float((a - r).abs().max()) < 1e-5 * float(r.abs().max()) + 1e-9


def test_no_boxes_on_this_worker(dev):
    """num_boxes_replica == 0: the matched terms are exact zeros that keep the heads in the graph."""
    crit = SetCriterion(None, HotPathDatasetConfig(), {}, train_range_max=10).to(dev)
    outs, targets, assign = make_case(dev, 2, 2, 16, 4, seed=9)
    targets["gt_box_present"].zero_()
    targets["num_boxes_replica"], targets["num_boxes"] = 0, 1.0
    assign["proposal_matched_mask"].zero_()
    fused = crit._fused_box_terms(outs, targets, assign)
    for k in ["loss_angle_cls", "loss_angle_reg", "loss_center", "loss_size", "loss_sem_cls_softmax_skip_none_gt_sample"]:
        assert float(fused[k].abs().max()) == 0.0
    (g,) = torch.autograd.grad(sum(v.sum() for v in fused.values()), [outs["angle_logits"]])
    assert float(g.abs().max()) == 0.0
