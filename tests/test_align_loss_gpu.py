"""Fused alignment losses (csrc/align_loss.hip through criterion.SetCriterion) against the torch
formulation of the same terms (criterion.py:924-943, 598-644 of the reference, restated in
criterion.py of this package and pinned against the reference by tests/test_criterion.py):
per-layer values and the gradient w.r.t. the region embedding, 1e-3 relative (north_star)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from coda_neurips2023_amd.criterion import SetCriterion

pytestmark = pytest.mark.gpu


def _crit(dev, ncls):
    cfg = SimpleNamespace(num_semcls=1, num_angle_bin=12)
    return SetCriterion(None, cfg, {}, train_range_max=ncls).to(dev)


def _case(dev, nl, b, nq, e, ncls, ngt, permuted, seed, shared=False):
    g = torch.Generator().manual_seed(seed)
    if permuted:  # the heads hand over a (layer, scene, query) VIEW of a (layer, query, scene) buffer
        emb = torch.randn(nl, nq, b, e, generator=g).to(dev).permute(0, 2, 1, 3)
    else:
        emb = torch.randn(nl, b, nq, e, generator=g).to(dev)
    emb = emb.detach().requires_grad_(True)
    targets = {
        "gt_text_correlation_embedding": F.normalize(torch.randn(b, nq, e, generator=g), dim=-1).to(dev),
        "gt_text_correlation_embedding_mask": (torch.rand(b, nq, 1, generator=g) < 0.3).float().to(dev),
        "text_features_clip": F.normalize(torch.randn(ncls, e, generator=g), dim=-1).unsqueeze(0).repeat(b, 1, 1).to(dev),
        "logit_scale": torch.tensor(14.2857, device=dev),
        "gt_box_seen_sem_cls_label": torch.randint(0, ncls, (b, ngt), generator=g).to(dev),
        "gt_box_seen_sem_cls_confi": torch.ones(b, ngt, device=dev),
        "weak_box_cate_label": torch.randint(0, ncls, (b, nq), generator=g).to(dev),
        "weak_confidence_weight": (torch.rand(b, nq, generator=g) * (torch.rand(b, nq, generator=g) < 0.5)).to(dev),
    }
    if shared:  # what model_3detr.py hands over: one prompt set, expanded over the scenes (stride 0)
        targets["text_features_clip"] = targets["text_features_clip"][0].unsqueeze(0).expand(b, -1, -1)
    targets["gt_text_correlation_embedding_mask"][0, 0, 0] = 1.0  # an empty mask is 0/0 in the reference too
    assign = {"per_prop_gt_inds": torch.randint(0, ngt, (nl, b, nq), generator=g).to(dev),
              "proposal_matched_mask": (torch.rand(nl, b, nq, generator=g) < 0.2).float().to(dev)}
    return emb, targets, assign


@pytest.mark.parametrize("nl,b,nq,e,ncls,permuted", [(8, 8, 256, 512, 10, True), (3, 2, 37, 512, 37, False),
                                                      (1, 1, 5, 64, 3, False), (2, 3, 16, 1024, 232, True)])
def test_fused_alignment_matches_torch(dev, nl, b, nq, e, ncls, permuted):
    crit = _crit(dev, ncls)
    names = ["loss_predicted_region_embed_l1", "loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi"]
    res = {}
    for fused in (True, False):
        crit.fused_alignment = fused
        emb, targets, assign = _case(dev, nl, b, nq, e, ncls, 7, permuted, seed=nl * 100 + nq)
        outs = {"text_correlation_embedding": emb}
        l1 = crit.stacked_loss_predicted_region_embed_l1(outs, targets, assign)[names[0]]
        ce = crit.stacked_loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi(outs, targets, assign)[names[1]]
        assert l1.shape == ce.shape == (nl,)
        w = torch.linspace(0.5, 1.5, nl, device=dev)
        ((l1 * w).sum() + 2.0 * (ce * w).sum()).backward()
        res[fused] = (l1.detach(), ce.detach(), emb.grad.detach().clone())
    for i, what in enumerate(["l1", "ce", "d emb"]):
        got, ref = res[True][i].double().cpu().numpy(), res[False][i].double().cpu().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < 1e-3, f"{what}: {err:.3e}"


@pytest.mark.parametrize("nl,b,nq,e,ncls", [(8, 8, 256, 512, 232), (8, 8, 256, 512, 1201), (2, 2, 64, 512, 100),
                                            (8, 8, 256, 512, 64)])
def test_gemm_route_at_the_stage2_class_counts(dev, nl, b, nq, e, ncls):
    """ncls >= 64 with one prompt set for all scenes (232 / 1201 prompts, models/model_3detr.py:321): the class logits and
    their gradient are dense products on the matrix cores (align_loss._AlignLossGemm: bf16x3 kernels at 16 384 rows, the
    library below 8192) -- values and gradient against the torch formulation, 1e-3 relative."""
    from coda_neurips2023_amd import align_loss, gemm
    crit = _crit(dev, ncls)
    names = ["loss_predicted_region_embed_l1", "loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi"]
    res = {}
    before = gemm.x3_calls
    for fused in (True, False):
        crit.fused_alignment = fused
        emb, targets, assign = _case(dev, nl, b, nq, e, ncls, 7, True, seed=ncls, shared=True)
        assert align_loss._shared_text(targets["text_features_clip"]) is not None
        outs = {"text_correlation_embedding": emb}
        l1 = crit.stacked_loss_predicted_region_embed_l1(outs, targets, assign)[names[0]]
        ce = crit.stacked_loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi(outs, targets, assign)[names[1]]
        w = torch.linspace(0.5, 1.5, nl, device=dev)
        ((l1 * w).sum() + 2.0 * (ce * w).sum()).backward()
        res[fused] = (l1.detach(), ce.detach(), emb.grad.detach().clone())
    if nl * b * nq >= 8192 and ncls > 128 and gemm._X3:
        assert gemm.x3_calls - before >= 2   # both products of the fused evaluation took the matrix-core route
    for i, what in enumerate(["l1", "ce", "d emb"]):
        got, ref = res[True][i].double().cpu().numpy(), res[False][i].double().cpu().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < 1e-3, f"{what}: {err:.3e}"


def test_single_layer_entry_points_use_the_same_code(dev):
    crit = _crit(dev, 10)
    emb, targets, assign = _case(dev, 1, 2, 32, 512, 10, 5, False, seed=3)
    outs = {"text_correlation_embedding": emb[0]}
    a1 = {k: v[0] for k, v in assign.items()}
    fused = crit.loss_predicted_region_embed_l1(outs, targets, a1)["loss_predicted_region_embed_l1"]
    crit.fused_alignment = False
    ref = crit.loss_predicted_region_embed_l1(outs, targets, a1)["loss_predicted_region_embed_l1"]
    assert abs(float(fused) - float(ref)) < 1e-4 * abs(float(ref))
