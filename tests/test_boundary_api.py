"""Boundary b3 (third row): the model / loss API that the reference's main.py and
engine.py call, replayed with the reference's call shapes and argparse names:

    model, _ = build_model(args, dataset_config)              main.py:988
    model.clip_model / model.res_encoder / model.if_keep_box   engine.py:85-117, main.py:356
    criterion = build_criterion(args, dataset_config)          main.py:997
    outputs = model(batch_data_label, curr_epoch=curr_epoch)   engine.py:144
    loss, loss_dict = criterion(outputs, batch_data_label)     engine.py:148
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from coda_neurips2023_amd import criterion as C
from coda_neurips2023_amd import model_3detr as M
from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig


def stage1_args(**over):
    """The flags scripts/coda_sunrgbd_stage1.sh passes, on top of main.py's defaults (main.py:37-304)."""
    ns = M.default_args(nqueries=32, preenc_npoints=128, enc_nlayers=1, dec_nlayers=2, enc_ffn_dim=64, dec_ffn_dim=64)
    weights = {attr: 0 for attr in C._WEIGHT_ARGS.values()}
    weights.update(loss_no_object_weight=0.05, loss_angle_cls_weight=0.1, loss_angle_reg_weight=0.5,
                   loss_center_weight=5.0, loss_size_weight=1.0, loss_no_object_contrast_weight=0.05,
                   loss_predicted_region_embed_l1_weight=1, loss_sem_cls_softmax_skip_none_gt_sample_weight=1)
    for k, v in weights.items():
        setattr(ns, k, v)
    for k, v in dict(matcher_giou_cost=3, matcher_cls_cost=1, matcher_center_cost=5, matcher_objectness_cost=5,
                     train_range_max=10, confidence_type="clip-max-prob", confidence_type_in_datalayer="clip-max-prob",
                     if_skip_no_seen_scene_objectness=False, if_only_seen_in_loss=False, only_image_class=False,
                     only_prompt_loss=False, if_clip_trainable=False).items():
        setattr(ns, k, v)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


class FakeClip(nn.Module):
    """Stands in for the CLIP module the reference loads (weights are not available here)."""

    def __init__(self):
        super().__init__()
        self.visual = nn.Linear(4, 4)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(100.0))


def clip_loader(args, dataset_config):
    text = torch.nn.functional.normalize(torch.randn(46, 512, generator=torch.Generator().manual_seed(0)), dim=-1)
    return {"text_features_fg_norm": text, "clip_model": FakeClip()}


def test_build_model_and_criterion_with_reference_call_shapes(monkeypatch):
    args, cfg = stage1_args(), HotPathDatasetConfig()
    # without a CLIP loader: the reference's "no clip here" branch (engine.py:99-100, 116-117)
    model, box_processor = M.build_model(args, cfg)
    assert not hasattr(model, "clip_model") and model.if_keep_box is False
    # with the deployment hook: same two-argument call, CLIP attributes present and frozen
    monkeypatch.setattr(M, "CLIP_LOADER", clip_loader)
    model, _ = M.build_model(args, cfg)
    model.if_keep_box = True  # main.py:356 pokes it
    assert hasattr(model, "clip_model") and hasattr(model, "res_encoder")
    model.train()
    model.clip_model.eval()    # engine.py:87
    model.res_encoder.eval()   # engine.py:91
    assert not any(p.requires_grad for p in model.clip_model.parameters())
    assert model.logit_scale is model.clip_model.logit_scale
    assert float(model.logit_scale.exp().clamp(max=100)) == pytest.approx(100.0, rel=1e-5)
    assert model.train_range_max == 10 and model.test_range_max == 46

    crit = C.build_criterion(args, cfg)
    assert isinstance(crit.matcher, C.Matcher) and crit.matcher.cost_giou == 3
    w = crit.loss_weight_dict
    assert "loss_no_object_weight" not in w and "loss_no_object_contrast_weight" not in w  # criterion.py:108-109
    assert w["loss_predicted_region_embed_l1_weight"] == 1 and w["loss_giou_weight"] == 0
    assert float(crit.semcls_percls_weights[-1]) == pytest.approx(0.05)
    assert crit.seen_semcls_percls_weights.shape == (11,)
    live = sorted(k for k in crit.loss_functions if crit._live(k))
    assert live == ["loss_angle", "loss_cardinality", "loss_center", "loss_predicted_region_embed_l1",
                    "loss_sem_cls_softmax_skip_none_gt_sample", "loss_size"]


def test_unsupported_live_terms_fail_loudly():
    cfg = HotPathDatasetConfig()
    with pytest.raises(NotImplementedError, match="loss_giou"):
        C.build_criterion(stage1_args(loss_giou_weight=1.0), cfg)
    with pytest.raises(NotImplementedError, match="loss_contrastive"):
        C.build_criterion(stage1_args(loss_contrastive_weight=0.5), cfg)
    with pytest.raises(NotImplementedError):
        C.build_criterion(stage1_args(only_image_class=True), cfg)


def _batch(dev, bsz, npts, ngt=8, seed=0):
    from coda_neurips2023_amd.box_util import get_3d_box_batch_tensor
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    pc, mn, mx = make_batch(bsz, npts, seed=seed)
    gen = torch.Generator().manual_seed(seed)
    nactual = torch.tensor([5, 0, 8][:bsz])
    centers = torch.from_numpy(mn)[:, None] + torch.rand(bsz, ngt, 3, generator=gen) * torch.from_numpy(mx - mn)[:, None]
    sizes = torch.rand(bsz, ngt, 3, generator=gen) * 1.0 + 0.2
    angles = (torch.rand(bsz, ngt, generator=gen) - 0.3) * 1.5
    scale = torch.from_numpy(mx - mn)[:, None]
    cam = torch.stack((centers[..., 0], -centers[..., 2], centers[..., 1]), -1)
    d = {"point_clouds": torch.from_numpy(pc), "point_cloud_dims_min": torch.from_numpy(mn),
         "point_cloud_dims_max": torch.from_numpy(mx),
         "gt_box_present": (torch.arange(ngt)[None] < nactual[:, None]).float(),
         "gt_box_sem_cls_label": torch.zeros(bsz, ngt, dtype=torch.int64),
         "gt_box_centers_normalized": (centers - torch.from_numpy(mn)[:, None]) / scale,
         "gt_box_sizes_normalized": sizes / scale,
         "gt_box_angles": angles, "gt_box_corners": get_3d_box_batch_tensor(sizes, angles, cam),
         "gt_angle_class_label": torch.randint(0, 12, (bsz, ngt), generator=gen),
         "gt_angle_residual_label": (torch.rand(bsz, ngt, generator=gen) - 0.5) * 0.2}
    return {k: v.to(dev) for k, v in d.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("layer_batched", [True, False])
def test_training_step_through_the_reference_call_sequence(dev, monkeypatch, layer_batched):
    """engine.py:144-159 on the GPU: forward, criterion, backward -- stage-1 recipe (L1 alignment term only,
    matcher with gIoU cost 3 on rotated boxes), both criterion drivers."""
    args, cfg = stage1_args(), HotPathDatasetConfig()
    bsz, nq = 3, args.nqueries
    gen = torch.Generator().manual_seed(4)
    img_emb = torch.nn.functional.normalize(torch.randn(bsz, nq, 512, generator=gen), dim=-1).to(dev)
    img_mask = (torch.rand(bsz, nq, 1, generator=gen) < 0.3).float().to(dev)

    def provider(inputs, outputs, curr_epoch=-1):  # stands in for the CLIP image-crop branch
        outputs["gt_text_correlation_embedding"] = img_emb
        outputs["gt_text_correlation_embedding_mask"] = img_mask
        return outputs

    monkeypatch.setattr(M, "CLIP_LOADER", lambda a, c: dict(clip_loader(a, c), region_embedding_provider=provider))
    torch.manual_seed(1)
    model, _ = M.build_model(args, cfg)
    model = model.to(dev)
    criterion = C.build_criterion(args, cfg).to(dev)
    criterion.layer_batched = layer_batched
    model.train()
    batch_data_label = _batch(dev, bsz, 1024)
    outputs = model(batch_data_label, curr_epoch=0)                 # engine.py:144
    assert outputs["outputs"]["text_features_clip"].shape == (bsz, 10, 512)
    loss, loss_dict = criterion(outputs, batch_data_label)          # engine.py:148
    assert torch.isfinite(loss) and float(loss) > 0
    for name in ["loss_angle_cls", "loss_angle_reg", "loss_center", "loss_size", "loss_predicted_region_embed_l1",
                 "loss_sem_cls_softmax_skip_none_gt_sample", "loss_cardinality"]:
        assert name in loss_dict and f"{name}_0" in loss_dict, name    # last layer + aux layer 0 (criterion.py:1213)
    assert "loss_feat_seen_softmax_weakly_loss_with_novel_cate_confi" not in loss_dict  # weight 0 in stage 1
    loss.backward()                                                  # engine.py:159
    grads = [p.grad for p in model.parameters() if p.requires_grad]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    test_training_step_through_the_reference_call_sequence.results = getattr(
        test_training_step_through_the_reference_call_sequence, "results", {})
    test_training_step_through_the_reference_call_sequence.results[layer_batched] = float(loss)
    res = test_training_step_through_the_reference_call_sequence.results
    if len(res) == 2:  # the layer-batched driver and the reference-shaped per-layer loop agree
        assert abs(res[True] - res[False]) < 1e-4 * abs(res[False])
