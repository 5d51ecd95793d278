"""Whole detector (tiny configuration) on the GPU against the reference's own
methods (golden fixture model_tiny.npz): indices bit-exact, every output key
within 1e-3 relative, parameter-gradient digests, open-vocabulary class scores."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from golden.weights import fill_deterministic, grad_digest  # noqa: E402
from test_model_structure import tiny_args  # noqa: E402

from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig  # noqa: E402
from coda_neurips2023_amd.model_3detr import build_model  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _golden_distance_mode(_distance_mode_default):
    """FPS / ball_query inside the model run in the mode the fixture was generated in."""
    from tests._modes import fixture_mode, set_distance_mode
    set_distance_mode(fixture_mode(G))
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_tiny.npz"))
RTOL = 1e-3


def close(got, ref, what, rtol=RTOL):
    got = got.detach().cpu().numpy()
    err = np.abs(got.astype(np.float64) - ref).max() / (np.abs(ref).max() + 1e-12)
    assert err < rtol, f"{what}: max err / max|ref| = {err:.3e}"


def _inputs(dev):
    return {"point_clouds": torch.from_numpy(G["pc"]).to(dev),
            "point_cloud_dims_min": torch.from_numpy(G["dims_min"]).to(dev),
            "point_cloud_dims_max": torch.from_numpy(G["dims_max"]).to(dev)}


def test_tiny_model_matches_reference(dev):
    """The fixture ran ONE reference model: a train-mode step (batch-statistics BN, running
    statistics updated once), then eval mode on the updated statistics; same sequence here."""
    model, _ = build_model(tiny_args(), HotPathDatasetConfig())
    fill_deterministic(model, seed=9)
    model.to(dev)
    for mode in ["train", "eval"]:
        _check_mode(model, dev, mode)


def _check_mode(model, dev, mode):
    model.train(mode == "train")
    model.zero_grad()
    inputs = _inputs(dev)
    pred = model(inputs)
    with torch.no_grad():
        was = model.training
        model.eval()  # do not touch the BN running statistics a second time
        enc_xyz, _, enc_inds = model.run_encoder(inputs["point_clouds"])
        model.train(was)
    assert np.array_equal(enc_inds.cpu().numpy(), G[f"{mode}_enc_inds"])
    assert np.array_equal(enc_xyz.cpu().numpy(), G[f"{mode}_enc_xyz"])
    o = pred["outputs"]
    int_keys = []
    for k in [f for f in G.files if f.startswith(f"{mode}_out/")]:
        name = k.split("/", 1)[1]
        close(o[name], G[k], f"{mode} outputs[{name}]")
    assert len(pred["aux_outputs"]) == 2
    for li, aux in enumerate(pred["aux_outputs"]):
        for name in ["sem_cls_logits", "center_normalized", "box_corners", "text_correlation_embedding"]:
            close(aux[name], G[f"{mode}_aux{li}/{name}"], f"{mode} aux{li}[{name}]")
    if mode == "train":
        loss = 0
        for name in ["sem_cls_logits", "text_correlation_embedding", "center_normalized", "size_normalized",
                     "angle_logits", "angle_residual", "box_corners"]:
            w = torch.from_numpy(G[f"train_lossw/{name}"]).to(dev)
            loss = loss + (o[name] * w).sum()
            for aux in pred["aux_outputs"]:
                loss = loss + 0.5 * (aux[name] * w).sum()
        loss.backward()
        assert abs(float(loss.detach()) - float(G["train_loss"])) < RTOL * abs(float(G["train_loss"])) + 1e-2
        # Gradient digests at the north-star tolerance.  The fixture's scene was chosen by
        # make_golden.py (kink-margin search, `scene_seed` / `kink_margin` in the file) so that no
        # ReLU pre-activation and no max-pool runner-up lies within fp32 rounding of a branch flip:
        # every fp32 implementation takes the same branches and the whole backward chain has to
        # agree to 1e-3 (it was 2e-2 in round 1, one flip moved the first SA layer by 1 %).
        GRAD_TOL = RTOL
        dig = grad_digest(model)
        gmax = max(abs(G[k][1]) for k in G.files if k.startswith("train_grad/"))
        for k in [f for f in G.files if f.startswith("train_grad/")]:
            name = k.split("/", 1)[1]
            ref = G[k]
            scale = max(abs(ref[1]), 1e-3 * gmax)  # numerically-zero gradients: global scale
            assert abs(dig[name][1] - ref[1]) < GRAD_TOL * scale + 1e-6, f"grad norm {name}"
            assert np.abs(dig[name][2:] - ref[2:]).max() < GRAD_TOL * max(np.abs(ref[2:]).max(), scale / 10) + 1e-6, name


def test_open_vocabulary_class_scores(dev):
    text = torch.from_numpy(G["text_features"])
    model, _ = build_model(tiny_args(), HotPathDatasetConfig(), text_features_fg_norm=text)
    fill_deterministic(model, seed=9)
    with torch.no_grad():
        model.logit_scale.fill_(float(np.log(1 / 0.07)))
    model.to(dev).train()
    with torch.no_grad():
        model(_inputs(dev))  # the fixture's model had taken one train-mode step (BN running stats)
        model.eval()
        pred = model(_inputs(dev), if_real_test=True)
    close(pred["outputs"]["sem_cls_prob"], G["class_scores"], "get_class_scores", rtol=2e-3)
    assert pred["outputs"]["text_features_clip"].shape == (2, 10, 512)


def test_prefetched_sampling_gives_identical_outputs(dev):
    """model.prefetch_sampling(batch) (FPS on a side stream) followed by forward on the same
    tensor: identical outputs to the in-line path; a modified or unknown tensor samples in line."""
    import numpy as np
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    model, _ = build_model(tiny_args(), HotPathDatasetConfig())
    fill_deterministic(model, seed=5)
    model.to(dev).eval()
    pc, mn, mx = make_batch(2, 5000, seed=3)
    inputs = {"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev),
              "point_cloud_dims_max": torch.from_numpy(mx).to(dev)}
    with torch.no_grad():
        ref = model(inputs)["outputs"]
        model.prefetch_sampling(inputs)
        assert len(model._sampling_prefetcher._pending) == 1
        # the object queries were sampled ahead as well (the tiny model's encoder hands its xyz through)
        assert "query_inds" in model._sampling_prefetcher._pending[0][2]
        import coda_neurips2023_amd.model_3detr as m3
        calls = []
        real = m3.furthest_point_sample
        m3.furthest_point_sample = lambda *a, **k: (calls.append(a[1]), real(*a, **k))[1]
        try:
            got = model(inputs)["outputs"]
        finally:
            m3.furthest_point_sample = real
        assert calls == []  # no in-line query sampling
        assert len(model._sampling_prefetcher._pending) == 0  # consumed
        for k in ["center_normalized", "sem_cls_logits", "box_corners"]:
            # same indices and groups; the shared MLP may run on de-duplicated rows (summation order)
            assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-5), k
        model.prefetch_sampling(inputs)
        inputs["point_clouds"].mul_(1.0)  # version bump: the stale entry must not be used
        other = model(inputs)["outputs"]
        assert len(model._sampling_prefetcher._pending) == 1
        assert torch.allclose(other["center_normalized"], ref["center_normalized"], rtol=1e-4, atol=1e-5)


def test_pre_encoded_entry_point_gives_the_same_outputs(dev):
    """forward(inputs, pre_encoded=run_pre_encoder(pc)) is forward(inputs): the set-abstraction stage can
    be run by the caller (model_3detr.run_pre_encoder) and handed back."""
    model, _ = build_model(tiny_args(), HotPathDatasetConfig())
    fill_deterministic(model, seed=9)
    model.to(dev).eval()  # eval: no running-statistics update between the two calls
    inputs = _inputs(dev)
    with torch.no_grad():
        ref = model(inputs)["outputs"]
        xyz, feat, inds = model.run_pre_encoder(inputs["point_clouds"])
        assert xyz.shape[1:] == (128, 3) and feat.shape[1:] == (256, 128) and inds.dtype == torch.int32
        got = model(inputs, pre_encoded=(xyz, feat, inds))["outputs"]
    for k in ["sem_cls_logits", "center_normalized", "size_normalized", "angle_logits", "box_corners",
              "text_correlation_embedding"]:
        err = float((got[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12))
        assert err < 1e-5, (k, err)
