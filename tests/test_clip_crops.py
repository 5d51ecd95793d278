"""Image side of the CLIP distillation branch up to the tower's input: the numpy oracle
against vectors from the reference's own projection function (tests/golden/clip_crops.npz), and the two HIP kernels
(coda_project_box_rects_f64, coda_crop_resize_f32 through coda_neurips2023_amd.clip_crops) against the oracle."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_crops.npz"))
CAM = {k[4:]: G[k] for k in G.files if k.startswith("cam_")}


def test_oracle_projection_matches_reference_vectors():
    from oracle import crop_oracle as CO
    uv, depth = CO.project(G["corners"], CAM)
    np.testing.assert_allclose(uv, G["uv_raw"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(depth, G["depth"], rtol=1e-12, atol=1e-12)
    _, _, rects, valid = CO.rects(G["corners"], G["sizes"], CAM)
    assert valid.sum() > 20 and (~valid).sum() >= 10           # boxes behind the camera, the zero-size box
    assert not valid[0, 5] and not valid[:, :3].any()
    assert (rects[valid][:, 2] > rects[valid][:, 0]).all()


def _gpu_inputs(dev):
    return {k: torch.from_numpy(v).to(dev) for k, v in CAM.items()}


@pytest.mark.gpu
def test_rect_kernel_matches_oracle(dev):
    from coda_neurips2023_amd import clip_crops as CC
    from oracle import crop_oracle as CO
    rects, valid, uv, depth = CC.project_box_rects(_gpu_inputs(dev), torch.from_numpy(G["corners"]).to(dev),
                                                   torch.from_numpy(G["sizes"]).to(dev), want_uv=True)
    uv_o, depth_o, rects_o, valid_o = CO.rects(G["corners"], G["sizes"], CAM)
    np.testing.assert_allclose(uv.cpu().numpy(), uv_o, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(depth.cpu().numpy(), depth_o, rtol=1e-12, atol=1e-12)
    assert np.array_equal(rects.cpu().numpy(), rects_o)
    assert np.array_equal(valid.cpu().numpy().astype(bool), valid_o)
    # a zx flip and a flipped image exercise the remaining branches
    inp = dict(CAM, zx_flip_array=np.array([[1.0], [-1.0], [1.0]]))
    r2, v2 = CC.project_box_rects({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in inp.items()},
                                  torch.from_numpy(G["corners"]).to(dev), torch.from_numpy(G["sizes"]).to(dev))
    _, _, r2o, v2o = CO.rects(G["corners"], G["sizes"], inp)
    assert np.array_equal(r2.cpu().numpy(), r2o) and np.array_equal(v2.cpu().numpy().astype(bool), v2o)


@pytest.mark.gpu
def test_crop_resize_matches_torch_pipeline(dev):
    """Every valid proposal's crop against the reference's per-box pipeline restated with torch CPU ops: equal
    after the uint8 rounding except where the interpolated value sits on a rounding boundary."""
    from coda_neurips2023_amd import clip_crops as CC
    from oracle import crop_oracle as CO
    gen = torch.Generator().manual_seed(3)
    b, h, w = 3, 531, 730
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    base = torch.stack(((xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy) % 251), -1)
    images = ((base[None] + torch.randint(0, 40, (b, h, w, 3), generator=gen)) % 256).to(torch.uint8)
    _, _, rects_o, valid_o = CO.rects(G["corners"], G["sizes"], CAM)
    rects, valid = CC.project_box_rects(_gpu_inputs(dev), torch.from_numpy(G["corners"]).to(dev),
                                        torch.from_numpy(G["sizes"]).to(dev))
    k = rects.shape[1]
    sel = torch.stack([torch.randperm(k, generator=gen)[:12] for _ in range(b)])
    out = CC.crop_resize(images.to(dev), sel.to(dev), rects, valid, 224).cpu()
    assert out.shape == (b * 12, 3, 224, 224)
    std = torch.tensor(CO.STD).view(3, 1, 1)
    checked = 0
    for i in range(b):
        for j in range(12):
            box = int(sel[i, j])
            got = out[i * 12 + j]
            if not valid_o[i, box]:
                white = (1.0 - torch.tensor(CO.MEAN)) / torch.tensor(CO.STD)
                assert torch.allclose(got, white.view(3, 1, 1).expand_as(got))
                continue
            ref = CO.crop_resize(images[i], rects_o[i, box], 224)
            lsb = ((got - ref) * std * 255.0).abs()          # difference in units of one uint8 step
            assert float(lsb.max()) <= 1.0 + 1e-3
            assert float((lsb > 0.5).float().mean()) < 2e-3   # only rounding-boundary pixels differ
            checked += 1
    assert checked >= 15


@pytest.mark.gpu
def test_region_embedding_provider_fills_embeddings_and_mask(dev):
    from coda_neurips2023_amd import clip_crops as CC

    class Tower(torch.nn.Module):   # stands in for the CLIP module (weights are not available here)
        def __init__(self):
            super().__init__()
            self.visual = torch.nn.Module()
            self.visual.input_resolution = 32
            self.proj = torch.nn.Linear(3, 512)

        def encode_image(self, x):
            return self.proj(x.mean(dim=(2, 3)))

    b = 3
    inputs = _gpu_inputs(dev)
    inputs["input_image"] = torch.randint(0, 256, (b, 531, 730, 3), dtype=torch.uint8,
                                          generator=torch.Generator().manual_seed(0)).to(dev)
    outputs = {"box_corners_xyz": torch.from_numpy(G["corners"]).to(dev), "size_unnormalized": torch.from_numpy(G["sizes"]).to(dev)}
    prov = CC.RegionEmbeddingProvider(Tower().to(dev), distillation_box_num=8, box_pool=128, rng=np.random.RandomState(1))
    out = prov(inputs, outputs, curr_epoch=0)
    emb, mask = out["gt_text_correlation_embedding"], out["gt_text_correlation_embedding_mask"]
    assert emb.shape == (b, G["corners"].shape[1], 512) and mask.shape == (b, G["corners"].shape[1], 1)
    assert 0 < int(mask.sum()) <= b * 8
    assert float(emb[mask.squeeze(-1) == 0].abs().max()) == 0.0 and float(emb[mask.squeeze(-1) > 0].abs().min()) >= 0.0
    from oracle import crop_oracle as CO
    valid_o = CO.rects(G["corners"], G["sizes"], CAM)[3]
    assert not (mask.squeeze(-1).cpu().numpy() > 0)[~valid_o].any()   # skipped proposals never carry an embedding


@pytest.mark.gpu
def test_provider_with_this_packages_tower_matches_the_oracles(dev):
    """crops (kernel) -> ImageTower (coda_vit_fwd, float32) against crop_oracle -> clip_tower_oracle on the same
    selection: the image branch end to end, on a small seeded tower (64 px, two blocks)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from golden.weights import fill_deterministic
    from coda_neurips2023_amd import clip_crops as CC, clip_tower
    from oracle import clip_tower_oracle, crop_oracle as CO
    tower = clip_tower.ImageTower(64, 64, 2, 128, 16)
    fill_deterministic(tower.visual, seed=5)
    tower = tower.to(dev)
    b = 3
    inputs = _gpu_inputs(dev)
    image = torch.randint(0, 256, (b, 531, 730, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(4))
    inputs["input_image"] = image.to(dev)
    outputs = {"box_corners_xyz": torch.from_numpy(G["corners"]).to(dev), "size_unnormalized": torch.from_numpy(G["sizes"]).to(dev)}
    prov = CC.RegionEmbeddingProvider(tower, distillation_box_num=6, box_pool=24, rng=np.random.RandomState(3))
    out = prov(inputs, outputs, curr_epoch=0)
    emb, mask = out["gt_text_correlation_embedding"].cpu().numpy(), out["gt_text_correlation_embedding_mask"].cpu().numpy()
    rng = np.random.RandomState(3)
    select = np.stack([rng.choice(np.arange(24), 6, replace=False) for _ in range(b)])
    _, _, rects, valid = CO.rects(G["corners"], G["sizes"], CAM)
    sd = {k: v.cpu().numpy() for k, v in tower.visual.state_dict().items()}
    checked = 0
    for i in range(b):
        for k in select[i]:
            if not valid[i, k]:
                assert mask[i, k, 0] == 0 and np.abs(emb[i, k]).max() == 0
                continue
            crop = np.asarray(CO.crop_resize(image[i], rects[i, k], 64))
            ref, _ = clip_tower_oracle.forward(sd, crop[None], 2, 16)
            assert mask[i, k, 0] == 1
            # crops may differ by one 8-bit step at a few pixels (bicubic rounding ties), the tower is float32
            np.testing.assert_allclose(emb[i, k], ref[0], rtol=0, atol=5e-3)
            checked += 1
    assert checked >= 6
