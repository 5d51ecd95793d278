"""PointnetSAModuleVotes on the GPU against the golden fixture produced by the
reference's own module (third_party_pointnet2/pointnet2/pointnet2_modules.py:161-268)
on CPU torch: indices bit-exact, fp32 features / gradients within 1e-3 relative
(north_star tolerance), train-mode and eval-mode BatchNorm."""
import numpy as np
import pytest
import torch

from coda_neurips2023_amd.pointnet2 import pointnet2_modules

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def _close(got, ref, what, tol=None):
    got = got.detach().cpu().numpy()
    scale = np.abs(ref).max() + 1e-12
    err = np.abs(got - ref).max() / scale
    assert err < (RTOL if tol is None else tol), f"{what}: max err / max|ref| = {err:.3e}"


@pytest.mark.parametrize("tag,c_in", [("xyz", 0), ("feat", 5)])
def test_sa_module_matches_reference(dev, golden_sa, tag, c_in):
    g = golden_sa
    mod = pointnet2_modules.PointnetSAModuleVotes(mlp=[c_in, 16, 32, 64], npoint=128, radius=0.3,
                                                  nsample=32, normalize_xyz=True)
    state0 = {k.split("/", 1)[1]: torch.from_numpy(np.asarray(g[k])) for k in g.files
              if k.startswith(f"{tag}_state0/")}
    mod.load_state_dict(state0, strict=True)
    mod.to(dev).train()
    xyz = torch.from_numpy(g[f"{tag}_xyz"]).to(dev)
    feats = torch.from_numpy(g[f"{tag}_feats"]).to(dev).requires_grad_(True) if c_in else None
    new_xyz, new_feat, inds = mod(xyz, feats)
    assert inds.dtype == torch.int32
    assert np.array_equal(inds.cpu().numpy(), g[f"{tag}_inds"])
    assert np.array_equal(new_xyz.cpu().numpy(), g[f"{tag}_new_xyz"])
    _close(new_feat, g[f"{tag}_new_feat_train"], "train-mode features")
    (new_feat * torch.from_numpy(g[f"{tag}_gw"]).to(dev)).sum().backward()
    for k, p in mod.named_parameters():
        _close(p.grad, g[f"{tag}_grad/{k}"], f"grad {k}")
    if c_in:
        _close(feats.grad, g[f"{tag}_feats_grad"], "grad features")
    for k, v in mod.state_dict().items():  # BN running statistics after one step
        ref = g[f"{tag}_state1/{k}"]
        if ref.dtype.kind == "f":
            _close(v, ref, f"state {k}")
        else:
            assert int(v) == int(ref)
    mod.eval()
    with torch.no_grad():
        _, new_feat_eval, _ = mod(xyz, feats)
    _close(new_feat_eval, g[f"{tag}_new_feat_eval"], "eval-mode features")


def test_pre_encoder_widths_match_reference_through_the_mfma_pipeline(dev, monkeypatch):
    """PointnetSAModuleVotes(mlp=[0, 64, 128, 256]) -- the pre-encoder's own widths, where the shared MLP runs on the
    hand-written fp32-MFMA pipeline (csrc/sa_mfma.hip) -- against the fixture the REFERENCE's module produced
    (tests/golden/make_golden.py: golden_sa_module_wide; a quarter of the BatchNorm gammas negative): sampled
    indices bit-exact, train- and eval-mode features, every parameter gradient and the running statistics within
    1e-3 relative (north_star tolerance)."""
    import os

    from coda_neurips2023_amd.pointnet2 import fused_sa_mlp
    from tests._modes import fixture_mode, set_distance_mode
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sa_module_wide.npz"))
    set_distance_mode(fixture_mode(g))
    monkeypatch.delenv("CODA_SA_MLP", raising=False)
    used = []
    real = fused_sa_mlp._MfmaMlpPool.apply
    monkeypatch.setattr(fused_sa_mlp._MfmaMlpPool, "apply", lambda *a: (used.append(1), real(*a))[1])
    mod = pointnet2_modules.PointnetSAModuleVotes(mlp=[0, 64, 128, 256], npoint=96, radius=0.25, nsample=64,
                                                  normalize_xyz=True)
    mod.load_state_dict({k.split("/", 1)[1]: torch.from_numpy(np.asarray(g[k])) for k in g.files
                         if k.startswith("state0/")}, strict=True)
    mod.to(dev).train()
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    new_xyz, new_feat, inds = mod(xyz)
    assert used, "the MFMA pipeline did not run"
    assert np.array_equal(inds.cpu().numpy(), g["inds"])
    assert np.array_equal(new_xyz.cpu().numpy(), g["new_xyz"])
    _close(new_feat, g["new_feat_train"], "train-mode features")
    (new_feat * torch.from_numpy(g["gw"]).to(dev)).sum().backward()
    for k, p in mod.named_parameters():
        _close(p.grad, g[f"grad/{k}"], f"grad {k}")
    for k, v in mod.state_dict().items():
        ref = g[f"state1/{k}"]
        if ref.dtype.kind == "f":
            _close(v, ref, f"state {k}")
        else:
            assert int(v) == int(ref)
    mod.eval()
    with torch.no_grad():
        _, new_feat_eval, _ = mod(xyz)
    _close(new_feat_eval, g["new_feat_eval"], "eval-mode features")


def test_a_prepared_front_can_be_used_twice(dev):
    """The packed front carries the statistics accumulators of the two MFMA layers, zeroed by the packing kernel.  A
    second forward on the same front (a prefetched batch used twice, an activation-checkpoint recompute) must not add
    into the first pass's sums: same features both times."""
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    pc, _, _ = make_batch(2, 5000, seed=8)
    xyz = torch.from_numpy(pc).to(dev)
    torch.manual_seed(2)
    mod = pointnet2_modules.PointnetSAModuleVotes(mlp=[0, 64, 128, 256], npoint=128, radius=0.2, nsample=64,
                                                  normalize_xyz=True).to(dev).train()
    front = mod.prepare(xyz)
    assert "packed" in front
    with torch.no_grad():
        a = mod(xyz, prepared=front)[1].clone()
        b = mod(xyz, prepared=front)[1].clone()
    # (equal up to the order of the fp64 statistics atomics; doubled sums would change the features by tens of percent)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


def test_prepared_front_equals_the_inline_forward(dev):
    """forward(xyz, prepared=prepare(xyz)) -- the sampling prefetcher's route through the MFMA pipeline -- equals
    forward(xyz): same indices, features, parameter gradients and running statistics (a quarter of the BatchNorm
    gammas negative: min-pooling channels)."""
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    pc, _, _ = make_batch(3, 6000, seed=21)
    xyz = torch.from_numpy(pc).to(dev)
    torch.manual_seed(4)
    mods = [pointnet2_modules.PointnetSAModuleVotes(mlp=[0, 64, 128, 256], npoint=256, radius=0.2, nsample=64,
                                                    normalize_xyz=True).to(dev).train() for _ in range(2)]
    mods[1].load_state_dict(mods[0].state_dict())
    with torch.no_grad():
        for m in mods:
            for bn in (layer.bn.bn for layer in m.mlp_module.children()):
                bn.weight.mul_(torch.where(torch.arange(bn.weight.numel(), device=dev) % 4 == 0, -1.0, 1.0))
    gw = torch.randn(3, 256, 256, device=dev)
    ref_xyz, ref_feat, ref_inds = mods[0](xyz)
    (ref_feat * gw).sum().backward()
    prepared = mods[1].prepare(xyz)
    assert prepared is not None and "packed" in prepared
    new_xyz, feat, inds = mods[1](xyz, prepared=prepared)
    assert torch.equal(inds, ref_inds) and torch.equal(new_xyz, ref_xyz)
    _close(feat, ref_feat.detach().cpu().numpy(), "features", tol=1e-5)
    (feat * gw).sum().backward()
    for (k, p), (_, q) in zip(mods[1].named_parameters(), mods[0].named_parameters()):
        _close(p.grad, q.grad.cpu().numpy(), f"grad {k}", tol=1e-4)
    for (k, v), (_, w) in zip(mods[1].state_dict().items(), mods[0].state_dict().items()):
        if v.dtype.is_floating_point:
            _close(v, w.cpu().numpy(), f"state {k}", tol=1e-5)


@pytest.mark.parametrize("dedup", ["1", "0"])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("impl", ["mfma", "fused"])
def test_fused_shared_mlp_matches_layer_path(dev, monkeypatch, train, dedup, impl):
    """The fused channels-last shared MLP + BN + ReLU + max-pool -- "mfma": the hand-written fp32-MFMA pipeline
    (csrc/sa_mfma.hip), "fused": the streaming kernels of csrc/sa_mlp.hip around library GEMMs -- against the
    per-layer Conv2d/BatchNorm2d/ReLU/max_pool2d path of the same module, same weights, at the pre-encoder's
    widths [3,64,128,256]: outputs, every parameter gradient and the BatchNorm running statistics."""
    import os

    from coda_neurips2023_amd.synthetic_scenes import make_batch
    pc, _, _ = make_batch(2, 8192, seed=77)
    xyz = torch.from_numpy(pc).to(dev)

    def run(kind):
        monkeypatch.setenv("CODA_SA_MLP", kind)
        monkeypatch.setenv("CODA_SA_DEDUP", dedup)  # padded copies of a group computed once / every row
        torch.manual_seed(3)
        mod = pointnet2_modules.PointnetSAModuleVotes(mlp=[0, 64, 128, 256], npoint=512, radius=0.2,
                                                      nsample=64, normalize_xyz=True).to(dev)
        with torch.no_grad():
            for k, p in mod.named_parameters():
                if "bn" in k:
                    p.copy_(torch.rand_like(p) + 0.5 if k.endswith("weight") else torch.randn_like(p) * 0.1)
            for k, b in mod.named_buffers():
                if k.endswith("running_var"):
                    b.copy_(torch.rand_like(b) + 0.5)
                if k.endswith("running_mean"):
                    b.copy_(torch.randn_like(b) * 0.1)
        mod.train(train)
        _, feat, _ = mod(xyz)
        gw = torch.randn(feat.shape, generator=torch.Generator().manual_seed(4)).to(dev)
        (feat * gw).sum().backward()
        return feat.detach(), {k: p.grad for k, p in mod.named_parameters()}, \
            {k: v.clone() for k, v in mod.state_dict().items() if "running" in k or "tracked" in k}

    f1, g1, s1 = run(impl)
    f2, g2, s2 = run("layers")
    assert f1.shape == f2.shape == (2, 256, 512)
    assert float((f1 - f2).abs().max() / f2.abs().max()) < 1e-5
    for k in g1:
        err = float((g1[k] - g2[k]).abs().max() / g2[k].abs().max())
        assert err < 1e-3, f"{k}: {err:.3e}"
    for k in s1:
        assert torch.allclose(s1[k].float(), s2[k].float(), rtol=1e-5, atol=1e-6), k


def test_group_compaction(dev):
    """compact_groups: distinct rows per group, multiplicities summing to nsample, offsets; dense
    groups (no copies) are left alone."""
    from coda_neurips2023_amd.pointnet2 import _ext, fused_sa_mlp
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    pc, _, _ = make_batch(2, 6000, seed=5)
    xyz = torch.from_numpy(pc).to(dev)
    inds = _ext.furthest_point_sampling(xyz, 256)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx, grouped = _ext.query_and_group_xyz(new_xyz, xyz, 0.2, 64, True, channels_last=True)
    out = fused_sa_mlp.compact_groups(idx, grouped)
    assert out is not None
    x, w, goff = out
    g = 2 * 256
    assert x.shape[0] % 2048 == 0 and goff.shape == (g + 1,) and goff.dtype == torch.int32
    total = int(goff[-1])
    idx2 = idx.view(g, 64).cpu()
    grouped2 = grouped.view(g, 64, 3).cpu()
    goff_c, x_c, w_c = goff.cpu(), x.cpu(), w.cpu()
    for gi in [0, 1, 17, 255, 256, g - 1]:
        cnt = len(set(idx2[gi].tolist()))
        a, b = int(goff_c[gi]), int(goff_c[gi + 1])
        assert b - a == cnt
        assert torch.equal(x_c[a:b], grouped2[gi, :cnt])
        assert float(w_c[a:b].sum()) == 64.0 and float(w_c[a]) == 64 - cnt + 1
    assert float(w_c[total:].abs().sum()) == 0.0 and float(x_c[total:].abs().sum()) == 0.0
    # radius large enough that every group is full: nothing to de-duplicate
    idx_full, grouped_full = _ext.query_and_group_xyz(new_xyz, xyz, 5.0, 16, True, channels_last=True)
    assert fused_sa_mlp.compact_groups(idx_full, grouped_full) is None
