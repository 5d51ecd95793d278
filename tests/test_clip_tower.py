"""The frozen CLIP image tower: oracle vs the reference's VisionTransformer (golden vectors),
this package's tower (coda_vit_fwd through clip_tower.VisionTransformer) vs both, in float32 and in the
reference's float16."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from golden.make_golden import TOWER_CASES, tower_images  # noqa: E402
from golden.weights import fill_deterministic  # noqa: E402

from coda_neurips2023_amd import clip_tower  # noqa: E402
from oracle import clip_tower_oracle  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_tower.npz"))


def build(name):
    res, patch, width, layers, heads, odim, n = TOWER_CASES[name]
    vit = fill_deterministic(clip_tower.VisionTransformer(res, patch, width, layers, heads, odim), seed=5).eval()
    return vit, tower_images(name, n, res), heads, patch


@pytest.mark.parametrize("name", sorted(TOWER_CASES))
def test_parameter_names_and_shapes_are_the_references(name):
    vit, _, _, _ = build(name)
    mine = ["%s %s" % (k, tuple(v.shape)) for k, v in sorted(vit.state_dict().items())]
    assert mine == list(GOLD[name + "_keys"])


@pytest.mark.parametrize("name", sorted(TOWER_CASES))
def test_oracle_matches_reference_golden(name):
    vit, images, heads, patch = build(name)
    sd = {k: v.numpy() for k, v in vit.state_dict().items()}
    cls, tok = clip_tower_oracle.forward(sd, images.numpy(), heads, patch)
    np.testing.assert_allclose(cls, GOLD[name + "_cls"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(tok, GOLD[name + "_tokens"], rtol=0, atol=2e-5)


def test_checkpoint_entry_builds_the_tower():
    vit, _, _, _ = build("small")
    sd = {"visual." + k: v for k, v in vit.state_dict().items()}
    sd["logit_scale"] = torch.tensor(4.6)
    tower = clip_tower.build_image_tower(sd, half=True)
    assert tower.dtype == torch.float16 and tower.visual.input_resolution == 64 and tower.visual.patch_size == 16
    assert tower.visual.ln_pre.weight.dtype == torch.float32  # convert_weights leaves LayerNorm in float32
    assert tower.visual.transformer.resblocks[0].attn.in_proj_weight.dtype == torch.float16
    assert not any(p.requires_grad for p in tower.parameters())
    with pytest.raises(RuntimeError, match="CPU not supported"):
        tower.encode_image(torch.zeros(1, 3, 64, 64))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TOWER_CASES))
def test_float32_tower_matches_reference(name):
    vit, images, _, _ = build(name)
    vit = vit.cuda()
    cls, tok = vit(images.cuda())
    # float32 GEMMs on the matrix cores + fp32 attention: summation order differs from the CPU run only
    np.testing.assert_allclose(cls.cpu().numpy(), GOLD[name + "_cls"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(tok.cpu().numpy(), GOLD[name + "_tokens"], rtol=0, atol=1e-4)
    only_cls = vit.embed(images.cuda(), tokens=False)
    assert torch.equal(only_cls, cls)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TOWER_CASES))
def test_float16_tower_within_half_precision_of_reference(name):
    vit, images, _, _ = build(name)
    vit = clip_tower.convert_weights(vit).cuda()
    cls, tok = vit(images.cuda())
    assert cls.dtype == torch.float16
    ref_cls, ref_tok = GOLD[name + "_cls"], GOLD[name + "_tokens"]
    # half-precision activations through <= 2 blocks: 2^-11 relative per rounding on values of order 1;
    # 2e-2 absolute on outputs whose scale is ~1 leaves room for the accumulated roundings, and the direction
    # of every embedding (what the L1 / cosine losses see) must agree to 1e-3
    np.testing.assert_allclose(cls.float().cpu().numpy(), ref_cls, rtol=0, atol=2e-2 * max(1.0, np.abs(ref_cls).max()))
    np.testing.assert_allclose(tok.float().cpu().numpy(), ref_tok, rtol=0, atol=2e-2 * max(1.0, np.abs(ref_tok).max()))
    cos = torch.nn.functional.cosine_similarity(cls.float().cpu(), torch.from_numpy(ref_cls), dim=-1)
    assert float(cos.min()) > 1 - 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n,l,heads", [(3, 197, 12), (2, 50, 4), (1, 257, 2), (5, 17, 1), (2, 128, 3), (2, 288, 2)])
def test_float16_attention_core(n, l, heads):
    from coda_neurips2023_amd import _lib
    gen = torch.Generator().manual_seed(l * 7 + n)
    w = heads * 64
    qkv = torch.randn(l, n, 3 * w, generator=gen).cuda().half()
    out = torch.empty(l, n, w, dtype=torch.float16, device="cuda")
    _lib.check(_lib.load().coda_vit_attention_f16(qkv.data_ptr(), out.data_ptr(), n, l, heads,
                                                  _lib.current_stream_handle()), "coda_vit_attention_f16")
    q, k, v = (t.float().reshape(l, n, heads, 64).permute(1, 2, 0, 3) for t in qkv.chunk(3, -1))
    ref = torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v            # (n,h,l,64) in float32 on the half inputs
    ref = ref.permute(2, 0, 1, 3).reshape(l, n, w)
    # probabilities and outputs are rounded to half once each: 2^-11 relative on |out| <= max|v|
    assert float((out.float() - ref).abs().max()) < 4e-3


@pytest.mark.gpu
def test_attention_core_rejects_long_sequences():
    from coda_neurips2023_amd import _lib
    qkv = torch.zeros(300, 1, 192, dtype=torch.float16, device="cuda")
    out = torch.empty(300, 1, 64, dtype=torch.float16, device="cuda")
    assert _lib.load().coda_vit_attention_f16(qkv.data_ptr(), out.data_ptr(), 1, 300, 1, None) == -2


@pytest.mark.gpu
def test_vit_b16_shape_runs_and_is_deterministic():
    """The reference's tower (ViT-B/16: 224 px, width 768, 12 layers, 12 heads, 512-d), seeded weights, 8 crops."""
    torch.manual_seed(0)
    tower = clip_tower.convert_weights(clip_tower.ImageTower(512, 224, 12, 768, 16)).cuda()
    x = torch.randn(8, 3, 224, 224, device="cuda")
    a = tower.encode_image(x)
    b = tower.encode_image(x)
    assert a.shape == (8, 512) and a.dtype == torch.float16 and torch.isfinite(a.float()).all()
    assert torch.equal(a, b)
    half = tower.encode_image(x[:4])      # images are independent units
    assert float((half.float() - a[:4].float()).abs().max()) < 2e-2


def test_converted_weight_cache_is_dropped_when_weights_change():
    """The C driver's converted weight copies must not outlive the weights: load_state_dict and .to() / .half() drop
    them, refresh() is the hook for loaders that write through p.data (which changes neither pointer nor version)."""
    from coda_neurips2023_amd import clip_tower
    vit = clip_tower.VisionTransformer(32, 16, 64, 1, 2, 16)
    vit._packed = ("stamp", "copies")
    vit.load_state_dict(vit.state_dict())
    assert vit._packed is None
    vit._packed = ("stamp", "copies")
    vit.half()
    assert vit._packed is None
    vit._packed = ("stamp", "copies")
    vit.refresh()
    assert vit._packed is None
