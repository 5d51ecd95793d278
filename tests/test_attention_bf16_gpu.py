"""bf16-MFMA mode of the fused attention core (mfma_dtype = 1 of coda_mha_*_opt_f32, csrc/attention_bf16.hip;
BASELINE.json configs[4]) against the plain torch fp32 reference of the same op.

Tolerance: bf16 operands carry 8 significand bits (unit round-off 2^-9 = 2e-3); with fp32
accumulation the outputs / gradients are expected within a few units of that relative to the
largest reference element.  Stated bar: 2e-2 (max error / max |reference|), and 1e-2 in the
relative L2 norm.  The dropout masks must be IDENTICAL to the fp32 mode's (same counter hash)."""
import pytest
import torch

from coda_neurips2023_amd import attention_core
from oracle.cpu_port import attention_ref, attention_ref_bf16
from tests.test_attention_gpu import make_qkv, rel

pytestmark = pytest.mark.gpu
TOL_MAX, TOL_L2 = 2e-2, 1e-2


@pytest.fixture(autouse=True)
def bf16_mode():
    with attention_core.mfma_dtype("bf16"):
        assert attention_core.get_mfma_dtype() == "bf16"
        yield
    assert attention_core.get_mfma_dtype() == "fp32"


def l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("l,s,b,h,d,packed,masked", [
    (2048, 2048, 2, 4, 64, True, False),    # encoder self-attention
    (512, 512, 2, 4, 64, True, False),      # decoder self-attention, 512 queries (configs[4])
    (512, 2048, 2, 4, 64, False, False),    # decoder cross-attention, 512 queries
    (256, 2048, 8, 4, 64, False, False),    # 32 batch*heads (XCD-aware grid)
    (100, 77, 3, 2, 64, False, True),       # ragged + mask
    (33, 31, 1, 1, 64, False, False),
    (1024, 1024, 1, 4, 64, True, True),     # long masked
    (1100, 1060, 1, 2, 64, False, False),   # long ragged (partial stages and tiles)
    (40, 160, 2, 4, 128, False, False),     # dec_dim 512
    (128, 128, 2, 4, 128, True, True),
    (1024, 1024, 1, 2, 128, True, False),
    (96, 64, 3, 4, 64, False, False),       # 12 batch*heads: plain grid
    (512, 2048, 8, 4, 64, False, False),    # configs[4]'s cross-attention at full batch: two query tiles per workgroup
    (544, 1280, 8, 4, 64, False, True),     # ... ragged last pair, mask
    (300, 700, 2, 2, 64, False, True),      # 512 <= keys < 1024: dK/dV with the keys owned per wave, ragged + mask
])
def test_bf16_forward_backward_match_fp32_reference(dev, l, s, b, h, d, packed, masked):
    leaves, q, k, v = make_qkv(dev, l, s, b, h, d, packed, seed=l + s)
    scale = d ** -0.5
    mask = None
    if masked:
        mask = torch.rand(b, h, l, s, device=dev) < 0.3
        mask[..., 0] = False
    out, _ = attention_core.attention(q, k, v, mask, scale, 0.0, False)
    ref, _ = attention_ref(q, k, v, mask, scale, 0.0, False)
    assert rel(out, ref) < TOL_MAX and l2(out, ref) < TOL_L2, (rel(out, ref), l2(out, ref))
    gw = torch.randn(out.shape, device=dev)
    grads = torch.autograd.grad((out * gw).sum(), leaves)
    grads_ref = torch.autograd.grad((ref * gw).sum(), leaves)
    for g, gr in zip(grads, grads_ref):
        assert rel(g, gr) < TOL_MAX and l2(g, gr) < TOL_L2, (rel(g, gr), l2(g, gr))


def test_bf16_dropout_mask_is_the_fp32_modes(dev):
    l, s, b, h, d, p = 96, 160, 2, 4, 64, 0.3
    g = torch.Generator().manual_seed(11)
    q = torch.randn(l, b, h, d, generator=g).to(dev)
    k = torch.randn(s, b, h, d, generator=g).to(dev)
    eye = torch.zeros(s, b, h, d)
    eye[torch.arange(64), :, :, torch.arange(64)] = 1.0  # V = [I; 0]: the output shows P's first 64 columns
    eye = eye.to(dev)
    torch.manual_seed(5)
    a_bf16, _ = attention_core.attention(q, k, eye, None, d ** -0.5, p, False)
    torch.manual_seed(5)
    with attention_core.mfma_dtype("fp32"):
        a_fp32, _ = attention_core.attention(q, k, eye, None, d ** -0.5, p, False)
    assert torch.equal(a_bf16 != 0, a_fp32 != 0)
    assert rel(a_bf16, a_fp32) < TOL_MAX


def test_bf16_dropout_backward_consistency(dev):
    """Backward regenerates the forward's mask: compare with autograd through the explicit masked formula."""
    l, s, b, h, d, p = 256, 2048, 2, 4, 64, 0.1
    g = torch.Generator().manual_seed(12)
    q = torch.randn(l, b, h, d, generator=g).to(dev).requires_grad_(True)
    k = torch.randn(s, b, h, d, generator=g).to(dev).requires_grad_(True)
    v = torch.randn(s, b, h, d, generator=g).to(dev).requires_grad_(True)
    scale = d ** -0.5
    # recover the keep mask 64 key columns at a time with one-hot V blocks
    keep = torch.empty(l, b, h, s, device=dev)
    with torch.no_grad():
        for c0 in range(0, s, d):
            sel = torch.zeros(s, b, h, d, device=dev)
            sel[torch.arange(c0, c0 + d), :, :, torch.arange(d)] = 1.0
            torch.manual_seed(77)
            a, _ = attention_core.attention(q, k, sel, None, scale, p, False)
            keep[..., c0:c0 + d] = (a != 0).float() / (1 - p)
    assert abs(float((keep != 0).float().mean()) - (1 - p)) < 0.01
    torch.manual_seed(77)
    out, _ = attention_core.attention(q, k, v, None, scale, p, False)
    scores = torch.einsum("lbhd,sbhd->lbhs", q * scale, k)
    ref = torch.einsum("lbhs,sbhd->lbhd", torch.softmax(scores, -1) * keep, v)
    assert rel(out, ref) < TOL_MAX
    gw = torch.randn(out.shape, device=dev)
    grads = torch.autograd.grad((out * gw).sum(), (q, k, v))
    grads_ref = torch.autograd.grad((ref * gw).sum(), (q, k, v))
    for a, r in zip(grads, grads_ref):
        assert rel(a, r) < TOL_MAX and l2(a, r) < TOL_L2


def test_bf16_fully_masked_rows_give_zero(dev):
    _, q, k, v = make_qkv(dev, 40, 50, 1, 2, 64, False, seed=4)
    mask = torch.zeros(1, 2, 40, 50, dtype=torch.bool, device=dev)
    mask[:, :, 7] = True
    out, _ = attention_core.attention(q, k, v, mask, 0.125, 0.0, False)
    assert torch.isfinite(out).all() and out[7].abs().max() == 0
    (gq,) = torch.autograd.grad(out.sum(), q)
    assert torch.isfinite(gq).all()


# Against the bf16-ROUNDING oracle (oracle/cpu_port.attention_ref_bf16: every matrix-product operand rounded to bf16
# where the kernels round it -- Q pre-scaled in the forward / dQ kernels, unscaled in the dK/dV kernel -- float32
# accumulation) the roundings of Q, K, V and dO are IDENTICAL on both sides; what is left is the rounding of the
# probabilities, which the kernels take against the RUNNING row maximum of their key tiles and the oracle against the
# final one: two independent 2^-9 roundings per probability, i.e. a relative L2 distance of ~sqrt(2) * 2^-9 / sqrt(3)
# = 1.6e-3 on every output (measured on MI355X: 1.3e-3 .. 2.0e-3), whatever the problem size.  Bars: outputs 2e-3 of
# the largest element / 2.5e-3 in L2, gradients 4e-3 / 3e-3 -- against the fp32 reference the same kernels sit at
# 4e-3 .. 8e-3, so a dropped or doubled operand rounding, a wrong fragment or a wrong scale now shows (VERDICT r3,
# missing 4); the floor itself could only be removed by re-stating each kernel's tile order in the oracle.
ORACLE_OUT, ORACLE_GRAD = (2e-3, 2.5e-3), (4e-3, 3e-3)


@pytest.mark.parametrize("l,s,b,h,d,packed,masked", [
    (2048, 2048, 1, 4, 64, True, False),    # encoder self-attention
    (512, 512, 2, 4, 64, True, False),      # decoder self-attention, 512 queries (configs[4])
    (512, 2048, 2, 4, 64, False, False),    # decoder cross-attention, 512 queries
    (100, 77, 3, 2, 64, False, True),       # ragged + mask
    (128, 128, 2, 4, 128, True, True),      # dec_dim 512
    (300, 700, 2, 2, 64, False, True),
    (512, 2048, 8, 4, 64, False, False),    # two query tiles per workgroup (forward and dQ)
])
def test_bf16_kernels_match_the_bf16_rounding_oracle(dev, l, s, b, h, d, packed, masked):
    leaves, q, k, v = make_qkv(dev, l, s, b, h, d, packed, seed=3 * l + s)
    scale = d ** -0.5
    mask = None
    if masked:
        mask = torch.rand(b, h, l, s, device=dev) < 0.3
        mask[..., 0] = False
    out, _ = attention_core.attention(q, k, v, mask, scale, 0.0, False)
    gw = torch.randn(out.shape, device=dev)
    grads = torch.autograd.grad((out * gw).sum(), leaves)
    cl = [t.detach().cpu().requires_grad_(True) for t in leaves]
    if len(cl) == 1:   # make_qkv's packed form: one (L,B,3*h*d) leaf cut into q, k, v
        cq, ck, cv = (t.reshape(l, b, h, d) for t in cl[0].chunk(3, dim=-1))
    else:
        cq, ck, cv = cl
    ref, _ = attention_ref_bf16(cq, ck, cv, None if mask is None else mask.cpu(), scale, 0.0, False)
    grads_ref = torch.autograd.grad((ref * gw.cpu()).sum(), cl)
    worst = [(rel(out.cpu(), ref), l2(out.cpu(), ref))] + [(rel(g.cpu(), r), l2(g.cpu(), r)) for g, r in zip(grads, grads_ref)]
    print(f"bf16 kernels vs bf16 oracle ({l}x{s} d{d}): max " + ", ".join(f"{a:.1e}" for a, _ in worst) + "; L2 "
          + ", ".join(f"{b_:.1e}" for _, b_ in worst))
    assert worst[0][0] < ORACLE_OUT[0] and worst[0][1] < ORACLE_OUT[1], worst
    assert all(a < ORACLE_GRAD[0] and b_ < ORACLE_GRAD[1] for a, b_ in worst[1:]), worst
