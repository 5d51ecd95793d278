"""Three-piece mode of the fused attention core (mfma_dtype = 2 of coda_mha_*_opt_f32, csrc/attention_bf16.hip with NS = 3):
every fp32 operand as hi + mid + lo bf16 pieces, six piece products per product, fp32 accumulation.

The claim to check is "fp32-level results": on the same inputs the error against a FLOAT64 evaluation of the op
must be of the size of the fp32-MFMA kernels' own error (same accumulation structure, products exact to 2^-24
either way).  Stated bar: each output / gradient within 2x of the fp32 kernels' error against float64, and
within the fp32 kernels' 1e-5 tolerance of the plain torch fp32 reference.  Dropout masks are the fp32 mode's."""
import pytest
import torch

from coda_neurips2023_amd import attention_core
from oracle.cpu_port import attention_ref
from tests.test_attention_gpu import make_qkv, rel

pytestmark = pytest.mark.gpu


def run(mode, q, k, v, leaves, mask, scale, gw, p=0.0):
    with attention_core.mfma_dtype(mode):
        assert attention_core.get_mfma_dtype() == mode
        torch.manual_seed(77)
        out, _ = attention_core.attention(q, k, v, mask, scale, p, False)
    assert attention_core.get_mfma_dtype() == "fp32"          # the scope is gone, the backward still runs in `mode`
    grads = torch.autograd.grad((out * gw).sum(), leaves)
    return [out.detach()] + [g.detach() for g in grads]


@pytest.mark.parametrize("l,s,b,h,d,packed,masked", [
    (2048, 2048, 2, 4, 64, True, False),    # encoder self-attention: forward and dQ three-piece, dK/dV fp32 kernel
    (256, 2048, 8, 4, 64, False, False),    # decoder shapes: fp32 kernels throughout
    (256, 256, 2, 4, 64, True, False),
    (1100, 1060, 1, 2, 64, False, False),   # long ragged (general variants)
    (1024, 1024, 1, 4, 64, True, True),     # long masked
    (100, 77, 3, 2, 64, False, True),
    (40, 160, 2, 4, 128, False, False),     # head_dim 128: no three-piece kernel, fp32 kernels throughout
])
def test_x3_error_is_fp32_sized(dev, l, s, b, h, d, packed, masked):
    leaves, q, k, v = make_qkv(dev, l, s, b, h, d, packed, seed=l + s)
    scale = d ** -0.5
    mask = None
    if masked:
        mask = torch.rand(b, h, l, s, device=dev) < 0.3
        mask[..., 0] = False
    gw = torch.randn(l, b, h, d, device=dev)
    # float64 evaluation of the same op
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    out64, _ = attention_ref(qd, kd, vd, mask, scale, 0.0, False)
    g64 = torch.autograd.grad((out64 * gw.double()).sum(), (qd, kd, vd))
    res = {}
    for mode in ("fp32", "bf16x3"):
        qq, kk, vv = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        res[mode] = run(mode, qq, kk, vv, (qq, kk, vv), mask, scale, gw)
    refs = [out64] + list(g64)
    for name, a32, ax3, r in zip(("out", "dq", "dk", "dv"), res["fp32"], res["bf16x3"], refs):
        e32 = float((a32.double() - r).abs().max() / r.abs().max())
        ex3 = float((ax3.double() - r).abs().max() / r.abs().max())
        assert ex3 < 2.0 * e32 + 2e-7, (name, e32, ex3)
        assert ex3 < 1e-5, (name, ex3)


def test_x3_dropout_masks_are_the_fp32_modes(dev):
    l, s, b, h, d, p = 256, 2048, 2, 4, 64, 0.1
    _, q, k, _ = make_qkv(dev, l, s, b, h, d, False, seed=5)
    eye = torch.eye(s, d).view(s, 1, 1, d).expand(s, b, h, d).contiguous().to(dev)
    masks = {}
    for mode in ("fp32", "bf16x3"):
        torch.manual_seed(123)
        with attention_core.mfma_dtype(mode):
            a, _ = attention_core.attention(q, k, eye, None, d ** -0.5, p, False)
        masks[mode] = a != 0
    assert torch.equal(masks["fp32"], masks["bf16x3"])
