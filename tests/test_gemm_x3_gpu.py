"""coda_gemm_x3_f32 (csrc/gemm_x3.hip): fp32 GEMM evaluated on the bf16 matrix cores with every operand split into
three exact bf16 pieces and all nine piece products accumulated in fp32.

The claim to check is that nothing of fp32 is lost: against a FLOAT64 evaluation of the same product the error must
be of the size of a native fp32 GEMM's (torch.mm = hipBLASLt fp32) on the same inputs -- stated bar: within 1.5x of
it, on well-scaled random operands, on operands with 12 decades of dynamic range, and on sums that cancel to 1e-6 of
their terms (where any dropped operand bit would show)."""
import pytest
import torch

from coda_neurips2023_amd import gemm

pytestmark = pytest.mark.gpu


def err(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


def operands(kind, m, n, k, gen, dev):
    a = torch.randn(m, k, generator=gen, dtype=torch.float64)
    b = torch.randn(k, n, generator=gen, dtype=torch.float64)
    if kind == "range":      # 12 decades of dynamic range along k
        scale = torch.logspace(-6, 6, k, dtype=torch.float64)
        a = a * scale[None, :]
        b = b / scale[:, None]
    elif kind == "cancel":   # pairs of terms that cancel to ~1e-6 of their size
        a[:, 1::2] = -a[:, 0::2] * (1 + 1e-6 * torch.randn(m, k // 2, generator=gen, dtype=torch.float64))
        b[1::2] = b[0::2]
    return a.float().to(dev), b.float().to(dev)


@pytest.mark.parametrize("kind", ["normal", "range", "cancel"])
@pytest.mark.parametrize("m,n,k", [(16384, 256, 256), (16384, 768, 256), (16384, 128, 256), (2048, 256, 256),
                                   (256, 256, 16384), (2048, 2048, 256), (64, 64, 32), (192, 320, 96)])
def test_error_against_float64_is_native_fp32_sized(dev, kind, m, n, k):
    gen = torch.Generator().manual_seed(m + n + k)
    a, b = operands(kind, m, n, k, gen, dev)          # a (m,k), b (k,n): fp32 values
    ref = a.double() @ b.double()
    native = err(torch.mm(a, b), ref)
    layouts = {
        "nn": (0, 0, a, b),
        "nt": (0, 1, a, b.t().contiguous()),
        "tn": (1, 0, a.t().contiguous(), b),
        "tt": (1, 1, a.t().contiguous(), b.t().contiguous()),
    }
    for name, (ta, tb, aa, bb) in layouts.items():
        out = gemm.gemm_x3(ta, tb, m, n, k, aa, bb)
        assert out is not None, name
        e = err(out, ref)
        assert e < 2.0 * native + 1e-7, (name, kind, e, native)


def test_bias_accumulate_strides_and_determinism(dev):
    gen = torch.Generator().manual_seed(1)
    m, n, k = 2048, 256, 256
    x = torch.randn(m, k, generator=gen).to(dev)
    packed_w = torch.randn(3 * n, k, generator=gen).to(dev)       # rows of a packed in_proj weight
    bias = torch.randn(n, generator=gen).to(dev)
    out = torch.full((m, 3 * n), float("nan"), device=dev)
    for j in range(3):                                              # column slices of a packed output
        r = gemm.gemm_x3(0, 1, m, n, k, x, packed_w[j * n:(j + 1) * n], out=out[:, j * n:(j + 1) * n], bias=bias)
        assert r is not None
    ref = (x.double() @ packed_w.double().t()) + bias.double().repeat(3)
    assert err(out, ref) < 1e-6
    acc = out.clone()
    gemm.gemm_x3(0, 1, m, n, k, x, packed_w[:n], out=acc[:, :n], accumulate=True)
    assert err(acc[:, :n], 2 * ref[:, :n] - bias.double()) < 1e-6
    # split-K weight gradient: deterministic, accumulating
    dy = torch.randn(16384, 256, generator=gen).to(dev)
    xx = torch.randn(16384, 256, generator=gen).to(dev)
    g1 = gemm.gemm_x3(1, 0, 256, 256, 16384, dy, xx)
    g2 = gemm.gemm_x3(1, 0, 256, 256, 16384, dy, xx)
    assert torch.equal(g1, g2)
    assert err(g1, dy.double().t() @ xx.double()) < 1e-6
    g3 = gemm.gemm_x3(1, 0, 256, 256, 16384, dy, xx, out=g1.clone(), accumulate=True)
    assert err(g3, 2 * (dy.double().t() @ xx.double())) < 1e-6


def test_shapes_outside_the_kernel_are_declined(dev):
    a = torch.randn(100, 64, device=dev)
    b = torch.randn(64, 64, device=dev)
    assert gemm.gemm_x3(0, 0, 100, 64, 64, a, b) is None
