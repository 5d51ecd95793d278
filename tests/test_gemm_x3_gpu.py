"""coda_gemm_x3_nt_f32 / coda_gemm_x3_split_f32 (csrc/gemm_x3.hip): the large token-wise linear layers evaluated on the
bf16 matrix cores with every fp32 operand carried as three exact bf16 pieces and the six piece products of order <= 2
accumulated in fp32.

The claim to check is that fp32 accuracy is kept: against a FLOAT64 evaluation of the same product the error must be of
the size of a native fp32 GEMM's (torch.mm = hipBLASLt fp32) on the same inputs -- stated bar: within 2x of it -- on
well-scaled random operands, on operands with 12 decades of dynamic range, and on sums that cancel to 1e-6 of their
terms (where any dropped operand bit would show).  Then the plumbing: the weight-piece cache behind gemm.linear / gemm.mm
(row slices of a packed weight, bias, accumulate, column slices of a packed output) and its invalidation rules."""
import pytest
import torch

from coda_neurips2023_amd import gemm

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def x3_on():
    saved = (gemm._X3, gemm._X3_TN)
    gemm.set_x3(True, force=True, tn=True)
    yield
    gemm.set_x3(saved[0], tn=saved[1])


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
@pytest.mark.parametrize("m,n,k", [(16384, 256, 256), (16384, 768, 256), (16384, 128, 256), (16384, 256, 128),
                                   (16384, 2048, 256), (16384, 256, 2048), (4096, 256, 256), (4224, 64, 32),
                                   (8192, 320, 96), (16384, 512, 4096)])
def test_error_against_float64_is_native_fp32_sized(dev, kind, m, n, k):
    gen = torch.Generator().manual_seed(m + n + k)
    a, b = operands(kind, m, n, k, gen, dev)          # a (m,k), b (k,n): fp32 values
    ref = a.double() @ b.double()
    native = err(torch.mm(a, b), ref)
    # y = x W^T with W = b^T (n x k), and dx = dy W with W = b (k x n): both orientations of the weight's pieces
    w_nt = b.t().contiguous().requires_grad_(True)
    w_nn = b.clone().requires_grad_(True)
    with torch.no_grad():
        out_nt = gemm._x3_route(0, 1, m, n, k, a, w_nt, None, None, False)
        out_nn = gemm._x3_route(0, 0, m, n, k, a, w_nn, None, None, False)
    for name, out in (("nt", out_nt), ("nn", out_nn)):
        assert out is not None, name
        e = err(out, ref)
        assert e < 2.0 * native + 1e-7, (name, kind, e, native)


def test_packed_weights_bias_accumulate_and_output_slices(dev):
    gen = torch.Generator().manual_seed(1)
    m, e, k = 4096, 256, 256
    x = torch.randn(m, k, generator=gen).to(dev)
    w_in = torch.nn.Parameter(torch.randn(3 * e, k, generator=gen).to(dev))   # a packed in_proj weight
    bias = torch.randn(3 * e, generator=gen).to(dev)
    with torch.no_grad():
        # whole weight, and row slices of it written into column slices of a packed output
        full = gemm.linear(x, w_in, bias)
        out = torch.full((m, 3 * e), float("nan"), device=dev)
        for j in range(3):
            gemm.linear(x, w_in[j * e:(j + 1) * e], bias[j * e:(j + 1) * e], out=out[:, j * e:(j + 1) * e])
        ref = x.double() @ w_in.double().t() + bias.double()
        assert err(full, ref) < 1e-6 and err(out, ref) < 1e-6
        assert sum(1 for v in gemm._planes.values() if v.base is w_in) == 1   # ONE set of pieces serves the slices
        assert all(v.nt.numel() == 3 * 3 * e * k for v in gemm._planes.values() if v.base is w_in)
        # dx = dq Wq + dk Wk + dv Wv: mm with accumulate on row slices (= column slices of the transposed pieces)
        dq, dk, dv = (torch.randn(m, e, generator=gen).to(dev) for _ in range(3))
        dx = gemm.mm(dq, w_in[:e])
        gemm.mm(dk, w_in[e:2 * e], out=dx, accumulate=True)
        gemm.mm(dv, w_in[2 * e:], out=dx, accumulate=True)
        ref_dx = torch.cat([dq, dk, dv], 1).double() @ w_in.double()
        assert err(dx, ref_dx) < 1e-6


def test_pieces_follow_the_weight(dev):
    """The cache is keyed by the weight's storage; an in-place update through torch bumps the version counter, a raw
    write (this package's optimizer, ``p.data`` loaders) needs refresh_weight_planes() -- both are honoured; a tensor
    that is not a weight (an activation) never enters the cache."""
    gen = torch.Generator().manual_seed(2)
    m, n, k = 4096, 128, 64
    x = torch.randn(m, k, generator=gen).to(dev)
    w = torch.nn.Parameter(torch.randn(n, k, generator=gen).to(dev))
    with torch.no_grad():
        y0 = gemm.linear(x, w)
        assert err(y0, x.double() @ w.double().t()) < 1e-6
        w.mul_(2.0)                                  # version counter moves
        assert err(gemm.linear(x, w), 2 * (x.double() @ w.double().t()) / 2) < 1e-6
        w.data.add_(1.0)                             # behind the version counter
        stale = gemm.linear(x, w)
        gemm.refresh_weight_planes()
        fresh = gemm.linear(x, w)
        ref = x.double() @ w.double().t()
        assert err(fresh, ref) < 1e-6 and err(stale, ref) > 1e-3
        before = len(gemm._planes)
        act = torch.randn(n, k, generator=gen).to(dev)    # not a weight: library route, nothing cached
        assert err(gemm.linear(x, act), x.double() @ act.double().t()) < 1e-5
        assert len(gemm._planes) == before
        # entries that were not used for a few refreshes are dropped
        for _ in range(6):
            gemm.refresh_weight_planes()
        assert not any(v.base is w for v in gemm._planes.values())


def test_torch_optimizer_step_refreshes_the_pieces(dev):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(4096, 64, generator=gen).to(dev)
    w = torch.nn.Parameter(torch.randn(64, 64, generator=gen).to(dev))
    opt = torch.optim.SGD([w], lr=0.5)
    with torch.no_grad():
        gemm.linear(x, w)
    w.grad = torch.ones_like(w)
    opt.step()
    with torch.no_grad():
        assert err(gemm.linear(x, w), x.double() @ w.double().t()) < 1e-6


def test_shapes_outside_the_kernel_take_the_library(dev):
    gen = torch.Generator().manual_seed(4)
    w = torch.nn.Parameter(torch.randn(72, 40, generator=gen).to(dev))
    x = torch.randn(4096, 40, generator=gen).to(dev)
    with torch.no_grad():
        assert gemm._x3_route(0, 1, 4096, 72, 40, x, w, None, None, False) is None
        assert err(gemm.linear(x, w), x.double() @ w.double().t()) < 1e-5


@pytest.mark.parametrize("kind", ["normal", "range", "cancel"])
@pytest.mark.parametrize("t,co,ci,pad", [(16384, 256, 256, 0), (16384, 128, 256, 0), (16384, 256, 128, 0),
                                         (16384, 2048, 256, 64), (8192, 384, 128, 0), (16384, 768, 256, 0)])
def test_weight_gradient_partials_against_float64(dev, kind, t, co, ci, pad):
    """coda_gemm_x3_tn_f32: dW = dY^T X over the token rows as partial sums per token slice (both operands split in the
    kernel, fragments through the transposing LDS read); the sum of the partials within 2x of the library's error."""
    gen = torch.Generator().manual_seed(t + co + ci)
    a, b = operands(kind, co, ci, t, gen, dev)        # a (co, t), b (t, ci): the product a @ b = dY^T X
    dy_full = torch.zeros((t, co + pad), device=dev)
    dy = dy_full[:, :co]
    dy.copy_(a.t())
    x = b.contiguous()
    ref = dy.double().t() @ x.double()
    native = err(torch.mm(dy.t(), x), ref)
    part = gemm.x3_tn_partials(dy, x)
    assert part is not None and part.shape[1:] == (co, ci)
    got = part.double().sum(0)
    e = float((got - ref).abs().max() / ref.abs().max())
    assert e < 2.0 * native + 1e-7, (kind, e, native, part.shape)
    again = gemm.x3_tn_partials(dy, x)
    assert torch.equal(part, again)                   # deterministic
    # a caller-provided buffer decides the slice count
    buf = torch.full((16, co, ci), float("nan"), device=dev)
    assert gemm.x3_tn_partials(dy, x, out=buf) is buf
    assert float((buf.double().sum(0) - ref).abs().max() / ref.abs().max()) < 2.0 * native + 1e-7


def test_weight_gradient_shapes_outside_the_kernel_are_declined(dev):
    dy = torch.randn(16384, 192, device=dev)
    x = torch.randn(16384, 256, device=dev)
    assert gemm.x3_tn_partials(dy, x) is None
    assert gemm.x3_tn_partials(dy[:4096, :128].contiguous(), x[:4096]) is None
