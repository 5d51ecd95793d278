"""Fused GenericMLP stacks (batched GEMMs + csrc/token_bn.hip) against the module path the
reference runs (Conv1d -> BatchNorm1d -> ReLU -> Dropout per head, models/helpers.py:45-112,
models/model_3detr.py:1617-1660): outputs, parameter / input gradients and BN running
statistics within 1e-3 relative (north_star tolerance); dropout checked through its
statistics and the consistency of the regenerated backward mask."""
import copy
from functools import partial

import numpy as np
import pytest
import torch

from coda_neurips2023_amd import _lib, fused_bn_mlp
from coda_neurips2023_amd.helpers import GenericMLP

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _close(got, ref, what, rtol=RTOL):
    got, ref = got.detach().double().cpu().numpy(), ref.detach().double().cpu().numpy()
    err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
    assert err < rtol, f"{what}: max err / max|ref| = {err:.3e}"


def _heads(dev, outs, hidden, cin, dropout):
    torch.manual_seed(3)
    mk = partial(GenericMLP, norm_fn_name="bn1d", activation="relu", use_conv=True, hidden_dims=hidden,
                 dropout=dropout, input_dim=cin)
    heads = [mk(output_dim=o).to(dev) for o in outs]
    for h in heads:  # non-trivial BN affine parameters
        for m in h.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                torch.nn.init.uniform_(m.weight, 0.5, 1.5)
                torch.nn.init.uniform_(m.bias, -0.3, 0.3)
    return heads


@pytest.mark.parametrize("outs,hidden,cin,tokens", [((2, 512, 3, 3, 12, 12), [256, 256], 256, (4, 64, 8)),
                                                     ((5, 7), [64, 32], 48, (2, 33, 3)),
                                                     ((16,), [128], 32, (1, 200, 2))])
def test_heads_match_module_path(dev, outs, hidden, cin, tokens):
    nl, nq, b = tokens
    heads = _heads(dev, outs, hidden, cin, dropout=0.0)
    # reference: the module path in float64 (fp32 conv/BN backward of either implementation carries
    # ~1e-3 cancellation noise in the summed input gradient; fp64 is the arbiter)
    ref_heads = [copy.deepcopy(h).double() for h in heads]
    x = torch.randn(nl, nq, b, cin, device=dev, requires_grad=True)
    x_ref = x.detach().double().requires_grad_(True)

    parsed = fused_bn_mlp.eligible(heads, x)
    assert parsed is not None
    got = [o.view(nl, nq, b, -1).permute(0, 2, 1, 3) for o in fused_bn_mlp.run_stacks(x.reshape(-1, cin), parsed)]
    feats = x_ref.permute(0, 2, 3, 1).reshape(nl * b, cin, nq)
    ref = [h(feats).transpose(1, 2).reshape(nl, b, nq, -1) for h in ref_heads]

    loss = loss_ref = 0
    for g, (a, r) in enumerate(zip(got, ref)):
        _close(a, r, f"head {g} output")
        w = torch.randn_like(a)
        loss = loss + (a * w).sum()
        loss_ref = loss_ref + (r * w.double()).sum()
    loss.backward()
    loss_ref.backward()
    _close(x.grad, x_ref.grad, "input gradient")
    for g, (h, hr) in enumerate(zip(heads, ref_heads)):
        for (k, p), (_, pr) in zip(h.named_parameters(), hr.named_parameters()):
            _close(p.grad, pr.grad, f"head {g} grad {k}")
        for (k, v), (_, vr) in zip(h.named_buffers(), hr.named_buffers()):
            if v.dtype.is_floating_point:
                _close(v, vr, f"head {g} buffer {k}")
            else:
                assert int(v) == int(vr), k


def test_projection_stack_without_tail(dev):
    """encoder_to_decoder_projection: three conv+BN+ReLU blocks, the last one IS the output."""
    torch.manual_seed(5)
    mlp = GenericMLP(input_dim=64, hidden_dims=[128, 128], output_dim=64, norm_fn_name="bn1d", activation="relu",
                     use_conv=True, output_use_activation=True, output_use_norm=True, output_use_bias=False).to(dev)
    ref_mlp = copy.deepcopy(mlp).double()
    x = torch.randn(300, 4, 64, device=dev, requires_grad=True)  # (npoints, B, C)
    x_ref = x.detach().double().requires_grad_(True)
    parsed = fused_bn_mlp.eligible([mlp], x)
    assert parsed is not None and parsed[0][1] is None and len(parsed[0][0]) == 3
    got = fused_bn_mlp.run_stacks(x.reshape(-1, 64), parsed).view(300, 4, 64)
    ref = ref_mlp(x_ref.permute(1, 2, 0)).permute(2, 0, 1)
    _close(got, ref, "projection output")
    w = torch.randn_like(got)
    (got * w).sum().backward()
    (ref * w.double()).sum().backward()
    _close(x.grad, x_ref.grad, "input gradient")
    for (k, p), (_, pr) in zip(mlp.named_parameters(), ref_mlp.named_parameters()):
        _close(p.grad, pr.grad, f"grad {k}")


def test_eval_mode_and_odd_shapes_use_the_module_path(dev):
    heads = _heads(dev, (3,), [64], 32, dropout=0.1)
    x = torch.randn(2, 5, 3, 32, device=dev)
    heads[0].eval()
    assert fused_bn_mlp.eligible(heads, x) is None
    heads[0].train()
    assert fused_bn_mlp.eligible(heads, x) is not None
    odd = GenericMLP(32, [30], 3, norm_fn_name="bn1d", use_conv=True).to(dev)  # 30 % 4 != 0
    assert fused_bn_mlp.eligible([odd], x) is None
    no_norm = GenericMLP(32, [32], 3, use_conv=True, hidden_use_bias=True).to(dev)
    assert fused_bn_mlp.eligible([no_norm], x) is None


def test_dropout_statistics_and_backward_mask(dev):
    """a = dropout(relu(z*1+0)) on positive z: the zero fraction is p, survivors are scaled by
    1/(1-p), a second seed draws another mask, and the backward regenerates the same mask."""
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g, r, c, p = 3, 4096, 256, 0.3
    z = torch.rand(g, r, c, device=dev) + 0.5
    prm = torch.zeros(g, 4, c, device=dev)
    prm[:, 0] = 1.0  # scale
    prm[:, 3] = 1.0  # invstd
    a = torch.empty_like(z)
    _lib.check(lib.coda_tok_bn_act_f32(z.data_ptr(), prm.data_ptr(), g, r, c, 1, p, 1234, None, a.data_ptr(), st), "act")
    frac = float((a == 0).float().mean())
    assert abs(frac - p) < 5e-3, frac
    kept = a != 0
    assert torch.allclose(a[kept], z[kept] / (1 - p), rtol=1e-6)
    per_group = (a == 0).float().mean(dim=(1, 2))
    assert float((per_group - p).abs().max()) < 1e-2
    a2 = torch.empty_like(z)
    _lib.check(lib.coda_tok_bn_act_f32(z.data_ptr(), prm.data_ptr(), g, r, c, 1, p, 99, None, a2.data_ptr(), st), "act")
    assert 0.35 < float(((a == 0) != (a2 == 0)).float().mean()) < 0.49  # 2p(1-p) = 0.42
    # backward with zero batch-mean terms: dz = coef_a * d, d = da * keep/(1-p)
    prmb = torch.zeros(g, 3, c, device=dev)
    prmb[:, 0] = 1.0
    da = torch.ones_like(z)
    dz = torch.empty_like(z)
    _lib.check(lib.coda_tok_bn_act_bwd_apply_f32(da.data_ptr(), z.data_ptr(), prm.data_ptr(), prmb.data_ptr(), g, r,
                                                 c, 1, p, 1234, None, dz.data_ptr(), st), "bwd_apply")
    assert torch.equal(dz != 0, kept)
    assert torch.allclose(dz[kept], torch.full_like(dz[kept], 1 / (1 - p)), rtol=1e-6)


@pytest.mark.parametrize("g,r,c", [(1, 16384, 256), (6, 12288, 256), (1, 2048, 512), (2, 1001, 4), (3, 77, 1024),
                                   (1, 5, 64)])
def test_statistics_as_partial_sums(dev, g, r, c):
    """The statistics kernels leave one partial sum per row block (no atomics); the finalize kernels add them in order:
    the summed parts are the fp64 column sums, the finalized parameters those of BatchNorm1d, bit-identical run to run."""
    lib = _lib.load()
    gen = torch.Generator(device="cpu").manual_seed(g * 1000 + c)
    z = (torch.randn(g, r, c, generator=gen) * 2 + 0.5).to(dev)
    nparts = lib.coda_tok_bn_parts(g, r, c)
    assert nparts >= 1
    parts = torch.full((nparts, g, 2, c), float("nan"), dtype=torch.float64, device=dev)   # every entry must be written
    _lib.check(lib.coda_tok_bn_stats_f32(z.data_ptr(), g, r, c, parts.data_ptr(), None), "stats")
    sums = parts.sum(0)
    zd = z.double()
    assert torch.allclose(sums[:, 0], zd.sum(1), rtol=1e-6, atol=1e-6 * r)
    assert torch.allclose(sums[:, 1], (zd * zd).sum(1), rtol=1e-6)
    gamma = torch.rand(g, c, device=dev) + 0.5
    beta = torch.randn(g, c, device=dev)
    outs = []
    for src, n in ((parts, nparts), (sums.contiguous(), 1), (parts, nparts)):
        prm = torch.empty(g, 4, c, device=dev)
        stat = torch.empty(g, 2, c, device=dev)
        _lib.check(lib.coda_tok_bn_finalize_f32(src.data_ptr(), n, gamma.data_ptr(), beta.data_ptr(), g, c, float(r),
                                                1e-5, prm.data_ptr(), stat.data_ptr(), None), "finalize")
        outs.append((prm, stat))
    mean = zd.mean(1)
    var = zd.var(1, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    prm, stat = outs[0]
    assert torch.allclose(prm[:, 2].double(), mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(prm[:, 3].double(), invstd, rtol=1e-5)
    assert torch.allclose(prm[:, 0].double(), gamma.double() * invstd, rtol=1e-5)
    assert torch.allclose(prm[:, 1].double(), beta.double() - mean * gamma.double() * invstd, rtol=1e-4, atol=1e-5)
    assert torch.allclose(stat[:, 1].double(), var * (r / max(r - 1, 1)), rtol=1e-5)
    assert torch.allclose(outs[1][0], prm, rtol=1e-6, atol=1e-7)          # already reduced sums, nparts = 1
    assert torch.equal(outs[2][0], prm) and torch.equal(outs[2][1], stat)  # deterministic
    # backward statistics: sum d, sum d * xhat (no ReLU, no dropout: d = da), local parts vs an explicit total
    da = torch.randn(g, r, c, generator=gen).to(dev)
    bparts = torch.full((nparts, g, 2, c), float("nan"), dtype=torch.float64, device=dev)
    _lib.check(lib.coda_tok_bn_act_bwd_stats_f32(da.data_ptr(), z.data_ptr(), prm.data_ptr(), g, r, c, 0, 0.0, 0, None,
                                                 bparts.data_ptr(), None), "bwd_stats")
    bs = bparts.sum(0)
    xhat = (zd - prm[:, 2].double().unsqueeze(1)) * prm[:, 3].double().unsqueeze(1)
    assert torch.allclose(bs[:, 0], da.double().sum(1), rtol=1e-5, atol=1e-5 * r ** 0.5)
    assert torch.allclose(bs[:, 1], (da.double() * xhat).sum(1), rtol=1e-5, atol=1e-5 * r ** 0.5)
    res = []
    for total in (None, (2 * bs).contiguous()):
        prmb = torch.empty(g, 3, c, device=dev)
        dgamma = torch.empty(g, c, device=dev)
        dbeta = torch.empty(g, c, device=dev)
        _lib.check(lib.coda_tok_bn_bwd_finalize_f32(bparts.data_ptr(), nparts, total.data_ptr() if total is not None
                                                    else None, gamma.data_ptr(), prm.data_ptr(), g, c, float(r),
                                                    prmb.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), None), "bwd_fin")
        res.append((prmb, dgamma, dbeta))
    prmb, dgamma, dbeta = res[0]
    assert torch.allclose(dbeta.double(), bs[:, 0], rtol=1e-6, atol=1e-6)
    assert torch.allclose(dgamma.double(), bs[:, 1], rtol=1e-6, atol=1e-6)
    assert torch.allclose(prmb[:, 1].double(), bs[:, 0] / r, rtol=1e-6, atol=1e-7)
    assert torch.allclose(prmb[:, 2].double(), bs[:, 1] / r, rtol=1e-6, atol=1e-7)
    assert torch.equal(prmb[:, 0], gamma * prm[:, 3])
    assert torch.allclose(res[1][0][:, 1:], 2 * prmb[:, 1:], rtol=1e-6, atol=1e-7)   # the all-reduced totals are used
    assert torch.equal(res[1][1], dgamma) and torch.equal(res[1][2], dbeta)          # ... but not for dgamma / dbeta


def test_bad_arguments_are_rejected(dev):
    lib = _lib.load()
    z = torch.zeros(1, 8, 16, device=dev)
    sums = torch.zeros(1, 2, 16, dtype=torch.float64, device=dev)
    assert lib.coda_tok_bn_stats_f32(z.data_ptr(), 1, 8, 10, sums.data_ptr(), None) == _lib.CODA_EINVAL  # c % 4
    assert lib.coda_tok_bn_stats_f32(z.data_ptr(), 1, 8, 12, sums.data_ptr(), None) == _lib.CODA_EINVAL  # 256 % 3
    assert lib.coda_tok_bn_stats_f32(z.data_ptr(), 1, 8, 16, None, None) == _lib.CODA_EINVAL
    assert lib.coda_tok_bn_act_f32(z.data_ptr(), z.data_ptr(), 1, 8, 16, 1, 1.0, 0, None, z.data_ptr(),
                                   None) == _lib.CODA_EINVAL  # p == 1
    assert lib.coda_tok_bn_stats_f32(None, 1, 0, 16, sums.data_ptr(), None) == _lib.CODA_OK  # empty input
    assert lib.coda_tok_bn_parts(1, 0, 16) == 1 and lib.coda_tok_bn_parts(1, 8, 10) == _lib.CODA_EINVAL
