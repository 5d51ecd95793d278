"""Token-wise glue kernels of the fused transformer layers (csrc/token_ln.hip through
fused_layers.py) against plain torch in float64, and the fused encoder / decoder containers
against the module-by-module path of the same classes (CODA_LAYERS=modules), which is the
reference's op sequence (models/transformer.py:457-494, 558-594).  Tolerance 1e-3 relative
(north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from coda_neurips2023_amd import fused_layers as fl
from coda_neurips2023_amd.transformer import (TransformerDecoder, TransformerDecoderLayer, TransformerEncoder,
                                              TransformerEncoderLayer)

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _close(got, ref, what, rtol=RTOL):
    got, ref = got.detach().double().cpu().numpy(), ref.detach().double().cpu().numpy()
    err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
    assert err < rtol, f"{what}: max err / max|ref| = {err:.3e}"


@pytest.mark.parametrize("rows,c", [((128, 8), 256), ((37, 3), 512), ((5, 2), 100), ((2048, 8), 256), ((3, 1), 1024)])
@pytest.mark.parametrize("variant", ["ln", "ln_pos", "bias_res_ln_pos", "res_only", "bias_res_ln"])
def test_add_ln_matches_torch(dev, rows, c, variant):
    torch.manual_seed(0)
    shape = (*rows, c)
    x = torch.randn(shape, device=dev, requires_grad=True)
    norm = torch.nn.LayerNorm(c).to(dev) if variant != "res_only" else None
    if norm is not None:
        torch.nn.init.uniform_(norm.weight, 0.5, 1.5)
        torch.nn.init.uniform_(norm.bias, -0.5, 0.5)
    bias = torch.randn(c, device=dev, requires_grad=True) if "bias" in variant or variant == "res_only" else None
    res = (3 * torch.randn(shape, device=dev)).requires_grad_(True) if "res" in variant else None
    pos = torch.randn(shape, device=dev, requires_grad=True) if "pos" in variant else None
    s, y, yp = fl.add_ln(x, norm=norm, bias=bias, res=res, pos=pos)

    d = lambda t: None if t is None else t.detach().double().requires_grad_(True)  # noqa: E731
    x64, b64, r64, p64 = d(x), d(bias), d(res), d(pos)
    s64 = x64 if b64 is None else x64 + b64
    s64 = s64 if r64 is None else r64 + s64
    y64 = yp64 = None
    if norm is not None:
        g64, be64 = d(norm.weight), d(norm.bias)
        y64 = F.layer_norm(s64, (c,), g64, be64, norm.eps)
        yp64 = y64 + p64 if p64 is not None else None
    loss = loss64 = 0
    for got, ref, name in [(s, s64, "s"), (y, y64, "y"), (yp, yp64, "yp")]:
        assert (got is None) == (ref is None), name
        if got is not None:
            _close(got, ref, name)
            w = torch.randn_like(got)
            loss = loss + (got * w).sum()
            loss64 = loss64 + (ref * w.double()).sum()
    loss.backward()
    loss64.backward()
    pairs = [(x, x64, "dx"), (bias, b64, "dbias"), (res, r64, "dres"), (pos, p64, "dpos")]
    if norm is not None:
        pairs += [(norm.weight, g64, "dgamma"), (norm.bias, be64, "dbeta")]
    for t, t64, name in pairs:
        if t is not None:
            _close(t.grad, t64.grad, name)


def test_add_ln_partial_use_of_outputs(dev):
    """Only y, only the stream, or neither normalised output used downstream."""
    torch.manual_seed(1)
    c = 256
    norm = torch.nn.LayerNorm(c).to(dev)
    x = torch.randn(64, 4, c, device=dev, requires_grad=True)
    res = torch.randn(64, 4, c, device=dev, requires_grad=True)
    pos = torch.randn(64, 4, c, device=dev)
    s, y, yp = fl.add_ln(x, norm=norm, res=res, pos=pos)
    (s * 2.0).sum().backward()
    assert torch.allclose(x.grad, torch.full_like(x, 2.0)) and torch.allclose(res.grad, torch.full_like(x, 2.0))
    assert norm.weight.grad is None or float(norm.weight.grad.abs().max()) == 0.0
    x.grad = res.grad = None
    s, y, yp = fl.add_ln(x, norm=norm, res=res, pos=pos)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    x64 = (x + res).detach().double().requires_grad_(True)
    (F.layer_norm(x64, (c,), norm.weight.detach().double(), norm.bias.detach().double(), norm.eps)
     * w.double()).sum().backward()
    _close(x.grad, x64.grad, "dx through y only", rtol=2e-3)


def test_add_ln_dropout(dev):
    torch.manual_seed(2)
    c, p = 256, 0.1
    norm = torch.nn.LayerNorm(c).to(dev)
    x = (torch.rand(4096, 2, c, device=dev) + 1.0).requires_grad_(True)
    res = torch.randn(4096, 2, c, device=dev, requires_grad=True)
    s, y, _ = fl.add_ln(x, norm=norm, res=res, p=p)
    v = (s - res).detach()
    dropped = v.abs() < 1e-6
    assert abs(float(dropped.float().mean()) - p) < 3e-3
    assert torch.allclose(v[~dropped], x.detach()[~dropped] / (1 - p), rtol=1e-5, atol=1e-5)
    _close(y, F.layer_norm(s.detach(), (c,), norm.weight, norm.bias, norm.eps), "y of the dropped-out stream")
    w = torch.randn_like(s)
    (s * w).sum().backward()
    assert torch.equal(x.grad == 0, dropped)
    assert torch.allclose(x.grad[~dropped], (w / (1 - p))[~dropped], rtol=1e-5)
    assert torch.allclose(res.grad, w)
    s2, _, _ = fl.add_ln(x, norm=norm, res=res, p=p)  # a fresh mask per call
    assert 0.1 < float((((s2 - res).abs() < 1e-6) != dropped).float().mean()) < 0.25  # 2p(1-p) = 0.18


@pytest.mark.parametrize("rows,c,p", [(2048, 256, 0.0), (16384, 128, 0.0), (77, 512, 0.0), (4096, 256, 0.1)])
def test_ffn_act(dev, rows, c, p):
    torch.manual_seed(3)
    h = torch.randn(rows, c, device=dev, requires_grad=True)
    b = torch.randn(c, device=dev, requires_grad=True)
    a = fl.ffn_act(h, b, p)
    ref = torch.relu(h.detach() + b.detach())
    w = torch.randn_like(a)
    (a * w).sum().backward()
    if p == 0.0:
        assert torch.allclose(a, ref, rtol=1e-6, atol=1e-6)
        g = w * (ref > 0)
        assert torch.allclose(h.grad, g, rtol=1e-6, atol=1e-6)
        _close(b.grad, g.double().sum(0), "dbias", rtol=1e-5)
    else:
        alive = ref > 0
        dropped = alive & (a == 0)
        assert abs(float(dropped.sum()) / float(alive.sum()) - p) < 5e-3
        kept = alive & ~dropped
        assert torch.allclose(a[kept], ref[kept] / (1 - p), rtol=1e-5)
        assert torch.allclose(h.grad[kept], w[kept] / (1 - p), rtol=1e-5)
        assert float(h.grad[~kept].abs().max()) == 0.0
        _close(b.grad, h.grad.double().sum(0), "dbias", rtol=1e-5)


@pytest.mark.parametrize("g,rows,c", [(3, 16384, 256), (1, 77, 512), (2, 2048, 128), (1, 5, 16), (1, 16384, 768),
                                      (2, 333, 12), (1, 100, 1000)])
def test_colsum_kernel(dev, g, rows, c):
    from coda_neurips2023_amd.fused_layers import _colsum_into
    x = torch.randn(g, rows, c, device=dev)
    out = torch.empty(g * c, device=dev)
    _colsum_into(out, x)
    _close(out.view(g, c), x.double().sum(1), "column sums", rtol=1e-5)


def _run_encoder(dev, env, monkeypatch, src, norm):
    monkeypatch.setenv("CODA_LAYERS", env)
    torch.manual_seed(7)
    layer = TransformerEncoderLayer(d_model=256, nhead=4, dim_feedforward=128, dropout=0.0)
    enc = TransformerEncoder(layer, 3, norm=torch.nn.LayerNorm(256) if norm else None).to(dev).train()
    x = src.clone().requires_grad_(True)
    out = enc(x)[1]
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    (out * w).sum().backward()
    return out, x.grad, {k: p.grad for k, p in enc.named_parameters()}


@pytest.mark.parametrize("nodes", ["layer", "ops"])
@pytest.mark.parametrize("norm", [False, True])
def test_fused_encoder_equals_module_path(dev, monkeypatch, norm, nodes):
    monkeypatch.setenv("CODA_LAYER_NODES", nodes)  # one autograd node per layer / per block
    src = torch.randn(320, 3, 256, generator=torch.Generator().manual_seed(0)).to(dev)
    out_f, gx_f, gp_f = _run_encoder(dev, "fused", monkeypatch, src, norm)
    out_m, gx_m, gp_m = _run_encoder(dev, "modules", monkeypatch, src, norm)
    _close(out_f, out_m, "encoder output")
    _close(gx_f, gx_m, "encoder input gradient")
    for k in gp_m:
        _close(gp_f[k], gp_m[k], f"grad {k}")


def _run_decoder(dev, env, monkeypatch, tgt, memory, pos, query_pos):
    monkeypatch.setenv("CODA_LAYERS", env)
    torch.manual_seed(9)
    layer = TransformerDecoderLayer(d_model=256, nhead=4, dim_feedforward=256, dropout=0.0)
    dec = TransformerDecoder(layer, 3, return_intermediate=True).to(dev).train()
    m = memory.clone().requires_grad_(True)
    qp = query_pos.clone().requires_grad_(True)
    out = dec(tgt, m, pos=pos, query_pos=qp)[0]
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(2)).to(dev)
    (out * w).sum().backward()
    return out, m.grad, qp.grad, {k: p.grad for k, p in dec.named_parameters()}


@pytest.mark.parametrize("nodes", ["stack", "layer", "ops"])
def test_fused_decoder_equals_module_path(dev, monkeypatch, nodes):
    # one autograd node for the whole decoder / per layer / per block
    monkeypatch.setenv("CODA_LAYER_NODES", "ops" if nodes == "ops" else "layer")
    monkeypatch.setenv("CODA_DECODER_NODE", "stack" if nodes == "stack" else "layers")
    gen = torch.Generator().manual_seed(4)
    nq, nmem, b = 64, 300, 3
    tgt = torch.zeros(nq, b, 256, device=dev)
    memory = torch.randn(nmem, b, 256, generator=gen).to(dev)
    pos = torch.randn(nmem, b, 256, generator=gen).to(dev)
    query_pos = torch.randn(nq, b, 256, generator=gen).to(dev)
    out_f, gm_f, gq_f, gp_f = _run_decoder(dev, "fused", monkeypatch, tgt, memory, pos, query_pos)
    out_m, gm_m, gq_m, gp_m = _run_decoder(dev, "modules", monkeypatch, tgt, memory, pos, query_pos)
    assert out_f.shape == out_m.shape == (3, nq, b, 256)
    _close(out_f, out_m, "decoder outputs")
    _close(gm_f, gm_m, "memory gradient")
    _close(gq_f, gq_m, "query_pos gradient")
    for k in gp_m:
        _close(gp_f[k], gp_m[k], f"grad {k}")


@pytest.mark.parametrize("rows,c,p", [(2048, 256, 0.0), (2048, 256, 0.1), (300, 512, 0.0), (37, 100, 0.0), (17, 1024, 0.0)])
def test_two_layer_norms_of_one_stream(dev, rows, c, p):
    """coda_tok_add_ln_fwd2_f32 / _bwd2_f32 (the decoder's layer-output norm + the next layer's norm1 in one pass, with
    the positional embedding's gradient folded in) against float64 torch: s = res + drop(x + bias); y = LN_a(s);
    y2 = LN_b(s); yp2 = y2 + pos."""
    from coda_neurips2023_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(rows + c)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev)  # noqa: E731
    x, res, pos = rnd(rows, c), 3 * rnd(rows, c), rnd(rows, c)
    bias, ga, ba, gb, bb = rnd(c), 1 + 0.3 * rnd(c), rnd(c), 1 + 0.3 * rnd(c), rnd(c)
    s = torch.empty(rows, c, device=dev); y = torch.empty_like(s); y2 = torch.empty_like(s); yp2 = torch.empty_like(s)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    st = _lib.current_stream_handle()
    _lib.check(lib.coda_tok_add_ln_fwd2_f32(P(x), P(bias), P(res), None, P(ga), P(ba), P(gb), P(bb), P(pos), rows, c, 1e-5, p,
                                            99, None, P(s), P(y), None, P(y2), P(yp2), P(mean), P(rstd), st), "fwd2")
    # the single-head kernel on the same inputs: identical s (same dropout mask), y, mean, rstd
    s1 = torch.empty_like(s); y1 = torch.empty_like(s); m1 = torch.empty_like(mean); r1 = torch.empty_like(rstd)
    _lib.check(lib.coda_tok_add_ln_fwd_f32(P(x), P(bias), P(res), None, P(ga), P(ba), rows, c, 1e-5, p, 99, None, P(s1), P(y1),
                                           None, P(m1), P(r1), st), "fwd")
    assert torch.equal(s, s1) and torch.equal(y, y1) and torch.equal(mean, m1) and torch.equal(rstd, r1)
    keep = ((s - res) != 0).double() if p > 0 else None  # the mask, read off the result (x + bias is never exactly 0)
    d64 = lambda t: t.double().requires_grad_(True)  # noqa: E731
    x64, b64, r64, p64, ga64, ba64, gb64, bb64 = map(d64, (x, bias, res, pos, ga, ba, gb, bb))
    v64 = x64 + b64
    if keep is not None:
        v64 = v64 * keep / (1.0 - p)
    s64 = r64 + v64
    ya64 = F.layer_norm(s64, (c,), ga64, ba64, 1e-5)
    yb64 = F.layer_norm(s64, (c,), gb64, bb64, 1e-5)
    _close(s, s64, "s"); _close(y, ya64, "y"); _close(y2, yb64, "y2"); _close(yp2, yb64 + p64, "yp2")
    # backward: upstream dy (through y), dy2 (through y2), dyp2 (through yp2), ds (the stream's own), + an extra tensor
    # and a running total for the positional gradient
    dy, dy2, dyp2, ds, extra, acc0 = rnd(rows, c), rnd(rows, c), rnd(rows, c), rnd(rows, c), rnd(rows, c), rnd(rows, c)
    (ya64 * dy.double()).sum().backward(retain_graph=True)
    (yb64 * dy2.double() + (yb64 + p64) * dyp2.double() + s64 * ds.double()).sum().backward()
    blocks = lib.coda_tok_add_ln_bwd_blocks(rows, c)
    pa = torch.full((blocks, 3, c), float("nan"), device=dev); pb = torch.full((blocks, 3, c), float("nan"), device=dev)
    dres = torch.empty_like(s); dx = torch.empty_like(s)
    for init in (1, 0):
        acc = acc0.clone()
        _lib.check(lib.coda_tok_add_ln_bwd2_f32(P(dy), None, P(dy2), P(dyp2), P(ds), P(s), P(mean), P(rstd), P(ga), P(gb), rows, c,
                                                p, 99, None, 2, P(extra), P(acc), init, P(dres), P(dx), P(pa), P(pb), st), "bwd2")
        want = dyp2.double() + extra.double() + (0 if init else acc0.double())
        _close(acc, want, f"dpos (init {init})", 1e-6)
    _close(dres, r64.grad, "dres"); _close(dx, x64.grad, "dx")
    sa, sb = pa.sum(0), pb.sum(0)
    _close(sa[0], ga64.grad, "dgamma a"); _close(sa[1], ba64.grad, "dbeta a"); _close(sa[2], b64.grad, "dbias")
    _close(sb[0], gb64.grad, "dgamma b"); _close(sb[1], bb64.grad, "dbeta b")
    assert float(sb[2].abs().max()) == 0.0
    # one head + the positional fold (the first decoder layer's norm1): equals the plain backward bit for bit
    pc = torch.empty_like(pa); pd = torch.empty_like(pa); dres1 = torch.empty_like(s); dres2 = torch.empty_like(s)
    acc = torch.empty_like(s)
    _lib.check(lib.coda_tok_add_ln_bwd2_f32(P(dy2), P(dyp2), None, None, P(ds), P(s), P(mean), P(rstd), P(gb), None, rows, c, 0.0,
                                            0, None, 1, P(extra), P(acc), 1, P(dres1), None, P(pc), None, st), "bwd2 one head")
    _lib.check(lib.coda_tok_add_ln_bwd_f32(P(dy2), P(dyp2), P(ds), P(s), P(mean), P(rstd), P(gb), rows, c, 0.0, 0, None, P(dres2),
                                           None, P(pd), None, st), "bwd")
    assert torch.equal(dres1, dres2) and torch.equal(pc, pd) and torch.equal(acc, dyp2 + extra)


@pytest.mark.parametrize("m,n,k,p", [(2048, 256, 256, 0.1), (2048, 128, 256, 0.0), (256, 512, 128, 0.3), (1024, 64, 384, 0.1)])
def test_ffn_activation_backward_in_the_product_epilogue(dev, m, n, k, p):
    """coda_sgemm_relu_dropout_bwd_f32 = the product da . w followed by coda_tok_bias_relu_dropout_bwd_f32, in one launch:
    dz against float64, the bias gradient's partials against the column sums of dz."""
    from coda_neurips2023_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    da = torch.randn(m, k, generator=g).to(dev)
    w = (torch.randn(k, n, generator=g) / k ** 0.5).to(dev)
    act = torch.relu(torch.randn(m, n, generator=g)).to(dev)
    act = act * (torch.rand(m, n, generator=g).to(dev) >= p) / (1.0 - p) if p > 0 else act
    blocks = lib.coda_sgemm_relu_dropout_bwd_blocks(m)
    assert blocks == m // 32
    dz = torch.full((m, n), float("nan"), device=dev)
    parts = torch.full((blocks, n), float("nan"), device=dev)
    _lib.check(lib.coda_sgemm_relu_dropout_bwd_f32(m, n, k, da.data_ptr(), k, w.data_ptr(), n, act.data_ptr(), p, dz.data_ptr(),
                                                   parts.data_ptr(), _lib.current_stream_handle()), "sgemm_relu_dropout_bwd")
    ref = (da.double() @ w.double()) / (1.0 - p) * (act > 0)
    _close(dz, ref, "dz", 1e-5)
    assert torch.equal(dz == 0, ~(act > 0) | (ref == 0).to(dz.device))
    _close(parts.double().sum(0), dz.double().sum(0), "dbias", 1e-5)
    _close(parts.view(blocks, n), dz.view(blocks, 32, n).sum(1), "partials", 1e-5)
    assert lib.coda_sgemm_relu_dropout_bwd_f32(100, n, k, da.data_ptr(), k, w.data_ptr(), n, act.data_ptr(), p, dz.data_ptr(),
                                               parts.data_ptr(), _lib.current_stream_handle()) == _lib.CODA_ENOSPC
