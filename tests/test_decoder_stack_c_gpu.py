"""The decoder stack driven from C++ (include/coda_stack.h, csrc/decoder_stack.hip; fused_blocks._DecoderStackC)
against the same node driven launch by launch from Python (fused_blocks._DecoderStack): identical entry points in
identical order, so outputs and every gradient agree to fp32 round-off; with dropout on (op seeds are derived
differently in the two drivers) the C driver is checked for determinism and against directional finite
differences of its own forward."""
import pytest
import torch

from coda_neurips2023_amd import fused_blocks as FB
from coda_neurips2023_amd.transformer import TransformerDecoder, TransformerDecoderLayer

pytestmark = pytest.mark.gpu


def _decoder(dev, nl, ffn, p):
    torch.manual_seed(9)
    layer = TransformerDecoderLayer(d_model=256, nhead=4, dim_feedforward=ffn, dropout=p)
    return TransformerDecoder(layer, nl, return_intermediate=True).to(dev).train()


def _inputs(dev, nq, ns, b, seed=4):
    gen = torch.Generator().manual_seed(seed)
    return (torch.zeros(nq, b, 256, device=dev), torch.randn(ns, b, 256, generator=gen).to(dev),
            torch.randn(ns, b, 256, generator=gen).to(dev), torch.randn(nq, b, 256, generator=gen).to(dev))


def _run(dec, in_c, monkeypatch, tgt, memory, pos, qpos, w):
    monkeypatch.setattr(FB, "STACK_IN_C", in_c)
    dec.zero_grad(set_to_none=True)
    m, qp = memory.clone().requires_grad_(True), qpos.clone().requires_grad_(True)
    out = dec(tgt, m, pos=pos, query_pos=qp)[0]
    (out * w).sum().backward()
    return out.detach(), m.grad, qp.grad, {k: p.grad.clone() for k, p in dec.named_parameters()}


def _close(a, b, what, rtol=2e-5):
    err = float((a - b).abs().max() / (b.abs().max() + 1e-20))
    assert err < rtol, (what, err)


@pytest.mark.parametrize("nl,nq,ns,b,ffn", [(2, 256, 2048, 8, 256), (3, 64, 300, 3, 256), (2, 50, 77, 3, 128),
                                            (1, 32, 32, 1, 64)])
def test_c_driver_equals_python_driver(dev, monkeypatch, nl, nq, ns, b, ffn):
    monkeypatch.setenv("CODA_LAYERS", "fused")
    monkeypatch.setenv("CODA_DECODER_NODE", "stack")
    dec = _decoder(dev, nl, ffn, 0.0)
    tgt, memory, pos, qpos = _inputs(dev, nq, ns, b)
    w = torch.randn((nl, nq, b, 256), generator=torch.Generator().manual_seed(2)).to(dev)
    ran = {}
    real_c, real_py = FB._DecoderStackC.forward, FB._DecoderStack.forward
    monkeypatch.setattr(FB._DecoderStackC, "forward", staticmethod(lambda *a: (ran.setdefault("c", True), real_c(*a))[1]))
    monkeypatch.setattr(FB._DecoderStack, "forward", staticmethod(lambda *a: (ran.setdefault("py", True), real_py(*a))[1]))
    out_c, gm_c, gq_c, gp_c = _run(dec, True, monkeypatch, tgt, memory, pos, qpos, w)
    out_p, gm_p, gq_p, gp_p = _run(dec, False, monkeypatch, tgt, memory, pos, qpos, w)
    assert ran == {"c": True, "py": True}   # both drivers really ran
    _close(out_c, out_p, "outputs")
    _close(gm_c, gm_p, "memory gradient")
    _close(gq_c, gq_p, "query_pos gradient")
    assert set(gp_c) == set(gp_p)
    for k in gp_p:
        _close(gp_c[k], gp_p[k], f"grad {k}")


def test_c_driver_with_dropout_is_deterministic_and_consistent(dev, monkeypatch):
    """Same seed -> same masks in forward and backward: the analytic directional derivative matches central
    differences of the (deterministic, mask-fixed) forward."""
    from coda_neurips2023_amd import attention_core as core
    monkeypatch.setenv("CODA_LAYERS", "fused")
    monkeypatch.setenv("CODA_DECODER_NODE", "stack")
    monkeypatch.setattr(FB, "STACK_IN_C", True)
    nl, nq, ns, b = 2, 64, 256, 2
    dec = _decoder(dev, nl, 256, 0.1)
    tgt, memory, pos, qpos = _inputs(dev, nq, ns, b, seed=11)
    w = torch.randn((nl, nq, b, 256), generator=torch.Generator().manual_seed(3)).to(dev)
    fixed = iter([])
    monkeypatch.setattr(core, "_next_seed", lambda: (123456789, None))   # every call of the test draws the same seed

    def f(mem):
        return (dec(tgt, mem, pos=pos, query_pos=qpos)[0] * w).sum()

    m = memory.clone().requires_grad_(True)
    loss = f(m)
    loss.backward()
    again = f(memory)
    assert float(again) == float(loss)                      # deterministic given the seed
    assert float((dec(tgt, memory, pos=pos, query_pos=qpos)[0] == 0).float().mean()) < 0.5
    v = torch.randn(memory.shape, generator=torch.Generator().manual_seed(5)).to(dev)
    v = v / v.norm()
    eps = 2e-2
    with torch.no_grad():
        fd = (f(memory + eps * v).double() - f(memory - eps * v).double()) / (2 * eps)
    an = (m.grad.double() * v.double()).sum()
    assert abs(float(fd - an)) < 2e-2 * abs(float(an)) + 1e-3, (float(fd), float(an))
    # masks change with the seed
    monkeypatch.setattr(core, "_next_seed", lambda: (987654321, None))
    assert float(f(memory)) != float(loss)
