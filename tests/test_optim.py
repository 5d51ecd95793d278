"""clip_grad_norm_ + AdamW in three launches (include/coda_optim.h) against torch.nn.utils.clip_grad_norm_ and
torch.optim.AdamW -- the calls of engine.py:161-164 / optimizer.py:35."""
import copy

import pytest
import torch

from coda_neurips2023_amd import optim


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 256), (768,), (3,), (5000, 7), (1,), (2049,), (64, 3, 1, 1), (10,), (524288,)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


def test_chunk_map_and_flat_offsets_cover_every_element_once():
    import numpy as np
    sizes = [256 * 256, 768, 3, 0, 2048, 2049, 1, 524288]
    cmap = optim.chunk_map(sizes, 2048)
    assert cmap.dtype == np.int32 and cmap.shape == (sum(-(-n // 2048) for n in sizes), 2)
    seen = {i: np.zeros(n, dtype=np.int32) for i, n in enumerate(sizes)}
    for t, c in cmap:
        seen[int(t)][c * 2048:min((c + 1) * 2048, sizes[t])] += 1
    assert all((v == 1).all() for v in seen.values())
    assert list(cmap[:, 0]) == sorted(cmap[:, 0])                       # tensor order
    offsets, total = optim.flat_offsets(sizes)
    assert all(o % 4 == 0 for o in offsets) and total % 4 == 0
    ends = [o + n for o, n in zip(offsets, sizes)]
    assert all(e <= o2 for e, o2 in zip(ends, offsets[1:])) and ends[-1] <= total
    assert optim.flat_offsets([]) == ([], 4)


def test_cpu_parameters_are_rejected():
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        optim.clip_grad_norm_([p], 0.1)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        optim.AdamW([p]).step()


@pytest.mark.gpu
@pytest.mark.parametrize("max_norm", [0.1, 1e6])
def test_clip_and_adamw_match_torch(dev, max_norm):
    mine = _params(dev)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    groups = lambda ps: [{"params": ps[:4], "weight_decay": 0.0}, {"params": ps[4:], "weight_decay": 0.1}]  # noqa: E731
    o_mine = optim.AdamW(groups(mine), lr=3e-3)
    o_ref = torch.optim.AdamW(groups(ref), lr=3e-3, fused=False, foreach=False)
    gen = torch.Generator().manual_seed(9)
    for step in range(6):
        # gradients as views into one flat buffer at odd offsets, like DDP's bucket views
        flat = torch.randn(sum(p.numel() for p in mine) + 1, generator=gen).to(dev) * (10.0 if step % 2 else 0.01)
        at = 1
        for a, b in zip(mine, ref):
            a.grad = flat[at:at + a.numel()].view_as(a)
            b.grad = a.grad.clone()
            at += a.numel()
        if step == 3:      # a parameter without gradient is left alone and keeps its own step count (as in torch)
            mine[7].grad = None
            ref[7].grad = None
        n_mine = optim.clip_grad_norm_(mine, max_norm)
        n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        assert torch.allclose(n_mine, n_ref, rtol=1e-5)
        for a, b in zip(mine, ref):
            if a.grad is not None:
                assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-12)
        o_mine.step()
        o_ref.step()
        for i, (a, b) in enumerate(zip(mine, ref)):
            assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), (step, i, float((a - b).abs().max()))
    # checkpoints interchange: torch loads ours, we load torch's
    sd = o_mine.state_dict()
    o_ref2 = torch.optim.AdamW(groups([torch.nn.Parameter(p.detach().clone()) for p in mine]), lr=3e-3)
    o_ref2.load_state_dict(copy.deepcopy(sd))
    o_mine2 = optim.AdamW(groups(mine), lr=3e-3)
    o_mine2.load_state_dict(o_ref.state_dict())
    for a in mine:
        a.grad = torch.ones_like(a)
    o_mine2.step()
    assert float(o_mine2.state[mine[0]]["step"]) == 7.0 and float(o_mine2.state[mine[7]]["step"]) == 6.0


def _reducer_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)  # gloo all-reduces CUDA tensors through the host
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    params = _params(dev, seed=rank)                     # different values per rank: rank 0's must win
    red = optim.FlatGradReducer(params)
    g = torch.Generator().manual_seed(100 + rank)
    for p in params:
        p.grad = torch.randn(p.shape, generator=g).to(dev)
    params[2].grad = None                                # no gradient on this rank / step
    red.reduce()
    torch.save({"params": [p.detach().cpu() for p in params], "grads": [p.grad.cpu() for p in params]},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_flat_grad_reducer_two_ranks(dev, tmp_path):
    """Two gloo ranks sharing cuda:0 (RCCL refuses two ranks per device): parameters broadcast from rank 0, gradients
    averaged over the ranks, every p.grad a 16-byte aligned view of one flat buffer."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_reducer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in (0, 1))
    ref_params = _params(torch.device("cpu"), seed=0)
    for a, b, c in zip(r0["params"], r1["params"], ref_params):
        assert torch.equal(a, b) and torch.equal(a, c.detach())
    for i, (a, b) in enumerate(zip(r0["grads"], r1["grads"])):
        assert torch.equal(a, b)
    gens = [torch.Generator().manual_seed(100 + r) for r in (0, 1)]
    for i, a in enumerate(r0["grads"]):
        per_rank = [torch.randn(a.shape, generator=g) for g in gens]
        want = torch.zeros_like(a) if i == 2 else (per_rank[0] + per_rank[1]) / 2
        assert torch.allclose(a, want, rtol=1e-6, atol=1e-7), i


@pytest.mark.gpu
def test_flat_grad_reducer_single_process(dev):
    params = _params(dev)
    red = optim.FlatGradReducer(params)
    grads = [torch.randn_like(p) for p in params]
    for p, g in zip(params, grads):
        p.grad = g.clone()
    red.reduce()
    for p, g in zip(params, grads):
        assert torch.equal(p.grad, g) and p.grad.data_ptr() % 16 == 0
        assert red.flat.data_ptr() <= p.grad.data_ptr() < red.flat.data_ptr() + red.flat.numel() * 4
    n = optim.clip_grad_norm_(params, 0.1)               # the clipping sees (and scales) the flat views
    want = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
    assert abs(float(n) - float(want)) < 1e-3 * float(want)


def _logic_only_reducer(sizes):
    """FlatGradReducer's hook bookkeeping without a device: stub segments, ``_fire`` records the issue order."""
    import types
    red = optim.FlatGradReducer.__new__(optim.FlatGradReducer)
    red._sync = True
    red.segments = [types.SimpleNamespace(params=[object()] * n, pending=n, fired=False, work=None) for n in sizes]
    red.issued = []

    def fire(seg):
        red.issued.append(red.segments.index(seg))
        seg.fired = True

    red._fire = fire
    return red


def test_segments_are_issued_in_index_order_whatever_order_the_gradients_land_in():
    """Round 3's advisor finding: every rank must issue all_reduce(segment 0) before all_reduce(segment 1), also
    when segment 1's gradients are complete first, or when segment 0 is left to reduce() on one rank only."""
    # rank A: segment 0 completes first
    a = _logic_only_reducer([2, 3])
    hooks = [a._hook_for(s) for s in a.segments]
    for k in (0, 0, 1, 1, 1):
        hooks[k](None)
    # rank B: segment 1 completes first -> deferred until segment 0 has fired
    b = _logic_only_reducer([2, 3])
    hooks = [b._hook_for(s) for s in b.segments]
    for k in (1, 1, 1, 0):
        hooks[k](None)
    assert b.issued == []
    hooks[0](None)
    # rank C: one parameter of segment 0 gets no gradient -> nothing fires from the hooks, reduce() issues 0, 1
    c = _logic_only_reducer([2, 3])
    hooks = [c._hook_for(s) for s in c.segments]
    for k in (1, 1, 1, 0):
        hooks[k](None)
    assert c.issued == []
    optim.FlatGradReducer.reduce(c)
    assert a.issued == b.issued == c.issued == [0, 1]
    assert all(s.pending == len(s.params) and not s.fired for s in c.segments)  # re-armed by reduce()


def test_rearm_resets_a_step_that_never_reached_reduce():
    """A step abandoned after backward (engine.py:155-157's exit, a caller's ``continue``) must not leave a segment
    marked as fired or counters half decremented: the module's forward pre-hook re-arms."""
    r = _logic_only_reducer([2, 2])
    hooks = [r._hook_for(s) for s in r.segments]
    for k in (0, 0, 1):          # segment 0 fired, segment 1 half way, reduce() never called
        hooks[k](None)
    assert r.issued == [0] and r.segments[1].pending == 1
    with torch.enable_grad():
        r.rearm()
    assert [(s.pending, s.fired) for s in r.segments] == [(2, False), (2, False)]
    for k in (0, 0, 1, 1):
        hooks[k](None)
    assert r.issued == [0, 0, 1]
    with r.no_sync():            # accumulation passes neither count nor re-arm
        r.rearm()
        hooks[0](None)
    assert r.segments[0].fired and r.segments[1].fired
