"""The multi-GPU exchange of the step: SyncBatchNorm statistics inside the fused batch-norm paths +
``optim.FlatGradReducer`` (segmented flat gradient all-reduce, early segment issued from a backward hook), as
main.py:993-996 wraps the model (SyncBatchNorm + DDP).

* two ``gloo`` ranks sharing cuda:0 (runs on the 1-GPU box; RCCL refuses two ranks on one device);
* two ``nccl`` (= RCCL) ranks on cuda:0 / cuda:1 -- skipped below two GPUs, so the first box that has them runs it;
* single process: the hooks fire the early segment DURING backward.

Each rank holds half of the batch; after ``backward`` + ``reduce`` every rank must hold the whole batch's gradient
/ 2 (the mean over the two ranks of per-rank sums), identical on both ranks, as views of one flat buffer, and the
batch-norm running statistics of the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from coda_neurips2023_amd import optim  # noqa: E402

RTOL = 1e-3


def test_reducer_partition_by_name_and_by_size():
    named = [(f"{m}.w{i}", torch.nn.Parameter(torch.zeros(1000))) for m in ("pre", "enc", "dec", "heads") for i in range(3)]
    first, rest = optim.FlatGradReducer.partition(named, early=("heads.", "dec."))
    assert [n for n, _ in first] == [n for n, _ in named if n.startswith(("dec.", "heads."))]
    assert [n for n, _ in rest] == [n for n, _ in named if n.startswith(("pre.", "enc."))]
    # no names: the tail of the registration order (gradients arrive roughly in reverse order) up to the byte bound
    first, rest = optim.FlatGradReducer.partition(named, None, segment_bytes=5 * 4000)
    assert [n for n, _ in first] == [n for n, _ in reversed(named)][:5]
    assert [n for n, _ in rest] == [n for n, _ in reversed(named)][5:]
    assert {id(p) for _, p in first} | {id(p) for _, p in rest} == {id(p) for _, p in named}


class _Net(torch.nn.Module):
    """Set-abstraction module -> two GenericMLP heads through the fused batch-norm paths (tests/test_syncbn_gpu.py)."""

    def __init__(self, sa, heads):
        super().__init__()
        self.sa, self.heads = sa, heads

    def forward(self, xyz):
        from coda_neurips2023_amd import fused_bn_mlp
        _, feat, _ = self.sa(xyz)
        tokens = feat.permute(0, 2, 1).reshape(-1, 64)
        parsed = fused_bn_mlp.eligible(list(self.heads), tokens)
        assert parsed is not None
        return feat, fused_bn_mlp.run_stacks(tokens, parsed)


def _loss(net, xyz, w_feat, w_head):
    feat, outs = net(xyz)
    loss = (feat * w_feat).sum()
    for o, w in zip(outs, w_head):
        loss = loss + (o * w).sum()
    return loss


def _worker(rank, world, port, tmpdir, backend):
    import test_syncbn_gpu as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                        # different initial values per rank: rank 0's must win
    sa, heads = S._modules(dev)
    if rank != 0:
        with torch.no_grad():
            for t in list(sa.parameters()) + list(heads.parameters()) + list(sa.buffers()) + list(heads.buffers()):
                if t.dtype.is_floating_point:
                    t.add_(0.25)
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(_Net(sa, heads))
    reducer = optim.FlatGradReducer(net, early=("heads.",))      # broadcasts rank 0's parameters and buffers
    xyz, w_feat, w_head = S._data(dev)
    sl = slice(2 * rank, 2 * rank + 2)
    fired = []
    fire = reducer._fire
    reducer._fire = lambda seg: (fired.append(reducer.segments.index(seg)), fire(seg))[1]
    for _ in range(2):                                   # two steps: the hooks re-arm
        net.zero_grad(set_to_none=True)
        _loss(net, xyz[sl].contiguous(), w_feat[sl], [w.view(4, 64, -1)[sl].reshape(128, -1) for w in w_head]).backward()
        reducer.reduce()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    in_flat = all(reducer.flat.data_ptr() <= p.grad.data_ptr() < reducer.flat.data_ptr() + 4 * reducer.flat.numel()
                  for p in net.parameters())
    torch.save({"grads": grads, "bufs": {k: b.detach().cpu() for k, b in net.named_buffers()}, "fired": fired,
                "in_flat": in_flat, "params": {k: p.detach().cpu() for k, p in net.named_parameters()}},
               os.path.join(tmpdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_reduce_to_the_whole_batch_gradient(dev, tmp_path, backend):
    import test_syncbn_gpu as S
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: fewer than two GPUs on this box")
    port = S._free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), backend), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))

    torch.manual_seed(100)
    sa, heads = S._modules(dev)
    net = _Net(sa, heads)
    xyz, w_feat, w_head = S._data(dev)
    for _ in range(2):                                   # running statistics after two steps
        net.zero_grad(set_to_none=True)
        _loss(net, xyz, w_feat, w_head).backward()
    ref_grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    ref_bufs = {k: b.detach().cpu() for k, b in net.named_buffers()}

    def close(got, ref, what):
        got, ref = got.double().numpy(), ref.double().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < RTOL, f"{what}: {err:.3e}"

    assert r0["in_flat"] and r1["in_flat"], "every p.grad must be a view of the flat buffer"
    for k in ref_grads:
        assert torch.equal(r0["params"][k], r1["params"][k]), f"parameter {k} differs between the ranks"
        assert torch.equal(r0["grads"][k], r1["grads"][k]), f"reduced gradient {k} differs between the ranks"
        close(r0["grads"][k] * 2, ref_grads[k], f"grad {k}")        # mean over 2 ranks of per-rank sums
    for k in ref_bufs:
        if ref_bufs[k].dtype.is_floating_point:
            close(r0["bufs"][k], ref_bufs[k], f"buffer {k} (rank 0)")
            close(r1["bufs"][k], ref_bufs[k], f"buffer {k} (rank 1)")
    # the early segment (index 0: the heads) was fired by its hook, before reduce() fired the rest -- both steps
    assert r0["fired"] == [0, 1, 0, 1] and r1["fired"] == [0, 1, 0, 1]


@pytest.mark.gpu
def test_hooks_fire_the_early_segment_during_backward(dev):
    """Single process: the early segment is packed while later parameters still have no gradient."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 8)).to(dev)
    ref = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
    reducer = optim.FlatGradReducer(net, early=("4.",), broadcast=False)
    seen = []
    fire = reducer._fire

    def spy(seg):
        seen.append((reducer.segments.index(seg), [p.grad is not None for p in net.parameters()]))
        return fire(seg)

    reducer._fire = spy
    x = torch.randn(16, 32, device=dev)
    net(x).square().sum().backward()
    # both segments fire from their hooks: the early one while the first layers still have no gradient, the late one
    # when its last gradient has landed (the end of backward); reduce() then only waits
    assert [s[0] for s in seen] == [0, 1], "both segments fire inside backward"
    assert seen[0][1][-2:] == [True, True] and not any(seen[0][1][:4]), "...the early one before the first layers' gradients exist"
    reducer.reduce()
    assert [s[0] for s in seen] == [0, 1]
    h = torch.relu(torch.nn.functional.linear(x, ref[0], ref[1]))
    h = torch.relu(torch.nn.functional.linear(h, ref[2], ref[3]))
    torch.nn.functional.linear(h, ref[4], ref[5]).square().sum().backward()
    for p, r in zip(net.parameters(), ref):
        assert torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-6)
        assert reducer.flat.data_ptr() <= p.grad.data_ptr() < reducer.flat.data_ptr() + 4 * reducer.flat.numel()
    # a step in which one parameter gets no gradient: nothing fires early for its segment, reduce() completes it,
    # and in a single process the missing gradient stays None (AdamW then skips the tensor, as torch does)
    net.zero_grad(set_to_none=True)
    with torch.no_grad():
        net[0].bias.requires_grad_(False)
    net(x).square().sum().backward()
    net[0].bias.requires_grad_(True)
    reducer.reduce()
    assert net[0].bias.grad is None and net[0].weight.grad is not None
    # accumulation over two backward calls: the first under no_sync()
    net.zero_grad(set_to_none=True)
    with reducer.no_sync():
        net(x).square().sum().backward()
    net(x).square().sum().backward()
    reducer.reduce()
    for p, r in zip(net.parameters(), ref):
        assert torch.allclose(p.grad, 2 * r.grad, rtol=1e-5, atol=1e-6)
