"""SyncBatchNorm semantics of the fused batch-norm paths, on ONE GPU: two `gloo` processes share
cuda:0 (gloo all-reduces CUDA tensors through the host; RCCL refuses two ranks on one device),
each holds half of the batch, and outputs / summed parameter gradients / running statistics must
equal a single process running plain BatchNorm on the whole batch -- which is what
nn.SyncBatchNorm.convert_sync_batchnorm + DDP mean in the reference (main.py:993-996).  Covers
the set-abstraction shared MLP (pointnet2/fused_sa_mlp.py) and the GenericMLP stacks
(fused_bn_mlp.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WIDTHS = {"streaming": [0, 16, 32, 64],      # pointnet2/fused_sa_mlp.py's streaming kernels around library GEMMs
          "mfma": [0, 64, 128, 256]}          # the pre-encoder's own widths: the hand-written pipeline of csrc/sa_mfma.hip


def _modules(dev, path="streaming"):
    from functools import partial

    from coda_neurips2023_amd.helpers import GenericMLP
    from coda_neurips2023_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(11)
    sa = PointnetSAModuleVotes(radius=0.4, nsample=16, npoint=64, mlp=list(WIDTHS[path]), normalize_xyz=True)
    if path == "mfma":
        from coda_neurips2023_amd.pointnet2 import fused_sa_mlp
        assert fused_sa_mlp.mfma_eligible(sa.mlp_module, 16), "this case is meant to run the MFMA pipeline"
    mk = partial(GenericMLP, norm_fn_name="bn1d", activation="relu", use_conv=True, hidden_dims=[64, 64], dropout=0.0,
                 input_dim=WIDTHS[path][-1])
    heads = torch.nn.ModuleList([mk(output_dim=5), mk(output_dim=32)])
    for m in list(sa.modules()) + list(heads.modules()):
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.uniform_(m.bias, -0.3, 0.3)
    return sa.to(dev).train(), heads.to(dev).train()


def _data(dev, path="streaming"):
    g = torch.Generator().manual_seed(5)
    xyz = torch.rand(4, 700, 3, generator=g) * 2
    w_feat = torch.randn(4, WIDTHS[path][-1], 64, generator=g)
    w_head = [torch.randn(4 * 64, 5, generator=g), torch.randn(4 * 64, 32, generator=g)]
    return xyz.to(dev), w_feat.to(dev), [w.to(dev) for w in w_head]


def _run(sa, heads, xyz, w_feat, w_head):
    """SA module -> features (B,64,npoint) -> tokens (B*npoint,64) -> the two heads; loss = weighted sums."""
    from coda_neurips2023_amd import fused_bn_mlp
    _, feat, _ = sa(xyz)
    tokens = feat.permute(0, 2, 1).reshape(-1, feat.shape[1])
    parsed = fused_bn_mlp.eligible(list(heads), tokens)
    assert parsed is not None
    outs = fused_bn_mlp.run_stacks(tokens, parsed)
    loss = (feat * w_feat).sum()
    for o, w in zip(outs, w_head):
        loss = loss + (o * w).sum()
    loss.backward()
    return feat.detach(), [o.detach() for o in outs]


def _state(sa, heads):
    grads = {f"sa.{k}": p.grad.detach().cpu() for k, p in sa.named_parameters()}
    grads.update({f"heads.{k}": p.grad.detach().cpu() for k, p in heads.named_parameters()})
    bufs = {f"sa.{k}": b.detach().cpu() for k, b in sa.named_buffers()}
    bufs.update({f"heads.{k}": b.detach().cpu() for k, b in heads.named_buffers()})
    return grads, bufs


def _worker(rank, world, port, tmpdir, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sa, heads = _modules(dev, path)
    sa = torch.nn.SyncBatchNorm.convert_sync_batchnorm(sa)
    heads = torch.nn.SyncBatchNorm.convert_sync_batchnorm(heads)
    xyz, w_feat, w_head = _data(dev, path)
    sl = slice(2 * rank, 2 * rank + 2)
    feat, outs = _run(sa, heads, xyz[sl].contiguous(), w_feat[sl], [w.view(4, 64, -1)[sl].reshape(128, -1) for w in w_head])
    grads, bufs = _state(sa, heads)
    torch.save({"feat": feat.cpu(), "outs": [o.cpu() for o in outs], "grads": grads, "bufs": bufs},
               os.path.join(tmpdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("path", list(WIDTHS))
def test_two_ranks_equal_one_process_on_the_whole_batch(dev, tmp_path, path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), path), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))

    sa, heads = _modules(dev, path)
    xyz, w_feat, w_head = _data(dev, path)
    feat, outs = _run(sa, heads, xyz, w_feat, w_head)
    grads, bufs = _state(sa, heads)

    def close(got, ref, what):
        got, ref = got.double().numpy(), ref.double().cpu().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < RTOL, f"{what}: {err:.3e}"

    close(torch.cat([r0["feat"], r1["feat"]]), feat, "SA features")
    for g in range(2):
        both = torch.cat([r0["outs"][g].view(2, 64, -1), r1["outs"][g].view(2, 64, -1)]).reshape(256, -1)
        close(both, outs[g], f"head {g} output")
    for k in grads:
        close(r0["grads"][k] + r1["grads"][k], grads[k], f"grad {k}")
    for k in bufs:
        if bufs[k].dtype.is_floating_point:
            close(r0["bufs"][k], bufs[k], f"buffer {k} (rank 0)")
            close(r1["bufs"][k], bufs[k], f"buffer {k} (rank 1)")
        else:
            assert int(r0["bufs"][k]) == int(bufs[k])
