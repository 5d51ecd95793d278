"""N>1 path on CPU: world_size-2 `gloo` processes.  The data-parallel exchange of the
hot path is the DDP gradient all-reduce (+ the num_boxes normaliser of the criterion);
scenes are sharded, nothing else is exchanged.  The per-rank compute uses the CPU
oracle patched into the kernel seams (oracle/cpu_port.py) because the HIP operators
need a GPU; what is under test here is the distributed host logic."""
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
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    from golden.weights import fill_deterministic
    from test_model_structure import tiny_args

    from coda_neurips2023_amd.dataset_config import HotPathDatasetConfig
    from coda_neurips2023_amd.model_3detr import build_model
    model, _ = build_model(tiny_args(), HotPathDatasetConfig())
    return fill_deterministic(model, seed=9).train()


def _batch(rank):
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    pc, mn, mx = make_batch(2, 1024, seed=100 + rank)
    return {"point_clouds": torch.from_numpy(pc), "point_cloud_dims_min": torch.from_numpy(mn),
            "point_cloud_dims_max": torch.from_numpy(mx)}


def _loss(model, batch):
    o = model(batch)["outputs"]
    return o["center_normalized"].square().mean() + o["text_correlation_embedding"].abs().mean() + \
        o["sem_cls_logits"].square().mean() + o["size_normalized"].mean() + o["angle_logits"].square().mean() + \
        o["angle_residual"].square().mean()


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import cpu_port

    from coda_neurips2023_amd.criterion import all_reduce_average
    try:
        with cpu_port.patched():
            # local gradient without DDP
            local = _build()
            _loss(local, _batch(rank)).backward()
            g_local = {k: p.grad.clone() for k, p in local.named_parameters() if p.grad is not None}
            # the same step under DDP
            ddp = torch.nn.parallel.DistributedDataParallel(_build())
            _loss(ddp, _batch(rank)).backward()
            g_ddp = {k: p.grad.clone() for k, p in ddp.module.named_parameters() if p.grad is not None}
        # DDP gradient == mean over ranks of the local gradients
        for k, g in g_local.items():
            t = g.clone()
            dist.all_reduce(t)
            t /= world
            assert torch.allclose(g_ddp[k], t, rtol=1e-4, atol=1e-6), k
        assert set(g_ddp) == set(g_local)
        # the criterion's num_boxes normaliser (criterion.py:1181)
        nb = all_reduce_average(torch.tensor(float(3 + 4 * rank)))
        assert abs(float(nb) - 5.0) < 1e-6
        np.save(os.path.join(tmpdir, f"ok{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_ddp_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0.npy") and os.path.exists(tmp_path / "ok1.npy")
