"""The headline step in a process that holds a one-rank RCCL process group (nothing wrapped, no collective issued):
how the mere existence of the communication library's streams changes the step.  Found the hardware-queue sharing
between the sampling side stream and the compute stream (22.3 vs 18.1 ms per step; CODA_PREFETCH_PRIORITY=0
reproduces it).      python tools/pg_probe.py plain|sync      (sync: with SyncBatchNorm conversion)"""
import os, sys, cProfile, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mod, step_fn, desc, kind = bench.build_workload("model", dev)
if sys.argv[1] == "sync":
    mod = torch.nn.SyncBatchNorm.convert_sync_batchnorm(mod)
calls = {}
for name, m in mod.named_modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.register_forward_pre_hook(lambda mod_, inp, name=name: calls.__setitem__(name, calls.get(name, 0) + 1))
pool = []
for i in range(3):
    pc, mn, mx = bench.make_batch(8, 20000, seed=1 + i)
    pool.append({"point_clouds": torch.from_numpy(pc).to(dev), "point_cloud_dims_min": torch.from_numpy(mn).to(dev), "point_cloud_dims_max": torch.from_numpy(mx).to(dev)})
opt, clip = bench.make_optimizer(mod.parameters())
def one(i):
    mod.prefetch_sampling(pool[(i + 1) % 3], wait_for=None)
    opt.zero_grad(set_to_none=True)
    step_fn(mod, pool[i % 3]).backward()
    clip(); opt.step()
for i in range(5): one(i)
torch.cuda.synchronize()
import time
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for i in range(10): one(i)
pr.disable(); torch.cuda.synchronize()
print(sys.argv[1], "ms/step", (time.perf_counter() - t) / 10 * 1e3)
print("BN module forwards:", calls)
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"])
dist.destroy_process_group()
