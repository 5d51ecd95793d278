"""72 weight gradients of the decoder stack (2048 rows, 256 x 256): one grouped launch vs one library GEMM each."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from coda_neurips2023_amd import gemm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
probs = [(torch.randn(2048, 256, generator=g).to(dev), torch.randn(2048, 256, generator=g).to(dev),
          torch.empty(256, 256, device=dev)) for _ in range(72)]


def grouped():
    d = gemm.DeferredWeightGrads()
    for dy, x, out in probs:
        d.add(out, dy, x)
    d.flush()


def single():
    for dy, x, out in probs:
        gemm.mm_tn(dy, x, out=out)


for name, fn in (("grouped", grouped), ("library x72", single)):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(20):
        fn()
    ev1.record()
    host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    dt = ev0.elapsed_time(ev1) / 20
    print(f"{name:12s}: {dt * 1e3:8.1f} us GPU  ({72 * 2 * 2048 * 256 * 256 / dt / 1e9:6.1f} TFLOP/s)   host {host * 1e6:7.1f} us")
