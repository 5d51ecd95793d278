"""Weight gradients with a LONG reduction (the encoder's 16 384 tokens, the SA MLP's ~0.7 M grouped rows):
dW = dy^T x.  Candidates: (a) split-K bmm + sum (linear_fn.tn_gemm, the current route), (b) the library's TN GEMM
with first-use candidate timing, (c) the split-rows atomics kernel coda_gemm_tn_f32, (d) the grouped kernel over
row chunks + sum."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coda_neurips2023_amd import _lib, gemm, linear_fn  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def grouped_chunks(dy, x, chunks):
    rows = dy.shape[0] // chunks
    part = torch.empty((chunks, dy.shape[1], x.shape[1]), device=dev)
    d = gemm.DeferredWeightGrads()
    for i in range(chunks):
        d.add(part[i], dy[i * rows:(i + 1) * rows], x[i * rows:(i + 1) * rows])
    d.flush()
    return part.sum(0)


for rows, m, n in [(16384, 768, 256), (16384, 256, 256), (16384, 128, 256), (16384, 256, 128),
                   (720896, 128, 64), (720896, 256, 128)]:
    g = torch.Generator().manual_seed(0)
    dy, x = torch.randn(rows, m, generator=g).to(dev), torch.randn(rows, n, generator=g).to(dev)
    ref = (dy.double().t() @ x.double())
    out = torch.empty(m, n, device=dev)
    res = {}
    lib.coda_gemm_set_tuning(0)
    res["bmm+sum"] = (timeit(lambda: linear_fn.tn_gemm(dy, x)), linear_fn.tn_gemm(dy, x))
    res["library"] = (timeit(lambda: gemm.mm_tn(dy, x, out=out, kernel=False)), gemm.mm_tn(dy, x, kernel=False))
    lib.coda_gemm_set_tuning(1)
    # a new leading dimension makes a new plan, so the tuned plan is timed separately from the untuned one
    out2 = torch.empty(m, n + 4, device=dev)[:, :n]
    res["library tuned"] = (timeit(lambda: gemm.mm_tn(dy, x, out=out2, kernel=False)), gemm.mm_tn(dy, x, out=out2, kernel=False).clone())
    lib.coda_gemm_set_tuning(-1)
    res["atomics kernel"] = (timeit(lambda: gemm.mm_tn(dy, x, out=out, kernel=True)), gemm.mm_tn(dy, x, kernel=True))
    for ch in (8, 32):
        if rows % (ch * 8) == 0 and m % 64 == 0 and n % 64 == 0:
            res[f"grouped x{ch} + sum"] = (timeit(lambda: grouped_chunks(dy, x, ch)), grouped_chunks(dy, x, ch))
    fl = 2.0 * rows * m * n
    print(f"rows {rows} m {m} n {n}:")
    for k, (us, val) in res.items():
        err = float((val.double() - ref).abs().max() / ref.abs().max())
        print(f"   {k:20s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s   rel err {err:.1e}")
