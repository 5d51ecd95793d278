#!/usr/bin/env python3
"""Does any kernel of the training step read memory it (or an earlier launch) has not written?  Every floating-point
tensor that torch.empty / empty_like / new_empty / empty_strided hands out is filled with NaN first (the caching
allocator otherwise returns whatever an earlier tensor left there -- usually benign finite values), then the whole-step
parity test of tests/test_reference_step_gpu.py runs on each case.  Dev tool (GPU).

    [POISON_ALL=1] python tools/poison_empty.py [case ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_empty, _empty_like, _empty_strided, _new_empty = torch.empty, torch.empty_like, torch.empty_strided, torch.Tensor.new_empty
COUNT = [0]


ALL = os.environ.get("POISON_ALL", "0") == "1"   # also byte workspaces and integer tensors (0xFF bytes: NaN / -1)


def _poison(t):
    if not (t.is_cuda and t.numel()):
        return t
    if t.is_floating_point():
        COUNT[0] += 1
        t.fill_(float("nan"))
    elif ALL and t.dtype in (torch.uint8, torch.int32, torch.int64):
        COUNT[0] += 1
        t.view(torch.uint8).fill_(255)
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
torch.empty_strided = lambda *a, **k: _poison(_empty_strided(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _poison(_new_empty(self, *a, **k))

import test_reference_step_gpu as T  # noqa: E402
from golden import step_inputs as SI  # noqa: E402

dev = torch.device("cuda:0")
for case in (sys.argv[1:] or list(SI.CASES)):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"step_full_{case}.npz"))
    COUNT[0] = 0
    res = T.run_product(dev, case, z)
    torch.cuda.synchronize()
    print(f"== {case}: {COUNT[0]} poisoned allocations; loss {float(res[0]):.6f}")
    model = res[4]
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print("   non-finite gradients:", bad[:12] if bad else "none")
    try:
        T.compare(*res, z, few_tokens=SI.CASES[case]["nq"] * SI.B <= 1024)
        print("   parity: ok")
    except AssertionError as e:
        print("   parity FAILED:", str(e)[:300])
