"""Host cost of the decoder stack (forward + backward, 8 layers): launch-bound shapes, so wall time = host time.
C driver (include/coda_stack.h) vs the per-launch Python driver."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CODA_LAYERS"] = "fused"
os.environ["CODA_DECODER_NODE"] = "stack"
from coda_neurips2023_amd import fused_blocks as FB  # noqa: E402
from coda_neurips2023_amd.transformer import TransformerDecoder, TransformerDecoderLayer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = TransformerDecoder(TransformerDecoderLayer(d_model=256, nhead=4, dim_feedforward=256, dropout=0.1), 8,
                         return_intermediate=True).to(dev).train()
nq, ns, b = 64, 64, 1
tgt = torch.zeros(nq, b, 256, device=dev)
mem = torch.randn(ns, b, 256, device=dev, requires_grad=True)
pos = torch.randn(ns, b, 256, device=dev)
qp = torch.randn(nq, b, 256, device=dev, requires_grad=True)
for name, in_c in (("C driver", True), ("Python driver", False), ("C driver", True)):
    FB.STACK_IN_C = in_c
    for _ in range(5):
        dec(tgt, mem, pos=pos, query_pos=qp)[0].sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    tf = 0.0
    for _ in range(n):
        a = time.perf_counter()
        out = dec(tgt, mem, pos=pos, query_pos=qp)[0]
        tf += time.perf_counter() - a
        out.sum().backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:14s}: {dt * 1e3:6.2f} ms per forward+backward (forward {tf / n * 1e3:5.2f} ms) at launch-bound shapes")
