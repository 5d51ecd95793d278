#!/usr/bin/env python3
"""Dev probe: does the packed (tokens, 8E) K/V layout of the decoder stack, read cold, explain the in-step slowdown of
the cross-attention kernels (135 vs 93 us dK/dV)?  256 x 2048, 8 scenes x 4 heads x 64: contiguous vs layer-slice of
the packed buffer, warm (back-to-back reps) vs cold (1 GB written between reps)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coda_neurips2023_amd import attention_core as core  # noqa: E402

dev = torch.device("cuda:0")
L, S, B, H, D, NL = 256, 2048, 8, 4, 64, 8
q = torch.randn(L, B, H, D, device=dev, requires_grad=True)
go = torch.randn(L, B, H, D, device=dev)
k_all = torch.randn(S, B, NL * H * D, device=dev)
v_all = torch.randn(S, B, NL * H * D, device=dev)
PAD = int(os.environ.get("PAD", "64"))   # floats of row padding for the third layout
k_pad = torch.randn(S, B, NL * H * D + PAD, device=dev)[..., :NL * H * D]
v_pad = torch.randn(S, B, NL * H * D + PAD, device=dev)[..., :NL * H * D]
flush = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)


def run(strided, cold, reps=16):
    core.enable_kernel_timing(0)
    for r in range(reps + 2):
        layer = r % NL
        if strided == 2:
            k = k_pad.unflatten(2, (NL, H, D))[:, :, layer].detach().requires_grad_(True)
            v = v_pad.unflatten(2, (NL, H, D))[:, :, layer].detach().requires_grad_(True)
        elif strided:
            k = k_all.view(S, B, NL, H, D)[:, :, layer].detach().requires_grad_(True)
            v = v_all.view(S, B, NL, H, D)[:, :, layer].detach().requires_grad_(True)
        else:
            k = k_all.view(S, B, NL, H, D)[:, :, layer].contiguous().requires_grad_(True)
            v = v_all.view(S, B, NL, H, D)[:, :, layer].contiguous().requires_grad_(True)
        if cold:
            flush.fill_(float(r))
        out, _ = core.attention(q, k, v, None, 0.125, 0.1, False)
        if cold:
            flush.fill_(float(r) + 0.5)
        out.backward(go)
    rec = core.collect_kernel_timing()
    core.disable_kernel_timing()
    line = f"{['contiguous  ', 'packed slice', 'padded slice'][int(strided)]} {'cold' if cold else 'warm'}:"
    for kind in ("fwd", "dkv", "dq"):
        ms = sorted(rec[(kind, L, S)][2:])
        line += f"  {kind} {1e3 * ms[len(ms) // 2]:6.1f} us"
    print(line)


for strided in (0, 1, 2):
    for cold in (False, True):
        run(strided, cold)
