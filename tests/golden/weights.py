"""Deterministic, framework-independent parameter filling shared by the golden
generator (applied to the REFERENCE modules) and the tests (applied to this
repo's modules): identical key names + shapes => identical weights, so the
fixtures only need to carry inputs and outputs."""
import zlib

import numpy as np
import torch


def fill_deterministic(module, seed=0):
    sd = module.state_dict()
    new = {}
    for name in sorted(sd):
        t = sd[name]
        if not t.dtype.is_floating_point:
            new[name] = t.clone()
            continue
        rng = np.random.default_rng(zlib.crc32(name.encode()) + seed)
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            v = rng.normal(0, 0.1, shape)
        elif t.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0, 1.0, shape) * (1.0 / np.sqrt(fan_in))
            if leaf == "gauss_B":
                v = rng.normal(0, 1.0, shape)
        elif leaf == "weight":  # norm scales
            v = rng.uniform(0.5, 1.5, shape)
        else:  # biases, scalars
            v = rng.normal(0, 0.05, shape)
        new[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape)
    module.load_state_dict(new, strict=True)
    return module


def grad_digest(module):
    """Per-parameter gradient digest: [sum, l2 norm, 16 strided samples]."""
    out = {}
    for name, p in module.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().cpu().double().reshape(-1)
        idx = np.linspace(0, g.numel() - 1, 16).astype(np.int64)
        out[name] = np.concatenate([[g.sum().item(), g.norm().item()], g[idx].numpy()]).astype(np.float64)
    return out
