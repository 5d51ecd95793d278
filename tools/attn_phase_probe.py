#!/usr/bin/env python3
"""Where the short one-kernel attention backward (mha_bwd_fused_short_kernel) spends its time: shader-clock stamps of
wave 0 of every workgroup, from a library built with -DCODA_ATTN_PROF (see the macro in csrc/attention.hip):

    cd coda_neurips2023_amd/csrc && hipcc <Makefile flags> -DCODA_ATTN_PROF -c attention.hip -o ../../tools/_build/attention_prof.o
    hipcc --offload-arch=gfx950 -shared -fPIC <other _build/*.o> tools/_build/attention_prof.o -lhipblaslt -o tools/_build/libcoda_hip_prof.so
    python tools/attn_phase_probe.py [l] [s]
Dev tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coda_neurips2023_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libcoda_hip_prof.so")
from coda_neurips2023_amd import attention_core as core  # noqa: E402

NAMES = ["start", "K/V in", "staged (loads+LDS)", "barrier", "S,dP", "softmax", "dV,dK", "dS^T+dQ", "partials out",
         "closing writes", "end"]
l = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = int(sys.argv[2]) if len(sys.argv) > 2 else 256
b, h, d = 8, 4, 64
dev = torch.device("cuda:0")
lib = _lib.load()
q, k, v, go = (torch.randn(n, b, h, d, device=dev) for n in (l, s, s, l))
out = torch.empty_like(q)
lse = torch.empty(b, h, l, device=dev)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
delta = torch.zeros(max(b * h * l, b * h * (s // 32) * 16), device=dev)
ws_bytes = lib.coda_mha_bwd_ws_bytes(b, h, l, s, d)
ws = torch.empty(ws_bytes // 4, device=dev)
st = _lib.current_stream_handle()
_lib.check(lib.coda_mha_fwd_opt_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), b, h, l, s,
                                    d, h * d, h * d, h * d, 0.125, 0.1, 5, None, 0, st), "fwd")
for _ in range(5):
    _lib.check(lib.coda_mha_bwd_ws_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(),
                                       go.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), b, h, l, s,
                                       d, h * d, h * d, h * d, 0, 0, 0, 0.125, 0.1, 5, None, ws.data_ptr(), ws_bytes, 0, st),
               "bwd")
torch.cuda.synchronize()
t = delta[: b * h * (s // 32) * 16].view(-1, 16)[:, :11].cpu()
print(f"l={l} s={s}: {t.shape[0]} workgroups; shader clocks since the workgroup's first stamp (wave 0), median / max")
prev = torch.zeros(t.shape[0])
for i, name in enumerate(NAMES):
    col = t[:, i]
    print(f"  {name:22s} at {col.median():9.0f} / {col.max():9.0f}   (+{(col - prev).median():8.0f})")
    prev = col
