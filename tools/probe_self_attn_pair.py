#!/usr/bin/env python3
"""dev probe (round 5): would the decoder's 256 x 256 self-attention backward gain from running its dQ and dK/dV
kernels CONCURRENTLY (one merged launch)?  Times delta -> (dQ ; dK/dV) in order on one stream against delta -> dQ on
one stream + dK/dV on another, per pair, 200 pairs back to back."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coda_neurips2023_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P = ctypes.c_void_p


def run(l, s, b=8, h=4, d=64, reps=200):
    torch.manual_seed(0)
    q = torch.randn(l, b, h, d, device=dev)
    k = torch.randn(s, b, h, d, device=dev)
    v = torch.randn(s, b, h, d, device=dev)
    out = torch.empty_like(q)
    lse = torch.empty(b, h, l, device=dev)
    st = lib.coda_mha_fwd_f32(P(q.data_ptr()), P(k.data_ptr()), P(v.data_ptr()), None, P(out.data_ptr()), P(lse.data_ptr()),
                              b, h, l, s, d, h * d, h * d, h * d, ctypes.c_float(d ** -0.5), ctypes.c_float(0.1), 7, None,
                              _lib.current_stream_handle())
    assert st == 0, st
    dout = torch.randn_like(q)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(b, h, l, device=dev)

    def part(parts, stream):
        st = lib.coda_mha_bwd_parts_f32(P(q.data_ptr()), P(k.data_ptr()), P(v.data_ptr()), None, P(out.data_ptr()),
                                        P(lse.data_ptr()), P(dout.data_ptr()), P(dq.data_ptr()), P(dk.data_ptr()),
                                        P(dv.data_ptr()), P(delta.data_ptr()), b, h, l, s, d, h * d, h * d, h * d, 0, 0, 0,
                                        ctypes.c_float(d ** -0.5), ctypes.c_float(0.1), 7, None, parts, P(stream.cuda_stream))
        assert st == 0, st

    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    part(1, main)
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / reps * 1e3

    def serial():
        part(4, main)
        part(2, main)

    def pair():
        side.wait_stream(main)
        part(4, main)
        part(2, side)
        main.wait_stream(side)

    t_dq = timed(lambda: part(4, main))
    t_dkv = timed(lambda: part(2, main))
    print(f"{l} x {s}: dQ {t_dq:.1f} us, dK/dV {t_dkv:.1f} us, in order {timed(serial):.1f} us, on two streams {timed(pair):.1f} us")


run(256, 256)
run(256, 2048)
run(512, 512)
