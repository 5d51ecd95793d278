#!/usr/bin/env python3
"""dev: what a co-tenant on a side stream costs the encoder's attention forward (three-piece kernels, 8 scenes x 4 heads x
2048 x 2048, 512 workgroups = two per CU): the forward alone, next to the furthest point sampling on a high-priority /
default-priority stream, and next to synthetic co-tenants (tools/hog.hip: 1 / 8 / 32 workgroups that hold a CU each).
Result: ANY co-tenant, even one workgroup on one CU, costs the forward ~100 us of its 290 --
workgroups go to the shader engines round-robin, the engine with the busy CU gets its full share and runs the surplus as a
second round (tools/where_probe.py); a grid cut to the free slots (tried: `reserved`) does not change that."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coda_neurips2023_amd import _lib
from coda_neurips2023_amd.pointnet2 import pointnet2_utils as pu

dev = torch.device("cuda:0")
lib = _lib.load()
l = s = 2048; b, h, d = 8, 4, 64
q, k, v = (torch.randn(n, b, h, d, device=dev) for n in (l, s, s))
out = torch.empty_like(q); lse = torch.empty((b, h, l), device=dev)
xyz = torch.rand(8, 20000, 3, device=dev) * 6.0


def fwd(reserved):
    st = lib.coda_mha_fwd_opt_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), b, h, l, s, d,
                                  h * d, h * d, h * d, d ** -0.5, 0.1, 77, None, 2, _lib.current_stream_handle())
    _lib.check(st, "fwd")


def timed(n, reserved):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in evs:
        e0.record(); fwd(reserved); e1.record()
    return evs


for reserved in (0,):
    for _ in range(5):
        fwd(reserved)
    torch.cuda.synchronize()
    evs = timed(20, reserved); torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    print(f"alone, reserved {reserved}: median {t[10]:.1f} us  min {t[0]:.1f}  max {t[-1]:.1f}")

for prio in (-1, 0):
    side = torch.cuda.Stream(device=dev, priority=prio)
    for reserved in (0,):
        res = []
        for rep in range(6):
            torch.cuda.synchronize()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                f0.record(side)
                idx = pu.furthest_point_sample(xyz, 2048)
                f1.record(side)
            evs = timed(5, reserved)
            torch.cuda.synchronize()
            res.append((f0.elapsed_time(f1) * 1e3, [e0.elapsed_time(e1) * 1e3 for e0, e1 in evs]))
        fps_us = sorted(r[0] for r in res)[len(res) // 2]
        per = [sorted(r[1][i] for r in res)[len(res) // 2] for i in range(5)]
        print(f"next to the sampling (stream priority {prio}), reserved {reserved}: sampling {fps_us:.0f} us; "
              f"five forwards back to back: " + " ".join(f"{x:.0f}" for x in per))

# ---- synthetic co-tenants (tools/hog.hip): is it the sampling kernel, or any second queue that holds CUs?
import ctypes
hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libhog.so"))
hog.hog_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
hout = torch.empty(64 * 1024, device=dev); hsrc = torch.rand(1 << 20, device=dev)
side = torch.cuda.Stream(device=dev, priority=-1)
for mode, nwg, threads, lds, iters, what in [(0, 8, 1024, 160 * 1024, 12000, "8 workgroups x 1024 threads, whole LDS, VALU chain"),
                                              (1, 8, 1024, 160 * 1024, 6000, "the same + barrier and LDS exchange per step"),
                                              (2, 8, 1024, 160 * 1024, 3000, "the same + a global load per step"),
                                              (0, 1, 1024, 160 * 1024, 12000, "ONE workgroup, VALU chain"),
                                              (0, 8, 64, 160 * 1024, 12000, "8 workgroups of ONE wave, whole LDS"),
                                              (0, 32, 1024, 160 * 1024, 12000, "32 workgroups x 1024 threads")]:
    for reserved in (0,):
        res = []
        for rep in range(4):
            torch.cuda.synchronize()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                f0.record(side)
                assert hog.hog_launch(mode, nwg, threads, iters, lds, hout.data_ptr(), hsrc.data_ptr(), side.cuda_stream) == 0
                f1.record(side)
            evs = timed(5, reserved)
            torch.cuda.synchronize()
            res.append((f0.elapsed_time(f1) * 1e3, [e0.elapsed_time(e1) * 1e3 for e0, e1 in evs]))
        hog_us = sorted(r[0] for r in res)[len(res) // 2]
        per = [sorted(r[1][i] for r in res)[len(res) // 2] for i in range(5)]
        print(f"next to {what} ({hog_us:.0f} us), reserved {reserved}: " + " ".join(f"{x:.0f}" for x in per))
