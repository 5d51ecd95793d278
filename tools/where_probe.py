#!/usr/bin/env python3
"""dev: on which XCD / CU do the workgroups of a 512-workgroup grid (256 threads, 57 KB of LDS: two per CU, the shape of the
encoder's attention forward) run -- alone, and next to one workgroup of another queue (tools/hog.hip)?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libhog.so"))
hog.hog_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
hog.where_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
hout = torch.empty(64 * 1024, device=dev); hsrc = torch.rand(1 << 20, device=dev)
out = torch.zeros(1024, dtype=torch.int32, device=dev); sink = torch.zeros(4, device=dev)
side = torch.cuda.Stream(device=dev, priority=-1)


def where(nwg, cotenant):
    torch.cuda.synchronize()
    if cotenant:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            assert hog.hog_launch(0, cotenant, 1024, 8000, 160 * 1024, hout.data_ptr(), hsrc.data_ptr(), side.cuda_stream) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hog.where_launch(4, 10, 57 * 1024, out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)  # (lets the co-tenant start)
    e0.record()
    assert hog.where_launch(nwg, 1500, 57 * 1024, out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    e1.record()
    torch.cuda.synchronize()
    o = out[:nwg].cpu().numpy().astype("uint32")
    xcc = o & 0xf
    hw = o >> 8
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = (xcc.astype(int) * 8 + se) * 32 + sh * 16 + cu
    import numpy as np
    match = float(np.mean(xcc == (np.arange(nwg) % 8)))
    per_xcc = np.bincount(xcc, minlength=8)
    per_cu = np.bincount(cuid)
    print(f"{nwg} workgroups, co-tenant workgroups {cotenant}: {e0.elapsed_time(e1) * 1e3:.0f} us; XCC == wg % 8 for {match * 100:.0f} %; "
          f"per XCC {per_xcc.tolist()}; distinct CUs {int((per_cu > 0).sum())}, workgroups per used CU max {int(per_cu.max())}, "
          f"CUs with 1 / 2 / 3+: {int((per_cu == 1).sum())} / {int((per_cu == 2).sum())} / {int((per_cu >= 3).sum())}")
    return xcc


for nwg in (512, 496):
    for cot in (0, 1, 8):
        where(nwg, cot)
        where(nwg, cot)
