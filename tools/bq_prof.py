"""Where a wave of the ball query's eight-lanes-per-centre kernel spends its time (round 5).

Builds a private copy of csrc/ball_query*.hip with -DCODA_BQ_PROF (shader-clock sums of the kernel's phases over all
waves) into tools/_build/libbq_prof.so -- on the CPU container: `python tools/bq_prof.py build` -- and on the GPU box
runs the fused operator on the bench's scenes:  python tools/bq_prof.py [scenes] [points]

Phases: 0 centre load, 1 cell-table reads + piece table, 2 first gathers issued, 3 the candidate loop, 4 the sorting
network, 5 output (coordinate gathers + stores)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_build", "libbq_prof.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = os.path.join(ROOT, "coda_neurips2023_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-munsafe-fp-atomics", "-DCODA_BQ_PROF", "-I" + os.path.join(ROOT, "include"), "-shared",
           os.path.join(src, "ball_query.hip"), os.path.join(src, "ball_query_grid.hip"), os.path.join(src, "version.hip"),
           "-o", OUT]
    subprocess.check_call(cmd)
    print("built", OUT)


def main():
    if sys.argv[1:2] == ["build"]:
        return build()
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from coda_neurips2023_amd.pointnet2 import _ext
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    m, ns = 2048, 64
    lib = ctypes.CDLL(OUT)
    lib.coda_ball_query_workspace_bytes.restype = ctypes.c_size_t
    pcs = [make_batch(8, n, seed=2000 + i)[0] for i in range(max(b // 8, 1))]
    xyz = torch.from_numpy(np.concatenate(pcs)[:b]).cuda().contiguous()
    inds = _ext.furthest_point_sampling(xyz, m)
    new = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = torch.empty((b, m, ns), dtype=torch.int32, device="cuda")
    grp = torch.empty((b, m, ns, 3), dtype=torch.float32, device="cuda")
    wsb = lib.coda_ball_query_workspace_bytes(b, n, m, ns)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    P = ctypes.c_void_p
    args = (P(new.data_ptr()), P(xyz.data_ptr()), P(idx.data_ptr()), P(grp.data_ptr()), b, n, m, ctypes.c_float(0.2), ns, 3,
            P(ws.data_ptr()), ctypes.c_size_t(wsb), P(0))
    host = (ctypes.c_ulonglong * (16384 * 12))()
    for _ in range(3):
        assert lib.coda_query_and_group_xyz_f32(*args) == 0
    torch.cuda.synchronize()
    lib.coda_bq_prof_read(host, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    assert lib.coda_query_and_group_xyz_f32(*args) == 0
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    assert lib.coda_bq_prof_read(host, 1) == 0
    pw = np.frombuffer(host, dtype=np.uint64).astype(np.float64).reshape(16384, 12)
    bw = pw[8192:]
    bw = bw[bw[:, 9] > 0]
    if len(bw):
        tb = bw[:, :5].sum(1)
        print(f"  build kernel, {len(bw)} waves: load+zero {bw[:, 0].mean():.0f}, count {bw[:, 1].mean():.0f}, reduce+barrier "
              f"{bw[:, 2].mean():.0f}, scan {bw[:, 3].mean():.0f}, scatter {bw[:, 4].mean():.0f} clocks; per wave "
              f"{tb.mean():.0f} (slowest {tb.max():.0f})")
    pw = pw[:8192]
    pw = pw[pw[:, 9] > 0]
    p = pw.sum(0)
    waves = p[9]
    ref = _ext.query_and_group_xyz(new, xyz, 0.2, ns, True, channels_last=True)
    assert torch.equal(ref[0], idx), "the instrumented build disagrees with the library"
    print(f"B={b} n={n}: {us:.1f} us for build + query with the probes in; {waves:.0f} waves, {p[8] / waves:.1f} steps per lane "
          f"(longest list of a wave / 8), {p[10] / waves:.1f} hits per centre (lane 0's)")
    names = ["centre load", "cell table + pieces", "first gathers", "candidate loop", "sorting network", "output"]
    tot = p[:6].sum() / waves
    for i, nm in enumerate(names):
        print(f"  {nm:22s} {p[i] / waves:9.0f} clocks per wave  ({p[i] / waves / tot * 100:4.1f} %)")
    tw = pw[:, :6].sum(1)
    order = np.argsort(tw)
    for q in (50, 90, 99, 100):
        i = order[min(len(order) - 1, int(len(order) * q / 100))] if q < 100 else order[-1]
        print(f"  p{q:<3d} wave: {tw[i]:7.0f} clocks, {pw[i, 8]:4.0f} steps, lane 0's centre {pw[i, 10]:4.0f} hits; phases "
              + " ".join(f"{pw[i, c]:.0f}" for c in range(6)))
    slow = order[-max(1, len(order) // 50):]
    print(f"  slowest 2 % of the waves: mean steps {float(pw[slow, 8].mean()):.1f}, phases "
          + " ".join(f"{pw[slow, c].mean():.0f}" for c in range(6)))
    print(f"  total {tot:.0f} clocks per wave = {tot / 2.4e3:.2f} us at 2.4 GHz; slowest wave {pw[:, :6].sum(1).max():.0f}, "
          f"fastest {pw[:, :6].sum(1).min():.0f}")


if __name__ == "__main__":
    main()
