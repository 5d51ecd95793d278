"""Where a round of the bucketed FPS kernel spends its time.

Builds a private copy of csrc/fps.hip with -DCODA_FPS_PROF (shader-clock sums of the phases of a round per wave,
see the macros there) into tools/_build/libfps_prof.so -- on the CPU container: `python tools/fps_prof.py build` --
and on the GPU box runs it on the bench's scenes:

    python tools/fps_prof.py [n] [m]

Phases: 0 box tests + ballot, 1 bucket updates, 2 the wave's candidate, 3 posting it (LDS), 4 the barrier (incl. the
wait for the slowest wave), 5 the read-back of the winner; 6 = buckets updated, 7 = rounds with any update.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_build", "libfps_prof.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = os.path.join(ROOT, "coda_neurips2023_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-munsafe-fp-atomics", "-DCODA_FPS_PROF", "-I" + os.path.join(ROOT, "include"), "-shared",
           os.path.join(src, "fps.hip"), os.path.join(src, "version.hip"), "-o", OUT]
    subprocess.check_call(cmd)
    print("built", OUT)


def main():
    if sys.argv[1:2] == ["build"]:
        return build()
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    b = 8
    lib = ctypes.CDLL(OUT)
    lib.coda_furthest_point_sampling_workspace_bytes.restype = ctypes.c_size_t
    pc, _, _ = make_batch(b, n, seed=1234)
    xyz = torch.from_numpy(pc).cuda().contiguous()
    idx = torch.empty((b, m), dtype=torch.int32, device="cuda")
    wsb = lib.coda_furthest_point_sampling_workspace_bytes(b, n, m)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    args = (ctypes.c_void_p(xyz.data_ptr()), b, n, m, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
            ctypes.c_size_t(wsb), ctypes.c_void_p(0))
    for _ in range(3):
        st = lib.coda_furthest_point_sampling_f32(*args)
        assert st == 0, st
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    lib.coda_furthest_point_sampling_f32(*args)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    host = (ctypes.c_ulonglong * (64 * 16 * 8))()
    assert lib.coda_fps_prof_read(host) == 0
    p = np.frombuffer(host, dtype=np.uint64).reshape(64, 16, 8)[:b].astype(np.float64)
    nw = int((p[0, :, 7] > 0).sum()) or 8
    p = p[:, :nw]
    rounds = m - 1
    cyc = p[..., :6].sum(-1)            # cycles per wave over all rounds
    print(f"n={n} m={m}: {ms:.3f} ms with the probes in = {ms / rounds * 1e3:.3f} us/round; "
          f"{cyc.mean() / rounds:.0f} shader clocks per round -> {cyc.mean() / (ms * 1e3):.0f} MHz")
    names = ["box tests", "bucket updates", "wave candidate", "post", "barrier", "winner read-back"]
    for i, nm in enumerate(names):
        v = p[..., i] / rounds
        print(f"  {nm:18s} mean {v.mean():7.1f}  min-wave {v.min():7.1f}  max-wave {v.max():7.1f} clocks/round")
    upd = p[..., 6]
    print(f"  buckets updated per round: {upd.sum(1).mean() / rounds:.2f} per scene "
          f"({upd.mean() / rounds:.3f} per wave, busiest wave {upd.max() / rounds:.3f}); "
          f"a wave has work in {p[..., 7].mean() / rounds * 100:.1f} % of the rounds; "
          f"clocks per update {(p[..., 1].sum() / max(upd.sum(), 1)):.0f}")


if __name__ == "__main__":
    main()
