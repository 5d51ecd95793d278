"""Where a sub-tile of the set-abstraction MFMA kernels spends its time.

Builds a private copy of the library with csrc/sa_mfma.hip compiled -DCODA_SA_PROF (shader-clock sums per phase, wave
and workgroup, see the macros there) into tools/_build/libcoda_sa_prof.so -- on the CPU container:
`python tools/sa_prof.py build` -- and on the GPU box runs the pre-encoder's forward + backward on the bench's scenes
through it:

    python tools/sa_prof.py

Phases (clocks per 64-row sub-tile, mean over waves and the first 64 workgroups):
  fwd: 0 top barrier, 1 staging (BN+ReLU -> LDS), 2 barrier, 3 prefetch issue + MFMA loop, 4 alias barrier,
       5 epilogue (statistics, tile -> LDS), 6 barrier, 8 pooling scan (CODA_SA_POOL=fused only), 7 row store
  dx:  0 top barrier, 1 y_in tile -> LDS + barrier, 2 dy staging, 3 barrier, 4 prefetch issue + MFMA loop,
       5 epilogue (ReLU mask, sums, tile -> LDS), 6 barrier, 7 row store
  dw:  0 top barrier, 1 activation staging, 2 barrier, 3 dy staging, 4 barrier, 5 prefetch issue + MFMA loop
"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_build", "libcoda_sa_prof.so")
KINDS = ["fwd layer 3 (128->256, pooled)", "fwd layer 2 (64->128)", "dx layer 3", "dx layer 2", "dw layer 3", "dw layer 2"]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = os.path.join(ROOT, "coda_neurips2023_amd", "csrc")
    subprocess.check_call(["make", "-C", src])
    obj = os.path.join(ROOT, "tools", "_build", "sa_mfma_prof.o")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
             "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include")]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DCODA_SA_PROF", "-c", os.path.join(src, "sa_mfma.hip"), "-o", obj])
    others = [o for o in glob.glob(os.path.join(src, "_build", "*.o")) if not o.endswith("/sa_mfma.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others +
                          ["-L/opt/rocm/lib", "-lhipblaslt", "-o", OUT])
    print("built", OUT)


def main():
    if sys.argv[1:2] == ["build"]:
        return build()
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from coda_neurips2023_amd import _lib
    _lib.LIB_PATH = OUT
    from coda_neurips2023_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from coda_neurips2023_amd.synthetic_scenes import make_batch
    dev = torch.device("cuda:0")
    pc, _, _ = make_batch(8, 20000, seed=1234)
    xyz = torch.from_numpy(pc).to(dev)[..., :3].contiguous()
    sa = PointnetSAModuleVotes(radius=0.2, nsample=64, npoint=2048, mlp=[0, 64, 128, 256], normalize_xyz=True).to(dev).train()
    for _ in range(3):
        sa.zero_grad()
        _, feats, _ = sa(xyz, None)
        feats.square().sum().backward()
    torch.cuda.synchronize()
    lib = _lib.load()
    host = (ctypes.c_ulonglong * (6 * 64 * 4 * 16))()
    lib.coda_sa_prof_read.restype = ctypes.c_int
    assert lib.coda_sa_prof_read(host) == 0
    p = np.frombuffer(host, dtype=np.uint64).reshape(6, 64, 4, 16).astype(np.float64)
    for k, name in enumerate(KINDS):
        tiles = p[k, :, :, 9]
        if tiles.sum() == 0:
            continue
        per = p[k, :, :, :9].sum((0, 1)) / tiles.sum()
        print(f"{name}: {tiles.mean():.1f} sub-tiles per workgroup, {per.sum():.0f} clocks per sub-tile")
        print("   phases: " + "  ".join(f"{i}:{v:.0f}" for i, v in enumerate(per)))
        if p[k, :, :, 10:14].sum() > 0:
            extra = p[k, :, :, 10:14].sum((0, 1)) / tiles.sum()
            print(f"   scan detail per sub-tile: fetch + test {extra[0]:.0f}, quads without a group start {extra[1]:.0f}, "
                  f"quads with one {extra[2]:.0f} ({extra[3]:.2f} of them)")
        spread = p[k, :, :, :9].sum(-1) / np.maximum(tiles, 1)
        print(f"   per wave-workgroup clocks/sub-tile: min {spread.min():.0f} max {spread.max():.0f}")


if __name__ == "__main__":
    main()
