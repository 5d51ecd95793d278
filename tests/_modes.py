"""Distance-arithmetic mode helpers shared by conftest.py and the GPU test modules."""
import os


def set_distance_mode(mode):
    """Put the oracle AND this thread's calls into the HIP library into distance-arithmetic mode `mode`
    (include/coda_pointnet2.h).  Golden fixtures carry the mode they were generated in."""
    from oracle import pointnet2_oracle as O
    O.set_fma_mode(int(mode))
    from coda_neurips2023_amd import _lib
    _lib.set_option("distance_mode", int(mode))  # a per-call argument of the library; sticky for this thread


def fixture_mode(npz):
    """Distance mode a golden .npz was generated in (files from before the key existed: 0)."""
    return int(npz["fma_mode"]) if "fma_mode" in npz.files else 0
