"""Per-call options instead of process-wide switches (the reference's ops are stateless,
third_party_pointnet2/pointnet2/_ext_src/src/ball_query.cpp:11-35): the C library exports no setter, the Python side
carries the values in a thread-local record, and two threads that run the attention core with different MFMA operand
types at the same time each get their own type's results."""
import threading

import pytest
import torch

from coda_neurips2023_amd import _lib


def test_the_library_exports_no_process_wide_setter():
    import ctypes
    import os
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in ("coda_set_distance_mode", "coda_set_fps_waves", "coda_set_ball_query_route", "coda_mha_set_mfma_dtype",
                 "coda_gemm_set_tuning"):
        assert not hasattr(lib, name), name
    # ... and nothing else that looks like one: the only mutable library state left is the measurement aid
    # coda_mha_timing_enable / coda_pointnet2 timing (bench.py's live roofline figures, off by default)
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout.split()
    setters = [x for x in syms if x.startswith("coda_") and "_set_" in x]
    assert not setters, setters
    for name in ("coda_furthest_point_sampling_opt_f32", "coda_ball_query_opt_f32", "coda_query_and_group_xyz_opt_f32",
                 "coda_three_nn_opt_f32", "coda_three_interpolate_opt_f32", "coda_mha_fwd_opt_f32",
                 "coda_mha_bwd_parts_opt_f32"):
        assert hasattr(lib, name), name


def test_options_are_thread_local_and_scoped():
    assert _lib.opt("mfma_dtype") == -1 and _lib.opt("bq_route") == 0
    seen = {}

    def other():
        seen["before"] = _lib.opt("mfma_dtype")
        _lib.set_option("mfma_dtype", 2)
        seen["after"] = _lib.opt("mfma_dtype")

    with _lib.options(mfma_dtype=1, fps_waves=8):
        assert _lib.opt("mfma_dtype") == 1 and _lib.opt("fps_waves") == 8
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert _lib.opt("mfma_dtype") == 1  # the other thread's sticky value stayed there
        with _lib.options(mfma_dtype=0):
            assert _lib.opt("mfma_dtype") == 0
        assert _lib.opt("mfma_dtype") == 1
    assert seen == {"before": -1, "after": 2}
    assert _lib.opt("mfma_dtype") == -1 and _lib.opt("fps_waves") == 0
    with pytest.raises(KeyError):
        _lib.set_option("no_such_option", 1)


@pytest.mark.gpu
def test_two_threads_run_fp32_and_bf16_attention_concurrently(dev):
    from coda_neurips2023_amd import attention_core as core
    gen = torch.Generator().manual_seed(0)
    l, s, b, h, d = 256, 512, 2, 4, 64
    q, k, v = (torch.randn(n, b, h, d, generator=gen).to(dev) for n in (l, s, s))
    scale = d ** -0.5
    refs = {}
    for mode in ("fp32", "bf16"):
        with core.mfma_dtype(mode):
            assert core.get_mfma_dtype() == mode
            refs[mode] = core.attention(q, k, v, None, scale, 0.0, False)[0].clone()
    assert not torch.equal(refs["fp32"], refs["bf16"])  # the modes do differ on these inputs
    torch.cuda.synchronize()
    bad = []
    barrier = threading.Barrier(2)

    def worker(mode):
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream), core.mfma_dtype(mode):
            barrier.wait()
            for i in range(150):
                qq = q.clone().requires_grad_(True)
                out = core.attention(qq, k, v, None, scale, 0.0, False)[0]
                if not torch.equal(out, refs[mode]):
                    bad.append((mode, i, "forward"))
                out.sum().backward()  # the backward runs on an autograd thread with the mode captured in the forward
            stream.synchronize()

    threads = [threading.Thread(target=worker, args=(m,)) for m in ("fp32", "bf16")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad[:5]
    assert core.get_mfma_dtype() == "fp32"  # nothing leaked into this thread
