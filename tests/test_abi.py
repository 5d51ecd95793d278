"""The C-ABI library loads on a CPU-only host and exports every symbol that
include/*.h declares (no compute calls: there is no GPU here)."""
import ctypes
import glob
import os
import re

import pytest

from coda_neurips2023_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(coda_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_nine_reference_ops():
    names = declared_symbols()
    for op in ["furthest_point_sampling", "gather_points", "gather_points_grad", "ball_query",
               "group_points", "group_points_grad", "three_nn", "three_interpolate",
               "three_interpolate_grad"]:
        assert f"coda_{op}_f32" in names, op


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build() first")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"libcoda_hip.so does not export {name}"


def test_python_signatures_cover_the_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.coda_version().decode().startswith("coda_hip gfx950")


def test_workspace_queries_run_without_gpu():
    lib = _lib.load()
    assert lib.coda_furthest_point_sampling_workspace_bytes(8, 2048, 256) == 0
    assert lib.coda_furthest_point_sampling_workspace_bytes(8, 20000, 2048) == 8 * 20000 * 16  # sorted records
    assert lib.coda_furthest_point_sampling_workspace_bytes(8, 100000, 2048) == 8 * 100000 * 4
    assert lib.coda_ball_query_workspace_bytes(8, 20000, 2048, 64) >= 0
    # attention backward (include/coda_attention.h): dS (+ the bf16 K^T pieces of the dQ GEMM, 24 KB per head and 64 keys) for
    # the encoder's long sequences, the key blocks' partial dQ tiles
    # for the decoder's short query sequences (blocks of 128 keys from 1024 keys on, of 32 below), nothing otherwise
    ws = lib.coda_mha_bwd_ws_bytes
    assert ws(8, 4, 2048, 2048, 64) == 4 * 8 * 4 * 2048 * 2048 + 8 * 4 * (2048 // 64) * 24576  # dS + the K^T pieces of dQ
    assert ws(8, 4, 256, 2048, 64) == 4 * 8 * 4 * (2048 // 128) * 256 * 64
    assert ws(8, 4, 256, 256, 64) == 4 * 8 * 4 * (256 // 32) * 256 * 64
    assert ws(8, 4, 512, 2048, 64) == 4 * 8 * 4 * 16 * 512 * 64
    for b, h, l, s, d in [(8, 4, 256, 2048, 128), (8, 4, 100, 2048, 64), (8, 4, 256, 1100, 64), (8, 4, 256, 77, 64),
                          (0, 4, 256, 256, 64), (8, 4, 1024, 256, 64)]:
        assert ws(b, h, l, s, d) == 0, (b, h, l, s, d)


def test_python_signatures_have_the_arity_of_the_header():
    """Every ctypes signature lists exactly as many arguments as the C declaration (a missing
    entry silently passes the trailing stream handle as a 32-bit int)."""
    decl = {}
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for name, args in re.findall(r"\b(coda_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
            args = args.strip()
            decl[name] = 0 if args in ("", "void") else len(args.split(","))
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        assert decl[name] == len(argtypes), f"{name}: header has {decl[name]} parameters, ctypes lists {len(argtypes)}"
