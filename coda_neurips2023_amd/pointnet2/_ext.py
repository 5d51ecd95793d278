"""Drop-in for the reference's pybind11 module ``pointnet2._ext``.

Same nine functions, argument order, dtypes, output shapes and error behaviour
as third_party_pointnet2/pointnet2/_ext_src/src/bindings.cpp:9-22 and the C++
wrappers it binds (ball_query.cpp, group_points.cpp, interpolate.cpp,
sampling.cpp), but every function forwards raw device pointers to the C ABI of
``libcoda_hip.so`` (``include/coda_pointnet2.h``) on torch's current stream.

Error behaviour mirrored from ``_ext_src/include/utils.h:8-28``: a
non-contiguous tensor, a wrong dtype, or a CPU tensor raises ``RuntimeError``
("... must be a contiguous tensor", "... must be a float tensor",
"... must be an int tensor", "CPU not supported").

Two extra entry points expose the fused kernels of this build
(``query_and_group_xyz``); they are not part of the reference module.
"""
import contextlib

import torch

from .. import _lib

# Optional per-operator device timing (bench.py): name -> [(start, end) events],
# recorded on the stream the kernels are launched on.  None = disabled.
_TIMING = None


def enable_kernel_timing(names):
    """Start recording HIP events around the named operators; returns the store."""
    global _TIMING
    _TIMING = {n: [] for n in names}
    return _TIMING


def disable_kernel_timing():
    global _TIMING
    _TIMING = None


@contextlib.contextmanager
def _timed(name):
    if _TIMING is None or name not in _TIMING:
        yield
        return
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    yield
    end.record()
    _TIMING[name].append((start, end))


def _check_contiguous(x, name):
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def _check_float(x, name):
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")


def _check_int(x, name):
    if x.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _check_device(ref, *others):
    if not ref.is_cuda:
        raise RuntimeError("CPU not supported")
    for name, x in others:
        if not x.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")


def _stream():
    return _lib.current_stream_handle()


def _ptr(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


def set_fps_waves(waves=0):
    """Waves per workgroup of the bucketed FPS kernels for the calls of THIS thread: 0 (default: 16, or 8 with two
    workgroups per scene), 8 or 16; same indices (a per-call argument of coda_furthest_point_sampling_opt_f32)."""
    if waves not in (0, 8, 16):
        raise ValueError("waves must be 0, 8 or 16")
    _lib.set_option("fps_waves", waves)


def check_sampling_status():
    """Raises if a two-workgroup furthest-point-sampling launch of this process lost its partner workgroup (its indices
    are wrong; include/coda_pointnet2.h, CODA_ELOST).  One read of a pinned host word: free to call anywhere; it speaks
    for every launch whose stream has been synchronised.  ``furthest_point_sampling`` itself refuses to launch on top
    of an unacknowledged loss, and ``SamplingPrefetcher.take`` calls this when it hands a finished sampling over."""
    lib = _lib.load()
    if lib.coda_fps_lost_partner_events(0):
        _lib.check(_lib.CODA_ELOST, "furthest_point_sampling")


def furthest_point_sampling(points, nsamples, _dbg=None):
    """(B,N,3) f32 -> (B,nsamples) i32.  sampling.cpp:67-88.  (``_dbg = (spin_limit, drop_half)``: the test hook
    coda_furthest_point_sampling_dbg_f32 that makes the two-workgroup kernel lose its partner.)"""
    _check_contiguous(points, "points")
    _check_float(points, "points")
    _check_device(points)
    lib = _lib.load()
    b, n = points.size(0), points.size(1)
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    ws_bytes = lib.coda_furthest_point_sampling_workspace_bytes(b, n, nsamples)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=points.device) if ws_bytes else None
    with torch.cuda.device(points.device), _timed("furthest_point_sampling"):
        if _dbg is not None:
            st = lib.coda_furthest_point_sampling_dbg_f32(_ptr(points), b, n, nsamples, _ptr(out), _ptr(ws), ws_bytes,
                                                          int(_dbg[0]), int(_dbg[1]), _stream())
        else:
            st = lib.coda_furthest_point_sampling_opt_f32(_ptr(points), b, n, nsamples, _ptr(out), _ptr(ws), ws_bytes,
                                                          _lib.opt("distance_mode"), _lib.opt("fps_waves"), _stream())
    _lib.check(st, "furthest_point_sampling")
    return out


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M).  sampling.cpp:17-41."""
    _check_contiguous(points, "points")
    _check_contiguous(idx, "idx")
    _check_float(points, "points")
    _check_int(idx, "idx")
    _check_device(points, ("idx", idx))
    lib = _lib.load()
    b, c, n = points.shape
    m = idx.size(1)
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed("gather_points"):
        st = lib.coda_gather_points_f32(_ptr(points), _ptr(idx), _ptr(out), b, c, n, m, _stream())
    _lib.check(st, "gather_points")
    return out


# The two scatter-add adjoints: deterministic by default (64-bit fixed-point accumulation: reproducible bit for bit,
# include/coda_pointnet2.h); CODA_SCATTER=atomic selects the float atomics of the plain entry points -- the reference's
# own, order-dependent form.
_SCATTER_ATOMIC = __import__("os").environ.get("CODA_SCATTER", "det") == "atomic"


def _det_workspace(lib, b, c, n, device):
    nbytes = lib.coda_scatter_add_det_workspace_bytes(b, c, n)
    return (torch.empty(nbytes // 8 + 1, dtype=torch.int64, device=device), nbytes) if nbytes else (None, 0)


def gather_points_grad(grad_out, idx, n):
    """(B,C,M) f32, (B,M) i32, n -> (B,C,n).  sampling.cpp:43-66."""
    _check_contiguous(grad_out, "grad_out")
    _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out")
    _check_int(idx, "idx")
    _check_device(grad_out, ("idx", idx))
    lib = _lib.load()
    b, c, m = grad_out.shape
    out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed("gather_points_grad"):
        if _SCATTER_ATOMIC:
            st = lib.coda_gather_points_grad_f32(_ptr(grad_out), _ptr(idx), _ptr(out), b, c, n, m, _stream())
        else:
            ws, ws_bytes = _det_workspace(lib, b, c, n, grad_out.device)
            st = lib.coda_gather_points_grad_det_f32(_ptr(grad_out), _ptr(idx), _ptr(out), b, c, n, m, _ptr(ws), ws_bytes,
                                                     _stream())
    _lib.check(st, "gather_points_grad")
    return out


_ROUTES = {"auto": 0, "grid": 1, "scan": 2}


def _route(algorithm):
    """``algorithm``: "auto" (the cell table in a workspace where it applies, else the scan) | "grid" | "scan" (brute
    force; include/coda_pointnet2.h).  All give identical results; the non-default
    ones exist for the parity tests and A/B timing.  The route is an argument of the call (coda_ball_query_opt_f32):
    "auto" defers to the calling thread's option, then to the library default (CODA_BQ)."""
    if algorithm not in _ROUTES:
        raise ValueError(f"unknown ball_query algorithm {algorithm!r}")
    return _ROUTES[algorithm] or _lib.opt("bq_route")


def _ball_query_workspace(lib, b, n, m, nsample, device, algorithm):
    """Workspace of the cell-binned search ("auto" / "grid"); "scan" needs none."""
    if algorithm not in _ROUTES:
        raise ValueError(f"unknown ball_query algorithm {algorithm!r}")
    ws_bytes = 0 if algorithm == "scan" else lib.coda_ball_query_workspace_bytes(b, n, m, nsample)
    if algorithm == "grid" and ws_bytes == 0:
        raise RuntimeError("grid ball_query not applicable to this shape (n < 1024 or nsample > 128)")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device) if ws_bytes else None
    return ws, ws_bytes


def ball_query(new_xyz, xyz, radius, nsample, algorithm="auto"):
    """(B,M,3), (B,N,3), radius, nsample -> (B,M,nsample) i32.  ball_query.cpp:11-35."""
    _check_contiguous(new_xyz, "new_xyz")
    _check_contiguous(xyz, "xyz")
    _check_float(new_xyz, "new_xyz")
    _check_float(xyz, "xyz")
    _check_device(new_xyz, ("xyz", xyz))
    lib = _lib.load()
    b, n = xyz.size(0), xyz.size(1)
    m = new_xyz.size(1)
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
    ws, ws_bytes = _ball_query_workspace(lib, b, n, m, nsample, new_xyz.device, algorithm)
    with torch.cuda.device(new_xyz.device), _timed("ball_query"):
        st = lib.coda_ball_query_opt_f32(_ptr(new_xyz), _ptr(xyz), _ptr(idx), b, n, m, float(radius), int(nsample),
                                         _ptr(ws), ws_bytes, _lib.opt("distance_mode"), _route(algorithm), _stream())
    _lib.check(st, "ball_query")
    return idx


def group_points(points, idx):
    """(B,C,N) f32, (B,M,S) i32 -> (B,C,M,S).  group_points.cpp:15-38."""
    _check_contiguous(points, "points")
    _check_contiguous(idx, "idx")
    _check_float(points, "points")
    _check_int(idx, "idx")
    _check_device(points, ("idx", idx))
    lib = _lib.load()
    b, c, n = points.shape
    npoints, nsample = idx.size(1), idx.size(2)
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed("group_points"):
        st = lib.coda_group_points_f32(_ptr(points), _ptr(idx), _ptr(out), b, c, n, npoints,
                                       nsample, _stream())
    _lib.check(st, "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """(B,C,M,S) f32, (B,M,S) i32, n -> (B,C,n).  group_points.cpp:40-63."""
    _check_contiguous(grad_out, "grad_out")
    _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out")
    _check_int(idx, "idx")
    _check_device(grad_out, ("idx", idx))
    lib = _lib.load()
    b, c = grad_out.size(0), grad_out.size(1)
    npoints, nsample = idx.size(1), idx.size(2)
    out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed("group_points_grad"):
        if _SCATTER_ATOMIC:
            st = lib.coda_group_points_grad_f32(_ptr(grad_out), _ptr(idx), _ptr(out), b, c, n, npoints, nsample, _stream())
        else:
            ws, ws_bytes = _det_workspace(lib, b, c, n, grad_out.device)
            st = lib.coda_group_points_grad_det_f32(_ptr(grad_out), _ptr(idx), _ptr(out), b, c, n, npoints, nsample,
                                                    _ptr(ws), ws_bytes, _stream())
    _lib.check(st, "group_points_grad")
    return out


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) f32, idx (B,n,3) i32].  interpolate.cpp:19-46."""
    _check_contiguous(unknowns, "unknowns")
    _check_contiguous(knows, "knows")
    _check_float(unknowns, "unknowns")
    _check_float(knows, "knows")
    _check_device(unknowns, ("knows", knows))
    lib = _lib.load()
    b, n = unknowns.size(0), unknowns.size(1)
    m = knows.size(1)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    with torch.cuda.device(unknowns.device), _timed("three_nn"):
        st = lib.coda_three_nn_opt_f32(_ptr(unknowns), _ptr(knows), _ptr(dist2), _ptr(idx), b, n, m,
                                       _lib.opt("distance_mode"), _stream())
    _lib.check(st, "three_nn")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """(B,C,m) f32, (B,n,3) i32, (B,n,3) f32 -> (B,C,n).  interpolate.cpp:48-74."""
    _check_contiguous(points, "points")
    _check_contiguous(idx, "idx")
    _check_contiguous(weight, "weight")
    _check_float(points, "points")
    _check_int(idx, "idx")
    _check_float(weight, "weight")
    _check_device(points, ("idx", idx), ("weight", weight))
    lib = _lib.load()
    b, c, m = points.shape
    n = idx.size(1)
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device), _timed("three_interpolate"):
        st = lib.coda_three_interpolate_opt_f32(_ptr(points), _ptr(idx), _ptr(weight), _ptr(out), b, c, m, n,
                                                _lib.opt("distance_mode"), _stream())
    _lib.check(st, "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """(B,C,n) f32, (B,n,3) i32, (B,n,3) f32, m -> (B,C,m).  interpolate.cpp:75-101."""
    _check_contiguous(grad_out, "grad_out")
    _check_contiguous(idx, "idx")
    _check_contiguous(weight, "weight")
    _check_float(grad_out, "grad_out")
    _check_int(idx, "idx")
    _check_float(weight, "weight")
    _check_device(grad_out, ("idx", idx), ("weight", weight))
    lib = _lib.load()
    b, c, n = grad_out.shape
    out = torch.empty((b, c, m), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device), _timed("three_interpolate_grad"):
        st = lib.coda_three_interpolate_grad_f32(_ptr(grad_out), _ptr(idx), _ptr(weight),
                                                 _ptr(out), b, c, n, m, _stream())
    _lib.check(st, "three_interpolate_grad")
    return out


# ---- fused entry points of this build (not in the reference module) ------------------

def query_and_group_xyz(new_xyz, xyz, radius, nsample, normalize_xyz, algorithm="auto",
                        channels_last=False):
    """ball_query + xyz grouping + centring (+ 1/radius) in one kernel.

    Returns (idx (B,M,S) i32, grouped_xyz (B,3,M,S) f32, or (B,M,S,3) when
    ``channels_last``); replaces the
    ball_query / transpose / group_points / sub / div sequence of
    QueryAndGroup.forward (pointnet2_utils.py:331-349).
    """
    _check_contiguous(new_xyz, "new_xyz")
    _check_contiguous(xyz, "xyz")
    _check_float(new_xyz, "new_xyz")
    _check_float(xyz, "xyz")
    _check_device(new_xyz, ("xyz", xyz))
    lib = _lib.load()
    b, n = xyz.size(0), xyz.size(1)
    m = new_xyz.size(1)
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
    shape = (b, m, nsample, 3) if channels_last else (b, 3, m, nsample)
    grouped = torch.empty(shape, dtype=torch.float32, device=new_xyz.device)
    ws, ws_bytes = _ball_query_workspace(lib, b, n, m, nsample, new_xyz.device, algorithm)
    with torch.cuda.device(new_xyz.device), _timed("query_and_group_xyz"):
        st = lib.coda_query_and_group_xyz_opt_f32(_ptr(new_xyz), _ptr(xyz), _ptr(idx), _ptr(grouped),
                                                  b, n, m, float(radius), int(nsample),
                                                  (1 if normalize_xyz else 0) | (2 if channels_last else 0),
                                                  _ptr(ws), ws_bytes, _lib.opt("distance_mode"), _route(algorithm),
                                                  _stream())
    _lib.check(st, "query_and_group_xyz")
    return idx, grouped
