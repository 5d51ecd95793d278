"""ctypes / numpy front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module.  The product path (``coda_neurips2023_amd``) never does.

``TorchExt`` exposes the nine functions of the reference's ``pointnet2._ext``
pybind module (third_party_pointnet2/pointnet2/_ext_src/src/bindings.cpp:9-22)
on CPU torch tensors, so that the reference's own Python layers can be imported
on top of it when the golden fixtures are generated (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpointnet2_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/pointnet2_oracle.c + box_giou_oracle.c with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("pointnet2_oracle.c", "box_giou_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_opt_n_threads.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


DEFAULT_FMA_MODE = 1  # same default as the library (include/coda_pointnet2.h, distance arithmetic mode)


def set_fma_mode(mode):
    lib().oracle_set_fma_mode(ctypes.c_int(mode))


def get_fma_mode():
    return int(lib().oracle_get_fma_mode())


def opt_n_threads(n):
    return lib().oracle_opt_n_threads(ctypes.c_int(n))


def furthest_point_sampling(xyz, m):
    xyz, p = _f(xyz)
    b, n, _ = xyz.shape
    out = np.zeros((b, m), np.int32)
    lib().oracle_furthest_point_sampling(b, n, m, p, out.ctypes.data_as(ctypes.c_void_p))
    return out


def gather_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().oracle_gather_points(b, c, n, m, pp, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().oracle_gather_points_grad(b, c, n, m, pg, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, pn = _f(new_xyz)
    xyz, px = _f(xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    out = np.zeros((b, m, nsample), np.int32)
    lib().oracle_ball_query(b, n, m, ctypes.c_float(radius), nsample, pn, px,
                            out.ctypes.data_as(ctypes.c_void_p))
    return out


def group_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib().oracle_group_points(b, c, n, npoints, nsample, pp, pi,
                              out.ctypes.data_as(ctypes.c_void_p))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().oracle_group_points_grad(b, c, n, npoints, nsample, pg, pi,
                                   out.ctypes.data_as(ctypes.c_void_p))
    return out


def three_nn(unknown, known):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().oracle_three_nn(b, n, m, pu, pk, dist2.ctypes.data_as(ctypes.c_void_p),
                          idx.ctypes.data_as(ctypes.c_void_p))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().oracle_three_interpolate(b, c, m, n, pp, pi, pw, out.ctypes.data_as(ctypes.c_void_p))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    lib().oracle_three_interpolate_grad(b, c, n, m, pg, pi, pw,
                                        out.ctypes.data_as(ctypes.c_void_p))
    return out


class TorchExt:
    """The nine ``pointnet2._ext`` functions over CPU torch tensors (oracle)."""

    @staticmethod
    def _t(a):
        import torch
        return torch.from_numpy(a)

    def gather_points(self, points, idx):
        return self._t(gather_points(points.numpy(), idx.numpy()))

    def gather_points_grad(self, grad_out, idx, n):
        return self._t(gather_points_grad(grad_out.numpy(), idx.numpy(), n))

    def furthest_point_sampling(self, points, nsamples):
        return self._t(furthest_point_sampling(points.detach().numpy(), nsamples))

    def three_nn(self, unknowns, knows):
        d, i = three_nn(unknowns.numpy(), knows.numpy())
        return self._t(d), self._t(i)

    def three_interpolate(self, points, idx, weight):
        return self._t(three_interpolate(points.detach().numpy(), idx.numpy(), weight.numpy()))

    def three_interpolate_grad(self, grad_out, idx, weight, m):
        return self._t(three_interpolate_grad(grad_out.numpy(), idx.numpy(), weight.numpy(), m))

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return self._t(ball_query(new_xyz.detach().numpy(), xyz.detach().numpy(), radius, nsample))

    def group_points(self, points, idx):
        return self._t(group_points(points.detach().numpy(), idx.numpy()))

    def group_points_grad(self, grad_out, idx, n):
        return self._t(group_points_grad(grad_out.numpy(), idx.numpy(), n))
