"""Plain fp32 GEMMs of the host-side mirrors through the C ABI (include/coda_gemm.h).

`linear`, `mm`, `mm_tn` are `F.linear` / `torch.mm` on 2-D CUDA tensors, computed by the same
vendor library (hipBLASLt) PyTorch-ROCm calls, but through `coda_gemm_f32`, whose matmul plans
are cached per shape: ~3x less host time per launch-sized GEMM than `torch.mm` / `torch.addmm`
(which rebuild descriptors and re-query the heuristic on every call).  Row strides are passed
through, so row / column slices of packed buffers (in_proj weights, q|k|v activations) need no
copy.  These helpers are used inside hand-written autograd nodes only: they do not record
autograd history themselves.
"""
import os

import torch

from . import _lib

_USE_TORCH = os.environ.get("CODA_GEMM", "") == "torch"  # dev A/B switch

def _plain(*ts):
    """True when every operand is a 2-D fp32 CUDA tensor (what coda_gemm_f32 takes); anything else
    (fp64 reference runs of the same modules, CPU tensors of the oracle port) goes to torch's own GEMM."""
    return not _USE_TORCH and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 for t in ts)


def _rows(t):
    """2-D fp32 CUDA tensor with unit inner stride (copied only if it is not)."""
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


# Launch-sized products (the decoder's 2048 x 256 x 256 family, ~200 per training step) go to the own
# fp32-MFMA kernel coda_sgemm_f32 (csrc/gemm_nn.hip): about half the host time per call of the library path
# and 7 vs 11 us on the GPU; larger problems stay with hipBLASLt, which reaches 75-80 % of the fp32 matrix peak
# there (tools/bench_gemm.py).  CODA_SGEMM=0 switches the own kernel off (A/B).
_OWN_SMALL = os.environ.get("CODA_SGEMM", "1") != "0"
_OWN_MAX_MN = 2048 * 256


# ---- the large token-wise products on the bf16 matrix cores (csrc/gemm_x3.hip) -------------------------------------
# y = x W^T + b and dx = dy W with >= _X3_MIN_ROWS token rows: every fp32 operand as three bf16 pieces, six piece
# products, fp32 accumulation -- error against float64 within 2x of the native fp32 GEMM (tests/test_gemm_x3_gpu.py).
# The WEIGHT's pieces are kept per weight (both orientations) and refreshed once per optimizer step in one launch;
# CODA_GEMM_X3=0 sends everything back to the library (A/B switch, and the reference point of the parity tests).
_X3 = os.environ.get("CODA_GEMM_X3", "1") != "0"
_X3_MIN_ROWS = 4096
_X3_CHECK = os.environ.get("CODA_X3_CHECK", "0") == "1"
_X3_CHECK_PREV = None
_X3_FORCE = False     # tests: every shape the kernels take, not only the ones they win on


def set_x3(on, force=False, tn=None):
    """Development / test switch (tools/bench_gemm_x3.py, tests): route eligible products through the x3 kernels;
    ``force``: also the shapes on which the library is faster; ``tn``: the weight-gradient kernel too."""
    global _X3, _X3_FORCE, _X3_TN
    _X3 = bool(on)
    _X3_FORCE = bool(force)
    if tn is not None:
        _X3_TN = bool(tn)


class _Planes:
    __slots__ = ("base", "version", "epoch", "used", "nt", "nn", "declared")


x3_calls = 0          # products that took the x3 route (diagnostics: tests, tools)
_planes = {}          # (data_ptr, rows, cols, stride) of the BASE weight -> _Planes (holds the base: its storage stays alive)
_planes_epoch = 0


def _weight_base(w):
    """(base, first row, rows) when ``w`` is a weight -- a leaf that requires grad / an nn.Parameter -- or a row slice
    of one (in_proj_weight[e:2 * e]); None for anything else (activations never enter the cache: their storage is
    recycled by the allocator, and a cache keyed by address would serve another tensor's pieces)."""
    base = w._base if w._base is not None else w
    if not (base.requires_grad or isinstance(base, torch.nn.Parameter)) or not base.is_leaf:
        # ... or a tensor a caller has DECLARED a weight for this step (declare_weight: the decoder's concatenated
        # key / value in-projection rows): the entry holds the tensor, so a matching address is the same storage
        e = _planes.get((w.data_ptr(), w.shape[0], w.shape[1], w.stride(0))) if w.dim() == 2 else None
        if (e is not None and e.declared and e.base is not None and e.version == w._version
                and e.epoch == _planes_epoch):
            return e.base, 0, w.shape[0]
        return None
    if base.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1:
        return None
    if base is w:
        return base, 0, w.shape[0]
    if base.dim() != 2 or base.stride(1) != 1 or w.shape[1] != base.shape[1] or w.stride() != base.stride():
        # another 2-D view of a parameter (a 1 x 1 convolution's weight flattened to (Cout, Cin)): the view itself is
        # the cached matrix (it keeps the parameter's storage alive and shares its version counter)
        return (w, 0, w.shape[0]) if w.stride(0) >= w.shape[1] else None
    off = (w.data_ptr() - base.data_ptr()) // 4
    if off % base.stride(0):
        return None
    return base, off // base.stride(0), w.shape[0]


def _split_items(entries):
    import numpy as np
    table = np.zeros((len(entries), 5), dtype=np.int64)  # the CodaX3SplitItem layout: src, nt, nn, (rows, cols), ld
    for i, e in enumerate(entries):
        r, c = e.base.shape
        table[i] = (e.base.data_ptr(), e.nt.data_ptr() if e.nt is not None else 0,
                    e.nn.data_ptr() if e.nn is not None else 0, r | (c << 32), e.base.stride(0))
    st = _lib.load().coda_gemm_x3_split_f32(table.ctypes.data, len(entries), _lib.current_stream_handle())
    _lib.check(st, "coda_gemm_x3_split_f32")


def _weight_planes(base):
    key = (base.data_ptr(), base.shape[0], base.shape[1], base.stride(0))
    e = _planes.get(key)
    if e is None or e.base.device != base.device:
        e = _Planes()
        r, c = base.shape
        # tiled plane sets (include/coda_gemm.h): rows padded to the 128-row tile, zero-filled once (the split kernel
        # writes the real rows only); an orientation whose contraction length is not a multiple of 32 has none
        e.nt = torch.zeros(3 * (-(-r // 128) * 128) * c, dtype=torch.bfloat16, device=base.device) if c % 32 == 0 else None
        e.nn = torch.zeros(3 * (-(-c // 128) * 128) * r, dtype=torch.bfloat16, device=base.device) if r % 32 == 0 else None
        e.version, e.epoch, e.used, e.base, e.declared = -1, -1, -1, None, False
        _planes[key] = e
    if e.version != base._version or e.epoch != _planes_epoch or e.base is None:
        e.base = base
        with torch.cuda.device(base.device):
            _split_items([e])
        e.version, e.epoch = base._version, _planes_epoch
    e.used = _planes_epoch
    return e


def declare_weight(t):
    """``t`` is a weight-like matrix assembled for THIS step from parameters (``torch.cat`` of the decoder layers'
    key / value in-projection rows): split it now, both orientations, so that the products of this step's forward and
    backward that use it (the very tensor, not a slice) take the x3 route.  The entry keeps ``t`` alive until the next
    ``refresh_weight_planes`` (after the optimizer step) or until eight newer declarations have been made."""
    if not (_X3 and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
            and (t.shape[0] % 32 == 0 or t.shape[1] % 32 == 0) and t._base is None):
        return t
    e = _weight_planes(t)
    e.declared = True
    old = [k for k, v in _planes.items() if v.declared and v is not e]
    for k in old[:-7]:
        del _planes[k]
    return t


def refresh_weight_planes():
    """The weights have changed behind the version counters (this package's optimizer writes through raw pointers;
    so do checkpoint loaders that assign through ``p.data``): every cached set of pieces that was used since the last
    refresh is recomputed NOW, all of them in one launch on the current stream; the others are dropped.  Called by
    ``optim.step`` and -- through torch's global post-step hook -- after every torch optimizer step."""
    global _planes_epoch
    live = []
    for key, e in list(_planes.items()):
        # kept: used within the last few refreshes (a loop with two optimizers refreshes twice per step)
        if e.used > _planes_epoch - 4 and e.base is not None and e.base.is_cuda and not e.declared:
            live.append(e)
        else:
            del _planes[key]
    _planes_epoch += 1
    by_dev = {}
    for e in live:
        by_dev.setdefault(e.base.device, []).append(e)
    for dev, group in by_dev.items():
        with torch.cuda.device(dev):
            _split_items(group)
        for e in group:
            e.version, e.epoch = e.base._version, _planes_epoch


try:  # torch optimizers bump the version counters themselves; the hook only turns ~25 one-weight launches into one
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post
    _reg_post(lambda *_a, **_k: refresh_weight_planes() if _planes else None)
except Exception:  # pragma: no cover
    pass


def _x3_route(transa, transb, m, n, k, a, b, out, bias, accumulate):
    """The product through coda_gemm_x3_nt_f32 when it is one of the large token-wise ones with a weight as B;
    returns ``out`` or None ("not this route")."""
    if transa or m < _X3_MIN_ROWS or m % 128 or n % 64 or k % 32 or a.stride(0) % 4 or a.data_ptr() % 16:
        return None
    if not _X3_FORCE and (m < 8192 or n < 256):
        # measured (tools/bench_gemm_x3.py): with 128 or fewer output columns the 128-row tiles fill half the chip and
        # the library's small-tile kernels win (16 384 x 128 x 256: 16.3 vs 14.3 us), likewise below 8192 rows
        return None
    wb = _weight_base(b)
    if wb is None:
        return None
    base, r0, nrows = wb
    rows, cols = base.shape
    if (rows % 32 and cols % 32) or r0 % 64:
        return None
    e = _weight_planes(base)
    if transb:   # y = x W^T: W (n x k) = rows r0 .. of the base: output columns r0 .. of the nt set, whole contraction
        tiled, w_cols, w_row0, w_col0 = e.nt, cols, r0, 0
    else:        # dx = dy W: W (k x n) = rows r0 .. of the base: the nn set's contraction range r0 .. r0 + k
        tiled, w_cols, w_row0, w_col0 = e.nn, rows, 0, r0
    if tiled is None or (e.nt is None and e.nn is None):
        return None
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if _X3_CHECK and accumulate:
        global _X3_CHECK_PREV
        _X3_CHECK_PREV = out.double().clone()
    global x3_calls
    x3_calls += 1
    st = _lib.load().coda_gemm_x3_nt_f32(m, n, k, a.data_ptr(), a.stride(0), tiled.data_ptr(), w_cols, w_row0, w_col0,
                                         out.data_ptr(), out.stride(0), bias.data_ptr() if bias is not None else None,
                                         1 if accumulate else 0, _lib.current_stream_handle())
    if st == _lib.CODA_ENOSPC:
        return None
    if st != 0:
        raise RuntimeError(f"coda_gemm_x3_nt_f32 failed ({st}) for transb={transb} m={m} n={n} k={k}")
    if _X3_CHECK:  # development: every x3 product against float64, next to the library's fp32 result
        b2 = (b.t() if transb else b).double()
        ref = a.double() @ b2
        if bias is not None:
            ref = ref + bias.double()
        lib_out = torch.mm(a, b.t() if transb else b)
        if bias is not None:
            lib_out = lib_out + bias
        got = out.double() - (_X3_CHECK_PREV if accumulate else 0.0)
        den = float(ref.abs().max()) + 1e-300
        e_x3 = float((got - ref).abs().max()) / den
        e_lib = float((lib_out.double() - ref).abs().max()) / den
        cs = ref.sum(0)
        cden = float(cs.abs().max()) + 1e-300
        c_x3 = float((got.sum(0) - cs).abs().max()) / cden
        c_lib = float((lib_out.double().sum(0) - cs).abs().max()) / cden
        print(f"x3check tb={transb} acc={int(bool(accumulate))} m={m} n={n} k={k} lda={a.stride(0)} ldc={out.stride(0)}: "
              f"max err x3 {e_x3:.2e} lib {e_lib:.2e} | column sums x3 {c_x3:.2e} lib {c_lib:.2e}"
              + ("   <<<<" if e_x3 > 3 * e_lib + 1e-7 or c_x3 > 3 * c_lib + 1e-7 else ""), flush=True)
    return out


# Weight gradients through coda_gemm_x3_tn_f32: OFF by default.  Stand-alone the kernel is 1.2-1.55x the library path
# (row chunks as a batched GEMM + a sum) and 3-6x closer to float64, but inside the step it does not pay: same-box A/B
# 555-558 scenes/s with it against 559-564 without (two rounds each) -- it needs 32-64 token slices to fill the chip where
# the library's chunks need 8, so the grouped reduction over the partial sums costs 0.235 instead of 0.087 ms per step
# and 16 MB of partials per 256 x 256 gradient go through HBM.  CODA_GEMM_X3_TN=1 switches it on (tests do).
_X3_TN = os.environ.get("CODA_GEMM_X3_TN", "0") == "1"


def x3_tn_partials(dy, x, out=None):
    """dy (T, Co), x (T, Ci) -> partial weight gradients (slices, Co, Ci) with sum over dim 0 = dy^T x, through
    coda_gemm_x3_tn_f32 (both operands split in the kernel, transposing LDS reads); None when the shape is not the
    kernel's (the caller then takes the library path: row chunks as a batched GEMM).  ``out``: a (slices, Co, Ci)
    buffer to fill -- its first dimension then decides the slice count."""
    if not (_X3 and _X3_TN) or not _plain(dy, x) or dy.stride(1) != 1 or x.stride(1) != 1:
        return None
    t, co = dy.shape
    ci = x.shape[1]
    if t < 8192 or t % 32 or co % 128 or ci % 128 or (dy.stride(0) | x.stride(0)) % 4 or (dy.data_ptr() | x.data_ptr()) % 16:
        return None
    tiles = (co // 128) * (ci // 128)
    if out is not None:
        slices = out.shape[0]
        if t % slices or (t // slices) % 32 or not out.is_contiguous():
            return None
    else:
        slices = 1  # enough (slice, tile) items for every CU, slices of at least 256 tokens
        while tiles * slices < 256 and (t // (2 * slices)) % 32 == 0 and t // (2 * slices) >= 256:
            slices *= 2
        out = torch.empty((slices, co, ci), dtype=torch.float32, device=dy.device)
    st = _lib.load().coda_gemm_x3_tn_f32(t, co, ci, dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0),
                                         out.data_ptr(), slices, _lib.current_stream_handle())
    if st == _lib.CODA_ENOSPC:
        return None
    if st != 0:
        raise RuntimeError(f"coda_gemm_x3_tn_f32 failed ({st}) for rows={t} m={co} n={ci} slices={slices}")
    if route_log is not None:
        key = ("x3-tn", 1, 0, co, ci, t)
        route_log[key] = route_log.get(key, 0) + 1
    return out


route_log = None      # tools/gemm_routes.py: a dict (route, transa, transb, m, n, k) -> calls, filled while not None


def _run(transa, transb, m, n, k, a, b, out, bias, accumulate):
    if _X3:
        r = _x3_route(transa, transb, m, n, k, a, b, out, bias, accumulate)
        if r is not None:
            if route_log is not None:
                key = ("x3", transa, transb, m, n, k)
                route_log[key] = route_log.get(key, 0) + 1
            return r
    if route_log is not None:
        key = ("lib/own", transa, transb, m, n, k)
        route_log[key] = route_log.get(key, 0) + 1
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if _OWN_SMALL and not transa and m * n <= _OWN_MAX_MN and m % 64 == 0 and n % 64 == 0 and k % 128 == 0:
        st = _lib.load().coda_sgemm_f32(transb, m, n, k, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                        out.data_ptr(), out.stride(0), bias.data_ptr() if bias is not None else None,
                                        1 if accumulate else 0, _lib.current_stream_handle())
        if st == 0:
            return out
        if st != _lib.CODA_ENOSPC:  # ENOSPC = "not this kernel's shape / alignment": library path below
            raise RuntimeError(f"coda_sgemm_f32 failed ({st}) for transb={transb} m={m} n={n} k={k}")
    st = _lib.load().coda_gemm_f32(transa, transb, m, n, k, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                   out.data_ptr(), out.stride(0), bias.data_ptr() if bias is not None else None,
                                   1 if accumulate else 0, _lib.current_stream_handle())
    if -4000 < st <= -3000:
        # hipBLASLt could not PLAN this problem (-(3000 + hipblasStatus): no heuristic result, odd leading
        # dimensions): PyTorch's own GEMM path for this shape.  Execution failures (-(2000 + status)) and
        # everything else are errors and raise below.
        a2 = a.t() if transa else a
        b2 = b.t() if transb else b
        ref = torch.mm(a2, b2) if bias is None else torch.addmm(bias, a2, b2)
        return out.add_(ref) if accumulate else out.copy_(ref)
    if st != 0:
        raise RuntimeError(f"coda_gemm_f32 failed ({st}) for transa={transa} transb={transb} m={m} n={n} k={k}")
    return out


def linear(x, w, bias=None, out=None):
    """x (M,K), w (N,K), bias (N,) or None -> x @ w.T + bias, (M,N)."""
    if not _plain(x, w) or x.shape[0] == 0:
        r = torch.mm(x, w.t()) if bias is None else torch.addmm(bias, x, w.t())
        return r if out is None else out.copy_(r)
    x, w = _rows(x), _rows(w)
    if bias is not None and bias.stride(0) != 1:
        bias = bias.contiguous()
    return _run(0, 1, x.shape[0], w.shape[0], x.shape[1], x, w, out, bias, False)


def mm(a, b, out=None, accumulate=False):
    """a (M,K), b (K,N) -> a @ b; ``accumulate`` adds to ``out`` instead of overwriting it."""
    assert out is not None or not accumulate
    if not _plain(a, b) or a.shape[0] == 0:
        if accumulate:
            return out.addmm_(a, b)
        return torch.mm(a, b) if out is None else torch.mm(a, b, out=out)
    a, b = _rows(a), _rows(b)
    return _run(0, 0, a.shape[0], b.shape[1], a.shape[1], a, b, out, None, accumulate)


def mm_tn(a, b, out=None, accumulate=False):
    """a (K,M), b (K,N) -> a.T @ b, (M,N); ``accumulate`` adds to ``out`` instead of overwriting it."""
    assert out is not None or not accumulate
    if not _plain(a, b) or a.shape[0] == 0:
        r = torch.mm(a.t(), b)
        if out is None:
            return r
        return out.add_(r) if accumulate else out.copy_(r)
    a, b = _rows(a), _rows(b)
    return _run(1, 0, a.shape[1], b.shape[1], a.shape[0], a, b, out, None, accumulate)


class DeferredWeightGrads:
    """Collects weight-gradient products ``out = dy^T x`` during a hand-written backward and issues them in one
    launch (``coda_grouped_gemm_tn_f32``) when ``flush()`` is called: nothing inside a backward pass reads a
    weight gradient, and 72 launch-sized products of the decoder stack fill the chip only together.  Shapes the
    grouped kernel does not take (columns not multiples of 64, rows not a multiple of 8, non-unit inner strides)
    are computed on the spot with ``mm_tn``."""

    def __init__(self, sums_only=False):
        # sums_only: long reductions (the encoder's 16 384 tokens, the heads' 2048 x 6) -- the products are issued on
        # the spot as ONE batched library GEMM over row chunks (the grouped kernel is built for launch-sized
        # problems), and only the sum over the chunks joins the grouped reduction at flush()
        self.sums_only = sums_only
        self._rows = []   # 8 int64 per problem: the CodaTnProblem layout (include/coda_gemm.h)
        self._sums = []   # 3 int64 per open column-sum reduction: the CodaColsumItem layout (coda_token_ops.h)
        self._keep = []   # operands stay referenced until the launch has been enqueued

    def add_colsum(self, partials, out, blocks, n, groups=1):
        """Close ``out (groups, n) = sum over blocks of partials (groups, blocks, n)`` at flush time (the bias /
        LayerNorm gradient reductions of the blocks: one grouped launch instead of one per block)."""
        self._sums.append((partials.data_ptr(), out.data_ptr(), blocks | (n << 32), groups))
        self._keep.append((partials, out))

    def add_split(self, out, dy, x, chunk=2048, min_rows=4096):
        """out (Co, Ci) <- dy^T x with the row reduction split into `chunk`-row pieces: the batched product now, the
        sum over the pieces at flush() (one grouped launch for all sums of a node instead of one torch.sum each)."""
        p = dy.shape[0]
        if out.is_contiguous() and p >= min_rows:
            part = x3_tn_partials(dy, x)
            if part is not None:
                self.add_colsum(part, out, part.shape[0], out.numel())
                return
        if (p >= min_rows and p % chunk == 0 and out.is_contiguous() and dy.is_contiguous() and x.is_contiguous()
                and dy.dtype == torch.float32 and dy.is_cuda):
            nc = p // chunk
            part = torch.bmm(dy.view(nc, chunk, -1).transpose(1, 2), x.view(nc, chunk, -1))
            self.add_colsum(part, out, nc, out.numel())
        else:
            mm_tn(dy, x, out=out)

    def add(self, out, dy, x):
        if self.sums_only:
            return self.add_split(out, dy, x)
        rows, m = dy.shape
        n = x.shape[1]
        if (GROUPED_TN and rows % 8 == 0 and m % 64 == 0 and n % 64 == 0 and dy.stride(1) == 1 and x.stride(1) == 1
                and out.stride(1) == 1 and dy.dtype == torch.float32 and dy.is_cuda
                and (dy.stride(0) | x.stride(0) | out.stride(0)) % 2 == 0
                and (dy.data_ptr() | x.data_ptr() | out.data_ptr()) % 8 == 0):  # the kernel's 8-byte accesses
            self._rows.append((dy.data_ptr(), x.data_ptr(), out.data_ptr(), rows | (m << 32), n, dy.stride(0),
                               x.stride(0), out.stride(0)))
            self._keep.append((dy, x, out))
        else:
            mm_tn(dy, x, out=out)

    def flush(self):
        import numpy as np
        # (the launch's grid is sized by its widest item: bias-sized and weight-sized reductions go separately)
        for sums in ([t for t in self._sums if (t[2] >> 32) <= 4096], [t for t in self._sums if (t[2] >> 32) > 4096]):
            if sums:
                table = np.array(sums, dtype=np.int64)
                st = _lib.load().coda_tok_colsum_finalize_grouped_f32(table.ctypes.data, len(sums),
                                                                     _lib.current_stream_handle())
                _lib.check(st, "coda_tok_colsum_finalize_grouped_f32")
        if self._rows:
            table = np.array(self._rows, dtype=np.int64)
            st = _lib.load().coda_grouped_gemm_tn_f32(table.ctypes.data, len(self._rows), _lib.current_stream_handle())
            _lib.check(st, "coda_grouped_gemm_tn_f32")
        self._rows, self._sums, self._keep = [], [], []


GROUPED_TN = os.environ.get("CODA_GROUPED_TN", "1") != "0"
# the encoder layers' and the heads' split-K sums and column sums closed by one grouped launch per node (A/B switch)
DEFER_SUMS = GROUPED_TN and os.environ.get("CODA_DEFER_SUMS", "1") != "0"
